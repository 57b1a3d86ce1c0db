"""CPU oracle of the callers either side of the rasterizer (numpy, fp64).

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); the product never imports it.

Restates, in plain per-element math,
  * incident-direction generation: `fibonacci_sphere_sampling` (utils/graphics_utils.py:9-37) with
    `rotation_between_z` (utils/sh_utils.py:36-68), as called by `sample_incident_rays` (scene/gaussian_model.py:23-31);
  * the 32x64 bilinear down-sample `EnvLight.direct_light` applies to its HDR map (scene/envmap.py:62-63,
    F.interpolate(mode='bilinear', align_corners=False));
  * the image-space epilogue of `render_view` (gaussian_renderer/svgss.py:187-262): division by the rendered opacity,
    channel split, sRGB (`rgb_to_srgb`, utils/graphics_utils.py:198-215), compositing over the background;
  * `depth2normal` (utils/image_utils.py:61-125);
  * `ssim` + `F.l1_loss` as called on the rendered image (utils/loss_utils.py:21-64; svgss.py:281-289, render.py:150-151).
Parity: PINNED -- tests/golden/{incident_dirs,lights,render_view}.npz hold inputs and outputs produced by running the
reference's own functions (and its whole `render_view` with a recording stub rasterizer) in the authoring container
(scripts/make_golden_view.py); tests/test_view_fixtures.py checks every function here against them.
"""
import math

import numpy as np


# ---- f1: incident directions -------------------------------------------------------------------------------------
def rotation_between_z(vec):
    """Rotation that takes +z to `vec` [n,3] (unit): R [n,3,3].  utils/sh_utils.py:36-68: Rodrigues form with axis
    z x vec = (-vy, vx, 0) and 1/(1 + cos) clamped at 1e-7; -I for vec.z + 1 <= 0."""
    v = np.asarray(vec, dtype=np.float64)
    v1, v2 = -v[:, 1], v[:, 0]
    c1 = np.maximum(v[:, 2] + 1.0, 1e-7)
    R = np.zeros((v.shape[0], 3, 3))
    R[:, 0, 0] = 1 + (-v2 * v2) / c1
    R[:, 0, 1] = v1 * v2 / c1
    R[:, 0, 2] = v2
    R[:, 1, 0] = v1 * v2 / c1
    R[:, 1, 1] = 1 + (-v1 * v1) / c1
    R[:, 1, 2] = -v1
    R[:, 2, 0] = -v2
    R[:, 2, 1] = v1
    R[:, 2, 2] = 1 + (-v2 * v2 - v1 * v1) / c1
    flip = ~(v[:, 2] + 1.0 > 0)
    R[flip] = -np.eye(3)
    return R


def fibonacci_dirs(normals, Ns, offsets=None):
    """utils/graphics_utils.py:9-37: hemisphere lattice around +z (z clamped at sin 10 deg), rotated to each normal and
    re-normalised; `offsets` [n,1] = the random azimuth offset of the training branch (None: evaluation lattice).
    Returns (dirs [n,Ns,3], areas [n,Ns,1] = 2 pi)."""
    n = np.asarray(normals, dtype=np.float64)
    # The reference evaluates the lattice in fp32; its azimuth delta * idx reaches ~900 rad at Ns = 384, where one fp32
    # ulp is 6e-5 rad.  The lattice parameters are therefore rounded to fp32 exactly where the reference rounds them
    # (product, sum with the offset, z, radius); everything downstream is exact arithmetic on those values.
    f32 = np.float32
    idx = np.arange(Ns, dtype=f32)[None]
    delta = f32(math.pi * (3.0 - math.sqrt(5.0)))
    z = np.maximum(f32(1) - f32(2) * idx / f32(2 * Ns - 1), f32(math.sin(10 / 180 * math.pi)))
    rad = np.sqrt(f32(1) - z * z)
    theta = delta * idx
    if offsets is not None:
        theta = np.asarray(offsets, dtype=f32).reshape(-1, 1) * f32(1) + theta
    theta = np.broadcast_to(theta, (n.shape[0], Ns)).astype(np.float64)
    z, rad = z.astype(np.float64), rad.astype(np.float64)
    local = np.stack([np.sin(theta) * rad, np.cos(theta) * rad, np.broadcast_to(z, theta.shape)], axis=-1)   # [n,Ns,3]
    d = np.einsum("nij,nsj->nsi", rotation_between_z(n), local)
    d = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-12)
    return d, np.full((n.shape[0], Ns, 1), 2 * math.pi)


# ---- f1: EnvLight's down-sample ----------------------------------------------------------------------------------
def resample_bilinear(img, out_h, out_w):
    """F.interpolate(mode='bilinear', align_corners=False) of an [H,W,C] image (scene/envmap.py:62-63): source
    coordinate = (dst + 0.5) * scale - 0.5 clamped at 0, neighbours clamped to the last row / column."""
    img = np.asarray(img, dtype=np.float64)
    H, W = img.shape[:2]

    def taps(n_out, n_in):
        s = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        s = np.maximum(s, 0.0)
        i0 = np.minimum(np.floor(s).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        return i0, i1, s - i0

    y0, y1, fy = taps(out_h, H)
    x0, x1, fx = taps(out_w, W)
    top = img[y0][:, x0] * (1 - fx)[None, :, None] + img[y0][:, x1] * fx[None, :, None]
    bot = img[y1][:, x0] * (1 - fx)[None, :, None] + img[y1][:, x1] * fx[None, :, None]
    return top * (1 - fy)[:, None, None] + bot * fy[:, None, None]


# ---- f2: image-space epilogue ------------------------------------------------------------------------------------
def rgb_to_srgb(img, clip=True):
    img = np.asarray(img, dtype=np.float64)
    out = np.where(img > 0.0031308, np.power(np.maximum(img, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * img)
    return np.clip(out, 0.0, 1.0) if clip else out


def unpack_svgss(opacity, feature, vfeature, bg, training):
    """svgss.py:187-246.  opacity [1,H,W], feature [S,H,W], vfeature [VS/4,H,W], bg [3] -> dict of images."""
    op = np.asarray(opacity, dtype=np.float64)
    f = np.asarray(feature, dtype=np.float64) / np.maximum(op, 1e-5)
    vf = np.asarray(vfeature, dtype=np.float64) / np.maximum(op, 1e-5)
    bgc = np.asarray(bg, dtype=np.float64)[:, None, None]

    def over(r):
        return r * op + (1 - op) * bgc

    res = {}
    if training:
        vis, local = f[0:1], f[1:4]
        res.update(local_lights=over(rgb_to_srgb(local)), visibility=over(vis))
        pbr, base, normal, rough, diffuse = vf[0:3], vf[3:6], vf[6:9], vf[9:10], vf[10:13]
        res.update(base_color=over(rgb_to_srgb(base)), diffuse=over(rgb_to_srgb(diffuse)), roughness=over(rough))
    else:
        light, local, vis = f[0:3], f[3:6], f[6:7]
        res.update(lights=over(rgb_to_srgb(light)), local_lights=over(rgb_to_srgb(local)), visibility=over(vis))
        pbr, base, normal, rough, direct, indirect = vf[0:3], vf[3:6], vf[6:9], vf[9:10], vf[10:13], vf[13:16]
        res.update(base_color=over(rgb_to_srgb(base)), direct=rgb_to_srgb(direct), indirect=rgb_to_srgb(indirect),
                   roughness=over(rough))
    res.update(pbr=rgb_to_srgb(over(pbr)), pbr_linear=pbr, normal=normal)
    return res


def depth2normal(depth, mask, fovx, fovy, prcppoint=(0.5, 0.5)):
    """utils/image_utils.py:61-125.  depth, mask [1,H,W] -> normal [3,H,W]: back-project every pixel with the pinhole
    intrinsics (NOTE the reference's K = diag(focal(FoVy, H), focal(FoVx, W)): x uses the y focal length and vice
    versa), take the four neighbour differences (replicate padding, masked), sum the four cross products, normalise."""
    d = np.asarray(depth, dtype=np.float64)[0]
    m = np.asarray(mask, dtype=np.float64)[0] != 0
    H, W = d.shape
    k00 = H / (2 * math.tan(fovy / 2))
    k11 = W / (2 * math.tan(fovx / 2))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    px = (xs - prcppoint[0] * W) * d / k00
    py = (ys - prcppoint[1] * H) * d / k11
    pos = np.stack([px, py, d], axis=-1)                                    # [H,W,3]
    p = np.pad(pos, ((1, 1), (1, 1), (0, 0)), mode="edge")
    mk = np.pad(m, ((1, 1), (1, 1)), mode="edge")[..., None]
    c = p[1:-1, 1:-1] * mk[1:-1, 1:-1]
    u = (p[:-2, 1:-1] - c) * mk[:-2, 1:-1]
    l_ = (p[1:-1, :-2] - c) * mk[1:-1, :-2]
    b = (p[2:, 1:-1] - c) * mk[2:, 1:-1]
    r = (p[1:-1, 2:] - c) * mk[1:-1, 2:]
    n = np.cross(u, l_) + np.cross(r, u) + np.cross(b, r) + np.cross(l_, b)
    n = n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
    return (n * mk[1:-1, 1:-1]).transpose(2, 0, 1)


def unpack_svgss_torch(opacity, feature, vfeature, bg, training):
    """The same as unpack_svgss in differentiable torch (any dtype / device): torch.autograd of this is the reference for
    the backward of the fused epilogue kernel."""
    import torch

    def srgb(img):
        out = torch.where(img > 0.0031308, torch.pow(torch.clamp(img, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * img)
        return out.clamp(0.0, 1.0)

    f = feature / opacity.clamp_min(1e-5)
    vf = vfeature / opacity.clamp_min(1e-5)
    bgc = bg[:, None, None]

    def over(r):
        return r * opacity + (1 - opacity) * bgc

    res = {}
    if training:
        vis, local = f.split([1, 3], dim=0)
        pbr, base, normal, rough, diffuse = vf.split([3, 3, 3, 1, 3], dim=0)
        res.update(local_lights=over(srgb(local)), visibility=over(vis), base_color=over(srgb(base)), diffuse=over(srgb(diffuse)),
                   roughness=over(rough))
    else:
        light, local, vis = f.split([3, 3, 1], dim=0)
        pbr, base, normal, rough, direct, indirect = vf.split([3, 3, 3, 1, 3, 3], dim=0)
        res.update(lights=over(srgb(light)), local_lights=over(srgb(local)), visibility=over(vis), base_color=over(srgb(base)),
                   direct=srgb(direct), indirect=srgb(indirect), roughness=over(rough))
    res.update(pbr=srgb(over(pbr)), normal=normal.expand(3, -1, -1))
    return res


# ---- f2: image losses --------------------------------------------------------------------------------------------
def ssim_window():
    """utils/loss_utils.py:21-23 `gaussian(11, 1.5)`: fp32 values, divided by their fp32 sum."""
    g = np.array([math.exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    return (g / g.sum(dtype=np.float32)).astype(np.float64)


def l1_ssim_torch(img1, img2):
    """(mean |img1 - img2|, mean SSIM map) in differentiable torch fp64 -- utils/loss_utils.py:44-61 with the 2-D window
    written out (outer product of the 1-D one), zero padding, one depthwise convolution per windowed moment."""
    import torch
    a, b = img1.to(torch.float64), img2.to(torch.float64)
    C = a.shape[0]
    g = torch.from_numpy(ssim_window()).to(a.device)
    w = (g[:, None] * g[None, :])[None, None].expand(C, 1, 11, 11)

    def blur(x):
        return torch.nn.functional.conv2d(x[None], w, padding=5, groups=C)[0]

    mu1, mu2 = blur(a), blur(b)
    s1, s2, s12 = blur(a * a) - mu1 * mu1, blur(b * b) - mu2 * mu2, blur(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return (a - b).abs().mean(), smap.mean()
