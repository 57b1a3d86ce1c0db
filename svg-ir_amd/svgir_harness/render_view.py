"""End-to-end svgss view: shading -> packing -> rasterizer -> unpacking, the sequence of the reference's
`render_view` (gaussian_renderer/svgss.py:51-262) with explicit tensors instead of its scene / camera classes.

Only the parts that belong to the hot path and its immediate callers are reproduced: the SV-BRDF shading and packing
(svgss.py:125-166, fused: gaussian_renderer/shading.py), the rasterizer call (:170-182) and the image-space unpacking
(:188-246: division by the rendered opacity, channel split, sRGB, compositing over the background).  The
`depth2normal` pseudo normal and the environment backdrop of the eval branch need the reference's camera class and
are left to the caller."""
import ctypes as C

import torch

from gaussian_renderer import _native as N
from gaussian_renderer import shading
from gaussian_renderer.svgss_rasterization import GaussianRasterizer

from . import runner

N.lib.svgir_unpack_planes.restype = C.c_int
N.lib.svgir_unpack_planes.argtypes = [C.c_int32]
N.lib.svgir_unpack_forward.restype = C.c_int
N.lib.svgir_unpack_forward.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 6
N.lib.svgir_unpack_backward.restype = C.c_int
N.lib.svgir_unpack_backward.argtypes = [C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 9
N.lib.svgir_depth2normal.restype = C.c_int
N.lib.svgir_depth2normal.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p]

for _n, _k in (("svgir_pack_rgss_forward", 5), ("svgir_pack_rgss_backward", 6)):
    getattr(N.lib, _n).restype = C.c_int
    getattr(N.lib, _n).argtypes = [C.c_int32] + [C.c_void_p] * _k
for _n, _k in (("svgir_unpack_rgss_forward", 6), ("svgir_unpack_rgss_backward", 9)):
    getattr(N.lib, _n).restype = C.c_int
    getattr(N.lib, _n).argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * _k

TRAIN_PLANES = ("pbr", "normal", "base_color", "roughness", "diffuse", "local_lights", "visibility")
EVAL_PLANES = ("pbr", "normal", "base_color", "roughness", "direct", "indirect", "lights", "local_lights", "visibility")


class _Unpack(torch.autograd.Function):
    """svgir_unpack_forward / svgir_unpack_backward (csrc/epilogue.hip): one pass over the rasterizer's planes."""

    @staticmethod
    def forward(ctx, opacity, feature, vfeature, bg, training):
        dev = opacity.device
        if dev.type != "cuda":
            raise RuntimeError("unpack: tensors must live on the GPU (libsvgir_raster.so has no CPU path)")
        op, fe, vf, bgc = (N.f32c(t, dev) for t in (opacity, feature, vfeature, bg))
        H, W = op.shape[-2], op.shape[-1]
        S, VC = (4, 13) if training else (7, 16)
        if fe.shape[0] != S or vf.shape[0] != VC:
            raise RuntimeError(f"unpack: expected {S} feature / {VC} vfeature planes, got {fe.shape[0]} / {vf.shape[0]}")
        out = torch.empty((N.lib.svgir_unpack_planes(int(training)), H, W), dtype=torch.float32, device=dev)
        N.check(N.lib.svgir_unpack_forward(W, H, int(training), bgc.data_ptr(), op.data_ptr(), fe.data_ptr(), vf.data_ptr(),
                                           out.data_ptr(), N.stream_ptr(dev)), "unpack_forward")
        ctx.save_for_backward(op, fe, vf, bgc)
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, g_out):
        op, fe, vf, bgc = ctx.saved_tensors
        dev = op.device
        H, W = op.shape[-2], op.shape[-1]
        g = N.f32c(g_out, dev)
        d_op, d_fe, d_vf = torch.empty_like(op), torch.empty_like(fe), torch.empty_like(vf)
        N.check(N.lib.svgir_unpack_backward(W, H, int(ctx.training), bgc.data_ptr(), op.data_ptr(), fe.data_ptr(), vf.data_ptr(),
                                            g.data_ptr(), d_op.data_ptr(), d_fe.data_ptr(), d_vf.data_ptr(), N.stream_ptr(dev)),
                "unpack_backward")
        return d_op, d_fe, d_vf, None, None


class _PackRgss(torch.autograd.Function):
    """render.py:83-91: features = [geo normal, view depth, depth^2] (svgir_pack_rgss_forward / _backward)."""

    @staticmethod
    def forward(ctx, means3D, normals, viewmatrix):
        dev = means3D.device
        m3, nr, vm = (N.f32c(t, dev) for t in (means3D, normals, viewmatrix))
        out = torch.empty((m3.shape[0], 5), dtype=torch.float32, device=dev)
        N.check(N.lib.svgir_pack_rgss_forward(m3.shape[0], N.ptr(m3), N.ptr(nr), vm.data_ptr(), N.ptr(out), N.stream_ptr(dev)), "pack_rgss")
        ctx.save_for_backward(m3, vm)
        return out

    @staticmethod
    def backward(ctx, g):
        m3, vm = ctx.saved_tensors
        dev = m3.device
        g = N.f32c(g, dev)
        d_m, d_n = torch.empty_like(m3), torch.empty_like(m3)
        N.check(N.lib.svgir_pack_rgss_backward(m3.shape[0], N.ptr(m3), vm.data_ptr(), N.ptr(g), N.ptr(d_m), N.ptr(d_n), N.stream_ptr(dev)),
                "pack_rgss_backward")
        return d_m, d_n, None


class _UnpackRgss(torch.autograd.Function):
    """render.py:107-114 (svgir_unpack_rgss_forward / _backward): [normal3, depth, depth2, depth_var] planes."""

    @staticmethod
    def forward(ctx, num_contrib, opacity, depth, feature):
        dev = opacity.device
        nc = num_contrib.to(torch.int32).contiguous()
        op, de, fe = (N.f32c(t, dev) for t in (opacity, depth, feature))
        H, W = op.shape[-2], op.shape[-1]
        if fe.shape[0] != 5:
            raise RuntimeError("unpack_rgss: expected the 5 stage-1 feature planes [normal 3, depth, depth^2]")
        out = torch.empty((6, H, W), dtype=torch.float32, device=dev)
        N.check(N.lib.svgir_unpack_rgss_forward(W, H, nc.data_ptr(), op.data_ptr(), de.data_ptr(), fe.data_ptr(), out.data_ptr(),
                                                N.stream_ptr(dev)), "unpack_rgss")
        ctx.save_for_backward(nc, op, de, fe)
        return out

    @staticmethod
    def backward(ctx, g):
        nc, op, de, fe = ctx.saved_tensors
        dev = op.device
        H, W = op.shape[-2], op.shape[-1]
        g = N.f32c(g, dev)
        d_op, d_de, d_fe = torch.empty_like(op), torch.empty_like(de), torch.empty_like(fe)
        N.check(N.lib.svgir_unpack_rgss_backward(W, H, nc.data_ptr(), op.data_ptr(), de.data_ptr(), fe.data_ptr(), g.data_ptr(),
                                                 d_op.data_ptr(), d_de.data_ptr(), d_fe.data_ptr(), N.stream_ptr(dev)), "unpack_rgss_backward")
        return None, d_op, d_de, d_fe


def pack_rgss(means3D, geo_normals, viewmatrix):
    return _PackRgss.apply(means3D, geo_normals, viewmatrix)


def unpack_rgss(rendered):
    """render.py:107-135.  `rendered` = the rgss rasterizer's 11-tuple; returns the reference's result dict entries that are
    derived in image space."""
    (num_rendered, num_contrib, image, normal, opacity, depth, feature, pseudo_normal, surface_xyz, weights, radii) = rendered
    planes = _UnpackRgss.apply(num_contrib, opacity, depth, feature)
    return dict(render=image, opacity=opacity, depth=depth, depth_var=planes[5:6], normal=normal, feature_normal=planes[0:3],
                feature_depth=planes[3:4], surface_xyz=surface_xyz, visibility_filter=radii > 0, radii=radii,
                num_rendered=num_rendered, num_contrib=num_contrib, weights=weights)


N.lib.svgir_depth2normal_backward.restype = C.c_int
N.lib.svgir_depth2normal_backward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                             C.c_float, C.c_void_p, C.c_void_p]


class _Depth2Normal(torch.autograd.Function):
    """svgir_depth2normal / svgir_depth2normal_backward: differentiable w.r.t. the depth like the reference's function (the
    stage-1 loss uses the pseudo normal without detaching, gaussian_renderer/render.py:158-160)."""

    @staticmethod
    def forward(ctx, depth, mask, fovx, fovy, px, py):
        dev = depth.device
        d, m = N.f32c(depth.detach(), dev), N.f32c(mask.to(torch.float32), dev)
        H, W = d.shape[-2], d.shape[-1]
        out = N.out_tensor((3, H, W), torch.float32, dev)
        with torch.cuda.device(dev):
            N.check(N.lib.svgir_depth2normal(W, H, d.data_ptr(), m.data_ptr(), fovx, fovy, px, py, out.data_ptr(), N.stream_ptr(dev)),
                    "depth2normal")
        ctx.save_for_backward(d, m)
        ctx.cam = (fovx, fovy, px, py, tuple(depth.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        d, m = ctx.saved_tensors
        fovx, fovy, px, py, shape = ctx.cam
        dev = d.device
        H, W = d.shape[-2], d.shape[-1]
        gd = N.out_tensor((H, W), torch.float32, dev)
        with torch.cuda.device(dev):
            N.check(N.lib.svgir_depth2normal_backward(W, H, d.data_ptr(), m.data_ptr(), N.f32c(g, dev).data_ptr(), fovx, fovy, px, py,
                                                      gd.data_ptr(), N.stream_ptr(dev)), "depth2normal_backward")
        return gd.reshape(shape), None, None, None, None, None


def depth2normal(depth, mask, fovx, fovy, prcppoint=(0.5, 0.5)):
    """utils/image_utils.py:61-125 as one HIP kernel per direction: depth, mask [1,H,W] -> pseudo normal [3,H,W],
    differentiable w.r.t. `depth` (pass `depth.detach()` where the reference detaches, svgss.py:345)."""
    return _Depth2Normal.apply(depth, mask, float(fovx), float(fovy), float(prcppoint[0]), float(prcppoint[1]))


def unpack(rendered, bg_color, is_training):
    """svgss.py:188-246.  `rendered` = the rasterizer's 9-tuple; returns the result dict of the reference."""
    (num_rendered, image, normal, opacity, depth, feature, vfeature, weights, radii) = rendered
    planes = _Unpack.apply(opacity, feature, vfeature, bg_color, bool(is_training))
    res = {k: planes[3 * i:3 * i + 3] for i, k in enumerate(TRAIN_PLANES if is_training else EVAL_PLANES)}
    res.update(render=image, depth=depth, opacity=opacity, visibility_filter=radii > 0, radii=radii,
               num_rendered=num_rendered, weights=weights)
    return res


def render_svgss_view(sc, mat, light, is_training, fused=False):
    """sc: runner.to_torch() scene (geometry + camera); mat: dict with base_color [P,12], roughness [P,4], normals
    [P,4,3], viewdirs [P,3], radiance / dirs [P,Ns,3], visibility / areas [P,Ns,1]; light: DirectLightMap-like (.env).
    Returns (results dict, means2D gradient carrier).  fused=True: the shading runs inside the rasterizer calls, for the surfels
    the view reads only (shading.render_shaded); same results."""
    means2D = torch.zeros_like(sc["means3D"], requires_grad=torch.is_grad_enabled())
    if fused:
        rendered, _ = shading.render_shaded(runner.settings(sc, "svgss"), sc["means3D"], means2D, sc["opacities"], sc["shs"],
                                            sc["scales"], sc["rotations"], mat["base_color"], mat["roughness"], mat["normals"],
                                            mat["viewdirs"], mat["radiance"], light, mat["visibility"], mat["dirs"], mat["areas"],
                                            is_training)
        return unpack(rendered, sc["bg"], is_training), means2D
    feats, vfeats, _ = shading.shade_and_pack(mat["base_color"], mat["roughness"], mat["normals"], mat["viewdirs"],
                                              mat["radiance"], light, mat["visibility"], mat["dirs"], mat["areas"],
                                              sc["viewmatrix"], is_training)
    rast = GaussianRasterizer(runner.settings(sc, "svgss"))
    rendered = rast(means3D=sc["means3D"], means2D=means2D, opacities=sc["opacities"], shs=sc["shs"],
                    scales=sc["scales"], rotations=sc["rotations"], features=feats, vfeatures=vfeats)
    return unpack(rendered, sc["bg"], is_training), means2D
