"""Seeded synthetic scenes of SURVEY.md section 8(d) (no datasets, no network).

Every generator returns a dict of float32 numpy arrays named like the reference's tensors plus the camera
fields of cameras.make_camera(); the same dict feeds the oracle (oracle/oracle.py) and, moved to the GPU, the
rasterizer bindings.
"""
import math

import numpy as np

from . import cameras


def _normalize(v, axis=-1):
    return v / np.maximum(np.linalg.norm(v, axis=axis, keepdims=True), 1e-12)


def quat_from_frame(n, rng):
    """Quaternions (r,x,y,z) of rotations whose local z axis is `n` (in-plane angle random)."""
    n = _normalize(n.astype(np.float64))
    a = np.where(np.abs(n[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    t0 = _normalize(np.cross(a, n))
    t1 = np.cross(n, t0)
    ang = rng.uniform(0, 2 * np.pi, size=(n.shape[0], 1))
    u = np.cos(ang) * t0 + np.sin(ang) * t1
    v = np.cross(n, u)
    R = np.stack([u, v, n], axis=-1)  # columns
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    q = np.zeros((n.shape[0], 4))
    # robust matrix -> quaternion
    for i in range(n.shape[0]):
        m = R[i]
        t = tr[i]
        if t > 0:
            s = math.sqrt(t + 1.0) * 2
            q[i] = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
        elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
            s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
            q[i] = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
        elif m[1, 1] > m[2, 2]:
            s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
            q[i] = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
        else:
            s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
            q[i] = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    return _normalize(q).astype(np.float32)


def _quat_from_frame_fast(n, rng):
    """Vectorised variant for large P: q = q_align(z->n) * q_spin(z, ang)."""
    n = _normalize(n.astype(np.float64))
    # shortest-arc quaternion from +z to n
    w = 1.0 + n[:, 2]
    xyz = np.stack([-n[:, 1], n[:, 0], np.zeros_like(w)], axis=-1)  # cross(z, n)
    flip = w < 1e-6
    w = np.where(flip, 0.0, w)
    xyz[flip] = np.array([1.0, 0.0, 0.0])
    qa = _normalize(np.concatenate([w[:, None], xyz], axis=-1))
    ang = rng.uniform(0, 2 * np.pi, size=n.shape[0])
    qs = np.stack([np.cos(ang / 2), np.zeros_like(ang), np.zeros_like(ang), np.sin(ang / 2)], axis=-1)
    # Hamilton product qa * qs
    r1, x1, y1, z1 = qa.T
    r2, x2, y2, z2 = qs.T
    q = np.stack([r1 * r2 - x1 * x2 - y1 * y2 - z1 * z2, r1 * x2 + x1 * r2 + y1 * z2 - z1 * y2,
                  r1 * y2 - x1 * z2 + y1 * r2 + z1 * x2, r1 * z2 + x1 * y2 - y1 * x2 + z1 * r2], axis=-1)
    return _normalize(q).astype(np.float32)


def random_cloud(P=10000, W=256, H=256, seed=0, sh_degree=0, variant="svgss", S=0, VS=0):
    """cfg1 "plumbing": uniform cloud, SH degree 0, no BRDF channels, black background."""
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    sc = {
        "means3D": rng.uniform(-1, 1, size=(P, 3)).astype(np.float32),
        "scales": np.exp(rng.uniform(math.log(0.005), math.log(0.05), size=(P, 3))).astype(np.float32),
        "rotations": _normalize(rng.normal(size=(P, 4))).astype(np.float32),
        "opacities": rng.uniform(0.05, 0.99, size=(P, 1)).astype(np.float32),
        "shs": rng.normal(0, 0.5, size=(P, M, 3)).astype(np.float32),
        "sh_degree": sh_degree,
        "bg": np.zeros(3, dtype=np.float32),
        "config": np.array([1.0, 1.0, 1.0], dtype=np.float32),
        "scale_modifier": 1.0,
    }
    sc["features"] = rng.normal(size=(P, S)).astype(np.float32)
    if variant == "svgss":
        sc["vfeatures"] = rng.normal(size=(P, VS)).astype(np.float32)
    sc.update(cameras.make_camera(W, H, cameras.orbit_eye(4.0, 30.0, 20.0)))
    sc["backward_geometry"] = True
    sc["computer_pseudo_normal"] = False
    return sc


def _surface_points(P, rng):
    """Noisy unit sphere + axis-aligned box mixture; returns (points, outward normals)."""
    n_s = P // 2
    n_b = P - n_s
    d = _normalize(rng.normal(size=(n_s, 3)))
    ps = d * (0.8 + 0.01 * rng.normal(size=(n_s, 1)))
    ns = d
    face = rng.integers(0, 6, size=n_b)
    uv = rng.uniform(-0.45, 0.45, size=(n_b, 2))
    pb = np.zeros((n_b, 3))
    nb = np.zeros((n_b, 3))
    ax = face // 2
    sgn = np.where(face % 2 == 0, 1.0, -1.0)
    for a in range(3):
        m = ax == a
        o = [i for i in range(3) if i != a]
        pb[m, a] = sgn[m] * 0.45
        pb[m, o[0]] = uv[m, 0]
        pb[m, o[1]] = uv[m, 1]
        nb[m, a] = sgn[m]
    pb += np.array([0.0, 0.0, -0.1])
    pts = np.concatenate([ps, pb], axis=0)
    nrm = np.concatenate([ns, nb], axis=0)
    perm = rng.permutation(P)
    return pts[perm], nrm[perm]


def surface_scene(P=200000, W=800, H=800, seed=1, sh_degree=3, variant="rgss", S=5, VS=0, bg=1.0,
                  azimuth=30.0, elevation=25.0, scale_lo=0.004, scale_hi=0.03):
    """cfg2/cfg3/cfg4/cfg5 geometry: surface-aligned surfels (local z = outward normal +- 20 deg jitter)."""
    rng = np.random.default_rng(seed)
    pts, nrm = _surface_points(P, rng)
    jit = nrm + math.tan(math.radians(20.0)) * 0.5 * rng.normal(size=nrm.shape)
    M = (sh_degree + 1) ** 2
    sc = {
        "means3D": pts.astype(np.float32),
        "scales": np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), size=(P, 3))).astype(np.float32),
        "rotations": _quat_from_frame_fast(jit, rng),
        "opacities": rng.beta(2.0, 1.0, size=(P, 1)).astype(np.float32),
        "shs": rng.normal(0, 0.3, size=(P, M, 3)).astype(np.float32),
        "sh_degree": sh_degree,
        "bg": np.full(3, bg, dtype=np.float32),
        "config": np.array([1.0, 1.0, 1.0], dtype=np.float32),
        "scale_modifier": 1.0,
        "backward_geometry": True,
        "computer_pseudo_normal": False,
    }
    sc.update(cameras.make_camera(W, H, cameras.orbit_eye(4.0, azimuth, elevation)))
    if variant == "rgss":
        # render.py:83-91: features = [geo normal (world), view depth, depth^2]
        xyz1 = np.concatenate([sc["means3D"], np.ones((P, 1), np.float32)], axis=-1)
        dep = (xyz1 @ sc["viewmatrix"])[:, 2:3]
        f = np.concatenate([_normalize(jit).astype(np.float32), dep, dep * dep], axis=-1).astype(np.float32)
        if S != 5:
            f = rng.normal(size=(P, S)).astype(np.float32)
        sc["features"] = f
    else:
        sc["features"] = rng.uniform(0, 1, size=(P, S)).astype(np.float32)
        sc["vfeatures"] = rng.uniform(0, 1, size=(P, VS)).astype(np.float32)
    return sc


def upstream_grads(sc, variant, seed=101):
    """dL/d(outputs) ~ N(0,1)/(H*W) for every differentiable output."""
    rng = np.random.default_rng(seed)
    H, W = sc["H"], sc["W"]
    S = sc["features"].shape[1]
    g = {k: (rng.normal(size=(c, H, W)) / (H * W)).astype(np.float32)
         for k, c in (("color", 3), ("normal", 3), ("depth", 1), ("opacity", 1), ("feature", S))}
    if variant == "svgss":
        g["vfeature"] = (rng.normal(size=(sc["vfeatures"].shape[1] // 4, H, W)) / (H * W)).astype(np.float32)
    return g


CONFIGS = {
    # name: (generator, kwargs)  -- BASELINE.json configs[0..4]
    "cfg1": (random_cloud, dict(P=10000, W=256, H=256, seed=0, sh_degree=0, variant="svgss", S=0, VS=0)),
    "cfg2": (surface_scene, dict(P=200000, W=800, H=800, seed=1, sh_degree=3, variant="rgss", S=5, VS=0, bg=1.0)),
    "cfg3_train": (surface_scene, dict(P=200000, W=800, H=800, seed=2, sh_degree=3, variant="svgss", S=4, VS=52, bg=1.0)),
    "cfg3_eval": (surface_scene, dict(P=200000, W=800, H=800, seed=2, sh_degree=3, variant="svgss", S=7, VS=64, bg=1.0)),
    "cfg4": (surface_scene, dict(P=300000, W=800, H=800, seed=3, sh_degree=3, variant="svgss", S=7, VS=64, bg=1.0)),
    "cfg5": (surface_scene, dict(P=2000000, W=1600, H=1600, seed=4, sh_degree=3, variant="svgss", S=7, VS=64, bg=1.0,
                                 scale_lo=0.002, scale_hi=0.012)),
    # the same 2 M-surfel stress scene at the generator's default surfel scales (0.004-0.03, as in cfg2-cfg4): ~4x the instances
    # (R = 18.3 M), the size SURVEY 8(a) a1 budgets the state blobs for -- the configuration where the composite kernels are nearest
    # to the HBM roof
    "cfg5_dense": (surface_scene, dict(P=2000000, W=1600, H=1600, seed=4, sh_degree=3, variant="svgss", S=7, VS=64, bg=1.0)),
}


def make(name, **override):
    gen, kw = CONFIGS[name]
    kw = dict(kw)
    kw.update(override)
    return gen(**kw)
