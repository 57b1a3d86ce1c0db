"""Generates tests/golden/*.npz by importing the REFERENCE's importable Python helpers from /root/reference.

Runs only in the authoring container (the reference tree is absent on the GPU box); the fixtures are data
(inputs + expected outputs), no reference source is copied.  Heavy / CUDA-only reference dependencies are
replaced by inert stubs just so that the modules import.

    python scripts/make_golden.py
"""
import importlib
import importlib.abc
import importlib.machinery
import inspect
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

STUBS = {"plyfile", "simple_knn", "custom_knn", "kornia", "torchvision", "cv2", "imageio", "nvdiffrast", "slangtorch",
         "pyexr", "submodules", "pbgi", "svgss_rasterization", "rgss_rasterization", "lpipsPyTorch", "dearpygui",
         "tqdm", "PIL", "matplotlib", "skimage", "scipy", "open3d", "trimesh", "bvh"}


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        m = _Stub(self.__name__ + "." + name)
        sys.modules[m.__name__] = m
        return m

    def __call__(self, *a, **k):
        return _Stub("stubcall")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUBS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        return _Stub(spec.name)

    def exec_module(self, module):
        module.__path__ = []


def main():
    os.makedirs(OUT, exist_ok=True)
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, REF)
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: _Stub("_C")
    rng = np.random.default_rng(1234)
    if "--shading-only" in sys.argv:   # (regenerates tests/golden/shading.npz alone; the other archives stay byte-identical)
        shading_fixtures()
        return

    # ---- SH evaluation (utils/sh_utils.py:71-128 as used at gaussian_renderer/svgss.py:92-96) ----
    from utils.sh_utils import eval_sh
    n = 257
    pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    campos = np.array([0.3, -2.5, 1.1], dtype=np.float32)
    sh = rng.normal(0, 0.6, size=(n, 16, 3)).astype(np.float32)
    d = torch.from_numpy(pos) - torch.from_numpy(campos)[None]
    d = d / d.norm(dim=-1, keepdim=True)
    cols = {}
    for deg in range(4):
        shs_view = torch.from_numpy(sh).transpose(1, 2).reshape(-1, 3, 16)
        rgb = torch.clamp_min(eval_sh(deg, shs_view, d) + 0.5, 0.0)
        cols[f"rgb_deg{deg}"] = rgb.numpy().astype(np.float32)
    np.savez(os.path.join(OUT, "sh_eval.npz"), pos=pos, campos=campos, sh=sh, **cols)

    # ---- quaternion -> rotation (utils/general_utils.py:231-239) ----
    from utils.general_utils import quaternion2rotmat
    q = rng.normal(size=(129, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    Rm = quaternion2rotmat(torch.from_numpy(q)).numpy().astype(np.float32)
    np.savez(os.path.join(OUT, "quat_rot.npz"), q=q, R=Rm)

    # ---- camera matrices (utils/graphics_utils.py:136-168, scene/cameras.py:69-80) ----
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    cams = {}
    for i in range(4):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        t = rng.normal(size=3)
        fovx, fovy = float(rng.uniform(0.4, 1.2)), float(rng.uniform(0.4, 1.2))
        w2c = getWorld2View2(Q, t)
        wvt = torch.tensor(w2c).transpose(0, 1)
        P = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy)
        full = (wvt.unsqueeze(0).bmm(P.transpose(0, 1).unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cams[f"R{i}"], cams[f"t{i}"] = Q.astype(np.float64), t.astype(np.float64)
        cams[f"fov{i}"] = np.array([fovx, fovy])
        cams[f"w2c{i}"], cams[f"P{i}"] = np.asarray(w2c, dtype=np.float32), P.numpy().astype(np.float32)
        cams[f"full{i}"], cams[f"center{i}"] = full.numpy().astype(np.float32), center.numpy().astype(np.float32)
    np.savez(os.path.join(OUT, "cameras.npz"), **cams)

    # ---- binding API surface (gaussian_renderer/{svgss,rgss}_rasterization.py) ----
    api = {}
    for mod in ("svgss_rasterization", "rgss_rasterization"):
        m = importlib.import_module("gaussian_renderer." + mod)
        api[mod + ".settings_fields"] = np.array(m.GaussianRasterizationSettings._fields)
        api[mod + ".forward_args"] = np.array(list(inspect.signature(m.GaussianRasterizer.forward).parameters))
        api[mod + ".function_forward_args"] = np.array(list(inspect.signature(m._RasterizeGaussians.forward).parameters))
        api[mod + ".function_backward_args"] = np.array(list(inspect.signature(m._RasterizeGaussians.backward).parameters))
        api[mod + ".public"] = np.array(sorted(k for k in vars(m) if k in (
            "GaussianRasterizationSettings", "GaussianRasterizer", "_RasterizeGaussians", "rasterize_gaussians", "_C",
            "cpu_deep_copy_tuple")))
    np.savez(os.path.join(OUT, "binding_api.npz"), **api)
    shading_fixtures()
    print("wrote", sorted(os.listdir(OUT)))



def shading_fixtures():
    """Fixtures for the per-splat SV-BRDF shading (gaussian_renderer/svgss.py:537-631, the REFERENCE's own
    rendering_equation4 / GGX_specular4 imported and run on CPU, forward values and autograd gradients)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from gaussian_renderer.svgss import rendering_equation4

    class EnvStandIn:
        """scene/direct_light_map.py:70-83,103-106 needs .cuda() in __init__; its direct_light() is 14 lines of
        torch ops (softplus + F.grid_sample, x2) which are reproduced here only to FEED the reference function;
        the lookup itself is pinned against F.grid_sample."""

        def __init__(self, env):
            self.env = env

        def direct_light(self, dirs):
            shape = dirs.shape
            dirs = dirs.reshape(-1, 3)
            envir_map = F.softplus(self.env).permute(0, 3, 1, 2)
            phi = torch.arccos(dirs[:, 2]).reshape(-1) - 1e-6
            theta = torch.atan2(dirs[:, 1], dirs[:, 0]).reshape(-1)
            query_y = (phi / np.pi) * 2 - 1
            query_x = -theta / np.pi
            grid = torch.stack((query_x, query_y)).permute(1, 0).unsqueeze(0).unsqueeze(0)
            light_rgbs = F.grid_sample(envir_map, grid, align_corners=True).squeeze().permute(1, 0).reshape(-1, 3)
            return light_rgbs.reshape(*shape) * 2.0

    out = {}
    # (a, b: the round-1 fixtures, unchanged draws; c: 200 surfels x 64 samples with a third of the corners at the glossy end of the
    # reference's roughness range, 0.09 .. 0.15, where fp32 is ill-conditioned)
    for tag, n, Ns, seed in (("a", 48, 8, 0), ("b", 33, 64, 1), ("c", 200, 64, 2)):
        g = torch.Generator().manual_seed(seed)
        rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
        base = (torch.sigmoid(rnd(n, 12)) * 0.77 + 0.03).requires_grad_(True)
        rough = torch.sigmoid(rnd(n, 4)) * 0.9 + 0.09
        if tag == "c":
            glossy = torch.rand(n, 4, generator=g, dtype=torch.float64) < 0.33
            rough = torch.where(glossy, 0.09 + 0.06 * torch.rand(n, 4, generator=g, dtype=torch.float64), rough)
        rough = rough.requires_grad_(True)
        nrm0 = F.normalize(rnd(n, 1, 3), dim=-1)
        normals = (nrm0 + 0.1 * rnd(n, 4, 3)).requires_grad_(True)
        viewdirs = F.normalize(nrm0[:, 0] + 0.6 * rnd(n, 3), dim=-1)
        dirs = F.normalize(nrm0 + 0.9 * rnd(n, Ns, 3), dim=-1)
        areas = torch.full((n, Ns, 1), 2 * np.pi, dtype=torch.float64)
        vis = (torch.rand(n, Ns, 1, generator=g, dtype=torch.float64) > 0.3).to(torch.float64)
        radiance = (0.2 * rnd(n, Ns, 3)).abs().requires_grad_(True)
        env = (3.0 * torch.rand(1, 16, 32, 3, generator=g, dtype=torch.float64)).requires_grad_(True)
        light = EnvStandIn(env)
        pbr, ex = rendering_equation4(base, rough, normals, viewdirs, radiance, light, visibility_precompute=vis,
                                      incident_dirs_precompute=dirs, incident_areas_precompute=areas)
        wts = {k: rnd(*ex[k].shape) for k in ("diffuse_light", "specular", "direct", "indirect")}
        w_pbr, w_inc, w_glob = rnd(*pbr.shape), rnd(n, 3), rnd(n, 3)
        loss = (pbr * w_pbr).sum() + sum((ex[k] * wts[k]).sum() for k in wts) \
            + (ex["incident_lights"].mean(-2) * w_inc).sum() + (ex["global_incident_lights"].mean(-2) * w_glob).sum()
        grads = torch.autograd.grad(loss, [base, rough, normals, radiance, env])
        pre = "shade_" + tag + "_"
        for k, v in dict(base=base, rough=rough, normals=normals, viewdirs=viewdirs, dirs=dirs, areas=areas, vis=vis,
                         radiance=radiance, env=env, pbr=pbr, w_pbr=w_pbr, w_inc=w_inc, w_glob=w_glob,
                         env_lookup=light.direct_light(dirs)).items():
            out[pre + k] = v.detach().numpy()
        for k in ("diffuse_light", "specular", "direct", "indirect"):
            out[pre + k] = ex[k].detach().numpy()
            out[pre + "w_" + k] = wts[k].numpy()
        out[pre + "mean_incident"] = ex["incident_lights"].mean(-2).detach().numpy()
        out[pre + "mean_global"] = ex["global_incident_lights"].mean(-2).detach().numpy()
        out[pre + "mean_local"] = ex["local_incident_lights"].mean(-2).detach().numpy()
        out[pre + "mean_vis"] = ex["incident_visibility"].mean(-2).detach().numpy()
        for k, gv in zip(("base", "rough", "normals", "radiance", "env"), grads):
            out[pre + "g_" + k] = gv.numpy()
        # The SAME reference function on the SAME inputs in fp32 -- the precision the reference actually trains and renders in
        # (everything on its hot path is torch.float32): forward values and autograd gradients, keys "f32_*".  Their distance from
        # the fp64 values above is the error the reference's own arithmetic carries (glossy corners: the GGX denominator cancels in
        # fp32); the tests budget the HIP kernels against it instead of against a flat tolerance.
        f32 = lambda t: t.detach().float()  # noqa: E731
        b32, r32, n32, ra32, e32 = (f32(t).requires_grad_(True) for t in (base, rough, normals, radiance, env))
        light32 = EnvStandIn(e32)
        pbr32, ex32 = rendering_equation4(b32, r32, n32, f32(viewdirs), ra32, light32, visibility_precompute=f32(vis),
                                          incident_dirs_precompute=f32(dirs), incident_areas_precompute=f32(areas))
        loss32 = (pbr32 * f32(w_pbr)).sum() + sum((ex32[k] * f32(wts[k])).sum() for k in wts) \
            + (ex32["incident_lights"].mean(-2) * f32(w_inc)).sum() + (ex32["global_incident_lights"].mean(-2) * f32(w_glob)).sum()
        grads32 = torch.autograd.grad(loss32, [b32, r32, n32, ra32, e32])
        assert pbr32.dtype == torch.float32 and all(g_.dtype == torch.float32 for g_ in grads32)
        out[pre + "f32_pbr"] = pbr32.detach().numpy()
        for k in ("diffuse_light", "specular", "direct", "indirect"):
            out[pre + "f32_" + k] = ex32[k].detach().numpy()
        out[pre + "f32_mean_incident"] = ex32["incident_lights"].mean(-2).detach().numpy()
        out[pre + "f32_mean_global"] = ex32["global_incident_lights"].mean(-2).detach().numpy()
        for k, gv in zip(("base", "rough", "normals", "radiance", "env"), grads32):
            out[pre + "f32_g_" + k] = gv.numpy()
    np.savez_compressed(os.path.join(OUT, "shading.npz"), **out)
    print("wrote shading.npz", len(out), "arrays")


if __name__ == "__main__":
    main()
