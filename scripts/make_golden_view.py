"""Reference-run fixtures for the callers either side of the rasterizer (tests/golden/render_view.npz, lights.npz,
incident_dirs.npz, render_view_rgss.npz, losses.npz).  Runs only in the authoring container: imports the REFERENCE's own Python from /root/reference
and executes it on CPU; the committed fixtures are data (inputs + the reference's outputs), no reference source.

What is executed, unmodified, from the reference:
  * gaussian_renderer/svgss.py:16-262 `render_view` for is_training True and False, with the reference's real
    `scene.cameras.Camera`, real `scene.direct_light_map.DirectLightMap`, real `rendering_equation4`, real
    `depth2normal` / `rgb_to_srgb`; only the CUDA rasterizer extension is replaced by a RECORDING STUB that stores the
    `features` / `vfeatures` / settings it is handed (-> pins the packing of svgss.py:143-166) and returns seeded
    synthetic "rendered" buffers (-> the rest of render_view pins the image-space unpacking of svgss.py:187-262);
  * scene/direct_light_map.py:70-83 `DirectLightMap.direct_light` and scene/envmap.py:53-72 `EnvLight.direct_light`
    (with and without `.transform`) on their own objects (EnvLight via __new__: its __init__ reads a file);
  * utils/graphics_utils.py:9-37 `fibonacci_sphere_sampling`, utils/sh_utils.py:36-68 `rotation_between_z`,
    scene/gaussian_model.py:23-31 `sample_incident_rays`.

The reference hard-codes device="cuda" / .cuda(); a TorchFunctionMode redirects those to the CPU for the duration of
the calls (this container has no GPU).

    python scripts/make_golden_view.py
"""
import os
import sys
import types

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402  (stub finder for the reference's CUDA-only imports)

OUT = mg.OUT


def _is_cuda(d):
    if isinstance(d, torch.device):
        return d.type == "cuda"
    return isinstance(d, str) and d.startswith("cuda")


class CudaToCpu(TorchFunctionMode):
    """Rewrites every device argument that names CUDA to the CPU."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if _is_cuda(kwargs.get("device")):
            kwargs["device"] = "cpu"
        args = tuple(torch.device("cpu") if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


class cpu_reference:
    """Context: CUDA device redirection + no-op torch.cuda housekeeping calls."""

    def __enter__(self):
        self.mode = CudaToCpu()
        self.mode.__enter__()
        self.saved = (torch.Tensor.cuda, torch.cuda.synchronize, torch.cuda.empty_cache, torch.nn.Module.cuda)
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.empty_cache = lambda *a, **k: None
        return self

    def __exit__(self, *exc):
        torch.Tensor.cuda, torch.cuda.synchronize, torch.cuda.empty_cache, torch.nn.Module.cuda = self.saved
        self.mode.__exit__(*exc)
        return False


def setup_reference():
    if not any(isinstance(f, mg._Finder) for f in sys.meta_path):
        sys.meta_path.insert(0, mg._Finder())
    if mg.REF not in sys.path:
        sys.path.insert(0, mg.REF)
    import torch.utils.cpp_extension as cpp
    cpp.load = lambda *a, **k: mg._Stub("_C")


def np32(t):
    return t.detach().cpu().numpy().astype(np.float32) if t.dtype.is_floating_point else t.detach().cpu().numpy()


class RecordingRasterizer(torch.nn.Module):
    """Stands in for gaussian_renderer.svgss_rasterization.GaussianRasterizer inside the reference's render_view."""
    log = []
    outputs = None   # dict of synthetic rendered buffers, set per call by the driver

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, features=None, vfeatures=None):
        RecordingRasterizer.log.append(dict(settings=self.raster_settings, features=features.detach().clone(),
                                            vfeatures=vfeatures.detach().clone(), shs_is_none=shs is None,
                                            colors_is_none=colors_precomp is None))
        o = RecordingRasterizer.outputs
        return (o["num_rendered"], o["image"], o["normal"], o["opacity"], o["depth"], o["feature"], o["vfeature"],
                o["weights"], o["radii"])


def render_view_fixtures():
    import gaussian_renderer.svgss as ref_svgss
    from scene.cameras import Camera
    from scene.direct_light_map import DirectLightMap
    ref_svgss.GaussianRasterizer = RecordingRasterizer
    out = {}
    n, Ns, H, W = 257, 24, 40, 56
    g = torch.Generator().manual_seed(20260)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    unif = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    geo_n = torch.nn.functional.normalize(rnd(n, 3), dim=-1)
    pc = types.SimpleNamespace(
        get_xyz=0.6 * rnd(n, 3), get_opacity=unif(n, 1), get_scaling=0.02 + 0.05 * unif(n, 3),
        get_rotation=torch.nn.functional.normalize(rnd(n, 4), dim=-1), get_shs=0.3 * rnd(n, 16, 3),
        active_sh_degree=3, max_sh_degree=3, config=[1.0, 1.0, 1.0],
        get_base_color=torch.sigmoid(rnd(n, 12)) * 0.77 + 0.03, get_roughness=torch.sigmoid(rnd(n, 4)) * 0.9 + 0.09,
        get_shading_normal=torch.nn.functional.normalize(geo_n[:, None] + 0.1 * rnd(n, 4, 3), dim=-1),
        get_radiances=(0.2 * rnd(n, Ns, 3)).abs(),
        _visibility_tracing=(unif(n, Ns, 1) > 0.3).float(),
        _incident_dirs=torch.nn.functional.normalize(geo_n[:, None] + 0.9 * rnd(n, Ns, 3), dim=-1),
        _incident_areas=torch.full((n, Ns, 1), 2 * np.pi))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, compute_SHs_python=False)
    A = rnd(3, 3).double().numpy()
    Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    T = np.array([0.1, -0.2, 3.5])
    fovx, fovy = 0.69, 0.52
    bg = torch.tensor([1.0, 1.0, 1.0])
    with cpu_reference():
        cam = Camera(colmap_id=0, R=Q, T=T, FoVx=fovx, FoVy=fovy, fx=None, fy=None, cx=None, cy=None, image=None,
                     image_name="x", uid=0, data_device="cpu", height=H, width=W)
        cam.random_patch = lambda *a, **k: torch.tensor([0.0, 0.0, float(H), float(W)])
        mask = (unif(1, H, W) > 0.15).float()
        cam.image_mask = mask
        light = DirectLightMap(H=8, light_init=3.0)
        light.env = torch.nn.Parameter(3.0 * unif(1, 8, 16, 3))
        for tag, training in (("train", True), ("eval", False)):
            S, VC = (4, 13) if training else (7, 16)
            RecordingRasterizer.outputs = dict(
                num_rendered=1234, image=unif(3, H, W), normal=rnd(3, H, W), opacity=unif(1, H, W),
                depth=2.0 + unif(1, H, W), feature=unif(S, H, W), vfeature=unif(VC, H, W), weights=unif(n, 1),
                radii=(unif(n) * 9).int())
            RecordingRasterizer.log.clear()
            res = ref_svgss.render_view(cam, pc, pipe, bg, scaling_modifier=1.0, override_color=None,
                                        is_training=training, dict_params={"env_light": light})
            rec = RecordingRasterizer.log[-1]
            pre = f"{tag}_"
            out[pre + "features"] = np32(rec["features"])
            out[pre + "vfeatures"] = np32(rec["vfeatures"])
            st = rec["settings"]
            out[pre + "settings_fields"] = np.array(type(st)._fields)
            for f in ("viewmatrix", "projmatrix", "campos", "patch_bbox", "prcppoint", "config", "bg"):
                out[pre + "settings_" + f] = np32(getattr(st, f))
            out[pre + "settings_scalars"] = np.array([st.image_height, st.image_width, st.tanfovx, st.tanfovy,
                                                      st.scale_modifier, st.sh_degree], dtype=np.float64)
            for k, v in RecordingRasterizer.outputs.items():
                if torch.is_tensor(v):
                    out[pre + "raster_" + k] = np32(v)
            for k, v in res.items():
                if torch.is_tensor(v) and k not in ("viewspace_points",):
                    out[pre + "res_" + k] = np32(v)
        for k in ("get_xyz", "get_base_color", "get_roughness", "get_shading_normal", "get_radiances",
                  "_visibility_tracing", "_incident_dirs", "_incident_areas"):
            out["pc_" + k.lstrip("_")] = np32(getattr(pc, k))
        out["env"] = np32(light.env)
        out["bg"] = np32(bg)
        out["image_mask"] = np32(mask)
        out["cam_fov"] = np.array([fovx, fovy])
        out["cam_hw"] = np.array([H, W])
        out["cam_prcppoint"] = np32(cam.prcppoint)
        out["cam_c2w"] = np32(cam.c2w)
        out["cam_intrinsics"] = np32(cam.intrinsics)
    np.savez_compressed(os.path.join(OUT, "render_view.npz"), **out)
    print("wrote render_view.npz", len(out), "arrays")


class RecordingRgssRasterizer(torch.nn.Module):
    """Stands in for gaussian_renderer.rgss_rasterization.GaussianRasterizer inside the reference's stage-1 render_view."""
    log = []
    outputs = None

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, features=None):
        RecordingRgssRasterizer.log.append(dict(settings=self.raster_settings, features=features.detach().clone()))
        o = RecordingRgssRasterizer.outputs
        return (o["num_rendered"], o["num_contrib"], o["image"], o["normal"], o["opacity"], o["depth"], o["feature"],
                o["pseudo_normal"], o["surface_xyz"], o["weights"], o["radii"])


def rgss_view_fixtures():
    """gaussian_renderer/render.py:16-135 (stage 1) with a recording stub rasterizer: pins the rgss feature packing
    ([geo normal, view depth, depth^2], :83-91) and the image-space tail (:107-114)."""
    import importlib
    importlib.import_module("gaussian_renderer.render")
    ref_render = sys.modules["gaussian_renderer.render"]   # (the package also exports a function called `render`)
    from scene.cameras import Camera
    ref_render.GaussianRasterizer = RecordingRgssRasterizer
    out = {}
    n, H, W = 193, 36, 52
    g = torch.Generator().manual_seed(777)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    unif = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    pc = types.SimpleNamespace(
        get_xyz=0.6 * rnd(n, 3), get_opacity=unif(n, 1), get_scaling=0.02 + 0.05 * unif(n, 3),
        get_rotation=torch.nn.functional.normalize(rnd(n, 4), dim=-1), get_shs=0.3 * rnd(n, 16, 3),
        active_sh_degree=3, max_sh_degree=3, get_geo_normal=torch.nn.functional.normalize(rnd(n, 3), dim=-1))
    pipe = types.SimpleNamespace(debug=False, compute_cov3D_python=False, compute_SHs_python=False)
    A = rnd(3, 3).double().numpy()
    Q, _ = np.linalg.qr(A)
    if np.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    T = np.array([-0.2, 0.1, 3.0])
    fovx, fovy = 0.8, 0.6
    bg = torch.tensor([0.0, 0.0, 0.0])
    with cpu_reference():
        cam = Camera(colmap_id=0, R=Q, T=T, FoVx=fovx, FoVy=fovy, fx=None, fy=None, cx=None, cy=None, image=None,
                     image_name="x", uid=0, data_device="cpu", height=H, width=W)
        cam.image_mask = (unif(1, H, W) > 0.1).float()
        RecordingRgssRasterizer.outputs = dict(
            num_rendered=999, num_contrib=(unif(H, W) * 3).int(), image=unif(3, H, W), normal=rnd(3, H, W), opacity=unif(1, H, W),
            depth=2.0 + unif(1, H, W), feature=rnd(5, H, W), pseudo_normal=rnd(3, H, W), surface_xyz=rnd(3, H, W),
            weights=unif(n, 1), radii=(unif(n) * 9).int())
        res = ref_render.render_view(cam, pc, pipe, bg, 1.0, None, computer_pseudo_normal=False)
        rec = RecordingRgssRasterizer.log[-1]
        out["features"] = np32(rec["features"])
        st = rec["settings"]
        out["settings_fields"] = np.array(type(st)._fields)
        out["settings_viewmatrix"] = np32(st.viewmatrix)
        out["settings_scalars"] = np.array([st.image_height, st.image_width, st.tanfovx, st.tanfovy, st.cx, st.cy], dtype=np.float64)
        for k, v in RecordingRgssRasterizer.outputs.items():
            if torch.is_tensor(v):
                out["raster_" + k] = np32(v)
        for k in ("depth_var", "pseudo_normal", "normal", "depth", "opacity", "render"):
            out["res_" + k] = np32(res[k])
        out["pc_xyz"], out["pc_geo_normal"] = np32(pc.get_xyz), np32(pc.get_geo_normal)
        out["image_mask"] = np32(cam.image_mask)
        out["cam_fov"] = np.array([fovx, fovy])
        out["cam_prcppoint"] = np32(cam.prcppoint)
    np.savez_compressed(os.path.join(OUT, "render_view_rgss.npz"), **out)
    print("wrote render_view_rgss.npz", len(out), "arrays")


def light_fixtures():
    from scene.direct_light_map import DirectLightMap
    from scene.envmap import EnvLight
    out = {}
    g = torch.Generator().manual_seed(77)
    dirs = torch.nn.functional.normalize(torch.randn(19, 7, 3, generator=g), dim=-1)
    out["dirs"] = np32(dirs)
    with cpu_reference():
        lm = DirectLightMap(H=16, light_init=3.0)
        lm.env = torch.nn.Parameter(3.0 * torch.rand(1, 16, 32, 3, generator=g))
        out["dlm_env"] = np32(lm.env)
        out["dlm_light"] = np32(lm.direct_light(dirs))
        el = EnvLight.__new__(EnvLight)
        torch.nn.Module.__init__(el)
        el.envmap = 4.0 * torch.rand(48, 96, 3, generator=g) ** 3      # HDR-like, not 32x64: exercises the resample
        el.transform = None
        out["el_envmap"] = np32(el.envmap)
        out["el_light"] = np32(el.direct_light(dirs))
        A = torch.randn(3, 3, generator=g).double().numpy()
        Q, _ = np.linalg.qr(A)
        el.transform = torch.tensor(Q, dtype=torch.float32)
        out["el_transform"] = np32(el.transform)
        out["el_light_transformed"] = np32(el.direct_light(dirs))
        # the 32x64 map the lookup actually samples (scene/envmap.py:62-63)
        out["el_resampled"] = np32(torch.nn.functional.interpolate(el.envmap.permute(2, 0, 1).unsqueeze(0), size=(32, 64),
                                                                   mode="bilinear", align_corners=False)[0].permute(1, 2, 0))
    np.savez_compressed(os.path.join(OUT, "lights.npz"), **out)
    print("wrote lights.npz", len(out), "arrays")


def incident_dir_fixtures():
    from scene.gaussian_model import sample_incident_rays
    from utils.graphics_utils import fibonacci_sphere_sampling
    from utils.sh_utils import rotation_between_z
    out = {}
    g = torch.Generator().manual_seed(5)
    normals = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1)
    normals[0] = torch.tensor([0.0, 0.0, 1.0])
    normals[1] = torch.tensor([0.0, 0.0, -1.0])          # the degenerate branch of rotation_between_z
    normals[2] = torch.nn.functional.normalize(torch.tensor([1e-4, -2e-4, -1.0]), dim=-1)
    out["normals"] = np32(normals)
    with cpu_reference():
        out["rot"] = np32(rotation_between_z(normals))
        for Ns in (8, 64, 384):
            d, a = fibonacci_sphere_sampling(normals, Ns, random_rotate=False)
            out[f"dirs_{Ns}"], out[f"areas_{Ns}"] = np32(d), np32(a)
        d, a = sample_incident_rays(normals, is_training=False, sample_num=24)
        out["sample_eval_24"] = np32(d)
        torch.manual_seed(99)
        d, a = sample_incident_rays(normals, is_training=True, sample_num=24)
        torch.manual_seed(99)
        out["sample_train_24_offsets"] = np32(torch.rand(300, 1) * 2 * np.pi)   # the draw the call above made
        out["sample_train_24"] = np32(d)
    np.savez_compressed(os.path.join(OUT, "incident_dirs.npz"), **out)
    print("wrote incident_dirs.npz", len(out), "arrays")


def loss_fixtures():
    """utils/loss_utils.py `ssim` and F.l1_loss exactly as gaussian_renderer/svgss.py:281-282 calls them on [3,H,W] images,
    with the autograd gradients w.r.t. the rendered image."""
    from utils.loss_utils import ssim
    import torch.nn.functional as F
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, (H, W) in (("a", (45, 70)), ("b", (64, 64)), ("c", (17, 23))):
        gt = torch.rand(3, H, W, generator=g)
        gt[:, : H // 3] = 0.25                                        # a flat region (sigma ~ 0)
        img = (gt + 0.15 * torch.randn(3, H, W, generator=g)).clamp(0, 1).requires_grad_(True)
        with cpu_reference():
            s = ssim(img, gt)
            l1 = F.l1_loss(img, gt)
            gs, = torch.autograd.grad(s, img, retain_graph=True)
            gl, = torch.autograd.grad(l1, img)
        out[f"{tag}_img"], out[f"{tag}_gt"] = np32(img.detach()), np32(gt)
        out[f"{tag}_ssim"], out[f"{tag}_l1"] = np32(s.detach()), np32(l1.detach())
        out[f"{tag}_dssim"], out[f"{tag}_dl1"] = np32(gs), np32(gl)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
    print("wrote losses.npz", len(out), "arrays")


if __name__ == "__main__":
    setup_reference()
    if "--losses-only" in sys.argv:
        loss_fixtures()
        sys.exit(0)
    incident_dir_fixtures()
    light_fixtures()
    render_view_fixtures()
    rgss_view_fixtures()
    loss_fixtures()
