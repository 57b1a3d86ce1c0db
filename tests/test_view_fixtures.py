"""Pins the oracles of the rasterizer's callers (oracle/epilogue_oracle.py, the packing / light lookups of
oracle/shading_oracle.py) and the harness' image-space unpacking against fixtures produced by RUNNING THE REFERENCE
(scripts/make_golden_view.py): its whole `render_view` (gaussian_renderer/svgss.py:16-262) with the real Camera /
DirectLightMap and a recording stub in place of the CUDA rasterizer, `DirectLightMap.direct_light`,
`EnvLight.direct_light` (with and without `.transform`), `fibonacci_sphere_sampling` / `rotation_between_z` /
`sample_incident_rays`."""
import os

import numpy as np
import pytest
import torch

from oracle import epilogue_oracle as eo
from oracle import shading_oracle as so

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def rv():
    return np.load(os.path.join(GOLD, "render_view.npz"))


def _t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_packing_matches_what_the_reference_hands_to_its_rasterizer(rv, tag):
    """svgss.py:143-166: the `features` / `vfeatures` tensors recorded at the reference's rasterizer call."""
    training = tag == "train"
    view = _t(rv[f"{tag}_settings_viewmatrix"])
    out = so.shade(_t(rv["pc_get_base_color"]), _t(rv["pc_get_roughness"]), _t(rv["pc_get_shading_normal"]),
                   torch.nn.functional.normalize(_t(rv[f"{tag}_settings_campos"])[None] - _t(rv["pc_get_xyz"]), dim=-1),
                   _t(rv["pc_get_radiances"]), _t(rv["pc_visibility_tracing"]), _t(rv["pc_incident_dirs"]),
                   _t(rv["pc_incident_areas"]), _t(rv["env"]))
    f, vf = so.pack(out, _t(rv["pc_get_base_color"]), _t(rv["pc_get_roughness"]), _t(rv["pc_get_shading_normal"]),
                    view[:3, :3], training)
    assert tuple(f.shape) == rv[f"{tag}_features"].shape and tuple(vf.shape) == rv[f"{tag}_vfeatures"].shape
    np.testing.assert_allclose(f.numpy(), rv[f"{tag}_features"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(vf.numpy(), rv[f"{tag}_vfeatures"], rtol=2e-4, atol=2e-6)
    # the settings the reference builds for the call (field order is part of the API)
    assert list(rv[f"{tag}_settings_fields"]) == ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                                                  "viewmatrix", "projmatrix", "patch_bbox", "prcppoint", "sh_degree", "campos",
                                                  "prefiltered", "debug", "config"]
    np.testing.assert_array_equal(rv[f"{tag}_settings_config"], [1.0, 1.0, 1.0])


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_unpacking_matches_the_rest_of_reference_render_view(rv, tag):
    """svgss.py:187-262 on the stub rasterizer's buffers: numpy and differentiable-torch restatements of the oracle."""
    training = tag == "train"
    ras = {k: rv[f"{tag}_raster_{k}"] for k in ("image", "normal", "opacity", "depth", "feature", "vfeature", "weights", "radii")}
    res = eo.unpack_svgss(ras["opacity"], ras["feature"], ras["vfeature"], rv["bg"], training)
    keys = ["pbr", "normal", "base_color", "roughness", "local_lights", "visibility"] + (["diffuse"] if training else ["lights", "direct", "indirect"])
    for k in keys:
        np.testing.assert_allclose(res[k], rv[f"{tag}_res_{k}"], rtol=2e-5, atol=2e-6, err_msg=k)
    # the differentiable torch restatement (reference for the fused kernel's backward) agrees too
    tt = eo.unpack_svgss_torch(_t(ras["opacity"]), _t(ras["feature"]), _t(ras["vfeature"]), _t(rv["bg"]), training)
    for k in keys:
        np.testing.assert_allclose(tt[k].numpy(), rv[f"{tag}_res_{k}"], rtol=2e-5, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_depth2normal_matches_reference(rv, tag):
    fovx, fovy = rv["cam_fov"]
    n = eo.depth2normal(rv[f"{tag}_raster_depth"], rv["image_mask"], fovx, fovy, rv["cam_prcppoint"])
    np.testing.assert_allclose(n, rv[f"{tag}_res_pseudo_normal"], rtol=0, atol=3e-5)


def test_eval_environment_backdrop_matches_reference(rv):
    """svgss.py:255-260: env lookup along the camera's world-space pixel directions composited behind the render."""
    H, W = [int(v) for v in rv["cam_hw"]]
    K, c2w = rv["cam_intrinsics"].astype(np.float64), rv["cam_c2w"].astype(np.float64)
    v, u = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = np.stack([(u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)], 0)
    d = d / np.linalg.norm(d, axis=0, keepdims=True)
    d = (c2w[:3, :3] @ d.reshape(3, -1)).reshape(3, H, W)
    env = so.env_lookup(_t(rv["env"]), _t(d.transpose(1, 2, 0))).numpy().transpose(2, 0, 1)
    np.testing.assert_allclose(eo.rgb_to_srgb(env), rv["eval_res_env_only"], rtol=2e-4, atol=2e-5)
    op = rv["eval_raster_opacity"].astype(np.float64)
    np.testing.assert_allclose(rv["eval_raster_image"] + (1 - op) * eo.rgb_to_srgb(env), rv["eval_res_render_env"], rtol=2e-4, atol=2e-5)


def test_light_lookups_match_reference_classes():
    g = np.load(os.path.join(GOLD, "lights.npz"))
    dirs = _t(g["dirs"])
    got = so.env_lookup(_t(g["dlm_env"]), dirs, softplus=True, scale=2.0)
    np.testing.assert_allclose(got.numpy(), g["dlm_light"], rtol=2e-5, atol=2e-6)
    # EnvLight: 32x64 bilinear down-sample of the HDR map, identity transfer, optional rotation of the lookup direction
    small = eo.resample_bilinear(g["el_envmap"], 32, 64)
    np.testing.assert_allclose(small, g["el_resampled"], rtol=2e-5, atol=2e-6)
    got = so.env_lookup(_t(small), dirs, softplus=False, scale=1.0)
    np.testing.assert_allclose(got.numpy(), g["el_light"], rtol=2e-5, atol=2e-6)
    T = _t(g["el_transform"])
    got = so.env_lookup(_t(small), dirs @ T.T, softplus=False, scale=1.0)
    np.testing.assert_allclose(got.numpy(), g["el_light_transformed"], rtol=2e-5, atol=2e-6)


def test_incident_direction_generation_matches_reference():
    g = np.load(os.path.join(GOLD, "incident_dirs.npz"))
    np.testing.assert_allclose(eo.rotation_between_z(g["normals"]), g["rot"], rtol=0, atol=2e-6)
    for Ns in (8, 64, 384):
        d, a = eo.fibonacci_dirs(g["normals"], Ns)
        np.testing.assert_allclose(d, g[f"dirs_{Ns}"], rtol=0, atol=3e-6)
        np.testing.assert_allclose(a, g[f"areas_{Ns}"], rtol=1e-6)
    d, _ = eo.fibonacci_dirs(g["normals"], 24)
    np.testing.assert_allclose(d, g["sample_eval_24"], rtol=0, atol=3e-6)
    d, _ = eo.fibonacci_dirs(g["normals"], 24, offsets=g["sample_train_24_offsets"])
    np.testing.assert_allclose(d, g["sample_train_24"], rtol=0, atol=3e-6)


def test_l1_ssim_oracle_matches_the_reference_losses():
    """oracle/epilogue_oracle.py::l1_ssim_torch against `ssim` / `F.l1_loss` of the reference run in the authoring container
    (scripts/make_golden_view.py::loss_fixtures), values and autograd gradients."""
    import torch
    from oracle import epilogue_oracle as eo
    fx = np.load(os.path.join(GOLD, "losses.npz"))
    for tag in ("a", "b", "c"):
        img = torch.from_numpy(fx[f"{tag}_img"]).to(torch.float64).requires_grad_(True)
        gt = torch.from_numpy(fx[f"{tag}_gt"]).to(torch.float64)
        l1, s = eo.l1_ssim_torch(img, gt)
        gs, = torch.autograd.grad(s, img, retain_graph=True)
        gl, = torch.autograd.grad(l1, img)
        assert abs(float(s) - float(fx[f"{tag}_ssim"])) < 2e-6, tag
        assert abs(float(l1) - float(fx[f"{tag}_l1"])) < 1e-6, tag
        ref = fx[f"{tag}_dssim"].astype(np.float64)
        assert np.abs(gs.numpy() - ref).max() <= 2e-4 * np.abs(ref).max(), tag    # (the reference's gradient is fp32 autograd)
        assert np.abs(gl.numpy() - fx[f"{tag}_dl1"]).max() <= 1e-9, tag
