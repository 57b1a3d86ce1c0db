"""End-to-end svgss view on the GPU (shading -> packing -> rasterizer -> unpacking, svgir_harness/render_view.py)
against the same chain assembled from the two CPU oracles in fp64.  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import epilogue_oracle as eo
from oracle import shading_oracle as so
from svgir_harness import render_view, runner, scenes, shade_inputs

pytestmark = pytest.mark.gpu


def _cmp(name, a, b, tol=2e-4, flip_frac=5e-4):
    a = a.detach().double().cpu().numpy().reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    scale = max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > tol * (scale + np.abs(b))
    assert bad.mean() <= flip_frac, f"{name}: {bad.sum()}/{bad.size} entries off (max {np.abs(a - b).max():.3e}, scale {scale:.3e})"


@pytest.mark.parametrize("is_training", [True, False])
def test_svgss_view_end_to_end(built, is_training):
    dev = torch.device("cuda:0")
    S, VS = (4, 52) if is_training else (7, 64)
    sc = scenes.surface_scene(P=4000, W=160, H=120, seed=61, sh_degree=2, variant="svgss", S=S, VS=VS, scale_lo=0.02,
                              scale_hi=0.07)
    P = sc["means3D"].shape[0]
    d = shade_inputs.make(P, 32, seed=4)
    d["roughness"] = d["roughness"].clamp_min(0.3)
    # ---- GPU ----
    sct = runner.to_torch(sc, dev)
    mat = {k: v.to(dev) for k, v in d.items() if k != "env"}
    with torch.no_grad():
        res, _ = render_view.render_svgss_view(sct, mat, shade_inputs.Light(d["env"].to(dev)), is_training)
    # ---- oracles, fp64 ----
    dd = {k: v.double() for k, v in d.items()}
    ref = so.shade(dd["base_color"], dd["roughness"], dd["normals"], dd["viewdirs"], dd["radiance"], dd["visibility"],
                   dd["dirs"], dd["areas"], dd["env"])
    view3 = torch.from_numpy(sc["viewmatrix"]).double()[:3, :3]
    f, vf = so.pack(ref, dd["base_color"], dd["roughness"], dd["normals"], view3, is_training)
    sc_o = dict(sc)
    sc_o["features"] = f.float().numpy()
    sc_o["vfeatures"] = vf.float().numpy()
    o = orc.OracleRun(sc_o, orc.SVGSS)
    R = o.forward()
    im = o.images()
    t = lambda k: torch.from_numpy(np.asarray(im[k], dtype=np.float64))  # noqa: E731
    rendered = (R, t("color"), t("normal"), t("opacity"), t("depth"), t("feature"), t("vfeature"),
                torch.from_numpy(np.asarray(im["weights"], dtype=np.float64)), torch.from_numpy(im["radii"]))
    exp = eo.unpack_svgss_torch(t("opacity"), t("feature"), t("vfeature"), torch.from_numpy(sc["bg"]).double(), is_training)
    exp.update(render=t("color"), depth=t("depth"), opacity=t("opacity"))
    assert res["num_rendered"] == R
    assert np.array_equal(res["radii"].cpu().numpy(), im["radii"])
    keys = ["render", "depth", "opacity", "pbr", "normal", "base_color", "roughness", "local_lights", "visibility"]
    keys += ["diffuse"] if is_training else ["lights", "direct", "indirect"]
    for k in keys:
        assert res[k].shape == exp[k].shape, k
        _cmp(k, res[k], exp[k].numpy())


@pytest.mark.parametrize("tag", ["train", "eval"])
def test_epilogue_kernels_match_reference_render_view(built, tag):
    """the fused unpack kernel and depth2normal against what the reference's own render_view produced
    from the same rasterizer buffers (tests/golden/render_view.npz), and the unpack backward against torch.autograd of the
    pinned restatement."""
    import os
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_view.npz"))
    training = tag == "train"
    ras = {k: torch.from_numpy(g[f"{tag}_raster_{k}"]).to(dev) for k in ("image", "normal", "opacity", "depth", "feature", "vfeature", "weights", "radii")}
    tup = (1234, ras["image"], ras["normal"], ras["opacity"], ras["depth"], ras["feature"], ras["vfeature"], ras["weights"], ras["radii"])
    bg = torch.from_numpy(g["bg"]).to(dev)
    got = render_view.unpack(tup, bg, training)
    keys = ["pbr", "normal", "base_color", "roughness", "local_lights", "visibility"] + (["diffuse"] if training else ["lights", "direct", "indirect"])
    for k in keys + ["render", "depth", "opacity"]:
        np.testing.assert_allclose(got[k].cpu().numpy(), np.broadcast_to(g[f"{tag}_res_{k}"], got[k].shape), rtol=2e-5, atol=2e-6, err_msg=k)
    assert np.array_equal(got["visibility_filter"].cpu().numpy(), g[f"{tag}_res_visibility_filter"])
    fovx, fovy = g["cam_fov"]
    pn = render_view.depth2normal(ras["depth"], torch.from_numpy(g["image_mask"]).to(dev), fovx, fovy, g["cam_prcppoint"])
    np.testing.assert_allclose(pn.cpu().numpy(), g[f"{tag}_res_pseudo_normal"], rtol=0, atol=3e-5)
    # backward: random upstream gradients on every result plane
    gen = torch.Generator().manual_seed(5)
    lv = {k: ras[k].clone().requires_grad_(True) for k in ("opacity", "feature", "vfeature")}
    res = render_view.unpack((1234, ras["image"], ras["normal"], lv["opacity"], ras["depth"], lv["feature"], lv["vfeature"],
                              ras["weights"], ras["radii"]), bg, training)
    wts = {k: torch.randn(res[k].shape, generator=gen) for k in keys}
    sum((res[k] * wts[k].to(dev)).sum() for k in keys).backward()
    ld = {k: ras[k].double().cpu().requires_grad_(True) for k in ("opacity", "feature", "vfeature")}
    ref = eo.unpack_svgss_torch(ld["opacity"], ld["feature"], ld["vfeature"], bg.double().cpu(), training)
    sum((ref[k] * wts[k].double()).sum() for k in keys).backward()
    for k in ("opacity", "feature", "vfeature"):
        _cmp("d_" + k, lv[k].grad, ld[k].grad.numpy(), tol=2e-4, flip_frac=2e-3)


def test_rgss_packing_and_unpacking_match_reference_render_view(built):
    """Stage 1 (gaussian_renderer/render.py): the features the reference hands to its rasterizer and the image-space tail,
    recorded from the reference's own render_view (tests/golden/render_view_rgss.npz); backward vs torch.autograd."""
    import os
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_view_rgss.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731
    xyz, nrm, vm = t("pc_xyz").requires_grad_(True), t("pc_geo_normal").requires_grad_(True), t("settings_viewmatrix")
    f = render_view.pack_rgss(xyz, nrm, vm)
    np.testing.assert_allclose(f.detach().cpu().numpy(), g["features"], rtol=2e-6, atol=2e-6)
    w = torch.randn(f.shape, generator=torch.Generator().manual_seed(1)).to(dev)
    (f * w).sum().backward()
    x64, n64 = xyz.detach().double().cpu().requires_grad_(True), nrm.detach().double().cpu().requires_grad_(True)
    d = (torch.cat([x64, torch.ones_like(x64[:, :1])], -1) @ vm.double().cpu())[:, 2:3]
    (torch.cat([n64, d, d * d], -1) * w.double().cpu()).sum().backward()
    _cmp("d_means3D", xyz.grad, x64.grad.numpy(), tol=1e-5, flip_frac=0)
    _cmp("d_normals", nrm.grad, n64.grad.numpy(), tol=1e-6, flip_frac=0)
    ras = {k: t("raster_" + k) for k in ("num_contrib", "image", "normal", "opacity", "depth", "feature", "pseudo_normal", "surface_xyz", "weights", "radii")}
    lv = {k: ras[k].clone().requires_grad_(True) for k in ("opacity", "depth", "feature")}
    res = render_view.unpack_rgss((999, ras["num_contrib"], ras["image"], ras["normal"], lv["opacity"], lv["depth"], lv["feature"],
                                   ras["pseudo_normal"], ras["surface_xyz"], ras["weights"], ras["radii"]))
    np.testing.assert_allclose(res["depth_var"].detach().cpu().numpy(), g["res_depth_var"], rtol=2e-5, atol=2e-5)
    wv = torch.randn(res["depth_var"].shape, generator=torch.Generator().manual_seed(2)).to(dev)
    wn = torch.randn(res["feature_normal"].shape, generator=torch.Generator().manual_seed(3)).to(dev)
    ((res["depth_var"] * wv).sum() + (res["feature_normal"] * wn).sum()).backward()
    ld = {k: ras[k].double().cpu().requires_grad_(True) for k in ("opacity", "depth", "feature")}
    mask = (ras["num_contrib"] > 0).double().cpu()
    xf = ld["feature"] / ld["opacity"].clamp_min(1e-5) * mask
    ((((xf[4:5] - ld["depth"].square()) * wv.double().cpu()).sum()) + (xf[0:3] * wn.double().cpu()).sum()).backward()
    for k in ("opacity", "depth", "feature"):
        _cmp("d_" + k, lv[k].grad, ld[k].grad.numpy(), tol=2e-5, flip_frac=1e-4)
    fovx, fovy = g["cam_fov"]
    pn = render_view.depth2normal(ras["depth"], t("image_mask"), fovx, fovy, g["cam_prcppoint"])
    np.testing.assert_allclose(pn.cpu().numpy(), g["res_pseudo_normal"], rtol=0, atol=3e-5)
