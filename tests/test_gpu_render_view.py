"""End-to-end svgss view on the GPU (shading -> packing -> rasterizer -> unpacking, svgir_harness/render_view.py)
against the same chain assembled from the two CPU oracles in fp64.  Tolerances as in test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import epilogue_oracle as eo
from oracle import shading_oracle as so
from svgir_harness import render_view, runner, scenes, shade_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_l1_ssim_kernels_match_reference_losses(built, tag):
    """csrc/loss.hip against the reference's `ssim` / `F.l1_loss` (fixture: scripts/make_golden_view.py::loss_fixtures) and
    against the fp64 oracle: values and the gradient w.r.t. the rendered image, for L1 alone, SSIM alone and the training
    loss's combination (1 - lambda) L1 + lambda (1 - SSIM), lambda = 0.2."""
    from svgir_harness import losses
    fx = np.load(os.path.join(GOLD, "losses.npz"))
    dev = torch.device("cuda:0")
    img = torch.from_numpy(fx[f"{tag}_img"]).to(dev).requires_grad_(True)
    gt = torch.from_numpy(fx[f"{tag}_gt"]).to(dev)
    l1, s = losses.l1_ssim(img, gt)
    assert abs(float(s) - float(fx[f"{tag}_ssim"])) < 1e-5 and abs(float(l1) - float(fx[f"{tag}_l1"])) < 1e-6
    gs, = torch.autograd.grad(s, img, retain_graph=True)
    gl, = torch.autograd.grad(l1, img, retain_graph=True)
    i64 = torch.from_numpy(fx[f"{tag}_img"]).to(torch.float64).requires_grad_(True)
    o_l1, o_s = eo.l1_ssim_torch(i64, torch.from_numpy(fx[f"{tag}_gt"]))
    os_, = torch.autograd.grad(o_s, i64, retain_graph=True)
    ol_, = torch.autograd.grad(o_l1, i64, retain_graph=True)
    assert abs(float(s) - float(o_s)) < 1e-5 and abs(float(l1) - float(o_l1)) < 1e-6
    for got, ref in ((gs, os_), (gl, ol_)):
        ref = ref.numpy()
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    assert np.abs(gs.cpu().numpy() - fx[f"{tag}_dssim"]).max() <= 3e-4 * np.abs(fx[f"{tag}_dssim"]).max()   # fp32 autograd of the reference
    lam = 0.2
    loss = (1 - lam) * l1 + lam * (1 - s)
    g, = torch.autograd.grad(loss, img)
    ref = ((1 - lam) * ol_ - lam * os_).numpy()
    assert np.abs(g.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    assert float(losses.ssim(img.detach(), gt)) == float(s)
    # the one-node form of the same combination (svgir_harness.losses.l1_ssim_loss): same value, a scaled upstream gradient included
    img2 = img.detach().clone().requires_grad_(True)
    fused = losses.l1_ssim_loss(img2, gt, lam)
    assert abs(float(fused) - float(loss)) <= 1e-6
    (3.0 * fused).backward()
    assert np.abs(img2.grad.cpu().numpy() - 3.0 * ref).max() <= 1e-4 * np.abs(3.0 * ref).max()


def test_l1_ssim_full_size_properties(built):
    """800 x 800: SSIM of an image with itself is 1 with zero gradient; SSIM and L1 are symmetric in their arguments' values."""
    from svgir_harness import losses
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    a = torch.rand(3, 800, 800, generator=g).to(dev).requires_grad_(True)
    b = torch.rand(3, 800, 800, generator=g).to(dev)
    l1, s = losses.l1_ssim(a, a.detach().clone())
    assert abs(float(s) - 1.0) < 1e-6 and float(l1) == 0.0
    gs, = torch.autograd.grad(s, a)
    assert float(gs.abs().max()) < 1e-7
    l1_ab, s_ab = losses.l1_ssim(a, b)
    l1_ba, s_ba = losses.l1_ssim(b, a.detach())
    assert abs(float(s_ab) - float(s_ba)) < 1e-6 and abs(float(l1_ab) - float(l1_ba)) < 1e-7


def test_depth2normal_is_differentiable_like_the_reference(built):
    """utils/image_utils.py:61-125 with autograd (tests/golden/depth2normal_grad.npz, scripts/make_golden_d2n.py): the stage-1
    loss differentiates through the pseudo normal (gaussian_renderer/render.py:158-160)."""
    from svgir_harness import render_view
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth2normal_grad.npz"))
    dev = torch.device("cuda:0")
    depth = torch.from_numpy(g["depth"]).to(dev).requires_grad_(True)
    n = render_view.depth2normal(depth, torch.from_numpy(g["mask"]).to(dev), float(g["fovx"]), float(g["fovy"]), g["prcppoint"])
    ref_n = g["normal"]
    assert np.abs(n.detach().cpu().numpy() - ref_n).max() < 2e-5
    (n * torch.from_numpy(g["upstream"]).to(dev)).sum().backward()
    got, ref = depth.grad.cpu().numpy(), g["depth_grad"]
    assert got.shape == ref.shape
    # Pixels whose summed cross product is EXACTLY zero (an image-border pixel next to a masked one: two of its four
    # differences vanish) go through normalize's 1 / eps = 1e12 branch; analytically their contributions cancel, numerically
    # both the reference (fp32 autograd) and the kernel leave 1e12 x rounding noise there.  They and the pixels they touch are
    # excluded; everything else is compared.
    degenerate = (np.abs(ref_n).sum(0) == 0) & g["mask"][0]
    touched = degenerate.copy()
    touched[1:] |= degenerate[:-1]; touched[:-1] |= degenerate[1:]; touched[:, 1:] |= degenerate[:, :-1]; touched[:, :-1] |= degenerate[:, 1:]
    ok = ~touched[None]
    assert ok.mean() > 0.85
    assert np.abs(got - ref)[ok].max() <= 2e-4 * np.abs(ref[ok]).max(), np.abs(got - ref)[ok].max() / np.abs(ref[ok]).max()
