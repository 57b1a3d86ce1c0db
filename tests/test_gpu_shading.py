"""GPU parity of the fused shading kernels (csrc/shade.hip) against the shading oracle and the reference-generated
fixtures (tests/golden/shading.npz).

Tolerance: 1e-4 relative to the tensor scale (fp32 kernel vs fp64 oracle) for roughness >= 0.3 on the synthetic cases.  The fixture
cases -- the reference's own `rendering_equation4` run on CPU in fp64 AND in fp32 (scripts/make_golden.py), roughness over the
reference's whole range 0.09 .. 0.99, fixture `c` with a third of the corners at 0.09 .. 0.15 where the GGX denominator
NoH^2 (a2 - 1) + 1 cancels in fp32 -- are budgeted against what the reference's own fp32 arithmetic loses: per tensor,
err(HIP vs reference fp64) <= 1.5 x err(reference fp32 vs reference fp64) + 1e-4 (errors = max abs / tensor scale), forward and
gradients.  (Rounds 1-4 used flat 2e-3 / 5e-3 here and only asserted that the reference's fp32 was that far off too; measured, it is
8e-6 .. 3.5e-4.)"""
import os

import numpy as np
import pytest
import torch

from oracle import shading_oracle as so

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shading.npz")


class _Light:
    def __init__(self, env):
        self.env = env


def _close(name, a, b, tol=1e-4):
    a = a.detach().double().cpu().reshape(-1)
    b = (b if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).detach().double().cpu().reshape(-1)
    scale = max(float(b.abs().max()), 1e-30)
    err = float((a - b).abs().max())
    assert err <= tol * scale + 1e-7, f"{name}: max err {err:.3e} vs scale {scale:.3e}"


def _err(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = (b if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).detach().double().cpu().reshape(-1)
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _within_budget(name, hip, ref64, ref32, factor=1.5, floor=1e-4):
    """err(HIP vs reference fp64) <= factor * err(reference fp32 vs reference fp64) + floor; returns the budget used."""
    e_hip, e_ref = _err(hip, ref64), _err(ref32, ref64)
    assert e_hip <= factor * e_ref + floor, f"{name}: HIP err {e_hip:.3e} > {factor} x reference-fp32 err {e_ref:.3e} + {floor:.0e}"
    return factor * e_ref + floor


def _fixture(tag):
    g = np.load(GOLD)
    pre = "shade_" + tag + "_"
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def _random_case(n, Ns, seed, He=32, We=64, rough_lo=0.09):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)  # noqa: E731
    nrm0 = torch.nn.functional.normalize(rnd(n, 1, 3), dim=-1)
    return dict(
        base=torch.sigmoid(rnd(n, 12)) * 0.77 + 0.03, rough=torch.sigmoid(rnd(n, 4)) * (0.99 - rough_lo) + rough_lo,
        normals=torch.nn.functional.normalize(nrm0 + 0.1 * rnd(n, 4, 3), dim=-1),
        viewdirs=torch.nn.functional.normalize(nrm0[:, 0] + 0.6 * rnd(n, 3), dim=-1),
        dirs=torch.nn.functional.normalize(nrm0 + 0.9 * rnd(n, Ns, 3), dim=-1),
        areas=torch.full((n, Ns, 1), 2 * np.pi, dtype=torch.float64),
        vis=(torch.rand(n, Ns, 1, generator=g, dtype=torch.float64) > 0.3).double(),
        radiance=(0.2 * rnd(n, Ns, 3)).abs(), env=3.0 * torch.rand(1, He, We, 3, generator=g, dtype=torch.float64))


@pytest.mark.parametrize("case", ["fix_a", "fix_b", "fix_c", "rand64", "rand384"])
def test_shading_forward_and_packing(built, case):
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    fix = case.startswith("fix")
    d = _fixture(case[-1]) if fix else _random_case(500, 384 if case == "rand384" else 64, 7, rough_lo=0.3)
    ftol = 1e-4
    ref = so.shade(d["base"], d["rough"], d["normals"], d["viewdirs"], d["radiance"], d["vis"], d["dirs"], d["areas"], d["env"])
    f32 = {k: v.float().to(dev) for k, v in d.items() if k in ("base", "rough", "normals", "viewdirs", "radiance", "vis", "dirs", "areas", "env")}
    with torch.no_grad():
        pbr, ex = shading.rendering_equation4(f32["base"], f32["rough"], f32["normals"], f32["viewdirs"], f32["radiance"],
                                              _Light(f32["env"]), visibility_precompute=f32["vis"],
                                              incident_dirs_precompute=f32["dirs"], incident_areas_precompute=f32["areas"])
    if fix:   # against the reference's own outputs, within what its own fp32 arithmetic loses (see the module docstring)
        got = dict(pbr=pbr, mean_incident=ex["incident_lights"].mean(-2), mean_global=ex["global_incident_lights"].mean(-2),
                   **{k: ex[k] for k in ("diffuse_light", "specular", "direct", "indirect")})
        ftol = max(_within_budget(k, v, d[k], d["f32_" + k]) for k, v in got.items())
        for k, v in got.items():   # (and the oracle restates the reference: fp64 vs fp64)
            assert _err(ref[k], d[k]) < 1e-9, k
    else:
        _close("pbr", pbr, ref["pbr"], tol=ftol)
        for k in ("diffuse_light", "specular", "direct", "indirect"):
            _close(k, ex[k], ref[k], tol=ftol)
        _close("mean_incident", ex["incident_lights"].mean(-2), ref["mean_incident"])
        _close("mean_global", ex["global_incident_lights"].mean(-2), ref["mean_global"])
    view = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))[0]
    vm = torch.eye(4, dtype=torch.float64)
    vm[:3, :3] = view
    for training in (True, False):
        fr, vr = so.pack(ref, d["base"], d["rough"], d["normals"], view, training)
        with torch.no_grad():
            f, vf, _ = shading.shade_and_pack(f32["base"], f32["rough"], f32["normals"], f32["viewdirs"], f32["radiance"],
                                              _Light(f32["env"]), f32["vis"], f32["dirs"], f32["areas"], vm.float().to(dev), training)
        assert f.shape == fr.shape and vf.shape == vr.shape
        _close("features", f, fr, tol=ftol)
        _close("vfeatures", vf, vr, tol=ftol)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_shading_backward_within_the_reference_fp32_budget(built, tag):
    """Gradients of the fixture's loss (the reference's autograd in fp64 = g_*, in fp32 = f32_g_*) through the HIP kernels."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _fixture(tag)
    names = ("base", "rough", "normals", "radiance", "env")
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                          visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    w = lambda k: d[k].float().to(dev)  # noqa: E731
    loss = (pbr * w("w_pbr")).sum() + sum((ex[k] * w("w_" + k)).sum() for k in ("diffuse_light", "specular", "direct", "indirect")) \
        + (ex["incident_lights"].mean(-2) * w("w_inc")).sum() + (ex["global_incident_lights"].mean(-2) * w("w_glob")).sum()
    loss.backward()
    for k in names:
        _within_budget("grad_" + k, lg[k].grad, d["g_" + k], d["f32_g_" + k])


@pytest.mark.parametrize("case", ["rand64"])
def test_shading_backward(built, case):
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(300, 64, 11, rough_lo=0.3)
    gtol = 3e-4
    names = ("base", "rough", "normals", "radiance", "env")
    lo = {k: d[k].clone().requires_grad_(True) for k in names}
    ref = so.shade(lo["base"], lo["rough"], lo["normals"], d["viewdirs"], lo["radiance"], d["vis"], d["dirs"], d["areas"], lo["env"])
    g = torch.Generator().manual_seed(3)
    w = {k: torch.randn(ref[k].shape, generator=g, dtype=torch.float64) for k in
         ("pbr", "diffuse_light", "specular", "direct", "indirect", "mean_incident", "mean_local", "mean_global")}
    sum(((ref[k] * w[k]).sum() for k in w)).backward()
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                          visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    wd = {k: v.float().to(dev) for k, v in w.items()}
    loss = (pbr * wd["pbr"]).sum() + sum((ex[k] * wd[k]).sum() for k in ("diffuse_light", "specular", "direct", "indirect"))
    loss = loss + (ex["incident_lights"].mean(-2) * wd["mean_incident"]).sum() \
        + (ex["global_incident_lights"].mean(-2) * wd["mean_global"]).sum() \
        + (ex["local_incident_lights"].mean(-2) * wd["mean_local"]).sum()
    loss.backward()
    for k in names:
        _close("grad_" + k, lg[k].grad, lo[k].grad, tol=gtol)


@pytest.mark.parametrize("training", [True, False])
def test_shade_and_pack_backward(built, training):
    """Gradients through the fused packing (features / vfeatures / reduced all used) vs autograd of the oracle."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(200, 96, 23, rough_lo=0.3)   # 96 samples: one full + one partial 64-sample chunk
    names = ("base", "rough", "normals", "radiance", "env")
    view = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))[0]
    vm = torch.eye(4, dtype=torch.float64)
    vm[:3, :3] = view
    lo = {k: d[k].clone().requires_grad_(True) for k in names}
    ref = so.shade(lo["base"], lo["rough"], lo["normals"], d["viewdirs"], lo["radiance"], d["vis"], d["dirs"], d["areas"], lo["env"])
    fr, vr = so.pack(ref, lo["base"], lo["rough"], lo["normals"], view, training)
    g = torch.Generator().manual_seed(5)
    wf = torch.randn(fr.shape, generator=g, dtype=torch.float64)
    wv = torch.randn(vr.shape, generator=g, dtype=torch.float64)
    wp = torch.randn(ref["pbr"].shape, generator=g, dtype=torch.float64)
    ((fr * wf).sum() + (vr * wv).sum() + (ref["pbr"] * wp).sum()).backward()
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    f, vf, red = shading.shade_and_pack(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                        c["vis"], c["dirs"], c["areas"], vm.float().to(dev), training)
    _close("features", f, fr, tol=1e-4)
    _close("vfeatures", vf, vr, tol=1e-4)
    ((f * wf.float().to(dev)).sum() + (vf * wv.float().to(dev)).sum() + (red[:, 0:12] * wp.float().to(dev)).sum()).backward()
    for k in names:
        _close("grad_" + k, lg[k].grad, lo[k].grad, tol=3e-4)
    # only one of the outputs used: the other upstream gradients arrive as None
    for v in lg.values():
        v.grad = None
    f2, vf2, _ = shading.shade_and_pack(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                        c["vis"], c["dirs"], c["areas"], vm.float().to(dev), training)
    for v in lo.values():
        v.grad = None
    ref = so.shade(lo["base"], lo["rough"], lo["normals"], d["viewdirs"], lo["radiance"], d["vis"], d["dirs"], d["areas"], lo["env"])
    fr, vr = so.pack(ref, lo["base"], lo["rough"], lo["normals"], view, training)
    (fr * wf).sum().backward()
    (f2 * wf.float().to(dev)).sum().backward()
    for k in ("radiance", "env"):
        ref_g = lo[k].grad if lo[k].grad is not None else torch.zeros_like(lo[k])
        _close("featonly_grad_" + k, lg[k].grad, ref_g, tol=3e-4)


def test_envlight_hdr_map_with_transform(built):
    """scene/envmap.py EnvLight: HDR lat-long map (bilinear down-sample to 32x64, no softplus, scale 1) looked up with
    rotated directions (`.transform`); gradients w.r.t. the material inputs."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(300, 64, 31, rough_lo=0.3)
    g = torch.Generator().manual_seed(8)
    hdr = (torch.rand(64, 128, 3, generator=g, dtype=torch.float64) * 2.0) ** 2
    rot = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))[0]

    class EnvLight:
        def __init__(self, envmap, transform):
            self.envmap, self.transform = envmap, transform

    env32 = torch.nn.functional.interpolate(hdr.permute(2, 0, 1).unsqueeze(0), size=(32, 64), mode="bilinear",
                                            align_corners=False)[0].permute(1, 2, 0).unsqueeze(0)
    names = ("base", "rough", "normals", "radiance")
    lo = {k: d[k].clone().requires_grad_(True) for k in names}
    ref = so.shade(lo["base"], lo["rough"], lo["normals"], d["viewdirs"], lo["radiance"], d["vis"], d["dirs"], d["areas"],
                   env32, softplus=False, scale=1.0, transform=rot)
    (ref["pbr"].sum() + 0.5 * ref["direct"].sum()).backward()
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"],
                                          EnvLight(hdr.float().to(dev), rot.float().to(dev)),
                                          visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    (pbr.sum() + 0.5 * ex["direct"].sum()).backward()
    _close("pbr", pbr, ref["pbr"])
    _close("direct", ex["direct"], ref["direct"])
    _close("mean_global", ex["global_incident_lights"].mean(-2), ref["mean_global"])
    for k in names:
        _close("grad_" + k, lg[k].grad, lo[k].grad, tol=3e-4)


@pytest.mark.parametrize("Ns,He,We", [(1, 32, 64), (5, 16, 32), (65, 64, 128), (130, 32, 64)])
def test_shading_odd_sample_counts_and_env_sizes(built, Ns, He, We):
    """Partial 64-sample chunks, a single sample, and an env map too large for the LDS gradient image (64x128: the
    backward falls back to global atomics on the env-gradient table)."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(120, Ns, 40 + Ns, He=He, We=We, rough_lo=0.3)
    names = ("base", "rough", "normals", "radiance", "env")
    lo = {k: d[k].clone().requires_grad_(True) for k in names}
    ref = so.shade(lo["base"], lo["rough"], lo["normals"], d["viewdirs"], lo["radiance"], d["vis"], d["dirs"], d["areas"], lo["env"])
    (ref["pbr"].sum() + ref["diffuse_light"].sum() + ref["mean_global"].sum()).backward()
    lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], lg["radiance"], _Light(lg["env"]),
                                          visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    (pbr.sum() + ex["diffuse_light"].sum() + ex["global_incident_lights"].mean(-2).sum()).backward()
    _close("pbr", pbr, ref["pbr"])
    _close("diffuse_light", ex["diffuse_light"], ref["diffuse_light"])
    for k in names:
        _close("grad_" + k, lg[k].grad, lo[k].grad, tol=3e-4)


def test_rendering_equation4_survives_the_reference_eval_chunk_loop(built):
    """gaussian_renderer/svgss.py:121-151 (is_training=False): rendering_equation4 per 100k-surfel chunk, then
    `torch.cat` over EVERY key of extra_results, then the features / vfeatures assembly.  The fused kernel returns the
    per-sample light tensors reduced to their mean; the concatenation and `.mean(-2)` must still give the reference's
    features (pinned by tests/golden/render_view.npz, recorded at the reference's rasterizer call)."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_view.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dev)  # noqa: E731
    base, rough, normal = t("pc_get_base_color"), t("pc_get_roughness"), t("pc_get_shading_normal")
    inc, vis, dirs, areas = t("pc_get_radiances"), t("pc_visibility_tracing"), t("pc_incident_dirs"), t("pc_incident_areas")
    viewdirs = torch.nn.functional.normalize(t("eval_settings_campos") - t("pc_get_xyz"), dim=-1)
    light = _Light(t("env"))
    chunk = 100
    brdf, extras = [], []
    with torch.no_grad():
        for i in range(0, base.shape[0], chunk):
            b, e = shading.rendering_equation4(base[i:i + chunk], rough[i:i + chunk], normal[i:i + chunk].detach(),
                                               viewdirs[i:i + chunk], inc[i:i + chunk], light,
                                               visibility_precompute=vis[i:i + chunk],
                                               incident_dirs_precompute=dirs[i:i + chunk],
                                               incident_areas_precompute=areas[i:i + chunk])
            brdf.append(b)
            extras.append(e)
    brdf = torch.cat(brdf, dim=0)
    extra = {k: torch.cat([e[k] for e in extras], dim=0) for k in extras[0]}      # svgss.py:136
    features = torch.cat([extra["incident_lights"].mean(-2), extra["local_incident_lights"].mean(-2),
                          extra["incident_visibility"].mean(-2)], dim=-1)            # svgss.py:148-151
    nv = (normal @ t("eval_settings_viewmatrix")[:3, :3]).transpose(1, 2).reshape(normal.shape[0], -1)
    vfeatures = torch.cat([brdf, base, nv, rough, extra["direct"], extra["indirect"]], dim=-1)
    _close("features", features, g["eval_features"], tol=2e-4)
    _close("vfeatures", vfeatures, g["eval_vfeatures"], tol=2e-3)
    # the training branch consumes the same dict without concatenation (svgss.py:143-147, 157-158)
    with torch.no_grad():
        b, e = shading.rendering_equation4(base, rough, normal, torch.nn.functional.normalize(t("train_settings_campos") - t("pc_get_xyz"), dim=-1),
                                           inc, light, visibility_precompute=vis, incident_dirs_precompute=dirs,
                                           incident_areas_precompute=areas)
    f_tr = torch.cat([e["incident_visibility"].mean(-2), e["local_incident_lights"].mean(-2)], dim=-1)
    _close("train_features", f_tr, g["train_features"], tol=2e-4)
    # and the fused packing kernel against the same recording
    for tag, training in (("train", True), ("eval", False)):
        vd = torch.nn.functional.normalize(t(f"{tag}_settings_campos") - t("pc_get_xyz"), dim=-1)
        with torch.no_grad():
            f, vf, _ = shading.shade_and_pack(base, rough, normal, vd, inc, light, vis, dirs, areas, t(f"{tag}_settings_viewmatrix"), training)
        _close(tag + "_packed_features", f, g[f"{tag}_features"], tol=2e-4)
        _close(tag + "_packed_vfeatures", vf, g[f"{tag}_vfeatures"], tol=2e-3)


def test_incident_direction_lattice_matches_reference(built):
    """svgir_incident_dirs against the reference's fibonacci_sphere_sampling / sample_incident_rays
    outputs (tests/golden/incident_dirs.npz), evaluation lattice and training lattice with the recorded random offsets."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "incident_dirs.npz"))
    normals = torch.from_numpy(g["normals"]).to(dev)
    for Ns in (8, 64, 384):
        lat = shading.FibonacciLattice(normals, Ns)
        d, a = lat.dirs(), lat.areas()
        assert tuple(d.shape) == (300, Ns, 3) and tuple(a.shape) == (300, Ns, 1)
        np.testing.assert_allclose(d.cpu().numpy(), g[f"dirs_{Ns}"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(a.cpu().numpy(), g[f"areas_{Ns}"], rtol=1e-6)
    lat = shading.FibonacciLattice(normals, 24, offsets=torch.from_numpy(g["sample_train_24_offsets"]).to(dev))
    np.testing.assert_allclose(lat.dirs().cpu().numpy(), g["sample_train_24"], rtol=0, atol=3e-6)
    d, a = shading.sample_incident_rays(normals, is_training=False, sample_num=24)
    np.testing.assert_allclose(d.cpu().numpy(), g["sample_eval_24"], rtol=0, atol=2e-6)
    d, a = shading.sample_incident_rays(normals, is_training=True, sample_num=24)      # random offsets: unit, upper hemisphere
    assert torch.allclose(d.norm(dim=-1), torch.ones_like(d[..., 0]), atol=1e-5)
    assert ((d * normals[:, None]).sum(-1) > 0.1).all()


@pytest.mark.parametrize("training,Ns", [(True, 64), (False, 384), (True, 7)])
def test_shading_with_in_kernel_directions_equals_streamed_directions(built, training, Ns):
    """The shading kernels fed with a FibonacciLattice (directions built in registers, areas = 2 pi) against the same
    kernels fed with the materialised [P,Ns,3] / [P,Ns,1] tensors, forward and backward."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(400, Ns, 11, rough_lo=0.3)
    geo = torch.nn.functional.normalize(d["normals"][:, 0].float(), dim=-1).to(dev)
    offs = torch.rand(400, device=dev) * 6.2831855 if training else None
    lat = shading.FibonacciLattice(geo, Ns, offs)
    dirs, areas = lat.dirs(), lat.areas()
    names = ("base", "rough", "normals", "radiance", "env")
    res = []
    for mode in ("lattice", "streamed"):
        lv = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
        vm = torch.eye(4, device=dev)
        f, vf, red = shading.shade_and_pack(lv["base"], lv["rough"], lv["normals"], d["viewdirs"].float().to(dev), lv["radiance"],
                                            _Light(lv["env"]), d["vis"].float().to(dev),
                                            lat if mode == "lattice" else dirs, None if mode == "lattice" else areas, vm, training)
        (f.sum() + (vf * vf).sum() + red[:, :60].sum()).backward()
        res.append((f, vf, red, {k: lv[k].grad for k in names}))
    for a_, b_ in zip(res[0][:3], res[1][:3]):
        _close("fwd", a_, b_, tol=2e-5)
    for k in names:
        _close("grad_" + k, res[0][3][k], res[1][3][k], tol=5e-5)


def test_envlight_resample_matches_reference(built):
    """EnvLight.direct_light's 32x64 down-sample (scene/envmap.py:62-63) as a HIP kernel, and the lookups through it,
    against the outputs of the reference's own class (tests/golden/lights.npz)."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lights.npz"))

    class EnvLight:
        def __init__(self, envmap, transform=None):
            self.envmap, self.transform = envmap, transform

    env, softplus, scale, tr = shading._env_of(EnvLight(torch.from_numpy(g["el_envmap"]).to(dev)))
    assert not softplus and scale == 1.0 and tr is None
    np.testing.assert_allclose(env.cpu().numpy(), g["el_resampled"], rtol=2e-6, atol=2e-7)
    # lookups: a single white surfel whose `global light` mean is the env lookup itself is awkward; use the oracle path
    got = so.env_lookup(env.double().cpu(), torch.from_numpy(g["dirs"]).double(), softplus=False, scale=1.0)
    np.testing.assert_allclose(got.numpy(), g["el_light"], rtol=2e-5, atol=2e-6)


def test_shading_cfg5_scale_in_chunks(built):
    """BASELINE configs[4]'s other half: the shading kernels at P = 2 000 000 surfels, Ns = 384 incident directions (66 % of
    the cfg5 step), in 4 chunks of 500 000 surfels like the reference's evaluation chunk loop (svgss.py:121-136), incident
    directions generated in the kernels (Fibonacci lattice, f1).  A random subset of 20 000 surfels (5 000 per chunk) is
    recomputed by the fp64 oracle on the CPU from the oracle's own lattice (oracle/epilogue_oracle.py) and compared:
    pbr / direct / indirect / diffuse_light / specular and the packed evaluation features."""
    from gaussian_renderer import shading
    from oracle import epilogue_oracle as eo
    dev = torch.device("cuda:0")
    P, Ns, chunk, per = 2_000_000, 384, 500_000, 5_000
    g = torch.Generator(device=dev).manual_seed(17)
    env = 3.0 * torch.rand(1, 32, 64, 3, generator=g, device=dev)
    view = torch.linalg.qr(torch.randn(3, 3, dtype=torch.float64))[0]
    vm = torch.eye(4, dtype=torch.float64)
    vm[:3, :3] = view
    vmf = vm.float().to(dev)
    worst = {}
    for c in range(P // chunk):
        rnd = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
        geo = torch.nn.functional.normalize(rnd(chunk, 3), dim=-1)
        base = torch.sigmoid(rnd(chunk, 12)) * 0.77 + 0.03
        rough = torch.sigmoid(rnd(chunk, 4)) * 0.69 + 0.3            # (>= 0.3: the fp32-well-conditioned range, see the module docstring)
        normals = torch.nn.functional.normalize(geo[:, None] + 0.1 * rnd(chunk, 4, 3), dim=-1)
        viewdirs = torch.nn.functional.normalize(geo + 0.5 * rnd(chunk, 3), dim=-1)
        vis = (torch.rand(chunk, Ns, 1, generator=g, device=dev) > 0.3).float()
        radiance = (0.2 * rnd(chunk, Ns, 3)).abs_()
        lat, _ = shading.sample_incident_rays(geo, False, Ns, materialize=False)
        with torch.no_grad():
            pbr, ex = shading.rendering_equation4(base, rough, normals, viewdirs, radiance, _Light(env), visibility_precompute=vis,
                                                  incident_dirs_precompute=lat, incident_areas_precompute=None)
            f, vf, _ = shading.shade_and_pack(base, rough, normals, viewdirs, radiance, _Light(env), vis, lat, None, vmf, False)
        torch.cuda.synchronize()
        assert pbr.shape == (chunk, 12) and f.shape == (chunk, 7) and vf.shape == (chunk, 64)
        pick = torch.randperm(chunk, generator=torch.Generator().manual_seed(c))[:per].to(dev)
        cpu = lambda t: t[pick].double().cpu()  # noqa: E731
        dn, an = eo.fibonacci_dirs(geo[pick].float().cpu().numpy(), Ns)
        dirs, areas = torch.from_numpy(dn), torch.from_numpy(an)
        ref = so.shade(cpu(base), cpu(rough), cpu(normals), cpu(viewdirs), cpu(radiance), cpu(vis), dirs, areas, env.double().cpu())
        fr, vr = so.pack(ref, cpu(base), cpu(rough), cpu(normals), view, False)
        _close(f"chunk{c} pbr", pbr[pick], ref["pbr"])
        for k in ("diffuse_light", "specular", "direct", "indirect"):
            _close(f"chunk{c} {k}", ex[k][pick], ref[k])
        _close(f"chunk{c} features", f[pick], fr)
        _close(f"chunk{c} vfeatures", vf[pick], vr)
        del radiance, vis, pbr, ex, f, vf
        torch.cuda.empty_cache()


@pytest.mark.parametrize("api", ["rendering_equation4", "shade_and_pack"])
def test_radiance_ratio_equals_the_reference_product(built, api):
    """`radiance_ratio=` (ABI 13) = the reference's get_radiances, nan_to_num(_radiances.detach() * _radiance_ratio, nan=0)
    (scene/gaussian_model.py:323-324), formed inside the kernels: the outputs and every material gradient must be BIT-identical to
    the calls fed with the product torch computes, the scalar's gradient must be what autograd returns for the product form (and what
    autograd of the fp64 oracle returns), and no [n, Ns, 3] gradient is produced for the detached cache."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(257, 96, 41, rough_lo=0.3)
    raw = d["radiance"].float().to(dev)
    names = ("base", "rough", "normals", "env")
    c = {k: d[k].float().to(dev) for k in ("viewdirs", "vis", "dirs", "areas")}
    vm = torch.eye(4, device=dev)
    g = torch.Generator().manual_seed(9)

    def run(fused):
        lg = {k: d[k].float().to(dev).requires_grad_(True) for k in names}
        ratio = torch.tensor(0.83, device=dev, requires_grad=True)
        if fused:
            rad, kw = raw, dict(radiance_ratio=ratio)
        else:
            rad, kw = torch.nan_to_num(raw.detach() * ratio, nan=0.0), {}
        if api == "rendering_equation4":
            pbr, ex = shading.rendering_equation4(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], rad, _Light(lg["env"]),
                                                  visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                                  incident_areas_precompute=c["areas"], **kw)
            outs = [pbr, ex["diffuse_light"], ex["specular"], ex["indirect"], ex["local_incident_lights"].mean(-2)]
        else:
            outs = list(shading.shade_and_pack(lg["base"], lg["rough"], lg["normals"], c["viewdirs"], rad, _Light(lg["env"]), c["vis"],
                                               c["dirs"], c["areas"], vm, True, **kw))
        return lg, ratio, outs

    # forward with non-finite cache entries: nan_to_num -> 0, +max, -max (their gradient w.r.t. the scalar is NaN in torch as well:
    # 0 * NaN -- hence the gradient checks below run on a finite cache)
    finite_raw = raw
    raw = finite_raw.clone()
    raw[3, 5, 1] = float("nan"); raw[7, 0, 0] = float("inf"); raw[9, 70, 2] = -float("inf")
    with torch.no_grad():
        for i, (oa, ob) in enumerate(zip(run(True)[2], run(False)[2])):
            assert torch.equal(oa, ob), f"output {i} (non-finite cache entries) differs from the product form"
    raw = finite_raw
    lg_a, ratio_a, out_a = run(True)
    lg_b, ratio_b, out_b = run(False)
    ws = [torch.randn(o.shape, generator=g).to(dev) for o in out_a]
    for i, (oa, ob) in enumerate(zip(out_a, out_b)):
        assert torch.equal(oa, ob), f"output {i} differs from the product form"
    sum((o * w).sum() for o, w in zip(out_a, ws)).backward()
    sum((o * w).sum() for o, w in zip(out_b, ws)).backward()
    for k in names:
        if k == "env":   # (float atomics: order of the adds)
            _close("grad_env", lg_a[k].grad, lg_b[k].grad, tol=1e-5)
        else:
            assert torch.equal(lg_a[k].grad, lg_b[k].grad), f"grad_{k} differs from the product form"
    ga, gb = float(ratio_a.grad), float(ratio_b.grad)
    assert np.isfinite(ga) and abs(ga - gb) <= 2e-5 * max(abs(gb), 1e-6) + 1e-6, (ga, gb)
    # ... and against autograd of the fp64 oracle on the cleaned product
    r64 = torch.tensor(0.83, dtype=torch.float64, requires_grad=True)
    inc = raw.double().cpu() * r64
    if api == "rendering_equation4":
        ref = so.shade(d["base"], d["rough"], d["normals"], d["viewdirs"], inc, d["vis"], d["dirs"], d["areas"], d["env"])
        ro = [ref["pbr"], ref["diffuse_light"], ref["specular"], ref["indirect"], ref["mean_local"]]
        sum((o * w.double().cpu().reshape(o.shape)).sum() for o, w in zip(ro, ws)).backward()
        assert abs(ga - float(r64.grad)) <= 5e-4 * max(abs(float(r64.grad)), 1e-6) + 1e-5, (ga, float(r64.grad))


def test_radiance_ratio_can_also_return_the_cache_gradient(built):
    """A caller that does differentiate the cache gets dL/d(raw radiance) = dL/d(incident) * ratio where the product is finite."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(130, 64, 43, rough_lo=0.3)
    c = {k: d[k].float().to(dev) for k in ("base", "rough", "normals", "viewdirs", "vis", "dirs", "areas", "env")}
    raw = d["radiance"].float().to(dev)
    raw[2, 3, 0] = float("nan")

    def run(fused):
        r = raw.clone().requires_grad_(True)
        ratio = torch.tensor(1.7, device=dev, requires_grad=True)
        rad, kw = (r, dict(radiance_ratio=ratio)) if fused else (torch.nan_to_num(r * ratio, nan=0.0), {})
        pbr, ex = shading.rendering_equation4(c["base"], c["rough"], c["normals"], c["viewdirs"], rad, _Light(c["env"]),
                                              visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                              incident_areas_precompute=c["areas"], **kw)
        (pbr.sum() + 0.3 * ex["specular"].sum()).backward()
        return r.grad, ratio.grad

    (ga, gra), (gb, grb) = run(True), run(False)
    assert torch.equal(ga, gb) and float(ga[2, 3, 0]) == 0.0   # (no gradient through the scrubbed entry)
    # the scalar: torch adds 0 * NaN = NaN for the scrubbed entry; the kernel's sum skips it and stays finite
    assert not bool(torch.isfinite(grb)) and bool(torch.isfinite(gra))
    r2 = raw.clone(); r2[2, 3, 0] = 0.0
    ratio = torch.tensor(1.7, device=dev, requires_grad=True)
    pbr, ex = shading.rendering_equation4(c["base"], c["rough"], c["normals"], c["viewdirs"], torch.nan_to_num(r2 * ratio, nan=0.0),
                                          _Light(c["env"]), visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                          incident_areas_precompute=c["areas"])
    (pbr.sum() + 0.3 * ex["specular"].sum()).backward()
    assert abs(float(gra) - float(ratio.grad)) <= 2e-5 * abs(float(ratio.grad)) + 1e-6


def test_radiance_ratio_zero_still_has_a_gradient(built):
    """ratio == 0: the incident radiance vanishes, the scalar's gradient does not (sum dL/d(incident) * raw)."""
    from gaussian_renderer import shading
    dev = torch.device("cuda:0")
    d = _random_case(90, 70, 47, rough_lo=0.3)
    c = {k: d[k].float().to(dev) for k in ("base", "rough", "normals", "viewdirs", "vis", "dirs", "areas", "env")}
    raw = d["radiance"].float().to(dev)
    out = []
    for fused in (True, False):
        ratio = torch.zeros((), device=dev, requires_grad=True)
        rad, kw = (raw, dict(radiance_ratio=ratio)) if fused else (torch.nan_to_num(raw * ratio, nan=0.0), {})
        pbr, ex = shading.rendering_equation4(c["base"], c["rough"], c["normals"], c["viewdirs"], rad, _Light(c["env"]),
                                              visibility_precompute=c["vis"], incident_dirs_precompute=c["dirs"],
                                              incident_areas_precompute=c["areas"], **kw)
        (pbr.sum() + 0.3 * ex["indirect"].sum()).backward()
        out.append((pbr.detach(), float(ratio.grad)))
    assert torch.equal(out[0][0], out[1][0])
    assert abs(out[1][1]) > 1e-3 and abs(out[0][1] - out[1][1]) <= 2e-5 * abs(out[1][1]) + 1e-6, (out[0][1], out[1][1])
