"""Shading only the surfels a view reads (include/svgir_raster.h: svgir_shade_params.subset, svgir_fused_shade).

The reference shades all P surfels before every rasterizer call (gaussian_renderer/svgss.py:116-141) although the call reads the
packed rows of the surfels that survive its culls only (svgss.py:143-182; svgss forward.cu:267-395).  The fused path must be a pure
optimisation: images and every gradient BIT-IDENTICAL to shade_and_pack(...) + GaussianRasterizer(...) over all P -- except
dL/d(env), a sum of float atomics whose order differs (compared with a tolerance)."""
import numpy as np
import pytest
import torch

from gaussian_renderer import _native as N
from gaussian_renderer import shading
from gaussian_renderer.svgss_rasterization import GaussianRasterizer
from svgir_harness import runner, scenes, shade_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MAT = ("base_color", "roughness", "normals", "radiance", "env")


def _geo_normals(rotations):
    q = torch.nn.functional.normalize(rotations, dim=-1)
    r, x, y, z = q.unbind(-1)
    return torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], dim=-1)


def _materials(sct, st, Ns, training, seed, lattice=True):
    dev = sct["means3D"].device
    P = sct["means3D"].shape[0]
    geo_n = _geo_normals(sct["rotations"])
    d = shade_inputs.make(P, Ns, seed=seed, device=dev, geo_normals=geo_n, with_dirs=not lattice)
    d["viewdirs"] = torch.nn.functional.normalize(st.campos[None, :] - sct["means3D"], dim=-1)
    if lattice:
        offs = torch.rand(P, device=dev, generator=torch.Generator(dev).manual_seed(seed)) * 6.2831855 if training else None
        d["dirs"], d["areas"] = shading.FibonacciLattice(torch.nn.functional.normalize(geo_n, dim=-1), Ns, offs), None
    return d


def _unfused(sct, st, d, leaves, training):
    f, vf, _ = shading.shade_and_pack(leaves["base_color"], leaves["roughness"], leaves["normals"], d["viewdirs"], leaves["radiance"],
                                      shade_inputs.Light(leaves["env"]), d["visibility"], d["dirs"], d["areas"], st.viewmatrix, training)
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    out = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=means2D, opacities=leaves["opacities"], shs=leaves["shs"],
                                 scales=leaves["scales"], rotations=leaves["rotations"], features=f, vfeatures=vf)
    return out, means2D, (f, vf)


def _fused(sct, st, d, leaves, training, **kw):
    means2D = torch.zeros_like(leaves["means3D"], requires_grad=True)
    out, red = shading.render_shaded(st, leaves["means3D"], means2D, leaves["opacities"], leaves["shs"], leaves["scales"],
                                     leaves["rotations"], leaves["base_color"], leaves["roughness"], leaves["normals"], d["viewdirs"],
                                     leaves["radiance"], shade_inputs.Light(leaves["env"]), d["visibility"], d["dirs"], d["areas"],
                                     training, **kw)
    return out, means2D, red


def _leaves(sct, d, grad_mat=True):
    lv = {k: sct[k].detach().clone().requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    lv.update({k: d[k].detach().clone().requires_grad_(grad_mat) for k in MAT})
    return lv


def _loss(out, gt):
    (R, color, normal, opacity, depth, feature, vfeature, weights, radii) = out
    return ((color * gt["color"]).sum() + (normal * gt["normal"]).sum() + (depth * gt["depth"]).sum() + (opacity * gt["opacity"]).sum() +
            (feature * gt["feature"]).sum() + (vfeature * gt["vfeature"]).sum())


def _compare(sct, st, d, grads, training, grad_mat=True):
    gt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in grads.items()}
    la, lb = _leaves(sct, d, grad_mat), _leaves(sct, d, grad_mat)
    oa, m2a, (f, vf) = _unfused(sct, st, d, la, training)
    _loss(oa, gt).backward()
    ob, m2b, _ = _fused(sct, st, d, lb, training)
    _loss(ob, gt).backward()
    torch.cuda.synchronize()
    assert oa[0] == ob[0]
    names = ("num_rendered", "color", "normal", "opacity", "depth", "feature", "vfeature", "weights", "radii")
    for n, a, b in zip(names[1:], oa[1:], ob[1:]):
        if n == "weights":   # float atomics across sub-tiles
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), n
        else:
            assert torch.equal(a, b), n
    assert torch.equal(m2a.grad, m2b.grad)
    for k in la:
        if la[k].grad is None:
            assert lb[k].grad is None, k
            continue
        if k == "env":
            ga, gb = la[k].grad, lb[k].grad
            assert torch.allclose(ga, gb, rtol=2e-5, atol=2e-6 * float(ga.abs().max())), (k, float((ga - gb).abs().max()))
        else:
            assert torch.equal(la[k].grad, lb[k].grad), k
    return oa, (f, vf)


def _partition(mask):
    """The layout of svgir_shade_params.subset: selected ids (index order) in front, the others from the back."""
    ids = torch.arange(mask.numel(), device=mask.device, dtype=torch.int32)
    return torch.cat([ids[mask], ids[~mask].flip(0)]).contiguous(), mask.sum().to(torch.int32).reshape(1).contiguous()


@pytest.mark.parametrize("Ns,training", [(64, True), (24, True), (200, False), (7, False)])
def test_shade_subset_rows_equal_the_all_surfel_call(built, Ns, training):
    """svgir_shade_forward / _backward with svgir_shade_params.subset: rows of the subset bit-identical to the all-P call (both
    forward kernels: quad layout below 128 samples, one wave per surfel above), all other rows zero, every output written."""
    dev = torch.device(DEV)
    P = 3001
    d = shade_inputs.make(P, Ns, seed=11, device=dev)
    vm = torch.eye(4, device=dev) + 0.1 * torch.randn(4, 4, device=dev, generator=torch.Generator(dev).manual_seed(1))
    mask = torch.rand(P, device=dev, generator=torch.Generator(dev).manual_seed(2)) < 0.37
    lst, cnt = _partition(mask)
    S, VS = (4, 52) if training else (7, 64)

    def run(subset):
        sp, keep, *_ = shading._params(d["base_color"], d["roughness"], d["normals"], d["viewdirs"], d["radiance"], d["visibility"],
                                       d["dirs"], d["areas"], d["env"], True, 2.0, viewmatrix=vm, training=training)
        if subset:
            sp.subset, sp.subset_count = lst.data_ptr(), cnt.data_ptr()
        red, f, vf = (N.out_tensor(s, torch.float32, dev) for s in ((P, 70), (P, S), (P, VS)))
        N.check(N.lib.svgir_shade_forward(sp, red.data_ptr(), f.data_ptr(), vf.data_ptr(), N.stream_ptr(dev)), "shade_forward")
        g = torch.Generator(dev).manual_seed(5)
        gf, gvf = torch.randn(P, S, device=dev, generator=g), torch.randn(P, VS, device=dev, generator=g)
        outs = [N.out_tensor(t.shape, torch.float32, dev) for t in (d["base_color"], d["roughness"], d["normals"], d["radiance"], d["env"])]
        gwork = torch.empty(d["env"].numel(), device=dev)
        N.check(N.lib.svgir_shade_backward(sp, None, gf.data_ptr(), gvf.data_ptr(), *[t.data_ptr() for t in outs], gwork.data_ptr(),
                                           None, N.stream_ptr(dev)), "shade_backward")
        torch.cuda.synchronize()
        return [red, f, vf] + outs

    full, sub = run(False), run(True)
    for i, (a, b) in enumerate(zip(full[:7], sub[:7])):
        assert torch.equal(a[mask], b[mask]), i
        assert not torch.isnan(b).any() and float(b[~mask].abs().max()) == 0.0, i
    assert torch.isfinite(sub[7]).all()   # dL_denv of the subset: the subset's contributions only (checked end to end below)


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("lattice", [True, False])
def test_fused_view_is_bit_identical_small(built, training, lattice):
    dev = torch.device(DEV)
    S, VS = (4, 52) if training else (7, 64)
    sc = scenes.surface_scene(P=6000, W=176, H=144, seed=41, sh_degree=2, variant="svgss", S=S, VS=VS, scale_lo=0.01, scale_hi=0.06)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = _materials(sct, st, 64 if training else 136, training, seed=3, lattice=lattice)
    for _ in range(3):   # (the third view of a workload runs the speculative launch sequence)
        oa, (f, vf) = _compare(sct, st, d, scenes.upstream_grads(sc, "svgss", seed=6), training)
    # what the fused forward left in the packed rows: the all-P values where the composite can read them, zeros elsewhere
    means2D = torch.zeros_like(sct["means3D"])
    fs, keep = shading.fused_shade(d["base_color"], d["roughness"], d["normals"], d["viewdirs"], d["radiance"], shade_inputs.Light(d["env"]),
                                   d["visibility"], d["dirs"], d["areas"], st.viewmatrix, training)
    from gaussian_renderer.svgss_rasterization import _C
    empty = torch.empty(0, device=dev)
    f2, vf2 = N.out_tensor((6000, S), torch.float32, dev), N.out_tensor((6000, VS), torch.float32, dev)
    out = _C.rasterize_gaussians(st.bg, sct["means3D"], f2, vf2, empty, sct["opacities"], sct["scales"], sct["rotations"], st.scale_modifier,
                                 empty, st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height,
                                 st.image_width, sct["shs"], st.sh_degree, st.campos, False, False, st.config, shade=fs)
    torch.cuda.synchronize()
    weights, radii = out[7][:, 0], out[8]
    shaded = vf2.abs().sum(1) > 0
    assert not torch.isnan(f2).any() and not torch.isnan(vf2).any()
    assert torch.equal(f2[shaded], f.detach()[shaded]) and torch.equal(vf2[shaded], vf.detach()[shaded])
    assert bool((shaded | (weights == 0)).all()), "every surfel that was blended was shaded"
    assert bool((radii[shaded] > 0).all()), "only surfels that pass the preprocess culls are shaded"
    frac = float(shaded.float().mean())
    assert 0.05 < frac < float((radii > 0).float().mean()) + 1e-6, frac


def test_fused_all_surfels_and_reduced_gradient(built):
    """all_surfels=True: the reference's semantics (every surfel shaded; `reduced` usable in a loss of its own, svgss.py:359-364)."""
    dev = torch.device(DEV)
    sc = scenes.surface_scene(P=3000, W=128, H=96, seed=43, sh_degree=1, variant="svgss", S=4, VS=52, scale_lo=0.01, scale_hi=0.06)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = _materials(sct, st, 32, True, seed=4)
    gt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in scenes.upstream_grads(sc, "svgss", seed=8).items()}
    wr = torch.randn(3000, 70, device=dev, generator=torch.Generator(dev).manual_seed(3))
    la, lb = _leaves(sct, d), _leaves(sct, d)
    f, vf, red_a = shading.shade_and_pack(la["base_color"], la["roughness"], la["normals"], d["viewdirs"], la["radiance"],
                                          shade_inputs.Light(la["env"]), d["visibility"], d["dirs"], d["areas"], st.viewmatrix, True)
    m2 = torch.zeros_like(la["means3D"], requires_grad=True)
    oa = GaussianRasterizer(st)(means3D=la["means3D"], means2D=m2, opacities=la["opacities"], shs=la["shs"], scales=la["scales"],
                                rotations=la["rotations"], features=f, vfeatures=vf)
    (_loss(oa, gt) + (red_a * wr).sum()).backward()
    ob, _, red_b = _fused(sct, st, d, lb, True, all_surfels=True, want_reduced=True)
    (_loss(ob, gt) + (red_b * wr).sum()).backward()
    assert torch.equal(red_a, red_b)
    for k in MAT:
        if k == "env":
            assert torch.allclose(la[k].grad, lb[k].grad, rtol=2e-5, atol=2e-6 * float(la[k].grad.abs().max()))
        else:
            assert torch.equal(la[k].grad, lb[k].grad), k
    with pytest.raises(RuntimeError):   # a loss on `reduced` without all_surfels: rows of unshaded surfels would be missing
        oc, _, red_c = _fused(sct, st, d, _leaves(sct, d), True, want_reduced=True)
        (_loss(oc, gt) + (red_c * wr).sum()).backward()


def test_listed_surfel_without_gradient_rows_reads_zeros(built):
    """The shading backward reads the dL_dfeatures / dL_dvfeatures row of every surfel with out_weights > 0; those two tensors live
    OUTSIDE the cleared gradient allocation (scratch_feature_grads) and are NaN-poisoned here.  The rows come from the composite
    backward's own alpha / transmittance replay: a surfel that is listed but owns no row (a 1-ulp disagreement between the two
    evaluations, an undersized row scratch) must read zeros, not uninitialised memory.  Forced here by marking surfels the forward did
    NOT blend -- a culled one (radius 0) and visible ones with weight 0 -- as blended before the backward runs."""
    assert N.POISON
    dev = torch.device(DEV)
    sc = scenes.surface_scene(P=5000, W=160, H=128, seed=47, sh_degree=1, variant="svgss", S=4, VS=52, scale_lo=0.01, scale_hi=0.06)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = _materials(sct, st, 32, True, seed=9)
    gt = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in scenes.upstream_grads(sc, "svgss", seed=10).items()}

    def run(tamper):
        lv = _leaves(sct, d)
        out, m2, _ = _fused(sct, st, d, lv, True)
        weights, radii = out[7], out[8]
        extra = torch.zeros(5000, dtype=torch.bool, device=dev)
        if tamper:
            culled = torch.nonzero(radii == 0)[:3, 0]
            unblended = torch.nonzero((radii > 0) & (weights[:, 0] == 0))[:40, 0]
            assert culled.numel() == 3   # (visible surfels without a blend weight exist in denser scenes only; taken along when there are any)
            extra[culled] = True; extra[unblended] = True
            weights.data[extra] = 1.0   # (.data: the saved tensor's version counter must not change)
        _loss(out, gt).backward()
        torch.cuda.synchronize()
        return lv, m2, extra

    (la, m2a, _), (lb, m2b, extra) = run(False), run(True)
    for k in list(la) + ["means2D"]:
        ga, gb = (m2a.grad, m2b.grad) if k == "means2D" else (la[k].grad, lb[k].grad)
        assert torch.isfinite(gb).all(), k
        if k == "env":
            assert torch.allclose(ga, gb, rtol=2e-5, atol=2e-6 * float(ga.abs().max()))
        else:
            assert torch.equal(ga, gb), k   # the extra surfels contribute exactly nothing
    for k in ("base_color", "roughness", "normals"):
        assert float(lb[k].grad[extra].abs().max()) == 0.0, k


@pytest.mark.parametrize("name", ["cfg3_train", "cfg3_eval"])
def test_fused_view_is_bit_identical_at_baseline_size(built, name):
    """BASELINE configs[2] at full size (P = 200 000, 800 x 800; Ns = 64 at the training widths with the in-kernel lattice, 384 at the
    evaluation widths): images and all gradients of the fused path against the all-P path."""
    dev = torch.device(DEV)
    sc = scenes.make(name)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    training = name == "cfg3_train"
    d = _materials(sct, st, 64 if training else 384, training, seed=5)
    oa, (f, vf) = _compare(sct, st, d, scenes.upstream_grads(sc, "svgss"), training, grad_mat=training)
    w = oa[7][:, 0]
    print(f"\n{name}: radii > 0: {float((oa[8] > 0).float().mean()):.3f}  weights > 0: {float((w > 0).float().mean()):.3f}")


def test_fused_forward_is_bit_identical_at_cfg5(built):
    """BASELINE configs[4]: 2 M surfels, 1600 x 1600, Ns = 384, evaluation widths, forward (the shading there is 74 % of the step when
    all 2 M surfels are shaded; 13 % of them receive a blend weight)."""
    dev = torch.device(DEV)
    sc = scenes.make("cfg5")
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = _materials(sct, st, 384, False, seed=5)
    with torch.no_grad():
        lv = {k: sct[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        lv.update({k: d[k] for k in MAT})
        f, vf, _ = shading.shade_and_pack(d["base_color"], d["roughness"], d["normals"], d["viewdirs"], d["radiance"], shade_inputs.Light(d["env"]),
                                          d["visibility"], d["dirs"], d["areas"], st.viewmatrix, False)
        m2 = torch.zeros_like(sct["means3D"])
        oa = GaussianRasterizer(st)(means3D=sct["means3D"], means2D=m2, opacities=sct["opacities"], shs=sct["shs"], scales=sct["scales"],
                                    rotations=sct["rotations"], features=f, vfeatures=vf)
        del f, vf
        ob, _ = shading.render_shaded(st, sct["means3D"], m2, sct["opacities"], sct["shs"], sct["scales"], sct["rotations"], d["base_color"],
                                      d["roughness"], d["normals"], d["viewdirs"], d["radiance"], shade_inputs.Light(d["env"]), d["visibility"],
                                      d["dirs"], d["areas"], False)
    torch.cuda.synchronize()
    assert oa[0] == ob[0]
    for i in (1, 2, 3, 4, 5, 6, 8):
        assert torch.equal(oa[i], ob[i]), i
