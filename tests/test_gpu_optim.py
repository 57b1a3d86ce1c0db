"""csrc/optim.hip: the fused Adam step, the densification statistics and the fused row compaction, against
PyTorch's own implementations of what the reference calls (torch.optim.Adam with the reference's settings, boolean-mask
indexing, torch.norm) evaluated in fp64 on the CPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(dev, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    # the reference's parameter block (scene/gaussian_model.py:745-768): names, shapes per Gaussian, learning rates
    spec = [("xyz", (3,), 1.6e-4), ("normal", (3,), 1e-3), ("rotation", (4,), 1e-3), ("scaling", (3,), 5e-3), ("opacity", (1,), 5e-2),
            ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 2.5e-3 / 20), ("base_color", (4, 3), 1e-2), ("roughness", (4, 1), 1e-2),
            ("incidents_dc", (1, 3), 2e-3), ("incidents_rest", (15, 3), 1e-4), ("visibility_dc", (1, 1), 2.5e-3),
            ("visibility_rest", (15, 1), 1.25e-4)]
    P = 3001
    return [{"params": [torch.nn.Parameter(torch.randn((P,) + shp, generator=g).to(device=dev, dtype=dtype))], "lr": lr, "name": n}
            for n, shp, lr in spec], g, P


def test_fused_adam_matches_torch_adam(built):
    from svgir_harness.optim import FusedAdam
    dev = torch.device("cuda:0")
    mine, g1, P = _groups(dev, torch.float32, 5)
    ref, g2, _ = _groups("cpu", torch.float64, 5)
    opt = FusedAdam(mine, lr=1e-4, eps=1e-15)
    oref = torch.optim.Adam(ref, lr=1e-4, eps=1e-15)
    gen = torch.Generator().manual_seed(9)
    for it in range(25):
        for gm, gr in zip(opt.param_groups, oref.param_groups):
            grad = torch.randn(gr["params"][0].shape, generator=gen, dtype=torch.float64) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=gen)))
            if it % 7 == 3 and gm["name"] == "normal":
                gm["params"][0].grad = None; gr["params"][0].grad = None   # a group without a gradient is skipped, its step count stays
                continue
            gm["params"][0].grad = grad.to(device=dev, dtype=torch.float32)
            gr["params"][0].grad = grad.to(torch.float32).to(torch.float64)
        if it == 10:   # learning-rate schedule on the position group (update_learning_rate)
            opt.param_groups[0]["lr"] = oref.param_groups[0]["lr"] = 8e-5
        opt.step(); oref.step()
    for gm, gr in zip(opt.param_groups, oref.param_groups):
        p, q = gm["params"][0], gr["params"][0]
        st, sr = opt.state[p], oref.state[q]
        assert float(st["step"]) == float(sr["step"]), gm["name"]
        for a, b, what in ((p, q, "param"), (st["exp_avg"], sr["exp_avg"], "exp_avg"), (st["exp_avg_sq"], sr["exp_avg_sq"], "exp_avg_sq")):
            a = a.detach().double().cpu().numpy(); b = b.detach().numpy()
            # fp32 state against the fp64 evaluation of the same recurrence (25 steps of rounding; the gradients span eight
            # decades with random signs, so small moments carry the cancellation error of large ones: the criterion is
            # normalised by the tensor's scale, plus a loose element-wise one)
            scale = np.abs(b).max()
            assert np.abs(a - b).max() <= 3e-6 * scale, (gm["name"], what, np.abs(a - b).max() / scale)
            err = np.abs(a - b) / (np.abs(b) + 1e-6 * scale)
            assert np.quantile(err, 0.99) < 1e-4, (gm["name"], what, np.quantile(err, 0.99))


def test_fused_adam_state_layout_survives_the_reference_prune(built):
    """The reference swaps parameters and indexes the moments inside the optimizer (scene/gaussian_model.py:1020-1036); the
    same statements run on FusedAdam, and prune_rows gives what they give."""
    from svgir_harness.optim import FusedAdam, prune_rows
    dev = torch.device("cuda:0")
    groups, g, P = _groups(dev, torch.float32, 6)
    opt = FusedAdam(groups, lr=1e-4, eps=1e-15)
    for grp in opt.param_groups:
        grp["params"][0].grad = torch.randn_like(grp["params"][0])
    opt.step()
    mask = (torch.rand(P, generator=g) > 0.37).to(dev)
    flat, names = [], []
    for grp in opt.param_groups:
        p = grp["params"][0]
        flat += [p.detach(), opt.state[p]["exp_avg"], opt.state[p]["exp_avg_sq"]]
        names += [grp["name"]] * 3
    book = [torch.rand(P, 1, device=dev), torch.randint(0, 9, (P,), device=dev, dtype=torch.int32)]
    pruned = prune_rows(flat + book, mask)
    for t, q in zip(flat + book, pruned):
        assert torch.equal(t[mask], q)
    # the reference's own statements
    for group in opt.param_groups:
        stored_state = opt.state.get(group["params"][0], None)
        assert stored_state is not None
        stored_state["exp_avg"] = stored_state["exp_avg"][mask]
        stored_state["exp_avg_sq"] = stored_state["exp_avg_sq"][mask]
        del opt.state[group["params"][0]]
        group["params"][0] = torch.nn.Parameter((group["params"][0][mask].requires_grad_(True)))
        opt.state[group["params"][0]] = stored_state
    for grp in opt.param_groups:
        grp["params"][0].grad = torch.randn_like(grp["params"][0])
    opt.step()
    assert all(float(opt.state[grp["params"][0]]["step"]) == 2.0 for grp in opt.param_groups)
    # edge cases: nothing kept / everything kept / empty input
    none = prune_rows([flat[0]], torch.zeros(P, dtype=torch.bool, device=dev))
    assert none[0].shape == (0, 3)
    allk = prune_rows([flat[0]], torch.ones(P, dtype=torch.bool, device=dev))
    assert torch.equal(allk[0], flat[0])
    assert prune_rows([torch.empty(0, 3, device=dev)], torch.empty(0, dtype=torch.bool, device=dev))[0].shape == (0, 3)


def test_densification_stats(built):
    from svgir_harness.optim import add_densification_stats
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    P = 70001
    vg = torch.randn(P, 3, generator=g).to(dev)
    flt = (torch.rand(P, generator=g) > 0.5).to(dev)
    w = torch.rand(P, 1, generator=g).to(dev)
    acc = [torch.rand(P, 1, generator=g).to(dev) for _ in range(3)]
    ref = [a.clone() for a in acc]
    ref[0] += w
    ref[1][flt] += torch.norm(vg[flt, :2], dim=-1, keepdim=True)
    ref[2][flt] += 1
    add_densification_stats(vg, flt, w, acc[0], acc[1], acc[2])
    for a, b in zip(acc, ref):
        assert torch.allclose(a, b, rtol=2e-7, atol=0)


# ---- GaussianModel.step() and densify_and_prune against the reference's own methods (tests/golden/densify.npz) ----
_SPEC = [("xyz", 1.6e-4), ("normal", 1e-3), ("rotation", 1e-3), ("scaling", 5e-3), ("opacity", 5e-2), ("f_dc", 2.5e-3),
         ("f_rest", 1.25e-4), ("base_color", 1e-2), ("roughness", 1e-2), ("incidents_dc", 2e-3), ("incidents_rest", 1e-4),
         ("visibility_dc", 2.5e-3), ("visibility_rest", 1.25e-4)]


def _densify_state(gold):
    from svgir_harness.optim import DensifyState, FusedAdam
    dev = torch.device("cuda:0")
    params = {n: torch.nn.Parameter(torch.from_numpy(gold["init_" + n]).to(dev)) for n, _ in _SPEC}
    opt = FusedAdam([{"params": [params[n]], "lr": lr, "name": n} for n, lr in _SPEC], lr=1e-4, eps=1e-15)
    return DensifyState(params, opt, percent_dense=0.01, use_pbr=True), dev


def _same(a, b, what, tol=2e-6):
    a = a.detach().double().cpu().numpy(); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.size == 0:
        return
    finite = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), finite) and np.array_equal(np.isnan(a), np.isnan(b)), what
    scale = max(np.abs(b[finite]).max(), 1e-30) if finite.any() else 1.0
    err = np.abs(a[finite] - b[finite]).max() if finite.any() else 0.0
    assert err <= tol * scale, (what, err / scale)


def test_a_parameter_without_a_fresh_gradient_is_skipped_like_torch_adam(built):
    """GaussianModel.step() ends in optimizer.zero_grad() (grads -> None): a group that drops out of the graph in a later
    iteration keeps its parameter, moments and step count (torch.optim.Adam skips `p.grad is None`), instead of coasting on
    its momentum with g = 0."""
    from svgir_harness.optim import FusedAdam
    dev = torch.device("cuda:0")
    a = torch.nn.Parameter(torch.linspace(-1, 1, 300, device=dev).reshape(100, 3).contiguous())
    b = torch.nn.Parameter(torch.linspace(0, 2, 100, device=dev).reshape(100, 1).contiguous())
    ra, rb = (torch.nn.Parameter(x.detach().clone()) for x in (a, b))
    opt = FusedAdam([{"params": [a], "lr": 1e-2, "name": "a"}, {"params": [b], "lr": 1e-2, "name": "b"}], lr=0.0, eps=1e-15)
    ref = torch.optim.Adam([{"params": [ra], "lr": 1e-2, "name": "a"}, {"params": [rb], "lr": 1e-2, "name": "b"}], lr=0.0, eps=1e-15)
    for it in range(4):
        ga, gb = torch.full_like(a, 0.5 + it), torch.full_like(b, -1.0 - it)
        a.grad, ra.grad = ga.clone(), ga.clone()
        if it < 2:          # group b drops out after two iterations
            b.grad, rb.grad = gb.clone(), gb.clone()
        opt.step(zero_grad=True)
        ref.step(); ref.zero_grad()
        assert a.grad is None and b.grad is None
    assert float(opt.state[b]["step"]) == float(ref.state[rb]["step"]) == 2.0
    for x, y in ((a, ra), (b, rb)):
        assert torch.allclose(x, y, rtol=0, atol=2e-6)
        assert torch.allclose(opt.state[x]["exp_avg"], ref.state[y]["exp_avg"], rtol=0, atol=1e-6)
    # the round-3 behaviour stays available: zero-filled gradients, written by the Adam kernel itself
    a.grad = torch.ones_like(a)
    opt.step(zero_grad="fill")
    assert a.grad is not None and float(a.grad.abs().max()) == 0.0


@pytest.mark.parametrize("fixture", ["densify.npz", "densify_nan.npz"])
def test_step_and_densify_and_prune_match_the_reference_methods(built, fixture):
    """densify_nan.npz: the same run with NaN / overflowing `_scaling` rows, all selected for densification -- the case the
    reference's get_scaling = nan_to_num(exp(.), nan=1e-6) exists for (NaN axes count as 1e-6: cloned, never 'big')."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture))
    st, dev = _densify_state(gold)
    # ---- GaussianModel.step(): NaN-gradient scrub + Adam + zero_grad, one launch ----
    for it in range(3):
        for n, _ in _SPEC:
            g = gold[f"grad{it}_{n}"]
            st.params[n].grad = torch.from_numpy(g).to(dev) if g.size else None
        st.step()
        for n, _ in _SPEC:   # zero_grad: None afterwards, like the reference's optimizer.zero_grad() on torch >= 2
            assert st.params[n].grad is None
    for n, _ in _SPEC:
        p = st.params[n]
        s = st.optimizer.state[p]
        _same(p, gold["step_" + n], "step " + n)
        _same(s["exp_avg"], gold["step_m_" + n], "step m " + n)
        _same(s["exp_avg_sq"], gold["step_v_" + n], "step v " + n, tol=4e-6)
    # ---- densify_and_prune ----
    for k in ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D"):
        setattr(st, k, torch.from_numpy(gold["stat_" + k]).to(dev))
    max_grad, min_opacity, extent, max_screen, max_grad_normal = [float(x) for x in gold["densify_args"]]
    st.densify_and_prune(max_grad, min_opacity, extent, max_screen, max_grad_normal, z=torch.from_numpy(gold["split_z"]).to(dev))
    for n, _ in _SPEC:
        p = st.params[n]
        s = st.optimizer.state[p]
        assert p.requires_grad and isinstance(p, torch.nn.Parameter)
        _same(p, gold["dens_" + n], "densify " + n, tol=3e-6)
        _same(s["exp_avg"], gold["dens_m_" + n], "densify m " + n)
        _same(s["exp_avg_sq"], gold["dens_v_" + n], "densify v " + n, tol=4e-6)
    for k in ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D"):
        _same(getattr(st, k), gold["dens_" + k], "densify " + k)
    # the optimizer keeps stepping on the new block
    for n, _ in _SPEC:
        st.params[n].grad = torch.ones_like(st.params[n]) * 1e-3
    st.step()
    P_new = gold["dens_xyz"].shape[0]
    assert all(st.params[n].shape[0] == P_new and st.optimizer.state[st.params[n]]["exp_avg"].shape[0] == P_new for n, _ in _SPEC)
    assert torch.isfinite(st.params["xyz"]).all()   # (groups outside replace_nangrad_to_zero keep their NaNs, as in the reference)


def test_append_rows_and_masks_edge_cases(built):
    from svgir_harness import optim as O
    dev = torch.device("cuda:0")
    a = torch.arange(12, dtype=torch.float32, device=dev).reshape(4, 3)
    b = torch.arange(4, dtype=torch.int32, device=dev)
    lst, cnt, n = O._scan(torch.tensor([True, False, True, False], device=dev))
    o = O.append_rows([a, b, a], lst, cnt, n, repeat=2, zero_new={2})
    assert torch.equal(o[0], torch.cat([a, a[[0, 2]], a[[0, 2]]]))
    assert torch.equal(o[1], torch.cat([b, b[[0, 2]], b[[0, 2]]]))
    assert torch.equal(o[2], torch.cat([a, torch.zeros(4, 3, device=dev)]))
    lst, cnt, n = O._scan(torch.zeros(4, dtype=torch.bool, device=dev))       # nothing selected
    o = O.append_rows([a], lst, cnt, n)
    assert n == 0 and torch.equal(o[0], a)
