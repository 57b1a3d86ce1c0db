"""Two views rendered concurrently on two HIP streams from two host threads (the way a per-GPU driver fills the issue slots one
view leaves idle: HISTORY.md 4) must give what each gives alone.  Exercises the library's shared state under concurrency: the
speculative-capacity cache, the per-device side stream of the backward, the allocator callbacks."""
import threading

import numpy as np
import pytest
import torch

from svgir_harness import cameras, runner, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _views(variant):
    kw = dict(P=5000, W=176, H=144, seed=31, sh_degree=2, variant=variant, scale_lo=0.01, scale_hi=0.06)
    kw.update(dict(S=4, VS=52) if variant == "svgss" else dict(S=5, VS=0))
    a = scenes.surface_scene(**kw)
    b = dict(a)
    b.update(cameras.make_camera(a["W"], a["H"], cameras.orbit_eye(4.0, 70.0, 20.0)))
    return a, b


def _run(sc, variant, grads, stream, rounds, out):
    dev = torch.device(DEV)
    with torch.cuda.stream(stream):
        sct = runner.to_torch(sc, dev)
        for _ in range(rounds):
            res, leaves = runner.render(sct, variant, requires_grad=True)
            runner.backward(res, grads, variant)
        stream.synchronize()
    out["res"] = {k: v.detach().cpu().numpy() for k, v in res.items() if torch.is_tensor(v)}
    out["num_rendered"] = res["num_rendered"]
    out["grads"] = {k: v.grad.detach().cpu().numpy() for k, v in leaves.items() if v.grad is not None}


@pytest.mark.parametrize("variant", ["svgss", "rgss"])
def test_two_views_on_two_streams(built, variant):
    views = _views(variant)
    grads = [scenes.upstream_grads(v, variant, seed=5 + i) for i, v in enumerate(views)]
    alone = [{}, {}]
    for i in range(2):
        _run(views[i], variant, grads[i], torch.cuda.Stream(DEV), 2, alone[i])
    both = [{}, {}]
    th = [threading.Thread(target=_run, args=(views[i], variant, grads[i], torch.cuda.Stream(DEV), 12, both[i])) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert alone[0]["num_rendered"] != alone[1]["num_rendered"]          # really two different views
    for i in range(2):
        assert both[i]["num_rendered"] == alone[i]["num_rendered"]
        for k, v in alone[i]["res"].items():
            if k == "weights":   # float atomics across sub-tiles: order-dependent in the last bits
                assert np.allclose(both[i]["res"][k], v, rtol=1e-5, atol=1e-6), k
            else:
                assert np.array_equal(both[i]["res"][k], v), k
        for k, v in alone[i]["grads"].items():
            g = both[i]["grads"][k]
            if variant == "svgss":   # gradient rows + ordered reduction: bit-reproducible
                assert np.array_equal(g, v), k
            else:                    # rgss sums its rows with float atomics
                assert np.allclose(g, v, rtol=2e-4, atol=1e-6 * max(1.0, np.abs(v).max())), k


def test_render_in_flight_helper(built):
    from svgir_harness import view_parallel as vp
    dev = torch.device(DEV)
    a, b = _views("svgss")
    scs = [a, b, a, b, a]

    def render(sc):
        res, _ = runner.render(runner.to_torch(sc, dev), "svgss", requires_grad=False)
        return res["color"].detach().cpu().numpy()

    seq = [render(sc) for sc in scs]
    par = vp.render_in_flight(render, scs, dev, in_flight=2)
    assert len(par) == 5 and all(np.array_equal(x, y) for x, y in zip(par, seq))
    with pytest.raises(ZeroDivisionError):
        vp.render_in_flight(lambda sc: 1 // 0, scs, dev)


def test_shading_on_a_side_stream_overlapping_the_binning(built):
    """svgir_params.features_ready: the features / vfeatures of a view are produced by the shading kernels on a side stream
    while the rasterizer's preprocess / sorts / cull of the same view run; only the composite kernel waits.  Same results as
    the sequential order, bit for bit."""
    from gaussian_renderer import shading
    from gaussian_renderer.svgss_rasterization import _C
    from svgir_harness import shade_inputs
    dev = torch.device(DEV)
    sc = scenes.surface_scene(P=6000, W=176, H=144, seed=41, sh_degree=2, variant="svgss", S=4, VS=52, scale_lo=0.01, scale_hi=0.06)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = shade_inputs.make(6000, 64, seed=3, device=dev)
    light = shade_inputs.Light(d["env"])
    empty = torch.empty(0, dtype=torch.float32, device=dev)

    def shade():
        f, vf, _ = shading.shade_and_pack(d["base_color"], d["roughness"], d["normals"], d["viewdirs"], d["radiance"], light,
                                          d["visibility"], d["dirs"], d["areas"], st.viewmatrix, True)
        return f, vf

    def raster(f, vf, ready=None):
        out = _C.rasterize_gaussians(st.bg, sct["means3D"], f, vf, empty, sct["opacities"], sct["scales"], sct["rotations"],
                                     st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx,
                                     st.tanfovy, st.image_height, st.image_width, sct["shs"], st.sh_degree, st.campos, False, False,
                                     st.config, features_ready=ready)
        return [o.clone() for o in out[1:7]], out[7].clone()   # images; out_weights (float atomics: order-dependent bits)

    f, vf = shade()
    ref, ref_w = raster(f, vf)
    torch.cuda.synchronize()
    main, side, ev = torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Event()
    for _ in range(5):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            f2, vf2 = shade()
        ev.record(side)
        f2.record_stream(main); vf2.record_stream(main)
        got, got_w = raster(f2, vf2, ready=ev)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, ref))
        assert torch.allclose(got_w, ref_w, rtol=1e-5, atol=1e-6)


def _call_args(sct, st, variant, empty):
    if variant == "svgss":
        return (st.bg, sct["means3D"], sct["features"], sct["vfeatures"], empty, sct["opacities"], sct["scales"], sct["rotations"],
                st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy,
                st.image_height, st.image_width, sct["shs"], st.sh_degree, st.campos, False, False, st.config)
    return (st.bg, sct["means3D"], sct["features"], empty, sct["opacities"], sct["scales"], sct["rotations"], st.scale_modifier, empty,
            st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.cx, st.cy, st.image_height, st.image_width, sct["shs"],
            st.sh_degree, st.campos, False, False, False)


@pytest.mark.parametrize("variant", ["svgss", "rgss"])
def test_forward_batch_from_one_thread_matches_single_calls(built, variant):
    """svgir_forward_batch: five views of one scene launched from ONE host thread on four streams -- every view's kernels are
    in the queue before the first view's instance count is awaited -- give bit for bit what five svgir_forward calls give, and the
    blobs they leave behind drive the backward like any other."""
    from svgir_harness import view_parallel as vp
    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import _C
    else:
        from gaussian_renderer.rgss_rasterization import _C
    dev = torch.device(DEV)
    base, _ = _views(variant)
    scs = []
    for i in range(5):
        v = dict(base)
        v.update(cameras.make_camera(base["W"], base["H"], cameras.orbit_eye(4.0, 15.0 + 60.0 * i, 10.0 + 5.0 * i)))
        scs.append(runner.to_torch(v, dev))
    sts = [runner.settings(s, variant) for s in scs]
    empty = torch.empty(0, dtype=torch.float32, device=dev)
    nimg = 7 if variant == "svgss" else 9      # leading tensors of the tuple that are images (+ n_contrib for rgss)
    for rounds in range(3):   # (the third round runs the speculative launch sequence for every view)
        single = [_C.rasterize_gaussians(*_call_args(scs[i], sts[i], variant, empty)) for i in range(5)]
        torch.cuda.synchronize()
        got = vp.render_views_in_flight(lambda i: (_call_args(scs[i], sts[i], variant, empty), {}), lambda i, res: res, range(5), dev,
                                        _C.rasterize_gaussians_batch, in_flight=4)
        torch.cuda.synchronize()
        assert len({s[0] for s in single}) > 1            # really different views
        for a, b in zip(single, got):
            assert a[0] == b[0]
            for j in range(1, nimg):
                assert torch.equal(a[j], b[j]), j
            assert torch.equal(a[-4], b[-4])              # radii
    # a backward on the blobs of a batched forward
    i = 3
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in scenes.upstream_grads(base, variant, seed=2).items()}
    sct, st = scs[i], sts[i]

    def bwd(out):
        if variant == "svgss":
            return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], sct["vfeatures"], out[8], empty, sct["scales"],
                                                   sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint,
                                                   st.patch_bbox, st.tanfovx, st.tanfovy, g["color"], g["normal"], g["depth"], g["opacity"],
                                                   g["feature"], g["vfeature"], sct["shs"], st.sh_degree, st.campos, out[9], out[0], out[10],
                                                   out[11], False, st.config)
        return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], out[10], empty, sct["scales"], sct["rotations"],
                                               st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, g["color"],
                                               g["normal"], g["opacity"], g["depth"], g["feature"], sct["shs"], st.sh_degree, st.campos,
                                               out[11], out[0], out[12], out[13], True, False)

    ga, ga2, gb = bwd(single[i]), bwd(single[i]), bwd(got[i])
    torch.cuda.synchronize()
    for a, a2, b in zip(ga, ga2, gb):
        if variant == "svgss":
            assert torch.equal(a, b)
        else:   # rgss sums its packed rows with float atomics: the same call twice differs by `noise`
            noise = float((a - a2).abs().max())
            assert float((a - b).abs().max()) <= 4.0 * noise + 1e-6 * float(a.abs().max()) + 1e-12


@pytest.mark.parametrize("variant", ["svgss", "rgss"])
def test_forward_only_keeps_no_states_and_a_backward_still_works(built, variant):
    """svgir_params.forward_only (evaluation loops): same images from a smaller binning blob (no state slots); a backward that
    comes anyway replays the composite for its states and gives the usual gradients."""
    if variant == "svgss":
        from gaussian_renderer.svgss_rasterization import _C
    else:
        from gaussian_renderer.rgss_rasterization import _C
    dev = torch.device(DEV)
    kw = dict(P=20000, W=256, H=192, seed=33, sh_degree=1, variant=variant, scale_lo=0.03, scale_hi=0.12)   # long lists: states are dumped
    kw.update(dict(S=4, VS=52) if variant == "svgss" else dict(S=5, VS=0))
    sc = scenes.surface_scene(**kw)
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, variant)
    empty = torch.empty(0, dtype=torch.float32, device=dev)
    args = _call_args(sct, st, variant, empty)
    for _ in range(3):
        a = _C.rasterize_gaussians(*args)
        b = _C.rasterize_gaussians(*args, forward_only=True)
    torch.cuda.synchronize()
    nimg = 7 if variant == "svgss" else 9
    assert a[0] == b[0]
    for j in range(1, nimg):
        assert torch.equal(a[j], b[j]), j
    assert b[-2].numel() < a[-2].numel(), "no state slots in the binding blob of a forward-only view"
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in scenes.upstream_grads(sc, variant, seed=4).items()}

    def bwd(out):
        if variant == "svgss":
            return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], sct["vfeatures"], out[8], empty, sct["scales"],
                                                   sct["rotations"], st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.prcppoint,
                                                   st.patch_bbox, st.tanfovx, st.tanfovy, g["color"], g["normal"], g["depth"], g["opacity"],
                                                   g["feature"], g["vfeature"], sct["shs"], st.sh_degree, st.campos, out[9], out[0], out[10],
                                                   out[11], False, st.config)
        return _C.rasterize_gaussians_backward(st.bg, sct["means3D"], sct["features"], out[10], empty, sct["scales"], sct["rotations"],
                                               st.scale_modifier, empty, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, g["color"],
                                               g["normal"], g["opacity"], g["depth"], g["feature"], sct["shs"], st.sh_degree, st.campos,
                                               out[11], out[0], out[12], out[13], True, False)

    ga, ga2 = bwd(a), bwd(a)
    before = N_stats()
    gb = bwd(b)
    torch.cuda.synchronize()
    assert N_stats()["rerun_slots"] == before["rerun_slots"] + 1, "the forward-only view's states were dumped by its backward"
    for x, x2, y in zip(ga, ga2, gb):
        if variant == "svgss":
            assert torch.equal(x, y)
        else:   # rgss sums its packed rows with float atomics: the same call twice differs by `noise`
            noise = float((x - x2).abs().max())
            assert float((x - y).abs().max()) <= 4.0 * noise + 1e-6 * float(x.abs().max()) + 1e-12


def N_stats():
    from gaussian_renderer import _native
    return _native.speculation_stats()
