"""Blob sizing without cliffs (include/svgir_raster.h: the binning blob is sized from the workload's recent views).

The model's surfel count changes every few hundred iterations (densify_and_prune, scene/gaussian_model.py:1229-1253).  The
speculation history must survive that -- counts are scaled by the ratio of the surfel counts -- and a workload's first view must not
allocate the worst case over the cull (four full candidate lists per tile)."""
import numpy as np
import pytest
import torch

from gaussian_renderer import _native as N
from svgir_harness import runner, scenes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _fwd_bwd(sct, grads):
    res, leaves = runner.render(sct, "svgss", requires_grad=True)
    runner.backward(res, grads, "svgss")
    return res


def test_growing_model_never_reruns_after_its_first_view(built):
    dev = torch.device(DEV)
    W, H = 208, 176     # (a resolution no other test uses: this workload starts without history)
    P = 6000
    base = scenes.surface_scene(P=9000, W=W, H=H, seed=91, sh_degree=1, variant="svgss", S=4, VS=52, scale_lo=0.02, scale_hi=0.09)
    grads = scenes.upstream_grads(base, "svgss", seed=3)

    def model(n):   # the first n surfels of the pool (densification appends)
        sc = dict(base)
        for k in ("means3D", "scales", "rotations", "opacities", "shs", "features", "vfeatures"):
            sc[k] = base[k][:n]
        return runner.to_torch(sc, dev)

    before = N.speculation_stats()
    Rs = []
    for it in range(12):
        if it and it % 3 == 0:
            P = int(P * 1.05) + 1      # +5 % every three views
        res = _fwd_bwd(model(P), grads)
        Rs.append(res["num_rendered"])
        if it == 0:
            first = N.speculation_stats()
    torch.cuda.synchronize()
    after = N.speculation_stats()
    assert Rs[-1] > Rs[0] * 1.1                                                  # the view really grew
    assert after["rerun_capacity"] == first["rerun_capacity"] == before["rerun_capacity"], (before, first, after)
    assert after["rerun_slots"] == before["rerun_slots"], (before, after)       # ... and no view's states had to be dumped again
    # one reference render of the last model from a clean history gives the same image (the speculation changes nothing)
    ref, _ = runner.render(model(P), "svgss", requires_grad=False)
    assert torch.equal(ref["color"], res["color"].detach())


@pytest.mark.parametrize("variant", ["svgss", "rgss"])
def test_first_view_blob_is_close_to_steady_state(built, variant):
    dev = torch.device(DEV)
    kw = dict(P=30000, W=304, H=240, seed=17, sh_degree=1, variant=variant, scale_lo=0.02, scale_hi=0.08)   # (fresh resolution again)
    kw.update(dict(S=4, VS=52) if variant == "svgss" else dict(S=5, VS=0))
    sc = runner.to_torch(scenes.surface_scene(**kw), dev)
    sizes = []
    for _ in range(4):
        raw = runner.forward_raw(sc, variant)
        sizes.append(raw["blobs"][1].numel())
        R = raw["num_rendered"]
    S, VS = kw["S"], kw["VS"]
    worst = int(N.lib.svgir_binning_bytes(R, kw["W"], kw["H"], S, VS))
    assert sizes[0] <= 1.3 * sizes[-1], (sizes, worst)          # first view: sized from its own cull, not from the worst case
    assert sizes[0] < 0.8 * worst, (sizes, worst)


def test_two_models_with_equal_widths_keep_separate_histories(built):
    """svgir_params.workload_scope (ABI 14): two models that share (device, image size, widths, variant) and alternate in one process feed
    ONE speculation history unless each carries its own scope -- the smaller model's binning blob is then sized for the larger one.  With
    scopes each is sized from its own views; svgir_reset_workload_history makes the next view a first view again; images never change."""
    dev = torch.device(DEV)
    kw = dict(W=272, H=208, seed=23, sh_degree=1, variant="svgss", S=4, VS=52)   # (a resolution no other test uses)
    small = runner.to_torch(scenes.surface_scene(P=9000, scale_lo=0.01, scale_hi=0.03, **kw), dev)
    big = runner.to_torch(scenes.surface_scene(P=12000, scale_lo=0.05, scale_hi=0.12, **kw), dev)

    def rounds(scoped, n=4):
        sizes = {"small": [], "big": []}
        for _ in range(n):
            for name, sc, scope in (("small", small, 11), ("big", big, 12)):
                with N.workload_scope(scope if scoped else 0):
                    raw = runner.forward_raw(sc, "svgss")
                sizes[name].append((raw["blobs"][1].numel(), raw["num_rendered"]))
        return sizes

    N.reset_workload_history(-1)
    shared = rounds(False)
    N.reset_workload_history(-1)
    own = rounds(True)
    assert shared["big"][-1][1] > 3 * shared["small"][-1][1], "the two models must differ in their instance counts"
    # one history: the small model's steady-state blob is laid out for the big model's instance count
    assert shared["small"][-1][0] > 0.6 * shared["big"][-1][0], shared
    # own scopes: sized from its own views
    assert own["small"][-1][0] < 0.5 * own["big"][-1][0], own
    assert own["big"][-1][0] <= shared["big"][-1][0]
    # a reset scope starts over: its next view waits for the exact instance count (blob = exact capacity, no 12.5 % headroom) ...
    before = N.speculation_stats()
    N.reset_workload_history(11)
    with N.workload_scope(11):
        again = runner.forward_raw(small, "svgss")
    with N.workload_scope(12):
        other = runner.forward_raw(big, "svgss")
    after = N.speculation_stats()
    assert again["blobs"][1].numel() <= own["small"][-1][0]
    assert other["blobs"][1].numel() == own["big"][-1][0]                       # ... and the other scope is untouched
    assert after["rerun_capacity"] == before["rerun_capacity"]
    ref, _ = runner.render(small, "svgss", requires_grad=False)
    assert torch.equal(ref["color"], again["color"])
