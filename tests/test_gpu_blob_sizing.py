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
