"""GPU probe: bytes of the state blobs and of the backward scratch per instance, first view of a workload (worst-case state slots) and
steady state (slots from the pair statistics of the previous views; gradient rows from the view's pair count)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from gaussian_renderer import _native as N
from svgir_harness import runner, scenes

dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg2", "cfg3_train", "cfg3_eval", "cfg5"]:
    variant = scenes.CONFIGS[name][1]["variant"]
    sc = runner.to_torch(scenes.make(name), dev)
    S = sc["features"].shape[1]; VS = sc["vfeatures"].shape[1] if variant == "svgss" else 0
    W, H, P = sc["W"], sc["H"], sc["means3D"].shape[0]
    rows = []
    for it in range(3):
        raw = runner.forward_raw(sc, variant)
        torch.cuda.synchronize()
        gb, bb, ib = raw["blobs"]
        R = raw["num_rendered"]
        var_id = N.SVGSS if variant == "svgss" else N.RGSS
        scr = int(N.lib.svgir_backward_scratch_bytes_for(var_id, P, bb.numel(), ib.data_ptr(), W, H, S, VS))
        worst_scr = int(N.lib.svgir_backward_scratch_bytes(var_id, P, int(N.lib.svgir_binning_bytes(R, W, H, S, VS)), W, H, S, VS))
        rows.append((bb.numel(), scr))
        del raw, gb, bb, ib
    worst_bin = int(N.lib.svgir_binning_bytes(R, W, H, S, VS))
    print(f"{name}: R = {R}; binning blob: worst case {worst_bin / R:7.1f} B/instance, first view {rows[0][0] / R:7.1f}, steady state {rows[-1][0] / R:7.1f}; "
          f"backward scratch: worst case {worst_scr / R:7.1f} B/instance, this view {rows[-1][1] / R:7.1f}; "
          f"together {(worst_bin + worst_scr) / R:7.1f} -> {(rows[-1][0] + rows[-1][1]) / R:7.1f} B/instance "
          f"({(worst_bin + worst_scr) / 2**20:.0f} -> {(rows[-1][0] + rows[-1][1]) / 2**20:.0f} MiB)")
