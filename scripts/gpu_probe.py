"""Development probe (GPU box): parity statistics against the oracle + per-stage timings for named configs.
Usage: python scripts/gpu_probe.py cfg2 cfg3_train [--small]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gaussian_renderer import _native
from oracle import oracle as orc
from svgir_harness import runner, scenes


def stats(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.size == 0:
        return {}
    d = np.abs(a - b)
    scale = max(np.abs(b).max(), 1e-30)
    rel = d / (1e-4 * scale + np.abs(b))
    return dict(max_abs=float(d.max()), scale=float(scale), frac_gt_1e4=float((rel > 1e-4).mean()),
                p9999=float(np.quantile(d, 0.9999) / scale), mean_rel=float(d.mean() / scale))


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["cfg1"]
    small = "--small" in sys.argv
    dev = torch.device("cuda:0")
    res = {}
    for name in names:
        kw = {}
        if small:
            kw = dict(P=20000, W=320, H=240)
        sc = scenes.make(name, **kw)
        variant = scenes.CONFIGS[name][1]["variant"]
        var_id = orc.SVGSS if variant == "svgss" else orc.RGSS
        grads = scenes.upstream_grads(sc, variant)
        sct = runner.to_torch(sc, dev)
        _native.set_profiling(True)
        out, leaves = runner.render(sct, variant, requires_grad=True)
        tf = _native.last_timings()
        runner.backward(out, grads, variant)
        tb = _native.last_timings()
        torch.cuda.synchronize()
        # repeat for warm timings
        tfs, tbs = [], []
        for _ in range(3):
            o2, l2 = runner.render(sct, variant, requires_grad=True)
            tfs.append(dict(_native.last_timings()))
            runner.backward(o2, grads, variant)
            tbs.append(dict(_native.last_timings()))
        _native.set_profiling(False)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            o2, l2 = runner.render(sct, variant, requires_grad=True)
            runner.backward(o2, grads, variant)
        torch.cuda.synchronize()
        wall = (time.time() - t0) / 5

        o = orc.OracleRun(sc, var_id)
        R = o.forward()
        o.backward(grads["color"], grads["normal"], grads["depth"], grads["opacity"], grads["feature"], grads.get("vfeature"))
        im, gr = o.images(), o.grads()
        r = dict(P=int(sc["means3D"].shape[0]), R_gpu=int(out["num_rendered"]), R_oracle=int(R),
                 radii_equal=bool(np.array_equal(out["radii"].cpu().numpy(), im["radii"])),
                 fwd_ms_first=tf, bwd_ms_first=tb, fwd_ms=tfs[-1], bwd_ms=tbs[-1], wall_fwd_bwd_ms=wall * 1e3)
        keys = ["color", "normal", "depth", "opacity", "feature"] + (["vfeature"] if variant == "svgss" else [])
        r["fwd"] = {k: stats(out[k].detach().cpu().numpy(), im[k]) for k in keys}
        r["fwd"]["weights"] = stats(out["weights"].detach().cpu().numpy(), im["weights"])
        if variant == "rgss":
            r["n_contrib_equal_frac"] = float((out["n_contrib"].cpu().numpy() == im["n_contrib"]).mean())
        gm = dict(means3D="means3D", scales="scales", rotations="rotations", opacities="opacity", shs="sh",
                  features="features", means2D="means2D")
        if variant == "svgss":
            gm["vfeatures"] = "vfeatures"
        r["bwd"] = {k: stats(leaves[k].grad.detach().cpu().numpy(), gr[v]) for k, v in gm.items() if leaves[k].grad is not None}
        res[name] = r
        print(json.dumps({name: r}, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
