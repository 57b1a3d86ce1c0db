"""GPU stress: random scene sizes / widths / image sizes in one process (exercises the speculative capacity logic, the
depth segments, the gradient rows and the channel-group fallback back to back) against the CPU oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import test_gpu_parity as T
from svgir_harness import scenes

# tiny images / tiny scenes: one threshold flip (a pixel = 3 entries) must not exceed the allowed *fraction*
_cmp0 = T._cmp
def _cmp(name, a, b, tol=T.TOL, flip_frac=T.FLIP_FRAC, flip_bound=T.FLIP_BOUND, rel=True, **kw):
    n = max(1, int(np.asarray(b).size))
    if "rel_frac" in kw:   # (the same for the relative criterion: with 50 entries ONE nearly cancelling sum is 2 % of them)
        kw["rel_frac"] = max(kw["rel_frac"], 5.0 / n)
    return _cmp0(name, a, b, tol=tol, flip_frac=max(flip_frac, 8.0 / n), flip_bound=flip_bound, rel=rel, **kw)
T._cmp = _cmp

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
widths = {"svgss": [(0, 0), (1, 4), (3, 8), (4, 52), (7, 64), (5, 0), (2, 4), (6, 24)], "rgss": [(0, 0), (1, 0), (3, 0), (5, 0), (4, 0), (8, 0)]}
n_ok = n_bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    variant = "svgss" if rng.random() < 0.6 else "rgss"
    S, VS = widths[variant][rng.integers(len(widths[variant]))]
    P = int(rng.choice([1, 50, 800, 3000, 9000, 25000]))
    W, H = int(rng.integers(17, 260)), int(rng.integers(17, 200))
    lo = float(rng.choice([0.01, 0.03, 0.08]))
    sc = scenes.surface_scene(P=P, W=W, H=H, seed=int(rng.integers(1 << 30)), sh_degree=int(rng.integers(0, 4)), variant=variant,
                              S=S, VS=VS, scale_lo=lo, scale_hi=lo * float(rng.choice([2.0, 5.0])))
    if rng.random() < 0.3:
        sc["opacities"] = (sc["opacities"] * 0.05).astype(np.float32)   # translucent: deep stacks, many segments
    grads = scenes.upstream_grads(sc, variant, seed=int(rng.integers(1 << 30)))
    # surfels of scale 0.08-0.4 make cov2D^-1 ill-conditioned in fp32: the oracle's own fp32-vs-fp64 gap on grad_scales /
    # grad_cov3D then exceeds the per-element relative criterion (1.2 % of the entries at seed 7, scene 18), so that criterion is
    # relaxed for those scenes; the normalised criterion (1e-4 of the tensor's scale) stays
    translucent = float(sc['opacities'].max()) < 0.06   # hundreds of blended terms per pixel: longer, more cancelling sums
    T.REL_FRAC = 2e-2 if lo >= 0.08 else (5e-3 if translucent else 1e-3)
    T.REL_TOL = 5e-3   # (tiny tensors: a dozen nearly cancelling sums are already 0.5 % of the entries; the full-size tests keep 2e-3)
    out, leaves, o, R = T._run_both(sc, variant, grads)
    tag = f"{it:2d} {variant} P={P} {W}x{H} S={S} VS={VS} R={R} scale_lo={lo} translucent={float(sc['opacities'].max()) < 0.06}"
    try:
        T._check_forward(out, o, R, variant)
        T._check_backward(leaves, o, variant)
    except AssertionError as e:
        n_bad += 1
        print("FAIL", tag, "::", str(e)[:200], flush=True)
        continue
    n_ok += 1
    print("ok", tag, flush=True)
print("stress passed:", n_ok, "failed:", n_bad)
sys.exit(1 if n_bad else 0)
