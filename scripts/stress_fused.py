"""GPU stress of the fused shading (svgir_params.shade): random scene sizes / image sizes / sample counts / lattice or streamed directions /
training or evaluation widths in ONE process -- the speculation history is shared across them -- each compared bit for bit with the all-P
path (tests/test_gpu_fused_shade.py::_compare: images, every gradient; dL/d env to the order of its float atomics).
    python scripts/stress_fused.py [seed] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_gpu_fused_shade as F
from svgir_harness import runner, scenes

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
dev = torch.device("cuda:0")
n_ok = n_bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    training = bool(rng.random() < 0.5)
    S, VS = (4, 52) if training else (7, 64)
    P = int(rng.choice([1, 40, 700, 3000, 9000, 30000]))
    W, H = int(rng.integers(17, 300)), int(rng.integers(17, 220))
    Ns = int(rng.choice([1, 7, 24, 64, 96, 128, 130, 200]))
    lo = float(rng.choice([0.01, 0.03, 0.08]))
    lattice = bool(rng.random() < 0.6)
    sc = scenes.surface_scene(P=P, W=W, H=H, seed=int(rng.integers(1 << 30)), sh_degree=int(rng.integers(0, 4)), variant="svgss", S=S, VS=VS,
                              scale_lo=lo, scale_hi=lo * float(rng.choice([2.0, 5.0])))
    if rng.random() < 0.3:
        sc["opacities"] = (sc["opacities"] * 0.05).astype(np.float32)   # translucent: deep stacks
    sct = runner.to_torch(sc, dev)
    st = runner.settings(sct, "svgss")
    d = F._materials(sct, st, Ns, training, seed=int(rng.integers(1 << 30)), lattice=lattice)
    grads = scenes.upstream_grads(sc, "svgss", seed=int(rng.integers(1 << 30)))
    tag = f"{it:2d} P={P} {W}x{H} Ns={Ns} training={training} lattice={lattice} scale_lo={lo}"
    try:
        oa, _ = F._compare(sct, st, d, grads, training, grad_mat=bool(rng.random() < 0.8))
    except AssertionError as e:
        n_bad += 1
        print("FAIL", tag, "::", str(e)[:300], flush=True)
        continue
    n_ok += 1
    print("ok", tag, "R =", oa[0], flush=True)
print("fused stress passed:", n_ok, "failed:", n_bad)
sys.exit(1 if n_bad else 0)
