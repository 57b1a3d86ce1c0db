"""GPU timing of csrc/optim.hip against the PyTorch calls it replaces (the reference's parameter block, P = 200 000).
    python scripts/optim_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from svgir_harness.optim import FusedAdam, prune_rows

dev = torch.device("cuda:0")
P = 200000
spec = [("xyz", (3,), 1.6e-4), ("normal", (3,), 1e-3), ("rotation", (4,), 1e-3), ("scaling", (3,), 5e-3), ("opacity", (1,), 5e-2),
        ("f_dc", (1, 3), 2.5e-3), ("f_rest", (15, 3), 1.25e-4), ("base_color", (4, 3), 1e-2), ("roughness", (4, 1), 1e-2),
        ("incidents_dc", (1, 3), 2e-3), ("incidents_rest", (15, 3), 1e-4), ("visibility_dc", (1, 1), 2.5e-3), ("visibility_rest", (15, 1), 1.25e-4)]


def groups():
    return [{"params": [torch.nn.Parameter(torch.randn((P,) + s, device=dev))], "lr": lr, "name": n} for n, s, lr in spec]


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, cls, kw in (("torch.optim.Adam (foreach)", torch.optim.Adam, {}), ("torch.optim.Adam (fused=True)", torch.optim.Adam, {"fused": True}),
                      ("FusedAdam (csrc/optim.hip)", FusedAdam, {})):
    g = groups()
    for grp in g:
        grp["params"][0].grad = torch.randn_like(grp["params"][0])
    try:
        opt = cls(g, lr=1e-4, eps=1e-15, **kw)
        print(f"{name:34s} step: {timeit(opt.step):.3f} ms")
    except Exception as e:   # noqa: BLE001
        print(f"{name:34s} unavailable: {e}")
floats = sum(int(torch.tensor(s).prod()) for _, s, _ in spec)
print(f"   (parameter block: {floats} floats per Gaussian; one Adam step moves {7 * 4 * floats * P / 1e6:.0f} MB)")

tens = []
for grp in groups():
    p = grp["params"][0].detach()
    tens += [p, torch.randn_like(p), torch.rand_like(p)]
tens += [torch.rand(P, 1, device=dev) for _ in range(4)] + [torch.randint(0, 9, (P,), device=dev, dtype=torch.int32)]
mask = torch.rand(P, device=dev) > 0.3
print(f"t[mask] x {len(tens)} tensors (torch):   {timeit(lambda: [t[mask] for t in tens], 20):.3f} ms")
print(f"prune_rows (one scan + one gather): {timeit(lambda: prune_rows(tens, mask), 20):.3f} ms")
