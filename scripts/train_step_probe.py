"""GPU probe: phase table of one stage-2 training iteration (svgir_harness.workloads.TrainStep, cfg3_train)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import torch
from svgir_harness import workloads

dev = torch.device("cuda:0")
ts = workloads.TrainStep(dev)
for _ in range(5):
    R, img, loss = ts.step()
torch.cuda.synchronize()
print("R", R, "loss", float(loss), "param elements", ts.n_param_elems)
t0 = time.perf_counter()
K = 30
for _ in range(K):
    ts.step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
print(f"train step: {ms:.3f} ms  ({ts.P / ms / 1e3:.1f} M surfels/s)")
tab = ts.stage_table(10)
print("phases (ms, HIP events, with a sync per step):", {k: round(v, 3) for k, v in tab.items()}, "sum", round(sum(tab.values()), 3))
