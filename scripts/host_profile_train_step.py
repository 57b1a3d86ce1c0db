"""Where the HOST time of one training iteration goes (svgir_harness.workloads.TrainStep.step): cProfile over 200 steps, top functions by
cumulative and by own time, next to the wall clock per step with the GPU drained every step (host + GPU serial) and free-running."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd"))
sys.path.insert(0, ROOT)
from svgir_harness import workloads  # noqa: E402

dev = torch.device("cuda:0")
ts = workloads.TrainStep(dev)
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("free-running: host returns after %.1f us per step; GPU done %.1f us per step" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    ts.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
