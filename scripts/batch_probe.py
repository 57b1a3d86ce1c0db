"""svgir_forward_batch: surfels/s of V views in flight from one host thread (bench.py one_thread_batch) for several V, next to the host
time of one forward + backward binding call (the Python side of a view)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd"))
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
for V in (1, 2, 3, 4, 6, 8):
    r = bench.one_thread_batch(args.workload or "cfg2", dev, args, V=V)
    print(V, "views in flight: %.1f M surfels/s, %.4f ms per view" % (r["value"] / 1e6, r["ms_per_view"]))
    torch.cuda.empty_cache()
w = bench.Workload(args.workload or "cfg2", dev, 0, 1, args)
for _ in range(5):
    w.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    w.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time of one view's forward + backward calls: %.1f us (GPU drained %.1f us later)" % ((t1 - t0) / 200 * 1e6, (t2 - t1) * 1e6))
