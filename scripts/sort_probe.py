"""Side-by-side of the library's tile sort and rocPRIM's radix_sort_pairs on the very same arrays (SURVEY 7 names the comparison).

For each workload: one forward of the view, the EMIT-order (tile key, Gaussian id) arrays rebuilt from the sorted instance list
(instances of a Gaussian in ascending tile order, Gaussians in depth order -- what duplicateWithKeys / emit_kernel produce), written
to a file, then scripts/probes/sort_probe.hip (compiled here with hipcc against the product's binning.hip and the rocPRIM headers of
the ROCm install) sorts them both ways.  Bench only: nothing of rocPRIM is linked into libsvgir_raster.so.
    python scripts/sort_probe.py [cfg2 cfg5 cfg5_dense]   (on the GPU box)"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd"))
sys.path.insert(0, ROOT)
from svgir_harness import runner, scenes  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out", "sort_probe")
os.makedirs(out_dir, exist_ok=True)
exe = os.path.join(out_dir, "sort_probe")
csrc = os.path.join(ROOT, "svg-ir_amd", "csrc")
flags = subprocess.check_output(["make", "-s", "-C", csrc, "print-hipflags"], text=True).split()
flags = [f for f in flags if f != "-fPIC"]
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-I/opt/rocm/include", os.path.join(ROOT, "scripts", "probes", "sort_probe.hip"),
                       os.path.join(csrc, "binning.hip"), "-o", exe])
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg2", "cfg5", "cfg5_dense"]:
    variant = scenes.CONFIGS[name][1]["variant"]
    sc = scenes.make(name)
    raw = runner.forward_raw(runner.to_torch(sc, dev), variant)
    torch.cuda.synchronize()
    pl, ranges = raw["point_list"].astype(np.int64), raw["ranges"].astype(np.int64)
    T = ranges.shape[0]
    tile = np.repeat(np.arange(T, dtype=np.int64), ranges[:, 1] - ranges[:, 0])      # tile of every sorted instance
    gid = pl[:tile.size]
    V = np.asarray(sc["viewmatrix"], dtype=np.float64).reshape(4, 4)
    depth = (np.c_[np.asarray(sc["means3D"], np.float64), np.ones(len(sc["means3D"]))] @ V)[:, 2].astype(np.float32)
    rank = np.empty(len(depth), dtype=np.int64)
    rank[np.argsort(depth.view(np.uint32), kind="stable")] = np.arange(len(depth))     # the library's depth key order (stable)
    order = np.lexsort((tile, rank[gid]))                                               # emit order: by Gaussian (depth), then tile
    keys, vals = tile[order].astype(np.uint32), gid[order].astype(np.uint32)
    path = os.path.join(out_dir, name + ".bin")
    with open(path, "wb") as f:
        np.array([keys.size], dtype=np.uint32).tofile(f); keys.tofile(f); vals.tofile(f)
    bits = max(1, int(np.ceil(np.log2(T))))
    del raw
    torch.cuda.empty_cache()
    r = subprocess.run([exe, path, str(bits)], capture_output=True, text=True)
    print(f"{name}: R = {keys.size}, T = {T}: " + (r.stdout.strip() or r.stderr.strip()))
    os.remove(path)
