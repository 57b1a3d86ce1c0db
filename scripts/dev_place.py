"""Development only: where the composite workgroups land (block index -> SE / CU / SIMD) and how long they run.
    SVGIR_RASTER_LIB=build/variants/<dev build>/libsvgir_raster.so python scripts/dev_place.py [workload]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from gaussian_renderer import _native
from svgir_harness import runner, scenes

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
variant = scenes.CONFIGS[wl][1]["variant"]
dev = torch.device("cuda:0")
sct = runner.to_torch(scenes.make(wl), dev)
lib = _native.lib
lib.svgir_dev_trace_read.restype = C.c_int
lib.svgir_dev_trace_read.argtypes = [C.c_int, C.c_void_p, C.c_int]
CAP = 1 << 17
buf = np.zeros((CAP, 8), dtype=np.uint64)
for it in range(3):
    out, _ = runner.render(sct, variant)
    torch.cuda.synchronize()
    n = lib.svgir_dev_trace_read(0, buf.ctypes.data, CAP)
rec = buf[:n].astype(np.int64)
dur, r0, r1 = rec[:, 0], rec[:, 1], rec[:, 2]
ln, tail = rec[:, 3] >> 32, rec[:, 3] & 0xffffffff
blk, hw = rec[:, 4] >> 32, rec[:, 4] & 0xffffffff
stage = rec[:, 6]
t0 = r0.min()
o = np.argsort(blk)
se, cu, simd = (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3
print("hw id bits of the first waves (raw):", [hex(int(h)) for h in hw[o][:8]])
print("  blk  cand   se cu simd   start_us  dur_us  stage_kcyc")
for i in o[:48]:
    print("%5d %5d   %d %2d %d   %8.1f %7.1f %8.1f" % (blk[i], tail[i], se[i], cu[i], simd[i], (r0[i] - t0) / 100.0, (r1[i] - r0[i]) / 100.0, stage[i] / 1e3))
# per (se, cu) load: waves, candidates, mean duration
key = se * 16 + cu
print("per (se,cu) [blk%8 is the XCD and is not in this id]: waves / candidates / mean wave us")
for k in np.unique(key)[:40]:
    m = key == k
    print("  se%d cu%2d: %3d waves %6d cand  mean %.1f us  blk%%8 = %s" % (k // 16, k % 16, m.sum(), tail[m].sum(), ((r1[m] - r0[m]) / 100.0).mean(), np.bincount(blk[m] % 8, minlength=8).tolist()))

xcd = blk % 8
clk = dur.astype(np.float64) / np.maximum((r1 - r0) / 100.0, 1e-3)
print("per XCD (blk % 8): waves / candidates / mean wave us / stage cycles per candidate / median clock MHz / last end us")
for x in range(8):
    m = xcd == x
    print("  xcd%d: %4d waves %6d cand  mean %.1f us  stage/cand %.0f  clk %.0f  end %.1f" % (
        x, m.sum(), tail[m].sum(), ((r1[m] - r0[m]) / 100.0).mean(), stage[m].sum() / max(1, tail[m].sum()), np.median(clk[m & (dur > 20000)]) if (m & (dur > 20000)).any() else 0,
        (r1[m].max() - t0) / 100.0))
# concurrency over time: live waves per 10 us
edges = np.arange(0, (r1.max() - t0) / 100.0 + 10, 10)
print("live waves at t =", [int(((r0 - t0) / 100.0 <= e).sum() - ((r1 - t0) / 100.0 <= e).sum()) for e in edges])
