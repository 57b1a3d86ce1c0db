"""Per-wave timeline of the forward composite (development builds only: scripts/build_variant.sh dev -DSVGIR_DEV).
    SVGIR_RASTER_LIB=build/variants/dev/libsvgir_raster.so python scripts/dev_trace.py [workload]"""
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
scan, stage = rec[:, 5], rec[:, 6]
blend, epi = rec[:, 7] >> 32, rec[:, 7] & 0xffffffff
t_begin, t_end = r0.min(), r1.max()
print(f"{wl}: {n} waves traced; kernel span {(t_end - t_begin) / 100.0:.1f} us (100 MHz realtime)")
ne = ln > 0
print(f"non-empty waves {ne.sum()}, total candidates {tail.sum()}, entries scanned(len) {ln[ne].sum()}")
clk = dur.astype(np.float64) / np.maximum((r1 - r0) / 100.0, 1e-3)   # shader cycles per us
print(f"shader clock ~ {np.median(clk[ne & (dur > 20000)]):.0f} MHz")
print("start delay (us) of non-empty waves: p50 %.1f p90 %.1f max %.1f" % tuple(np.quantile((r0[ne] - t_begin) / 100.0, [0.5, 0.9, 1.0])))
print("end time (us) of non-empty waves: p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.quantile((r1[ne] - t_begin) / 100.0, [0.5, 0.9, 0.99, 1.0])))
print("wave duration (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.quantile((r1[ne] - r0[ne]) / 100.0, [0.5, 0.9, 0.99, 1.0])))
tot = dur[ne].sum()
print("cycle shares: scan %.2f stage %.2f blend %.2f epilogue %.2f" % (scan[ne].sum() / tot, stage[ne].sum() / tot, blend[ne].sum() / tot, epi[ne].sum() / tot))
print("cycles per candidate (blend) %.0f ; cycles per scan step %.0f ; cycles per staged candidate %.0f" % (
    blend[ne].sum() / max(1, tail.sum()), scan[ne].sum() / max(1, np.ceil(ln[ne] / 64).sum()), stage[ne].sum() / max(1, tail.sum())))
order = np.argsort(-(r1 - t_begin))[:12]
print("last-finishing waves:  blk   len  cand  start_us  end_us  dur_us   scan  stage  blend (kcycles)   cu")
for i in order:
    print("   %6d %5d %5d %8.1f %7.1f %7.1f %6.1f %6.1f %6.1f   se%d cu%d simd%d" % (
        blk[i], ln[i], tail[i], (r0[i] - t_begin) / 100.0, (r1[i] - t_begin) / 100.0, (r1[i] - r0[i]) / 100.0,
        scan[i] / 1e3, stage[i] / 1e3, blend[i] / 1e3, (hw[i] >> 13) & 7, (hw[i] >> 8) & 15, (hw[i] >> 4) & 3))
# waves per SIMD concurrently alive at the midpoint
mid = (t_begin + t_end) // 2
alive = (r0 <= mid) & (r1 >= mid)
print("waves alive at mid-kernel:", alive.sum())
