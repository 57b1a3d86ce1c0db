"""Per-wave timeline of the backward composite (development builds only: scripts/build_variant.sh dev -DSVGIR_DEV).
    SVGIR_RASTER_LIB=build/variants/dev/libsvgir_raster.so python scripts/dev_trace_bwd.py [workload]"""
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
sc = scenes.make(wl)
sct = runner.to_torch(sc, dev)
grads = scenes.upstream_grads(sc, variant)
lib = _native.lib
reader = lib.svgir_dev_trace_read_bwd_plain if variant == "rgss" and hasattr(lib, "svgir_dev_trace_read_bwd_plain") else lib.svgir_dev_trace_read_bwd
reader.restype = C.c_int
reader.argtypes = [C.c_void_p, C.c_int]
CAP = 1 << 17
buf = np.zeros((CAP, 8), dtype=np.uint64)
for it in range(3):
    out, leaves = runner.render(sct, variant, requires_grad=True)
    runner.backward(out, grads, variant)
    torch.cuda.synchronize()
    n = reader(buf.ctypes.data, CAP)
rec = buf[:n].astype(np.int64)
dur, r0, r1 = rec[:, 0], rec[:, 1], rec[:, 2]
items, cands = rec[:, 3] >> 32, rec[:, 3] & 0xffffffff
setup, stage = rec[:, 5], rec[:, 6]
phA, phB = rec[:, 7] >> 32, rec[:, 7] & 0xffffffff
t_begin, t_end = r0.min(), r1.max()
print(f"{wl} backward: {n} waves with work; kernel span {(t_end - t_begin) / 100.0:.1f} us; segments {items.sum()}, candidates {cands.sum()}")
print("start delay (us): p50 %.1f p90 %.1f max %.1f" % tuple(np.quantile((r0 - t_begin) / 100.0, [0.5, 0.9, 1.0])))
print("end time (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.quantile((r1 - t_begin) / 100.0, [0.5, 0.9, 0.99, 1.0])))
print("wave duration (us): p50 %.1f p90 %.1f max %.1f ; items per wave max %d" % (*np.quantile((r1 - r0) / 100.0, [0.5, 0.9, 1.0]), items.max()))
tot = dur.sum()
print("cycle shares: setup %.2f stage/epilogue %.2f phaseA %.2f phaseB(+loop) %.2f" % (setup.sum() / tot, stage.sum() / tot, phA.sum() / tot, phB.sum() / tot))
print("cycles per candidate: total %.0f phaseA %.0f phaseB %.0f stage %.0f ; setup cycles per segment %.0f" % (
    tot / cands.sum(), phA.sum() / cands.sum(), phB.sum() / cands.sum(), stage.sum() / cands.sum(), setup.sum() / items.sum()))
mid = (t_begin + t_end) // 2
print("waves alive at mid-kernel:", ((r0 <= mid) & (r1 >= mid)).sum())
