"""Per-wave phase split of shade_bwd_kernel (development builds only: scripts/build_variant.sh dev -DSVGIR_DEV).
    SVGIR_RASTER_LIB=build/variants/dev/libsvgir_raster.so python scripts/dev_trace_shade.py [workload]
Only wave 0 of every workgroup reports (a sample of 1 / SHADE_BWAVES of the waves)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svg-ir_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gaussian_renderer import _native

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_train"
args = argparse.Namespace(no_shade=False, samples=0, streamed_dirs=False)
dev = torch.device("cuda:0")
wl = bench.Workload(name, dev, 0, 1, args)
reader = _native.lib.svgir_dev_trace_read_shade
reader.restype = C.c_int
reader.argtypes = [C.c_void_p, C.c_int, C.c_int]
CAP = 1 << 17
buf = np.zeros((CAP, 8), dtype=np.uint64)
fbuf = np.zeros((CAP, 8), dtype=np.uint64)
for it in range(3):
    wl.step()
    torch.cuda.synchronize()
    nf = reader(fbuf.ctypes.data, CAP, 1)
    n = reader(buf.ctypes.data, CAP, 0)
fr = fbuf[:nf].astype(np.int64)
ft = fr[:, 0].sum()
if nf == 0:
    print(f"{name} shade_fwd: no records (the quad kernel of the training sample counts carries no probes)")
else:
  print(f"{name} shade_fwd: {nf} reporting waves (one of four); kernel span {(fr[:, 2].max() - fr[:, 1].min()) / 100.0:.1f} us; cycles per surfel {ft / nf:.0f}: "
      f"prologue {fr[:, 5].sum() / nf:.0f} staging {fr[:, 6].sum() / nf:.0f} corner-x-sample loop {(fr[:, 7] >> 32).sum() / nf:.0f} "
      f"reductions + epilogue {(fr[:, 7] & 0xffffffff).sum() / nf:.0f}")
rec = buf[:n].astype(np.int64)
dur, r0, r1 = rec[:, 0], rec[:, 1], rec[:, 2]
surfels = rec[:, 3] >> 32
pro, stage = rec[:, 5], rec[:, 6]
loop, epi = rec[:, 7] >> 32, rec[:, 7] & 0xffffffff
t_begin, t_end = r0.min(), r1.max()
print(f"{name} shade_bwd: {n} reporting waves; kernel span {(t_end - t_begin) / 100.0:.1f} us; surfels per reporting wave p50 {np.median(surfels):.0f}")
print("wave duration (us): p50 %.1f p90 %.1f max %.1f" % tuple(np.quantile((r1 - r0) / 100.0, [0.5, 0.9, 1.0])))
tot = dur.sum()
print("cycle shares: prologue %.3f staging %.3f corner-x-sample loop %.3f row stores + epilogue %.3f (sum %.3f)" % (
    pro.sum() / tot, stage.sum() / tot, loop.sum() / tot, epi.sum() / tot, (pro.sum() + stage.sum() + loop.sum() + epi.sum()) / tot))
ns = surfels.sum()
print("cycles per surfel: total %.0f prologue %.0f staging %.0f loop %.0f epilogue %.0f" % (tot / ns, pro.sum() / ns, stage.sum() / ns, loop.sum() / ns, epi.sum() / ns))
