#!/bin/bash
# GPU-side: rebuild render_bwd.o with in-kernel s_memtime phase counters and print the phase split.  usage: bwd_timing.sh <workload>
W=${1:-cfg2}
cd svg-ir_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -I../../include -DRENDER_TIMING ${RENDER_DEFS:-} -c render_bwd.hip -o render_bwd.o 2>&1 | grep -E "error" -A5
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsvgir_raster.so api.o binning.o geom_bwd.o grad_reduce.o image_ops.o preprocess.o render_bwd.o render_fwd.o shade.o
cd ../..
python - $W <<'PY'
import ctypes as C, sys, json, subprocess, io, contextlib
sys.path.insert(0, "svg-ir_amd"); sys.path.insert(0, ".")
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "10", "--warmup", "2", "--workload", sys.argv[1], "--no-shade"]
import runpy
from gaussian_renderer import _native as N
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
r = json.loads(buf.getvalue())
out = (C.c_ulonglong * 16)()
N.lib.svgir_debug_bwd_timing(out, 1)
names = ["setup", "staging", "phase A", "phase B"]
tot = sum(out[:4]); waves = out[7]; cand = out[5]; live = out[6]
print(f"render_bwd {r['stage_ms']['render_bwd']:.3f} ms; waves {waves/12:.0f}/launch, staged candidates {cand/12:.0f}/launch, live {live/12:.0f}/launch")
for nm, v in zip(names, out[:4]):
    print(f"  {nm:10s} {v / max(waves,1):10.0f} cyc/wave  {v / max(cand,1):8.0f} cyc/staged-candidate  {100.0 * v / tot:5.1f}%")
print(f"  total      {tot / max(waves,1):10.0f} cyc/wave  {tot / max(cand,1):8.0f} cyc/staged-candidate")
if sum(out[8:12]):
    for nm, v in zip(["A: loop+slot", "A: alpha", "A: replay math", "A: panel/octant"], out[8:12]):
        print(f"  {nm:16s} {v / max(cand,1):8.0f} cyc/staged-candidate  {v / max(live,1):8.0f} cyc/live-candidate")
PY
