#!/bin/bash
# A/B of run-time switches (environment variables) and / or experiment builds on the stage table of some workloads:
#   gpurun -- 'WORKLOADS="cfg5 cfg4" bash scripts/gpu_ab.sh base: xcd:SVGIR_FWD_XCD=1 "v3:SVGIR_RASTER_LIB=$PWD/build/variants/v3/libsvgir_raster.so"'
# every argument is  <label>:<VAR=value ...>  (nothing behind the colon = the product as it is)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for W in ${WORKLOADS:-cfg5}; do
  for spec in "$@"; do
    L=${spec%%:*}; E=${spec#*:}
    env $E timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --repeats ${REPEATS:-5} --no-cpu-baseline --no-shaded --no-concurrent ${BENCH_FLAGS:---no-shade} \
        > gpurun_out/ab_${L}_$W.json 2> gpurun_out/ab_${L}_$W.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_${L}_$W.json").read().strip().splitlines()[-1])
    s=d.get("stage_ms",{})
    print("%-12s %-10s ms/step %.4f | "%("$L", "$W", d["ms_per_step"]) + " ".join("%s %.1f"%(k,1e3*v) for k,v in s.items() if k in ("preprocess","cull","prepass","render","render_bwd","grad_reduce","geom_bwd","sort_depth","sort_tile","shade_fwd","shade_bwd","seg_build")))
except Exception as e:
    print("$L $W FAILED", e); print(open("gpurun_out/ab_${L}_$W.err").read()[-800:])
PY
  done
done 2>&1 | tee -a gpurun_out/ab.log
