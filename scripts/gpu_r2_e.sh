#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
for V in ${VARIANTS:-A}; do
  LIB=$R/build/variants/$V/libsvgir_raster.so
  [ $V = A ] && LIB=$R/svg-ir_amd/libsvgir_raster.so
  for W in ${WORKLOADS:-cfg2 cfg3_train cfg3_eval}; do
    SVGIR_RASTER_LIB=$LIB timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-shade > gpurun_out/r2e_bench_${V}_$W.json 2> gpurun_out/r2e_bench_${V}_$W.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2e_bench_${V}_$W.json").read().strip().splitlines()[-1])
    print("$V $W ms/step %.3f"%d["ms_per_step"], {k:v for k,v in d["stage_ms"].items() if k in ("cull","render","render_bwd","grad_reduce")})
except Exception as e:
    print("$V $W FAILED", e); print(open("gpurun_out/r2e_bench_${V}_$W.err").read()[-1500:])
PY
  done
done
for W in ${TRACE:-cfg2}; do
SVGIR_RASTER_LIB=$R/build/variants/dev/libsvgir_raster.so timeout 300 python scripts/dev_trace.py $W 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2e_trace_$W.log
done
