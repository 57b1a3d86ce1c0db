#!/bin/bash
# round-2 session a: parity tests of the batched composite kernels + KB/WPE variants
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r2a_tests.log
cat gpurun_out/r2a_tests.log
for V in A B C D; do
  LIB=$R/build/variants/$V/libsvgir_raster.so
  [ $V = A ] && LIB=$R/svg-ir_amd/libsvgir_raster.so
  for W in cfg2 cfg3_train cfg3_eval; do
    SVGIR_RASTER_LIB=$LIB timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r2a_bench_${V}_$W.json 2> gpurun_out/r2a_bench_${V}_$W.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r2a_bench_${V}_$W.json").read().strip().splitlines()[-1])
    print("$V $W ms/step %.3f"%d["ms_per_step"], {k:v for k,v in d["stage_ms"].items() if k in ("render","render_bwd","grad_reduce","shade_fwd","shade_bwd")})
except Exception as e:
    print("$V $W FAILED", e); print(open("gpurun_out/r2a_bench_${V}_$W.err").read()[-1500:])
PY
  done
done
