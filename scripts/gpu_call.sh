#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in product c512 c1024; do
  if [ $V = product ]; then unset SVGIR_RASTER_LIB; else export SVGIR_RASTER_LIB=$PWD/build/variants/$V/libsvgir_raster.so; fi
  for W in cfg5 cfg5_dense; do
    timeout 300 python bench.py --workload $W --steps 10 --warmup 5 --repeats 3 --no-cpu-baseline --no-shaded --no-concurrent --no-shade > gpurun_out/s_$W.json 2> gpurun_out/s_$W.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/s_$W.json").read().strip().splitlines()[-1])
    st=d.get("stage_ms") or {}
    print("$V $W ms/step %.4f"%d["ms_per_step"], {k:round(v,4) for k,v in st.items()})
except Exception as e:
    print("$V $W FAILED", e); print(open("gpurun_out/s_$W.err").read()[-1500:])
PY
  done
done
