#!/bin/bash
# Bench line (stage table) of one workload for the product library and for every experiment build named:
#   gpurun -- 'W=cfg2 bash scripts/gpu_variants.sh v1 v2 ...'      (build/variants/<name>/libsvgir_raster.so)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
W=${W:-cfg2}
for V in product "$@"; do
  if [ "$V" = product ]; then unset SVGIR_RASTER_LIB; else export SVGIR_RASTER_LIB=$PWD/build/variants/$V/libsvgir_raster.so; fi
  timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --repeats ${REPEATS:-7} --no-cpu-baseline --no-shaded --no-concurrent ${BENCH_FLAGS:---no-shade} \
      > gpurun_out/var_${V}_$W.json 2> gpurun_out/var_${V}_$W.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/var_${V}_$W.json").read().strip().splitlines()[-1])
    s=d.get("stage_ms",{})
    print("%-10s $W ms/step %.4f | "%("$V", d["ms_per_step"]) + " ".join("%s %.1f"%(k,1e3*v) for k,v in s.items() if k in ("cull","render","render_bwd","grad_reduce","geom_bwd","sort_depth","sort_tile","shade_fwd","shade_bwd")))
except Exception as e:
    print("$V $W FAILED", e); print(open("gpurun_out/var_${V}_$W.err").read()[-800:])
PY
done 2>&1 | tee -a gpurun_out/variants.log
