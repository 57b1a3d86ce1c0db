#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in product ident; do
  if [ $V = product ]; then unset SVGIR_RASTER_LIB; else export SVGIR_RASTER_LIB=$PWD/build/variants/$V/libsvgir_raster.so; fi
  for W in cfg2 cfg3_train cfg5; do
  timeout 300 python bench.py --workload $W --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-shaded --no-concurrent --no-shade 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V $W', d['ms_per_step'], 'render', d['stage_ms']['render'], 'cull', d['stage_ms']['cull'])"
  done
done
