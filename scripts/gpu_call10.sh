#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for V in product lds24 lds16 lds16w5; do
if [ "$V" = product ]; then unset SVGIR_RASTER_LIB; else export SVGIR_RASTER_LIB=$PWD/build/variants/$V/libsvgir_raster.so; fi
echo "variant $V"; SVGIR_PBGI_POOL=64 timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | grep -E "update_radiance|shell scene"
done | tee gpurun_out/c10_lds.log
