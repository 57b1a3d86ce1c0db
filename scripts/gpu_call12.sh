#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pbgi.py -q -x 2>&1 | tail -3 | tee gpurun_out/c12_tests.log
for V in product coop32; do
if [ "$V" = product ]; then unset SVGIR_RASTER_LIB; else export SVGIR_RASTER_LIB=$PWD/build/variants/$V/libsvgir_raster.so; fi
echo "variant $V"; python -m pytest tests/test_gpu_pbgi.py -q -x 2>&1 | tail -1; timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | grep -E "update_radiance|shell scene"
done | tee gpurun_out/c12_coop.log
