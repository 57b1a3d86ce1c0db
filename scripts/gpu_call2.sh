#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pbgi.py tests/test_gpu_bvh.py -q -x 2>&1 | tail -8 | tee gpurun_out/c2_tests.log
SVGIR_RASTER_LIB=$PWD/build/variants/dev/libsvgir_raster.so timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | tail -12 | tee gpurun_out/c2_tracer_dev.log
timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | tail -12 | tee gpurun_out/c2_tracer.log
