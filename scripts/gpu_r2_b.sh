#!/bin/bash
set -u
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
SVGIR_RASTER_LIB=$R/build/variants/dev/libsvgir_raster.so timeout 300 python scripts/dev_trace.py cfg2 2>&1 | tee gpurun_out/r2b_trace_cfg2.log
SVGIR_RASTER_LIB=$R/build/variants/dev/libsvgir_raster.so timeout 300 python scripts/dev_trace.py cfg3_train 2>&1 | tee gpurun_out/r2b_trace_cfg3.log
