#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/c1_tests.log
W=cfg2 REPEATS=5 bash scripts/gpu_variants.sh pfl2 pfk pfboth
python scripts/train_step_probe.py 2>&1 | tail -6 | tee gpurun_out/c1_train.log
SVGIR_RASTER_LIB=$PWD/build/variants/dev/libsvgir_raster.so timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | tail -12 | tee gpurun_out/c1_tracer.log
