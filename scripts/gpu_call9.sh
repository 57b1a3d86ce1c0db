#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pbgi.py -q -x 2>&1 | tail -3 | tee gpurun_out/c9_tests.log
for POOL in 64 256 1024; do
echo "chunk $POOL"; SVGIR_PBGI_POOL=$POOL timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | grep -E "update_radiance|shell scene"
done | tee gpurun_out/c9_pool.log
