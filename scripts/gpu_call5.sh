#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for POOL in 256 1024 4096 16384; do
echo "pool $POOL"; SVGIR_PBGI_POOL=$POOL timeout 600 python scripts/tracer_cfg3_probe.py 200000 shell 2>&1 | grep -E "update_radiance|shell scene"
done | tee gpurun_out/c5_pool.log
