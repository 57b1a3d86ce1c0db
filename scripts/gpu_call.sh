#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
TAG=g4 TESTS=0 WORKLOADS="cfg2 cfg2 cfg3_train" bash scripts/gpu_iter.sh
SVGIR_NO_KEY_SPEC=1 TAG=g4n TESTS=0 WORKLOADS="cfg2" bash scripts/gpu_iter.sh
