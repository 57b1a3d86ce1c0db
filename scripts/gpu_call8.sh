#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python -m pytest tests/test_gpu_parity.py -q -x -k "cfg5_dense" ) 2>&1 | grep -E "passed|failed|real|Error|assert" | tee gpurun_out/c8_dense.log
