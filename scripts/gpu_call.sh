#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=g10 TESTS=1 WORKLOADS="cfg3_train cfg5 cfg5_dense" bash scripts/gpu_iter.sh
timeout 600 python scripts/stress.py 9 30 2>&1 | tail -2
