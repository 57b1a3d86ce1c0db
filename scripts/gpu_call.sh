#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=g11 TESTS=1 WORKLOADS="cfg2 cfg2" bash scripts/gpu_iter.sh
