#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=g1 TESTS=1 WORKLOADS="cfg2 cfg3_train" PROFILE="cfg2" bash scripts/gpu_iter.sh
rm -rf gpurun_out/g1_prof_*
