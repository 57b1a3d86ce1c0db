#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
W=cfg2 REPEATS=15 bash scripts/gpu_variants.sh syncc
W=cfg3_train REPEATS=7 BENCH_FLAGS=" " bash scripts/gpu_variants.sh syncc
