#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=g9 TESTS=1 WORKLOADS="cfg2 cfg3_train" bash scripts/gpu_iter.sh
timeout 600 python scripts/stress.py 5 30 2>&1 | tail -2
timeout 600 python bench.py --workload train_step --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train_step', d['ms_per_step'], d.get('phase_ms'))"
