#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_view_parallel.py tests/test_gpu_pbgi.py tests/test_gpu_bvh.py -q -x 2>&1 | tail -8 | tee gpurun_out/c6_tests.log
python bench.py --workload train_step --steps 20 --warmup 5 --repeats 10 > gpurun_out/c6_train_step.json 2> gpurun_out/c6_train_step.err; tail -c 3000 gpurun_out/c6_train_step.json; tail -3 gpurun_out/c6_train_step.err
python bench.py --workload tracers > gpurun_out/c6_tracers.json 2> gpurun_out/c6_tracers.err; tail -c 3000 gpurun_out/c6_tracers.json; tail -3 gpurun_out/c6_tracers.err
for W in cfg2 cfg3_train; do scripts/pmc_issue.sh $W c6_$W > /dev/null 2>&1; cp gpurun_out/c6_${W}_issue.json gpurun_out/issue_$W.json; head -c 600 gpurun_out/issue_$W.json; echo; done
