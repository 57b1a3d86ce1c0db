#!/bin/bash
# GPU iteration: parity tests, then short bench lines of the named workloads (stage table included).
#   gpurun -- 'TAG=x WORKLOADS="cfg2 cfg3_train" bash scripts/gpu_iter.sh'
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${TAG:-it}
if [ "${TESTS:-1}" = 1 ]; then timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -15; fi
for W in ${WORKLOADS:-cfg2 cfg3_train cfg3_eval}; do
  timeout 300 python bench.py --workload $W --steps 30 --warmup 5 --repeats ${REPEATS:-9} --no-cpu-baseline --no-shaded ${BENCH_FLAGS:---no-shade} \
      > gpurun_out/${TAG}_bench_$W.json 2> gpurun_out/${TAG}_bench_$W.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench_$W.json").read().strip().splitlines()[-1])
    print("$W ms/step %.4f value %.4g"%(d["ms_per_step"], d["value"]), json.dumps(d.get("stage_ms")))
except Exception as e:
    print("$W FAILED", e); print(open("gpurun_out/${TAG}_bench_$W.err").read()[-1500:])
PY
done
R=$PWD
for W in ${PROFILE:-}; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$W -o p -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-shaded --no-shade --workload $W > $R/gpurun_out/${TAG}_prof_$W.log 2>&1)
  DB=$(find gpurun_out/${TAG}_prof_$W -name "*.db" | head -1)
  python scripts/rocprof_summary.py $DB gpurun_out/${TAG}_${W}_kernel_stats.txt "python3 bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-shaded --no-shade --workload $W" | cut -c1-170 | head -34
done
