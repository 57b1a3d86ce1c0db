#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/c15_tests.log
W=cfg3_train REPEATS=7 BENCH_FLAGS=" " bash scripts/gpu_variants.sh
W=cfg2 REPEATS=9 bash scripts/gpu_variants.sh
python - <<'PY'
import json
for w in ("cfg2","cfg3_train"):
    d=json.loads(open(f"gpurun_out/var_product_{w}.json").read().strip().splitlines()[-1]); print(w, d["ms_per_step"], {k:round(v*1e3,1) for k,v in d["stage_ms"].items()})
PY
timeout 600 python scripts/stress.py 5 40 2>&1 | tail -2
