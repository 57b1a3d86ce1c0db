#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrency.py -q -x 2>&1 | tail -2
timeout 600 python scripts/stress.py 5 40 2>&1 | grep -E "FAIL|stress passed"
W=cfg3_train REPEATS=7 BENCH_FLAGS=" " bash scripts/gpu_variants.sh
python - <<'PY'
import json
for w in ("cfg3_train",):
    d=json.loads(open(f"gpurun_out/var_product_{w}.json").read().strip().splitlines()[-1]); print(w, d["ms_per_step"], {k:round(v*1e3,1) for k,v in d["stage_ms"].items()})
PY
