#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_concurrency.py -q -x 2>&1 | tail -3 | tee gpurun_out/c11_tests.log
W=cfg2 REPEATS=9 bash scripts/gpu_variants.sh
W=cfg3_train REPEATS=5 BENCH_FLAGS=" " bash scripts/gpu_variants.sh
python - <<'PY'
import json
for w in ("cfg2","cfg3_train"):
    d=json.loads(open(f"gpurun_out/var_product_{w}.json").read().strip().splitlines()[-1]); print(w, d["ms_per_step"], d["stage_ms"])
PY
