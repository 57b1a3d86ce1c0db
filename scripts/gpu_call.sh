#!/bin/bash
# scratch: one GPU iteration (edited per call)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['metric'][:60], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('issue_frac'), d['roofline'].get('bound'), d.get('value_shaded'), d['cpu_baseline']['value'])"
