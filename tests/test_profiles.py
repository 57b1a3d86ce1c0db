"""The committed PMC summaries must describe the committed kernels: `bench.py` reports `roofline.traffic` / `issue_frac` only when the
summary's kernel-source hash (and the hash of the built library) match -- a stale summary would silently turn them into null."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("cfg2", "cfg3_train", "cfg3_eval", "cfg4", "cfg5")


def _bench():
    sys.path.insert(0, ROOT)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench
    finally:
        sys.argv = argv
    return bench


@pytest.mark.parametrize("kind", ["traffic", "issue"])
def test_pmc_summaries_match_the_kernel_sources(kind):
    bench = _bench()
    want = bench.kernel_source_hash()
    for w in WORKLOADS:
        path = os.path.join(ROOT, "profiles", f"{kind}_{w}.json")
        assert os.path.exists(path), path
        rec = json.load(open(path))
        assert rec["kernel_source_hash"] == want, f"{path} was measured on other kernel sources: re-run scripts/gpu_round.sh"
        assert rec["workload"] == w and rec["kernels"], path
