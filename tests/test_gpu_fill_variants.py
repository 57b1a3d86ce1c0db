"""The composite forward's launch variants (csrc/api.hip high_fill; csrc/render_fwd.hip; csrc/binning.hip order_xcd_kernel).

A launch that fills the machine several times over runs the forward composite at three waves per SIMD with one longest-first dispatch
list per XCD; everything else runs at two waves per SIMD with one global list.  The choice comes from the workload's history (the
non-empty sub-tiles of its latest view) and never changes a result.  The small scenes of the suite never reach the high-fill threshold on
their own, so the variants are forced here through the environment (read once per process: hence the child processes) and run against
the oracle: random scenes of both rasterizers, every specialised width, deep translucent stacks, and the fused shading."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fill,xcd", [("1", "1"), ("1", "0"), ("0", "1")])
def test_forced_variants_match_the_oracle(built, fill, xcd):
    env = dict(os.environ, SVGIR_FWD_FILL=fill, SVGIR_FWD_XCD=xcd)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress.py"), "11", "14"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "failed: 0" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_fused.py"), "5", "6"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "failed: 0" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
