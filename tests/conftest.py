import os
import sys

import pytest

# every buffer the library promises to write completely (ABI 5: forward outputs, gradient tensors) is NaN-filled by the
# binding layer before the call, so an element the kernels forget shows up in the parity checks
os.environ.setdefault("SVGIR_POISON", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "svg-ir_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Build (if needed) the HIP library and the oracle once per session."""
    import __graft_entry__ as g
    g.build()
    return True
