"""GPU: every HIP kernel, called through the C ABI, against a torch fp32 reference of the same op
(tolerances are in tests/gpu_selfcheck.py next to each check; index work is bit-exact)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['gemm_nt', 'gemm_tn', 'attention', 'layernorm', 'masking', 'misc_kernels', 'patch_gather3d'])
def test_kernel_family(name):
    import gpu_selfcheck as sc
    sc.RESULTS.clear()
    getattr(sc, name)()
    bad = [r for r in sc.RESULTS if not r[3]]
    assert sc.RESULTS and not bad, bad[:10]
