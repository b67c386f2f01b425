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


def test_kernel_shape_fuzz():
    """A fixed-seed slice of tests/gpu_fuzz.py: random GEMM shapes with random epilogue options (+ their weight gradients),
    random attention lengths / head widths, random two-segment LayerNorms — each against a torch fp32 reference."""
    import random

    import torch

    import gpu_fuzz as fz
    fz.FAILS.clear()
    rng = random.Random(7)
    torch.manual_seed(7)
    fz.fuzz_gemm(rng, 60)
    fz.fuzz_attn(rng, 25)
    fz.fuzz_ln(rng, 25)
    assert not fz.FAILS, fz.FAILS[:10]
