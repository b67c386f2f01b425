"""DAV_BENCH_LIB=<path of another libdavfusion_hip.so>: the micro benches load that build instead of the tree's (same-box A/B against
an older library, e.g. tools/runs_r05/lib_r04/ built by tools/runs_r05/build_r04_lib.sh).  Import before deepavfusion_amd.ops."""
import ctypes
import os

from deepavfusion_amd import _lib

if os.environ.get('DAV_BENCH_LIB'):
    _lib.LIB_PATH = os.environ['DAV_BENCH_LIB']
    _old = ctypes.CDLL(_lib.LIB_PATH)
    _lib.ABI_VERSION = _old.dav_abi_version()
    for _k in [k for k in _lib.SIGNATURES if not hasattr(_old, k)]:
        del _lib.SIGNATURES[_k]
