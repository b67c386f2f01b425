"""ctypes binding of libdavfusion_hip.so (the C ABI declared in include/dav_kernels.h).

The product path has NO fallback: if the shared library is missing or a kernel
returns an error code, a RuntimeError is raised.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C deepavfusion_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must be imported first: the library has to bind to the HIP runtime torch ships,
#                      a second libamdhip64 in the process sees no device)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdavfusion_hip.so')

_p, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_size_t

ABI_VERSION = 9      # DAV_ABI_VERSION of include/dav_kernels.h this package was written against (struct layouts, signatures)

# name -> argtypes (must match include/dav_kernels.h)
SIGNATURES = {
    'dav_abi_version': [],
    'dav_build_flags': [],
    'dav_last_error_string': [],
    'dav_tune': [_i, _i],
    'dav_gemm_nt_bf16': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _f, _i, _p],
    'dav_gemm_nt_ln_bf16': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _f, _i, _p, _p],
    'dav_ln_fold_grouped': [_p, _i, _p],
    'dav_rowstats_cast': [_p, _l, _i, _i, _i, _p, _p, _p],
    'dav_layernorm_bwd_twin': [_p, _l, _p, _i, _p, _l, _p, _i, _i, _i, _f, _p, _p, _p, _p,
                               _p, _l, _i, _p, _l, _p, _l,
                               _p, _l, _i, _p, _l, _p, _l,
                               _p, _p, _p, _p, _sz, _p],
    'dav_nt_issue_log': [_i, _p, _i],
    'dav_nt_tune_set': [_p, _i],
    'dav_gemm_tn_bf16': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _i, _p],
    'dav_gemm_tn_grouped_bf16': [_p, _i, _p],
    'dav_gemm_tn_gang_workspace_bytes': [_p, _i],
    'dav_gemm_tn_gang_bf16': [_p, _i, _p, _sz, _p],
    'dav_attn_fwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p],
    'dav_attn_bwd': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _p],
    'dav_attn_bwd_part': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _i, _p],
    'dav_add_cast': [_p, _p, _p, _p, _l, _p],
    'dav_attn_bwd_ctx': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _i, _i, _p],
    'dav_attn_bias_fwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p, _i, _i, _p],
    'dav_attn_bias_bwd': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _p, _i, _i, _p, _i, _p],
    'dav_attn_drop_fwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p, _i, _f, _p],
    'dav_attn_drop_bwd': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _p, _i, _f, _i, _i, _p],
    'dav_window_unfold': [_p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _p, _i, _p],
    'dav_window_fold': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p],
    'dav_relpos_bias_build': [_p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _p],
    'dav_relpos_bias_bwd': [_p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    'dav_layernorm_fwd': [_p, _l, _i, _p, _l, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p],
    'dav_layernorm_bwd': [_p, _l, _i, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p,
                          _p, _l, _i, _p, _l, _p, _l,
                          _p, _l, _i, _p, _l, _p, _l,
                          _p, _p, _p, _sz, _p],
    'dav_layernorm_bwd_workspace_bytes': [_i, _i],
    'dav_layernorm_bwd_reduce_grouped': [_p, _i, _p],
    'dav_mask_build': [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    'dav_patch_gather': [_p, _i, _i, _i, _i, _p, _i, _p, _p],
    'dav_patch_gather3d': [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p],
    'dav_unshuffle_fwd': [_p, _p, _p, _p, _i, _i, _i, _i, _p, _l, _i, _p],
    'dav_rows_gather_cast': [_p, _l, _i, _p, _i, _i, _i, _p, _p],
    'dav_rows_axpy': [_p, _p, _p, _i, _i, _i, _p, _p],
    'dav_rows_scale_cast': [_p, _p, _i, _i, _i, _p, _p],
    'dav_dropout_rows': [_p, _i, _p, _p, _f, _p, _i, _i, _i, _p, _i, _p],
    'dav_unshuffle_bwd_reduce': [_p, _l, _i, _p, _i, _i, _i, _i, _p, _p, _p],
    'dav_patch_mse_fwd': [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    'dav_patch_mse_bwd': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    'dav_pair_expand': [_p, _p, _i, _i, _i, _i, _p, _p],
    'dav_pair_reduce': [_p, _i, _i, _i, _i, _p, _p, _p],
    'dav_cast_bf16': [_p, _p, _l, _p],
    'dav_cast_transpose_bf16': [_p, _p, _i, _i, _p],
    'dav_l2norm_workspace_bytes': [_l],
    'dav_l2norm': [_p, _l, _f, _p, _p, _sz, _p],
    'dav_gemm_nt_f32': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _i, _p, _p, _p, _i, _p, _p, _i, _i, _i, _f, _i, _p],
    'dav_gemm_tn_f32': [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p],
    'dav_attn_fwd_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p],
    'dav_attn_bwd_f32': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _i, _p],
    'dav_attn_bias_fwd_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p, _i, _i, _p],
    'dav_attn_bias_bwd_f32': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _p, _i, _i, _p, _i, _p],
    'dav_attn_drop_fwd_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _l, _i, _l, _i, _l, _i, _l, _i, _f, _p, _i, _f, _p],
    'dav_attn_drop_bwd_f32': [_p] * 10 + [_i] * 6 + [_l, _i] * 8 + [_f, _p, _i, _f, _i, _p],
    'dav_patch_gather_f32': [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p],
    'dav_rows_gather_f32': [_p, _l, _i, _p, _i, _i, _i, _p, _l, _p],
    'dav_pair_expand_f32': [_p, _p, _i, _i, _i, _i, _p, _p],
    'dav_pair_reduce_f32': [_p, _i, _i, _i, _i, _p, _p, _p],
    'dav_patch_mse_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p],
    'dav_add_f32': [_p, _p, _p, _l, _p],
    'dav_logmel': [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _f, _i, _i, _p, _p],
    'dav_log10_eps': [_p, _f, _l, _p, _p],
    'dav_batch_begin': [_i],
    'dav_batch_lane': [],
    'dav_batch_region': [_i],
    'dav_batch_skip': [_i],
    'dav_batch_suspend': [_i],
    'dav_batch_end': [],
    'dav_batch_abort': [],
    'dav_batch_stats': [_p, _p],
    'dav_adamw_flat': [_p, _p, _p, _p, _p, _l, _p, _p, _i, _f, _f, _f, _p, _f, _p, _i, _p, _p, _p],
    'dav_step_guard': [_p, _p, _p, _f, _f, _p, _p, _p],
    'dav_cast_transpose_grouped': [_p, _i, _p],
}

class DavLnReduce(C.Structure):
    _fields_ = [('workspace', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('rows', C.c_int), ('D', C.c_int)]


class DavNtLn(C.Structure):
    _fields_ = [('stats', C.c_void_p), ('stats2', C.c_void_p), ('A2', C.c_void_p), ('ln_c', C.c_void_p), ('eps', C.c_float),
                ('a_r0', C.c_int), ('a_r1', C.c_int), ('stats_out', C.c_void_p), ('twin_out', C.c_void_p), ('ld_twin', C.c_int)]


class DavLnFold(C.Structure):
    _fields_ = [('w', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('bias', C.c_void_p),
                ('w_ln_bf16', C.c_void_p), ('ln_c', C.c_void_p), ('ln_d', C.c_void_p), ('N', C.c_int), ('K', C.c_int)]


class DavTranspose(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y_bf16', C.c_void_p), ('R', C.c_int), ('C', C.c_int)]


class DavTnProblem(C.Structure):
    _fields_ = [('A', C.c_void_p), ('B', C.c_void_p), ('C', C.c_void_p), ('bias_grad', C.c_void_p),
                ('Mc', C.c_int), ('N', C.c_int), ('K', C.c_int), ('lda', C.c_int), ('ldb', C.c_int), ('ldc', C.c_int),
                ('a_rowmap', C.c_int * 3), ('b_rowmap', C.c_int * 3), ('flags', C.c_int)]


ERRORS = {-1: 'bad shape', -2: 'unsupported dtype', -3: 'insufficient workspace', -4: 'HIP error', -5: 'misaligned pointer/stride'}

_lib = None


def load():
    """Load the HIP library once; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: the DeepAVFusion HIP kernels are not built '
                           '(run __graft_entry__.build() or make -C deepavfusion_amd/csrc). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _sz if name.endswith('workspace_bytes') else (C.c_char_p if name == 'dav_last_error_string' else _i)
    if lib.dav_abi_version() != ABI_VERSION:
        raise RuntimeError('libdavfusion_hip.so ABI version mismatch')
    _lib = lib
    load_nt_tuning(NT_TUNING_PATH)
    return lib


def kernel_source_hash():
    """sha1 over the kernel sources and the tuned tile-configuration table: the PMC-derived constants under profiles/ (HBM traffic of the
    dominant kernel, fabric bytes per step) carry the hash of the tree they were measured on, and bench.py reports them only while it
    still matches (tools/traffic_json.py, tools/step_traffic.py write it)."""
    import glob
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(here, 'csrc', '*.hip')) + glob.glob(os.path.join(here, 'csrc', '*.h'))) + [os.path.join(here, 'tuning', 'nt_gfx950.json')]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


NT_TUNING_PATH = os.environ.get('DAV_NT_TUNE_FILE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuning', 'nt_gfx950.json')


def load_nt_tuning(path):
    """Hands the tuned tile-configuration table (tools/mix_sweep.py --write) to the library; DAV_NT_TUNE=0 or a missing file
    leave the built-in rules alone.  Returns the number of entries installed."""
    if os.environ.get('DAV_NT_TUNE', '1') == '0' or not path or not os.path.exists(path):
        _lib.dav_nt_tune_set(None, 0)
        return 0
    import json
    blob = []
    for e in json.load(open(path))['entries']:
        blob += [int(e['cfg']), int(e['b_kn']), len(e['problems'])]
        for q in e['problems']:
            blob += [int(x) for x in q]            # M, N, K, epilogue flags
    arr = (C.c_int * len(blob))(*blob)
    n = _lib.dav_nt_tune_set(arr, len(blob))
    if n < 0:
        raise RuntimeError(f'{path}: malformed tuning table ({ERRORS.get(n, n)})')
    return n


def check(code: int, what: str):
    if code != 0:
        detail = ''
        if code == -4 and _lib is not None:
            detail = f" ({_lib.dav_last_error_string().decode()})"
        raise RuntimeError(f'{what} failed: {ERRORS.get(code, code)}{detail}')
