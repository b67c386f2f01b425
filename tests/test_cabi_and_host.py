"""CPU: the C-ABI library loads and exports every symbol include/dav_kernels.h declares; host logic
(lr schedule, param groups, sin-cos tables, state-dict contract, FLOP model) matches the golden fixtures."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from deepavfusion_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'dav_kernels.h')).read()
    declared = set(re.findall(r'\b(dav_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 20
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/dav_kernels.h but not exported'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.dav_abi_version() == _lib.ABI_VERSION == 9


def test_no_cpu_fallback():
    from deepavfusion_amd import ops
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.cast_bf16(x, torch.zeros(4, 8, dtype=torch.bfloat16))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'deepavfusion_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src, f'{f} mentions the oracle'
    # outside the package only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/
    import re
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for d in ('tools', 'configs'):
        for dirpath, _, files in os.walk(os.path.join(ROOT, d)):
            for f in files:
                if f.endswith('.py'):
                    assert not pat.search(open(os.path.join(dirpath, f)).read()), f'{d}/{f} imports the oracle'
    assert not pat.search(open(os.path.join(ROOT, 'train.py')).read())
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    assert len(pat.findall(bench)) == 2 and bench.index('from oracle') > bench.index('CPU baseline')


def test_state_dict_contract_matches_reference_shapes():
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from oracle import avmae_oracle as O
    from oracle.configs import CONFIGS as OC
    for name in ('micro', 'tiny'):
        m = build_avmae(CONFIGS[name])
        shapes = O.state_shapes(OC[name])           # pinned by strict load into the reference (tests/golden/gen_golden.py)
        sd = m.state_dict()
        assert set(sd) == set(shapes)
        assert all(tuple(v.shape) == tuple(shapes[k]) for k, v in sd.items())
        assert [n for n, p in m.named_parameters() if not p.requires_grad] == [n for n in O.FROZEN if n.startswith('encoder.')]
        m.load_state_dict(O.closed_form_state(OC[name], 0), strict=True)
    assert list(CONFIGS) == list(OC)
    for a, b in zip(CONFIGS.values(), OC.values()):
        assert (a.embed_dim, a.depth, a.fusion_tkns, a.audio_size) == (b.embed_dim, b.depth, b.fusion_tkns, b.audio_size)
        if hasattr(a, 'decoder_dim'):
            assert (a.decoder_dim, a.audio_mask_ratio, a.image_size) == (b.decoder_dim, b.audio_mask_ratio, b.image_size)
        else:
            assert (a.video_size, a.video_patch) == (b.video_size, b.video_patch)


def test_video_state_dict_contract_and_pos_table(golden):
    """VideoEarlyFusion (configs[4]): names/shapes as pinned by the strict load into the reference
    (tests/golden/gen_golden.py gen_video), the 3-D sin-cos table equals the reference's own initial buffer."""
    from deepavfusion_amd.build_model import build_video_earlyfusion
    from deepavfusion_amd.configs import CONFIGS
    from oracle import avmae_oracle as O
    from oracle.configs import CONFIGS as OC
    g = golden('e2e_video_micro')
    m = build_video_earlyfusion(CONFIGS['video_micro'])
    shapes = O.state_shapes(OC['video_micro'])
    sd = m.state_dict()
    assert set(sd) == set(shapes) and all(tuple(v.shape) == tuple(shapes[k]) for k, v in sd.items())
    assert set(g['grad_names'].tolist()) == {n for n, p in m.named_parameters() if p.requires_grad}
    assert [n for n, p in m.named_parameters() if not p.requires_grad] == ['video.pos_embed', 'audio.pos_embed']
    assert np.allclose(sd['video.pos_embed'].numpy(), g['video_pos_embed_init'], atol=1e-6)
    m.load_state_dict(O.closed_form_state(OC['video_micro'], 0), strict=True)
    ids = m.params_layer_ids()
    # as in the reference, the frozen pos-embeds are not listed (models/video_vits.py:185-192)
    assert len([p for p, _ in ids if p is not None]) == len([p for p in m.parameters() if p.requires_grad])
    assert max(l for _, l in ids) == OC['video_micro'].depth + 1


def test_sincos_tables(golden):
    from deepavfusion_amd.util.pos_embed import get_2d_sincos_pos_embed
    g = golden('posembed')
    n = 0
    for k in g.files:
        if not k.startswith('2d_'):
            continue
        _, dim, grid = k.split('_')[:3]
        grid = tuple(int(x) for x in grid.split('x'))
        full = get_2d_sincos_pos_embed(int(dim), grid).astype(np.float32)
        if k.endswith('_sub'):
            assert np.allclose(full[::3, ::5], g[k], atol=1e-6)
        elif k.endswith('_sum'):
            assert abs(full.astype(np.float64).sum() - float(g[k])) < 1e-3
        else:
            assert np.allclose(full, g[k], atol=1e-6)
        n += 1
    assert n >= 10


class _NS(dict):
    __getattr__ = dict.__getitem__


def test_lr_schedule_and_param_groups(golden):
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    g = golden('lr_groups')
    model = build_avmae(CONFIGS['micro'])
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    name_of = {id(p): n for n, p in model.named_parameters()}
    assert len(groups) == int(g['n_groups'])
    for gi, grp in enumerate(groups):
        assert sorted(name_of[id(p)] for p in grp['params']) == sorted(g[f'group{gi}.names'].tolist())
        assert grp['weight_decay'] == float(g[f'group{gi}.weight_decay'])
        assert bool(grp.get('pretrained', False)) == bool(g[f'group{gi}.pretrained'])
    opt = torch.optim.AdamW(groups, lr=1e-3, betas=(0.9, 0.95))
    args = _NS(opt=_NS(lr=1e-3, warmup_epochs=2, epochs=10, pt_warmup_epochs='10/2', pt_lr_mult_start=0, pt_lr_mult_end=1))
    for e, row in zip(g['epochs'], g['lr_table']):
        lr = lr_sched.adjust_learning_rate(opt, float(e), args)
        assert abs(lr - row[0]) < 1e-12
        for gi, pg in enumerate(opt.param_groups):
            assert abs(pg['lr'] - row[1 + gi]) < 1e-12


def test_flop_model_matches_survey_table():
    import bench
    from deepavfusion_amd.configs import CONFIGS
    gf = {k: 3 * bench.necessary_fwd_flops_per_pair(CONFIGS[k]) / 1e9 for k in ('base', 'base_m75', 'base_as', 'large')}
    for k, ref in (('base', 176.7), ('base_m75', 187.2), ('base_as', 185.9), ('large', 377.8)):
        assert abs(gf[k] - ref) / ref < 0.01, (k, gf[k], ref)


def test_len_keep_truncation_quirk():
    assert int(320 * (1 - 0.8)) == 63 and int(196 * (1 - 0.75)) == 49 and int(320 * (1 - 0.75)) == 80


def test_bucket_layout_follows_backward_order():
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util.distributed import GradReducer
    from deepavfusion_amd.util.flat import FlatParams
    model = build_avmae(CONFIGS['micro'])
    flat = FlatParams(reversed(list(model.parameters())))
    names = {id(p): n for n, p in model.named_parameters()}
    assert names[id(flat.params[0])].startswith('image_decoder_pred')        # first gradient the backward finishes
    order = [names[id(p)] for p in flat.params]
    assert order.index('encoder.image.patch_embed.proj.weight') > order.index('encoder.image.blocks.0.norm1.weight') > order.index('encoder.image.blocks.1.norm1.weight') > order.index('image_decoder_embed.weight')
    assert all(p.grad.data_ptr() == flat.flat_g.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    red = GradReducer(flat, bucket_mb=1.0, first_bucket_mb=0.25)
    assert sum(len(b[2]) for b in red.buckets) == len(flat.params) and red.buckets[-1][1] == flat.total
    assert all(b[0] % 64 == 0 for b in red.buckets)


def test_init_distributed_mode_single_process_on_multi_gpu_host(monkeypatch):
    """``python train.py`` without a launcher on an 8-GPU host must NOT open a process group nobody else joins
    (reference launcher.py spawns the ranks first; util/distributed.py:66-100 only then rendezvous) — and must still seed."""
    import random
    import torch
    import torch.distributed as dist
    from deepavfusion_amd.util import distributed as du

    class NS(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    called = []
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: called.append(('set_device', d)))
    monkeypatch.setattr(dist, 'init_process_group', lambda *a, **k: called.append(('init', k)))
    monkeypatch.setattr(dist, 'barrier', lambda *a, **k: called.append(('barrier',)))
    monkeypatch.setattr(du, 'setup_for_distributed', lambda *a, **k: None)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    args = NS(env=NS(world_size=1, rank=0, dist_url='tcp://127.0.0.1:50000', seed=3, ngpu=1))
    du.init_distributed_mode(0, args)
    assert args.env.distributed is False and args.env.world_size == 1 and ('set_device', 0) in called
    assert not any(c[0] == 'init' for c in called)
    a = (random.random(), float(torch.rand(())))
    du.init_distributed_mode(0, args)
    assert a == (random.random(), float(torch.rand(())))                     # seeded on the single-process path too
    # spawned by train.py (env.ngpu = 8): ranks exist -> rendezvous over dist_url with world 8, per-rank seed
    called.clear()
    args = NS(env=NS(world_size=1, rank=0, dist_url='tcp://127.0.0.1:50000', seed=3, ngpu=8, spawned=True))
    du.init_distributed_mode(5, args)
    init = [c for c in called if c[0] == 'init'][0][1]
    assert args.env.distributed and init['world_size'] == 8 and init['rank'] == 5 and init['backend'] == 'nccl'
    # torch.distributed.run: env:// with the launcher's ranks
    called.clear()
    monkeypatch.setenv('WORLD_SIZE', '4'); monkeypatch.setenv('RANK', '2'); monkeypatch.setenv('LOCAL_RANK', '2')
    args = NS(env=NS(world_size=1, rank=0, dist_url='tcp://127.0.0.1:50000', seed=None, ngpu=4))
    du.init_distributed_mode(0, args)
    init = [c for c in called if c[0] == 'init'][0][1]
    assert init['init_method'] == 'env://' and init['world_size'] == 4 and init['rank'] == 2 and ('set_device', 2) in called


def test_nt_tuning_table_loads_and_rejects_malformed_blobs():
    """The shipped tile-configuration table installs through dav_nt_tune_set (host-only call), DAV_NT_TUNE=0 leaves the
    built-in rules alone, and blobs that are cut short or name an unknown configuration are refused."""
    import ctypes as C
    from deepavfusion_amd import _lib
    lib = _lib.load()
    n = _lib.load_nt_tuning(_lib.NT_TUNING_PATH)
    assert n >= 1
    ok = (C.c_int * 7)(3, 0, 1, 256, 256, 64, 4)
    assert lib.dav_nt_tune_set(ok, 7) == 1
    assert lib.dav_nt_tune_set(ok, 6) < 0                       # truncated entry
    bad = (C.c_int * 7)(99, 0, 1, 256, 256, 64, 4)              # unknown tile configuration
    assert lib.dav_nt_tune_set(bad, 7) < 0
    assert lib.dav_nt_tune_set(None, 0) == 0                    # cleared
    os.environ['DAV_NT_TUNE'] = '0'
    try:
        assert _lib.load_nt_tuning(_lib.NT_TUNING_PATH) == 0
    finally:
        del os.environ['DAV_NT_TUNE']
    assert _lib.load_nt_tuning(_lib.NT_TUNING_PATH) == n        # back to the shipped table for the rest of the session


def _zero_args(argtypes):
    import ctypes as C
    return [None if t is C.c_void_p else t(0) for t in argtypes]


def test_every_entry_point_rejects_empty_input_with_the_documented_code():
    """Error contract of include/dav_kernels.h: argument validation comes before any HIP call and answers with a negative
    code (never abort / never undefined behaviour).  Empty input — every size 0, every pointer NULL — is a bad shape for each
    kernel entry point; the call returns before a launch, so this runs without a GPU."""
    from deepavfusion_amd import _lib
    lib = _lib.load()
    host_only = {'dav_abi_version', 'dav_build_flags', 'dav_last_error_string', 'dav_tune', 'dav_nt_issue_log', 'dav_nt_tune_set', 'dav_batch_begin',
                 'dav_batch_lane', 'dav_batch_region', 'dav_batch_skip', 'dav_batch_suspend', 'dav_batch_end', 'dav_batch_abort',
                 'dav_batch_stats', 'dav_layernorm_bwd_workspace_bytes', 'dav_l2norm_workspace_bytes',
                 'dav_gemm_tn_gang_workspace_bytes'}
    kernels = sorted(set(_lib.SIGNATURES) - host_only)
    assert len(kernels) >= 40
    for name in kernels:
        rc = getattr(lib, name)(*_zero_args(_lib.SIGNATURES[name]))
        assert rc == -1, f'{name}(empty input) returned {rc}, expected DAV_ERR_SHAPE (-1)'
    # the launch batcher refuses calls outside a batch the same way
    for name in ('dav_batch_lane', 'dav_batch_end'):
        assert getattr(lib, name)() == -1
    assert lib.dav_batch_region(1) == -1 and lib.dav_batch_skip(1) == -1
    assert lib.dav_tune(99, 0) == -1


def test_error_codes_for_dtype_alignment_and_workspace():
    """-2 unsupported dtype combination, -3 workspace too small, -5 misaligned pointer / stride: each from a call whose sizes
    are valid, whose pointers are never dereferenced (validation returns first) — the conventions block of include/dav_kernels.h."""
    import ctypes as C
    from deepavfusion_amd import _lib
    lib = _lib.load()
    p = lambda v: C.c_void_p(v)
    S = _lib.SIGNATURES

    def call(name, **over):
        args = _zero_args(S[name])
        for k, v in over.items():
            args[int(k[1:])] = v
        return getattr(lib, name)(*args)
    # dav_gemm_nt_bf16(A, B, M, N, K, lda, ldb, ..., C @16, ldc, c_is_bf16 @18, ..., beta @23): accumulate into a bf16 output
    nt = dict(a0=p(4096), a1=p(8192), a2=C.c_int(128), a3=C.c_int(128), a4=C.c_int(64), a5=C.c_int(64), a6=C.c_int(64),
              a16=p(12288), a17=C.c_int(128))
    assert call('dav_gemm_nt_bf16', **nt, a18=C.c_int(1), a23=C.c_int(1)) == -2
    assert call('dav_gemm_nt_bf16', **{**nt, 'a0': p(4096 + 8)}) == -5                    # A not 16-byte aligned
    assert call('dav_gemm_nt_bf16', **{**nt, 'a5': C.c_int(60)}) == -1                    # lda not a multiple of 8
    # dav_gemm_tn_bf16(A, B, Mc, N, K, lda, ldb, ...)
    assert call('dav_gemm_tn_bf16', a0=p(4096 + 2), a1=p(8192), a2=C.c_int(64), a3=C.c_int(8), a4=C.c_int(8), a5=C.c_int(8), a6=C.c_int(8)) == -5
    # dav_l2norm(x, n, scale, out, workspace, workspace_bytes, stream)
    assert call('dav_l2norm', a0=p(4096), a1=C.c_long(1024), a3=p(8192), a4=p(12288), a5=C.c_size_t(16)) == -3
    assert lib.dav_l2norm_workspace_bytes(1024) >= 4096
    # dav_adamw_flat(p, g, m, v, p_bf16, n, seg_end, hyper, nseg, ...)
    assert call('dav_adamw_flat', a0=p(4096 + 4), a1=p(8192), a2=p(12288), a3=p(16384), a5=C.c_long(1024), a8=C.c_int(1)) == -5
    assert call('dav_adamw_flat', a0=p(4096), a1=p(8192), a2=p(12288), a3=p(16384), a5=C.c_long(1022), a8=C.c_int(1)) == -1   # n % 4
    # dav_layernorm_bwd: workspace given but too small for (rows, D)
    need = lib.dav_layernorm_bwd_workspace_bytes(4 * 7, 64)
    assert need > 0
    ln = S['dav_layernorm_bwd']
    args = _zero_args(ln)
    args[2], args[6], args[7] = C.c_int(7), C.c_int(4), C.c_int(64)            # r0, B, D
    args[8] = p(4096)                                                          # dy_bf16
    args[-3], args[-2] = p(8192), C.c_size_t(need - 1)                         # workspace, workspace_bytes
    assert lib.dav_layernorm_bwd(*args) == -3
    # dav_gemm_tn_gang_bf16(problems, count, workspace, workspace_bytes, stream): the workspace query prices header + problem table +
    # one ticket per 256 x 256 tile; a smaller workspace, a misaligned one and a problem asking for the fused optimizer pass are refused
    pr = (_lib.DavTnProblem * 2)()
    for q, (N, K) in zip(pr, ((768, 3072), (192, 264))):
        q.A, q.B, q.C = 4096, 8192, 12288
        q.Mc, q.N, q.K, q.lda, q.ldb, q.ldc = 3136, N, K, N, K, K
    need = lib.dav_gemm_tn_gang_workspace_bytes(pr, 2)
    assert need == 128 + 2 * 96 + 8 * (3 * 12 + 1 * 2)
    assert lib.dav_gemm_tn_gang_bf16(pr, 2, p(1 << 20), C.c_size_t(need - 1), None) == -3
    assert lib.dav_gemm_tn_gang_bf16(pr, 2, p((1 << 20) + 4), C.c_size_t(need), None) == -5
    pr[1].flags = 3
    assert lib.dav_gemm_tn_gang_workspace_bytes(pr, 2) == 0 and lib.dav_gemm_tn_gang_bf16(pr, 2, p(1 << 20), C.c_size_t(need), None) == -1
    pr[1].flags, pr[1].Mc = 0, 100                                       # a ragged contraction (not a multiple of 64 rows) is fine here ...
    assert lib.dav_gemm_tn_gang_workspace_bytes(pr, 2) == need
    assert lib.dav_gemm_tn_grouped_bf16(pr, 2, None) == -1                # ... and refused by the 128 x 128 grouped kernel
    pr[1].Mc = 0
    assert lib.dav_gemm_tn_gang_workspace_bytes(pr, 2) == 0
    assert _lib.ERRORS[-2] and _lib.ERRORS[-3] and _lib.ERRORS[-5]
    # ---- ABI 8: LayerNorm folded into the GEMMs (dav_gemm_nt_ln_bf16 + DavNtLn, dav_rowstats_cast, dav_layernorm_bwd_twin, dav_ln_fold_grouped)
    ln = _lib.DavNtLn()
    ntl = dict(nt, a26=C.byref(ln))
    ln.stats, ln.ln_c, ln.eps = 16384, 20480, 1e-6
    assert call('dav_gemm_nt_ln_bf16', **{**ntl, 'a4': C.c_int(2048), 'a5': C.c_int(2048), 'a6': C.c_int(2048)}) == -1      # K > 1024: the statistics of a row do not fit the prologue
    ln.a_r0, ln.a_r1 = 3, 5                                                  # two-source rows without the second source
    assert call('dav_gemm_nt_ln_bf16', **ntl) == -1
    ln = _lib.DavNtLn(); ntl = dict(nt, a26=C.byref(ln))
    ln.stats_out, ln.twin_out, ln.ld_twin = 16384, 20480 + 8, 128            # twin not 16-byte aligned
    assert call('dav_gemm_nt_ln_bf16', **ntl) == -5
    ln.twin_out = 20480
    assert call('dav_gemm_nt_ln_bf16', **{**ntl, 'a18': C.c_int(1)}) == -1   # statistics are those of an fp32 result
    assert call('dav_gemm_nt_ln_bf16', **{**ntl, 'a25': C.c_int(60 << 4)}) == -1      # producers: the 32 x 64 / 32 x 32 wave tiles only
    assert call('dav_rowstats_cast', a0=p(4096), a2=C.c_int(2), a3=C.c_int(3), a4=C.c_int(96), a5=p(8192), a6=p(12288)) == -1      # D % 64
    assert call('dav_rowstats_cast', a0=p(4096), a2=C.c_int(2), a3=C.c_int(3), a4=C.c_int(128), a5=p(8192 + 2), a6=p(12288)) == -5
    tw = S['dav_layernorm_bwd_twin']
    args = _zero_args(tw)
    args[0], args[1], args[2], args[3] = p(4096), C.c_long(7 * 64), p(8192), C.c_int(7)      # xb0, batch stride, st0, r0
    args[8], args[9], args[10] = C.c_int(4), C.c_int(64), C.c_float(1e-6)                    # B, D, eps
    args[11], args[13] = p(12288), p(16384)                                                  # dy_bf16, gamma
    args[-3], args[-2] = p(20480), C.c_size_t(lib.dav_layernorm_bwd_workspace_bytes(4 * 7, 64) - 1)
    assert lib.dav_layernorm_bwd_twin(*args) == -3
    fold = (_lib.DavLnFold * 1)()
    fold[0].w, fold[0].gamma, fold[0].beta, fold[0].w_ln_bf16, fold[0].ln_c, fold[0].ln_d, fold[0].N, fold[0].K = 4096, 8192, 12288, 16384, 20480, 24576, 8, 60
    assert lib.dav_ln_fold_grouped(fold, 1, None) == -1                      # K % 8
    fold[0].K, fold[0].w = 64, 4096 + 4
    assert lib.dav_ln_fold_grouped(fold, 1, None) == -5
    assert lib.dav_ln_fold_grouped(fold, 49, None) == -1                     # more pairs than one launch carries
    # ---- ABI 9: dropout (dav_attn_drop_fwd / _bwd + fp32 twins, dav_dropout_rows)
    for fn in ('dav_attn_drop_fwd', 'dav_attn_drop_fwd_f32'):
        a = _zero_args(S[fn])
        for i, v in zip(range(5), (4096, 8192, 12288, 16384, 20480)):
            a[i] = p(v)
        for i, v in zip(range(5, 11), (2, 3, 40, 70, 64, 64)):            # B, H, Nq, Nk, dqk, dv
            a[i] = C.c_int(v)
        for i, (bs, rs) in zip(range(11, 19, 2), ((70 * 192, 192),) * 3 + ((40 * 192, 192),)):
            a[i], a[i + 1] = C.c_long(bs), C.c_int(rs)
        a[19] = C.c_float(0.125)
        a[20], a[21], a[22] = None, C.c_int(96), C.c_float(1.25)
        assert getattr(lib, fn)(*a) == -1                                    # no mask: the plain entry points are the ones to call
        a[20], a[21] = p(24576), C.c_int(64)                                 # keep_ld < Nk (rounded up to 32 for the bf16 kernels)
        assert getattr(lib, fn)(*a) == -1
        a[21], a[22] = C.c_int(96), C.c_float(0.0)                           # keep_scale = 1 / (1 - p) > 0
        assert getattr(lib, fn)(*a) == -1
        if fn == 'dav_attn_drop_fwd':
            a[20], a[22] = p(24576 + 2), C.c_float(1.25)                     # the bf16 kernels read mask rows four bytes at a time
            assert getattr(lib, fn)(*a) == -5
    dr = _zero_args(S['dav_dropout_rows'])
    dr[0], dr[3], dr[4], dr[9] = p(4096), p(8192), C.c_float(1.25), p(12288)
    dr[6], dr[7], dr[8] = C.c_int(2), C.c_int(3), C.c_int(10)               # D % 4
    assert lib.dav_dropout_rows(*dr) == -1
    dr[8], dr[4] = C.c_int(16), C.c_float(0.0)
    assert lib.dav_dropout_rows(*dr) == -1                                   # a mask needs its scale
    dr[4], dr[3] = C.c_float(1.25), p(8192 + 1)
    assert lib.dav_dropout_rows(*dr) == -5


def test_written_first_contribution_bookkeeping(monkeypatch):
    """engine.flush_wgrads under wgrad_overwrite_begin(): which queued weight-gradient problems may WRITE their tile
    (DavTnProblem.flags bit 0) — pure host logic, the grouped launch is intercepted."""
    from deepavfusion_amd import engine as E, ops
    launches = []
    monkeypatch.setattr(ops, 'gemm_tn_grouped', lambda probs: launches.append([dict(p) for p in probs]))
    lin = [torch.nn.Linear(8, 8) for _ in range(5)]
    g = [torch.zeros(8, 8) for _ in range(5)]

    def prob(i, mc, full=True, col=0):
        C = g[i] if col == 0 else g[i].view(-1)[col:]
        return dict(A=None, B=None, Mc=mc, N=8, K=8 if full else 4, C=C, lda=8, ldb=8, ldc=8, bias_grad=None, ready=(),
                    gbase=g[i].data_ptr(), weight=lin[i].weight if full else None)
    with E.deferred_wgrads():
        E.wgrad_overwrite_begin()
        E._OVERWRITE['touched'].add(g[3].data_ptr())               # weight 3 got an immediate (un-deferred) contribution earlier
        E._DEFERRED.extend([prob(0, 128), prob(1, 64), prob(1, 192),    # weight 1 twice (decoder_embed: tokens and fusion tokens)
                            prob(2, 64, full=False), prob(2, 64, full=False, col=4),      # weight 2 as two column blocks (pair path)
                            prob(3, 64), prob(4, 256)])
        E.flush_wgrads()
        E._DEFERRED.append(prob(4, 64))                            # a later flush into weight 4: accumulates
        E.flush_wgrads()
        kept = E.wgrad_overwrite_end()
    assert E._OVERWRITE is None
    flat = [(p['C'].data_ptr(), p['Mc'], bool(p.get('overwrite'))) for l in launches for p in l]
    assert len(launches) == 3 and [len(l) for l in launches] == [6, 1, 1]          # the second problem of weight 1 cannot share a launch
    assert [p['Mc'] for p in launches[0]] == sorted((p['Mc'] for p in launches[0]), reverse=True)   # longest contraction first
    ow = {(ptr, mc): o for ptr, mc, o in flat}
    assert ow[(g[0].data_ptr(), 128)] and ow[(g[4].data_ptr(), 256)]              # first full-weight contribution: written
    assert ow[(g[1].data_ptr(), 64)] != ow[(g[1].data_ptr(), 192)]                 # exactly one of the two problems of weight 1 writes ...
    assert launches[1][0]['C'].data_ptr() == g[1].data_ptr() and not launches[1][0].get('overwrite')   # ... the one issued first
    assert not any(o for ptr, mc, o in flat if ptr in (g[2].data_ptr(), g[2].view(-1)[4:].data_ptr()))   # column blocks accumulate
    assert not ow[(g[3].data_ptr(), 64)]                                           # touched before: accumulates
    assert not ow[(g[4].data_ptr(), 64)]                                           # second flush into a written weight: accumulates
    assert {id(p) for p in kept} == {id(lin[0].weight), id(lin[1].weight), id(lin[4].weight)}
    # outside a captured step nothing is ever written
    launches.clear()
    with E.deferred_wgrads():
        E._DEFERRED.append(prob(0, 64))
    assert launches and not launches[0][0].get('overwrite')


def test_bench_starts_its_own_ranks_when_launched_plainly():
    """`python bench.py --gpus N` without torch.distributed.run around it (missing #1): the process starts N
    fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, before any GPU call, and forwards rank 0's one line."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, DAV_BENCH_SPAWN_DRY='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '3'], env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['dry'] and d['world'] == 4 and d['rank'] == 0 and d['master'] == '127.0.0.1' and d['port'] > 0 and d['argv_gpus'] == 4


def _asm_first_readers(asm_text):
    """-> (checked, offenders): inline-asm VALU instructions that read a register of the most recent (program order) MFMA result
    with NO compiler-visible VALU reader of that same result earlier IN THE SAME BASIC BLOCK.  The compiler pads ITS OWN first
    reader (s_nop) and looks through branches when it does; an asm statement gets nothing (CDNA4 guide section 5.7 item 2)."""
    def regs(tok):
        tok = tok.strip().rstrip(',')
        m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.fullmatch(r'v(\d+)', tok)
        return {int(m.group(1))} if m else set()
    checked, offenders = 0, []
    writers = []            # [registers of an MFMA result, seen-a-compiler-reader flag], program order within a kernel
    in_asm = False
    for line in asm_text.splitlines():
        t = line.strip()
        if t.startswith('.amdhsa_kernel'):
            writers = []
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if t.endswith(':') or t.startswith('s_cbranch') or t.startswith('s_branch'):
            for w in writers:          # a new basic block: a reader on another path protects nothing here
                w[1] = False
            continue
        if not t or t[0] in '.;':
            continue
        op, _, rest = t.partition(' ')
        toks = rest.split(';')[0].split(',')
        if op.startswith('v_mfma'):
            d = regs(toks[0])
            writers = [w for w in writers if not (w[0] & d)] + [[d, False]]
            writers = writers[-64:]
            continue
        if not op.startswith('v_'):
            continue
        src = set().union(*[regs(x) for x in toks[1:]]) if len(toks) > 1 else set()
        dst = regs(toks[0])
        for w in writers:
            if w[0] & src:
                if in_asm:
                    checked += 1
                    if not w[1]:
                        offenders.append(t)
                else:
                    w[1] = True
        writers = [w for w in writers if not (w[0] & dst and not (w[0] & src))] if dst else writers
    return checked, offenders


def test_isa_audit_no_inline_asm_is_the_first_reader_of_an_mfma_result():
    """Compile csrc/attention.hip to gfx950 assembly and require that no inline-asm VALU instruction is the first reader of an
    MFMA result (advisor finding, round 3: the raw v_max3 chain of the softmax used to be; its head is now a plain
    __builtin_fmaxf over one register of each score MFMA, which hipcc pads).  The checker is validated on two synthetic
    snippets: the hazardous form must be flagged, the protected form must pass."""
    bad = """
	.amdhsa_kernel k
	v_mfma_f32_16x16x32_bf16 v[2:5], v[2:5], v[6:9], 0
	s_cbranch_scc1 .L1
	;;#ASMSTART
	v_max3_f32 v1, v2, v3, v4
	;;#ASMEND
"""
    good = """
	.amdhsa_kernel k
	v_mfma_f32_16x16x32_bf16 v[2:5], v[2:5], v[6:9], 0
	s_nop 7
	v_max_f32_e32 v1, v2, v6
	;;#ASMSTART
	v_max3_f32 v1, v1, v3, v4
	;;#ASMEND
"""
    assert _asm_first_readers(bad) == (1, ['v_max3_f32 v1, v2, v3, v4'])
    assert _asm_first_readers(good) == (1, [])
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    csrc = os.path.join(ROOT, 'deepavfusion_amd', 'csrc')
    asm = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-honor-nans', f'-I{ROOT}/include', f'-I{csrc}', '-S',
                          '--cuda-device-only', '-o', '-', os.path.join(csrc, 'attention.hip')], capture_output=True, text=True, timeout=300)
    assert asm.returncode == 0, asm.stderr[-2000:]
    checked, offenders = _asm_first_readers(asm.stdout)
    assert checked > 0 and not offenders, offenders[:6]


def test_switch_table_is_complete_and_current():
    """SWITCHES.md (tools/gen_switch_table.py): every DAV_* environment switch the sources read has a documented default, meaning
    and covering test, and the committed table is the generated one."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gen_switch_table.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]


def test_profiled_traffic_constants_belong_to_this_tree():
    """bench.py's roofline.traffic and step_fabric are PMC-derived constants read from profiles/*.json; each record carries the hash of
    the kernel sources + tuned table it was measured on (deepavfusion_amd._lib.kernel_source_hash) and bench.py drops a stale one.
    The committed records of the bench workload must be current, or the driver's line would carry nulls."""
    import json
    from deepavfusion_amd._lib import kernel_source_hash
    h = kernel_source_hash()
    assert h == kernel_source_hash() and len(h) == 16
    for name in ('dominant_kernel_traffic.json', 'step_traffic.json', 'wgrad_traffic.json'):
        rec = json.load(open(os.path.join(ROOT, 'profiles', name)))['base_b64']
        assert rec.get('kernel_source_hash') == h, (f'profiles/{name} was measured on another version of the kernels / tuned table: '
                                                     're-run tools/collect_r06.sh quick on the GPU box and tools/refresh_profiles_r06.sh')
        assert rec.get('measured_on'), f'profiles/{name}: no record of the box / date the PMC passes ran on'
