#!/usr/bin/env python3
"""Generates SWITCHES.md: every DAV_* environment switch of the package / bench.py / train.py with its default, what it selects and
the test that covers it.  The descriptions live HERE; the list of switches is grepped from the sources, and the script fails when a
switch in the code has no entry (tests/test_cabi_and_host.py runs it in check mode, so the table cannot go stale).

    python tools/gen_switch_table.py            # rewrite SWITCHES.md
    python tools/gen_switch_table.py --check    # exit 1 if SWITCHES.md is out of date or a switch is undocumented
"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOT_SWITCHES = {'DAV_ABI_VERSION', 'DAV_OK', 'DAV_LAUNCH', 'DAV_LAUNCH_NOW', 'DAV_EXPERIMENTAL'} | {f'DAV_ERR_{x}' for x in ('ALIGN', 'DTYPE', 'HIP', 'SHAPE', 'WORKSPACE')}

# name: (default, class, what it selects, covered by)
#   class: schedule | kernel | dp | optimizer | debug (timing ablations: results are WRONG by construction) | harness
S = {
    'DAV_BATCH': ('auto', 'schedule', "launch batching policy (engine.BATCH_POLICY): unset = one HIP stream per independent chain with batched regions; 1 = the towers / decoders as lanes of one launch batch; 0 = no batching at all", 'test_lane_batched_and_stream_schedules_agree, test_full_size_step_is_schedule_independent_and_repeatable'),
    'DAV_LANE_MIN_ROWS': ('2^30', 'schedule', "with DAV_BATCH unset: lanes from this many rows (B x tokens per tower block) upwards", 'test_lane_batched_and_stream_schedules_agree (policy values)'),
    'DAV_FUSION_STREAM': ('1', 'schedule', "lanes schedule only: the fusion block on its own stream (1) or as a third lane of the batch (0)", 'bench.py roofline.lanes_schedule (runs it), test_full_size_step...'),
    'DAV_DEC_JOIN': ('0', 'schedule', "streams schedule: cross-join the two decoders' streams every n decoder blocks (0 = two unjoined branches).  Under rocprofv3 the second branch of the replayed graph starts 2-3 ms late and joins give the overlap back; without the profiler the branches start together and the step is flat (tools/runs_r06/dec_overlap_probe.py, profiles/r06_dec_overlap.txt): the traces under profiles/ understate the decoders' overlap", 'test_decoder_cross_joins_change_nothing_but_the_schedule'),
    'DAV_STREAMS': ('1', 'schedule', "0: everything on the current stream (serial schedule; tools/instep_gemm_bound.py uses it)", 'tools/instep_gemm_bound.py (profiles/r04_gemm_instep_bound.txt)'),
    'DAV_WGRAD_GANG': ('1', 'kernel', "weight gradients of a flush as ONE gang-scheduled launch of 256 x 256 tiles (dav_gemm_tn_gang_bf16: per-XCD ticket queues of tiles that share operand panels); 0 = the 128 x 128 grouped kernel, one launch per layer (profiles/r05_tn_gang_*.txt)", 'test_gang_weight_gradients_equal_the_grouped_kernel_and_the_oracle, test_baseline_config_shapes_vs_oracle (default path), tools/tn_gang_bench.py check()'),
    'DAV_WGRAD_MERGE': ('0 (all layers of a captured segment)', 'schedule', "encoder layers whose queued weight gradients share one launch (n > 0: a flush every n layers); with 0 an EAGER data-parallel backward (grad-ready hook installed, no captured segments) still flushes every 3 layers, so that gradient buckets become ready under the backward and operands are released (engine.WGRAD_EAGER_DP_MERGE)", 'test_gang_weight_gradients_equal_the_grouped_kernel_and_the_oracle (1 and 0)'),
    'DAV_TN_GANG_DEBUG': ('0', 'debug', "dav_gemm_tn_gang_bf16 timing ablations: 2 = no epilogue, 4 = no MFMAs, 8 / 16 = queue chosen by block id / by a deliberately wrong placement instead of the hardware XCC id", '- (tools/tn_gang_bench.py, profiles/r05_tn_gang_*.txt)'),
    'DAV_WGRAD_OVERWRITE': ('1', 'kernel', "captured step: the first weight-gradient contribution to a Linear weight WRITES its tile (AdamW skips that zero-fill); 0 = accumulate / zero-fill", 'test_written_first_gradients_equal_accumulated_ones'),
    'DAV_LN_FUSE': ('0', 'kernel', "1: LayerNorms folded into the GEMMs either side of them (dav_gemm_nt_ln_bf16: twin + row statistics from the producer's epilogue, gamma-folded weights in the consumer; no LayerNorm-forward launches, the backward re-makes the LayerNorm outputs for the weight gradients); auto: only when no backward follows; 0 (default): the LayerNorm kernels — the folded step is 0.7 ms slower, the folded forward alone 0.1 ms slower (profiles/r06_ln_fuse.txt)", 'test_ln_folded_and_layernorm_kernel_paths_agree, test_ln_folded_captured_step_tracks_the_kernel_path, test_kernel_family[ln_fused]'),
    'DAV_NT_ALT': ('0', 'kernel', "EXPERIMENTAL builds only: 31 / 51 = the software-pipelined / loader-wave body in place of configuration 3 (faster alone, +0.6 / +1.9 ms in the step: profiles/r05_experiments.txt)", '- (make EXPERIMENTAL=1)'),
    'DAV_NT_TUNE': ('1', 'kernel', "0: ignore the tuned tile-configuration table (deepavfusion_amd/tuning/nt_gfx950.json), rules only", 'test_nt_tuning_table_loads_and_rejects_malformed_blobs, test_baseline_config_shapes_vs_oracle[large-32] (entries must fire)'),
    'DAV_NT_TUNE_FILE': ('tuning/nt_gfx950.json', 'kernel', "another tuned table", 'test_nt_tuning_table_loads_and_rejects_malformed_blobs'),
    'DAV_SEGMENTS': ('0 (1 graph; 5 when data-parallel)', 'dp', "graphs per captured step (any run)", 'test_segmented_graph_step_matches_single_graph_and_schedules_every_bucket'),
    'DAV_DP_SEGMENTS': ('5', 'dp', "graphs per captured data-parallel step, between which finished gradient buckets are reduced", 'test_dp_switches_over_one_rank_rccl'),
    'DAV_DP_ALGO': ('allreduce', 'dp', "allreduce | rs_ag (reduce-scatter + all-gather per bucket)", 'test_dp_switches_world2_and_world4_gloo, test_dp_switches_over_one_rank_rccl'),
    'DAV_DP_BUCKET_MB': ('64', 'dp', "gradient bucket size (MB of fp32) when the caller passes none", 'test_dp_switches_world2_and_world4_gloo'),
    'DAV_DP_FIRST_BUCKET_MB': ('8', 'dp', "size of the first bucket (the decoders' last gradients start moving early)", 'test_dp_switches_world2_and_world4_gloo'),
    'DAV_DP_BF16': ('0', 'dp', "1 (NOT the reference's arithmetic): gradient buckets reduced as bf16 on the wire", 'test_dp_switches_world2_and_world4_gloo, test_dp_switches_over_one_rank_rccl'),
    'DAV_FORCE_DIST': ('0', 'harness', "1: build the data-parallel machinery on a 1-rank process group (tests / tools on one GPU)", 'test_dp_step_over_one_rank_rccl'),
    'DAV_DIST_TIMEOUT_S': ('1800', 'dp', "process-group timeout in seconds", '- (init_distributed_mode)'),
    'DAV_TUNE': ('', 'harness', "bench.py: comma list knob:value for dav_tune (launch-geometry experiments)", '- (bench only)'),
    'DAV_DUMP_MIX': ('', 'harness', "bench.py: file that receives the recorded launch mix of one step (tools/mix_sweep.py input)", '- (bench only)'),
    'DAV_BENCH_SPAWN_DRY': ('0', 'harness', "bench.py: CPU test hook of the rank spawner (no GPU call)", 'test_bench_starts_its_own_ranks_when_launched_plainly'),
    'DAV_NT_DEBUG': ('0', 'debug', "NT GEMM dead-code ablations (no stores / no epilogue / no MFMAs): timing only, results wrong", '- (DESIGN_HISTORY section 3)'),
    'DAV_TN_DEBUG': ('0', 'debug', "weight-gradient GEMM ablations: timing only", '- (DESIGN_HISTORY section 3)'),
    'DAV_ATTN_DEBUG': ('0', 'debug', "attention ablations (no tile loop / no staging): timing only", '- (DESIGN_HISTORY section 3)'),
}


def found():
    names = set()
    files = [os.path.join(ROOT, f) for f in ('bench.py', 'train.py')]
    for pat in ('deepavfusion_amd/*.py', 'deepavfusion_amd/util/*.py', 'deepavfusion_amd/models/*.py', 'deepavfusion_amd/csrc/*.hip', 'deepavfusion_amd/csrc/*.h'):
        files += glob.glob(os.path.join(ROOT, pat))
    for f in files:
        names |= set(re.findall(r'\bDAV_[A-Z0-9_]+\b', open(f, errors='replace').read()))
    return names - NOT_SWITCHES


def render():
    names = found()
    missing = sorted(names - set(S))
    if missing:
        raise SystemExit(f'switches in the sources without an entry in tools/gen_switch_table.py: {missing}')
    out = ['# Environment switches', '',
           'Generated by `tools/gen_switch_table.py` (a CPU test fails when this file is stale or a switch is undocumented).  Defaults are the',
           'product path; everything else exists for same-box A/B measurements, tests or the first multi-GPU runs.  "debug" switches are',
           'timing ablations whose results are wrong by construction.  Combinations other than the ones the named tests run are untested.', '']
    for cls, title in (('schedule', 'Step schedule'), ('kernel', 'Kernel selection'), ('optimizer', 'Optimizer'), ('dp', 'Data parallel'),
                       ('harness', 'Harness / tools'), ('debug', 'Timing ablations')):
        rows = sorted(n for n in names if S[n][1] == cls)
        if not rows:
            continue
        out += [f'## {title}', '', '| switch | default | selects | covered by |', '|---|---|---|---|']
        out += [f'| `{n}` | {S[n][0]} | {S[n][2]} | {S[n][3]} |' for n in rows]
        out.append('')
    stale = sorted(set(S) - names)
    if stale:
        out += ['(entries kept for switches that no longer appear in the sources: ' + ', '.join(stale) + ')', '']
    return '\n'.join(out)


if __name__ == '__main__':
    text = render()
    path = os.path.join(ROOT, 'SWITCHES.md')
    if '--check' in sys.argv:
        if not os.path.exists(path) or open(path).read() != text:
            raise SystemExit('SWITCHES.md is out of date: run python tools/gen_switch_table.py')
    else:
        open(path, 'w').write(text)
        print(f'{path}: {len(found())} switches')
