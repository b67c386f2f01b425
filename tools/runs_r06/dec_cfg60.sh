#!/bin/bash
# Configuration 60 (256 x 256 tiles) for the DECODERS' GEMMs only (M = B x 228 / B x 352 rows): the decoders' two streams overlap little
# (profiles/r06_dec_overlap.txt), so a tile that is faster alone and owns the CU may pay there although it lost as a step-wide choice (round 5).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
DAV_DUMP_MIX=$GRAFT_REPO_ROOT/$OUT/mix.json timeout 300 python bench.py --no-graph --no-roofline --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $OUT/mix.err
for V in "$@"; do
python3 - $OUT/mix.json deepavfusion_amd/tuning/nt_gfx950.json $OUT/nt_dec60_$V.json "$V" <<'PY'
import json, sys
mix = json.load(open(sys.argv[1]))['nt']
table = json.load(open(sys.argv[2]))
which = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] else 'all'
key = lambda e: (e['b_kn'], tuple(sorted(map(tuple, e['problems']))))
held = {key(e): e for e in table['entries']}
n = 0
seen = set()
for cfg, bt, probs, flags in mix:
    if not all(M in (64 * 228, 64 * 352) for M, N, K in probs):
        continue
    if which == 'k2048' and not all(K == 2048 for M, N, K in probs):
        continue
    if which == 'wide' and not all(K == 2048 or N >= 1536 for M, N, K in probs):
        continue
    if which == 'audio' and not all(M == 64 * 352 for M, N, K in probs):
        continue
    if which == 'audiowide' and not all(M == 64 * 352 and (K == 2048 or N >= 1536) for M, N, K in probs):
        continue
    if which == 'fwd' and bt:
        continue
    if which == 'bwd' and not bt:
        continue
    e = {'cfg': 60, 'b_kn': bt, 'problems': [[M, N, K, fl] for (M, N, K), fl in zip(probs, flags)], 'rule_cfg': cfg}
    if key(e) not in seen:
        seen.add(key(e)); n += 1
    held[key(e)] = e
table['entries'] = list(held.values())
json.dump(table, open(sys.argv[3], 'w'), indent=1)
print(which, n, 'decoder launch kinds -> cfg 60')
PY
done
P=$GRAFT_REPO_ROOT/deepavfusion_amd/tuning/nt_gfx950.json
ARGS="DAV_NT_TUNE_FILE=$P"
for V in "$@"; do ARGS="$ARGS DAV_NT_TUNE_FILE=$GRAFT_REPO_ROOT/$OUT/nt_dec60_$V.json"; done
bash tools/ab_env2.sh $ARGS | sed -e "s#$GRAFT_REPO_ROOT/##" | cut -c1-110
