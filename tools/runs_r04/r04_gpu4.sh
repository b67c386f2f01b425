set -x
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04/prof_tails -o tails -- python3 $R/bench.py --steps 6 --warmup 2 --no-roofline --no-cpu-baseline > $R/gpurun_out/r04/prof_tails.json 2> $R/gpurun_out/r04/prof_tails.err
cd $R
ls gpurun_out/r04/prof_tails
python - <<'PY'
import sqlite3,glob
f=glob.glob('gpurun_out/r04/prof_tails/*.db')[0]
db=sqlite3.connect(f); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; sym=[t for t in tabs if 'info_kernel_symbol' in t][0]
rows=cur.execute(f"select k.kernel_name, d.start, d.end from {kd} d join {sym} k on d.kernel_id=k.id order by d.start").fetchall()
import collections
agg=collections.defaultdict(list)
for n,s,e in rows: agg[n.split('(')[0][:90]].append((e-s)/1e3)
tot=sum(sum(v) for v in agg.values())
for n,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:45]:
    print(f'{len(v):6d} {sum(v)/1e3:9.2f} ms  avg {sum(v)/len(v):8.1f} us  {n}')
PY
