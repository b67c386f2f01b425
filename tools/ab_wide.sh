for i in 1 2; do
for w in 0 1; do
DAV_NT_WIDE=$w timeout 300 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/ab_$w.json 2> gpurun_out/ab_$w.err
python -c "
import json; d=json.load(open('gpurun_out/ab_$w.json')); print('wide=$w', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
done
