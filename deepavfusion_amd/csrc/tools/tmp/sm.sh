for i in 1 2; do
for w in 5 7; do
DAV_NT_SMALL=$w timeout 300 python bench.py --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/sm_$w.json 2> gpurun_out/sm_$w.err
python -c "
import json; d=json.load(open('gpurun_out/sm_$w.json')); print('small=$w', d['value'], d['ms_per_step'])"
done
done
