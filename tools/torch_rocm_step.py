#!/usr/bin/env python3
"""Runs the stock-stack comparator (tests/torch_rocm_step.py: the oracle's torch restatement of the reference step on cuda,
bf16 autocast, hipBLASLt + SDPA) as a child process and stores its JSON line: profiles/r04_torch_rocm_step.json.
Measurement only — see the header of tests/torch_rocm_step.py.  Extra arguments are forwarded."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'torch_rocm_step.py')] + sys.argv[1:], stdout=subprocess.PIPE, check=True).stdout.decode()
line = [l for l in out.splitlines() if l.startswith('{')][-1]
dst = os.environ.get('DAV_STOCK_OUT', os.path.join(ROOT, 'gpurun_out', 'r04_torch_rocm_step.json'))
os.makedirs(os.path.dirname(dst), exist_ok=True)
with open(dst, 'a') as f:
    f.write(line + '\n')
print(line)
