#!/usr/bin/env python3
"""How late does the second branch of a captured fork start?  One kernel on the capture stream, then two independent chains of N kernels on
two streams (as the two MAE decoders of the step), joined at the end; replayed under rocprofv3 --kernel-trace and analysed by
graph_fork_probe_report.py.  Variants (argv[1]): plain | join<k> (cross-joins after every k kernels) | ext (fork through an EXTERNAL event:
an event-record node in the graph)."""
import sys

import torch

variant = sys.argv[1] if len(sys.argv) > 1 else 'plain'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device('cuda')
a = torch.randn(4096, 1024, device=dev, dtype=torch.bfloat16)
wa = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
b = torch.randn(2048, 1024, device=dev, dtype=torch.bfloat16)
wb = torch.randn(1024, 1024, device=dev, dtype=torch.bfloat16)
x0 = torch.zeros(1 << 20, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
k = int(variant[4:]) if variant.startswith('join') else 0


def body():
    main = torch.cuda.current_stream()
    x0.add_(1.0)                                   # the fork node
    if variant == 'ext':
        ev = torch.cuda.Event(external=True)
        ev.record(main)
        s2.wait_event(ev)
    else:
        s2.wait_stream(main)
    ya, yb = a, b
    for i in range(N):
        ya = torch.tanh(ya @ wa) if i % 2 else ya * 1.0001          # chain A on the capture stream: GEMM, elementwise, ...
        with torch.cuda.stream(s2):
            yb = torch.tanh(yb @ wb) if i % 2 else yb * 1.0001      # chain B on the side stream
        if k and (i + 1) % k == 0 and i + 1 < N:
            main.wait_stream(s2)
            s2.wait_stream(main)
    main.wait_stream(s2)
    x0.add_(ya.float().sum() + yb.float().sum())  # the join node
    return ya, yb


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    torch.cuda.synchronize()
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
print('done', variant)
