#!/usr/bin/env python3
"""Isolated timing of the four fused fusion-block tail kernels (csrc/fusion_tail.hip) at the bench shapes (B = 64: 32 workgroups),
each replayed 20 x inside a hipGraph: what a tail costs with the GPU to itself (in the step it runs beside the towers)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops      # noqa: E402

dev = 'cuda'
B, D, Da, nF = int(os.environ.get('B', '64')), 768, 192, 32
bf = lambda *s: (torch.randn(*s, device=dev) * 0.5).bfloat16()
f32 = lambda *s: torch.randn(*s, device=dev) * 0.5
wt = lambda o, i: (torch.randn(o, i, device=dev) * 0.03).bfloat16()
dims = dict(B=B, D=D, Da=Da, Hd=D, nmm=16, nv=8, na=8, eps2=1e-5)
T = dict(Wpv=wt(D, D), Wpa=wt(D, D), Wk=wt(Da, 2 * D), Wv=wt(D, 2 * D), Wp=wt(D, D), W1=wt(D, D), W2=wt(D, D),
         WpvT=wt(D, D), WpaT=wt(D, D), WkT=wt(2 * D, Da), WvT=wt(2 * D, D), WpT=wt(D, D), W1T=wt(D, D), W2T=wt(D, D),
         bpv=f32(D), bpa=f32(D), bk=f32(Da), bv=f32(D), bp=f32(D), b1=f32(D), b2=f32(D), g2=f32(D), be2=f32(D),
         xmm32=f32(B, nF, D), o_v=bf(B * 8, D), o_a=bf(B * 8, D), xvo_b=bf(B * 8, D), xao_b=bf(B * 8, D),
         kv_p=f32(B * 8, Da), ka_p=f32(B * 8, Da), vv_p=f32(B * 8, D), va_p=f32(B * 8, D), Kp=bf(B * 64, Da), Vp=bf(B * 64, D),
         xmm1=f32(B, nF, D), o2=bf(B * 16, D), h2=bf(B * nF, D), z=bf(B * nF, D), u=bf(B * nF, D), mean2=f32(B * nF), rstd2=f32(B * nF).abs() + 0.5,
         out=f32(B, nF, D), g=f32(B, nF, D), gb=bf(B * nF, D), dz=bf(B * nF, D), dh2=bf(B * nF, D), g1=f32(B, nF, D), g1b=bf(B * nF, D),
         do2=bf(B * 16, D), ln2_partial=f32(B // 2, 2 * D), dKp=bf(B * 64, Da), dVp=bf(B * 64, D), dkv_p=bf(B * 8, Da), dka_p=bf(B * 8, Da),
         dvv_p=bf(B * 8, D), dva_p=bf(B * 8, D), dxvo_b=bf(B * 8, D), dxao_b=bf(B * 8, D), dov=bf(B * 8, D), doa=bf(B * 8, D))
W_MB = {'tail1_fwd': 2 * D * D + Da * 2 * D + D * 2 * D, 'tail2_fwd': 3 * D * D, 'tail2_bwd': 3 * D * D, 'tail1_bwd': Da * 2 * D + D * 2 * D + 2 * D * D}
FL = {'tail1_fwd': 2 * 16 * (2 * D * D) + 2 * 16 * D * (2 * Da + 2 * D), 'tail2_fwd': 2 * D * D * (32 + 64 + 64), 'tail2_bwd': 2 * D * D * (64 + 64 + 32),
      'tail1_bwd': 2 * 16 * D * (2 * Da + 2 * D) + 2 * 16 * 2 * D * D}
for stage in ('tail1_fwd', 'tail2_fwd', 'tail2_bwd', 'tail1_bwd'):
    fn = lambda: ops.fusion_tail(stage, dims, **T)
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    mb = W_MB[stage] * 2 / 1e6
    print(f'{stage}: {us:7.1f} us  ({B // 2} workgroups; {mb:.1f} MB of weights per workgroup -> {mb * 1e6 / us / 1e3:.0f} GB/s per CU; '
          f'{FL[stage] * (B // 2) / us / 1e6:.1f} TFLOP/s)')
