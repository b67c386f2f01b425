#!/usr/bin/env python3
"""Time of the device-side log-mel front-end for one bench batch: 64 waveforms of 10 s at 16 kHz -> [64, 1, 128, 640]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd.util.audio_transforms import LogMelSpectrogram  # noqa: E402

fe = LogMelSpectrogram().cuda()
w = (torch.randn(64, 160000, device='cuda') * 0.1).clamp(-1, 1)
for _ in range(3):
    out = fe(w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = fe(w)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
flop = 64 * 641 * 401 * 800 * 4.0
print(f'log-mel of 64 x 10 s: {ms:.3f} ms per batch, {flop / ms / 1e9:.1f} TFLOP/s fp32 (direct DFT), out {tuple(out.shape)}')
