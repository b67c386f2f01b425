"""CPU restatement of the reference's audio front-end — TEST INFRASTRUCTURE ONLY (never imported by deepavfusion_amd).

    aT.Pad(rate, dur) -> [aT.RandomVol()] -> aT.MelSpectrogram(sample_rate, n_fft=int(rate*0.05), hop_length=int(rate/64),
    n_mels) -> aT.Log() -> [:, :, :-1]           reference train.py:50-54, util/audio_transforms.py:8-35, datasets.py:242

`aT.MelSpectrogram` is torchaudio.transforms.MelSpectrogram with its defaults.  torchaudio (pinned by the reference's
environment) is NOT installed here and not vendored by the reference, so its published algorithm is restated:
Spectrogram = torch.stft(n_fft, hop, win_length=n_fft, window=hann_window(n_fft, periodic=True), center=True,
pad_mode='reflect', normalized=False, onesided=True) -> |.|^2;  MelScale = melscale_fbanks(n_freqs, f_min=0, f_max=rate/2,
n_mels, rate, norm=None, mel_scale='htk').  The STFT half is pinned to torch's own torch.stft (imported below and checked
against a direct DFT in tests/test_oracle_golden.py); the filterbank half is "parity unpinned" (no torchaudio, no fixture in
the reference)."""
import math

import numpy as np
import torch


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels] (float64)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(hz_to_mel_htk(f_min), hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def pad(waveform, dur, rate):
    """aT.Pad: mirror-extend [C, S] until it has int(dur * rate) samples, then cut."""
    samples = int(dur * rate)
    while waveform.shape[-1] < samples:
        waveform = torch.cat((waveform, torch.flip(waveform, dims=(1,))), dim=1)
    return waveform[:, :samples]


def log_mel(waveform, sample_rate=16000, n_mels=128, eps=1e-7, drop_last=True, dtype=torch.float64):
    """waveform [B, S] -> [B, 1, n_mels, S // hop (+1 without drop_last)]."""
    n_fft, hop = int(sample_rate * 0.05), int(sample_rate / 64)
    x = waveform.to(dtype)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft, periodic=True, dtype=dtype),
                      center=True, pad_mode='reflect', normalized=False, onesided=True, return_complex=True).abs() ** 2     # [B, F, T]
    fb = torch.from_numpy(melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate)).to(dtype)
    mel = torch.matmul(spec.transpose(1, 2), fb).transpose(1, 2)                                                           # [B, M, T]
    out = torch.log10(mel + eps)
    if drop_last:
        out = out[:, :, :-1]
    return out.unsqueeze(1)


def dft_power_direct(frame_windowed):
    """|DFT|^2 of one windowed frame by the defining sum (pins torch.stft's conventions in the tests)."""
    n = len(frame_windowed)
    k = np.arange(n // 2 + 1)[:, None] * np.arange(n)[None, :]
    ang = 2.0 * math.pi * (k % n) / n
    re = (frame_windowed[None, :] * np.cos(ang)).sum(1)
    im = (frame_windowed[None, :] * np.sin(ang)).sum(1)
    return re * re + im * im
