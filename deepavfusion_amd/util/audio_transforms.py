"""Audio transforms of the reference (util/audio_transforms.py:8-35 + the torchaudio MelSpectrogram it composes,
train.py:50-54) with the spectrogram computed ON THE GPU from raw waveforms instead of in CPU
data-loader workers.  Same class names and constructor arguments:

    aT.Compose([aT.Pad(rate=…, dur=…), aT.RandomVol(), aT.MelSpectrogram(sample_rate=…, n_fft=…, hop_length=…, n_mels=…), aT.Log()])

works on a batch of waveforms [B, S] (or [B, 1, S] / [1, S]) resident on the device and yields [B, 1, n_mels, frames];
``LogMelSpectrogram`` is the fused form (one kernel, incl. the ``[:, :, :-1]`` of datasets.py:242).
Pad / RandomVol are index / scale plumbing (torch); the STFT, mel projection and log run in csrc/mel.hip."""
import math
import random

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream


class Compose(torch.nn.Module):
    def __init__(self, transforms):
        super().__init__()
        self.transforms = list(transforms)

    def forward(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class RandomVol(torch.nn.Module):
    """util/audio_transforms.py:8-17: a random gain in dB, then clamp to [-1, 1].  The reference applies this transform per
    SAMPLE inside the dataset: ONE draw per call, whatever the waveform's shape ([samples] or [channels, samples] — a stereo
    clip gets one gain for both channels).  ``per_sample=True`` (set by the GPU front-end of this package, which transforms a
    whole batch [B, samples] at once) draws one gain per row of the first dimension instead — what B per-sample calls would
    have drawn.  The mode is explicit: [channels, samples] and [B, samples] cannot be told apart by shape."""
    def __init__(self, gain=(-6, 6), per_sample=False):
        super().__init__()
        self.gain = gain
        self.per_sample = per_sample

    def forward(self, waveform):
        if self.per_sample and waveform.dim() >= 2:
            g = torch.tensor([random.uniform(self.gain[0], self.gain[1]) for _ in range(waveform.shape[0])],
                             dtype=waveform.dtype, device=waveform.device).view(-1, *([1] * (waveform.dim() - 1)))
            return torch.clamp(waveform * torch.pow(10.0, g / 20.0), -1, 1)
        g = random.uniform(self.gain[0], self.gain[1])
        return torch.clamp(waveform * (10.0 ** (g / 20.0)), -1, 1)


class Pad(torch.nn.Module):
    """util/audio_transforms.py:19-27 (note the reference's positional order: dur, rate)."""
    def __init__(self, dur, rate):
        super().__init__()
        self.samples = int(dur * rate)

    def forward(self, waveform):
        while waveform.shape[-1] < self.samples:
            waveform = torch.cat((waveform, torch.flip(waveform, dims=(-1,))), dim=-1)
        return waveform[..., :self.samples]


def _fbank_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    m_pts = np.linspace(mel(f_min), mel(f_max), n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))


class MelSpectrogram(torch.nn.Module):
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, hop_length=…, n_mels=…) with its defaults (win_length = n_fft,
    periodic Hann, power 2, centre / reflect padding, HTK mel scale, no filter normalisation) on dav_logmel."""

    def __init__(self, sample_rate=16000, n_fft=400, hop_length=None, n_mels=128, f_min=0.0, f_max=None, log_eps=None, drop_last=False):
        super().__init__()
        self.sample_rate, self.n_fft, self.hop, self.n_mels = sample_rate, n_fft, hop_length or n_fft // 2, n_mels
        self.log_eps, self.drop_last = log_eps, drop_last
        n = np.arange(n_fft, dtype=np.float64)
        fb = _fbank_htk(n_fft // 2 + 1, f_min, f_max if f_max is not None else sample_rate / 2.0, n_mels, sample_rate)
        nz = fb > 0
        lo = np.where(nz.any(0), nz.argmax(0), 0).astype(np.int32)
        hi = np.where(nz.any(0), fb.shape[0] - nz[::-1].argmax(0), 0).astype(np.int32)
        self.register_buffer('window', torch.from_numpy(0.5 - 0.5 * np.cos(2.0 * math.pi * n / n_fft)).float(), persistent=False)
        self.register_buffer('cos_tab', torch.from_numpy(np.cos(2.0 * math.pi * n / n_fft)).float(), persistent=False)
        self.register_buffer('sin_tab', torch.from_numpy(np.sin(2.0 * math.pi * n / n_fft)).float(), persistent=False)
        self.register_buffer('fbank', torch.from_numpy(fb).float().contiguous(), persistent=False)
        self.register_buffer('band_lo', torch.from_numpy(lo), persistent=False)
        self.register_buffer('band_hi', torch.from_numpy(hi), persistent=False)

    def forward(self, waveform):
        if not waveform.is_cuda:
            raise RuntimeError('the log-mel front-end runs on an MI355X (cuda) device; there is no CPU fallback')
        x = waveform.to(torch.float32)
        if x.dim() == 3:
            x = x.reshape(-1, x.shape[-1])
        x = x.contiguous()
        if self.window.device != x.device:
            self.to(x.device)
        B, S = x.shape
        frames = S // self.hop + 1 - (1 if self.drop_last else 0)
        out = torch.empty(B, 1, self.n_mels, frames, dtype=torch.float32, device=x.device)
        lib = _lib.load()
        _lib.check(lib.dav_logmel(_ptr(x), B, S, self.n_fft, self.hop, self.n_mels, _ptr(self.window), _ptr(self.cos_tab),
                                  _ptr(self.sin_tab), _ptr(self.fbank), _ptr(self.band_lo), _ptr(self.band_hi),
                                  float(self.log_eps if self.log_eps is not None else 0.0), int(self.log_eps is not None),
                                  int(self.drop_last), _ptr(out), _stream()), 'dav_logmel')
        return out


class Log(torch.nn.Module):
    """util/audio_transforms.py:29-35: log10(spec + eps)."""
    def __init__(self, eps=1e-7):
        super().__init__()
        self.eps = eps

    def forward(self, spec):
        x = spec.contiguous()
        y = torch.empty_like(x)
        _lib.check(_lib.load().dav_log10_eps(_ptr(x), float(self.eps), x.numel(), _ptr(y), _stream()), 'dav_log10_eps')
        return y


class LogMelSpectrogram(MelSpectrogram):
    """MelSpectrogram -> Log -> [:, :, :-1] (train.py:50-54, datasets.py:242) in ONE kernel: waveforms [B, S] -> [B, 1, n_mels, S // hop]."""
    def __init__(self, sample_rate=16000, n_mels=128, eps=1e-7):
        super().__init__(sample_rate=sample_rate, n_fft=int(sample_rate * 0.05), hop_length=int(sample_rate / 64), n_mels=n_mels,
                         log_eps=eps, drop_last=True)
