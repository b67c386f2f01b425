"""GPU: the log-mel front-end (csrc/mel.hip, deepavfusion_amd/util/audio_transforms.py) against the
oracle restatement of the reference's transform chain (train.py:50-54, util/audio_transforms.py, datasets.py:242).
Tolerance (fp32 DFT sums of 800 terms against the float64 oracle; the reference itself computes in fp32): in mel POWER,
|got - ref| <= 5e-5 * ref + 3e-6 * (largest mel power of that waveform) — a strong tone leaves fp32 rounding noise ~1e-7 of its
own power in every other band, which is a large RELATIVE error in a band that is 1e5 times quieter; in log10 units this is
<= 3e-5 wherever a band is within 40 dB of the loudest one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(got, ref, eps=1e-7):
    got, ref = got.double().cpu(), ref.double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    pg, pr = 10.0 ** got - eps, 10.0 ** ref - eps
    peak = pr.amax(dim=(1, 2, 3), keepdim=True)
    assert bool(((pg - pr).abs() <= 5e-5 * pr + 3e-6 * peak + 1e-9).all()), float(((pg - pr).abs() / (5e-5 * pr + 3e-6 * peak + 1e-9)).max())
    near = pr > 1e-4 * peak
    if bool(near.any()):
        assert float((got - ref)[near].abs().max()) <= 3e-5


def test_logmel_matches_oracle_and_reference_shapes():
    from deepavfusion_amd.util import audio_transforms as aT
    from oracle import audio_oracle as A
    torch.manual_seed(1)
    rate = 16000
    t = torch.arange(2 * rate) / rate
    waves = torch.stack([
        (0.3 * torch.randn(2 * rate)).clamp(-1, 1),                                        # noise
        0.5 * torch.sin(2 * np.pi * 440.0 * t) + 0.2 * torch.sin(2 * np.pi * 3000.0 * t),   # tones
        torch.zeros(2 * rate),                                                              # silence -> the eps floor
        (torch.rand(2 * rate) * 2 - 1) * torch.linspace(0, 1, 2 * rate),                    # ramped noise
    ])
    fused = aT.LogMelSpectrogram(sample_rate=rate, n_mels=128).cuda()
    out = fused(waves.cuda())
    assert out.shape == (4, 1, 128, 128)                                                    # 2 s -> 64 * 2 frames
    _check(out, A.log_mel(waves, rate, 128))
    # the reference's own composition, class by class (MelSpectrogram -> Log), then the dataset's [:, :, :-1]
    chain = aT.Compose([aT.MelSpectrogram(sample_rate=rate, n_fft=int(rate * 0.05), hop_length=int(rate / 64), n_mels=128), aT.Log()])
    out2 = chain(waves.cuda())[:, :, :, :-1]
    assert torch.allclose(out2, out, atol=1e-6)
    # 10 s of audio -> the (128, 640) log-mel the model is built for (train.py:65)
    long = (0.1 * torch.randn(2, 10 * rate)).clamp(-1, 1)
    o10 = fused(long.cuda())
    assert o10.shape == (2, 1, 128, 640)
    _check(o10, A.log_mel(long, rate, 128))


def test_pad_randomvol_and_other_geometries():
    from deepavfusion_amd.util import audio_transforms as aT
    from oracle import audio_oracle as A
    torch.manual_seed(2)
    short = (0.2 * torch.randn(1, 5000)).clamp(-1, 1)
    p = aT.Pad(dur=1.0, rate=16000)(short.cuda())
    assert torch.equal(p.cpu(), A.pad(short, 1.0, 16000))                                    # mirror extension, then cut
    v = aT.RandomVol()(short.cuda())
    assert float(v.abs().max()) <= 1.0
    ratio = (v.cpu() / short)[short.abs() > 1e-3]
    inside = ratio[(v.cpu().abs() < 1.0)[short.abs() > 1e-3]]
    assert float(inside.max() - inside.min()) < 1e-5 and 0.5 <= float(inside.mean()) <= 2.0      # one gain in [-6, 6] dB
    # a batch gets one gain PER SAMPLE (the reference applies the transform per sample inside its dataset)
    wb = (0.05 * torch.randn(6, 4000)).clamp(-1, 1)
    vb = aT.RandomVol(per_sample=True)(wb.cuda()).cpu()
    gains = [float((vb[i] / wb[i])[wb[i].abs() > 1e-3].median()) for i in range(6)]
    for i in range(6):
        r = (vb[i] / wb[i])[wb[i].abs() > 1e-3]
        assert float(r.max() - r.min()) < 1e-4 and 0.5 <= gains[i] <= 2.0
    assert max(gains) - min(gains) > 1e-3
    # default mode: ONE draw per call whatever the shape (a stereo [2, samples] clip gets one gain, as in the reference)
    st = (0.05 * torch.randn(2, 4000)).clamp(-1, 1)
    vs = aT.RandomVol()(st.cuda()).cpu()
    gs = [float((vs[i] / st[i])[st[i].abs() > 1e-3].median()) for i in range(2)]
    assert abs(gs[0] - gs[1]) < 1e-5
    # another rate / mel count (8 kHz, 64 mels): n_fft 400, hop 125
    w8 = (0.3 * torch.randn(3, 12000)).clamp(-1, 1)
    _check(aT.LogMelSpectrogram(sample_rate=8000, n_mels=64).cuda()(w8.cuda()), A.log_mel(w8, 8000, 64))
