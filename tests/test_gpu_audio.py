"""VGGish audio front-end on the MI355X (csrc/logmel.hip) against the oracle restatement and the reference's own
waveform_to_examples output (tests/golden/g6_logmel.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_py

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_logmel_matches_reference_fixture(avt, dev):
    from avtex.audio_frontend import log_mel_device, waveform_to_examples_device

    g = np.load(os.path.join(GOLD, "g6_logmel.npz"))
    lm = log_mel_device(g["wave"], 16000, dev)
    assert lm.dtype == torch.float64 and lm.shape == (298, 64)
    ex = avt.ops.logmel_examples(lm, 100, 10)
    assert ex.shape == g["examples"].shape and ex.dtype == torch.float32
    # float64 table vs the reference's float64 examples: summation order / twiddle last bits only
    ref = g["examples"]
    got64 = np.stack([lm.cpu().numpy()[10 * e : 10 * e + 100] for e in range(ref.shape[0])])
    np.testing.assert_allclose(got64, ref, rtol=0, atol=1e-10)
    # the float32 examples VGGish consumes = the cast validate.py:160-161 applies (<= 1 ulp of float32)
    np.testing.assert_allclose(ex.cpu().numpy(), ref.astype(np.float32), rtol=2e-7, atol=1e-7)
    ex2 = waveform_to_examples_device(g["wave"], 16000, dev)
    assert torch.equal(ex, ex2)


@pytest.mark.parametrize("n,dtype", [(399, np.float32), (400, np.float32), (559, np.float64), (560, np.float64),
                                     (16000 + 240, np.float32), (16000 * 7 + 13, np.float64)])
def test_logmel_ragged_lengths_vs_oracle(avt, dev, n, dtype):
    """Empty / one-frame / ragged tails (mel_features.frame drops incomplete frames and examples), both sample types."""
    from avtex.audio_frontend import log_mel_device

    rng = np.random.default_rng(n)
    wave = (0.3 * rng.standard_normal(n)).astype(dtype)
    lm = log_mel_device(wave, 16000, dev)
    ref = ref_py.log_mel(wave)
    assert tuple(lm.shape) == ref.shape
    if ref.shape[0]:
        np.testing.assert_allclose(lm.cpu().numpy(), ref, rtol=0, atol=1e-10)
    ex = avt.ops.logmel_examples(lm, 100, 10)
    ref_ex = ref_py.logmel_examples(ref)
    assert tuple(ex.shape) == ref_ex.shape
    if ref_ex.shape[0]:
        np.testing.assert_allclose(ex.cpu().numpy(), ref_ex.astype(np.float32), rtol=2e-7, atol=1e-7)


def test_logmel_silence_and_stereo(avt, dev):
    """All-zero input gives log(0.01) everywhere; [n,2] input is averaged to mono first (vggish_utils.py:44-46)."""
    from avtex.audio_frontend import log_mel_device

    lm = log_mel_device(np.zeros(1600, np.float32), 16000, dev)
    assert torch.allclose(lm, torch.full_like(lm, float(np.log(0.01))), rtol=0, atol=1e-15)
    rng = np.random.default_rng(1)
    st = rng.standard_normal((4000, 2))
    a, b = log_mel_device(st, 16000, dev), log_mel_device(st.mean(axis=1), 16000, dev)
    assert torch.equal(a, b)
