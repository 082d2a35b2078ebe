"""Waveform -> VGGish log-mel examples, once per video.

`waveform_to_examples_device` is what validate() runs: the STFT / mel / log / framing arithmetic on the MI355X
(csrc/logmel.hip, float64 like the reference) from tables built here.  `waveform_to_examples` is the same arithmetic
in host NumPy for CPU-side callers (the training DataLoader's workers, dataset.py) where no device is bound.

Same arithmetic as the reference's TF-VGGish front-end
(contrastive_video_textures/utils/vggish_utils.py:27-69, mel_features.py:21-205,
vggish_params.py:27-38): 16 kHz mono, STFT 25 ms / 10 ms with a periodic Hann,
512-point FFT magnitude, 64 HTK-mel bands 125-7500 Hz, log(mel + 0.01), framed
into examples of 1.0 s (100 frames) at a hop of 0.1 s (10 frames) [quirk Q5:
the segment stride is 0.2 s, the audio hop 0.1 s].  Returns float64 like the
reference; callers cast to fp32 (validate.py:160-161).
"""
import numpy as np

SAMPLE_RATE = 16000
STFT_WINDOW_SECONDS = 0.025
STFT_HOP_SECONDS = 0.010
NUM_MEL_BINS = 64
MEL_MIN_HZ = 125.0
MEL_MAX_HZ = 7500.0
LOG_OFFSET = 0.01
EXAMPLE_WINDOW_SECONDS = 1.0
EXAMPLE_HOP_SECONDS = 0.1
_MEL_BREAK_HZ = 700.0
_MEL_Q = 1127.0


def frame(data, window_length, hop_length):
    """[n, ...] -> [num_frames, window_length, ...] strided view; incomplete tail dropped."""
    n = data.shape[0]
    num = 1 + int(np.floor((n - window_length) / hop_length))
    shape = (num, window_length) + data.shape[1:]
    strides = (data.strides[0] * hop_length,) + data.strides
    return np.lib.stride_tricks.as_strided(data, shape=shape, strides=strides)


def _hz_to_mel(hz):
    return _MEL_Q * np.log(1.0 + (hz / _MEL_BREAK_HZ))


def mel_matrix(num_mel_bins, num_spec_bins, sample_rate, lo_hz, hi_hz):
    nyq = sample_rate / 2.0
    if lo_hz < 0.0 or lo_hz >= hi_hz or hi_hz > nyq:
        raise ValueError("bad mel band edges %.1f..%.1f (nyquist %.1f)" % (lo_hz, hi_hz, nyq))
    spec_mel = _hz_to_mel(np.linspace(0.0, nyq, num_spec_bins))
    edges = np.linspace(_hz_to_mel(lo_hz), _hz_to_mel(hi_hz), num_mel_bins + 2)
    m = np.empty((num_spec_bins, num_mel_bins))
    for i in range(num_mel_bins):
        lo, ce, hi = edges[i : i + 3]
        up = (spec_mel - lo) / (ce - lo)
        down = (hi - spec_mel) / (hi - ce)
        m[:, i] = np.maximum(0.0, np.minimum(up, down))
    m[0, :] = 0.0  # HTK drops the DC bin
    return m


def log_mel_spectrogram(data, sample_rate=SAMPLE_RATE):
    win = int(round(sample_rate * STFT_WINDOW_SECONDS))
    hop = int(round(sample_rate * STFT_HOP_SECONDS))
    fft_len = 2 ** int(np.ceil(np.log(win) / np.log(2.0)))
    hann = 0.5 - (0.5 * np.cos(2 * np.pi / win * np.arange(win)))  # periodic
    spec = np.abs(np.fft.rfft(frame(data, win, hop) * hann, int(fft_len)))
    mel = np.dot(spec, mel_matrix(NUM_MEL_BINS, spec.shape[1], sample_rate, MEL_MIN_HZ, MEL_MAX_HZ))
    return np.log(mel + LOG_OFFSET)


def waveform_to_examples(data, sample_rate):
    """-> float64 [num_examples, 100, 64]."""
    data = np.asarray(data)
    if data.ndim > 1:
        data = np.mean(data, axis=1)
    if sample_rate != SAMPLE_RATE:
        data = _resample(data, sample_rate, SAMPLE_RATE)
    log_mel = log_mel_spectrogram(data, SAMPLE_RATE)
    rate = 1.0 / STFT_HOP_SECONDS
    return frame(log_mel, int(round(EXAMPLE_WINDOW_SECONDS * rate)), int(round(EXAMPLE_HOP_SECONDS * rate)))


def _resample(data, sr_in, sr_out):
    """The reference calls resampy (kaiser_best); this image has scipy only, so use its polyphase
    resampler.  Not bit-identical to resampy — parity for non-16 kHz input is unpinned."""
    from fractions import Fraction

    from scipy.signal import resample_poly

    fr = Fraction(int(sr_out), int(sr_in)).limit_denominator(1000)
    return resample_poly(data, fr.numerator, fr.denominator)


_TABLES = {}


def _device_tables(device):
    """Periodic Hann window and HTK mel matrix (float64) resident on `device`."""
    key = str(device)
    if key not in _TABLES:
        import torch

        win = int(round(SAMPLE_RATE * STFT_WINDOW_SECONDS))
        fft_len = 2 ** int(np.ceil(np.log(win) / np.log(2.0)))
        hann = 0.5 - (0.5 * np.cos(2 * np.pi / win * np.arange(win)))
        mel = mel_matrix(NUM_MEL_BINS, fft_len // 2 + 1, SAMPLE_RATE, MEL_MIN_HZ, MEL_MAX_HZ)
        _TABLES[key] = (torch.from_numpy(hann).to(device), torch.from_numpy(np.ascontiguousarray(mel)).to(device), fft_len)
    return _TABLES[key]


def log_mel_device(data, sample_rate, device):
    """-> float64 device tensor [num_frames, 64] (mel_features.log_mel_spectrogram on the GPU)."""
    import torch

    from . import ops

    data = np.asarray(data)
    if data.ndim > 1:
        data = np.mean(data, axis=1)
    if sample_rate != SAMPLE_RATE:
        data = _resample(data, sample_rate, SAMPLE_RATE)
    if data.dtype not in (np.float32, np.float64):
        data = data.astype(np.float64)
    hann, mel, fft_len = _device_tables(device)
    wave = torch.from_numpy(np.ascontiguousarray(data)).to(device)
    return ops.logmel(wave, hann, mel, int(round(SAMPLE_RATE * STFT_HOP_SECONDS)), fft_len, LOG_OFFSET)


def waveform_to_examples_device(data, sample_rate, device):
    """-> float32 device tensor [num_examples, 100, 64]: waveform_to_examples + the .float() of validate.py:160-161."""
    from . import ops

    rate = 1.0 / STFT_HOP_SECONDS
    return ops.logmel_examples(log_mel_device(data, sample_rate, device), int(round(EXAMPLE_WINDOW_SECONDS * rate)),
                               int(round(EXAMPLE_HOP_SECONDS * rate)))
