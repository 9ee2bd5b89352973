"""Oracle: VGGish audio encoder and its log-mel front end (test infrastructure only).

VGGish follows /root/reference/models/backbone.py:16-66 (pre-processing twin:
abaw5_pre_processing/base/vggish/vggish.py:9-32).  The front end follows
abaw5_pre_processing/base/vggish/mel_features.py:21-49 (my_frame), :75-92 (periodic Hann),
:95-114 (STFT magnitude), :134-204 (HTK mel matrix, DC row zeroed), :207-236 (log mel) and
vggish_input.py:37-98 (framing into 96x64 examples with a FRACTIONAL hop rounded by Python's
round-half-to-even; int16/32768 scaling and the 1 s edge pad), constants vggish_params.py.
All arithmetic is float64 numpy like the reference.
"""
import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
WINDOW_SAMPLES = 400   # 25 ms
HOP_SAMPLES = 160      # 10 ms
FFT_LENGTH = 512
NUM_MEL_BINS = 64
MEL_MIN_HZ, MEL_MAX_HZ = 125.0, 7500.0
LOG_OFFSET = 0.01
CONV_KEYS = (0, 3, 6, 8, 11, 13)       # nn.Sequential indices of the conv layers
POOL_AFTER = (0, 3, 8, 13)             # a 2x2 max-pool follows these convs


def vggish_forward(x, sd, prefix=""):
    """x: [n,96,64] log-mel examples -> [n,128]."""
    x = torch.as_tensor(x)[:, None, :, :].float()
    for i in CONV_KEYS:
        x = F.relu(F.conv2d(x, sd[f"{prefix}features.{i}.weight"], sd[f"{prefix}features.{i}.bias"], 1, 1))
        if i in POOL_AFTER:
            x = F.max_pool2d(x, 2, 2)
    x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)  # (H, W, C) flatten
    x = F.relu(F.linear(x, sd[prefix + "embeddings.0.weight"], sd[prefix + "embeddings.0.bias"]))
    x = F.relu(F.linear(x, sd[prefix + "embeddings.2.weight"], sd[prefix + "embeddings.2.bias"]))
    return F.linear(x, sd[prefix + "embeddings.4.weight"], sd[prefix + "embeddings.4.bias"])


def hertz_to_mel(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_matrix(num_mel_bins=NUM_MEL_BINS, num_spectrogram_bins=FFT_LENGTH // 2 + 1, sample_rate=SAMPLE_RATE,
               lower=MEL_MIN_HZ, upper=MEL_MAX_HZ):
    bins_mel = hertz_to_mel(np.linspace(0.0, sample_rate / 2.0, num_spectrogram_bins))
    edges = np.linspace(hertz_to_mel(lower), hertz_to_mel(upper), num_mel_bins + 2)
    w = np.empty((num_spectrogram_bins, num_mel_bins))
    for i in range(num_mel_bins):
        lo, ce, up = edges[i:i + 3]
        w[:, i] = np.maximum(0.0, np.minimum((bins_mel - lo) / (ce - lo), (up - bins_mel) / (up - ce)))
    w[0, :] = 0.0
    return w


def periodic_hann(n):
    return 0.5 - 0.5 * np.cos(2 * np.pi / n * np.arange(n))


def log_mel_spectrogram(samples):
    """[S] float64 samples at 16 kHz -> [num_frames, 64] log-mel."""
    n = 1 + int(np.floor((len(samples) - WINDOW_SAMPLES) / HOP_SAMPLES))
    frames = np.stack([samples[i * HOP_SAMPLES:i * HOP_SAMPLES + WINDOW_SAMPLES] for i in range(n)])
    spec = np.abs(np.fft.rfft(frames * periodic_hann(WINDOW_SAMPLES), FFT_LENGTH))
    return np.log(spec @ mel_matrix() + LOG_OFFSET)


def example_starts(num_frames, window_frames, hop_frames):
    """Start rows of the examples: Python round() (half to even) of i * hop, hop fractional."""
    n = 1 + int(np.floor((num_frames - window_frames) / hop_frames))
    return [round(hop_frames * i) for i in range(max(n, 0))]


def waveform_to_examples(samples, sample_rate, window_sec, hop_sec):
    if sample_rate != SAMPLE_RATE:
        raise ValueError("the oracle has no resampler: feed 16 kHz audio (the reference resamples with resampy)")
    lm = log_mel_spectrogram(np.asarray(samples, dtype=np.float64))
    win = int(round(window_sec * 100.0))
    starts = example_starts(lm.shape[0], win, hop_sec * 100.0)
    return np.stack([lm[s:s + win] for s in starts])


def wav_int16_to_examples(pcm_int16, sample_rate, window_sec, hop_sec):
    """vggish_input.py:84-98: int16 -> [-1,1), pad one second of edge samples, then frame."""
    samples = np.asarray(pcm_int16).astype(np.float64) / 32768.0
    samples = np.pad(samples, (0, sample_rate), "edge")
    return waveform_to_examples(samples, sample_rate, window_sec, hop_sec)
