"""Mel front end with torchaudio-0.7 ``transforms.MelSpectrogram`` semantics and buffer names, as the reference
configures it (core/pipelines/voice2pose.py:27-30; pose2pose.py:25-28): win 400 (periodic Hann), hop 160,
n_fft 512, f 55..7500 Hz, 80 HTK mel bins, power 2, no log.  Computed by the gfx950 kernels: the STFT is an
fp32-MFMA GEMM against a windowed DFT basis (ops.mel_spectrogram).

torchaudio 0.7.0 itself is not available offline, so this follows its documented algorithm; checkpoints that
carry ``spectrogram.window`` / ``mel_scale.fb`` buffers override the values computed here."""
import math

import torch
from torch import nn

from . import ops


def htk_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate):
    """Triangular HTK-mel filters, (n_freqs, n_mels), fp32 arithmetic like torchaudio 0.7 create_fb_matrix."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_lo = 2595.0 * math.log10(1.0 + f_min / 700.0)
    mel_hi = 2595.0 * math.log10(1.0 + f_max / 700.0)
    edges = 700.0 * (10 ** (torch.linspace(mel_lo, mel_hi, n_mels + 2) / 2595.0) - 1.0)
    width = edges[1:] - edges[:-1]
    dist = edges.unsqueeze(0) - freqs.unsqueeze(1)  # (n_freqs, n_mels+2)
    rising = -dist[:, :-2] / width[:-1]
    falling = dist[:, 2:] / width[1:]
    return torch.minimum(rising, falling).clamp_min(0.0)


class _Window(nn.Module):
    def __init__(self, win_length):
        super().__init__()
        self.register_buffer('window', torch.hann_window(win_length, periodic=True))  # torchaudio's default window_fn


class _MelScale(nn.Module):
    def __init__(self, n_freqs, f_min, f_max, n_mels, sample_rate):
        super().__init__()
        self.register_buffer('fb', htk_filterbank(n_freqs, f_min, f_max, n_mels, sample_rate))


class MelSpectrogram(nn.Module):
    def __init__(self, sample_rate=16000, n_fft=512, win_length=400, hop_length=160, f_min=55, f_max=7500.0, n_mels=80):
        super().__init__()
        if (n_fft, win_length, hop_length) != (ops.N_FFT, ops.WIN, ops.HOP):
            raise NotImplementedError('the STFT kernel path is specialised to n_fft=512 / win=400 / hop=160')
        self.spectrogram = _Window(win_length)
        self.mel_scale = _MelScale(n_fft // 2 + 1, float(f_min), float(f_max), n_mels, sample_rate)
        self._basis = None
        self._basis_key = None
        self._bins = None
        self._bins_key = None

    def _dft_basis(self):
        w = self.spectrogram.window
        key = (w.data_ptr(), w._version, w.device)
        if self._basis is None or self._basis_key != key:
            self._basis = ops.dft_basis(w).to(w.device)
            self._basis_key = key
        return self._basis

    def _fb_bins(self):
        fb = self.mel_scale.fb
        key = (fb.data_ptr(), fb._version, fb.device)
        if self._bins is None or self._bins_key != key:
            self._bins = ops.fb_bin_ranges(fb)
            self._bins_key = key
        return self._bins

    @torch.no_grad()
    def forward(self, waveform):
        """(B, L) -> (B, n_mels, 1 + L // hop)."""
        return ops.mel_spectrogram(waveform, self._dft_basis(), self.mel_scale.fb, self._fb_bins())
