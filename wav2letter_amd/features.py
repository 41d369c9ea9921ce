"""MFSC / log-mel features on the device (the input pipeline next to the hot path).

Reference: fl::lib::audio::Mfsc [UNVENDORED] as the trainer configures it (recipes/slimIPL/src/Train.cpp:277-290:
useEnergy = usePower = zeroMeanFrame = false, --filterbanks=80 / 40, 25 ms frames every 10 ms) and as
recipes/streaming_convnets/inference/inference/module/feature/LogMelFeature.cpp:78-95 does for streaming.  Pre-emphasis, the Hamming window and the DFT are linear in the samples of a frame,
so they are folded (fp64, on the host, once) into ONE [frame x 2*bins] matrix; the frames of an utterance are
overlapping rows of its samples (row t starts at sample t*stride), so the spectra of all frames of a batch are
one GEMM whose A operand is the audio itself with leading dimension `stride` -- the same zero-copy operand view
as the conv_glu convolutions.  Then |.|, the mel filterbank GEMM, log(max(., floor)) and the transposition to
the network input layout [B][NFEAT][T].
"""
import math

import numpy as np
import torch

from . import _lib
from .ops import _p, _s, check


def _filterbank(num_filters, nfft, fs):
    nb = nfft // 2 + 1
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    imel = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    pts = imel(np.linspace(mel(0.0), mel(fs / 2.0), num_filters + 2)) * (nb - 1) * 2.0 / fs
    k = np.arange(nb, dtype=np.float64)[:, None]
    hi = (k - pts[None, :-2]) / (pts[None, 1:-1] - pts[None, :-2])
    lo = (pts[None, 2:] - k) / (pts[None, 2:] - pts[None, 1:-1])
    return np.maximum(np.minimum(hi, lo), 0.0)


class Mfsc:
    """log-mel filterbank features of a batch of equal-length utterances: audio [B][n_samples] -> [B][F][T]"""

    def __init__(self, num_filters=80, fs=16000, frame_ms=25, stride_ms=10, preem=0.97, use_power=False, mel_floor=1.0,
                 device="cuda"):
        self.F, self.fs = num_filters, fs
        self.N = int(round(1e-3 * frame_ms * fs))
        self.S = int(round(1e-3 * stride_ms * fs))
        self.nfft = 1 << (self.N - 1).bit_length()
        self.nb = self.nfft // 2 + 1
        self.use_power, self.floor = bool(use_power), float(mel_floor)
        n = np.arange(self.N, dtype=np.float64)
        win = 0.54 - 0.46 * np.cos(2.0 * np.pi * n / (self.N - 1))
        # frame -> windowed pre-emphasised frame: y = W P x,  (P x)[i] = x[i] - a x[i-1], (P x)[0] = (1 - a) x[0]
        WP = np.diag(win)
        P = np.eye(self.N) - preem * np.eye(self.N, k=-1)
        P[0, 0] = 1.0 - preem
        WP = WP @ P
        ang = 2.0 * np.pi * np.outer(n, np.arange(self.nb)) / self.nfft
        G = np.concatenate([WP.T @ np.cos(ang), -(WP.T @ np.sin(ang))], axis=1)   # [N][2*nb]: re | im
        self.G = torch.tensor(G.astype(np.float32), device=device)
        # spectrum rows are zero-padded to a multiple of 32 columns so that the mel GEMM has whole K tiles
        self.ld = (self.nb + 31) // 32 * 32
        H = np.zeros((self.ld, num_filters))
        H[:self.nb] = _filterbank(num_filters, self.nfft, fs)
        self.H = torch.tensor(H.astype(np.float32), device=device)

    def num_frames(self, n_samples):
        return 0 if n_samples < self.N else 1 + (n_samples - self.N) // self.S

    def __call__(self, audio):
        if audio.dim() != 2 or not audio.is_cuda or audio.dtype != torch.float32:
            raise _lib.W2LInvalidArgument("Mfsc: audio must be a float32 [B][n_samples] tensor on the device")
        B, ns = audio.shape
        T = self.num_frames(ns)
        if T <= 0:
            raise _lib.W2LInvalidArgument("Mfsc: utterance shorter than one frame")
        if ns % self.S:
            pad = self.S - ns % self.S                      # whole strides per utterance: row r = b*Tp + t of the flat audio
            audio = torch.nn.functional.pad(audio, (0, pad))
            ns += pad
        audio = audio.contiguous()
        Tp = ns // self.S
        L = _lib.lib()
        rows = (B * ns - self.N) // self.S + 1             # rows that straddle two utterances are computed and dropped
        reim = torch.empty(B * Tp, 2 * self.nb, device=audio.device, dtype=torch.float32)
        # frames = overlapping rows of the flat audio: lda = stride, K = frame length
        check(L.w2l_gemm_f32(rows, 2 * self.nb, self.N, _p(audio), self.S, 1, _p(self.G), 2 * self.nb, 0,
                             _p(reim), 2 * self.nb, None, 0, 1, _s()), "mfsc spectrum gemm")
        spec = torch.empty(B * Tp, self.ld, device=audio.device, dtype=torch.float32)
        check(L.w2l_mfsc_spectrum(_p(reim), _p(spec), rows, self.nb, self.ld, int(self.use_power), _s()), "mfsc spectrum")
        mel = torch.empty(B * Tp, self.F, device=audio.device, dtype=torch.float32)
        check(L.w2l_gemm_f32(rows, self.F, self.ld, _p(spec), self.ld, 1, _p(self.H), self.F, 0, _p(mel), self.F,
                             None, 0, 1, _s()), "mfsc mel gemm")
        out = torch.empty(B, self.F, T, device=audio.device, dtype=torch.float32)
        check(L.w2l_mfsc_log_transpose(_p(mel), _p(out), B, Tp, T, self.F, self.floor, _s()), "mfsc log")
        return out
