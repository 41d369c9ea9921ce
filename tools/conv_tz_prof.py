"""The TDS convolutions of am_tds_ctc.arch at B = 32 (C = 10 / 14 / 18: forward + ReLU, backward-data + addend, backward-filter), `reps` calls
each, for `rocprofv3 --kernel-trace --stats`: per-kernel average durations of the block-Toeplitz generation.   python tools/conv_tz_prof.py [reps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from wav2letter_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
a = torch.randn(4096, 4096, device="cuda")
for _ in range(10):
    a @ a
torch.cuda.synchronize()
for (Cc, T) in [(10, 750), (14, 375), (18, 188)]:
    B, H, kw = 32, 80, 21
    d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
    x = torch.randn(B, T, H, Cc, device="cuda")
    w = torch.randn(kw, Cc, Cc, device="cuda") / (kw * Cc) ** 0.5
    b = torch.randn(Cc, device="cuda")
    dy = torch.randn(B, T, H, Cc, device="cuda")
    add = torch.randn(B, T, H, Cc, device="cuda")
    y = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    for _ in range(reps):
        assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
    for _ in range(reps):
        assert L.w2l_conv_backward_data_add(C.byref(d), dy.data_ptr(), w.data_ptr(), add.data_ptr(), dx.data_ptr(), s) == 0
    for _ in range(reps):
        assert L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s) == 0
    torch.cuda.synchronize()
print("done")
