"""One shape of one TDS convolution kernel, repeated, for rocprofv3 / counter passes (am_tds_ctc.arch stages: C = 10 / 14 / 18,
kw = 21, H = 80, B = 32):   python tools/conv_one.py [fwd | fwd3 | filter] [C]
  fwd     forward through the product library (the round-1 counter passes r17_conv_sq*)
  fwd3    forward on the third-generation kernel through the probe library (W2L_TDS_RS3=1: the r2h / r2i rs3 passes)
  filter  filter gradient through the product library (the r2i rsf3 passes)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib

kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 10
assert kind in ("fwd", "fwd3", "filter"), kind
T = {10: 750, 14: 375, 18: 188}[Cc]
B, H, kw = 32, 80, 21
d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
x = torch.randn(B, T, H, Cc, device="cuda"); w = torch.randn(kw, Cc, Cc, device="cuda"); b = torch.randn(Cc, device="cuda")
y = torch.empty_like(x); dy = torch.randn_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)


def go(L, reps):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(reps):
        if kind == "filter":
            assert L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s) == 0
        else:
            assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
    torch.cuda.synchronize()


if kind == "fwd3":
    os.environ["W2L_TDS_RS3"] = "1"
    os.environ["W2L_TDS_RS_C14"] = "1"
    with _lib.use_probe() as P:
        go(P, 5)
else:
    go(_lib.lib(), 8 if kind == "filter" else 5)
