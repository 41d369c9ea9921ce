"""filter-gradient time against the amount of work (B): slope = per-tile cost, intercept = launch + set-up + the reductions"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
a = torch.randn(4096, 4096, device="cuda")
for _ in range(20):
    a @ a
for Cc, T in ((10, 750), (18, 188)):
    row = []
    for B in (8, 16, 32, 64, 128):
        H, kw = 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        x = torch.randn(B, T, H, Cc, device="cuda"); dy = torch.randn(B, T, H, Cc, device="cuda")
        dw = torch.empty(kw, Cc, Cc, device="cuda"); db = torch.empty(Cc, device="cuda")
        for _ in range(3):
            L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"C={Cc}: " + "  ".join(f"B={B}: {t:7.1f} us" for B, t in zip((8, 16, 32, 64, 128), row)))
