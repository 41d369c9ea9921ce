"""one shape of the TDS filter gradient (product library) for rocprofv3 passes: python tools/conv_one3f.py [C]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = {10: 750, 14: 375, 18: 188}[Cc]
B, H, kw = 32, 80, 21
d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
x = torch.randn(B, T, H, Cc, device="cuda"); dy = torch.randn(B, T, H, Cc, device="cuda")
dw = torch.empty(kw, Cc, Cc, device="cuda"); db = torch.empty(Cc, device="cuda")
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
for _ in range(8):
    L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s)
torch.cuda.synchronize()
