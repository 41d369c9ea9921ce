import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
B, T, H, Cc, kw = 32, 750, 80, 10, 21
d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
x = torch.randn(B, T, H, Cc, device="cuda"); w = torch.randn(kw, Cc, Cc, device="cuda"); b = torch.randn(Cc, device="cuda")
y = torch.empty_like(x)
for _ in range(5):
    L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
torch.cuda.synchronize()
