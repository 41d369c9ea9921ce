"""one shape of the third-generation TDS convolution kernel (probe library, W2L_TDS_RS3=1) for counter passes:
python tools/conv_one3.py [C]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = {10: 750, 14: 375, 18: 188}[Cc]
os.environ["W2L_TDS_RS3"] = "1"
os.environ["W2L_TDS_RS_C14"] = "1"
B, H, kw = 32, 80, 21
d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
x = torch.randn(B, T, H, Cc, device="cuda"); w = torch.randn(kw, Cc, Cc, device="cuda"); b = torch.randn(Cc, device="cuda")
y = torch.empty_like(x)
with _lib.use_probe() as P:
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        P.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
    torch.cuda.synchronize()
