import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
M, K, N = [int(v) for v in sys.argv[1:4]]
which = sys.argv[4] if len(sys.argv) > 4 else "fwd"
x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
dy = torch.randn(M, N, device="cuda"); y = torch.empty(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N, device="cuda")
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    if which == "fwd": L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
    elif which == "dx": L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s)
    else: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s)
torch.cuda.synchronize()
