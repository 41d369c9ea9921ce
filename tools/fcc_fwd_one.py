"""w2l_fcc_forward alone at the config-4 criterion shape, timed with events.  (profiles/r04_run45_fcc_helper_ablations.log was taken with a
temporary timing-only switch W2L_FCC_HABL in fcc_fwd_dpp2's helper wave -- 1 no x-row loads, 2 no u / q stores -- that is not in the tree.)
usage: fcc_fwd_one.py [B T N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
if os.environ.get("W2L_FCC_HABL") is not None or os.environ.get("W2L_FCC_1WAVE") is not None: _lib.use_probe().__enter__()
B, T, N = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 2000, 30)
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
x = torch.randn(B, T, N, device="cuda"); trans = (torch.eye(N) * 4 + 0.1 * torch.randn(N, N)).cuda()
ts = torch.full((B,), 100, dtype=torch.int32, device="cuda"); loss = torch.empty(B, device="cuda")
ws = torch.empty(L.w2l_fcc_workspace_size(B, T, N), dtype=torch.uint8, device="cuda")
def f(): assert L.w2l_fcc_forward(B, T, N, 4, x.data_ptr(), ts.data_ptr(), trans.data_ptr(), loss.data_ptr(), ws.data_ptr(), s) == 0
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print("fcc_forward B=%d T=%d N=%d habl=%s 1wave=%s: %.1f us (%.0f cycles per frame at 2.4 GHz)" % (B, T, N, os.environ.get("W2L_FCC_HABL", "-"),
      os.environ.get("W2L_FCC_1WAVE", "-"), e0.elapsed_time(e1) / 20 * 1e3, e0.elapsed_time(e1) / 20 * 1e3 * 2400 / T))
