"""CTC forward + backward through the C ABI at one shape, repeated, for rocprofv3 passes (per-kernel time of ctc_rows_lse / ctc_scan /
ctc_rows_grad):   python tools/ctc_one.py [T] [reps]      (B = 32, N = 9998, L <= 80: the shapes of bench.py's CTC leg)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib, criterion as Cr
if os.environ.get("W2L_CTC_LSE_VAR"): _lib.use_probe().__enter__()   # probe-library variants of ctc_rows_lse

T = int(sys.argv[1]) if len(sys.argv) > 1 else 188
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, N, Lt = 32, 9998, 80
L = _lib.lib()
g = torch.Generator(device="cpu").manual_seed(11)
x = torch.randn(B, T, N, generator=g).cuda()
tgt = torch.full((B, Lt), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(20, Lt + 1, (1,), generator=g))
    tgt[b, :l] = torch.randint(0, N - 1, (l,), generator=g, dtype=torch.int32)
tgt = tgt.cuda()
ts = Cr.batch_target_size(tgt, T, ctc=True)
ws = torch.empty(L.w2l_ctc_workspace_size(B, T, N, Lt), dtype=torch.uint8, device="cuda")
loss = torch.empty(B, device="cuda"); grad = torch.ones(B, device="cuda"); dx = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    assert L.w2l_ctc_forward(B, T, N, Lt, 4, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), loss.data_ptr(), ws.data_ptr(), st) == 0
    assert L.w2l_ctc_backward(B, T, N, Lt, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), grad.data_ptr(), dx.data_ptr(), ws.data_ptr(), st) == 0
torch.cuda.synchronize()
print("loss", float(loss.sum()))
