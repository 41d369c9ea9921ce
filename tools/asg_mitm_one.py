"""the four ASG criterion calls (FCC / FAC forward / backward) alone at the config-4 criterion shape, timed with events.
W2L_ASG_NOMITM=1 (probe library): the round-4 / round-5 full-length scans.   usage: asg_mitm_one.py [B T N L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
if os.environ.get("W2L_ASG_NOMITM") is not None or os.environ.get("W2L_MITM_ONLY") is not None: _lib.use_probe().__enter__()
B, T, N, Lt = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 2000, 30, 300)
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, N, generator=g).cuda(); trans = (torch.eye(N) * 4 + 0.1 * torch.randn(N, N, generator=g)).cuda()
tgt = torch.full((B, Lt), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(max(1, Lt // 5), Lt + 1, (1,), generator=g))
    y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
    for i in range(1, l):
        if y[i] == y[i - 1]: y[i] = (y[i] + 1) % 28
    tgt[b, :l] = y
tgt = tgt.cuda()
ts = torch.empty(B, dtype=torch.int32, device="cuda")
assert L.w2l_batch_target_size(B, Lt, T, tgt.data_ptr(), ts.data_ptr(), s) == 0
loss = torch.empty(B, device="cuda"); gl = torch.ones(B, device="cuda")
dx = torch.empty_like(x); dt = torch.empty(N, N, device="cuda")
wf = torch.empty(L.w2l_fcc_workspace_size(B, T, N), dtype=torch.uint8, device="cuda")
wa = torch.empty(L.w2l_fac_workspace_size(B, T, N, Lt), dtype=torch.uint8, device="cuda")
calls = {
    "fcc_fwd": lambda: L.w2l_fcc_forward(B, T, N, 4, x.data_ptr(), ts.data_ptr(), trans.data_ptr(), loss.data_ptr(), wf.data_ptr(), s),
    "fcc_bwd": lambda: L.w2l_fcc_backward(B, T, N, trans.data_ptr(), gl.data_ptr(), dx.data_ptr(), dt.data_ptr(), wf.data_ptr(), s),
    "fac_fwd": lambda: L.w2l_fac_forward(B, T, N, Lt, 4, x.data_ptr(), tgt.data_ptr(), ts.data_ptr(), trans.data_ptr(), loss.data_ptr(), wa.data_ptr(), s),
    "fac_bwd": lambda: L.w2l_fac_backward(B, T, N, Lt, tgt.data_ptr(), ts.data_ptr(), gl.data_ptr(), dx.data_ptr(), dt.data_ptr(), wa.data_ptr(), s),
}
out = []
for name, f in calls.items():
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    out.append("%s %.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
print("B=%d T=%d N=%d L=%d nomitm=%s only=%s: " % (B, T, N, Lt, os.environ.get("W2L_ASG_NOMITM", "-"), os.environ.get("W2L_MITM_ONLY", "-")) + "  ".join(out), flush=True)
