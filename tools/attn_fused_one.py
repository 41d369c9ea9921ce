"""w2l_attn_fused_forward at one geometry, timed with events; W2L_AF_ABL / W2L_AF_BPW (probe build) select ablations
usage: attn_fused_one.py B H T d csz p"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
if os.environ.get("W2L_AF_ABL") is not None or os.environ.get("W2L_AF_BPW") is not None: _lib.use_probe().__enter__()
B, H, T, d, csz = [int(v) for v in sys.argv[1:6]]
p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.2
L = _lib.lib(); Cc = H * d
q, k, v = [torch.randn(B, T, Cc, device="cuda") for _ in range(3)]
E = torch.randn(2 * csz - 1, d, device="cuda") * 0.5
n0 = csz - 1; rlo = max(0, n0 - (T - 1)); W = min(2 * csz - 1, n0 + T) - rlo
P = torch.empty(B, H, T, T, device="cuda"); Pd = torch.empty_like(P); ctx = torch.empty(B, T, Cc, device="cuda")
D = _lib.AttnFusedDesc(B=B, H=H, T=T, d=d, ld=Cc, ldc=Cc, W=W, n0=n0, rlo=rlo, scale=d ** -0.5, dropP=p, dropSeed=1, dropStream=2)
s = torch.cuda.current_stream().cuda_stream
def run():
    st = L.w2l_attn_fused_forward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr(), None, P.data_ptr(),
                                  Pd.data_ptr() if p > 0 else None, ctx.data_ptr(), s)
    assert st == 0, st
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("attn_fused B=%d H=%d T=%d d=%d csz=%d p=%.1f abl=%s bpw=%s: %.1f us" % (B, H, T, d, csz, p, os.environ.get("W2L_AF_ABL", "-"),
      os.environ.get("W2L_AF_BPW", "-"), e0.elapsed_time(e1) / 50 * 1e3))
