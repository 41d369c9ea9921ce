"""w2l_attn_fused_backward at one geometry, timed with events, next to the unfused launch sequence it replaces;
W2L_AB_BPW (probe build) selects the query blocks per workgroup.  usage: attn_fused_bwd_one.py B H T d csz p"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
if os.environ.get("W2L_AB_BPW") is not None: _lib.use_probe().__enter__()
B, H, T, d, csz = [int(v) for v in sys.argv[1:6]]
p = float(sys.argv[6]) if len(sys.argv) > 6 else 0.2
L = _lib.lib(); Cc = H * d
q, k, v, dctx = [torch.randn(B, T, Cc, device="cuda") for _ in range(4)]
E = torch.randn(2 * csz - 1, d, device="cuda") * 0.5
n0 = csz - 1; rlo = max(0, n0 - (T - 1)); W = min(2 * csz - 1, n0 + T) - rlo; ldr = (W + 3) // 4 * 4
P = torch.empty(B, H, T, T, device="cuda"); Pd = torch.empty_like(P); ctx = torch.empty(B, T, Cc, device="cuda")
D = _lib.AttnFusedDesc(B=B, H=H, T=T, d=d, ld=Cc, ldc=Cc, W=W, n0=n0, rlo=rlo, scale=d ** -0.5, dropP=p, dropSeed=1, dropStream=2)
s = torch.cuda.current_stream().cuda_stream
assert L.w2l_attn_fused_forward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr(), None, P.data_ptr(),
                                Pd.data_ptr() if p > 0 else None, ctx.data_ptr(), s) == 0
if p == 0: Pd = P
nws = L.w2l_attn_fused_backward_workspace(C.byref(D), 1)
ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
dq, dk, dv = [torch.empty(B, T, Cc, device="cuda") for _ in range(3)]
dE = torch.empty(2 * csz - 1, d, device="cuda")
def fused():
    st = L.w2l_attn_fused_backward(C.byref(D), q.data_ptr(), k.data_ptr(), v.data_ptr(), E.data_ptr(), P.data_ptr(), dctx.data_ptr(),
                                   dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dE.data_ptr(), ws.data_ptr(), nws, s)
    assert st == 0, st
TC, TT = T * Cc, T * T
dS = torch.empty(B, H, T, T, device="cuda"); dR = torch.empty(B * T * H, ldr, device="cuda"); dEp = torch.empty(B, W, d, device="cuda")
G = _lib.BgemmDesc
def unfused():
    g = G(M=T, N=T, K=d, G1=B, G2=H, sam=Cc, sak=1, a1=TC, a2=d, sbk=1, sbn=Cc, b1=TC, b2=d, ldc=T, c1=H * TT, c2=TT)
    L.w2l_bgemm_bf16(C.byref(g), dctx.data_ptr(), v.data_ptr(), dS.data_ptr(), s)
    g = G(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    L.w2l_bgemm_bf16(C.byref(g), Pd.data_ptr(), dctx.data_ptr(), dv.data_ptr(), s)
    if p > 0: L.w2l_dropout_inplace(dS.data_ptr(), B * H * TT, p, 1, 2, s)
    L.w2l_attn_softmax_backward(P.data_ptr(), dS.data_ptr(), dR.data_ptr(), B, H, T, ldr, rlo, W, n0, d ** -0.5, s)
    g = G(M=T, N=d, K=T, G1=B, G2=H, sam=T, sak=1, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    L.w2l_bgemm_bf16(C.byref(g), dS.data_ptr(), k.data_ptr(), dq.data_ptr(), s)
    g = G(M=T, N=d, K=T, G1=B, G2=H, sam=1, sak=T, a1=H * TT, a2=TT, sbk=Cc, sbn=1, b1=TC, b2=d, ldc=Cc, c1=TC, c2=d)
    L.w2l_bgemm_bf16(C.byref(g), dS.data_ptr(), q.data_ptr(), dk.data_ptr(), s)
    g = G(M=B * T * H, N=d, K=W, G1=1, G2=1, sam=ldr, sak=1, sbk=d, sbn=1, ldc=d, accumulate=1, bandMode=1, bandT=T, bandH=H, bandOff=n0 - rlo)
    L.w2l_bgemm_bf16(C.byref(g), dR.data_ptr(), E[rlo:].data_ptr(), dq.data_ptr(), s)
    g = G(M=W, N=d, K=T * H, G1=B, G2=1, sam=1, sak=ldr, a1=T * H * ldr, sbk=d, sbn=1, b1=T * Cc, ldc=d, c1=W * d, bandMode=2, bandT=T, bandH=H,
          bandOff=n0 - rlo)
    L.w2l_bgemm_bf16(C.byref(g), dR.data_ptr(), q.data_ptr(), dEp.data_ptr(), s)
    L.w2l_fill(dE.data_ptr(), dE.numel(), 0.0, s)
    L.w2l_colsum(dEp.data_ptr(), dE[rlo:].data_ptr(), B, W * d, s)
def timeit(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("attention backward B=%d H=%d T=%d d=%d csz=%d p=%.1f bpw=%s: fused %.1f us, unfused sequence %.1f us" % (
    B, H, T, d, csz, p, os.environ.get("W2L_AB_BPW", "-"), timeit(fused), timeit(unfused)))
