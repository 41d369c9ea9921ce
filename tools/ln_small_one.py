"""per-frame LayerNorm (groups of 1024 ... 2160 floats): forward (residual + dropout) and backward (+ masked copy) timings and the
output checksums.  A/B: W2L_HIP_SO=.../libw2l_hip_probe.so W2L_LN_WAVE=0|1 python tools/ln_small_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib, ops
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

tag = os.environ.get("W2L_LN_WAVE", "default")
for G, inner in [(11968, 1200), (11968, 1520), (11968, 2160), (3008, 1024), (48000, 320)]:
    g = torch.Generator(device="cuda").manual_seed(1)
    a0 = torch.randn(G * inner, device="cuda", generator=g); x = torch.randn(G * inner, device="cuda", generator=g)
    dy = torch.randn(G * inner, device="cuda", generator=g); gb = torch.tensor([1.3, 0.2], device="cuda")
    a = a0.clone(); r = torch.empty_like(a); y = torch.empty_like(a); mr = torch.empty(2 * G, device="cuda")
    st = torch.empty(int(L.w2l_layernorm_scratch_doubles(G, inner)), device="cuda", dtype=torch.float64)
    dr = torch.empty_like(a); dm = torch.empty_like(a); dgb = torch.empty(2, device="cuda")
    P = lambda t: t.data_ptr()
    f1 = lambda: L.w2l_residual_layernorm_forward(G, inner, P(a), P(x), P(r), P(y), P(gb), 1e-5, 0.1, 7, 3, P(st), P(mr), s)     # LN1 form: dropout + residual, r apart
    f2 = lambda: L.w2l_residual_layernorm_forward(G, inner, P(r), None, P(r), P(y), P(gb), 1e-5, 0.0, 0, 0, P(st), P(mr), s)      # plain LayerNorm of r
    b1 = lambda: L.w2l_layernorm_backward(G, inner, P(r), P(dy), P(gb), P(mr), P(dr), P(dgb), P(a), P(dm), 1.1, P(st), s)
    b2 = lambda: L.w2l_layernorm_backward(G, inner, P(r), P(dy), P(gb), P(mr), P(dr), P(dgb), None, None, 1.0, P(st), s)
    a.copy_(a0); f1(); torch.cuda.synchronize()
    ck = [float(y.double().sum()), float(y.double().abs().sum()), float(mr.double().sum())]
    b1(); torch.cuda.synchronize()
    ck += [float(dr.double().abs().sum()), float(dm.double().abs().sum()), float(dgb.double().sum())]
    n = G * inner * 4 / 1e6
    t = [timeit(f) for f in (f1, f2, b1, b2)]
    print(f"[wave={tag}] {G} x {inner}: fwd(dropout+residual) {t[0]:.1f} us ({5 * n / t[0]:.2f} TB/s) | fwd(plain) {t[1]:.1f} us ({2 * n / t[1]:.2f} TB/s) | "
          f"bwd(+mask) {t[2]:.1f} us ({5 * n / t[2]:.2f} TB/s) | bwd {t[3]:.1f} us ({3 * n / t[3]:.2f} TB/s) | checks " + " ".join(f"{c:.9e}" for c in ck), flush=True)
