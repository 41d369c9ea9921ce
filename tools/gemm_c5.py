"""GEMM shapes of a Transformer block at BASELINE config 5 (M = 16 x 188 = 3008 frames, width 1024 / 4096): each fl::Linear
call alone, and the three independent q / k / v projections (forward, weight gradient) on ONE stream against THREE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib

L = _lib.use_probe().__enter__() if "--probe" in sys.argv else _lib.lib()
sys.argv = [a for a in sys.argv if a != "--probe"]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3008
main = torch.cuda.current_stream()
side = [torch.cuda.Stream(), torch.cuda.Stream()]
a = torch.randn(4096, 4096, device="cuda"); c = torch.empty(4096, 4096, device="cuda")
for _ in range(60): L.w2l_linear_forward(4096, 4096, 4096, a.data_ptr(), a.data_ptr(), None, c.data_ptr(), 0, main.cuda_stream)
torch.cuda.synchronize()


def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main)
    for _ in range(n): f()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K, N in [(1024, 1024), (1024, 4096), (4096, 1024)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda"); dy = torch.randn(M, N, device="cuda"); dx = torch.empty(M, K, device="cuda"); dw = torch.empty(K, N, device="cuda")
    fl = 2.0 * M * K * N
    s = main.cuda_stream
    t = {"fwd": timeit(lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 0, s)),
         "bwd_data": timeit(lambda: L.w2l_linear_backward_data(M, K, N, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, None, 1.0, s)),
         "bwd_weight": timeit(lambda: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))}
    print(f"M={M} in={K} out={N}: " + "  ".join(f"{k} {v:6.1f} us = {fl / v / 1e6:5.1f} TF/s" for k, v in t.items()), flush=True)

# q / k / v: three independent GEMMs off the same input
K = N = 1024
x = torch.randn(M, K, device="cuda")
ws = [torch.randn(K, N, device="cuda") / 32 for _ in range(3)]
ys = [torch.empty(M, N, device="cuda") for _ in range(3)]
dys = [torch.randn(M, N, device="cuda") for _ in range(3)]
dws = [torch.empty(K, N, device="cuda") for _ in range(3)]
ev = [torch.cuda.Event() for _ in range(4)]


def one_stream(kind):
    s = main.cuda_stream
    for i in range(3):
        if kind == "fwd": L.w2l_linear_forward(M, K, N, x.data_ptr(), ws[i].data_ptr(), None, ys[i].data_ptr(), 0, s)
        else: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dys[i].data_ptr(), dws[i].data_ptr(), s)


def three_streams(kind):
    ev[0].record(main)
    for i in range(3):
        st = main if i == 0 else side[i - 1]
        if i: st.wait_event(ev[0])
        s = st.cuda_stream
        if kind == "fwd": L.w2l_linear_forward(M, K, N, x.data_ptr(), ws[i].data_ptr(), None, ys[i].data_ptr(), 0, s)
        else: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dys[i].data_ptr(), dws[i].data_ptr(), s)
        if i:
            ev[i].record(st)
            main.wait_event(ev[i])


fl3 = 3 * 2.0 * M * K * N
for kind in ("fwd", "bwd_weight"):
    t1 = timeit(lambda: one_stream(kind)); t3 = timeit(lambda: three_streams(kind))
    print(f"q/k/v {kind}: one stream {t1:6.1f} us = {fl3 / t1 / 1e6:5.1f} TF/s   three streams {t3:6.1f} us = {fl3 / t3 / 1e6:5.1f} TF/s", flush=True)
# fused N = 3072 for comparison (what a concatenated weight would give)
w3 = torch.randn(K, 3 * N, device="cuda") / 32; y3 = torch.empty(M, 3 * N, device="cuda")
t = timeit(lambda: L.w2l_linear_forward(M, K, 3 * N, x.data_ptr(), w3.data_ptr(), None, y3.data_ptr(), 0, main.cuda_stream))
print(f"q/k/v fwd as ONE GEMM out=3072: {t:6.1f} us = {fl3 / t / 1e6:5.1f} TF/s", flush=True)
