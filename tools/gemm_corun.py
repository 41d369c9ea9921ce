"""GEMM throughput with a co-resident kernel holding CU slots (stand-in for the RCCL all-reduce that runs on a side
stream under the backward pass): how much do the persistent GEMMs lose when `blocks` workgroups of another kernel
occupy part of the chip?"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib

L = _lib.lib()
L.w2l_selftest_spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
main = torch.cuda.current_stream()
side = torch.cuda.Stream()
sink = torch.zeros(4, device="cuda")
a = torch.randn(4096, 4096, device="cuda"); c = torch.empty(4096, 4096, device="cuda")
for _ in range(60): L.w2l_linear_forward(4096, 4096, 4096, a.data_ptr(), a.data_ptr(), None, c.data_ptr(), 0, main.cuda_stream)
torch.cuda.synchronize()

def run(M, K, N, blocks, threads, lds, n=30):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; y = torch.empty(M, N, device="cuda")
    f = lambda: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), None, y.data_ptr(), 0, main.cuda_stream)
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if blocks:
        L.w2l_selftest_spin(blocks, threads, lds, 30000, sink.data_ptr(), side.cuda_stream)   # 30 ms of co-resident spinning
    e0.record(main)
    for _ in range(n): f()
    e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (M, K, N) in [(24000, 800, 2400), (24000, 2400, 800), (6016, 1440, 4320), (6016, 4320, 1440), (4096, 4096, 4096)]:
    run(M, K, N, 0, 0, 0)
    base = run(M, K, N, 0, 0, 0)
    out = [f"alone {base * 1e3:.0f} us"]
    for (blocks, threads, lds) in [(16, 256, 16384), (32, 256, 16384), (64, 256, 16384), (64, 512, 32768)]:
        t = run(M, K, N, blocks, threads, lds)
        out.append(f"{blocks}x{threads}/{lds // 1024}K: {t * 1e3:.0f} us ({t / base:.2f}x)")
    again = run(M, K, N, 0, 0, 0)
    out.append(f"alone again {again * 1e3:.0f} us")
    print(f"[corun t160={os.environ.get('W2L_GEMM_T160', '1')}] M={M} K={K} N={N}: " + " | ".join(out), flush=True)
