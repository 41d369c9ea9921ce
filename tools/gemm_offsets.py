"""Same GEMM, same layout, buffers shifted by byte offsets: is the K=800 slowdown a placement (channel aliasing) effect?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
from wav2letter_amd.ops import _p, _s, check

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

L = _lib.lib()
M, N, K = 24000, 2400, 800
PAD = 64 << 20
pool = torch.empty((M * K + K * N + M * N) + 3 * PAD // 4 + 1024, device="cuda")
pool.normal_()
print(f"[offsets] pool base {pool.data_ptr():#x}", flush=True)

def view(off_bytes, n):
    o = off_bytes // 4
    return pool[o:o + n]

def run(aoff, boff, coff, bkc):
    a = view(aoff, M * K)
    b = view(PAD + M * K * 4 + boff, K * N)
    c = view(2 * PAD + (M * K + K * N) * 4 + coff, M * N)
    ldb = K if bkc else N
    t = timeit(lambda: check(L.w2l_gemm_f32(M, N, K, _p(a), K, 1, _p(b), ldb, int(bkc), _p(c), N, None, 0, 1, _s()), "gemm"))
    return 2.0 * M * N * K / t / 1e9

for bkc in (True, False):
    for (ao, bo, co) in [(0, 0, 0), (0, 0, 0), (128, 128, 128), (4096, 4096, 4096), (1 << 20, 1 << 20, 1 << 20), (0, 0, 0), (128, 0, 0), (4096, 0, 0), (65536, 0, 0), (1 << 20, 0, 0), (0, 128, 0), (0, 4096, 0), (0, 65536, 0),
                         (0, 1 << 20, 0), (0, 0, 128), (0, 0, 4096), (0, 0, 65536), (0, 0, 1 << 20), (0, 0, 3 << 20), (2 << 20, 5 << 20, 7 << 20)]:
        print(f"[offsets] bkc={int(bkc)} a+{ao} b+{bo} c+{co}: {run(ao, bo, co, bkc):6.1f} TF/s", flush=True)
