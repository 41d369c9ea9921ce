"""the shader clock and socket power the part holds while one GEMM shape runs back to back for a few seconds (rocm-smi polled from a
thread), next to the TFLOP/s of that shape: is the in-step deficit against 4096^3 a clock / power effect?
  python tools/gemm_clock.py"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream

samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", out)
            pw = re.search(r"Power \(W\):\s*([\d.]+)", out)
            samples.append((time.time(), int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:
            samples.append((time.time(), -2, -2.0))
        time.sleep(0.05)

th = threading.Thread(target=poll, daemon=True); th.start()
time.sleep(1.0)
idle = list(samples)
print("idle:", idle[-3:], flush=True)
for name, M, K, N in [("fc1 fwd", 24000, 800, 2400), ("fc1 dX-like (KC KC)", 24000, 800, 2400), ("fc3 fwd", 6016, 1440, 4320), ("4096^3", 4096, 4096, 4096), ("8192^3", 8192, 8192, 8192)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    wt = torch.randn(N, K, device="cuda")
    kc = "KC KC" in name
    def go():
        if kc: L.w2l_linear_backward_data(M, N, K, x.data_ptr(), wt.data_ptr(), y.data_ptr(), 0, None, 1.0, s)   # y[M][N] = x[M][K] wt[N][K]^T
        else: L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
    for _ in range(5): go()
    torch.cuda.synchronize()
    n0 = len(samples); t0 = time.time(); it = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(50): go()
        it += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    ss = [q for q in samples[n0:] if q[1] > 0]
    clk = sorted(q[1] for q in ss); pw = sorted(q[2] for q in ss)
    med = lambda v: v[len(v) // 2] if v else -1
    print(f"{name:22s} M={M} K={K} N={N}: {ms * 1e3:.0f} us {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s | sclk median {med(clk)} MHz (min {clk[0] if clk else -1}, max {clk[-1] if clk else -1}), power median {med(pw)} W, {len(ss)} samples", flush=True)
stop = True
