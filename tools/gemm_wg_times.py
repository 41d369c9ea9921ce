"""per-workgroup start / end times of one GEMM launch (probe library, W2L_GEMM_DBG): how far apart do the persistent workers finish?
  W2L_HIP_SO=.../libw2l_hip_probe.so python tools/gemm_wg_times.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
dbg = torch.zeros(4 * 1024, dtype=torch.int64, device="cuda")
os.environ["W2L_GEMM_DBG"] = str(dbg.data_ptr())
from wav2letter_amd import _lib
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
for name, M, K, N, which in [("fc1 fwd", 24000, 800, 2400, "fwd"), ("fc1' dX (KC KC)", 24000, 800, 2400, "dx"), ("fc2 fwd", 24000, 2400, 800, "fwd"), ("fc3 fwd", 6016, 1440, 4320, "fwd"),
                             ("fc3 dW", 6016, 1440, 4320, "dw"), ("4096^3", 4096, 4096, 4096, "fwd")]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda"); y = torch.empty(M, N, device="cuda")
    wt = torch.randn(N, K, device="cuda"); dy = torch.randn(M, N, device="cuda"); dw = torch.empty(K, N, device="cuda")
    def go():
        if which == "fwd": L.w2l_linear_forward(M, K, N, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
        elif which == "dx": L.w2l_linear_backward_data(M, N, K, x.data_ptr(), wt.data_ptr(), y.data_ptr(), 0, None, 1.0, s)
        else: L.w2l_linear_backward_weight(M, K, N, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s)
    for _ in range(20): go()
    torch.cuda.synchronize()
    for rep in range(2):
        dbg.zero_(); go(); torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 4)
        d = d[d[:, 1] > 0]
        t0 = d[:, 0].min(); st = (d[:, 0] - t0) / 100.0; en = (d[:, 1] - t0) / 100.0   # us
        xcc = d[:, 3] & 0xf
        span = en.max()
        cu = (d[:, 3] & 0xf) * 4096 + ((d[:, 2] >> 8) & 0xff)      # (XCC, SE / SH / CU bits of HW_ID)
        cus = np.unique(cu)
        cuEnd = np.array([en[cu == c].max() for c in cus]); cuFirst = np.array([en[cu == c].min() for c in cus]); perCu = np.array([(cu == c).sum() for c in cus])
        print(f"{name:16s} per CU: {len(cus)} CUs ({perCu.min()}-{perCu.max()} workgroups each); last workgroup of a CU ends: min {cuEnd.min():.1f} median {np.median(cuEnd):.1f} max {cuEnd.max():.1f}; "
              f"mean CU idle at the end {np.mean(span - cuEnd):.1f} us = {np.mean(span - cuEnd) / span * 100:.1f} %; a CU runs ONE workgroup for {np.mean(cuEnd - cuFirst):.1f} us on average "
              f"= {np.mean(cuEnd - cuFirst) / span * 100:.1f} % of the launch", flush=True)
        print(f"{name:16s} M={M} K={K} N={N}: {len(d)} workgroups, kernel span {span:.1f} us; start: max {st.max():.1f} us; end: min {en.min():.1f} median {np.median(en):.1f} "
              f"p90 {np.percentile(en, 90):.1f} max {en.max():.1f}; mean idle at the end {np.mean(span - en):.1f} us = {np.mean(span - en) / span * 100:.1f} % | "
              f"per-XCD median end: " + " ".join(f"{np.median(en[xcc == q]):.0f}" for q in range(8)), flush=True)
