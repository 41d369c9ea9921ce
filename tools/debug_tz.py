"""error map of one convolution case through the library against float64 (debugging aid)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wav2letter_amd import _lib
from oracle import tds_tz_model as M
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream
for (B, CI, CO, H, T, kw, stride, padl, padr) in [(2, 10, 14, 32, 61, 21, 2, 10, 10), (2, 14, 18, 16, 60, 21, 2, 10, 10), (1, 10, 14, 16, 41, 21, 2, 10, 10)]:
    rng = np.random.default_rng(1)
    To = (T + padl + padr - kw) // stride + 1
    x = rng.normal(size=(B, T, H, CI)).astype(np.float32); w = rng.normal(size=(kw, CI, CO)).astype(np.float32); b = rng.normal(size=CO).astype(np.float32)
    d = _lib.ConvDesc(B, T, H, CI, CO, kw, stride, padl, padr)
    xd, wd, bd = torch.tensor(x).cuda(), torch.tensor(w).cuda(), torch.tensor(b).cuda()
    y = torch.full((B, To, H, CO), float("nan"), device="cuda")
    assert L.w2l_conv_forward(C.byref(d), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), 0, s) == 0
    torch.cuda.synchronize()
    ref = M.direct(x, w, b, kw, padl, False, False, None, To, stride=stride)
    e = np.abs(y.cpu().numpy() - ref)
    print(f"fwd {CI}->{CO} T={T} H={H}: max err {np.nanmax(e):.2e}, nan {np.isnan(e).sum()}")
    bad = e > 1e-3
    print("  bad per t:", bad.sum(axis=(0, 2, 3)).tolist())
    print("  bad per h:", bad.sum(axis=(0, 1, 3)).tolist())
    print("  bad per co:", bad.sum(axis=(0, 1, 2)).tolist())
    print("  bad per b:", bad.sum(axis=(1, 2, 3)).tolist())
    dy = rng.normal(size=(B, To, H, CO)).astype(np.float32)
    dx = torch.full((B, T, H, CI), float("nan"), device="cuda")
    assert L.w2l_conv_backward_data(C.byref(d), torch.tensor(dy).cuda().data_ptr(), wd.data_ptr(), dx.data_ptr(), 0, s) == 0
    torch.cuda.synchronize()
    rd = M.direct_backward_data(dy, w, T, kw, stride, padl)
    e = np.abs(dx.cpu().numpy() - rd)
    print(f"bwd-data: max err {np.nanmax(e):.2e}, nan {np.isnan(e).sum()}; bad per t {(e > 1e-3).sum(axis=(0, 2, 3)).tolist()}")
