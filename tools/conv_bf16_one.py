"""the bf16 convolution kernels of the config-3 step, one shape at a time, with timing ablations (probe library):
  python tools/conv_bf16_one.py [abl ...]     abl bits: 1 no staging, 2 no MFMAs, 4 no stores (forward / backward-data only)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib, ops

SHAPES = [  # (Cin, Cout, T, kw, stride, padl, padr) at B = 64, H = 80
    (15, 15, 750, 9, 1, 7, 1), (19, 19, 375, 9, 1, 7, 1), (23, 23, 188, 11, 1, 9, 1), (27, 27, 188, 11, 1, 10, 0),
    (15, 19, 750, 10, 2, 7, 1), (19, 23, 375, 12, 2, 9, 1), (23, 27, 188, 11, 1, 10, 0)]
B, H = 64, 80
abls = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


with _lib.use_probe():
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for (ci, co, T, kw, st, pl, pr) in SHAPES:
        x = torch.randn(B, T, H, ci, device="cuda")
        w = torch.randn(kw, ci, co, device="cuda") * 0.05
        os.environ.pop("W2L_TBF_ABL", None)
        y, imgs, d = ops.tds_conv_bf16(x, w, None, pl, pr, stride=st)
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        mb = (x.numel() + y.numel()) * 4 / 1e6
        for abl in abls:
            os.environ["W2L_TBF_ABL"] = str(abl)
            tf = timed(lambda: L.w2l_tds_conv_bf16_forward(C.byref(d), x.data_ptr(), imgs[0].data_ptr(), None, y.data_ptr(), 0, s))
            tb = timed(lambda: L.w2l_tds_conv_bf16_backward_data(C.byref(d), dy.data_ptr(), imgs[1].data_ptr(), x.data_ptr(), dx.data_ptr(), s))
            tw = timed(lambda: L.w2l_tds_conv_bf16_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), s))
            print(f"[conv_bf16] {ci:2d}->{co:2d} T={T:3d} kw={kw:2d} s={st} abl={abl}: fwd {tf:6.1f} us ({mb / tf:5.2f} TB/s)  bwd-data {tb:6.1f} us"
                  f" ({(mb + x.numel() * 4 / 1e6) / tb:5.2f} TB/s)  filter {tw:6.1f} us ({mb / tw:5.2f} TB/s)", flush=True)
