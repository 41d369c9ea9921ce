"""the LayerNorm-with-images kernels at one geometry against LayerNorm + conversion launch, timed with events;
W2L_LI_ABL (probe build) = timing-only ablations: 1 no transposed image, 2 no row image, 4 no LDS tile, 8 no fp32 result stores
usage: ln_images_one.py groups inner [p]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib, ops
if os.environ.get("W2L_LI_ABL") is not None: _lib.use_probe().__enter__()
groups, inner = int(sys.argv[1]), int(sys.argv[2])
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
L = _lib.lib(); s = torch.cuda.current_stream().cuda_stream
a = torch.relu(torch.randn(groups, inner, device="cuda")); x = torch.randn(groups, inner, device="cuda")
gb = torch.tensor([1.3, -0.2], device="cuda")
r = torch.empty_like(a); y = torch.empty_like(a); mr = torch.empty(2 * groups, device="cuda")
ldR, ldT = (inner + 63) // 64 * 64, (groups + 63) // 64 * 64
rows = torch.zeros(groups, ldR, dtype=torch.bfloat16, device="cuda"); trans = torch.zeros(inner + 1, ldT, dtype=torch.bfloat16, device="cuda")
k = _lib.Bf16ImageSink(rowMajor=rows.data_ptr(), ldRows=ldR, transposed=trans.data_ptr(), ldTrans=ldT)
stats = torch.empty(L.w2l_layernorm_scratch_doubles(groups, inner), dtype=torch.float64, device="cuda")
dy = torch.randn(groups, inner, device="cuda"); dr = torch.empty_like(a); dgb = torch.empty(2, device="cuda")
sums = torch.empty(2 * groups + 64, dtype=torch.float64, device="cuda")
def f_img(): assert L.w2l_residual_layernorm_forward_images(groups, inner, a.data_ptr(), x.data_ptr(), r.data_ptr(), y.data_ptr(), gb.data_ptr(), 1e-5, p, 7, 5, mr.data_ptr(), C.byref(k), s) == 0
def f_plain():
    assert L.w2l_residual_layernorm_forward(groups, inner, a.data_ptr(), x.data_ptr(), r.data_ptr(), y.data_ptr(), gb.data_ptr(), 1e-5, p, 7, 5, stats.data_ptr(), mr.data_ptr(), s) == 0
    assert L.w2l_bf16_convert(y.data_ptr(), groups, inner, inner, rows.data_ptr(), ldR, trans.data_ptr(), ldT, s) == 0
def b_img(): assert L.w2l_layernorm_backward_images(groups, inner, r.data_ptr(), dy.data_ptr(), gb.data_ptr(), mr.data_ptr(), dr.data_ptr(), dgb.data_ptr(), None, None, 1.0, sums.data_ptr(), C.byref(k), p, 9, 6, s) == 0
def b_plain():
    assert L.w2l_layernorm_backward(groups, inner, r.data_ptr(), dy.data_ptr(), gb.data_ptr(), mr.data_ptr(), dr.data_ptr(), dgb.data_ptr(), None, None, 1.0, sums.data_ptr(), s) == 0
    assert L.w2l_bf16_convert_dropout(dr.data_ptr(), groups, inner, inner, rows.data_ptr(), ldR, trans.data_ptr(), ldT, p, 9, 6, s) == 0
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f_plain()
print("LayerNorm %d x %d abl=%s: forward images %.1f us (LayerNorm + convert %.1f), backward images %.1f us (LayerNorm + convert %.1f)" % (
    groups, inner, os.environ.get("W2L_LI_ABL", "-"), timeit(f_img), timeit(f_plain), timeit(b_img), timeit(b_plain)))
