"""TDS convolution probe: the three TDS stages of am_tds_ctc.arch (C = 10 / 14 / 18, kw = 21, H = 80, B = 32) forward,
backward-data, backward-filter through the C ABI: float64 torch reference on a slice, A/B against the previous kernels
(probe library, W2L_TDS_RS_OFF=1), hipEvent timings after a burn-in.   python tools/conv_rs.py [--quick]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from wav2letter_amd import _lib


def run(L, d, x, w, b, dy, reps=20):
    s = torch.cuda.current_stream().cuda_stream
    y = torch.empty_like(dy)
    dx = torch.empty_like(x)
    dw = torch.empty_like(w)
    db = torch.empty_like(b)
    fns = {
        "fwd": lambda: L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s),
        "bwd_data": lambda: L.w2l_conv_backward_data(C.byref(d), dy.data_ptr(), w.data_ptr(), dx.data_ptr(), 0, s),
        "bwd_filter": lambda: L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s),
    }
    times = {}
    for k, f in fns.items():
        for _ in range(3):
            st = f()
            assert st == 0, (k, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        times[k] = e0.elapsed_time(e1) / reps * 1e3
    return dict(y=y, dx=dx, dw=dw, db=db), times


def main():
    quick = "--quick" in sys.argv
    torch.manual_seed(0)
    # burn-in: clocks ramp during the first kernels of a process
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    prod = _lib.lib()
    for (Cc, T) in [(10, 750), (14, 375), (18, 188)]:
        B, H, kw = (4 if quick else 32), 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        x = torch.randn(B, T, H, Cc, device="cuda")
        w = torch.randn(kw, Cc, Cc, device="cuda") / (kw * Cc) ** 0.5
        b = torch.randn(Cc, device="cuda")
        dy = torch.randn(B, T, H, Cc, device="cuda")
        new, tn = run(prod, d, x, w, b, dy)
        # float64 reference on utterance 0: conv over time with kernel [co][ci][kw]
        xr = x[0].double().permute(2, 1, 0)[None]                      # [1][C][H][T]
        wr = w.double().permute(2, 1, 0)[:, :, None, :]                # [co][ci][1][kw]
        xr.requires_grad_(True)
        yr = F.conv2d(F.pad(xr, (10, 10)), wr, b.double())
        ref_y = torch.relu(yr)[0].permute(2, 1, 0)
        e_y = ((new["y"][0].double() - ref_y).abs().max() / ref_y.abs().max()).item()
        yr.backward(dy[0].double().permute(2, 1, 0)[None])
        ref_dx = xr.grad[0].permute(2, 1, 0)
        e_dx = ((new["dx"][0].double() - ref_dx).abs().max() / ref_dx.abs().max()).item()
        os.environ["W2L_TDS_RS_OFF"] = "1"
        with _lib.use_probe() as P:
            old, to = run(P, d, x, w, b, dy)
        os.environ.pop("W2L_TDS_RS_OFF")
        e_ab = {k: ((new[k].double() - old[k].double()).abs().max() / old[k].double().abs().max()).item() for k in new}
        flops = 2.0 * B * T * H * kw * Cc * Cc
        print(f"C={Cc} T={T} B={B}: vs fp64 y {e_y:.2e} dx {e_dx:.2e} | vs previous kernels " +
              " ".join(f"{k} {v:.1e}" for k, v in e_ab.items()))
        for k in tn:
            print(f"    {k:10s} new {tn[k]:8.1f} us = {flops / tn[k] / 1e6:6.1f} TF/s   previous {to[k]:8.1f} us = {flops / to[k] / 1e6:6.1f} TF/s")
        # determinism of the overlap-add
        again, _ = run(prod, d, x, w, b, dy, reps=1)
        print("    run-to-run identical:", all(torch.equal(new[k], again[k]) for k in ("y", "dx")))
        if "--stagger" in sys.argv:
            for st in (0, 1, 2, 3, 4, 6, 8):
                os.environ["W2L_TDS_RS_STAGGER"] = str(st)
                with _lib.use_probe() as P:
                    _, ta = run(P, d, x, w, b, dy, reps=10)
                os.environ.pop("W2L_TDS_RS_STAGGER")
                print(f"    stagger {st}: fwd {ta['fwd']:7.1f} us  bwd_data {ta['bwd_data']:7.1f} us")
        if "--abl" in sys.argv:
            # timing-only ablations of the role-swapped kernel (probe library): where does the time go?
            for abl, what in [(0, "full"), (1, "no MFMAs"), (2, "no overlap-add"), (4, "no epilogue"), (8, "no staging"),
                              (32, "no zero fill"), (3, "no MFMA+add"), (12, "no staging+epilogue"), (47, "launch + tile loop only")]:
                os.environ["W2L_TDS_RS_ABL"] = str(abl)
                with _lib.use_probe() as P:
                    _, ta = run(P, d, x, w, b, dy, reps=10)
                os.environ.pop("W2L_TDS_RS_ABL")
                print(f"    abl {abl:2d} ({what:22s}): fwd {ta['fwd']:7.1f} us")


def small_shapes():
    """the shapes of the reduced full-network test (T = 96 -> 48 / 24 / 12 frames, B = 2): every element of y, dx (with
    an addend, as the TDS block's backward calls it) against the previous kernels and a float64 reference"""
    prod = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    for (Cc, T) in [(10, 48), (14, 24), (18, 12), (10, 50), (18, 15), (14, 77), (10, 1), (18, 2)]:
        B, H, kw = 2, 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        g = torch.Generator(device="cpu").manual_seed(Cc * 100 + T)
        x = torch.randn(B, T, H, Cc, generator=g).cuda()
        w = (torch.randn(kw, Cc, Cc, generator=g) / (kw * Cc) ** 0.5).cuda()
        b = torch.randn(Cc, generator=g).cuda()
        dy = torch.randn(B, T, H, Cc, generator=g).cuda()
        add = torch.randn(B, T, H, Cc, generator=g).cuda()
        outs = {}
        for name, L in (("new", prod), ("old", None)):
            def go(L):
                y = torch.empty_like(x); dx = torch.empty_like(x)
                assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
                assert L.w2l_conv_backward_data_add(C.byref(d), dy.data_ptr(), w.data_ptr(), add.data_ptr(), dx.data_ptr(), s) == 0
                torch.cuda.synchronize()
                return y, dx
            if L is None:
                os.environ["W2L_TDS_RS_OFF"] = "1"
                with _lib.use_probe() as P:
                    outs[name] = go(P)
                os.environ.pop("W2L_TDS_RS_OFF")
            else:
                outs[name] = go(L)
        xr = x.double().permute(0, 3, 2, 1).requires_grad_(True)          # [B][C][H][T]
        wr = w.double().permute(2, 1, 0)[:, :, None, :]
        yr = F.conv2d(F.pad(xr, (10, 10)), wr, b.double())
        yr.backward(dy.double().permute(0, 3, 2, 1))
        ref_y = torch.relu(yr).permute(0, 3, 2, 1)
        ref_dx = xr.grad.permute(0, 3, 2, 1) + add.double()
        def e(a, r):
            return ((a.double() - r).abs().max() / r.abs().max()).item()
        print(f"small C={Cc} T={T}: new vs fp64 y {e(outs['new'][0], ref_y):.1e} dx+add {e(outs['new'][1], ref_dx):.1e} | "
              f"old vs fp64 y {e(outs['old'][0], ref_y):.1e} dx+add {e(outs['old'][1], ref_dx):.1e}")


if __name__ == "__main__":
    if "--small" in sys.argv:
        small_shapes()
    else:
        main()
