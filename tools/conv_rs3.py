"""Third-generation TDS convolution kernel (conv_tds_rs3.hpp, the product path for C = 10 / 18) against the previous
generation (probe library, W2L_TDS_RS3_OFF=1) and a float64 reference: small / ragged shapes element by element (forward + ReLU + bias, backward-data + addend),
then the three TDS stages of am_tds_ctc.arch at B = 32 with hipEvent timings.   python tools/conv_rs3.py [--abl]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from wav2letter_amd import _lib
from tools.conv_rs import run


def ref64(x, w, b, dy, add):
    xr = x.double().permute(0, 3, 2, 1).requires_grad_(True)          # [B][C][H][T]
    wr = w.double().permute(2, 1, 0)[:, :, None, :]
    yr = F.conv2d(F.pad(xr, (10, 10)), wr, b.double())
    yr.backward(dy.double().permute(0, 3, 2, 1))
    return torch.relu(yr).permute(0, 3, 2, 1), xr.grad.permute(0, 3, 2, 1) + add.double()


def small():
    s = torch.cuda.current_stream().cuda_stream
    worst = 0.0
    for (Cc, T, B) in [(10, 48, 2), (18, 12, 2), (14, 24, 2), (10, 50, 2), (18, 15, 3), (14, 77, 2), (10, 1, 1), (18, 2, 2), (10, 129, 3),
                       (18, 188, 5), (10, 750, 3), (14, 375, 3), (10, 64, 1), (18, 46, 7), (10, 331, 9), (18, 97, 33)]:
        H, kw = 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        g = torch.Generator(device="cpu").manual_seed(Cc * 100 + T)
        x = torch.randn(B, T, H, Cc, generator=g).cuda()
        w = (torch.randn(kw, Cc, Cc, generator=g) / (kw * Cc) ** 0.5).cuda()
        b = torch.randn(Cc, generator=g).cuda()
        dy = torch.randn(B, T, H, Cc, generator=g).cuda()
        add = torch.randn(B, T, H, Cc, generator=g).cuda()
        os.environ["W2L_TDS_RS3"] = "1"
        with _lib.use_probe() as P:
            y = torch.full_like(x, float("nan")); dx = torch.full_like(x, float("nan"))
            assert P.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
            assert P.w2l_conv_backward_data_add(C.byref(d), dy.data_ptr(), w.data_ptr(), add.data_ptr(), dx.data_ptr(), s) == 0
            torch.cuda.synchronize()
            y2 = torch.full_like(x, float("nan"))
            P.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y2.data_ptr(), 1, s)
            torch.cuda.synchronize()
        os.environ.pop("W2L_TDS_RS3")
        ry, rdx = ref64(x, w, b, dy, add)
        ey = ((y.double() - ry).abs().max() / ry.abs().max()).item()
        edx = ((dx.double() - rdx).abs().max() / rdx.abs().max()).item()
        worst = max(worst, ey, edx) if ey == ey and edx == edx else float("nan")
        print(f"rs3 C={Cc} T={T} B={B}: vs fp64 y {ey:.1e} dx+add {edx:.1e}  deterministic {torch.equal(y, y2)}")
    print("worst", worst)


def big():
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    prod = _lib.lib()
    for (Cc, T) in [(10, 750), (14, 375), (18, 188)]:
        B, H, kw = 32, 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        x = torch.randn(B, T, H, Cc, device="cuda")
        w = torch.randn(kw, Cc, Cc, device="cuda") / (kw * Cc) ** 0.5
        b = torch.randn(Cc, device="cuda")
        dy = torch.randn(B, T, H, Cc, device="cuda")
        os.environ["W2L_TDS_RS3_OFF"] = "1"
        os.environ["W2L_TDS_RSF3_OFF"] = "1"
        with _lib.use_probe() as P:
            old, to = run(P, d, x, w, b, dy)
        os.environ.pop("W2L_TDS_RS3_OFF"); os.environ.pop("W2L_TDS_RSF3_OFF")
        os.environ["W2L_TDS_RS3"] = "1"
        with _lib.use_probe() as P:
            new, tn = run(P, d, x, w, b, dy)
        e_ab = {k: ((new[k].double() - old[k].double()).abs().max() / old[k].double().abs().max()).item() for k in ("y", "dx", "dw", "db")}
        flops = 2.0 * B * T * H * kw * Cc * Cc
        print(f"C={Cc} T={T} B={B}: rs3 vs previous generation " + " ".join(f"{k} {v:.1e}" for k, v in e_ab.items()))
        for k in ("fwd", "bwd_data", "bwd_filter"):
            print(f"    {k:10s} rs3 {tn[k]:8.1f} us = {flops / tn[k] / 1e6:6.1f} TF/s   previous {to[k]:8.1f} us = {flops / to[k] / 1e6:6.1f} TF/s")
        dbg = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
        os.environ["W2L_TDS_RS3_DBG"] = str(dbg.data_ptr())
        with _lib.use_probe() as P:
            s = torch.cuda.current_stream().cuda_stream
            yy = torch.empty_like(dy)
            P.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), yy.data_ptr(), 1, s)
            torch.cuda.synchronize()
        os.environ.pop("W2L_TDS_RS3_DBG")
        t = dbg.view(256, 8).double()
        nn = t[:, 6].clamp(min=1)
        names = ["consumer work", "consumer wait", "mover stage", "mover fetch", "mover epilogue", "mover wait", "tiles", "mover vmcnt(0)"]
        print("    cycles per round (mean over workgroups; rounds with a tile): " +
              "  ".join(f"{nm} {(t[:, i] / nn).mean().item():.0f}" for i, nm in enumerate(names)) + f"  tiles/WG {nn.mean().item():.1f}")
        if "--abl" in sys.argv:
            if Cc == 14:
                continue
            for abl, what in [(2, "no overlap-add"), (32, "no fragment reads"), (34, "MFMA + movers"), (28, "consumers + epilogue LDS"),
                              (92, "consumers only"), (94, "MFMA + fragment reads only"), (126, "MFMA only"), (127, "barriers + loop only"),
                              (35, "movers only")]:
                os.environ["W2L_TDS_RS_ABL"] = str(abl)
                with _lib.use_probe() as P:
                    _, ta = run(P, d, x, w, b, dy, reps=10)
                os.environ.pop("W2L_TDS_RS_ABL")
                print(f"    abl {abl:2d} ({what:26s}): fwd {ta['fwd']:7.1f} us")
        os.environ.pop("W2L_TDS_RS3")


if __name__ == "__main__" and "--scale" not in sys.argv:
    small()
    if "--small" not in sys.argv:
        big()


def scale():
    """time against the amount of work (B): slope = per-round cost, intercept = launch + set-up"""
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    os.environ["W2L_TDS_RS3"] = "1"
    for (Cc, T) in [(10, 750), (18, 188)]:
        for abl in (0, 126, 127, 35):
            row = []
            for B in (8, 16, 32, 64, 128):
                H, kw = 80, 21
                d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
                x = torch.randn(B, T, H, Cc, device="cuda")
                w = torch.randn(kw, Cc, Cc, device="cuda") / (kw * Cc) ** 0.5
                b = torch.randn(Cc, device="cuda")
                dy = torch.randn(B, T, H, Cc, device="cuda")
                os.environ["W2L_TDS_RS_ABL"] = str(abl)
                with _lib.use_probe() as P:
                    _, ta = run(P, d, x, w, b, dy, reps=10)
                row.append(ta["fwd"])
            print(f"C={Cc} abl {abl:3d}: " + "  ".join(f"B={B}: {t:7.1f}" for B, t in zip((8, 16, 32, 64, 128), row)))
    os.environ.pop("W2L_TDS_RS_ABL")


if __name__ == "__main__" and "--scale" in sys.argv:
    scale()
