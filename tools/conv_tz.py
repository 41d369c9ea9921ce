"""Block-Toeplitz TDS convolution kernels (conv_tds_tz.hpp / conv_tds_tzf.hpp, the product path for C = 10 / 14 / 18 with
H % 16 == 0) against a float64 reference on small / ragged shapes (forward + bias + ReLU, backward-data + addend,
backward-filter + bias gradient), then the three TDS stages of am_tds_ctc.arch at B = 32 with hipEvent timings next to
the previous generation (probe library, W2L_TDS_TZ_OFF=1 / W2L_TDS_TZF_OFF=1) and the per-round cycle counters of wave 0.
    python tools/conv_tz.py [--small] [--abl]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from wav2letter_amd import _lib
from tools.conv_rs import run


def ref64(x, w, b, dy, add, kw, padl):
    xr = x.double().permute(0, 3, 2, 1).requires_grad_(True)          # [B][C][H][T]
    wr = w.double().permute(2, 1, 0)[:, :, None, :].clone().requires_grad_(True)
    bb = b.double().clone().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (padl, kw - 1 - padl)), wr, bb)
    yr.backward(dy.double().permute(0, 3, 2, 1))
    dw = wr.grad[:, :, 0, :].permute(2, 1, 0)                          # [kw][ci][co]
    return torch.relu(yr).permute(0, 3, 2, 1).detach(), xr.grad.permute(0, 3, 2, 1) + add.double(), dw, bb.grad


def small():
    s = torch.cuda.current_stream().cuda_stream
    L = _lib.lib()
    worst = 0.0
    cases = [(10, 48, 2, 80), (18, 12, 2, 80), (14, 24, 2, 80), (10, 50, 2, 16), (18, 15, 3, 32), (14, 77, 2, 80), (10, 1, 1, 80), (18, 2, 2, 16),
             (10, 129, 3, 80), (18, 188, 5, 80), (10, 750, 3, 80), (14, 375, 3, 80), (14, 33, 5, 16), (18, 97, 33, 80)]
    for (Cc, T, B, H) in cases:
        for kw, padl in ((21, 10), (21, 20), (9, 0)):
            d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, padl, kw - 1 - padl)
            g = torch.Generator(device="cpu").manual_seed(Cc * 100 + T + kw)
            x = torch.randn(B, T, H, Cc, generator=g).cuda()
            w = (torch.randn(kw, Cc, Cc, generator=g) / (kw * Cc) ** 0.5).cuda()
            b = torch.randn(Cc, generator=g).cuda()
            dy = torch.randn(B, T, H, Cc, generator=g).cuda()
            add = torch.randn(B, T, H, Cc, generator=g).cuda()
            y = torch.full_like(x, float("nan")); dx = torch.full_like(x, float("nan"))
            dw = torch.full_like(w, float("nan")); db = torch.full_like(b, float("nan"))
            assert L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s) == 0
            assert L.w2l_conv_backward_data_add(C.byref(d), dy.data_ptr(), w.data_ptr(), add.data_ptr(), dx.data_ptr(), s) == 0
            assert L.w2l_conv_backward_filter(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), s) == 0
            torch.cuda.synchronize()
            y2 = torch.full_like(x, float("nan"))
            L.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y2.data_ptr(), 1, s)
            torch.cuda.synchronize()
            ry, rdx, rdw, rdb = ref64(x, w, b, dy, add, kw, padl)
            e = [((a.double() - r.cuda()).abs().max() / r.abs().max()).item() for a, r in ((y, ry), (dx, rdx), (dw, rdw), (db, rdb))]
            bad = any(not (v < 1e-4) for v in e)
            worst = max([worst] + e) if not any(v != v for v in e) else float("nan")
            print(f"C={Cc} T={T} B={B} H={H} kw={kw} padl={padl}: y {e[0]:.1e} dx+add {e[1]:.1e} dw {e[2]:.1e} db {e[3]:.1e}  deterministic {torch.equal(y, y2)}" +
                  ("   <<<<<< BAD" if bad else ""))
    print("worst", worst)


def counters(P, d, x, w, b, y, s, name):
    dbg = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
    os.environ[name] = str(dbg.data_ptr())
    try:
        P.w2l_conv_forward(C.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, s)
        torch.cuda.synchronize()
    finally:
        os.environ.pop(name)
    return dbg.view(512, 16)


def big():
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(20):
        a @ a
    torch.cuda.synchronize()
    prod = _lib.lib()
    total = {"new": 0.0, "old": 0.0}
    for (Cc, T, nblk) in [(10, 750, 5), (14, 375, 6), (18, 188, 10)]:
        B, H, kw = 32, 80, 21
        d = _lib.ConvDesc(B, T, H, Cc, Cc, kw, 1, 10, 10)
        x = torch.randn(B, T, H, Cc, device="cuda")
        w = torch.randn(kw, Cc, Cc, device="cuda") / (kw * Cc) ** 0.5
        b = torch.randn(Cc, device="cuda")
        dy = torch.randn(B, T, H, Cc, device="cuda")
        os.environ["W2L_TDS_TZ_OFF"] = "1"
        os.environ["W2L_TDS_TZF_OFF"] = "1"
        with _lib.use_probe() as P:
            old, to = run(P, d, x, w, b, dy)
        os.environ.pop("W2L_TDS_TZ_OFF"); os.environ.pop("W2L_TDS_TZF_OFF")
        new, tn = run(prod, d, x, w, b, dy)
        e_ab = {k: ((new[k].double() - old[k].double()).abs().max() / old[k].double().abs().max()).item() for k in ("y", "dx", "dw", "db")}
        flops = 2.0 * B * T * H * kw * Cc * Cc
        print(f"C={Cc} T={T} B={B}: tz vs previous generation " + " ".join(f"{k} {v:.1e}" for k, v in e_ab.items()))
        for k in ("fwd", "bwd_data", "bwd_filter"):
            print(f"    {k:10s} tz {tn[k]:8.1f} us = {flops / tn[k] / 1e6:6.1f} TF/s = {flops / tn[k] / 1e6 / 157.3:.3f}   previous {to[k]:8.1f} us = {flops / to[k] / 1e6:6.1f} TF/s")
            total["new"] += nblk * tn[k]; total["old"] += nblk * to[k]
        with _lib.use_probe() as P:
            s = torch.cuda.current_stream().cuda_stream
            yy = torch.empty_like(dy)
            t = counters(P, d, x, w, b, yy, s, "W2L_TDS_TZ_DBG")
        raw = t.cpu()
        t = t.double()
        nn = t[:, 5].clamp(min=1)
        live = t[:, 5] > 0
        names = ["stage issue + set-up", "chain", "DMA wait", "epilogue", "barrier"]
        print("    forward, cycles per round of wave 0 (mean over workgroups): " +
              "  ".join(f"{nm} {(t[live, i] / nn[live]).mean().item():.0f}" for i, nm in enumerate(names)) +
              f"  rounds/WG {nn[live].mean().item():.2f} (max {nn.max().item():.0f}, {int(live.sum())} workgroups)")
        # residency: which CU (XCC, SE, CU) each workgroup ran on and when (100 MHz wall clock)
        lv = raw[raw[:, 5] > 0]
        hw, xcc = lv[:, 6] & 0xffffffff, (lv[:, 6] >> 32) & 0xf
        cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
        t0 = int(lv[:, 7].min())
        ent, loop, ex = (lv[:, 7] - t0).double() / 100, (lv[:, 8] - t0).double() / 100, (lv[:, 9] - t0).double() / 100
        ncu = len(set(cu.tolist()))
        second = ent > ent.median()
        print(f"    residency: {len(lv)} workgroups on {ncu} distinct CUs; entry {ent.min().item():.1f}..{ent.max().item():.1f} us (median {ent.median().item():.1f}), "
              f"set-up (entry -> first round) mean {(loop - ent).mean().item():.2f} us, lifetime mean {(ex - ent).mean().item():.1f} us, last exit {ex.max().item():.1f} us; "
              f"workgroups entering after the median entry: {int(second.sum())}")
        # how many workgroups are alive on a CU at the midpoint of each workgroup's life
        mid = (ent + ex) / 2
        alive = [(int(((cu == cu[i]) & (ent <= mid[i]) & (ex >= mid[i])).sum())) for i in range(len(lv))]
        print(f"    co-resident workgroups per CU at a workgroup's midpoint: mean {sum(alive) / len(alive):.2f}, max {max(alive)}")
        for wgs in (128, 256, 384, 512, 768, 1024):
            os.environ["W2L_TDS_TZ_WGS"] = str(wgs)
            with _lib.use_probe() as P:
                _, ta = run(P, d, x, w, b, dy, reps=10)
            os.environ.pop("W2L_TDS_TZ_WGS")
            print(f"    at most {wgs:4d} workgroups: fwd {ta['fwd']:7.1f} us  bwd-data {ta['bwd_data']:7.1f} us")
        if "--abl" in sys.argv:
            for abl, what in [(1, "one MFMA per chain"), (2, "no fragment reads"), (4, "no stores"), (8, "no DMA"), (12, "no DMA, no stores"), (14, "MFMA chain only")]:
                os.environ["W2L_TDS_RS_ABL"] = str(abl)
                with _lib.use_probe() as P:
                    _, ta = run(P, d, x, w, b, dy, reps=10)
                os.environ.pop("W2L_TDS_RS_ABL")
                print(f"    abl {abl:2d} ({what:20s}): fwd {ta['fwd']:7.1f} us")
    fl = 3 * 2.0 * 32 * 80 * 21 * (5 * 750 * 100 + 6 * 375 * 196 + 10 * 188 * 324)
    for k in total:
        print(f"TDS convolutions of one step ({k}): {total[k] / 1e3:.2f} ms = {fl / total[k] / 1e6:.1f} TF/s = {fl / total[k] / 1e6 / 157.3:.3f} of the fp32 MFMA peak")


if __name__ == "__main__":
    if "--big-only" not in sys.argv:
        small()
    if "--small" not in sys.argv:
        big()
