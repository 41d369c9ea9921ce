"""The ASG alpha recursion at the north-star stress shape (B = 32, N = 9998): forward with the step epilogue folded into the
streaming kernel (default) against the separate fcc_big_step launch (probe library, W2L_FCC_FOLD=0): same losses bit for bit,
time per step.   python tools/fcc_fold.py [T]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import _lib
from wav2letter_amd.criterion import CriterionScaleMode, FullConnectionCriterion

T = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B, N = 32, 9998
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B, T, N, generator=g).cuda()
tgt = torch.zeros(B, 4, dtype=torch.int32).cuda()
A = (torch.randn(N, N, generator=g) * 0.1 + 4 * torch.eye(N)).cuda()


def run(reps=3):
    crit = FullConnectionCriterion(N, CriterionScaleMode.TARGET_SZ_SQRT).cuda()
    crit.transitions.data = A
    with torch.no_grad():
        loss = crit(x, tgt)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            loss = crit(x, tgt)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    return loss.clone(), min(ts)


res = {}
for fold in ("1", "0", "1", "0"):
    os.environ["W2L_FCC_FOLD"] = fold
    with _lib.use_probe():
        loss, dt = run()
    res.setdefault(fold, []).append((loss, dt))
    print(f"W2L_FCC_FOLD={fold}: forward {dt * 1e3:8.2f} ms = {dt * 1e6 / (T - 1):6.2f} us per step = "
          f"{(4.0 * N * N + 8.0 * B * N) * (T - 1) / dt / 1e9:7.1f} GB/s ({(4.0 * N * N + 8.0 * B * N) * (T - 1) / dt / 8e12:.3f} of 8 TB/s)", flush=True)
print("losses identical (folded vs separate step kernel):", torch.equal(res["1"][0][0], res["0"][0][0]),
      "| run to run (folded):", torch.equal(res["1"][0][0], res["1"][1][0]))
prod, dtp = run()
print(f"product library: forward {dtp * 1e3:8.2f} ms = {dtp * 1e6 / (T - 1):6.2f} us per step; equals the probe's folded run:", torch.equal(prod, res["1"][0][0]))
