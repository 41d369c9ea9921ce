"""BASELINE config C4 on one GPU: conv_glu LibriSpeech (17 WN-conv + GLU layers, 208.9 M parameters) with the ASG
criterion, N = 30 tokens, T = 2000 frames of 40 filterbanks, batch 64: one full training step (forward, ASG forward /
backward, backward, clip + SGD).  Prints ms/step, utterances/s and the GEMM throughput of the step.
  python tools/c4_step.py [steps] [batch]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wav2letter_amd import CriterionScaleMode, _lib, recipes
from wav2letter_amd.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T, nfeat, nlabel, Lmax = 2000, 40, 30, 300
fl = recipes.CONV_GLU_FLAGS
tr = Trainer(recipes.conv_glu_librispeech_arch(), nfeat, nlabel, "asg", CriterionScaleMode.TARGET_SZ_SQRT, transdiag=fl["transdiag"])
tr.init_params(seed=1)
Tout = tr.plan(B, T, Lmax)
tr.to_device()
g = torch.Generator().manual_seed(4)
x = torch.randn(B, nfeat, T, generator=g).cuda()
tgt = torch.full((B, Lmax), -1, dtype=torch.int32)
for b in range(B):
    l = int(torch.randint(60, Lmax + 1, (1,), generator=g))
    y = torch.randint(0, 28, (l,), generator=g, dtype=torch.int32)
    for i in range(1, l):
        if y[i] == y[i - 1]:
            y[i] = (y[i] + 1) % 28
    tgt[b, :l] = y
tgt = tgt.cuda()

def step():
    loss = tr.forward_backward(x, tgt)
    tr.update(lr=fl["lr"], lrcrit=fl["lrcrit"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
    return loss

step()
torch.cuda.synchronize()
L = _lib.lib()
L.w2l_profile_enable(1)
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
L.w2l_profile_report_kind(0, C.byref(n_), C.byref(ms_), C.byref(w_))
L.w2l_profile_enable(0)
out = {"config": f"C4 conv_glu LibriSpeech ASG: B={B}, T={T}, 40 fbank, N=30, fp32, Tout={Tout}", "ms_per_step": round(dt * 1e3, 1),
       "utterances_per_sec": round(B / dt, 2), "loss_mean": float(loss.float().mean().item()),
       "gemm": {"launches_per_step": n_.value // steps, "ms_per_step": round(ms_.value / steps, 1),
                "TFLOP_per_step": round(w_.value / steps / 1e12, 2), "achieved_TFLOPs": round(w_.value / (ms_.value * 1e-3) / 1e12, 1),
                "frac_of_fp32_mfma_peak": round(w_.value / (ms_.value * 1e-3) / 1e12 / 157.3, 3)},
       "conv_path": os.environ.get("W2L_CONV_GLDS", "1")}
print("[c4] " + json.dumps(out), flush=True)
