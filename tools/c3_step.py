"""BASELINE config 3 on one GPU: streaming_convnets LibriSpeech TDS-CTC (am_500ms_future_context.arch, 115.1 M parameters),
batch 64, T = 1500, 9998 word pieces: full training steps (forward, CTC, backward, clip + SGD) in fp32 or mixed precision.
  python tools/c3_step.py [steps] [f32|bf16] [batch]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wav2letter_amd import CriterionScaleMode, _lib, recipes
from wav2letter_amd.trainer import Trainer

if os.environ.get("W2L_USE_PROBE"):   # A/B runs of probe-library switches
    _lib.use_probe().__enter__()

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
T, nfeat, nlabel, Lmax = 1500, 80, 9998, 80
fl = recipes.STREAMING_TDS_FLAGS
device = torch.device("cuda:0")
x, tgt = bench.make_batch(B, T, nfeat, nlabel, Lmax, 3, device)
tr = Trainer(recipes.streaming_tds_arch(), nfeat, nlabel, "ctc", CriterionScaleMode.TARGET_SZ_SQRT, device=device)
tr.init_params(seed=1)
Tout = tr.plan(B, T, Lmax)
tr.to_device()
tr.set_mixed_precision(mode == "bf16")


def step():
    loss = tr.forward_backward(x, tgt)
    tr.update(lr=fl["lr"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
    return loss


step()
torch.cuda.synchronize()
L = _lib.lib()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# the per-kind figures: further steps with the event brackets on (a bracket costs ~4.6 us, 140 per step: not in the timed steps)
L.w2l_profile_enable(1)
for _ in range(steps):
    step()
torch.cuda.synchronize()
out = {"config": f"C3 streaming TDS-CTC: B={B}, T={T}, Tout={Tout}, {mode}", "ms_per_step": round(dt * 1e3, 2),
       "utterances_per_sec": round(B / dt, 1), "loss_mean": float(loss.float().mean().item())}
for name, kind in (("gemm_f32", 0), ("gemm_bf16", 6), ("tds_conv_fwd", 2), ("tds_conv_bwd_data", 4), ("tds_conv_bwd_filter", 5)):
    n_, ms_, w_ = C.c_int(0), C.c_double(0), C.c_double(0)
    L.w2l_profile_report_kind(kind, C.byref(n_), C.byref(ms_), C.byref(w_))
    if n_.value:
        out[name] = {"launches_per_step": n_.value // steps, "ms_per_step": round(ms_.value / steps, 2),
                     "TFLOPs": round(w_.value / (ms_.value * 1e-3) / 1e12, 1)}
L.w2l_profile_enable(0)
print("[c3] " + json.dumps(out), flush=True)
