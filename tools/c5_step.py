"""BASELINE config 5 on one GPU: sota/2019 Transformer-CTC (am_transformer_ctc.arch, 322.6 M parameters), batch 16,
T = 1500 (188 frames after the three max-pools), 9998 word pieces: one full training step (forward, CTC, backward,
clip + SGD).   python tools/c5_step.py [steps] [f32|bf16] [batch] [drop|nodrop]"""
import json, os, re, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from wav2letter_amd import CriterionScaleMode, _lib, recipes
from wav2letter_amd.trainer import Trainer

if os.environ.get("W2L_USE_PROBE"):   # A/B runs of probe-library switches
    _lib.use_probe().__enter__()

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
drop = (sys.argv[4] if len(sys.argv) > 4 else "drop") == "drop"
T, nfeat, nlabel, Lmax = 1500, 80, 9998, 80
fl = recipes.TRANSFORMER_CTC_FLAGS
arch = recipes.transformer_ctc_arch()
if not drop:
    arch = re.sub(r"^DO [0-9.]+$", "DO 0.0", arch, flags=re.M).replace("460 0.2 0.2", "460 0.0 0.0")
device = torch.device("cuda:0")
x, tgt = bench.make_batch(B, T, nfeat, nlabel, Lmax, 5, device)
tr = Trainer(arch, nfeat, nlabel, "ctc", CriterionScaleMode.TARGET_SZ_SQRT, device=device)
tr.init_params(seed=1)
Tout = tr.plan(B, T, Lmax)
tr.to_device()
tr.set_mixed_precision(mode == "bf16")
tr.set_optimizer(fl["netoptim"], fl["critoptim"])
it = [0]

def step():
    it[0] += 1
    tr.set_step(it[0])
    loss = tr.forward_backward(x, tgt)
    tr.update(lr=fl["lr"], lrcrit=fl["lrcrit"], momentum=fl["momentum"], max_grad_norm=fl["maxgradnorm"], total_batch=B)
    return loss

step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("[c5] " + json.dumps({"config": f"C5 Transformer-CTC: B={B}, T={T}, Tout={Tout}, {mode}, dropout/layerdrop {'on' if drop else 'off'}",
                            "ms_per_step": round(dt * 1e3, 1), "utterances_per_sec": round(B / dt, 2),
                            "loss_mean": float(loss.float().mean().item())}), flush=True)
