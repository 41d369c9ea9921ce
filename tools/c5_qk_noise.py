"""config-5 full network in the mixed-precision mode (the case of tests/test_gpu_trainer.py::test_transformer_ctc_config5_full_network_bf16):
per tensor kind the worst relative L2 / cosine of the parameter gradients against the float64 restatement, with the fused attention
backward (product library) and with the unfused launch sequence (probe library, W2L_AB_OFF=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_trainer as tg
from oracle import pyoracle
from wav2letter_amd import _lib
pyoracle.lib()
want = None
for mode in ("fused", "unfused"):
    if mode == "unfused":
        os.environ["W2L_AB_OFF"] = "1"
        _lib.use_probe().__enter__()
    rng = np.random.default_rng(51)
    nfeat, nlabel, B, T, L, x, tgt = tg._config5_case(rng)
    arch = tg._transformer_ctc_arch_no_dropout()
    tr, ref, params, _ = tg.build(arch, nfeat, nlabel, "ctc", 4, 0.0, rng, B, T, L)
    tr.set_mixed_precision(True)
    xd = torch.tensor(x.reshape(B, nfeat, T)).cuda(); td = torch.tensor(tgt).cuda()
    loss = tr.forward_backward(xd, td).cpu().numpy()
    if want is None:
        em_ref = ref.forward(x, params)
        o = pyoracle.CTC(em_ref, tgt, scale_mode=4); o.forward()
        want = ref.backward(o.backward().astype(np.float32), len(params))
    g = tr.grads.cpu().numpy()
    worst = {}
    for i, (name, _n, _off) in enumerate(tr.param_table()):
        got = np.asarray(tr.export_from(i, g), np.float64).reshape(-1)
        w = np.asarray(want[i], np.float64).reshape(-1)
        if w.size <= 2 or name == "tr.wk.b": continue
        l2 = np.linalg.norm(got - w) / max(1e-30, np.linalg.norm(w))
        cos = float(got @ w) / max(1e-30, np.linalg.norm(got) * np.linalg.norm(w))
        if l2 > worst.get(name, (0, 0, 0))[0]: worst[name] = (round(l2, 3), round(cos, 4), i)
    print(mode, {k: v for k, v in worst.items() if k.startswith("tr.")}, flush=True)
    del tr
