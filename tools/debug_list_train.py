"""debug: the Python Trainer on the list fixture of tools/mk_list_fixture.py with the C++ Train's data conventions (per-utterance
normalisation, frames padded to 64, batches of 3 alternating)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wav2letter_amd import data, recipes, text
from wav2letter_amd.features import Mfsc
from wav2letter_amd.trainer import Trainer
d = sys.argv[1]
samples = data.read_list(d + "/train.lst")
letters = open(d + "/tokens.txt").read().split()
lex = text.load_lexicon(d + "/lexicon.txt")
dic = text.create_token_dict(letters, "asg", replabel=2)
mf = Mfsc(num_filters=40)
feats = []
for s in samples:
    a, _ = data.read_audio(s.path)
    f = mf(torch.tensor(a).cuda()[None])[0]
    feats.append((f - f.mean()) / f.std(unbiased=False))
rows = [text.target_indices(s.transcript.split(), lex, dic, "asg", replabel=2, wordsep="|") for s in samples]
T = 64
def batch(idx):
    x = torch.zeros(len(idx), 40, T, device="cuda")
    for b, i in enumerate(idx):
        x[b, :, :feats[i].shape[1]] = feats[i]
    return x, torch.tensor(text.pad_targets([rows[i] for i in idx])).cuda()
tr = Trainer(open(d + "/arch/net.arch").read(), 40, dic.index_size(), "asg", 4, 0.0)
tr.init_params(3)
b0, b1 = batch([0, 1, 2]), batch([3, 4, 5])
L = max(b0[1].shape[1], b1[1].shape[1])
tr.plan(3, T, L)
tr.to_device()
for u in range(1, 41):
    x, t = b0 if u % 2 else b1
    if t.shape[1] < L:
        t = torch.nn.functional.pad(t, (0, L - t.shape[1]), value=-1)
    loss = tr.forward_backward(x, t.contiguous())
    gn = float(tr.grads.double().norm().item()) / 3.0
    tr.update(lr=0.05, lrcrit=0.002, momentum=0.8, max_grad_norm=1.0, total_batch=3)
    print("update %d loss %.4f grad norm / B %.4g" % (u, float(loss.mean().item()), gn), flush=True)
