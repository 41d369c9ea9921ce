"""Checkpoint interchange and learning-rate schedule.

The reference serialises `(config, network, criterion, netoptim, critoptim)` with cereal
(recipes/slimIPL/src/Train.cpp:718-827) -- a format that needs Flashlight to read.  This module defines
a documented, dependency-free container that keeps the REFERENCE's parameter order and array layouts
(Flashlight `params()` order; conv `(kw,1,cin,cout)`, linear `(out,in)`, WeightNorm `v, g, bias` --
recipes/utilities/convlm_serializer/Utils.cpp:112-143, recipes/joint_training_vox_populi/README.md:19-32),
so that a maintainer can convert either way with a few lines of cereal code:

    bytes 0..7    magic  b"W2LAMD01"
    bytes 8..11   little-endian uint32  n = length of the JSON header
    bytes 12..    JSON header (utf-8): {"nfeat", "nlabel", "criterion", "arch_sha256", "step", "flags": {...},
                   "optim": [netoptim, critoptim],
                   "tensors": [{"name", "kind": "network"|"criterion"|"momentum"|"state2", "shape": [...], "numel"}]}
    then, 16-byte aligned, every tensor's float32 data in header order (little endian, reference layout).

Network tensors go through `Trainer.export_from / import_param` (internal layout <-> reference layout);
criterion parameters (ASG transitions `(N, N)`, `[to][from]`) and the optimizer arenas are raw: "momentum" is SGD's velocity,
Adagrad's squared-gradient sums or Adadelta's accGrad (header "optim" says which), "state2" is Adadelta's accDelta.
"""
import hashlib
import json
import math
import struct

import numpy as np

MAGIC = b"W2LAMD01"


def _align16(n):
    return (n + 15) // 16 * 16


def save(path, trainer, arch_text, criterion, step=0, flags=None, momentum=True):
    """write network + criterion parameters (and the optimizer's momentum arena) of `trainer`"""
    host = trainer.params.detach().cpu().numpy() if trainer.params is not None else trainer.host_params
    table = trainer.param_table()
    tensors, blobs = [], []
    for i, (name, numel, _) in enumerate(table):
        ref = trainer.export_from(i, host)
        tensors.append({"name": name, "kind": "network", "numel": int(numel), "shape": [int(numel)]})
        blobs.append(np.ascontiguousarray(ref, np.float32))
    ncrit = trainer.n_floats - trainer.n_net
    if ncrit:
        # ASG transitions: exactly N*N floats, (N, N) [to][from] (the arena slot may be padded to a multiple of 4)
        n = trainer.nlabel
        nn = n * n if criterion == "asg" and n * n <= ncrit else int(ncrit)
        shape = [n, n] if nn == n * n else [nn]
        tensors.append({"name": "criterion.transitions", "kind": "criterion", "numel": nn, "shape": shape})
        blobs.append(np.ascontiguousarray(host[trainer.n_net:trainer.n_net + nn], np.float32))
    if momentum and trainer.mom is not None:
        tensors.append({"name": "netoptim.momentum(internal arena)", "kind": "momentum", "numel": int(trainer.n_floats),
                        "shape": [int(trainer.n_floats)]})
        blobs.append(np.ascontiguousarray(trainer.mom.detach().cpu().numpy(), np.float32))
        if getattr(trainer, "state2", None) is not None:
            tensors.append({"name": "netoptim.accDelta(internal arena)", "kind": "state2", "numel": int(trainer.n_floats),
                            "shape": [int(trainer.n_floats)]})
            blobs.append(np.ascontiguousarray(trainer.state2.detach().cpu().numpy(), np.float32))
    header = json.dumps({"nfeat": trainer.nfeat, "nlabel": trainer.nlabel, "criterion": criterion,
                         "optim": list(getattr(trainer, "_optim", ("sgd", "sgd"))),
                         "arch_sha256": hashlib.sha256(arch_text.encode()).hexdigest(), "step": int(step),
                         "flags": flags or {}, "tensors": tensors}).encode()
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(header)))
        f.write(header)
        pos = 12 + len(header)
        for b in blobs:
            pad = _align16(pos) - pos
            f.write(b"\0" * pad)
            f.write(b.tobytes())
            pos += pad + b.nbytes


def read(path):
    """-> (header dict, [float32 arrays in header order])"""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != MAGIC:
        raise ValueError(f"{path}: not a W2LAMD01 checkpoint")
    (n,) = struct.unpack("<I", raw[8:12])
    header = json.loads(raw[12:12 + n].decode())
    pos, arrays = 12 + n, []
    for t in header["tensors"]:
        pos = _align16(pos)
        nb = 4 * t["numel"]
        if pos + nb > len(raw):
            raise ValueError(f"{path}: truncated at tensor {t['name']}")
        arrays.append(np.frombuffer(raw, np.float32, t["numel"], pos).copy())
        pos += nb
    return header, arrays


def load(path, trainer, arch_text=None):
    """restore parameters (and momentum, if present) into `trainer`; returns the saved step.
    Raises ValueError on a geometry / architecture mismatch (the reference aborts in cereal the same way)."""
    import torch
    header, arrays = read(path)
    if (header["nfeat"], header["nlabel"]) != (trainer.nfeat, trainer.nlabel):
        raise ValueError("checkpoint was written for a different NFEAT / NLABEL")
    if arch_text is not None and hashlib.sha256(arch_text.encode()).hexdigest() != header["arch_sha256"]:
        raise ValueError("checkpoint was written for a different architecture file")
    want = getattr(trainer, "criterion", None)
    if want is not None and header.get("criterion") not in (None, want):
        raise ValueError(f"checkpoint was written for criterion {header['criterion']!r}, this trainer runs {want!r}")
    table = trainer.param_table()
    net = [(t, a) for t, a in zip(header["tensors"], arrays) if t["kind"] == "network"]
    if len(net) != len(table) or any(t["numel"] != row[1] for (t, _), row in zip(net, table)):
        raise ValueError("parameter list does not match this network")
    for i, (_, a) in enumerate(net):
        trainer.import_param(i, a)
    for t, a in zip(header["tensors"], arrays):
        if t["kind"] == "criterion":
            if t["numel"] > trainer.n_floats - trainer.n_net or t["numel"] + 3 < trainer.n_floats - trainer.n_net:
                raise ValueError("criterion parameters do not match")
            trainer.host_params[trainer.n_net:trainer.n_net + t["numel"]] = a
    if (trainer.n_floats > trainer.n_net) != any(t["kind"] == "criterion" for t in header["tensors"]):
        raise ValueError("criterion parameters do not match (one side has transitions, the other has none)")
    mom = [a for t, a in zip(header["tensors"], arrays) if t["kind"] == "momentum"]
    st2 = [a for t, a in zip(header["tensors"], arrays) if t["kind"] == "state2"]
    if (mom and mom[0].size != trainer.n_floats) or (st2 and st2[0].size != trainer.n_floats):
        raise ValueError("optimizer arenas do not match this trainer")
    optim = tuple(header.get("optim", ("sgd", "sgd")))
    if mom and optim != tuple(getattr(trainer, "_optim", ("sgd", "sgd"))):
        # the arena's MEANING depends on the optimizer (velocity / squared-gradient sums / accGrad)
        raise ValueError(f"checkpoint holds the state of optimizers {optim}; call trainer.set_optimizer{optim} before load()")
    if ("adadelta" in optim) != bool(st2) and mom:
        raise ValueError("Adadelta checkpoint without its accDelta arena")
    if trainer.params is not None:
        trainer.params.copy_(torch.from_numpy(trainer.host_params))
        if mom:
            trainer.mom.copy_(torch.from_numpy(mom[0]))
        if st2:
            trainer.state2.copy_(torch.from_numpy(st2[0]))
    else:
        if mom:
            trainer._pending_mom = mom[0]   # applied by Trainer.to_device() (load() before to_device() is the natural order)
        if st2:
            trainer._pending_state2 = st2[0]
    trainer.set_step(header["step"])
    return header["step"]


def learning_rate(flags, cur_batch, cur_epoch, n_batches=None, base=None):
    """The reference's schedule (recipes/slimIPL/src/Train.cpp:1171-1175, :1334-1348):
        lr = base * 0.5^(0 if e < lr_decay else 1 + (e - lr_decay) // lr_decay_step)
                  * (cos(pi/2 * batch / nbatches) if lrcosine else gamma^(batch / stepsize))
                  * min(batch / warmup, 1)
    `flags`: dict of gflags values (lr, lr_decay, lr_decay_step, lrcosine, gamma, stepsize, warmup);
    `base` overrides flags['lr'] (pass flags['lrcrit'] for the criterion optimizer)."""
    lr = float(flags.get("lr", 1.0) if base is None else base)
    lr_decay = int(flags.get("lr_decay", 2 ** 31 - 1))
    lr_decay_step = max(1, int(flags.get("lr_decay_step", 2 ** 31 - 1)))
    after = cur_epoch - lr_decay
    lr *= 0.5 ** (0 if after < 0 else 1 + after // lr_decay_step)
    if flags.get("lrcosine", False):
        if not n_batches:
            raise ValueError("lrcosine needs the total number of batches")
        lr *= math.cos(cur_batch / n_batches * math.pi / 2.0)
    else:
        lr *= float(flags.get("gamma", 1.0)) ** (cur_batch / float(flags.get("stepsize", 2 ** 31 - 1)))
    warmup = float(flags.get("warmup", 1))
    return lr * min(cur_batch / warmup if warmup > 0 else 1.0, 1.0)
