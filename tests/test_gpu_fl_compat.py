"""The fl:: C++ surface (include/fl_compat/flashlight.h) driven by a COMPILED C++ caller (tests/cpp/fl_compat_test.cpp,
built by __graft_entry__.build() with plain g++ against libw2l_hip.so), checked against the oracle:

  * fl::pkg::speech::ASGLoss / CTCLoss: forward({emission (N,T,B), target (L,B)}) -> loss (B), loss.backward(),
    param(0).grad(), viterbiPath, viterbiPathWithTarget  (reference shapes: cpc/CPCCriterion.h:30-50, Train.cpp:406-410)
  * buildSequentialModule / ModulePlugin(dlopen createModule) + SGDOptimizer + clipGradNorm: one training step in the
    reference's order (Train.cpp:1454-1804) equals the same step through the ctypes Trainer
"""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "fl_compat_test")
PLUGIN = os.path.join(ROOT, "tests", "cpp", "libplugin_model.so")
PLUGIN_LAYERS = os.path.join(ROOT, "tests", "cpp", "libplugin_layers.so")
TOL = 1e-4


def run(args):
    if not os.path.exists(BIN):
        pytest.fail(f"{BIN} missing: __graft_entry__.build() compiles it (make -C tests/cpp)")
    out = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    return out.stdout


@pytest.mark.parametrize("kind,mode,B,T,N,L", [("asg", 4, 3, 60, 30, 20), ("asg", 0, 2, 9, 100, 5), ("ctc", 4, 3, 40, 50, 12)])
def test_criteria_through_compiled_cpp(oracle, tmp_path, kind, mode, B, T, N, L):
    rng = np.random.default_rng(B * 100 + T)
    em = rng.normal(size=(B, T, N)).astype(np.float32)
    tgt = np.full((B, L), -1, np.int32)
    for b in range(B):
        l = int(rng.integers(1, min(L, T) + 1))
        tgt[b, :l] = rng.integers(0, N - 1, size=l)
    A = (rng.normal(size=(N, N)) * 0.3 + 2 * np.eye(N)).astype(np.float32)
    gw = rng.uniform(0.5, 1.5, size=B).astype(np.float32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<4i", N, T, B, L))
        f.write(em.tobytes()); f.write(tgt.tobytes()); f.write(A.tobytes()); f.write(gw.tobytes())
    stdout = run(["crit", kind, str(mode), fin, fout])
    raw = open(fout, "rb").read()
    off = 0

    def take(n, dt):
        nonlocal off
        a = np.frombuffer(raw, dt, n, off)
        off += 4 * n
        return a
    loss = take(B, np.float32)
    dem = take(B * T * N, np.float32).reshape(B, T, N)
    if kind == "asg":
        assert "AutoSegmentationCriterion" in stdout
        dA = take(N * N, np.float32).reshape(N, N)
        path = take(B * T, np.int32).reshape(B, T)
        fpath = take(B * T, np.int32).reshape(B, T)
        ol, odx, odA = oracle.asg(em, A, tgt, mode, grad=gw.astype(np.float64))
        assert np.abs(dA - odA).max() < TOL * np.abs(odA).max()
        assert (path == oracle.viterbi(em, A)).all()
        fac = oracle.FAC(em, A, tgt, scale_mode=mode)
        fac.forward()
        assert (fpath == fac.viterbi()).all()
    else:
        assert "ConnectionistTemporalClassificationCriterion" in stdout
        path = take(B * T, np.int32).reshape(B, T)
        o = oracle.CTC(em, tgt, scale_mode=mode)
        ol = o.forward()
        odx = o.backward(gw.astype(np.float64))
        assert (path == oracle.ctc_viterbi(em)).all()
    assert off == len(raw)
    assert np.abs(loss - ol).max() < TOL * max(1.0, np.abs(ol).max())
    assert np.abs(dem - odx).max() < TOL * np.abs(odx).max()


@pytest.mark.parametrize("via", ["arch", "plugin", "layers"])
def test_training_step_through_fl_surface_equals_trainer(tmp_path, via):
    """network from an arch FILE (buildSequentialModule), from a dlopen'ed createModule plugin that assembles arch text, or from
    one that builds it out of LAYER OBJECTS (`encoder->add(std::make_shared<fl::Conv2D>(...))`, tests/cpp/plugin_layers.cpp, the
    style of recipes/slimIPL/100h_supervised.cpp:24-43), CTCLoss, zeroGrad,
    loss.backward(), grads / batch, clipGradNorm, SGD steps -- the compiled C++ caller reproduces the ctypes Trainer's
    losses before and after the update (same init seed, dropout off)"""
    from wav2letter_amd import recipes
    from wav2letter_amd.trainer import Trainer
    nfeat, nlabel, B, T, L = 8, 12, 3, 40, 5
    arch = ("V -1 NFEAT 1 0\nC2 1 4 5 1 2 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\n"
            f"TDS 4 5 {nfeat} 0.0 {4 * nfeat * 2}\nTDS 4 5 {nfeat} 0.0 0\nV 0 {4 * nfeat} 1 0\nRO 1 0 3 2\nL {4 * nfeat} NLABEL\n")
    rng = np.random.default_rng(21)
    x = rng.normal(size=(B, nfeat, T)).astype(np.float32)
    tgt = rng.integers(0, nlabel - 1, size=(B, L)).astype(np.int32)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<3i", T, B, L))
        f.write(x.tobytes()); f.write(tgt.tobytes())
    if via == "arch":
        target = str(tmp_path / "net.arch")
        open(target, "w").write(arch)
    else:
        target = PLUGIN if via == "plugin" else PLUGIN_LAYERS
        assert os.path.exists(target)
    stdout = run(["net", target, str(nfeat), str(nlabel), fin, fout])
    assert "TDS" in stdout or "Model" in stdout
    raw = open(fout, "rb").read()
    l0 = np.frombuffer(raw, np.float32, B, 0)
    l1 = np.frombuffer(raw, np.float32, B, 4 * B)
    checksum = np.frombuffer(raw, np.float32, 1, 8 * B)[0]
    nparams = np.frombuffer(raw, np.int32, 1, 8 * B + 4)[0]
    gnorm = np.frombuffer(raw, np.float32, 1, 8 * B + 8)[0]

    tr = Trainer(arch, nfeat, nlabel, "ctc", 4)
    tr.init_params(seed=1)     # the facade initialises with seed 1 too
    tr.plan(B, T, L)
    tr.to_device()
    xd, td = torch.tensor(x).cuda(), torch.tensor(tgt).cuda()
    em = tr.forward(xd, train=False)
    assert abs(em.double().sum().item() - checksum) < 1e-3 * max(1.0, abs(checksum))
    assert nparams == len(tr.param_table())
    # the reference step order with dropout-free arch: train-mode forward == eval forward
    want0 = tr.forward_backward(xd, td).cpu().numpy().copy()
    tr.update(lr=0.05, momentum=0.5, max_grad_norm=1.0, total_batch=B)
    want_norm = tr.grad_norm() / B
    want1 = tr.forward_backward(xd, td).cpu().numpy().copy()
    assert np.abs(l0 - want0).max() < TOL * np.abs(want0).max()
    assert abs(gnorm - want_norm) < 1e-3 * want_norm
    assert np.abs(l1 - want1).max() < 1e-3 * np.abs(want1).max()
    assert (l1 < l0).any()      # the update did something


@pytest.mark.parametrize("recipe", ["tds_ctc", "conv_glu", "transformer_ctc"])
def test_train_binary_reads_reference_cfg_and_prints_reference_log_keys(tmp_path, recipe):
    """`Train train --flagsfile=<the reference's train.cfg, unchanged> --rundir=... --archdir=... --tokensdir=...`:
    the recipe's own flags + arch files drive the C++ Trainer over the fl:: surface; the log line carries the reference's
    keys in the reference's order (recipes/slimIPL/src/MyLogger.cpp:40-106), 001_log / 001_config land in the run dir"""
    from wav2letter_amd import recipes
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    assert os.path.exists(exe), "build() links wav2letter_amd/bin/Train"
    d = tmp_path
    if recipe == "tds_ctc":
        cfg, arch_rel, arch = recipes.tds_ctc_train_cfg(), "am_arch/am_tds_ctc.arch", recipes.tds_ctc_arch()
        tokens, ntok = "librispeech-train-all-unigram-10000.tokens", 9997
        extra = ["--w2l_synth_frames=320", "--batchsize=2", "--w2l_synth_target_len=20"]
    elif recipe == "transformer_ctc":   # BASELINE config 5: 24 TR blocks, --netoptim=adadelta, bare --mfsc / --sqnorm flags
        cfg, arch_rel, arch = recipes.transformer_ctc_train_cfg(), "am_arch/am_transformer_ctc.arch", recipes.transformer_ctc_arch()
        tokens, ntok = "librispeech-train-all-unigram-10000.tokens", 9997
        extra = ["--w2l_synth_frames=320", "--batchsize=2", "--w2l_synth_target_len=12"]
    else:
        cfg, arch_rel, arch = recipes.conv_glu_train_cfg(), "network.arch", recipes.conv_glu_librispeech_arch()
        tokens, ntok = "tokens.txt", 28
        extra = ["--w2l_synth_frames=400", "--batchsize=2", "--w2l_synth_target_len=60", "--linseg=0"]
    os.makedirs(d / "arch" / os.path.dirname(arch_rel), exist_ok=True)
    open(d / "arch" / arch_rel, "w").write(arch)
    os.makedirs(d / "am")
    open(d / "am" / tokens, "w").write("".join(f"tok{i}\n" for i in range(ntok)))
    open(d / "train.cfg", "w").write(cfg)
    cmd = [exe, "train", f"--flagsfile={d / 'train.cfg'}", f"--rundir={d / 'runs'}", f"--archdir={d / 'arch'}",
           f"--tokensdir={d / 'am'}", "--w2l_synth_updates=3", "--reportiters=1"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch:")]
    assert len(lines) == 3
    keys = [kv.split(":")[0].strip() for kv in lines[-1].split(" | ")]
    assert keys[:18] == ["epoch", "nupdates", "lr", "lrcriterion", "runtime", "bch(ms)", "smp(ms)", "fwd(ms)", "crit-fwd(ms)", "bwd(ms)",
                         "optim(ms)", "loss", "train-TER", "train-WER", "avg-isz", "avg-tsz", "max-tsz", "avr-batchsz"]
    vals = dict((kv.split(":")[0].strip(), kv.split(":", 1)[1].strip()) for kv in lines[-1].split(" | "))
    assert int(vals["nupdates"]) == 3 and float(vals["loss"]) > 0 and np.isfinite(float(vals["loss"]))
    assert float(vals["fwd(ms)"]) > 0 and float(vals["bwd(ms)"]) > 0
    name = {"tds_ctc": "am_tds_ctc_librispeech", "conv_glu": "librispeech_conv_glu", "transformer_ctc": "am_transformer_ctc_librispeech"}[recipe]
    assert os.path.exists(d / "runs" / name / "001_log") and os.path.exists(d / "runs" / name / "001_config")
    assert "--criterion=" + ("asg" if recipe == "conv_glu" else "ctc") in open(d / "runs" / name / "001_config").read()
    crit = "AutoSegmentationCriterion" if recipe == "conv_glu" else "ConnectionistTemporalClassificationCriterion"
    assert crit in out.stdout
    if recipe == "transformer_ctc":
        assert "[Network Optimizer] Adadelta (rho=0.9)" in out.stdout and "Transformer" in out.stdout
    # a bad flag value fails like the reference (exception text, non-zero exit)
    bad = subprocess.run(cmd + ["--criterion=seq2seq"], capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "criterion" in bad.stderr


def test_train_binary_data_parallel_path_single_rank(tmp_path):
    """`Train train --enable_distributed=true --world_rank=0 --world_size=1 --rndv_filepath=...` (the reference's flags,
    Train.cpp:188-199): the C++ host creates the RCCL communicator itself (librccl.so is dlopen()ed), synchronises the
    replicas, routes every gradient through fl::CoalescingReducer and all-reduces the batch size -- with one rank that
    must reproduce the single-process run bit for bit (same synthetic data: rank 0's shard)"""
    from wav2letter_amd import recipes
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    d = tmp_path
    arch = ("V -1 NFEAT 1 0\nC2 1 4 5 1 2 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\nTDS 4 5 8 0.0 64\nV 0 32 1 0\nRO 1 0 3 2\nL 32 NLABEL\n")
    os.makedirs(d / "arch"); os.makedirs(d / "rndv")
    open(d / "arch" / "net.arch", "w").write(arch)
    base = [exe, "train", f"--archdir={d / 'arch'}", "--arch=net.arch", "--criterion=ctc", "--filterbanks=8", "--w2l_nlabel=12",
            "--batchsize=3", "--w2l_synth_frames=64", "--w2l_synth_target_len=6", "--w2l_synth_updates=4", "--reportiters=1",
            "--lr=0.05", "--momentum=0.5", "--maxgradnorm=1.0", "--onorm=target", "--sqnorm=true"]

    def losses(extra):
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
        rows = [l for l in out.stdout.splitlines() if l.startswith("epoch:")]
        assert len(rows) == 4
        return [dict((kv.split(":")[0].strip(), kv.split(":", 1)[1].strip()) for kv in r.split(" | "))["loss"] for r in rows], out.stdout
    single, _ = losses([])
    dist, text = losses(["--enable_distributed=true", "--world_rank=0", "--world_size=1", f"--rndv_filepath={d / 'rndv'}"])
    assert "[Distributed] world rank 0 of 1 (RCCL)" in text
    assert dist == single
    # a rank outside the world is refused like any bad flag
    bad = subprocess.run(base + ["--enable_distributed=true", "--world_rank=2", "--world_size=2"], capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "rank" in bad.stderr


def test_train_binary_lr_decay_and_specaugment_start(tmp_path):
    """the Trainer's schedule pieces the Transformer-CTC recipe uses (Train.cpp:1170-1175, :1334-1348, :1026-1048, :1453-1461):
    lr = lr0 * 0.5^(0 before --lr_decay, then 1 + (epoch - lr_decay) / lr_decay_step) * warm-up, and SpecAugment on the
    features from --saug_start_update on"""
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    d = tmp_path
    arch = ("V -1 NFEAT 1 0\nC2 1 4 5 1 2 1 -1 -1\nR\nDO 0.0\nLN 0 1 2\nTDS 4 5 8 0.0 64\nV 0 32 1 0\nRO 1 0 3 2\nL 32 NLABEL\n")
    os.makedirs(d / "arch")
    open(d / "arch" / "net.arch", "w").write(arch)
    cmd = [exe, "train", f"--archdir={d / 'arch'}", "--arch=net.arch", "--criterion=ctc", "--filterbanks=8", "--w2l_nlabel=12",
           "--batchsize=3", "--w2l_synth_frames=64", "--w2l_synth_target_len=6", "--w2l_synth_updates=4", "--reportiters=1",
           "--lr=0.8", "--momentum=0.0", "--maxgradnorm=1.0", "--onorm=target", "--sqnorm",
           "--lr_decay=2", "--lr_decay_step=1", "--w2l_synth_batches_per_epoch=1",
           "--saug_start_update=3", "--saug_fmaskf=2", "--saug_fmaskn=1", "--saug_tmaskt=5", "--saug_tmaskn=1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert "[SpecAugment from update 3] SpecAugment ( W: 0, F: 2, mF: 1, T: 5, p: 1, mT: 1 )" in out.stdout
    rows = [dict((kv.split(":")[0].strip(), kv.split(":", 1)[1].strip()) for kv in r.split(" | "))
            for r in out.stdout.splitlines() if r.startswith("epoch:")]
    assert [float(r["lr"]) for r in rows] == [0.8, 0.4, 0.2, 0.1]
    assert all(np.isfinite(float(r["loss"])) for r in rows)
    # the same run without SpecAugment: identical until the start update, different from it on
    base = subprocess.run([c for c in cmd if not c.startswith("--saug")], capture_output=True, text=True, timeout=600)
    brow = [l.split(" | ") for l in base.stdout.splitlines() if l.startswith("epoch:")]
    loss = lambda rws: [dict((kv.split(":")[0].strip(), kv.split(":", 1)[1].strip()) for kv in r)["loss"] for r in rws]
    a, b = [r["loss"] for r in rows], loss(brow)
    assert a[:2] == b[:2] and a[2:] != b[2:]


_TINY_ARCH = ("V -1 NFEAT 1 0\nC2 1 4 5 1 2 1 -1 -1\nR\nDO 0.1\nLN 0 1 2\nTDS 4 5 8 0.1 64\nV 0 32 1 0\nRO 1 0 3 2\nL 32 NLABEL\n")


def _tiny_train_cmd(d, rundir, updates, extra=()):
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    return [exe, "train", f"--archdir={d / 'arch'}", "--arch=net.arch", "--criterion=ctc", "--filterbanks=8", "--w2l_nlabel=12",
            "--batchsize=3", "--w2l_synth_frames=64", "--w2l_synth_target_len=6", f"--w2l_synth_updates={updates}", "--reportiters=1",
            "--lr=0.05", "--momentum=0.5", "--maxgradnorm=1.0", "--onorm=target", "--sqnorm=true", f"--rundir={rundir}",
            "--runname=exp"] + list(extra)


@pytest.mark.parametrize("optim", ["sgd", "adadelta"])
def test_train_binary_continue_is_bit_identical_and_fork_starts_fresh(tmp_path, optim):
    """`Train continue <directory>` (recipes/slimIPL/src/Train.cpp:117, :124-150, :460-467): three updates, stop, continue to
    five == five uninterrupted updates BIT FOR BIT (network, optimizer state; dropout is live, so the seed stream and the
    sample stream must resume where they stopped); run files are NNN_log / NNN_config / NNN_model_last.bin (:644-651, :767);
    the container is wav2letter_amd/checkpoint.py's (read back by the Python side, arch hash included).  `Train fork <model>`
    (:118, :151-166, :452-459) starts a new run from the model's network + criterion with fresh optimizers."""
    from wav2letter_amd import checkpoint
    from wav2letter_amd.trainer import Trainer
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    d = tmp_path
    os.makedirs(d / "arch")
    open(d / "arch" / "net.arch", "w").write(_TINY_ARCH)
    opt = [f"--netoptim={optim}", f"--critoptim={optim}"] + (["--lr=1.0"] if optim == "adadelta" else [])

    def run(cmd):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-2000:])
        return out.stdout
    run(_tiny_train_cmd(d, d / "A", 5, opt))
    run(_tiny_train_cmd(d, d / "B", 3, opt))
    text = run([exe, "continue", str(d / "B" / "exp"), "--w2l_synth_updates=5"])
    assert "Loaded model for continue training" in text
    rows = [l for l in text.splitlines() if l.startswith("epoch:")]
    assert [int(dict((kv.split(":")[0].strip(), kv.split(":", 1)[1].strip()) for kv in r.split(" | "))["nupdates"]) for r in rows] == [4, 5]
    for f in ("001_log", "001_config", "001_model_last.bin", "002_log", "002_config", "002_model_last.bin"):
        assert os.path.exists(d / "B" / "exp" / f), f
    ha, ta = checkpoint.read(str(d / "A" / "exp" / "001_model_last.bin"))
    hb, tb = checkpoint.read(str(d / "B" / "exp" / "002_model_last.bin"))
    assert ha["step"] == hb["step"] == 5 and ha["flags"]["nbupdates"] == hb["flags"]["nbupdates"] == "5"
    assert [t["kind"] for t in ha["tensors"]] == [t["kind"] for t in hb["tensors"]]
    assert "momentum" in [t["kind"] for t in ha["tensors"]] and (("state2" in [t["kind"] for t in ha["tensors"]]) == (optim == "adadelta"))
    for t, a, b in zip(ha["tensors"], ta, tb):
        assert np.array_equal(a, b), t["name"]
    # three updates differ from five (the comparison above is not vacuous)
    _, t3 = checkpoint.read(str(d / "B" / "exp" / "001_model_last.bin"))
    assert not np.array_equal(t3[0], ta[0])
    # the Python side reads the C++ container: same arch hash, reference-layout tensors land in the same arena
    tr = Trainer(_TINY_ARCH, 8, 12, "ctc", 4)
    tr.set_optimizer(optim, optim)
    tr.init_params(seed=99)
    assert checkpoint.load(str(d / "A" / "exp" / "001_model_last.bin"), tr, _TINY_ARCH) == 5
    assert "--criterion=ctc" in ha["flags"]["gflags"] and ha["optim"] == [optim, optim]
    for i, (name, n, off) in enumerate(tr.param_table()):
        assert np.array_equal(tr.export_from(i, tr.host_params), ta[i]), name
    # fork: new run directory, network from the model, optimizers fresh
    text = run([exe, "fork", str(d / "B" / "exp" / "001_model_last.bin"), f"--rundir={d / 'C'}", "--runname=forked", "--w2l_synth_updates=2"])
    assert "for fork" in text and os.path.exists(d / "C" / "forked" / "001_model_last.bin")
    hc, _ = checkpoint.read(str(d / "C" / "forked" / "001_model_last.bin"))
    assert hc["step"] == 2 and hc["flags"]["nbupdates"] == "2"
    # continue without a model is refused
    bad = subprocess.run([exe, "continue", str(d / "nowhere")], capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "model_last.bin" in bad.stderr


def test_train_binary_overlapped_bucket_reducer_single_rank(tmp_path):
    """the C++ Train's gradient reduction is the bucketed, event-gated one (fl::CoalescingReducer over the planned network's
    flat gradient arena: first update one collective, then one per bucket on the side stream, last bucket first, each behind the
    event backward() records when that part of the arena is final) -- with one rank it must leave the run bit-identical"""
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    d = tmp_path
    os.makedirs(d / "arch"); os.makedirs(d / "rndv")
    open(d / "arch" / "net.arch", "w").write(_TINY_ARCH)
    a = subprocess.run(_tiny_train_cmd(d, d / "S", 4), capture_output=True, text=True, timeout=600)
    b = subprocess.run(_tiny_train_cmd(d, d / "D", 4, ["--enable_distributed=true", "--world_rank=0", "--world_size=1",
                                                       f"--rndv_filepath={d / 'rndv'}"]), capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1500:], b.stderr[-1500:])
    import re
    m = re.search(r"gradient collectives of the last update: (\d+) \((\d+) issued on the side stream", b.stdout)
    assert m and int(m.group(2)) >= 2 and int(m.group(1)) == int(m.group(2)), b.stdout[-800:]
    from wav2letter_amd import checkpoint
    _, ta = checkpoint.read(str(d / "S" / "exp" / "001_model_last.bin"))
    _, tb = checkpoint.read(str(d / "D" / "exp" / "001_model_last.bin"))
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
    assert not os.path.exists(d / "rndv" / "w2l_nccl_id.1")   # a world of one publishes no rendezvous record


def _write_wav16(path, x):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.asarray(x, "<i2").tobytes())


def test_cpp_loadsound_and_mfsc_match_the_python_pipeline(tmp_path):
    """include/fl_compat/audio.h through the compiled C++ caller: loadSound of a WAV and of a FLAC file holding the same samples
    and fl::lib::audio::Mfsc on the device give the features of wav2letter_amd.features.Mfsc (the same folded pre-emphasis /
    window / DFT matrix and mel filterbank, computed independently in C++)"""
    from tests import flac_encode as FE
    from wav2letter_amd.features import Mfsc
    rng = np.random.default_rng(2)
    n = 16000 + 777
    t = np.arange(n) / 16000.0
    x = np.round((0.3 * np.sin(2 * np.pi * 310 * t) + 0.05 * rng.normal(size=n)) * 20000).astype(np.int16)
    _write_wav16(tmp_path / "a.wav", x)
    (tmp_path / "a.flac").write_bytes(FE.encode(x.astype(np.int64), kind="fixed", order=2, porder=3))
    want = Mfsc(num_filters=40)(torch.tensor(x.astype(np.float32) / 32768.0).cuda()[None])[0].cpu().numpy()   # [F][T]
    for name in ("a.wav", "a.flac"):
        out = tmp_path / (name + ".bin")
        run(["mfsc", str(tmp_path / name), "40", str(out)])
        raw = out.read_bytes()
        T, F, rate, ns = np.frombuffer(raw[:16], np.int32)
        got = np.frombuffer(raw[16:], np.float32).reshape(F, T)
        assert (T, F, rate, ns) == (want.shape[1], 40, 16000, n)
        assert np.abs(got - want).max() < 2e-4 * np.abs(want).max()


def test_train_binary_on_list_files(tmp_path):
    """`Train train` on REAL list files (recipes/slimIPL/src/Train.cpp:277-339): `id path duration transcript` lines over WAV and
    FLAC audio, letter tokens + lexicon, ASG with two replabels; features, targets and batching happen in the binary
    (fl_compat/audio.h, data.h, text.h).  The loss falls, the log line carries train-TER / train-WER from the Viterbi path, the
    run is reproducible, and `continue` picks the list data up again."""
    from tests import flac_encode as FE
    exe = os.path.join(ROOT, "wav2letter_amd", "bin", "Train")
    d = tmp_path
    os.makedirs(d / "arch")
    os.makedirs(d / "audio")
    from wav2letter_amd import recipes
    (d / "arch" / "net.arch").write_text(recipes.conv_glu_small_arch(widths=(32, 48), kws=(5, 5)))
    letters = ["|", "'"] + [chr(c) for c in range(ord("a"), ord("z") + 1)]
    (d / "tokens.txt").write_text("\n".join(letters) + "\n")
    words = ["hello", "aaa", "bee", "zoo", "add"]
    (d / "lexicon.txt").write_text("".join(f"{w}\t{' '.join(w)} |\n" for w in words))
    rng = np.random.default_rng(0)
    lines = []
    for k, (n, tr) in enumerate([(9600, "hello bee"), (6400, "aaa"), (8000, "zoo hello"), (4800, "bee"), (7300, "add zoo"), (5100, "hello")]):
        t = np.arange(n) / 16000.0
        sig = np.round((0.3 * np.sin(2 * np.pi * (200 + 150 * k) * t) + 0.05 * rng.normal(size=n)) * 30000).astype(np.int16)
        if k % 2:
            p = d / "audio" / f"u{k}.flac"
            p.write_bytes(FE.encode(sig.astype(np.int64), kind="fixed", order=2, porder=2))
        else:
            p = d / "audio" / f"u{k}.wav"
            _write_wav16(p, sig)
        lines.append(f"u{k} {p if k < 3 else 'audio/' + p.name} {n / 16.0:.1f} {tr}")    # absolute paths (as the recipes write) and --datadir-relative ones
    (d / "train.lst").write_text("\n".join(lines) + "\n")
    cmd = [exe, "train", f"--archdir={d / 'arch'}", "--arch=net.arch", "--criterion=asg", "--replabel=2", "--filterbanks=40", f"--tokensdir={d}",
           "--tokens=tokens.txt", f"--lexicon={d / 'lexicon.txt'}", f"--datadir={d}", "--train=train.lst", "--batchsize=3", "--iter=60",
           "--reportiters=20", "--lr=0.05", "--lrcrit=0.002", "--momentum=0.8", "--maxgradnorm=1.0", "--onorm=target", "--sqnorm=true",
           f"--rundir={d / 'run'}", "--runname=exp"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "[Data] 6 samples in 1 list(s), 6 on this rank, 2 batches of 3 per epoch; 40 MFSC features; 30 classes" in out.stdout

    def rows(text):
        r = []
        for line in text.splitlines():
            if line.startswith("epoch:"):
                r.append({k.strip(): v.strip() for k, v in (item.split(":", 1) for item in line.split(" | "))})
        return r
    rr = rows(out.stdout)
    assert [int(r["nupdates"]) for r in rr] == [20, 40, 60] and [int(r["epoch"]) for r in rr] == [10, 20, 30]
    losses = [float(r["loss"]) for r in rr]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert all(0.0 <= float(r["train-TER"]) <= 300.0 and 0.0 <= float(r["train-WER"]) <= 300.0 for r in rr)
    assert float(rr[-1]["train-TER"]) < 100.0
    assert rr[0]["avr-batchsz"].strip() == "3.00" and int(rr[0]["max-tsz"]) == 10    # "hello bee" -> h e l <1> o | b e <1> |
    # same command again: the same numbers (fixed seeds, deterministic kernels)
    out2 = subprocess.run(cmd[:-2] + [f"--rundir={d / 'run2'}", "--runname=exp"], capture_output=True, text=True, timeout=600, env=env)
    assert [r["loss"] for r in rows(out2.stdout)] == [r["loss"] for r in rr]
    # continue: four more updates on the same lists
    out3 = subprocess.run([exe, "continue", str(d / "run" / "exp"), "--iter=64", "--reportiters=2"], capture_output=True, text=True, timeout=600, env=env)
    assert out3.returncode == 0, out3.stdout[-2000:] + out3.stderr[-2000:]
    r3 = rows(out3.stdout)
    assert [int(r["nupdates"]) for r in r3] == [62, 64] and "[Data] 6 samples" in out3.stdout


_TINY_ARCH_NODROP = ("V -1 NFEAT 1 0\nC2 1 4 5 1 2 1 -1 -1\nR\nLN 0 1 2\nTDS 4 5 8 0 64\nV 0 32 1 0\nRO 1 0 3 2\nL 32 NLABEL\n")


def test_train_binary_two_ranks_equal_one_process(tmp_path):
    """the C++ data-parallel path with world_size = 2 on ONE GPU (round-3 verdict, a12): two `Train` processes rendezvous through
    fl::distributedInit's host-memory test collective (--rndv_filepath=shm:<file>; RCCL refuses two ranks on one device) and run
    allReduceParameters, the bucketed event-gated fl::CoalescingReducer (first update: the arena in one piece; then one
    collective per bucket on the side stream, last bucket first, behind backward()'s events), the batch-size reduce and the
    barrier.  Both ranks must end bit-identical, and equal to ONE process that trains on the concatenated batch
    (--w2l_synth_emulate_world=2 draws the two shards) up to the order of the gradient sum (1e-5 of each tensor's scale).
    Reference: recipes/slimIPL/src/Train.cpp:188-196, :1078-1079, :1651-1660, :1721-1747."""
    import re
    from wav2letter_amd import checkpoint
    d = tmp_path
    os.makedirs(d / "arch")
    open(d / "arch" / "net.arch", "w").write(_TINY_ARCH_NODROP)
    shm = f"/dev/shm/w2l_test_collective_{os.getpid()}"
    common = ["--netoptim=sgd", "--critoptim=sgd"]
    procs = []
    for r in (0, 1):
        cmd = _tiny_train_cmd(d, d / f"R{r}", 4, common + ["--enable_distributed=true", f"--world_rank={r}", "--world_size=2",
                                                          f"--rndv_filepath=shm:{shm}", "--w2l_save_all_ranks=true"])
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, (o[-1500:], e[-1500:])
    assert "[Distributed] world rank 0 of 2 (host-memory test collective)" in outs[0][1]
    assert "[Distributed] world rank 1 of 2 (host-memory test collective)" in outs[1][1]
    m = re.search(r"gradient collectives of the last update: (\d+) \((\d+) issued on the side stream", outs[0][1])
    assert m and int(m.group(2)) >= 2, outs[0][1][-800:]          # the bucketed path ran with two ranks
    one = subprocess.run([c if not c.startswith("--batchsize=") else "--batchsize=6" for c in _tiny_train_cmd(d, d / "ONE", 4, common + ["--w2l_synth_emulate_world=2"])],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, (one.stdout[-1500:], one.stderr[-1500:])
    _, t0 = checkpoint.read(str(d / "R0" / "exp" / "001_model_last.bin"))
    h1, t1 = checkpoint.read(str(d / "ONE" / "exp" / "001_model_last.bin"))
    # both replicas end bit-identical (rank 1 saves its own under --w2l_save_all_ranks)
    _, tr1 = checkpoint.read(str(d / "R1" / "exp" / "001_model_last.bin.rank1"))
    assert len(tr1) == len(t0) and all(np.array_equal(x, y) for x, y in zip(t0, tr1))
    for t, a, b in zip(h1["tensors"], t0, t1):
        scale = max(1e-6, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 1e-5 * scale, (t["name"], float(np.abs(a - b).max()), scale)
    # not vacuous: ONE process on its own stream of 6 utterances (no shard emulation) ends somewhere else
    other = subprocess.run([c if not c.startswith("--batchsize=") else "--batchsize=6" for c in _tiny_train_cmd(d, d / "OTHER", 4, common)],
                           capture_output=True, text=True, timeout=600)
    assert other.returncode == 0
    _, t2 = checkpoint.read(str(d / "OTHER" / "exp" / "001_model_last.bin"))
    assert any(np.abs(a - b).max() > 1e-3 * max(1e-6, float(np.abs(b).max())) for a, b in zip(t2, t1))
