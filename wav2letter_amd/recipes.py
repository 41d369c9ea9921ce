"""The reference's recipe inputs this build targets, regenerated programmatically.

The GPU box has no /root/reference, so the arch texts bench.py / smoke() need are
produced here from compact specs; tests/test_recipes.py checks (when /root/reference is
present) that each generator reproduces the reference file line for line, i.e. the
recipes load unchanged:
  recipes/sota/2019/am_arch/am_tds_ctc.arch                        (BASELINE config 2)
  recipes/conv_glu/librispeech/network.arch, conv_glu/wsj/network.arch   (configs 4, 1)
  recipes/streaming_convnets/librispeech/am_500ms_future_context.arch    (config 3)
  recipes/sota/2019/am_arch/am_transformer_ctc.arch                      (config 5)
"""


def tds_ctc_arch():
    """sota/2019 TDS-CTC: 3 x (strided C2 + R + DO + LN + k TDS) + Linear, 203.4 M params"""
    lines = ["SAUG 80 27 2 100 1.0 2", "V -1 NFEAT 1 0"]
    stages = [(1, 10, 2400, [0.05, 0.05, 0.05, 0.1, 0.1]),
              (10, 14, 3360, [0.15] * 6),
              (14, 18, 4320, [0.15, 0.15, 0.15, 0.15, 0.2, 0.2, 0.25, 0.25, 0.25, 0.25])]
    for cin, c, l2, drops in stages:
        lines += [f"C2 {cin} {c} 21 1 2 1 -1 -1", "R", "DO 0.0", "LN 0 1 2"]
        lines += [f"TDS {c} 21 80 {p} {l2}" for p in drops]
    lines += ["V 0 1440 1 0", "RO 1 0 3 2", "L 1440 NLABEL"]
    return "\n".join(lines) + "\n"


def tds_ctc_librivox_arch():
    """sota/2019 TDS-CTC for the LibriVox-scale model (am_tds_ctc_librivox.arch): 21 x 3 sub-sampling convolutions over
    (time, mel), channel counts 16 / 16 / 32 / 48, 19 TDS blocks"""
    lines = ["SAUG 80 27 2 100 1.0 2", "V -1 NFEAT 1 0"]
    stages = [(1, 16, "21 3 2", 2400, [0.05, 0.05]), (16, 16, "21 3 2", 2400, [0.05, 0.05]), (16, 32, "21 3 2", 4800, [0.1] * 5),
              (32, 48, "21 1 1", 7200, [0.1] * 6)]
    for cin, c, geom, l2, drops in stages:
        lines += [f"C2 {cin} {c} {geom} 1 -1 -1", "R", "DO 0.0", "LN 0 1 2"]
        lines += [f"TDS {c} 21 80 {p} {l2}" for p in drops]
    lines += ["V 0 3840 1 0", "RO 1 0 3 2", "L 3840 NLABEL"]
    return "\n".join(lines) + "\n"


def transformer_ctc_arch():
    """sota/2019 Transformer-CTC (am_transformer_ctc.arch): three WN-Conv(k=3) + GLU + max-pool(2) stages, 24 Transformer
    blocks of width 1024 (4 heads, 4096-wide MLP, +-460 frames of relative position), Linear; 322.6 M params"""
    lines = ["V -1 1 NFEAT 0"]
    for cin, cout in (("NFEAT", 1024), (512, 1024), (512, 2048)):
        lines += [f"WN 3 C {cin} {cout} 3 1 -1", "GLU 2", "DO 0.2", "M 1 1 2 1"]
    lines += ["RO 2 0 3 1"] + ["TR 1024 4096 4 460 0.2 0.2"] * 24 + ["DO 0.2", "L 1024 NLABEL"]
    return "\n".join(lines) + "\n"


# recipes/sota/2019/librispeech/train_am_transformer_ctc.cfg:11-41
TRANSFORMER_CTC_FLAGS = dict(criterion="ctc", netoptim="adadelta", critoptim="adadelta", lr=0.4, lrcrit=0.4, momentum=0.0,
                             maxgradnorm=1.0, onorm="target", sqnorm=True, batchsize=8)


def transformer_ctc_train_cfg():
    """recipes/sota/2019/librispeech/train_am_transformer_ctc.cfg, regenerated (checked line for line by tests/test_recipes.py)"""
    fl = [("runname", "am_transformer_ctc_librispeech"), ("rundir", "[...]"), ("archdir", "[...]"), ("arch", "am_arch/am_transformer_ctc.arch"),
          ("tokensdir", "[MODEL_DST]/am"), ("tokens", "librispeech-train-all-unigram-10000.tokens"),
          ("lexicon", "[MODEL_DST]/am/librispeech-train+dev-unigram-10000-nbest10.lexicon"),
          ("train", "[DATA_DST]/lists/train-clean-100.lst,[DATA_DST]/lists/train-clean-360.lst,[DATA_DST]/lists/train-other-500.lst"),
          ("valid", "dev-clean:[DATA_DST]/lists/dev-clean.lst,dev-other:[DATA_DST]/lists/dev-other.lst"),
          ("criterion", "ctc"), ("mfsc", None), ("usewordpiece", "true"), ("wordseparator", "_"), ("labelsmooth", "0.05"),
          ("dataorder", "output_spiral"), ("inputbinsize", "25"), ("softwstd", "4"), ("memstepsize", "5000000"), ("pcttraineval", "1"),
          ("pctteacherforcing", "99"), ("sampletarget", "0.01"), ("netoptim", "adadelta"), ("critoptim", "adadelta"), ("lr", "0.4"),
          ("lrcrit", "0.4"), ("linseg", "0"), ("momentum", "0.0"), ("maxgradnorm", "1.0"), ("onorm", "target"), ("sqnorm", None),
          ("nthread", "6"), ("batchsize", "8"), ("filterbanks", "80"), ("minisz", "200"), ("mintsz", "2"), ("enable_distributed", None),
          ("warmup", "32000"), ("saug_start_update", "32000"), ("lr_decay", "180"), ("lr_decay_step", "40")]
    return ("# Replace `[...]`, `[MODEL_DST]`, `[DATA_DST]`, with appropriate paths\n" +
            "".join(f"--{k}\n" if v is None else f"--{k}={v}\n" for k, v in fl))


def conv_glu_librispeech_arch():
    """conv_glu LibriSpeech: 17 WN-Conv+GLU layers, widths x1.1, kernel 13..29, 208.9 M params"""
    lines = ["V -1 1 NFEAT 0"]
    outs = [400, 440, 484, 532, 584, 642, 706, 776, 852, 936, 1028, 1130, 1242, 1366, 1502, 1652, 1816]
    drops = ["0.2", "0.214", "0.22898", "0.2450086", "0.262159202", "0.28051034614", "0.30014607037",
             "0.321156295296", "0.343637235966", "0.367691842484", "0.393430271458", "0.42097039046",
             "0.450438317792", "0.481969000038", "0.51570683004", "0.551806308143", "0.590432749713"]
    cin = "NFEAT"
    for i, (co, p) in enumerate(zip(outs, drops)):
        pad = 170 if i == 0 else 0
        lines += [f"WN 3 C {cin} {co} {13 + i} 1 {pad}", "GLU 2", f"DO {p}"]
        cin = co // 2
    lines += ["RO 2 0 3 1", "WN 0 L 908 1816", "GLU 0", "DO 0.590432749713", "WN 0 L 908 NLABEL"]
    return "\n".join(lines) + "\n"


def streaming_tds_arch():
    """streaming_convnets LibriSpeech (BASELINE config 3, am_500ms_future_context.arch): 4 stages of asymmetric-padded
    strided C2 + per-frame LayerNorm + TDS blocks with c = 15 / 19 / 23 / 27 channels, fc width = c*80, 115.1 M params"""
    lines = ["V -1 NFEAT 1 0", "SAUG 80 27 2 100 1.0 2"]
    stages = [("PD 0 5 3", 1, 15, 10, 2, [(9, 1)] * 2), ("PD 0 7 1", 15, 19, 10, 2, [(9, 1)] * 3),
              ("PD 0 9 1", 19, 23, 12, 2, [(11, 1)] * 3 + [(11, 0)]), ("PD 0 10 0", 23, 27, 11, 1, [(11, 0)] * 5)]
    for pd, cin, c, kw, stride, blocks in stages:
        lines += [pd, f"C2 {cin} {c} {kw} 1 {stride} 1 0 0", "R", "DO 0.1", "LN 1 2"]
        lines += [f"TDS {c} {k} 80 0.1 0 {rpad} 0" for k, rpad in blocks]
    lines += ["RO 2 1 0 3", "V 2160 -1 1 0", "L 2160 NLABEL", "V NLABEL 0 -1 1"]
    return "\n".join(lines) + "\n"


STREAMING_TDS_FLAGS = dict(criterion="ctc", lr=0.4, momentum=0.0, maxgradnorm=0.5, onorm="target", sqnorm=True,
                           filterbanks=80, batchsize=8)   # recipes/streaming_convnets/librispeech/train_am_500ms_future_context.cfg


def conv_glu_wsj_arch():
    """conv_glu WSJ (BASELINE config 1): 15 WN-Conv+GLU layers with SAME padding (-1), kernels 13, 3..15, 21"""
    lines = ["V -1 1 NFEAT 0"]
    outs = [200, 200, 200, 250, 250, 300, 350, 400, 450, 500, 500, 500, 600, 600, 750]
    kws = [13, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 21]
    cin = "NFEAT"
    for co, kw in zip(outs, kws):
        lines += [f"WN 3 C {cin} {co} {kw} 1 -1", "GLU 2", "DO 0.25"]
        cin = co // 2
    lines += ["RO 2 0 3 1", "WN 0 L 375 1000", "GLU 0", "DO 0.25", "WN 0 L 500 NLABEL"]
    return "\n".join(lines) + "\n"


def tds_ctc_small_arch(c=(4, 6), h=8, kw=5, l2mult=2, drop=0.0):
    """a reduced TDS-CTC of the same topology (tests)"""
    lines = [f"V -1 NFEAT 1 0"]
    cin = 1
    for ci in c:
        lines += [f"C2 {cin} {ci} {kw} 1 2 1 -1 -1", "R", f"DO {drop}", "LN 0 1 2",
                  f"TDS {ci} {kw} {h} {drop} {ci * h * l2mult}", f"TDS {ci} {kw} {h} {drop} 0"]
        cin = ci
    lines += [f"V 0 {c[-1] * h} 1 0", "RO 1 0 3 2", f"L {c[-1] * h} NLABEL"]
    return "\n".join(lines) + "\n"


def conv_glu_small_arch(widths=(16, 24), kws=(5, 4), pad0=4, drop=0.0):
    lines = ["V -1 1 NFEAT 0"]
    cin = "NFEAT"
    for i, (co, kw) in enumerate(zip(widths, kws)):
        lines += [f"WN 3 C {cin} {co} {kw} 1 {pad0 if i == 0 else -1}", "GLU 2", f"DO {drop}"]
        cin = co // 2
    lines += ["RO 2 0 3 1", f"WN 0 L {cin} {2 * cin}", "GLU 0", f"DO {drop}", f"WN 0 L {cin} NLABEL"]
    return "\n".join(lines) + "\n"


# train.cfg essentials of the two headline recipes (flags the hot path consumes)
TDS_CTC_FLAGS = dict(criterion="ctc", lr=0.3, momentum=0.5, maxgradnorm=1.0, onorm="target", sqnorm=True,
                     filterbanks=80, batchsize=4)       # recipes/sota/2019/librispeech/train_am_tds_ctc.cfg:11-30
CONV_GLU_FLAGS = dict(criterion="asg", lr=0.6, lrcrit=0.006, momentum=0.8, maxgradnorm=0.2, onorm="target",
                      sqnorm=True, filterbanks=40, batchsize=4, transdiag=4, replabel=2, linseg=1)
#                                                        recipes/conv_glu/librispeech/train.cfg:12-26


def tds_ctc_train_cfg():
    """recipes/sota/2019/librispeech/train_am_tds_ctc.cfg, regenerated (tests/test_recipes.py checks it against the
    reference file line for line when /root/reference is present): the flags file `Train train --flagsfile=...` reads"""
    fl = [("runname", "am_tds_ctc_librispeech"), ("rundir", "[...]"), ("archdir", "[...]"), ("arch", "am_arch/am_tds_ctc.arch"),
          ("tokensdir", "[MODEL_DST]/am"), ("tokens", "librispeech-train-all-unigram-10000.tokens"),
          ("lexicon", "[MODEL_DST]/am/librispeech-train+dev-unigram-10000-nbest10.lexicon"),
          ("train", "[DATA_DST]/lists/train-clean-100.lst,[DATA_DST]/lists/train-clean-360.lst,[DATA_DST]/lists/train-other-500.lst"),
          ("valid", "dev-clean:[DATA_DST]/lists/dev-clean.lst,dev-other:[DATA_DST]/lists/dev-other.lst"),
          ("batchsize", "4"), ("lr", "0.3"), ("momentum", "0.5"), ("maxgradnorm", "1"), ("onorm", "target"), ("sqnorm", "true"),
          ("mfsc", "true"), ("nthread", "10"), ("criterion", "ctc"), ("memstepsize", "8338608"), ("wordseparator", "_"),
          ("usewordpiece", "true"), ("filterbanks", "80"), ("gamma", "0.5"), ("enable_distributed", "true"), ("stepsize", "200"),
          ("framesizems", "30"), ("framestridems", "10"), ("seed", "2"), ("lr_decay", "10000")]
    return "# Replace `[...]`, `[MODEL_DST]`, `[DATA_DST]`, with appropriate paths\n" + "".join(f"--{k}={v}\n" for k, v in fl)


def conv_glu_train_cfg():
    """recipes/conv_glu/librispeech/train.cfg, regenerated (checked like tds_ctc_train_cfg)"""
    fl = [("runname", "librispeech_conv_glu"), ("rundir", "[...]"), ("tokensdir", "[MODEL_DST]/am"), ("archdir", "[...]"),
          ("train", "[DATA_DST]/lists/train-clean-100.lst,[DATA_DST]/lists/train-clean-360.lst,[DATA_DST]/lists/train-other-500.lst"),
          ("valid", "dev-clean:[DATA_DST]/lists/dev-clean.lst,dev-other:[DATA_DST]/lists/dev-other.lst"),
          ("lexicon", "[MODEL_DST]/am/lexicon_train+dev.txt"), ("arch", "network.arch"), ("tokens", "tokens.txt"), ("criterion", "asg"),
          ("lr", "0.6"), ("lrcrit", "0.006"), ("linseg", "1"), ("momentum", "0.8"), ("maxgradnorm", "0.2"), ("replabel", "2"),
          ("surround", "|"), ("onorm", "target"), ("sqnorm", "true"), ("mfsc", "true"), ("nthread", "6"), ("batchsize", "4"),
          ("transdiag", "4"), ("filterbanks", "40")]
    return ("# Training config for Librispeech using Gated ConvNets\n# Replace `[...]`, `[MODEL_DST]`, `[DATA_DST]` with appropriate paths\n"
            + "".join(f"--{k}={v}\n" for k, v in fl))
