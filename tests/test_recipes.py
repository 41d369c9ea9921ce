"""The reference's recipes must load unchanged: every .arch file parses with the reference's
own grammar/arity rules, every train .cfg parses as a gflags file, and the arch generators
in wav2letter_amd.recipes reproduce the reference files line for line."""
import glob
import os

import pytest

REF = "/root/reference/recipes"
need_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


def _lines(text):
    return [l.strip() for l in text.splitlines() if l.strip() and not l.strip().startswith("#")]


@need_ref
def test_generators_reproduce_reference_arch_files():
    from wav2letter_amd import recipes
    assert _lines(recipes.tds_ctc_arch()) == _lines(open(f"{REF}/sota/2019/am_arch/am_tds_ctc.arch").read())
    assert _lines(recipes.conv_glu_librispeech_arch()) == _lines(open(f"{REF}/conv_glu/librispeech/network.arch").read())
    assert _lines(recipes.conv_glu_wsj_arch()) == _lines(open(f"{REF}/conv_glu/wsj/network.arch").read())
    assert _lines(recipes.streaming_tds_arch()) == _lines(
        open(f"{REF}/streaming_convnets/librispeech/am_500ms_future_context.arch").read())
    assert _lines(recipes.tds_ctc_librivox_arch()) == _lines(open(f"{REF}/sota/2019/am_arch/am_tds_ctc_librivox.arch").read())
    assert _lines(recipes.transformer_ctc_arch()) == _lines(open(f"{REF}/sota/2019/am_arch/am_transformer_ctc.arch").read())


@need_ref
def test_generators_reproduce_reference_train_cfgs():
    """the flags files the `Train` binary is exercised with on the GPU box ARE the reference's (line for line)"""
    from wav2letter_amd import recipes
    assert _lines(recipes.tds_ctc_train_cfg()) == _lines(open(f"{REF}/sota/2019/librispeech/train_am_tds_ctc.cfg").read())
    assert _lines(recipes.conv_glu_train_cfg()) == _lines(open(f"{REF}/conv_glu/librispeech/train.cfg").read())
    assert _lines(recipes.transformer_ctc_train_cfg()) == _lines(open(f"{REF}/sota/2019/librispeech/train_am_transformer_ctc.cfg").read())


@need_ref
def test_all_reference_arch_files_parse():
    from wav2letter_amd.trainer import arch_check
    files = sorted(glob.glob(f"{REF}/**/*.arch", recursive=True))
    assert len(files) >= 30
    from wav2letter_amd._lib import W2LInvalidArgument
    rejected = []
    for f in files:
        try:
            n = arch_check(open(f).read(), 80, 9998)
        except W2LInvalidArgument as e:
            assert "LN 3" in str(e), (f, e)
            rejected.append(os.path.relpath(f, REF))
            continue
        assert n == len(_lines(open(f).read())), f
    # the reference's own builder rejects exactly these two pre-migration files (SequentialBuilder.cpp:366-375)
    assert sorted(rejected) == ["self_training/librispeech/am/baseline.arch", "seq2seq_tds/librispeech/network.arch"]


@need_ref
def test_all_reference_train_cfgs_parse():
    from wav2letter_amd.trainer import flags_check
    files = [f for f in glob.glob(f"{REF}/**/*.cfg", recursive=True)]
    assert len(files) >= 100
    for f in files:
        assert flags_check(open(f).read()) > 0, f


def test_grammar_errors_match_reference_behaviour():
    from wav2letter_amd._lib import W2LInvalidArgument
    from wav2letter_amd.trainer import arch_check
    for bad in ["V 1 2 3", "RO 0 1 2", "FOO 1 2", "TDS 10", "LN 3", "DO", "PD 0 1", "C2 1 2 3"]:
        with pytest.raises(W2LInvalidArgument):
            arch_check(bad, 80, 30)
    assert arch_check("# comment\n\nV -1 NFEAT 1 0\nL NFEAT NLABEL\n", 80, 30) == 2


def test_headline_archs_build_and_have_the_published_parameter_counts():
    from wav2letter_amd import recipes
    from wav2letter_amd.trainer import Trainer
    t = Trainer(recipes.tds_ctc_arch(), 80, 9998, "ctc", device="cpu")
    n = sum(n for _, n, _ in t.param_table())
    # 203.4 M parameters (LayerNorm scalars counted as 2 per LN here)
    assert abs(n - 203.4e6) < 0.1e6, n
    t = Trainer(recipes.conv_glu_librispeech_arch(), 40, 30, "asg", transdiag=4.0, device="cpu")
    n = sum(n for _, n, _ in t.param_table())
    assert abs(n - 208.9e6) < 0.1e6, n


def test_librivox_arch_with_two_dimensional_subsampling_convolutions_builds():
    """am_tds_ctc_librivox.arch: `C2 cin cout 21 3 2 1 -1 -1` (21 frames x 3 mel rows) -- parameter shapes in the
    reference's order and dims, and the network plans at a recipe-sized batch"""
    from wav2letter_amd import recipes
    from wav2letter_amd.trainer import Trainer
    t = Trainer(recipes.tds_ctc_librivox_arch(), 80, 9998, "ctc", device="cpu")
    table = t.param_table()
    assert table[0][1] == 21 * 3 * 1 * 16 and table[1][1] == 16          # first C2: weight (kw, kh, cin, cout), bias
    convs = [n for name, n, _ in table if name == "conv.w"]
    assert 21 * 3 * 16 * 16 in convs and 21 * 3 * 16 * 32 in convs and 21 * 1 * 32 * 48 in convs
    assert t.plan(4, 800, 40) == 100                                      # three stride-2 stages, the fourth keeps T
    from wav2letter_amd._lib import W2LInvalidArgument
    with pytest.raises(W2LInvalidArgument):                               # even kh / stride on the mel axis: not a recipe geometry
        Trainer("V -1 NFEAT 1 0\nC2 1 4 5 2 1 1 -1 -1\n", 8, 5, "ctc", device="cpu")
    with pytest.raises(W2LInvalidArgument):
        Trainer("V -1 NFEAT 1 0\nC2 1 4 5 3 1 2 -1 -1\n", 8, 5, "ctc", device="cpu")


def test_transformer_ctc_arch_builds():
    """am_transformer_ctc.arch (BASELINE config 5): parameter table in the reference's order (position table, w1, w2, wq,
    wk, wv, wf, norm1, norm2 per block -- TransformerCPC.cpp:79-94), parameter count, and the frames left after the three
    max-pools"""
    from wav2letter_amd import recipes
    from wav2letter_amd.trainer import Trainer
    t = Trainer(recipes.transformer_ctc_arch(), 80, 9998, "ctc", device="cpu")
    table = t.param_table()
    names = [n for n, _, _ in table]
    i = names.index("tr.posemb")
    assert names[i:i + 15] == ["tr.posemb", "tr.w1.w", "tr.w1.b", "tr.w2.w", "tr.w2.b", "tr.wq.w", "tr.wq.b", "tr.wk.w", "tr.wk.b",
                               "tr.wv.w", "tr.wv.b", "tr.wf.w", "tr.wf.b", "tr.norm1.weight+bias", "tr.norm2.weight+bias"]
    assert table[i][1] == (2 * 460 - 1) * 256 and table[i + 1][1] == 1024 * 4096
    assert names.count("tr.posemb") == 24
    n = sum(m for name, m, _ in table) - 24 * 2          # LayerNorm pairs are stored as 2 floats, counted as 2 params each
    assert abs(n - 322.6e6) < 0.5e6, n
    assert t.plan(2, 1500, 40) == 188
    from wav2letter_amd._lib import W2LInvalidArgument
    with pytest.raises(W2LInvalidArgument):               # a Transformer block wants (C, T, B, 1): the Reorder is missing
        Trainer("V -1 1 NFEAT 0\nTR 80 320 4 460 0.2\n", 80, 30, "ctc", device="cpu")
    with pytest.raises(W2LInvalidArgument):               # pre-LayerNorm variant: not a BASELINE recipe
        Trainer("V -1 1 NFEAT 0\nRO 2 0 3 1\nTR 80 320 4 460 0.2 0.0 1\n", 80, 30, "ctc", device="cpu")


def test_unsupported_layers_fail_loudly():
    from wav2letter_amd._lib import W2LInvalidArgument
    from wav2letter_amd.trainer import Trainer
    with pytest.raises(W2LInvalidArgument):
        Trainer("V -1 1 NFEAT 0\nRO 2 0 3 1\nCFR 80 320 4 460 0.2 0.1 31\n", 80, 30, "ctc", device="cpu")
    with pytest.raises(W2LInvalidArgument):
        Trainer("V -1 NFEAT 1 0\nL 80 NLABEL\n", 80, 30, "seq2seq", device="cpu")


def test_layer_objects_are_their_arch_lines(tmp_path):
    """include/fl_compat/flashlight.h: fl::View / LayerNorm / Conv2D / GatedLinearUnit / Dropout / Reorder / Transformer / Linear /
    TDSBlock / Pool2D / WeightNorm constructed the way the reference's model plugin constructs them
    (recipes/slimIPL/100h_supervised.cpp:16-33) are the corresponding lines of the arch grammar
    (cpc/SequentialBuilder.cpp:92-626); a combination the grammar does not have throws, a lone layer object does not run.
    tests/cpp/layers_test.cpp, plain g++ against the library (loads without a GPU)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "wav2letter_amd")
    if not os.path.exists(os.path.join(lib, "libw2l_hip.so")):
        pytest.skip("library not built")
    exe = str(tmp_path / "layers_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "layers_test.cpp"),
                    "-o", exe, "-L" + lib, "-lw2l_hip", "-Wl,-rpath," + lib], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.strip() == "ok"
