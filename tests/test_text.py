"""Token dictionary, lexicon, target generation and the evaluation remap (wav2letter_amd/text.py): the host
logic either side of the criteria, pinned by the class counts of the BASELINE recipes, round trips and worked examples."""
import numpy as np
import pytest

from wav2letter_amd import text

LETTERS = ["|", "'"] + [chr(c) for c in range(ord("a"), ord("z") + 1)]          # recipes/conv_glu/librispeech/prepare.py:59-64


def test_token_dictionary_class_counts_of_the_recipes():
    """Train.cpp:235-251: 28 letters + 2 replabels = the 30 classes of conv_glu's ASG (config 4), no blank; 9997 word pieces
    + the blank LAST = the 9998 classes of the CTC recipes (configs 2, 3, 5)"""
    asg = text.create_token_dict(LETTERS, "asg", replabel=2)
    assert asg.index_size() == 30 and asg.get_index("<1>") == 28 and asg.get_index("<2>") == 29 and not asg.contains(text.BLANK)
    ctc = text.create_token_dict([f"_w{i}" for i in range(9997)], "ctc")
    assert ctc.index_size() == 9998 and ctc.get_index(text.BLANK) == 9997
    shared = text.Dictionary(["a A", "b"])                    # entries on one line share an index
    assert shared.get_index("a") == shared.get_index("A") == 0 and shared.get_entry(0) == "a" and shared.index_size() == 2
    with pytest.raises(ValueError):
        text.Dictionary(["a", "a"])


def test_letter_lexicon_targets_and_replabels():
    d = text.create_token_dict(LETTERS, "asg", replabel=2)
    lex = text.load_lexicon(["hello\th e l l o |", "aaa\ta a a |", "bee\tb e e |", "bee\tb e |"])
    assert lex["bee"] == [["b", "e", "e", "|"], ["b", "e", "|"]]
    tgt = text.target_indices(["hello", "aaa"], lex, d, "asg", replabel=2, wordsep="|")
    names = [d.get_entry(i) for i in tgt]
    assert names == ["h", "e", "l", "<1>", "o", "|", "a", "<2>", "|"]            # no label twice in a row
    assert all(tgt[i] != tgt[i + 1] for i in range(len(tgt) - 1))
    plain = text.target_indices(["hello", "aaa"], lex, d, "ctc", wordsep="|")    # CTC keeps the repeats
    assert [d.get_entry(i) for i in plain] == list("hello|aaa|")
    assert text.unpack_replabels(tgt, d, 2) == plain
    # a run longer than max_reps + 1 restarts: a a a a -> a <2> a
    a = d.get_index("a")
    assert [d.get_entry(i) for i in text.pack_replabels([a] * 4, d, 2)] == ["a", "<2>", "a"]
    # out-of-lexicon word: letters + separator on the right; unknown characters are an error unless skipped
    assert [d.get_entry(i) for i in text.target_indices(["zed"], lex, d, "ctc", wordsep="|")] == list("zed|")
    with pytest.raises(KeyError):
        text.target_indices(["z3d"], lex, d, "ctc", wordsep="|")
    assert text.target_indices(["z3d", "bee"], lex, d, "ctc", wordsep="|", skip_unk=True) == [d.get_index(c) for c in "bee|"]
    rows = text.pad_targets([tgt, plain[:3]])
    assert rows.dtype == np.int32 and rows.shape == (2, 9) and (rows[1, 3:] == -1).all()


def test_pack_unpack_round_trip_random():
    d = text.create_token_dict(LETTERS, "asg", replabel=3)
    rng = np.random.default_rng(0)
    for _ in range(200):
        seq = [int(v) for v in rng.integers(0, 4, size=rng.integers(0, 30))]
        packed = text.pack_replabels(seq, d, 3)
        assert all(packed[i] != packed[i + 1] for i in range(len(packed) - 1))
        assert text.unpack_replabels(packed, d, 3) == seq


def test_wordpiece_targets_sampling_and_remap():
    pieces = ["_the", "_c", "at", "_cat", "s", "_", "t", "h", "e", "c", "a"]
    d = text.create_token_dict(pieces, "ctc")
    lex = text.load_lexicon(["the _the", "cat _cat", "cat _c at", "cats _cat s"])
    assert [d.get_entry(i) for i in text.target_indices(["the", "cat"], lex, d, "ctc", wordsep="_")] == ["_the", "_cat"]
    rng = np.random.default_rng(1)
    seen = {tuple(text.wrd2target(["cat"], lex, d, "_", sample_pct=1.0, rng=rng)) for _ in range(40)}
    assert seen == {("_cat",), ("_c", "at")}                                     # --sampletarget picks among the n-best spellings
    # out-of-lexicon word with word pieces: separator on the LEFT
    assert text.wrd2target(["eat"], lex, d, "_", fallback_sep_left=True, fallback_sep_right=False) == ["_", "e", "a", "t"]
    # a CTC Viterbi path over frames: collapse, drop blanks, split pieces into letters, strip the leading separator
    b = d.get_index(text.BLANK)
    path = [b, d.get_index("_the"), d.get_index("_the"), b, b, d.get_index("_c"), d.get_index("at"), d.get_index("at"), b, d.get_index("s")]
    ltr = text.tkn_prediction_to_ltr(path, d, "ctc", use_wordpiece=True, wordsep="_")
    assert ltr == list("the_cats")
    assert text.tkn2wrd(ltr, "_") == ["the", "cats"]
    tgt = text.pad_targets([text.target_indices(["the", "cat"], lex, d, "ctc", wordsep="_")], 6)[0]
    assert text.tkn2wrd(text.tkn_target_to_ltr(tgt, d, "ctc", use_wordpiece=True, wordsep="_"), "_") == ["the", "cat"]
    ter, wer = text.eval_output([path], [tgt], d, "ctc", use_wordpiece=True, wordsep="_")
    assert ter.errors == 1 and ter.length == 7 and abs(ter.value() - 100.0 / 7) < 1e-9      # one inserted letter
    assert wer.errors == 1 and wer.length == 2 and wer.value() == 50.0


def test_asg_prediction_remap_with_replabels_and_surround():
    d = text.create_token_dict(LETTERS, "asg", replabel=2)
    i = d.get_index
    # frames: | h h e l <1> <1> o | |  -> collapse -> | h e l <1> o | -> unpack -> | h e l l o | -> trim the surround token
    path = [i("|"), i("h"), i("h"), i("e"), i("l"), i("<1>"), i("<1>"), i("o"), i("|"), i("|")]
    assert text.tkn_prediction_to_ltr(path, d, "asg", surround="|", replabel=2, wordsep="|") == list("hello")
    assert text.edit_distance("kitten", "sitting") == 3 and text.edit_distance([], [1, 2]) == 2
    m = text.EditDistanceMeter()
    m.add(list("abc"), list("abd"))
    m.add([], list("xy"))
    assert m.value() == 100.0 * 3 / 5


def test_generated_recipe_tokens_match_the_prepare_script():
    """the token file recipes/conv_glu/librispeech/prepare.py writes (when the reference tree is present)"""
    import os
    p = "/root/reference/recipes/conv_glu/librispeech/prepare.py"
    if not os.path.exists(p):
        pytest.skip("reference tree not present")
    src = open(p).read()
    assert 'fout.write("|\\n")' in src and "fout.write(\"'\\n\")" in src and 'range(ord("a"), ord("z") + 1)' in src
    assert '"{word}\\t{tokens} |\\n"' in src          # the lexicon line format load_lexicon parses


def test_cpp_header_mirror_compiles_and_agrees(tmp_path):
    """include/fl_compat/text.h (the same logic on the reference's C++ names) through tests/cpp/text_test.cpp: plain g++, no
    device code -- the worked examples of this file, asserted in C++"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "text_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(root, "tests", "cpp", "text_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, check=True)
    assert "text ok" in out.stdout
