"""Token dictionary, lexicon, target generation and the prediction -> letters / words remap of the Trainer (SURVEY 8f row f3:
the data formats either side of the hot path).  Host logic, no device code.

In-repo witnesses (call sites and file formats):
  token dictionary of a run          recipes/slimIPL/src/Train.cpp:235-251 (tokens file, then `<1>` .. `<replabel>`, then the
                                     blank LAST for CTC -- 28 letters + 2 replabels = the 30 classes of conv_glu (config 4),
                                     9997 word pieces + blank = the 9998 classes of TDS-CTC / Transformer-CTC)
  tokens.txt / lexicon formats       recipes/conv_glu/librispeech/prepare.py:59-84 (`|`, `'`, a..z; `word<TAB>w o r d |`),
                                     recipes/sota/2019 cfgs (`--wordseparator=_ --usewordpiece=true`, n-best spellings)
  evaluation remap                   recipes/slimIPL/src/Train.cpp:829-872 (viterbiPath -> tknPrediction2Ltr, tknTarget2Ltr,
                                     tkn2Wrd -> token / word edit distance)
The functions themselves (fl::lib::text::Dictionary, wrd2Target, packReplabels, unpackReplabels, tknPrediction2Ltr,
tknIdx2Ltr, tkn2Wrd) are un-vendored Flashlight ([UNVENDORED]): restated from their published behaviour, pinned here by
round trips and by the class counts above (tests/test_text.py).
"""
from typing import Dict, Iterable, List, Optional, Sequence

BLANK = "#"          # fl::pkg::speech::kBlankToken
UNK = "<unk>"


class Dictionary:
    """fl::lib::text::Dictionary: one entry per line; several whitespace-separated entries on a line share one index"""

    def __init__(self, source=None):
        self.entry2idx: Dict[str, int] = {}
        self.idx2entry: Dict[int, str] = {}
        if source is None:
            return
        lines = open(source).read().splitlines() if isinstance(source, str) else list(source)
        for line in lines:
            toks = line.split()
            if not toks:
                continue
            idx = len(self.idx2entry)
            for t in toks:
                if t in self.entry2idx:
                    raise ValueError(f"duplicate dictionary entry {t!r}")
                self.entry2idx[t] = idx
            self.idx2entry[idx] = toks[0]

    def add_entry(self, entry: str) -> int:
        if entry in self.entry2idx:
            raise ValueError(f"duplicate dictionary entry {entry!r}")
        idx = len(self.idx2entry)
        self.entry2idx[entry] = idx
        self.idx2entry[idx] = entry
        return idx

    def index_size(self) -> int:
        return len(self.idx2entry)

    def contains(self, entry: str) -> bool:
        return entry in self.entry2idx

    def get_index(self, entry: str) -> int:
        if entry not in self.entry2idx:
            raise KeyError(f"unknown dictionary entry {entry!r}")
        return self.entry2idx[entry]

    def get_entry(self, idx: int) -> str:
        return self.idx2entry[idx]


def replabel_token(r: int) -> str:
    return f"<{r}>"


def create_token_dict(tokens, criterion: str, replabel: int = 0) -> Dictionary:
    """the class inventory of a run (Train.cpp:235-251): tokens file, `<1>`..`<replabel>`, and for CTC the blank LAST"""
    d = Dictionary(tokens)
    for r in range(1, replabel + 1):
        d.add_entry(replabel_token(r))
    if criterion == "ctc":
        d.add_entry(BLANK)
    return d


def load_lexicon(source, max_words: int = -1) -> Dict[str, List[List[str]]]:
    """`word<TAB or space>tok tok ...` per line; a word may appear on several lines (n-best spellings, all kept, in file order).
    `max_words` is the reference's second argument of loadWords(FLAGS_lexicon, FLAGS_maxword): at most that many distinct words
    are kept (-1: all); reading stops at the first new word beyond the limit, as the reference's loop does"""
    lines = open(source).read().splitlines() if isinstance(source, str) else list(source)
    lex: Dict[str, List[List[str]]] = {}
    for line in lines:
        parts = line.split()
        if len(parts) < 2:
            continue
        if parts[0] not in lex and 0 <= max_words <= len(lex):
            break
        lex.setdefault(parts[0], []).append(parts[1:])
    return lex


def split_wrd(word: str) -> List[str]:
    return list(word)       # code points (fl::lib::splitWrd walks UTF-8 characters)


def wrd2target(words: Sequence[str], lexicon: Dict[str, List[List[str]]], token_dict: Dictionary, wordsep: str = "",
               fallback2ltr: bool = True, fallback_sep_left: bool = False, fallback_sep_right: bool = True,
               skip_unk: bool = False, sample_pct: float = 0.0, rng=None) -> List[str]:
    """transcription words -> token strings: the first spelling of the lexicon (or, with probability `sample_pct`, a random
    one: --sampletarget); out-of-lexicon words fall back to their letters with the word separator on the chosen side(s)
    when every letter is a token, else they are skipped (`skip_unk`) or an error"""
    out: List[str] = []
    for w in words:
        sp = lexicon.get(w)
        if sp:
            k = 0
            if sample_pct > 0 and rng is not None and len(sp) > 1 and rng.random() < sample_pct:
                k = int(rng.integers(0, len(sp)))
            out += sp[k]
            continue
        letters = split_wrd(w)
        if fallback2ltr and all(token_dict.contains(c) for c in letters):
            if fallback_sep_left and wordsep:
                out.append(wordsep)
            out += letters
            if fallback_sep_right and wordsep:
                out.append(wordsep)
        elif skip_unk:
            continue
        else:
            raise KeyError(f"word {w!r} is not in the lexicon and cannot be spelled with the token set")
    return out


def pack_replabels(tokens: Sequence[int], token_dict: Dictionary, max_reps: int) -> List[int]:
    """`a a a b` -> `a <2> b`: runs of one token become the token + the replabel counting the EXTRA repetitions (ASG cannot
    emit the same label twice in a row); runs longer than max_reps + 1 restart"""
    if not tokens or max_reps <= 0:
        return list(tokens)
    rep_idx = {r: token_dict.get_index(replabel_token(r)) for r in range(1, max_reps + 1)}
    out: List[int] = []
    prev, reps = -1, 0
    for t in tokens:
        if t == prev and reps < max_reps:
            reps += 1
        else:
            if reps > 0:
                out.append(rep_idx[reps])
                reps = 0
            out.append(t)
            prev = t
    if reps > 0:
        out.append(rep_idx[reps])
    return out


def unpack_replabels(tokens: Sequence[int], token_dict: Dictionary, max_reps: int) -> List[int]:
    if not tokens or max_reps <= 0:
        return list(tokens)
    value = {token_dict.get_index(replabel_token(r)): r for r in range(1, max_reps + 1)}
    out: List[int] = []
    prev = -1
    for t in tokens:
        if t not in value:
            out.append(t)
            prev = t
        elif prev != -1:
            out += [prev] * value[t]
            prev = -1
    return out


def target_indices(words: Sequence[str], lexicon, token_dict: Dictionary, criterion: str, replabel: int = 0, wordsep: str = "",
                   surround: str = "", **kw) -> List[int]:
    """one transcription -> the int32 target row of the criterion (before -1 padding)"""
    toks = wrd2target(words, lexicon, token_dict, wordsep, **kw)
    if surround:
        toks = [surround] + toks + [surround]
    idx = [token_dict.get_index(t) for t in toks]
    if criterion == "asg" and replabel > 0:
        idx = pack_replabels(idx, token_dict, replabel)
    return idx


def pad_targets(rows: Sequence[Sequence[int]], length: Optional[int] = None):
    """[B][L] int32, -1 padded (the layout w2l_*_forward takes)"""
    import numpy as np
    L = max([len(r) for r in rows] + [1]) if length is None else length
    out = np.full((len(rows), L), -1, np.int32)
    for b, r in enumerate(rows):
        out[b, :min(L, len(r))] = list(r)[:L]
    return out


def uniq(tokens: Iterable[int]) -> List[int]:
    out: List[int] = []
    for t in tokens:
        if not out or out[-1] != t:
            out.append(t)
    return out


def tkn_idx_to_ltr(tokens: Sequence[int], token_dict: Dictionary, use_wordpiece: bool, wordsep: str) -> List[str]:
    out: List[str] = []
    for t in tokens:
        e = token_dict.get_entry(int(t))
        out += split_wrd(e) if use_wordpiece else [e]
    if out and wordsep:
        if out[0] == wordsep:
            out = out[1:]
        if out and out[-1] == wordsep:
            out = out[:-1]
    return out


def _remap_labels(tokens: List[int], token_dict: Dictionary, surround: str, replabel: int) -> List[int]:
    if replabel > 0:
        tokens = unpack_replabels(tokens, token_dict, replabel)
    if surround and token_dict.contains(surround):
        s = token_dict.get_index(surround)
        if tokens and tokens[-1] == s:
            tokens = tokens[:-1]
        if tokens and tokens[0] == s:
            tokens = tokens[1:]
    return tokens


def tkn_prediction_to_ltr(path: Sequence[int], token_dict: Dictionary, criterion: str, surround: str = "", replabel: int = 0,
                          use_wordpiece: bool = False, wordsep: str = "") -> List[str]:
    """a Viterbi path (one label per frame) -> letters: collapse repeated frames, drop the CTC blank, undo replabels"""
    toks = [int(t) for t in path if int(t) >= 0]
    if not toks:
        return []
    if criterion in ("ctc", "asg"):
        toks = uniq(toks)
    if criterion == "ctc":
        blank = token_dict.get_index(BLANK)
        toks = [t for t in toks if t != blank]
    toks = _remap_labels(toks, token_dict, surround, replabel if criterion == "asg" else 0)
    return tkn_idx_to_ltr(toks, token_dict, use_wordpiece, wordsep)


def tkn_target_to_ltr(target: Sequence[int], token_dict: Dictionary, criterion: str, surround: str = "", replabel: int = 0,
                      use_wordpiece: bool = False, wordsep: str = "") -> List[str]:
    toks = [int(t) for t in target if int(t) >= 0]      # -1 padding of the batch
    if not toks:
        return []
    toks = _remap_labels(toks, token_dict, surround, replabel if criterion == "asg" else 0)
    return tkn_idx_to_ltr(toks, token_dict, use_wordpiece, wordsep)


def tkn2wrd(letters: Sequence[str], wordsep: str) -> List[str]:
    words: List[str] = []
    cur = ""
    for t in letters:
        if t == wordsep:
            if cur:
                words.append(cur)
                cur = ""
        else:
            cur += t
    if cur:
        words.append(cur)
    return words


def edit_distance(a: Sequence, b: Sequence) -> int:
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


class EditDistanceMeter:
    """fl::EditDistanceMeter as the Trainer logs it: 100 * (ins + del + sub) / reference length over everything added"""

    def __init__(self):
        self.errors = 0
        self.length = 0

    def add(self, hyp: Sequence, ref: Sequence):
        self.errors += edit_distance(hyp, ref)
        self.length += len(ref)

    def value(self) -> float:
        return 100.0 * self.errors / self.length if self.length else 0.0


def eval_output(paths, targets, token_dict: Dictionary, criterion: str, surround: str = "", replabel: int = 0,
                use_wordpiece: bool = False, wordsep: str = ""):
    """Train.cpp:829-872 for a batch: (token error rate, word error rate) meters over Viterbi paths [B][T] and targets [B][L]"""
    ter, wer = EditDistanceMeter(), EditDistanceMeter()
    for p, t in zip(paths, targets):
        lp = tkn_prediction_to_ltr(p, token_dict, criterion, surround, replabel, use_wordpiece, wordsep)
        lt = tkn_target_to_ltr(t, token_dict, criterion, surround, replabel, use_wordpiece, wordsep)
        ter.add(lp, lt)
        wer.add(tkn2wrd(lp, wordsep), tkn2wrd(lt, wordsep))
    return ter, wer
