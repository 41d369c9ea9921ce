"""List files and batch partitioning of the input pipeline (host side, no device code).

* `.lst` format: one sample per line, `id path duration_ms transcript...` -- written by the recipes' data preparation
  (data/librispeech/utils.py:36-46, read back at :49-57) and consumed through --train / --valid / --datadir
  (recipes/slimIPL/src/Train.cpp:327-339).
* Partitioning across ranks follows Flashlight's `partitionByRoundRobin` [UNVENDORED, recalled]: the sample list is
  cut into GLOBAL batches of world * batchsize consecutive samples, rank r takes the r-th slice of batchsize samples
  of every global batch; what is left over after the last whole global batch is split evenly, in order.  Every rank
  therefore runs the same number of full batches (the all-reduce never waits for a straggler).
* Length bucketing (--batching_strategy / --batching_max_duration, Train.cpp:337-338): sort by duration so that a
  padded batch wastes few frames; `max_duration` caps the padded audio time of one batch.
"""
from dataclasses import dataclass
from typing import List, Sequence


@dataclass
class Sample:
    sample_id: str
    path: str
    duration_ms: float
    transcript: str


def parse_list(text: str) -> List[Sample]:
    """the lines of a .lst file (blank lines ignored; a sample without transcript has an empty one)"""
    out = []
    for ln, line in enumerate(text.splitlines(), 1):
        line = line.strip()
        if not line:
            continue
        parts = line.split(" ", 3)
        if len(parts) < 3:
            raise ValueError(f"list line {ln}: expected 'id path duration [transcript]', got {line!r}")
        try:
            dur = float(parts[2])
        except ValueError:
            raise ValueError(f"list line {ln}: duration {parts[2]!r} is not a number") from None
        out.append(Sample(parts[0], parts[1], dur, parts[3] if len(parts) > 3 else ""))
    return out


def read_list(path: str) -> List[Sample]:
    with open(path, "r") as f:
        return parse_list(f.read())


def partition_round_robin(n_samples: int, rank: int, world: int, batch_size: int, allow_empty: bool = False) -> List[int]:
    """sample indices of rank `rank` (see the module docstring)"""
    if not (0 <= rank < world) or batch_size <= 0 or n_samples < 0:
        raise ValueError("partition_round_robin: bad arguments")
    per_global = world * batch_size
    n_global = n_samples // per_global
    out = []
    for g in range(n_global):
        base = g * per_global + rank * batch_size
        out.extend(range(base, base + batch_size))
    rest = n_samples - n_global * per_global
    if rest >= world or (allow_empty and rest > 0):
        # the tail batch (fl::partitionByRoundRobin, un-vendored, as recalled): equal shares of rest // world; with
        # allow_empty (validation sets) the first rest % world ranks take one sample more, so that the union over
        # ranks is every sample; without it the remainder is dropped so that all ranks see equal batch counts
        per, remaining = divmod(rest, world)
        base = n_global * per_global + rank * per
        if allow_empty:
            base += min(rank, remaining)
            per += 1 if rank < remaining else 0
        out.extend(range(base, base + per))
    return out


def batches(indices: Sequence[int], durations_ms: Sequence[float], batch_size: int, max_duration_ms: float = 0.0,
            sort_by_length: bool = False) -> List[List[int]]:
    """consecutive batches of `batch_size` samples; with `sort_by_length` the rank's samples are ordered by duration first
    (padded batches waste few frames); `max_duration_ms` > 0 additionally closes a batch when its padded duration
    (longest sample x batch members) would exceed the cap"""
    idx = list(indices)
    if sort_by_length:
        idx.sort(key=lambda i: (durations_ms[i], i))
    out, cur, longest = [], [], 0.0
    for i in idx:
        d = durations_ms[i]
        too_long = max_duration_ms > 0 and cur and max(longest, d) * (len(cur) + 1) > max_duration_ms
        if len(cur) == batch_size or too_long:
            out.append(cur)
            cur, longest = [], 0.0
        cur.append(i)
        longest = max(longest, d)
    if cur:
        out.append(cur)
    return out


def decode_flac(data: bytes):
    """a FLAC file's bytes -> (int32 samples [n][channels], sample rate, bits per sample): wav2letter_amd/csrc/host/flac.cpp through the
    C ABI (w2l_flac_info / w2l_flac_decode) -- frame CRCs and the STREAMINFO MD5 of the decoded audio are verified there; a
    malformed or corrupted stream raises ValueError with the decoder's message"""
    import ctypes as C
    import numpy as np
    from . import _lib
    L = _lib.lib()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    rate, ch, bps, total = C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint64(0)
    if L.w2l_flac_info(buf, len(data), C.byref(rate), C.byref(ch), C.byref(bps), C.byref(total)) != 0:
        raise ValueError(L.w2l_flac_last_error().decode())
    cap = total.value if total.value else max(1, len(data) * 8)       # unknown length: a frame never codes a sample in under a bit
    out = np.empty((cap, ch.value), np.int32)
    done, md5 = C.c_uint64(0), C.c_int(0)
    if L.w2l_flac_decode(buf, len(data), out.ctypes.data_as(C.c_void_p), cap, C.byref(done), C.byref(md5)) != 0:
        raise ValueError(L.w2l_flac_last_error().decode())
    return out[:done.value], rate.value, bps.value


def read_audio(path: str):
    import numpy as np
    low = path.lower()
    if low.endswith(".wav"):
        import wave
        with wave.open(path, "rb") as w:
            nch, width, rate, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
        if width == 1:
            a = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        elif width == 2:
            a = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        elif width == 3:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            a = (np.where(v >= 1 << 23, v - (1 << 24), v)).astype(np.float32) / float(1 << 23)
        elif width == 4:
            a = np.frombuffer(raw, "<i4").astype(np.float32) / float(1 << 31)
        else:
            raise ValueError(f"{path}: unsupported sample width {width}")
        if nch > 1:
            a = a.reshape(-1, nch).mean(axis=1)
        return np.ascontiguousarray(a, np.float32), rate
    if low.endswith(".flac"):
        with open(path, "rb") as f:
            pcm, rate, bps = decode_flac(f.read())
        a = pcm.astype(np.float32) / float(1 << (bps - 1))
        return np.ascontiguousarray(a.mean(axis=1) if a.shape[1] > 1 else a[:, 0], np.float32), rate
    if low.endswith(".f32"):
        return np.fromfile(path, "<f4"), 16000
    if low.endswith(".raw") or low.endswith(".pcm"):
        return np.fromfile(path, "<i2").astype(np.float32) / 32768.0, 16000
    raise ValueError(f"{path}: no decoder for this container (WAV, FLAC and raw PCM are read; the reference uses libsndfile)")


def pad_batch(audios):
    """[B][max samples] float32, zero padded, plus the per-utterance sample counts (the `inputSizes` of the batch)"""
    import numpy as np
    n = np.array([len(a) for a in audios], np.float32)
    out = np.zeros((len(audios), int(n.max()) if len(audios) else 0), np.float32)
    for b, a in enumerate(audios):
        out[b, :len(a)] = a
    return out, n
