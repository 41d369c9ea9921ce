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


def read_audio(path: str):
    """one utterance as float32 samples in [-1, 1) plus its sample rate.  The reference decodes through libsndfile
    (fl::pkg::speech::loadSound, un-vendored; the recipes' lists point at .flac / .wav files): here RIFF / WAV PCM (8 / 16 /
    24 / 32 bit, via the standard library) and headerless float32 `.f32` / int16 `.raw` / `.pcm` at 16 kHz; FLAC needs a
    decoder this image does not carry and raises.  Multi-channel files are averaged to mono, as the Trainer's `--channels=1`
    pipelines expect a single channel."""
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
    if low.endswith(".f32"):
        return np.fromfile(path, "<f4"), 16000
    if low.endswith(".raw") or low.endswith(".pcm"):
        return np.fromfile(path, "<i2").astype(np.float32) / 32768.0, 16000
    raise ValueError(f"{path}: no decoder for this container in this image (WAV / raw PCM only; the reference uses libsndfile)")


def pad_batch(audios):
    """[B][max samples] float32, zero padded, plus the per-utterance sample counts (the `inputSizes` of the batch)"""
    import numpy as np
    n = np.array([len(a) for a in audios], np.float32)
    out = np.zeros((len(audios), int(n.max()) if len(audios) else 0), np.float32)
    for b, a in enumerate(audios):
        out[b, :len(a)] = a
    return out, n
