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
