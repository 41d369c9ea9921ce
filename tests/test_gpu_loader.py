"""The prefetching input pipeline (wav2letter_amd/loader.py) on the device: batches come out in list
order, bit-identical to featurising the same padded audio directly, with the utterances' sample counts; a decode error
surfaces in the consumer; an abandoned iteration shuts the threads down."""
import os
import threading
import wave

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_wav(path, samples):
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(samples.astype("<i2").tobytes())


@pytest.mark.parametrize("fmt", ["wav_int16", "wav_float", "flac"])
def test_prefetch_loader_matches_direct_featurisation(tmp_path, fmt):
    from wav2letter_amd import data
    from wav2letter_amd.features import Mfsc
    from wav2letter_amd.loader import PrefetchLoader, read_audio_int16
    rng = np.random.default_rng(3)
    lens = [16000 + 517 * i for i in range(11)]
    audio, paths = [], []
    for i, n in enumerate(lens):
        a = (rng.normal(size=n) * 4000).astype(np.int16)
        audio.append(a)
        if fmt == "flac":     # the LibriSpeech container, through the library's decoder (csrc/host/flac.cpp)
            from tests import flac_encode as FE
            p = str(tmp_path / f"u{i}.flac")
            with open(p, "wb") as f:
                f.write(FE.encode(a.astype(np.int64), kind="fixed", order=2, porder=2, block=4096))
        else:
            p = str(tmp_path / f"u{i}.wav")
            _write_wav(p, a)
        paths.append(p)
    samples = data.parse_list("\n".join(f"id{i} {p} {1000.0 * n / 16000:.1f} a b" for i, (p, n) in enumerate(zip(paths, lens))))
    batches = data.batches(range(len(samples)), [s.duration_ms for s in samples], 4, sort_by_length=True)
    assert [len(b) for b in batches] == [4, 4, 3]
    mfsc = Mfsc(num_filters=40)
    loader = PrefetchLoader(samples, batches, mfsc, workers=3, depth=2, read=None if fmt == "wav_float" else read_audio_int16)
    got = []
    for feats, sizes, ids in loader:
        got.append((feats.clone(), sizes.clone(), list(ids)))
    assert [g[2] for g in got] == batches
    for feats, sizes, ids in got:
        longest = max(lens[i] for i in ids)
        host = np.zeros((len(ids), longest), np.float32)
        for b, i in enumerate(ids):
            host[b, :lens[i]] = audio[i].astype(np.float32) / 32768.0
        want = mfsc(torch.tensor(host).cuda())
        assert torch.equal(feats, want)
        assert sizes.cpu().tolist() == [float(lens[i]) for i in ids]
    # a second pass over the same loader object gives the same batches
    again = [ids for _f, _s, ids in loader]
    assert again == batches


def test_prefetch_loader_errors_and_shutdown(tmp_path):
    from wav2letter_amd.features import Mfsc
    from wav2letter_amd.loader import PrefetchLoader
    good = str(tmp_path / "a.wav")
    _write_wav(good, np.zeros(16000))
    mfsc = Mfsc(num_filters=40)
    before = threading.active_count()
    with pytest.raises(Exception):
        for _ in PrefetchLoader([good, str(tmp_path / "missing.wav")], [[0], [1], [0]], mfsc):
            pass
    other = str(tmp_path / "b.wav")
    with wave.open(other, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000)
        w.writeframes(np.zeros(8000, "<i2").tobytes())
    with pytest.raises(ValueError):
        for _ in PrefetchLoader([other], [[0]], mfsc):
            pass
    it = iter(PrefetchLoader([good], [[0]] * 50, mfsc))
    next(it)
    it.close()                      # abandoned after one batch: the generator's finally stops the assembler thread
    assert threading.active_count() <= before + 1
    with pytest.raises(ValueError):
        PrefetchLoader([good], [[0]], mfsc, depth=1)
