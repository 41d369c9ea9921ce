"""Host side of the input pipeline: list files, rank partitioning, length-bucketed batching."""
import pytest

from wav2letter_amd import data


def test_parse_list_librispeech_lines():
    text = ("1272-128104-0000 /data/LibriSpeech/dev-clean/1272/128104/1272-128104-0000.flac 5855.0 mister quilter is the apostle\n"
            "\n"
            "1272-128104-0001 /data/x.flac 4815.0 nor is mister quilter's manner less interesting\n"
            "utt-without-text /data/y.flac 1000\n")
    s = data.parse_list(text)
    assert [x.sample_id for x in s] == ["1272-128104-0000", "1272-128104-0001", "utt-without-text"]
    assert s[0].duration_ms == 5855.0 and s[0].transcript == "mister quilter is the apostle"
    assert s[1].transcript.endswith("less interesting") and s[2].transcript == ""
    with pytest.raises(ValueError):
        data.parse_list("only two\n")
    with pytest.raises(ValueError):
        data.parse_list("id path notanumber text\n")


@pytest.mark.parametrize("n,world,bs", [(64, 8, 4), (70, 8, 4), (67, 4, 8), (5, 2, 4), (0, 2, 4), (33, 1, 32)])
def test_partition_round_robin_is_a_partition(n, world, bs):
    """ranks get disjoint index sets of equal size; whole global batches are covered completely, rank r takes the r-th
    slice of each; nothing is sampled twice"""
    parts = [data.partition_round_robin(n, r, world, bs) for r in range(world)]
    sizes = {len(p) for p in parts}
    assert len(sizes) == 1                                   # same number of samples (hence batches) on every rank
    flat = [i for p in parts for i in p]
    assert len(flat) == len(set(flat)) and all(0 <= i < n for i in flat)
    n_global = n // (world * bs)
    assert set(range(n_global * world * bs)) <= set(flat)     # every sample of a whole global batch is used
    for r, p in enumerate(parts):
        for g in range(n_global):
            assert p[g * bs:(g + 1) * bs] == list(range(g * world * bs + r * bs, g * world * bs + (r + 1) * bs))
    with pytest.raises(ValueError):
        data.partition_round_robin(10, 3, 2, 4)


def test_batches_by_length_and_duration_cap():
    dur = [5000, 1000, 3000, 2000, 9000, 1500, 2500]
    idx = list(range(7))
    plain = data.batches(idx, dur, 3)
    assert plain == [[0, 1, 2], [3, 4, 5], [6]]
    srt = data.batches(idx, dur, 3, sort_by_length=True)
    assert srt == [[1, 5, 3], [6, 2, 0], [4]]
    capped = data.batches(idx, dur, 4, max_duration_ms=9000, sort_by_length=True)
    for b in capped:
        assert max(dur[i] for i in b) * len(b) <= 9000 or len(b) == 1
    assert sorted(i for b in capped for i in b) == idx


@pytest.mark.parametrize("n,world,bs", [(70, 8, 4), (67, 4, 8), (5, 2, 4), (3, 8, 4), (0, 2, 4), (33, 1, 32), (71, 8, 4)])
def test_partition_round_robin_allow_empty_covers_every_sample(n, world, bs):
    """validation sets (allowEmpty = true in the reference): the union over ranks is range(n), disjoint, and the
    first rest % world ranks carry the extra sample of the tail batch (fl::partitionByRoundRobin)"""
    parts = [data.partition_round_robin(n, r, world, bs, allow_empty=True) for r in range(world)]
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(n))
    sizes = [len(p) for p in parts]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_read_audio_wav_and_raw_and_batch_padding(tmp_path):
    """the loader side of a .lst line: WAV PCM (16 / 24 bit, stereo -> mono) and raw PCM back as float32 in [-1, 1), a padded
    batch with its input sizes; containers without a decoder here are refused (FLAC has its own tests: test_flac.py)"""
    import wave
    import numpy as np
    import pytest
    from wav2letter_amd import data
    t = np.arange(1600) / 16000.0
    sig = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    pcm = np.round(sig * 32767).astype("<i2")
    with wave.open(str(tmp_path / "a.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    a, rate = data.read_audio(str(tmp_path / "a.wav"))
    assert rate == 16000 and a.dtype == np.float32 and a.shape == (1600,) and np.abs(a - sig).max() < 1e-4
    st = np.stack([pcm, -pcm], axis=1)                                   # stereo: the channels cancel
    with wave.open(str(tmp_path / "s.wav"), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(8000); w.writeframes(st.tobytes())
    s, rate = data.read_audio(str(tmp_path / "s.wav"))
    assert rate == 8000 and s.shape == (1600,) and np.abs(s).max() < 1e-6
    v24 = (np.round(sig * (2 ** 23 - 1)).astype(np.int32)) & 0xFFFFFF
    b24 = np.stack([v24 & 0xFF, (v24 >> 8) & 0xFF, (v24 >> 16) & 0xFF], axis=1).astype(np.uint8)
    with wave.open(str(tmp_path / "h.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(3); w.setframerate(16000); w.writeframes(b24.tobytes())
    h, _ = data.read_audio(str(tmp_path / "h.wav"))
    assert np.abs(h - sig).max() < 1e-6
    pcm.tofile(str(tmp_path / "r.raw"))
    r, _ = data.read_audio(str(tmp_path / "r.raw"))
    assert np.array_equal(r, a)
    with pytest.raises(ValueError):
        data.read_audio(str(tmp_path / "x.ogg"))
    batch, sizes = data.pad_batch([a, a[:700]])
    assert batch.shape == (2, 1600) and list(sizes) == [1600.0, 700.0] and (batch[1, 700:] == 0).all()


def test_cpp_header_mirror_of_list_and_partitioning(tmp_path):
    """include/fl_compat/data.h through tests/cpp/data_test.cpp (g++, no device code); the worked partition below is checked
    against the Python implementation too"""
    import os
    import subprocess
    from wav2letter_amd import data
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "data_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(root, "tests", "cpp", "data_test.cpp"), "-o", exe], check=True)
    assert "data ok" in subprocess.run([exe], capture_output=True, text=True, check=True).stdout
    assert data.partition_round_robin(20, 1, 2, 4) == [4, 5, 6, 7, 12, 13, 14, 15, 18, 19]


def test_read_audio_int16_keeps_pcm_words(tmp_path):
    """the prefetching loader's decoder (wav2letter_amd/loader.py): 16-bit mono WAV / raw PCM stay int16 (scaled on the device),
    equal to data.read_audio * 32768; anything else falls back to the float decoder"""
    import wave
    import numpy as np
    from wav2letter_amd import data
    from wav2letter_amd.loader import read_audio_int16
    a = (np.arange(-500, 500) * 37).astype(np.int16)
    p = str(tmp_path / "a.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(a.astype("<i2").tobytes())
    got, rate = read_audio_int16(p)
    assert got.dtype == np.int16 and rate == 16000 and np.array_equal(got, a)
    f, _ = data.read_audio(p)
    assert np.array_equal(f, a.astype(np.float32) / 32768.0)
    a.astype("<i2").tofile(str(tmp_path / "b.raw"))
    got, rate = read_audio_int16(str(tmp_path / "b.raw"))
    assert got.dtype == np.int16 and np.array_equal(got, a)
    st = str(tmp_path / "s.wav")
    with wave.open(st, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.stack([a, a], 1).astype("<i2").tobytes())
    got, _ = read_audio_int16(st)
    assert got.dtype == np.float32 and len(got) == len(a)
