"""The FLAC decoder of the input pipeline (wav2letter_amd/csrc/host/flac.cpp) through the C ABI: bit-exact
round trips against the specification-level encoder of tests/flac_encode.py over every subframe type, residual coding, stereo
decorrelation and header code, self-verification (frame CRC-8 / CRC-16, STREAMINFO MD5), and refusal of corrupted streams.
Host code: runs without a GPU (the library loads on CPU boxes)."""
import numpy as np
import pytest

from tests import flac_encode as FE


def _speech_like(rng, n, ch=1, bps=16):
    t = np.arange(n)
    x = np.zeros((n, ch))
    for c in range(ch):
        x[:, c] = (0.3 * np.sin(2 * np.pi * (180 + 40 * c) * t / 16000) + 0.2 * np.sin(2 * np.pi * 1310 * t / 16000 + c)
                   + 0.05 * rng.normal(size=n)) * np.hanning(n)
    return np.round(x * (1 << (bps - 2))).astype(np.int64)


def _roundtrip(x, bps, **kw):
    from wav2letter_amd import data
    blob = FE.encode(x, bps=bps, **kw)
    pcm, rate, gb = data.decode_flac(blob)
    assert gb == bps and rate == kw.get("rate", 16000)
    want = np.asarray(x, np.int64)
    if want.ndim == 1:
        want = want[:, None]
    assert pcm.shape == want.shape and np.array_equal(pcm.astype(np.int64), want)
    return blob


@pytest.mark.parametrize("kind,order,porder,method", [("verbatim", 0, 0, 0), ("fixed", 0, 0, 0), ("fixed", 1, 2, 0), ("fixed", 2, 3, 0),
                                                      ("fixed", 3, 4, 1), ("fixed", 4, 0, 1), ("lpc", 1, 0, 0), ("lpc", 3, 2, 0),
                                                      ("lpc", 8, 3, 0), ("lpc", 12, 1, 1), ("lpc", 32, 0, 0)])
def test_flac_subframe_types_and_residual_codings(kind, order, porder, method):
    rng = np.random.default_rng(order * 7 + porder)
    x = _speech_like(rng, 9000)
    lpc = None
    if kind == "lpc":
        prec, shift = 12, 9
        base = {1: [0.95], 3: [2.6, -2.4, 0.78], 8: [1.9, -1.1, 0.3, -0.2, 0.1, -0.05, 0.02, 0.01]}.get(order)
        if base is None:
            base = list(rng.normal(size=order) * 0.2)
            base[0] += 1.0
        lpc = (prec, shift, [int(round(c * (1 << shift))) for c in base])
    _roundtrip(x, 16, kind=kind, order=order, porder=porder, method=method, lpc=lpc, block=4096)


@pytest.mark.parametrize("stereo", ["independent", "left_side", "right_side", "mid_side"])
@pytest.mark.parametrize("bps", [8, 16, 24])
def test_flac_stereo_decorrelations_and_sample_sizes(stereo, bps):
    rng = np.random.default_rng(bps)
    x = _speech_like(rng, 5000, ch=2, bps=bps)
    x[:, 1] = x[:, 0] // 2 + rng.integers(-3, 4, size=len(x))      # correlated channels, odd side values (mid/side parity bit)
    _roundtrip(x, bps, kind="fixed", order=2, porder=2, stereo=stereo, block=1152, rate=44100)


def test_flac_header_codes_constant_wasted_escape_and_metadata():
    rng = np.random.default_rng(3)
    # variable block size stream, explicit 8 / 16-bit block-size tails, explicit sample-rate tails, sample size from STREAMINFO
    x = _speech_like(rng, 7000)
    _roundtrip(x, 16, kind="fixed", order=2, porder=1, block=1000, variable=True, explicit_rate=True, explicit_size=False)
    _roundtrip(x[:300], 16, kind="fixed", order=1, block=200, rate=22050, explicit_rate=True)
    _roundtrip(x, 16, kind="lpc", order=2, lpc=(10, 7, [250, -124]), block=576, rate=16001, explicit_rate=True)   # 16-bit Hz tail
    # constant blocks (digital silence) and a DC block
    z = np.zeros(5000, np.int64)
    z[2048:4096] = -1234
    _roundtrip(z, 16, kind="constant", block=2048)
    # wasted bits: 16-bit container, 12 significant bits
    _roundtrip((x >> 4) << 4, 16, kind="fixed", order=2, block=4096)
    # an escaped (raw) partition, Rice2 parameters
    noisy = rng.integers(-30000, 30000, size=4096)
    _roundtrip(noisy, 16, kind="fixed", order=0, porder=2, method=1, escape_partition=1, block=4096)
    # ID3 tag in front, PADDING + VORBIS_COMMENT after STREAMINFO, no MD5, unknown total length
    from wav2letter_amd import data
    blob = FE.encode(x, id3=True, extra_metadata=True, md5=False, total_known=False)
    pcm, rate, bps = data.decode_flac(blob)
    assert np.array_equal(pcm[:, 0], x[:, 0]) and rate == 16000 and bps == 16
    # 8 channels, 20-bit samples
    x8 = np.stack([_speech_like(rng, 1200, bps=20)[:, 0] + c for c in range(8)], 1)
    _roundtrip(x8, 20, kind="fixed", order=2, block=512)


def test_flac_corruption_is_refused_and_files_load(tmp_path):
    from wav2letter_amd import _lib, data
    from wav2letter_amd.loader import read_audio_int16
    rng = np.random.default_rng(9)
    x = _speech_like(rng, 20000)
    blob = bytearray(FE.encode(x, kind="lpc", order=4, porder=3, lpc=(12, 9, [1400, -900, 200, -60]), block=4096))
    ok, _, _ = data.decode_flac(bytes(blob))
    assert np.array_equal(ok[:, 0], x[:, 0])
    # a flipped bit inside a frame body: the frame's CRC-16 catches it
    bad = bytearray(blob); bad[len(bad) // 2] ^= 0x10
    with pytest.raises(ValueError, match="CRC|synchronisation|flac"):
        data.decode_flac(bytes(bad))
    # a flipped bit in a frame header: CRC-8
    first = blob.find(b"\xff\xf8", 42)
    bad = bytearray(blob); bad[first + 3] ^= 0x02
    assert first >= 42
    with pytest.raises(ValueError, match="CRC-8"):
        data.decode_flac(bytes(bad))
    # a wrong signature in STREAMINFO
    bad = bytearray(blob); bad[4 + 4 + 18] ^= 0xFF
    with pytest.raises(ValueError, match="MD5"):
        data.decode_flac(bytes(bad))
    # truncated file, not a FLAC file
    with pytest.raises(ValueError):
        data.decode_flac(bytes(blob[:len(blob) * 2 // 3]))
    with pytest.raises(ValueError, match="fLaC"):
        data.decode_flac(b"RIFF" + bytes(100))
    # through the readers of the input pipeline
    p = tmp_path / "u.flac"
    p.write_bytes(bytes(blob))
    a, rate = data.read_audio(str(p))
    assert rate == 16000 and a.dtype == np.float32 and np.array_equal(a, (x[:, 0] / 32768.0).astype(np.float32))
    i16, rate = read_audio_int16(str(p))
    assert i16.dtype == np.int16 and np.array_equal(i16, x[:, 0].astype(np.int16))
    st = FE.encode(np.stack([x[:, 0], -x[:, 0]], 1), stereo="mid_side")
    (tmp_path / "s.flac").write_bytes(st)
    m, _ = data.read_audio(str(tmp_path / "s.flac"))
    assert np.abs(m).max() == 0.0                                     # the two channels cancel in the mono mix
    assert _lib.lib().w2l_flac_last_error() is not None
