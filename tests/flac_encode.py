"""A small FLAC ENCODER for the decoder's tests (test infrastructure, written from the format specification RFC 9639 like the
decoder it exercises -- no libFLAC in this image).  It can be told which subframe type, predictor order, Rice partitioning,
stereo decorrelation, block-size / sample-rate / sample-size header codes to use, so that every branch of
wav2letter_amd/csrc/host/flac.cpp is reached; the STREAMINFO MD5 is hashlib's, the frame CRCs are computed here.

encode(samples [n][channels] int array, bps, rate, block=4096, kind="fixed", ...) -> bytes
"""
import hashlib

import numpy as np


class BitWriter:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, v, k):
        if k == 0:
            return
        v = int(v) & ((1 << k) - 1)
        self.acc = (self.acc << k) | v
        self.n += k
        while self.n >= 8:
            self.n -= 8
            self.out.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def unary(self, q):
        while q >= 32:
            self.put(0, 32)
            q -= 32
        self.put(1, q + 1)

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)

    def bytes(self):
        assert self.n == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def utf8_number(v):
    if v < 0x80:
        return bytes([v])
    for nbytes, lead in ((2, 0xC0), (3, 0xE0), (4, 0xF0), (5, 0xF8), (6, 0xFC), (7, 0xFE)):
        bits = (7 - nbytes) + 6 * (nbytes - 1) if nbytes < 7 else 36
        if v < (1 << bits):
            out = []
            for _ in range(nbytes - 1):
                out.append(0x80 | (v & 0x3F))
                v >>= 6
            out.append(lead | v)
            return bytes(reversed(out))
    raise ValueError("number too large")


def _residual(w, res, order, bs, porder, method, escape_partition=-1):
    w.put(method, 2)
    w.put(porder, 4)
    pbits, esc = (5, 31) if method else (4, 15)
    parts = 1 << porder
    i = 0
    for pt in range(parts):
        cnt = bs - order if porder == 0 else ((bs >> porder) - order if pt == 0 else bs >> porder)
        part = res[i:i + cnt]
        i += cnt
        if pt == escape_partition:
            raw = max(1, int(max(abs(int(v)) for v in part)).bit_length() + 1) if cnt else 1
            w.put(esc, pbits)
            w.put(raw, 5)
            for v in part:
                w.put(v, raw)
            continue
        mean = float(np.mean(np.abs(np.asarray(part, np.float64)))) if cnt else 0.0
        k = min(esc - 1, max(0, int(np.ceil(np.log2(mean + 1.0)))))
        w.put(k, pbits)
        for v in part:
            v = int(v)
            u = (v << 1) if v >= 0 else ((-v << 1) - 1)
            w.unary(u >> k)
            w.put(u & ((1 << k) - 1), k)


def _subframe(w, x, bps, kind, order, porder, method, lpc, escape_partition):
    x = [int(v) for v in x]
    bs = len(x)
    wasted = 0
    if kind != "constant" and any(x):
        while all((v >> wasted) & 1 == 0 for v in x):
            wasted += 1
    if wasted:
        x = [v >> wasted for v in x]
        bps -= wasted
    if kind == "constant":
        assert all(v == x[0] for v in x)
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind == "fixed":
        code = 8 + order
    else:
        code = 32 + order - 1
    w.put(0, 1)
    w.put(code, 6)
    if wasted:
        w.put(1, 1)
        w.unary(wasted - 1)
    else:
        w.put(0, 1)
    if kind == "constant":
        w.put(x[0], bps)
        return
    if kind == "verbatim":
        for v in x:
            w.put(v, bps)
        return
    for v in x[:order]:
        w.put(v, bps)
    res = []
    if kind == "fixed":
        coef = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}[order]
        for i in range(order, bs):
            res.append(x[i] - sum(c * x[i - 1 - j] for j, c in enumerate(coef)))
    else:
        prec, shift, coef = lpc
        w.put(prec - 1, 4)
        w.put(shift, 5)
        for c in coef:
            w.put(c, prec)
        for i in range(order, bs):
            res.append(x[i] - (sum(c * x[i - 1 - j] for j, c in enumerate(coef)) >> shift))
    _residual(w, res, order, bs, porder, method, escape_partition)


_BS_CODES = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
_SR_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
_SZ_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def encode(samples, bps=16, rate=16000, block=4096, kind="fixed", order=2, porder=0, method=0, stereo="independent", lpc=None,
           escape_partition=-1, variable=False, explicit_rate=False, explicit_size=True, md5=True, id3=False, extra_metadata=False,
           total_known=True):
    """samples: [n][channels] integers.  kind: constant | verbatim | fixed | lpc (per subframe; a block that is not constant
    falls back to verbatim when `kind` is constant).  stereo: independent | left_side | right_side | mid_side (2 channels)."""
    x = np.asarray(samples, np.int64)
    if x.ndim == 1:
        x = x[:, None]
    n, ch = x.shape
    body = bytearray()
    pos, frame = 0, 0
    sizes = []
    while pos < n:
        bs = min(block, n - pos)
        blk = x[pos:pos + bs]
        w = BitWriter()
        w.put(0x3FFE, 14)
        w.put(0, 1)
        w.put(1 if variable else 0, 1)
        if bs in _BS_CODES and not (frame % 3 == 2 and bs <= 65536 and bs > 256):
            bcode, btail = _BS_CODES[bs], None
        elif bs <= 256:
            bcode, btail = 6, (bs - 1, 8)
        else:
            bcode, btail = 7, (bs - 1, 16)
        if explicit_rate and rate % 1000 == 0 and rate // 1000 < 256:
            scode, stail = 12, (rate // 1000, 8)
        elif explicit_rate and rate < 65536:
            scode, stail = 13, (rate, 16)
        elif explicit_rate and rate % 10 == 0 and rate // 10 < 65536:
            scode, stail = 14, (rate // 10, 16)
        else:
            scode, stail = (_SR_CODES.get(rate, 0) if frame % 2 else 0), None
        w.put(bcode, 4)
        w.put(scode, 4)
        chcode = {"independent": ch - 1, "left_side": 8, "right_side": 9, "mid_side": 10}[stereo]
        w.put(chcode, 4)
        w.put(_SZ_CODES.get(bps, 0) if explicit_size else 0, 3)
        w.put(0, 1)
        for byte in utf8_number(pos if variable else frame):
            w.put(byte, 8)
        if btail:
            w.put(*btail)
        if stail:
            w.put(*stail)
        w.put(crc8(bytes(w.out)), 8)
        chans = [blk[:, c] for c in range(ch)]
        widths = [bps] * ch
        if stereo == "left_side":
            chans = [blk[:, 0], blk[:, 0] - blk[:, 1]]
            widths = [bps, bps + 1]
        elif stereo == "right_side":
            chans = [blk[:, 0] - blk[:, 1], blk[:, 1]]
            widths = [bps + 1, bps]
        elif stereo == "mid_side":
            chans = [(blk[:, 0] + blk[:, 1]) >> 1, blk[:, 0] - blk[:, 1]]
            widths = [bps, bps + 1]
        for c, (cx, cw) in enumerate(zip(chans, widths)):
            k, o = kind, order
            if k == "constant" and not np.all(cx == cx[0]):
                k = "verbatim"
            if k in ("fixed", "lpc") and o > bs:
                k = "verbatim"
            po = porder
            while po > 0 and ((bs >> po) << po != bs or (bs >> po) < o):
                po -= 1
            _subframe(w, cx, cw, k, o, po, method, lpc, escape_partition if po == porder else -1)
        w.align()
        raw = bytes(w.out)
        fr = raw + crc16(raw).to_bytes(2, "big")
        sizes.append(len(fr))
        body += fr
        pos += bs
        frame += 1
    bytes_per = (bps + 7) // 8
    pcm = bytearray()
    for row in x:
        for v in row:
            pcm += int(v).to_bytes(bytes_per, "little", signed=True)
    si = BitWriter()
    si.put(block if n >= block else max(16, n), 16)
    si.put(block, 16)
    si.put(min(sizes), 24)
    si.put(max(sizes), 24)
    si.put(rate, 20)
    si.put(ch - 1, 3)
    si.put(bps - 1, 5)
    si.put(n if total_known else 0, 36)
    digest = hashlib.md5(bytes(pcm)).digest() if md5 else bytes(16)
    head = bytearray()
    if id3:
        head += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 12]) + bytes(12)
    head += b"fLaC"
    head += bytes([0x00 if extra_metadata else 0x80, 0, 0, 34]) + si.bytes() + digest
    if extra_metadata:   # a PADDING block and a (vendor-only) VORBIS_COMMENT block, the second marked last
        head += bytes([0x01, 0, 0, 8]) + bytes(8)
        vc = (4).to_bytes(4, "little") + b"test" + (0).to_bytes(4, "little")
        head += bytes([0x84, 0, 0, len(vc)]) + vc
    return bytes(head) + bytes(body)
