// flac.cpp -- FLAC (native container) decoder for the input pipeline: the LibriSpeech lists the recipes
// prepare point at .flac files (data/librispeech/utils.py:36-46) and the reference decodes them through libsndfile
// (fl::pkg::speech::loadSound, un-vendored) on its loader threads.  libsndfile / libFLAC are not in this image, so this is a
// from-the-format-specification decoder (https://xiph.org/flac/format.html, RFC 9639): host C++, integer arithmetic, bit-exact
// by construction of the format -- every frame carries a CRC-16 of its bytes, every frame header a CRC-8, and STREAMINFO an
// MD5 of the decoded samples, all three are checked here, so a real file verifies itself end to end.
//
// Supported: the whole subset a lossless encoder emits -- CONSTANT / VERBATIM / FIXED (order 0-4) / LPC (order 1-32)
// subframes, Rice and Rice2 residuals with escaped partitions, wasted bits, all channel assignments (independent, left/side,
// right/side, mid/side), fixed and variable block sizes, 4-32 bits per sample, up to 8 channels.  Metadata blocks other than
// STREAMINFO are skipped; an ID3v2 tag in front of the stream is skipped.  Ogg-encapsulated FLAC is not.
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../../include/w2l_hip.h"

#define W2L_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_flacErr;

struct Md5 {   // RFC 1321
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  uint64_t bytes = 0;
  uint8_t buf[64];
  size_t fill = 0;
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
        0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
        0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
        0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
        0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
        0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 5, 9,  14, 20,
                              4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C; C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t* p, size_t n) {
    bytes += n;
    while (n) {
      const size_t take = 64 - fill < n ? 64 - fill : n;
      memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill == 64) { block(buf); fill = 0; }
    }
  }
  void finish(uint8_t out[16]) {
    const uint64_t bits = bytes * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    const uint32_t v[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(v[i / 4] >> (8 * (i % 4)));
  }
};

struct Bits {   // MSB-first reader over a byte range: 64-bit window, refilled bytewise
  const uint8_t* p;
  size_t n, byte = 0;
  uint64_t acc = 0;
  int have = 0;
  bool fail = false;
  Bits(const uint8_t* d, size_t bytes) : p(d), n(bytes) {}
  void fill() {
    while (have <= 56 && byte < n) { acc |= (uint64_t)p[byte++] << (56 - have); have += 8; }
  }
  size_t pos() const { return byte * 8 - (size_t)have; }   // bits consumed
  uint32_t get(int k) {   // k <= 32
    if (k == 0) return 0;
    fill();
    if (have < k) { fail = true; have = 0; acc = 0; return 0; }
    const uint32_t v = (uint32_t)(acc >> (64 - k));
    acc <<= k; have -= k;
    return v;
  }
  uint64_t get64(int k) { return k > 32 ? ((uint64_t)get(k - 32) << 32) | get(32) : get(k); }
  int64_t sget(int k) {   // two's complement, k <= 33
    if (k == 0) return 0;
    const uint64_t v = get64(k);
    const uint64_t sign = 1ull << (k - 1);
    return (int64_t)(v ^ sign) - (int64_t)sign;
  }
  uint32_t unary() {   // zeros before the next one
    uint32_t q = 0;
    for (;;) {
      fill();
      if (have == 0) { fail = true; return q; }
      if (acc == 0) { q += (uint32_t)have; have = 0; continue; }
      const int lz = __builtin_clzll(acc);
      if (lz >= have) { q += (uint32_t)have; acc = 0; have = 0; continue; }   // (cannot happen: bits past `have` are zero)
      q += (uint32_t)lz;
      acc <<= lz; acc <<= 1; have -= lz + 1;
      return q;
    }
  }
  void align() { const int drop = have & 7; acc <<= drop; have -= drop; }
};

uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}
uint16_t crc16(const uint8_t* p, size_t n) {
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= (uint16_t)p[i] << 8;
    for (int b = 0; b < 8; ++b) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
  }
  return c;
}

struct Info {
  int minBlock = 0, maxBlock = 0, rate = 0, channels = 0, bps = 0;
  uint64_t total = 0;
  uint8_t md5[16] = {0};
  size_t firstFrame = 0;   // byte offset of the first audio frame
};

bool parse_header(const uint8_t* d, size_t n, Info& info) {
  size_t off = 0;
  if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {   // ID3v2 tag: 10-byte header + syncsafe size
    const size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (d[9] & 0x7f);
    off = 10 + sz;
  }
  if (off + 4 > n || memcmp(d + off, "fLaC", 4) != 0) { g_flacErr = "flac: no 'fLaC' stream marker"; return false; }
  off += 4;
  bool haveInfo = false;
  for (;;) {
    if (off + 4 > n) { g_flacErr = "flac: truncated metadata"; return false; }
    const bool last = (d[off] & 0x80) != 0;
    const int type = d[off] & 0x7f;
    const size_t len = ((size_t)d[off + 1] << 16) | ((size_t)d[off + 2] << 8) | d[off + 3];
    off += 4;
    if (off + len > n) { g_flacErr = "flac: truncated metadata block"; return false; }
    if (type == 0) {
      if (len < 34) { g_flacErr = "flac: short STREAMINFO"; return false; }
      Bits b(d + off, len);
      info.minBlock = (int)b.get(16); info.maxBlock = (int)b.get(16);
      b.get(24); b.get(24);
      info.rate = (int)b.get(20); info.channels = (int)b.get(3) + 1; info.bps = (int)b.get(5) + 1;
      info.total = b.get64(36);
      memcpy(info.md5, d + off + 18, 16);
      haveInfo = true;
    }
    off += len;
    if (last) break;
  }
  if (!haveInfo || info.rate == 0 || info.bps < 4) { g_flacErr = "flac: missing or invalid STREAMINFO"; return false; }
  info.firstFrame = off;
  return true;
}

// one subframe of `bs` samples at `bps` bits into out[]; false on a malformed stream
bool subframe(Bits& b, int bs, int bps, int64_t* out) {
  if (b.get(1)) { g_flacErr = "flac: subframe padding bit set"; return false; }
  const int type = (int)b.get(6);
  int wasted = 0;
  if (b.get(1)) wasted = (int)b.unary() + 1;
  bps -= wasted;
  if (bps < 1) { g_flacErr = "flac: wasted bits exceed the sample size"; return false; }
  int order = 0;
  auto residual = [&](int predOrder) -> bool {
    const int method = (int)b.get(2);
    if (method > 1) { g_flacErr = "flac: reserved residual coding method"; return false; }
    const int pbits = method ? 5 : 4, esc = method ? 31 : 15;
    const int porder = (int)b.get(4);
    const int parts = 1 << porder;
    if ((bs >> porder) << porder != bs && porder > 0) { g_flacErr = "flac: block size not divisible by the partition count"; return false; }
    int i = predOrder;
    for (int pt = 0; pt < parts; ++pt) {
      int cnt = porder == 0 ? bs - predOrder : (pt == 0 ? (bs >> porder) - predOrder : (bs >> porder));
      if (cnt < 0) { g_flacErr = "flac: predictor order exceeds the partition"; return false; }
      const int k = (int)b.get(pbits);
      if (k == esc) {
        const int raw = (int)b.get(5);
        for (int j = 0; j < cnt; ++j) out[i++] = b.sget(raw);
      } else {
        for (int j = 0; j < cnt; ++j) {
          const uint64_t u = ((uint64_t)b.unary() << k) | (k ? b.get(k) : 0u);
          out[i++] = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
        }
      }
      if (b.fail) { g_flacErr = "flac: truncated residual"; return false; }
    }
    return true;
  };
  if (type == 0) {                       // CONSTANT
    const int64_t v = b.sget(bps);
    for (int i = 0; i < bs; ++i) out[i] = v;
  } else if (type == 1) {                // VERBATIM
    for (int i = 0; i < bs; ++i) out[i] = b.sget(bps);
  } else if (type >= 8 && type <= 12) {  // FIXED, order type - 8
    order = type - 8;
    if (order > bs) { g_flacErr = "flac: fixed predictor order exceeds the block"; return false; }
    for (int i = 0; i < order; ++i) out[i] = b.sget(bps);
    if (!residual(order)) return false;
    for (int i = order; i < bs; ++i) {
      int64_t p = 0;
      switch (order) {
        case 1: p = out[i - 1]; break;
        case 2: p = 2 * out[i - 1] - out[i - 2]; break;
        case 3: p = 3 * out[i - 1] - 3 * out[i - 2] + out[i - 3]; break;
        case 4: p = 4 * out[i - 1] - 6 * out[i - 2] + 4 * out[i - 3] - out[i - 4]; break;
        default: break;
      }
      out[i] += p;
    }
  } else if (type >= 32) {               // LPC, order (type & 31) + 1
    order = (type & 31) + 1;
    if (order > bs) { g_flacErr = "flac: LPC order exceeds the block"; return false; }
    for (int i = 0; i < order; ++i) out[i] = b.sget(bps);
    const int prec = (int)b.get(4) + 1;
    if (prec == 16) { g_flacErr = "flac: invalid LPC precision"; return false; }
    const int shift = (int)b.sget(5);
    if (shift < 0) { g_flacErr = "flac: negative LPC shift"; return false; }
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = b.sget(prec);
    if (!residual(order)) return false;
    for (int i = order; i < bs; ++i) {
      int64_t acc = 0;
      for (int j = 0; j < order; ++j) acc += coef[j] * out[i - 1 - j];
      out[i] += acc >> shift;
    }
  } else {
    g_flacErr = "flac: reserved subframe type";
    return false;
  }
  if (wasted)
    for (int i = 0; i < bs; ++i) out[i] *= (int64_t)1 << wasted;
  if (b.fail) { g_flacErr = "flac: truncated subframe"; return false; }
  return true;
}

}  // namespace

// stream parameters of a FLAC file held in memory.  Reference: the header libsndfile reads for fl::pkg::speech::loadSound.
W2L_API int w2l_flac_info(const uint8_t* data, size_t bytes, int* sampleRate, int* channels, int* bitsPerSample,
                          uint64_t* totalSamples) {
  Info info;
  if (!data || !parse_header(data, bytes, info)) return W2L_EINVAL;
  if (sampleRate) *sampleRate = info.rate;
  if (channels) *channels = info.channels;
  if (bitsPerSample) *bitsPerSample = info.bps;
  if (totalSamples) *totalSamples = info.total;
  return W2L_OK;
}

W2L_API const char* w2l_flac_last_error(void) { return g_flacErr.c_str(); }

// decode into out[sample][channel] (interleaved int32, `capacity` inter-channel samples).  *decoded = inter-channel samples
// written; *md5 = 1 the STREAMINFO signature matches the decoded audio, 0 it does not (W2L_EINVAL is returned as well), -1 the
// file carries no signature (all zero).  A frame that fails its CRC-8 / CRC-16 or does not parse is an error: W2L_EINVAL.
W2L_API int w2l_flac_decode(const uint8_t* data, size_t bytes, int32_t* out, uint64_t capacity, uint64_t* decoded, int* md5) {
  Info info;
  if (!data || !out || !parse_header(data, bytes, info)) return W2L_EINVAL;
  const int ch = info.channels;
  std::vector<int64_t> sub[8];
  Md5 sig;
  std::vector<uint8_t> raw;
  const int bytesPer = (info.bps + 7) / 8;
  uint64_t done = 0;
  size_t off = info.firstFrame;
  while (off + 2 <= bytes && (info.total == 0 || done < info.total)) {
    if (data[off] != 0xFF || (data[off + 1] & 0xFE) != 0xF8) { g_flacErr = "flac: lost frame synchronisation"; return W2L_EINVAL; }
    Bits b(data + off, bytes - off);
    b.get(14);
    if (b.get(1)) { g_flacErr = "flac: reserved header bit set"; return W2L_EINVAL; }
    b.get(1);   // blocking strategy: the coded number is a frame (0) or sample (1) index -- not needed for sequential decoding
    const int bsCode = (int)b.get(4), srCode = (int)b.get(4), chCode = (int)b.get(4), szCode = (int)b.get(3);
    if (b.get(1)) { g_flacErr = "flac: reserved header bit set"; return W2L_EINVAL; }
    {   // "UTF-8" coded frame / sample number: leading ones of the first byte give the byte count
      const uint32_t first = b.get(8);
      int extra = 0;
      if (first & 0x80) {
        int ones = 0;
        for (int m = 0x80; m && (first & m); m >>= 1) ++ones;
        if (ones < 2 || ones > 7) { g_flacErr = "flac: bad coded frame number"; return W2L_EINVAL; }
        extra = ones - 1;
      }
      for (int i = 0; i < extra; ++i)
        if ((b.get(8) & 0xC0) != 0x80) { g_flacErr = "flac: bad coded frame number"; return W2L_EINVAL; }
    }
    int bs;
    if (bsCode == 0) { g_flacErr = "flac: reserved block size code"; return W2L_EINVAL; }
    else if (bsCode == 1) bs = 192;
    else if (bsCode <= 5) bs = 576 << (bsCode - 2);
    else if (bsCode == 6) bs = (int)b.get(8) + 1;
    else if (bsCode == 7) bs = (int)b.get(16) + 1;
    else bs = 256 << (bsCode - 8);
    if (srCode == 12) b.get(8);
    else if (srCode == 13 || srCode == 14) b.get(16);
    else if (srCode == 15) { g_flacErr = "flac: invalid sample rate code"; return W2L_EINVAL; }
    if (b.fail) { g_flacErr = "flac: truncated frame header"; return W2L_EINVAL; }
    const size_t hdrBytes = b.pos() >> 3;
    const uint32_t c8 = b.get(8);
    if (b.fail || crc8(data + off, hdrBytes) != c8) { g_flacErr = "flac: frame header CRC-8 mismatch"; return W2L_EINVAL; }
    static const int kBps[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    int bps = kBps[szCode];
    if (bps < 0) { g_flacErr = "flac: reserved sample size code"; return W2L_EINVAL; }
    if (bps == 0) bps = info.bps;
    int nch;
    if (chCode < 8) nch = chCode + 1;
    else if (chCode <= 10) nch = 2;
    else { g_flacErr = "flac: reserved channel assignment"; return W2L_EINVAL; }
    if (nch != ch || bps != info.bps) { g_flacErr = "flac: frame parameters differ from STREAMINFO"; return W2L_EINVAL; }
    for (int c = 0; c < nch; ++c) {
      sub[c].resize((size_t)bs);
      // the side channel of a stereo pair carries one more bit
      const bool side = (chCode == 8 && c == 1) || (chCode == 9 && c == 0) || (chCode == 10 && c == 1);
      if (!subframe(b, bs, bps + (side ? 1 : 0), sub[c].data())) return W2L_EINVAL;
    }
    b.align();
    const size_t bodyBytes = b.pos() >> 3;
    const uint32_t c16 = b.get(16);
    if (b.fail || crc16(data + off, bodyBytes) != c16) { g_flacErr = "flac: frame CRC-16 mismatch"; return W2L_EINVAL; }
    if (chCode == 8) for (int i = 0; i < bs; ++i) sub[1][i] = sub[0][i] - sub[1][i];            // left, side -> right = left - side
    else if (chCode == 9) for (int i = 0; i < bs; ++i) sub[0][i] = sub[0][i] + sub[1][i];       // side, right -> left = right + side
    else if (chCode == 10)
      for (int i = 0; i < bs; ++i) {
        const int64_t s = sub[1][i];
        const int64_t m = (sub[0][i] * 2) | (s & 1);
        sub[0][i] = (m + s) >> 1;
        sub[1][i] = (m - s) >> 1;
      }
    uint64_t take = (uint64_t)bs;
    if (info.total && done + take > info.total) take = info.total - done;
    if (done + take > capacity) { g_flacErr = "flac: output buffer too small"; return W2L_EINVAL; }
    raw.resize((size_t)take * nch * bytesPer);
    size_t r = 0;
    for (uint64_t i = 0; i < take; ++i)
      for (int c = 0; c < nch; ++c) {
        const int64_t v = sub[c][i];
        out[(done + i) * nch + c] = (int32_t)v;
        for (int k = 0; k < bytesPer; ++k) raw[r++] = (uint8_t)((uint64_t)v >> (8 * k));
      }
    sig.update(raw.data(), raw.size());
    done += take;
    off += b.pos() >> 3;
  }
  if (info.total && done != info.total) { g_flacErr = "flac: stream ends before the announced sample count"; return W2L_EINVAL; }
  if (decoded) *decoded = done;
  bool any = false;
  for (int i = 0; i < 16; ++i) any = any || info.md5[i];
  int ok = -1;
  if (any) {
    uint8_t dig[16];
    sig.finish(dig);
    ok = memcmp(dig, info.md5, 16) == 0 ? 1 : 0;
  }
  if (md5) *md5 = ok;
  if (ok == 0) { g_flacErr = "flac: MD5 signature of the decoded audio does not match STREAMINFO"; return W2L_EINVAL; }
  return W2L_OK;
}
