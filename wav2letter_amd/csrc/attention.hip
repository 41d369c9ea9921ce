// attention.hip -- the attention core of fl::Transformer (arch token `TR`, recipes/sota/2019/am_arch/am_transformer_ctc.arch
// :15-38; block structure recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151 selfAttention, :153-182 forward;
// fl::multiheadAttention / relativePositionEmbeddingRotate themselves are [UNVENDORED] Flashlight) and the time-axis max
// pool between the convolutional front end's stages (`M 1 1 2 1`, am_transformer_ctc.arch:5).
//
//   scores[b][h][i][j] = (q_i . k_j + q_i . E[j - i + n0]) / sqrt(d)   n0 = csz - 1, E the (2 csz - 1, d) embedding table
//   P = softmax_j(scores);  ctx_i = sum_j P[i][j] v_j
//
// Activations are frame-major [B][T][heads*d]: head h of frame (b, t) is the d contiguous floats at ((b*T + t)*heads + h)*d,
// so the per-(utterance, head) products are strided batched GEMMs straight out of the q / k / v buffers (no transposes),
// and the relative term is ONE plain GEMM R = Qflat[B*T*heads][d] . Ewin^T over the 2T-1 table rows a T-frame utterance
// can reach; the softmax kernel gathers R's skewed diagonal (the reference materialises the skew by a pad + reshape).
// The attention products are < 1 % of a TR block's flops: the batched GEMM is a plain double-buffered
// 64x64x16 MFMA tile kernel, not a tuned one.
#include "common.hpp"

namespace w2l {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBgK = 16, kBgPitch = 68;

struct BgOperand {
  const float* p;
  long long sR, sK;  // element strides of the 64-wide tile dimension (m of A / n of B) and of k
  int vec;           // float4 loads along the contiguous dimension are legal (alignment, extents % 4 == 0)
};

struct BgRegs { float v[4]; };

// operand access modes, fixed per launch (template parameters: the tile loop has no mode branches)
constexpr int kBgRowVec = 0;   // the 64-wide tile dimension is contiguous: one float4 of 4 rows per thread (k = tid / 16)
constexpr int kBgKVec = 1;     // k is contiguous: one float4 of 4 k per thread (row = tid / 4)
constexpr int kBgGeneric = 2;  // anything else, or extents / alignment that forbid float4: four bounds-checked scalar loads

// 64 x 16 tile of an operand -> registers
template <int MODE>
__device__ __forceinline__ BgRegs bg_load(const BgOperand& o, int tid, int rv, int k0, int K) {
  BgRegs r;
  if (MODE == kBgRowVec) {
    const int k = k0 + (tid >> 4), rq = (tid & 15) * 4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (k < K && rq < rv) t = *(const f32x4*)(o.p + (long long)k * o.sK + rq);   // extents are multiples of 4
    r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
  } else if (MODE == kBgKVec) {
    const int row = tid >> 2, k = k0 + (tid & 3) * 4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (row < rv && k < K) t = *(const f32x4*)(o.p + (long long)row * o.sR + k);
    r.v[0] = t[0]; r.v[1] = t[1]; r.v[2] = t[2]; r.v[3] = t[3];
  } else if (o.sR == 1) {
    const int k = k0 + (tid >> 4), rq = (tid & 15) * 4;
    const float* src = o.p + (long long)k * o.sK + rq;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = (k < K && rq + i < rv) ? src[i] : 0.f;
  } else {
    const int row = tid >> 2, k = k0 + (tid & 3) * 4;
    const float* src = o.p + (long long)row * o.sR + (long long)k * o.sK;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = (row < rv && k + i < K) ? src[(long long)i * o.sK] : 0.f;
  }
  return r;
}
template <int MODE>
__device__ __forceinline__ void bg_store(const BgOperand& o, int tid, const BgRegs& r, float (*s)[kBgPitch]) {
  if (MODE == kBgRowVec || (MODE == kBgGeneric && o.sR == 1)) {
    f32x4 t = {r.v[0], r.v[1], r.v[2], r.v[3]};
    *(f32x4*)&s[tid >> 4][(tid & 15) * 4] = t;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) s[(tid & 3) * 4 + i][tid >> 2] = r.v[i];
  }
}

struct BgP {
  BgOperand A, B;
  float* C;
  int M, N, K, G2, tilesN;
  long long a1, a2, b1, b2, ldc, c1, c2;
  int accumulate;
  int bandMode = 0, bandT = 0, bandH = 0, bandOff = 0;   // w2l_bgemm_desc::bandMode (bf16 kernel only)
};

// K range [kLo, kHi) that can hold non-zeros for the 64-row tile at row r0 of a BANDED A operand (the skewed score gradient dR of
// the relative-position products: row (b, i, h) is non-zero only at the T table rows w = j - i + bandOff, j in [0, T)):
//   mode 1  rows = (b, i, h) flattened, k = w          (dq += dR E)
//   mode 2  rows = w, k = (i, h) flattened             (dE = dR^T q, per utterance)
// Skipped K tiles multiply exact zeros: the result is bit-identical to the full product.
__device__ __forceinline__ void bg_band(const BgP& p, int r0, int& kLo, int& kHi) {
  kLo = 0; kHi = p.K;
  if (p.bandMode == 1) {
    const int r1 = min(r0 + 63, p.M - 1);
    const int q0 = r0 / p.bandH, q1 = r1 / p.bandH;          // (b, i) flattened
    if (q0 / p.bandT == q1 / p.bandT) {                       // one utterance
      const int iLo = q0 % p.bandT, iHi = q1 % p.bandT;
      kLo = max(0, p.bandOff - iHi);
      kHi = min(p.K, p.bandOff - iLo + p.bandT);
    }
  } else if (p.bandMode == 2) {
    const int w1 = min(r0 + 63, p.M - 1);
    const int iLo = max(0, p.bandOff - w1), iHi = min(p.bandT - 1, p.bandOff - r0 + p.bandT - 1);
    kLo = iLo * p.bandH;
    kHi = min(p.K, (iHi + 1) * p.bandH);
  }
  if (kHi < kLo) kHi = kLo;
}

template <int AM, int BM>
__global__ __launch_bounds__(256) void bgemm_k(BgP p) {
  __shared__ __attribute__((aligned(16))) float As[2][kBgK][kBgPitch];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBgK][kBgPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / p.tilesN, tn = blockIdx.x - tm * p.tilesN;
  const int g1 = blockIdx.y / p.G2, g2 = blockIdx.y - g1 * p.G2;
  BgOperand a = p.A, b = p.B;
  a.p += g1 * p.a1 + g2 * p.a2 + (long long)tm * 64 * a.sR;
  b.p += g1 * p.b1 + g2 * p.b2 + (long long)tn * 64 * b.sR;
  const int mv = min(64, p.M - tm * 64), nv = min(64, p.N - tn * 64);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nk = (p.K + kBgK - 1) / kBgK;
  BgRegs ra = bg_load<AM>(a, tid, mv, 0, p.K), rb = bg_load<BM>(b, tid, nv, 0, p.K);
  bg_store<AM>(a, tid, ra, As[0]);
  bg_store<BM>(b, tid, rb, Bs[0]);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      ra = bg_load<AM>(a, tid, mv, (kt + 1) * kBgK, p.K);
      rb = bg_load<BM>(b, tid, nv, (kt + 1) * kBgK, p.K);
    }
#pragma unroll
    for (int kk = 0; kk < kBgK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[cur][2 * kk + lh][wm + li], Bs[cur][2 * kk + lh][wn + li], acc, 0, 0, 0);
    if (more) {
      bg_store<AM>(a, tid, ra, As[cur ^ 1]);
      bg_store<BM>(b, tid, rb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  float* C = p.C + g1 * p.c1 + g2 * p.c2;
  const int n = tn * 64 + wn + li;
  if (n < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = tm * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) {
        float* dst = C + (long long)m * p.ldc + n;
        *dst = p.accumulate ? *dst + acc[r] : acc[r];
      }
    }
  }
}


// ---- the same batched product with bf16 MULTIPLIES (mixed precision, BASELINE config 5 "bf16 MFMA attention"): fp32 operands
// in memory (q / k / v, probabilities, gradients: whatever the block holds), rounded to nearest-even bf16 on the way into LDS,
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 result.  64 x 64 x 32 tiles; the LDS image of an operand tile is
// [row][k] with k contiguous (80-byte rows = 16 bytes x 5: conflict-free ds_read_b128 fragments of 8 k), whatever the
// operand's orientation in memory: a k-contiguous source stores eight k with one ds_write_b128, a row-contiguous source
// (k strided) scatters eight rows of one k.  Eight elements per thread, operand and K tile.
typedef __bf16 bg_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bg_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float bg_f32x2_t __attribute__((ext_vector_type(2)));
constexpr int kBhK = 32, kBhPitch = 40;   // bf16 elements

struct BhRegs { float v[8]; };

__device__ __forceinline__ uint32_t bh_pack2(float a, float b) {
  const bg_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bg_bf16x2_t));
}

template <int MODE>
__device__ __forceinline__ BhRegs bh_load(const BgOperand& o, int tid, int rv, int k0, int K) {
  BhRegs r;
  if (MODE == kBgRowVec) {           // rows contiguous: k = tid / 8, rows (tid % 8) * 8 .. + 8
    const int k = k0 + (tid >> 3), rq = (tid & 7) * 8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (k < K) {
      const float* src = o.p + (long long)k * o.sK + rq;
      if (rq < rv) a = *(const f32x4*)src;            // extents are multiples of 4
      if (rq + 4 < rv) b = *(const f32x4*)(src + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  } else if (MODE == kBgKVec) {      // k contiguous: row = tid / 4, k (tid % 4) * 8 .. + 8
    const int row = tid >> 2, k = k0 + (tid & 3) * 8;
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (row < rv) {
      const float* src = o.p + (long long)row * o.sR + k;
      if (k < K) a = *(const f32x4*)src;
      if (k + 4 < K) b = *(const f32x4*)(src + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.v[i] = a[i]; r.v[4 + i] = b[i]; }
  } else if (o.sR == 1) {
    const int k = k0 + (tid >> 3), rq = (tid & 7) * 8;
    const float* src = o.p + (long long)k * o.sK + rq;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (k < K && rq + i < rv) ? src[i] : 0.f;
  } else {
    const int row = tid >> 2, k = k0 + (tid & 3) * 8;
    const float* src = o.p + (long long)row * o.sR + (long long)k * o.sK;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = (row < rv && k + i < K) ? src[(long long)i * o.sK] : 0.f;
  }
  return r;
}
template <int MODE>
__device__ __forceinline__ void bh_store(const BgOperand& o, int tid, const BhRegs& r, uint16_t (*s)[kBhPitch]) {
  if (MODE == kBgRowVec || (MODE == kBgGeneric && o.sR == 1)) {
    const int kl = tid >> 3, rq = (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[rq + i][kl] = (uint16_t)(bh_pack2(r.v[i], 0.f) & 0xffffu);
  } else {
    const int row = tid >> 2, kq = (tid & 3) * 8;
    const uint4 q = make_uint4(bh_pack2(r.v[0], r.v[1]), bh_pack2(r.v[2], r.v[3]), bh_pack2(r.v[4], r.v[5]), bh_pack2(r.v[6], r.v[7]));
    *(uint4*)&s[row][kq] = q;
  }
}

template <int AM, int BM>
__global__ __launch_bounds__(256) void bgemm_bf_k(BgP p) {
  __shared__ __attribute__((aligned(16))) uint16_t As[2][64][kBhPitch];
  __shared__ __attribute__((aligned(16))) uint16_t Bs[2][64][kBhPitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / p.tilesN, tn = blockIdx.x - tm * p.tilesN;
  const int g1 = blockIdx.y / p.G2, g2 = blockIdx.y - g1 * p.G2;
  BgOperand a = p.A, b = p.B;
  a.p += g1 * p.a1 + g2 * p.a2 + (long long)tm * 64 * a.sR;
  b.p += g1 * p.b1 + g2 * p.b2 + (long long)tn * 64 * b.sR;
  const int mv = min(64, p.M - tm * 64), nv = min(64, p.N - tn * 64);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  int kLo, kHi;
  bg_band(p, tm * 64, kLo, kHi);
  const int kt0 = kLo / kBhK, nk = (kHi + kBhK - 1) / kBhK;   // K tiles kt0 .. nk (an empty band leaves C = 0 / untouched below)
  BhRegs ra = bh_load<AM>(a, tid, mv, kt0 * kBhK, p.K), rb = bh_load<BM>(b, tid, nv, kt0 * kBhK, p.K);
  bh_store<AM>(a, tid, ra, As[kt0 & 1]);
  bh_store<BM>(b, tid, rb, Bs[kt0 & 1]);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      ra = bh_load<AM>(a, tid, mv, (kt + 1) * kBhK, p.K);
      rb = bh_load<BM>(b, tid, nv, (kt + 1) * kBhK, p.K);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bg_bf16x8_t fa = *(const bg_bf16x8_t*)&As[cur][wm + li][16 * ks + 8 * lh];
      const bg_bf16x8_t fb = *(const bg_bf16x8_t*)&Bs[cur][wn + li][16 * ks + 8 * lh];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
    }
    if (more) {
      bh_store<AM>(a, tid, ra, As[cur ^ 1]);
      bh_store<BM>(b, tid, rb, Bs[cur ^ 1]);
    }
    __syncthreads();
  }
  float* C = p.C + g1 * p.c1 + g2 * p.c2;
  const int n = tn * 64 + wn + li;
  if (n < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = tm * 64 + wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) {
        float* dst = C + (long long)m * p.ldc + n;
        *dst = p.accumulate ? *dst + acc[r] : acc[r];
      }
    }
  }
}

// ---- softmax over the keys of one query row, relative-position term gathered from R ------------------------------------
// one wave per row (b, h, i); S row in place -> P.  R row of (b, i, h): entry w holds q_i . E[rlo + w]
struct SmP {
  float* S;
  const float* P;
  float* dR;
  const float* R;
  const int* keyLen;   // [B] keys j >= keyLen[b] are padding (log(0) added to their scores); NULL: none
  int B, H, T, ldr, rlo, W, n0;
  float scale;
};

__global__ __launch_bounds__(256) void attn_softmax_fwd_k(SmP p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.B * p.H * p.T) return;
  const int i = row % p.T, bh = row / p.T, h = bh % p.H, b = bh / p.H;
  float* s = p.S + (size_t)row * p.T;
  const float* r = p.R ? p.R + ((size_t)(b * p.T + i) * p.H + h) * p.ldr : nullptr;
  const int kl = p.keyLen ? min(p.keyLen[b], p.T) : p.T;
  float mx = -INFINITY;
  for (int j = lane; j < p.T; j += 64) {
    float v = s[j];
    if (r) {
      const int w = j - i + p.n0 - p.rlo;
      if (w >= 0 && w < p.W) v += r[w];
    }
    v = j < kl ? v * p.scale : -INFINITY;
    s[j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < p.T; j += 64) {
    float e = j < kl ? expf(s[j] - mx) : 0.f;
    s[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;   // an utterance with no valid key: zeros (the reference's softmax gives NaN)
  for (int j = lane; j < p.T; j += 64) s[j] *= inv;
}

// dS_j = scale * P_j (dP_j - sum_k P_k dP_k) in place over dP; dR row = the skewed copy of dS (zero where no key maps)
__global__ __launch_bounds__(256) void attn_softmax_bwd_k(SmP p) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= p.B * p.H * p.T) return;
  const int i = row % p.T, bh = row / p.T, h = bh % p.H, b = bh / p.H;
  float* d = p.S + (size_t)row * p.T;
  const float* pr = p.P + (size_t)row * p.T;
  float dot = 0.f;
  for (int j = lane; j < p.T; j += 64) dot += pr[j] * d[j];
  dot = wave_sum(dot);
  if (p.dR) {
    // the skewed copy first, from the incoming dP (d is overwritten in place below)
    float* dr = p.dR + ((size_t)(b * p.T + i) * p.H + h) * p.ldr;
    for (int w = lane; w < p.ldr; w += 64) {
      const int j = w + p.rlo - p.n0 + i;
      dr[w] = (w < p.W && j >= 0 && j < p.T) ? p.scale * pr[j] * (d[j] - dot) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // those loads have landed before d changes
  }
  for (int j = lane; j < p.T; j += 64) d[j] = p.scale * pr[j] * (d[j] - dot);
}

// ---- padding mask of the keys (forwardSequentialModuleWithPadMask, recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:58-81,
// and TransformerCPC.cpp:138-144): valid input frames  n_b = ceil(inputSizes[b] * Tin / max(inputSizes)),  mask[t][b] = t < n_b on
// the Tin input frames, resized to the block's Tk frames (af::resize, nearest: source index round(j * Tin / Tk), clamped) and added
// to the scores as log(mask).  The mask is monotone, so it is kept as the count of valid keys per utterance.
// `full` (device scalar, may be null): the size the Tin input frames correspond to when the batch is padded BEYOND its longest
// utterance (a caller that rounds T up for the sake of few distinct plans); the reference pads to the longest only, where the
// denominator is max(sizes).
__global__ void attn_key_len_k(const float* __restrict__ sizes, const float* __restrict__ full, int B, int Tin, int Tk,
                               int* __restrict__ keyLen) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = sizes[0];
  for (int q = 1; q < B; ++q) mx = fmaxf(mx, sizes[q]);
  if (full) mx = fmaxf(mx, *full);
  const float nb = ceilf(sizes[b] * (float)Tin / mx);
  const float xf = (float)Tin / (float)Tk;
  int n = 0;
  for (int j = 0; j < Tk; ++j) {
    int src = (int)roundf((float)j * xf);
    if (src >= Tin) src = Tin - 1;
    if ((float)src < nb) n = j + 1; else break;
  }
  keyLen[b] = n;
}

// ---- fl::Pool2D(wx, 1, sx, 1, MAX) over time on frame-major rows ------------------------------------------------------
__global__ __launch_bounds__(256) void pool_time_fwd_k(const float* __restrict__ x, float* __restrict__ y, int T, int To,
                                                       int F, int w, int stride, size_t n) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const int f = idx % F;
    const size_t bt = idx / F;
    const int to = bt % To;
    const size_t b = bt / To;
    const float* src = x + ((size_t)b * T + (size_t)to * stride) * F + f;
    float m = src[0];
    for (int k = 1; k < w; ++k) m = fmaxf(m, src[(size_t)k * F]);
    y[idx] = m;
  }
}
// gradient to the FIRST maximum of each window; windows may overlap (w > stride): one thread per input element gathers
__global__ __launch_bounds__(256) void pool_time_bwd_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ dx, int T, int To, int F, int w, int stride, size_t n) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (size_t)gridDim.x * 256) {
    const int f = idx % F;
    const size_t bt = idx / F;
    const int t = bt % T;
    const size_t b = bt / T;
    float g = 0.f;
    // windows to with to*stride <= t < to*stride + w
    int lo = t - w + 1;
    lo = lo <= 0 ? 0 : (lo + stride - 1) / stride;
    for (int to = lo; to < To && to * stride <= t; ++to) {
      const float* src = x + ((size_t)b * T + (size_t)to * stride) * F + f;
      int arg = 0;
      float m = src[0];
      for (int k = 1; k < w; ++k) {
        float v = src[(size_t)k * F];
        if (v > m) { m = v; arg = k; }
      }
      if (to * stride + arg == t) g += dy[((size_t)b * To + to) * F + f];
    }
    dx[idx] = g;
  }
}

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace w2l

using namespace w2l;
#define W2L_S ((hipStream_t)stream)

static int bgemm_launch(const w2l_bgemm_desc* d, const float* A, const float* B, float* C, bool bf16, w2l_stream_t stream) {
  if (!d || !A || !B || !C) return W2L_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->G1 <= 0 || d->G2 <= 0) return W2L_EINVAL;
  if ((long long)d->G1 * d->G2 > 65535) return W2L_EINVAL;
  BgP p;
  auto mult4 = [](long long v) { return (v & 3) == 0; };
  p.A = {A, d->sam, d->sak, 0};
  p.B = {B, d->sbn, d->sbk, 0};
  // float4 loads: the contiguous dimension's extent and every other stride a multiple of 4 floats, base 16-byte aligned
  if (d->sam == 1) p.A.vec = al16(A) && mult4(d->sak) && mult4(d->a1) && mult4(d->a2) && mult4(d->M);
  else if (d->sak == 1) p.A.vec = al16(A) && mult4(d->sam) && mult4(d->a1) && mult4(d->a2) && mult4(d->K);
  if (d->sbn == 1) p.B.vec = al16(B) && mult4(d->sbk) && mult4(d->b1) && mult4(d->b2) && mult4(d->N);
  else if (d->sbk == 1) p.B.vec = al16(B) && mult4(d->sbn) && mult4(d->b1) && mult4(d->b2) && mult4(d->K);
  p.C = C;
  p.M = d->M; p.N = d->N; p.K = d->K; p.G2 = d->G2;
  p.tilesN = (d->N + 63) / 64;
  p.a1 = d->a1; p.a2 = d->a2; p.b1 = d->b1; p.b2 = d->b2; p.ldc = d->ldc; p.c1 = d->c1; p.c2 = d->c2;
  p.accumulate = d->accumulate;
  p.bandMode = d->bandMode; p.bandT = d->bandT; p.bandH = d->bandH; p.bandOff = d->bandOff;
  if (p.bandMode < 0 || p.bandMode > 2 || (p.bandMode && (p.bandT <= 0 || p.bandH <= 0))) return W2L_EINVAL;
  const long long tiles = (long long)((d->M + 63) / 64) * p.tilesN;
  if (tiles > 0x7fffffffLL) return W2L_EINVAL;
  const int am = !p.A.vec ? kBgGeneric : (d->sam == 1 ? kBgRowVec : kBgKVec);
  const int bm = !p.B.vec ? kBgGeneric : (d->sbn == 1 ? kBgRowVec : kBgKVec);
  const dim3 grid((unsigned)tiles, (unsigned)(d->G1 * d->G2));
#define W2L_BG(AMv, BMv)                                                                      \
  do {                                                                                        \
    if (bf16) hipLaunchKernelGGL((bgemm_bf_k<AMv, BMv>), grid, dim3(256), 0, W2L_S, p);       \
    else hipLaunchKernelGGL((bgemm_k<AMv, BMv>), grid, dim3(256), 0, W2L_S, p);               \
  } while (0)
  switch (am * 3 + bm) {
    case 0: W2L_BG(0, 0); break;
    case 1: W2L_BG(0, 1); break;
    case 2: W2L_BG(0, 2); break;
    case 3: W2L_BG(1, 0); break;
    case 4: W2L_BG(1, 1); break;
    case 5: W2L_BG(1, 2); break;
    case 6: W2L_BG(2, 0); break;
    case 7: W2L_BG(2, 1); break;
    default: W2L_BG(2, 2); break;
  }
#undef W2L_BG
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_bgemm_f32(const w2l_bgemm_desc* d, const float* A, const float* B, float* C, w2l_stream_t stream) {
  return bgemm_launch(d, A, B, C, false, stream);
}
// the same product with bf16 multiplies (operands rounded to nearest even on the way into LDS), fp32 accumulation and result:
// the attention products of a Transformer block in the mixed-precision mode
W2L_API int w2l_bgemm_bf16(const w2l_bgemm_desc* d, const float* A, const float* B, float* C, w2l_stream_t stream) {
  return bgemm_launch(d, A, B, C, true, stream);
}

static int sm_params(SmP& p, int B, int H, int T, int ldr, int rlo, int W, int n0, float scale) {
  if (B <= 0 || H <= 0 || T <= 0 || (long long)B * H * T > 0x7fffffffLL) return W2L_EINVAL;
  p.B = B; p.H = H; p.T = T; p.ldr = ldr; p.rlo = rlo; p.W = W; p.n0 = n0; p.scale = scale;
  return W2L_OK;
}

W2L_API int w2l_attn_key_lengths_full(const float* inputSizes, const float* fullSize, int B, int Tin, int Tk, int* keyLen,
                                      w2l_stream_t stream) {
  if (!inputSizes || !keyLen || B <= 0 || Tin <= 0 || Tk <= 0) return W2L_EINVAL;
  hipLaunchKernelGGL(attn_key_len_k, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, W2L_S, inputSizes, fullSize, B, Tin, Tk, keyLen);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_attn_key_lengths(const float* inputSizes, int B, int Tin, int Tk, int* keyLen, w2l_stream_t stream) {
  if (!inputSizes || !keyLen || B <= 0 || Tin <= 0 || Tk <= 0) return W2L_EINVAL;
  hipLaunchKernelGGL(attn_key_len_k, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, W2L_S, inputSizes, (const float*)nullptr, B, Tin, Tk, keyLen);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_attn_softmax_forward(float* S, const float* R, const int* keyLen, int B, int H, int T, int ldr, int rlo, int W,
                                     int n0, float scale, w2l_stream_t stream) {
  SmP p{};
  if (!S || sm_params(p, B, H, T, ldr, rlo, W, n0, scale) != W2L_OK) return W2L_EINVAL;
  p.S = S; p.R = R; p.keyLen = keyLen;
  hipLaunchKernelGGL(attn_softmax_fwd_k, dim3((unsigned)((B * H * T + 3) / 4)), dim3(256), 0, W2L_S, p);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_attn_softmax_backward(const float* P, float* dS, float* dR, int B, int H, int T, int ldr, int rlo, int W,
                                      int n0, float scale, w2l_stream_t stream) {
  SmP p{};
  if (!P || !dS || sm_params(p, B, H, T, ldr, rlo, W, n0, scale) != W2L_OK) return W2L_EINVAL;
  p.S = dS; p.P = P; p.dR = dR;
  hipLaunchKernelGGL(attn_softmax_bwd_k, dim3((unsigned)((B * H * T + 3) / 4)), dim3(256), 0, W2L_S, p);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_pool_time_forward(const float* x, float* y, int B, int T, int F, int w, int stride, w2l_stream_t stream) {
  if (!x || !y || B <= 0 || F <= 0 || w <= 0 || stride <= 0 || T < w) return W2L_EINVAL;
  const int To = (T - w) / stride + 1;
  const size_t n = (size_t)B * To * F;
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pool_time_fwd_k, dim3((unsigned)g), dim3(256), 0, W2L_S, x, y, T, To, F, w, stride, n);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API int w2l_pool_time_backward(const float* x, const float* dy, float* dx, int B, int T, int F, int w, int stride,
                                   w2l_stream_t stream) {
  if (!x || !dy || !dx || B <= 0 || F <= 0 || w <= 0 || stride <= 0 || T < w) return W2L_EINVAL;
  const int To = (T - w) / stride + 1;
  const size_t n = (size_t)B * T * F;
  size_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pool_time_bwd_k, dim3((unsigned)g), dim3(256), 0, W2L_S, x, dy, dx, T, To, F, w, stride, n);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
