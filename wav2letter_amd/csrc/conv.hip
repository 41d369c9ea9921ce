// conv.hip -- time convolution (fl::Conv2D kw x 1) forward / backward as an
// IM2COL-FREE implicit GEMM on the fp32 MFMA engine (gemm.hpp).
//
// Reference semantics: cross-correlation over time, y[to] += x[to*stride + tap - pad] * w[tap]
// (recipes/streaming_convnets/inference/inference/module/nn/backend/fbgemm/Conv1dFbGemm.cpp:104-185;
// arch tokens C / C2 / TDS in recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:203-301).
// In the reference this is cudnnConvolutionForward/BackwardData/BackwardFilter on
// WHCN tensors; here activations live FRAME-MAJOR  x[B][T][H][C]  (C fastest) so
// that a GEMM row m = (b, t, h) is a contiguous C-vector and the K index
// (tap, c_in) of the implicit GEMM maps to a row shift of tap frames -- the
// unfolded matrix is never materialised:
//   forward   Y[(b,to,h)][co] = sum_{tap,ci} X[b][to*s+tap-pl][h][ci] * W[(tap,ci)][co]
//   bwd-data  dX[(b,ti,h)][ci] = sum_{tap,co} dY[b][(ti+pl-tap)/s][h][co] * W[(tap,ci)][co]
//   bwd-filt  dW[(tap,ci)][co] = sum_{(b,to,h)} X[b][to*s+tap-pl][h][ci] * dY[(b,to,h)][co]
// Weights are stored [kw][Cin][Cout] (== [K][Cout], "k-rows" B operand).  Small
// output widths (TDS: C = 10/14/18) use the 16x16x4 skinny kernel.
#include <cstdlib>
#include <map>
#include <mutex>

#include "gemm.hpp"

namespace w2l {

struct FastDiv {
  uint32_t mul, shr, d;
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)__umulhi(n, mul) + n) >> shr); }
};
static FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shr = s;
  f.mul = (uint32_t)((((1ull << s) - d) << 32) / d + 1);
  return f;
}

struct ConvGeom {
  int T_src;       // frames of the tensor being read
  int T_rows;      // frames of the row space (output frames for fwd, input frames for bwd-data)
  int H, C;        // rows per frame, channels of the tensor being read
  int sA, sB, off, div;  // src frame = (t_row*sA + tap*sB + off) / div  (must divide, be in [0,T_src))
  FastDiv dH, dT, dC;
};

__device__ __forceinline__ bool conv_src_frame(const ConvGeom& g, int trow, int tap, int& tsrc) {
  int num = trow * g.sA + tap * g.sB + g.off;
  if (num < 0) return false;
  if (g.div != 1) {
    if (num % g.div) return false;
    num /= g.div;
  }
  tsrc = num;
  return num < g.T_src;
}

// A operand of forward / backward-data: element (kk=(tap,c), m=(b,trow,h))
struct ConvAOp {
  static constexpr bool kFast = false;
  struct Ptrs {};
  __device__ __forceinline__ void init(Ptrs&, int, int) const {}
  __device__ __forceinline__ void load_fast(float (&)[16], const Ptrs&, int) const {}
  const float* src;
  ConvGeom g;
  int M, K;
  static constexpr int kPad = 1;
  static constexpr int kPadSkinny = 2;

  template <int BI, int BK>
  __device__ __forceinline__ void load_t(float (&r)[16], int i0, int k0, int tid) const {
    constexpr int RPP = 256 / BK;
    constexpr int NP = BI / RPP;
    const int kk = k0 + tid % BK;
    const int rr = tid / BK;
    const uint32_t tap = g.dC.div((uint32_t)kk);
    const int c = kk - (int)tap * g.C;
    const bool kok = kk < K;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int m = i0 + rr + RPP * j;
      float v = 0.f;
      if (kok && m < M) {
        const uint32_t bt = g.dH.div((uint32_t)m);
        const int h = m - (int)bt * g.H;
        const uint32_t b = g.dT.div(bt);
        const int trow = (int)bt - (int)b * g.T_rows;
        int ts;
        if (conv_src_frame(g, trow, (int)tap, ts))
          v = src[(((size_t)b * g.T_src + ts) * g.H + h) * g.C + c];
      }
      r[j] = v;
    }
  }
  template <int BI, int BK>
  __device__ __forceinline__ void store_t(float* lds, int ldS, const float (&r)[16], int tid) const {
    constexpr int RPP = 256 / BK;
    constexpr int NP = BI / RPP;
    const int kc = tid % BK, rr = tid / BK;
#pragma unroll
    for (int j = 0; j < NP; ++j) lds[kc * ldS + rr + RPP * j] = r[j];
  }
  __device__ __forceinline__ void load(float (&r)[16], int i0, int k0, int tid) const { load_t<128, 32>(r, i0, k0, tid); }
  __device__ __forceinline__ void store(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<128, 32>(lds, ldS, r, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void load_bk(float (&r)[16], int i0, int k0, int tid) const { load_t<BM, BK>(r, i0, k0, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void store_bk(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<BM, BK>(lds, ldS, r, tid); }
};

// B operand of backward-data: element (kk=(tap,co), j=ci) = W[(tap*Cin + ci)*Cout + co]
struct ConvWTOp {
  static constexpr bool kFast = false;
  struct Ptrs {};
  __device__ __forceinline__ void init(Ptrs&, int, int) const {}
  __device__ __forceinline__ void load_fast(float (&)[16], const Ptrs&, int) const {}
  const float* w;
  int Cin, Cout, K;  // K = kw*Cout
  FastDiv dCo;
  static constexpr int kPad = 1;

  __device__ __forceinline__ void load(float (&r)[16], int n0, int k0, int tid) const {
    const int kk = k0 + tid % 32;
    const int rr = tid / 32;
    const uint32_t tap = dCo.div((uint32_t)kk);
    const int co = kk - (int)tap * Cout;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ci = n0 + rr + 8 * j;
      r[j] = (kk < K && ci < Cin) ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* lds, int ldS, const float (&r)[16], int tid) const {
    const int kc = tid % 32, rr = tid / 32;
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[kc * ldS + rr + 8 * j] = r[j];
  }
  template <int BN, int BK>
  __device__ __forceinline__ void load_small(float (&r)[(BK * BN + 255) / 256], int n0, int k0, int tid) const {
    constexpr int NE = (BK * BN + 255) / 256;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      const int k = e % BK, n = e / BK;
      const int kk = k0 + k, ci = n0 + n;
      const uint32_t tap = dCo.div((uint32_t)kk);
      const int co = kk - (int)tap * Cout;
      r[j] = (e < BK * BN && kk < K && ci < Cin) ? w[((size_t)tap * Cin + ci) * Cout + co] : 0.f;
    }
  }
  template <int BN, int BK>
  __device__ __forceinline__ void store_small(float* lds, int ldS, const float (&r)[(BK * BN + 255) / 256], int tid) const {
    constexpr int NE = (BK * BN + 255) / 256;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      const int e = tid + 256 * j;
      if (e < BK * BN) lds[(e % BK) * ldS + e / BK] = r[j];
    }
  }
};

// A operand of backward-filter: element (k = m = (b,to,h), i = (tap,ci)) -- "k-rows"
struct ConvFilterAOp {
  static constexpr bool kFast = false;
  struct Ptrs {};
  __device__ __forceinline__ void init(Ptrs&, int, int) const {}
  __device__ __forceinline__ void load_fast(float (&)[16], const Ptrs&, int) const {}
  const float* src;
  ConvGeom g;  // T_rows = To (row space of dY), reads x with C = Cin
  int Mi;      // kw*Cin  (extent of i)
  int Kred;    // B*To*H  (reduction length)
  static constexpr int kPad = 4;
  static constexpr int kPadSkinny = 4;

  template <int BI, int BK>
  __device__ __forceinline__ void load_t(float (&r)[16], int i0, int k0, int tid) const {
    constexpr int RPP = 256 / BI;  // k-rows per pass
    constexpr int NP = BK / RPP;
    const int i = i0 + tid % BI;
    const int kr = tid / BI;
    const uint32_t tap = g.dC.div((uint32_t)i);
    const int c = i - (int)tap * g.C;
    const bool iok = i < Mi;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int m = k0 + kr + RPP * j;
      float v = 0.f;
      if (iok && m < Kred) {
        const uint32_t bt = g.dH.div((uint32_t)m);
        const int h = m - (int)bt * g.H;
        const uint32_t b = g.dT.div(bt);
        const int trow = (int)bt - (int)b * g.T_rows;
        int ts;
        if (conv_src_frame(g, trow, (int)tap, ts))
          v = src[(((size_t)b * g.T_src + ts) * g.H + h) * g.C + c];
      }
      r[j] = v;
    }
  }
  template <int BI, int BK>
  __device__ __forceinline__ void store_t(float* lds, int ldS, const float (&r)[16], int tid) const {
    constexpr int RPP = 256 / BI;
    constexpr int NP = BK / RPP;
    const int ic = tid % BI, kr = tid / BI;
#pragma unroll
    for (int j = 0; j < NP; ++j) lds[(kr + RPP * j) * ldS + ic] = r[j];
  }
  __device__ __forceinline__ void load(float (&r)[16], int i0, int k0, int tid) const { load_t<128, 32>(r, i0, k0, tid); }
  __device__ __forceinline__ void store(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<128, 32>(lds, ldS, r, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void load_bk(float (&r)[16], int i0, int k0, int tid) const { load_t<BM, BK>(r, i0, k0, tid); }
  template <int BM, int BK>
  __device__ __forceinline__ void store_bk(float* lds, int ldS, const float (&r)[16], int tid) const { store_t<BM, BK>(lds, ldS, r, tid); }
};

// column sums: out[n] = sum_m x[m][n]   (bias gradients).  Two deterministic passes, no atomics: row-block
// partial sums (V-wide loads, 4 rows in flight per wave) into the library scratch, then a sum over row-blocks
// in fixed order.  HBM-bound: M*N*4 bytes read once.
float* sk_scratch(hipStream_t s, size_t bytes);
constexpr int kColsumMaxParts = 128;

template <int V>
__global__ __launch_bounds__(256) void colsum_partial_k(const float* __restrict__ x, float* __restrict__ partial,
                                                        size_t M, int N, size_t rowsPerBlock) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * V;
  __shared__ float sm[4][64 * V];
  size_t m0 = (size_t)blockIdx.y * rowsPerBlock;
  size_t m1 = m0 + rowsPerBlock;
  if (m1 > M) m1 = M;
  vec_t a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (n < N) {
    const float* col = x + n;
    size_t m = m0 + wave;
    for (; m + 12 < m1; m += 16) {
      vec_t v0 = __builtin_nontemporal_load((const vec_t*)(col + m * N));
      vec_t v1 = __builtin_nontemporal_load((const vec_t*)(col + (m + 4) * N));
      vec_t v2 = __builtin_nontemporal_load((const vec_t*)(col + (m + 8) * N));
      vec_t v3 = __builtin_nontemporal_load((const vec_t*)(col + (m + 12) * N));
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; m < m1; m += 4) a0 += *(const vec_t*)(col + m * N);
  }
  vec_t a = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int v = 0; v < V; ++v) sm[wave][lane * V + v] = a[v];
  __syncthreads();
  for (int c = threadIdx.x; c < 64 * V; c += 256) {
    int nn = blockIdx.x * 64 * V + c;
    if (nn < N) partial[(size_t)blockIdx.y * N + nn] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
  }
}

// 64 columns per workgroup; the four waves take every fourth row-block (independent loads in flight), fixed-order sums
__global__ __launch_bounds__(256) void colsum_finish_k(const float* __restrict__ partial, float* __restrict__ out, int parts, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  __shared__ float sm[4][64];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int p = wave;
    for (; p + 12 < parts; p += 16) {
      const float v0 = partial[(size_t)p * N + n], v1 = partial[(size_t)(p + 4) * N + n];
      const float v2 = partial[(size_t)(p + 8) * N + n], v3 = partial[(size_t)(p + 12) * N + n];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; p < parts; p += 4) s0 += partial[(size_t)p * N + n];
  }
  sm[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) out[n] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// out[n] = sum_g tmp[g N + n], g ascending
__global__ __launch_bounds__(64) void colsum_fold_k(const float* __restrict__ tmp, float* __restrict__ out, int G, int N) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += tmp[(size_t)g * N + n];
  out[n] = s;
}

int colsum(const float* x, float* out, size_t M, int N, hipStream_t s) {
  if (N <= 0) return W2L_OK;
  if (M == 0) {
    W2L_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s));
    return W2L_OK;
  }
  // NARROW matrices (the bias gradient of a TDS convolution: [B T H][C], C = 10 .. 27): a row is a fraction of one wave's
  // 64 lanes and an odd C forbids vector loads (colsum_partial_k<1> on [957 k][27]: 304 us, 0.34 TB/s, run j4).  G
  // consecutive rows are read as ONE row of G N floats (same bytes), summed by the wide kernels, and the G groups folded.
  if (N < 64 && M >= 4096) {
    int G = 0;
    for (int g = 16; g >= 4; g >>= 1)
      if (M % (size_t)g == 0 && (g * N) % 4 == 0) { G = g; break; }
    if (G) {
      float* scratch = sk_scratch(s, kSkScratchBytes);
      if (!scratch) return W2L_EHIP;
      float* tmp = scratch + (size_t)(1 << 20);          // behind the row-block partials of the inner call (<= 128 x 1008 floats)
      const int st = colsum(x, tmp, M / (size_t)G, G * N, s);
      if (st != W2L_OK) return st;
      hipLaunchKernelGGL(colsum_fold_k, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, tmp, out, G, N);
      W2L_LAUNCH_CHECK();
      return W2L_OK;
    }
  }
  size_t rowsPerBlock = (M + kColsumMaxParts - 1) / kColsumMaxParts;
  if (rowsPerBlock < 64) rowsPerBlock = 64;
  const int parts = (int)((M + rowsPerBlock - 1) / rowsPerBlock);
  float* partial = sk_scratch(s, kSkScratchBytes);  // shared 64 MiB stream scratch
  if (!partial || (size_t)parts * N * sizeof(float) > (size_t)kSkSlots * 2 * kSlabFloats * sizeof(float)) return W2L_EHIP;
  const bool al = (((uintptr_t)x) & 15) == 0;
  const int V = (al && N % 4 == 0) ? 4 : (al && N % 2 == 0) ? 2 : 1;
  dim3 grid((unsigned)((N + 64 * V - 1) / (64 * V)), (unsigned)parts);
  if (V == 4)
    hipLaunchKernelGGL(colsum_partial_k<4>, grid, dim3(256), 0, s, x, partial, M, N, rowsPerBlock);
  else if (V == 2)
    hipLaunchKernelGGL(colsum_partial_k<2>, grid, dim3(256), 0, s, x, partial, M, N, rowsPerBlock);
  else
    hipLaunchKernelGGL(colsum_partial_k<1>, grid, dim3(256), 0, s, x, partial, M, N, rowsPerBlock);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_finish_k, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, partial, out, parts, N);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

static inline int out_len(int T, int kw, int stride, int padl, int padr) {
  int n = T + padl + padr - kw;
  return n < 0 ? 0 : n / stride + 1;
}

static inline int pick_vec_rows(const float* p, int ld, int extent) {
  if ((((uintptr_t)p) & 15) == 0 && ld % 4 == 0 && extent % 4 == 0) return 4;
  if ((((uintptr_t)p) & 7) == 0 && ld % 2 == 0 && extent % 2 == 0) return 2;
  return 1;
}

template <class AOp>
static int conv_launch_rowsB(const AOp& a, const float* Bp, int ldb, int N, int K, const GemmOut& o, int epi,
                             int splitk, hipStream_t s) {
  // B is a plain "k-rows" matrix [K][N]
  if (N <= 16) return launch_skinny<AOp, PlainOp<false, 1>, 16>(a, PlainOp<false, 1>{Bp, ldb, N, K}, o, epi, splitk, s);
  if (N <= 32) return launch_skinny<AOp, PlainOp<false, 1>, 32>(a, PlainOp<false, 1>{Bp, ldb, N, K}, o, epi, splitk, s);
  int v = pick_vec_rows(Bp, ldb, N);
  if (v == 4) return launch128(a, PlainOp<false, 4>{Bp, ldb, N, K}, o, epi, splitk, s);
  if (v == 2) return launch128(a, PlainOp<false, 2>{Bp, ldb, N, K}, o, epi, splitk, s);
  return launch128(a, PlainOp<false, 1>{Bp, ldb, N, K}, o, epi, splitk, s);
}

// conv_tds.hip: slab-in-LDS kernels for few-channel convolutions (TDS, C2 sub-sampling)
bool tds_conv_applicable(const w2l_conv_desc* d);
int tds_conv_forward(const w2l_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                     hipStream_t s);
int tds_conv_backward_data(const w2l_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                           const float* add, hipStream_t s);
int tds_conv_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                             hipStream_t s);

// =====================================================================================================
// conv_glu path (fl::Conv2D kw x 1 over time with H == 1, stride 1: recipes/conv_glu/*/network.arch).
// In the frame-major layout [B][T][C] the im2col row of output frame (b, t) -- frames t .. t+kw-1 -- is ONE contiguous
// run of kw*C_in floats, so the convolution is a plain GEMM on OVERLAPPING rows: A = x viewed with leading dimension
// C_in, K = kw*C_in.  No gather, no index arithmetic: the operand goes through the LDS-DMA kernels like a dense
// matrix (the register-staged implicit-GEMM operand ran the C4 layers at 78-84 TF/s forward, 54-70 backward-data,
// 60-73 backward-filter; profiles/r01_run44_convglu.log).  Rows are indexed by the GLOBAL frame r = b*Tp + t', the
// kw-1 rows per utterance that straddle into the next one are dropped by the epilogue's row remap.
//   forward : y  = X[r][(tap,ci)] . W[(tap,ci)][co]                    (X = x, or its zero-padded copy)
//   bwd-data: dx = DYP[r][(tap',co)] . Wf[(tap',co)][ci],  Wf[tap'] = W[kw-1-tap']^T, DYP = dy re-pitched to Tp frames
//             per utterance with kw-1 zero frames after (and, for the first utterance, before) each one
//   bwd-filt: dw[(tap,ci)][co] = sum_r X[r][(tap,ci)] . DYP[r][co]     (X as a k-row operand: same memory)
// K need not be a multiple of the K tile: the rows past K of the k-row operand lie outside its byte range and read as
// zeros (buffer addressing), what the overlapping operand holds there is finite activation data.
float* conv_scratch(hipStream_t s, size_t bytes) {  // library-owned, grows on demand, one buffer per stream
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  auto& e = cache[{dev, s}];
  if (e.second < bytes) {
    if (e.first) { (void)hipStreamSynchronize(s); (void)hipFree(e.first); }
    e.first = nullptr; e.second = 0;
    void* p = nullptr;
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    e.first = (float*)p; e.second = want;
  }
  return e.first;
}

// dst[b][t'][c] = src[b][t' - off][c] if 0 <= t' - off < Tsrc else 0     (frames of C floats; V floats per access)
template <int V>
__global__ __launch_bounds__(256) void pad_frames_k(const float* __restrict__ src, float* __restrict__ dst, int Tsrc, int Tdst,
                                                    int C, int off) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const int b = blockIdx.y;
  const int cv = C / V;                                   // vectors per frame
  const size_t nv = (size_t)Tdst * cv;
  const vec_t* sv = (const vec_t*)src + (size_t)b * Tsrc * cv;
  vec_t* dv = (vec_t*)dst + (size_t)b * nv;
  const size_t shift = (size_t)off * cv, lim = (size_t)Tsrc * cv;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < nv; e += (size_t)gridDim.x * 256) {
    vec_t v = 0.f;
    if (e >= shift && e - shift < lim) v = __builtin_nontemporal_load(sv + (e - shift));  // frames are contiguous: no division
    dv[e] = v;
  }
}

static bool glds_conv_enabled() {
  const char* e = tune_env("W2L_CONV_GLDS");
  return !(e && e[0] == '0');
}
static bool glds_conv_applicable(const w2l_conv_desc* d) {
  return glds_conv_enabled() && d->H == 1 && d->stride == 1 && d->kw * d->Cin >= 64 && d->Cout >= 32 && d->Cin >= 4 &&
         (int64_t)d->B * (d->T + d->padl + d->padr) < (1ll << 30);
}
static int pad_frames(const float* src, float* dst, int B, int Tsrc, int Tdst, int C, int off, hipStream_t s) {
  const bool a16 = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0, a8 = ((((uintptr_t)src) | ((uintptr_t)dst)) & 7) == 0;
  const int V = (C % 4 == 0 && a16) ? 4 : (C % 2 == 0 && a8) ? 2 : 1;
  unsigned gx = (unsigned)(((size_t)Tdst * (C / V) + 255) / 256);
  if (gx > 2048) gx = 2048;
  if (V == 4) hipLaunchKernelGGL(pad_frames_k<4>, dim3(gx, (unsigned)B), dim3(256), 0, s, src, dst, Tsrc, Tdst, C, off);
  else if (V == 2) hipLaunchKernelGGL(pad_frames_k<2>, dim3(gx, (unsigned)B), dim3(256), 0, s, src, dst, Tsrc, Tdst, C, off);
  else hipLaunchKernelGGL(pad_frames_k<1>, dim3(gx, (unsigned)B), dim3(256), 0, s, src, dst, Tsrc, Tdst, C, off);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

static int glds_conv_forward(const w2l_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                             hipStream_t s) {
  const int Tp = d->T + d->padl + d->padr, To = Tp - d->kw + 1;
  const int K = d->kw * d->Cin;
  const float* xs = x;
  if (d->padl || d->padr) {
    float* xp = conv_scratch(s, (size_t)d->B * Tp * d->Cin * sizeof(float));
    if (!xp) return W2L_EUNSUPPORTED;
    int st = pad_frames(x, xp, d->B, d->T, Tp, d->Cin, d->padl, s);
    if (st) return st;
    xs = xp;
  }
  GemmOut o{y, bias, d->B * Tp - d->kw + 1, d->Cout, K, d->Cout, 0};
  gemm_set_row_remap(o, Tp, To, 0);
  const int epi = (bias ? EPI_BIAS : 0) | (relu ? EPI_RELU : 0);
  return gemm_glds_raw(xs, d->Cin, true, (size_t)d->B * Tp * d->Cin * sizeof(float), w, d->Cout, false,
                       (size_t)K * d->Cout * sizeof(float), o, epi, s);
}

// Mixed-precision mode (w2l_set_matmul_precision(1): the network passes of configs 3 / 5): the same overlapping-row GEMMs on
// bf16 images.  The activation image keeps the frame pitch (ld = channels, NOT padded: row m of the GEMM must run on into
// frame m + 1), so the K padding of a row reads the next frame's data -- finite numbers against the zero rows that pad the
// weight image -- and the last rows read past the image, where the buffer range returns zeros.
static inline size_t al4(size_t floats) { return (floats + 3) & ~(size_t)3; }
static bool bf16_conv_ok(int chan, size_t rows) { return chan % 16 == 0 && rows * (size_t)chan * 2 < 0x7fffffffull; }

static int glds_conv_forward_bf16(const w2l_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int relu,
                                  hipStream_t s) {
  const int Tp = d->T + d->padl + d->padr, To = Tp - d->kw + 1;
  const int K = d->kw * d->Cin, Kp = (K + 63) / 64 * 64;
  const size_t rows = (size_t)d->B * Tp, xN = rows * d->Cin;
  const bool padded = d->padl || d->padr;
  const size_t xpF = padded ? al4(xN) : 0, imgF = al4((xN + 1) / 2), wimgF = al4(((size_t)d->Cout * Kp + 1) / 2);
  float* sc = conv_scratch(s, (xpF + imgF + wimgF) * sizeof(float));
  if (!sc) return W2L_EUNSUPPORTED;
  const float* xs = x;
  if (padded) {
    int st = pad_frames(x, sc, d->B, d->T, Tp, d->Cin, d->padl, s);
    if (st) return st;
    xs = sc;
  }
  uint16_t* img = (uint16_t*)(sc + xpF);
  uint16_t* wimg = (uint16_t*)(sc + xpF + imgF);
  int st = w2l_bf16_convert(xs, rows, d->Cin, (size_t)d->Cin, img, (size_t)d->Cin, nullptr, 0, (w2l_stream_t)s);
  if (st) return st;
  st = w2l_bf16_convert(w, (size_t)K, d->Cout, (size_t)d->Cout, nullptr, 0, wimg, (size_t)Kp, (w2l_stream_t)s);   // [Cout][Kp], k contiguous
  if (st) return st;
  GemmOut o{y, bias, d->B * Tp - d->kw + 1, d->Cout, K, d->Cout, 0};
  gemm_set_row_remap(o, Tp, To, 0);
  const int epi = (bias ? EPI_BIAS : 0) | (relu ? EPI_RELU : 0);
  return gemm_bf16_images(img, d->Cin, 2ull * xN, wimg, Kp, 0, o, epi, s);
}

static int glds_conv_backward_data_bf16(const w2l_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                                        const float* add, hipStream_t s) {
  const int Tp = d->T + d->padl + d->padr, To = Tp - d->kw + 1;
  const int K = d->kw * d->Cout, Kp = (K + 63) / 64 * 64;
  const size_t front = (size_t)(d->kw - 1) * d->Cout, dypN = (size_t)d->B * Tp * d->Cout;
  const size_t wfN = (size_t)d->kw * d->Cout * d->Cin;
  const size_t dypF = al4(front + dypN), wfF = al4(wfN), imgF = al4((front + dypN + 1) / 2), wimgF = al4(((size_t)d->Cin * Kp + 1) / 2);
  float* sc = conv_scratch(s, (dypF + wfF + imgF + wimgF) * sizeof(float));
  if (!sc) return W2L_EUNSUPPORTED;
  float* wf = sc + dypF;
  uint16_t* img = (uint16_t*)(sc + dypF + wfF);
  uint16_t* wimg = (uint16_t*)(sc + dypF + wfF + imgF);
  W2L_HIP_CHECK(hipMemsetAsync(sc, 0, front * sizeof(float), s));
  int st = pad_frames(dy, sc + front, d->B, To, Tp, d->Cout, 0, s);
  if (st) return st;
  for (int tap = 0; tap < d->kw; ++tap) {  // wf[kw-1-tap] = w[tap]^T
    st = w2l_transpose(w + (size_t)tap * d->Cin * d->Cout, wf + (size_t)(d->kw - 1 - tap) * d->Cout * d->Cin, 1, d->Cin, d->Cout,
                       (w2l_stream_t)s);
    if (st) return st;
  }
  // the re-pitched dy (kw - 1 zero frames in front) as ONE image of (front + dypN) / Cout rows; wf [K][Cin] -> [Cin][Kp]
  st = w2l_bf16_convert(sc, (front + dypN) / d->Cout, d->Cout, (size_t)d->Cout, img, (size_t)d->Cout, nullptr, 0, (w2l_stream_t)s);
  if (st) return st;
  st = w2l_bf16_convert(wf, (size_t)K, d->Cin, (size_t)d->Cin, nullptr, 0, wimg, (size_t)Kp, (w2l_stream_t)s);
  if (st) return st;
  GemmOut o{dx, nullptr, d->B * Tp, d->Cin, K, d->Cin, 0};
  if (d->padl || d->padr) gemm_set_row_remap(o, Tp, d->T, d->padl);
  int epi = 0;
  if (add) { o.addend = add; epi |= EPI_ACCUM; }
  else if (accumulate) epi |= EPI_ACCUM;
  return gemm_bf16_images(img, d->Cout, 2ull * (front + dypN), wimg, Kp, 0, o, epi, s);
}

// scratch: [kw-1 zero frames | dyp [B][Tp][Cout]] [wf [kw][Cout][Cin]] ([xp [B][Tp][Cin]] for the filter gradient)
static int glds_conv_backward_data(const w2l_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                                   const float* add, hipStream_t s) {
  const int Tp = d->T + d->padl + d->padr, To = Tp - d->kw + 1;
  const size_t front = (size_t)(d->kw - 1) * d->Cout, dypN = (size_t)d->B * Tp * d->Cout;
  const size_t wfN = (size_t)d->kw * d->Cout * d->Cin;
  const size_t dypAl = (front + dypN + 3) & ~(size_t)3;
  float* sc = conv_scratch(s, (dypAl + wfN) * sizeof(float));
  if (!sc) return W2L_EUNSUPPORTED;
  float* wf = sc + dypAl;
  W2L_HIP_CHECK(hipMemsetAsync(sc, 0, front * sizeof(float), s));
  int st = pad_frames(dy, sc + front, d->B, To, Tp, d->Cout, 0, s);
  if (st) return st;
  for (int tap = 0; tap < d->kw; ++tap) {  // wf[kw-1-tap] = w[tap]^T
    st = w2l_transpose(w + (size_t)tap * d->Cin * d->Cout, wf + (size_t)(d->kw - 1 - tap) * d->Cout * d->Cin, 1, d->Cin, d->Cout,
                       (w2l_stream_t)s);
    if (st) return st;
  }
  const int K = d->kw * d->Cout;
  GemmOut o{dx, nullptr, d->B * Tp, d->Cin, K, d->Cin, 0};
  if (d->padl || d->padr) gemm_set_row_remap(o, Tp, d->T, d->padl);
  int epi = 0;
  if (add) { o.addend = add; epi |= EPI_ACCUM; }
  else if (accumulate) epi |= EPI_ACCUM;
  return gemm_glds_raw(sc, d->Cout, true, (front + dypN) * sizeof(float), wf, d->Cin, false, wfN * sizeof(float), o, epi, s);
}

static int glds_conv_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                     hipStream_t s) {
  const int Tp = d->T + d->padl + d->padr, To = Tp - d->kw + 1;
  const size_t dypN = ((size_t)d->B * Tp * d->Cout + 3) & ~(size_t)3, xpN = (size_t)d->B * Tp * d->Cin;
  const bool padded = d->padl || d->padr;
  float* sc = conv_scratch(s, (dypN + (padded ? xpN : 0)) * sizeof(float));
  if (!sc) return W2L_EUNSUPPORTED;
  int st = pad_frames(dy, sc, d->B, To, Tp, d->Cout, 0, s);
  if (st) return st;
  const float* xs = x;
  if (padded) {
    st = pad_frames(x, sc + dypN, d->B, d->T, Tp, d->Cin, d->padl, s);
    if (st) return st;
    xs = sc + dypN;
  }
  GemmOut o{dw, nullptr, d->kw * d->Cin, d->Cout, d->B * Tp, d->Cout, 0};
  st = gemm_glds_raw(xs, d->Cin, false, xpN * sizeof(float), sc, d->Cout, false, (size_t)d->B * Tp * d->Cout * sizeof(float), o, 0, s);
  if (st) return st;
  if (dbias) return colsum(dy, dbias, (size_t)d->B * To, d->Cout, s);
  return W2L_OK;
}

static bool tds_path() {
  const char* e = tune_env("W2L_CONV_TDS");
  return !(e && e[0] == '0');
}

}  // namespace w2l

using namespace w2l;

W2L_API int w2l_conv_out_len(int T, int kw, int stride, int padl, int padr) {
  return out_len(T, kw, stride, padl, padr);
}

// Flashlight PaddingMode::SAME for one axis (pad = -1 in arch files): the same
// p = ceil(total/2) on both sides.
W2L_API int w2l_conv_same_pad(int T, int kw, int stride) {
  int total = (T % stride == 0) ? (kw - 1) - stride + 1 : (kw - 1) - (T % stride) + 1;
  if (total < 0) total = 0;
  return (total + 1) / 2;
}

static int check_desc(const w2l_conv_desc* d) {
  if (!d || d->B <= 0 || d->T <= 0 || d->H <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->kw <= 0 ||
      d->stride <= 0 || d->padl < 0 || d->padr < 0)
    return W2L_EINVAL;
  if (out_len(d->T, d->kw, d->stride, d->padl, d->padr) <= 0) return W2L_EINVAL;
  if ((int64_t)d->B * d->T * d->H >= (1ll << 31)) return W2L_EUNSUPPORTED;
  return W2L_OK;
}

W2L_API int w2l_conv_forward(const w2l_conv_desc* d, const float* x, const float* w, const float* bias,
                             float* y, int relu, w2l_stream_t stream) {
  int st = check_desc(d);
  if (st) return st;
  if (!x || !w || !y) return W2L_EINVAL;
  if (tds_path() && tds_conv_applicable(d)) {
    st = tds_conv_forward(d, x, w, bias, y, relu, (hipStream_t)stream);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  if (glds_conv_applicable(d)) {
    if (matmul_bf16_mode() && bf16_conv_ok(d->Cin, (size_t)d->B * (d->T + d->padl + d->padr))) {
      st = glds_conv_forward_bf16(d, x, w, bias, y, relu, (hipStream_t)stream);
      if (st != W2L_EUNSUPPORTED) return st;
    }
    st = glds_conv_forward(d, x, w, bias, y, relu, (hipStream_t)stream);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  const int To = out_len(d->T, d->kw, d->stride, d->padl, d->padr);
  const int M = d->B * To * d->H, K = d->kw * d->Cin, N = d->Cout;
  ConvGeom g{d->T, To, d->H, d->Cin, d->stride, 1, -d->padl, 1,
             make_fastdiv((uint32_t)d->H), make_fastdiv((uint32_t)To), make_fastdiv((uint32_t)d->Cin)};
  ConvAOp a{x, g, M, K};
  GemmOut o{y, bias, M, N, K, N, 0, nullptr, 1.f};
  int epi = (bias ? EPI_BIAS : 0) | (relu ? EPI_RELU : 0);
  return conv_launch_rowsB(a, w, N, N, K, o, epi, 1, (hipStream_t)stream);
}

W2L_API int w2l_conv_backward_data(const w2l_conv_desc* d, const float* dy, const float* w, float* dx,
                                   int accumulate, w2l_stream_t stream) {
  int st = check_desc(d);
  if (st) return st;
  if (!dy || !w || !dx) return W2L_EINVAL;
  if (tds_path() && tds_conv_applicable(d)) {
    st = tds_conv_backward_data(d, dy, w, dx, accumulate, nullptr, (hipStream_t)stream);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  if (glds_conv_applicable(d)) {
    if (matmul_bf16_mode() && bf16_conv_ok(d->Cout, (size_t)d->B * (d->T + d->padl + d->padr) + d->kw)) {
      st = glds_conv_backward_data_bf16(d, dy, w, dx, accumulate, nullptr, (hipStream_t)stream);
      if (st != W2L_EUNSUPPORTED) return st;
    }
    st = glds_conv_backward_data(d, dy, w, dx, accumulate, nullptr, (hipStream_t)stream);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  const int To = out_len(d->T, d->kw, d->stride, d->padl, d->padr);
  const int M = d->B * d->T * d->H, K = d->kw * d->Cout, N = d->Cin;
  // rows are input frames ti; source frame of dY = (ti + padl - tap) / stride
  ConvGeom g{To, d->T, d->H, d->Cout, 1, -1, d->padl, d->stride,
             make_fastdiv((uint32_t)d->H), make_fastdiv((uint32_t)d->T), make_fastdiv((uint32_t)d->Cout)};
  ConvAOp a{dy, g, M, K};
  ConvWTOp b{w, d->Cin, d->Cout, K, make_fastdiv((uint32_t)d->Cout)};
  GemmOut o{dx, nullptr, M, N, K, N, 0, nullptr, 1.f};
  int epi = accumulate ? EPI_ACCUM : 0;
  hipStream_t s = (hipStream_t)stream;
  if (N <= 16) return launch_skinny<ConvAOp, ConvWTOp, 16>(a, b, o, epi, 1, s);
  if (N <= 32) return launch_skinny<ConvAOp, ConvWTOp, 32>(a, b, o, epi, 1, s);
  return launch128(a, b, o, epi, 1, s);
}

// dx = add + backward-data(dy): the residual / upstream gradient joins in the epilogue instead of
// a device-to-device copy followed by an accumulating launch (TDS block backward).
W2L_API int w2l_conv_backward_data_add(const w2l_conv_desc* d, const float* dy, const float* w, const float* add,
                                       float* dx, w2l_stream_t stream) {
  int st = check_desc(d);
  if (st) return st;
  if (!dy || !w || !dx || !add) return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (tds_path() && tds_conv_applicable(d)) {
    st = tds_conv_backward_data(d, dy, w, dx, 0, add, s);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  if (glds_conv_applicable(d)) {
    if (matmul_bf16_mode() && bf16_conv_ok(d->Cout, (size_t)d->B * (d->T + d->padl + d->padr) + d->kw)) {
      st = glds_conv_backward_data_bf16(d, dy, w, dx, 0, add, s);
      if (st != W2L_EUNSUPPORTED) return st;
    }
    st = glds_conv_backward_data(d, dy, w, dx, 0, add, s);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  if (dx != add)
    W2L_HIP_CHECK(hipMemcpyAsync(dx, add, (size_t)d->B * d->T * d->H * d->Cin * sizeof(float), hipMemcpyDeviceToDevice, s));
  return w2l_conv_backward_data(d, dy, w, dx, 1, stream);
}

W2L_API int w2l_conv_backward_filter(const w2l_conv_desc* d, const float* x, const float* dy, float* dw,
                                     float* dbias, w2l_stream_t stream) {
  int st = check_desc(d);
  if (st) return st;
  if (!x || !dy || !dw) return W2L_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (tds_path() && tds_conv_applicable(d)) {
    st = tds_conv_backward_filter(d, x, dy, dw, dbias, s);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  if (glds_conv_applicable(d)) {
    st = glds_conv_backward_filter(d, x, dy, dw, dbias, s);
    if (st != W2L_EUNSUPPORTED) return st;
  }
  const int To = out_len(d->T, d->kw, d->stride, d->padl, d->padr);
  const int Kred = d->B * To * d->H, Mi = d->kw * d->Cin, N = d->Cout;
  ConvGeom g{d->T, To, d->H, d->Cin, d->stride, 1, -d->padl, 1,
             make_fastdiv((uint32_t)d->H), make_fastdiv((uint32_t)To), make_fastdiv((uint32_t)d->Cin)};
  ConvFilterAOp a{x, g, Mi, Kred};
  GemmOut o{dw, nullptr, Mi, N, Kred, N, 0, nullptr, 1.f};
  // small output, long reduction: split K until the grid covers the chip a few times
  const bool skinny = N <= 32;
  const int bm = skinny ? 256 : 128, bn = skinny ? (N <= 16 ? 16 : 32) : 128, bk = skinny ? 16 : 32;
  const int tiles = ((Mi + bm - 1) / bm) * ((N + bn - 1) / bn);
  const int kTiles = (Kred + bk - 1) / bk;
  int splitk = (1024 + tiles - 1) / tiles;
  if (splitk > kTiles / 8) splitk = kTiles / 8;
  if (splitk < 1) splitk = 1;
  int epi = 0;
  if (!skinny) splitk = 1;  // the 128x128 engine splits K itself (stream-K, deterministic)
  if (splitk > 1) {
    W2L_HIP_CHECK(hipMemsetAsync(dw, 0, (size_t)Mi * N * sizeof(float), s));
    epi = EPI_ATOMIC;
  }
  st = conv_launch_rowsB(a, dy, N, N, Kred, o, epi, splitk, s);
  if (st) return st;
  if (dbias) return colsum(dy, dbias, (size_t)Kred, N, s);
  return W2L_OK;
}

W2L_API int w2l_colsum(const float* x, float* out, size_t M, int N, w2l_stream_t stream) {
  if (!x || !out || N <= 0) return W2L_EINVAL;
  return colsum(x, out, M, N, (hipStream_t)stream);
}
