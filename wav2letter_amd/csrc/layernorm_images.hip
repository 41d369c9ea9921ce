// layernorm_images.hip -- the per-frame LayerNorm of the mixed-precision mode writing the bf16 operand images of its result in
// the SAME pass (BASELINE configs 3 and 5: the TDS block's and the Transformer block's LayerNorm output is the A operand of the
// next fl::Linear, its gradient the operand of the previous one's backward products; before this file every such matrix was
// written in fp32, read back by a conversion launch and written again twice -- convert.hip).
//
//   forward   r = dropout(a) + x,  y = LayerNorm(r)                 (elementwise.hip::residual_ln_small_k)
//   backward  dr = LayerNorm backward of (r -> y) given dy (+ the ReLU / dropout mask of the producer: dmask)   (ln_bwd_small_k)
//   images    row-major [row][ldRows] and transposed [column][ldTrans] bf16 (nearest even) of y, resp. of dr -- optionally of
//             dropout(dr) with the library's stateless hash over the flat index: what w2l_bf16_convert_dropout produced
//
// The transposed image needs runs along the ROW index, so one workgroup of 16 waves owns 16 consecutive rows (a wave normalises one
// row, the row in registers between statistics and apply, wave reductions only -- no block barrier per row), leaves the rounded
// rows in an LDS tile [16][inner] and writes the tile out column by column as 32-byte runs.  HBM-bound like the kernels it
// replaces: + 4 bytes per element of writes, instead of a second kernel's 4 read + 4 written.
// Reference: fl::LayerNorm in the arch grammar (recipes/joint_training_vox_populi/cpc/SequentialBuilder.cpp:358-377); the casts of
// the AMP mode recipes/slimIPL/src/Train.cpp:209-216, :1681-1760.
#include "common.hpp"

namespace w2l {

int ln_param_grad(const double* sums, int groups, float* dGammaBeta, hipStream_t stream);   // elementwise.hip

constexpr int kLiRows = 16;         // rows per workgroup: 32-byte runs of the transposed image (32 rows / 64-byte runs: the same
                                    // time on rows of 2160 floats, slower on 1024: fewer workgroups; profiles/r04_run23_*)
constexpr int kLiThreads = 1024;    // sixteen waves, one row each: every row of the tile is in flight at once
constexpr int kLiMaxV = 9;          // float4 per lane: rows of at most 64 * 4 * 9 = 2304 floats
constexpr size_t kLiMaxInner = 64 * 4 * kLiMaxV;

typedef float li_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 li_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t li_pack2(float a, float b) {
  const li_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, li_bf16x2));
}

struct LiP {
  // forward
  float* a; const float* x; float* r; float* y; float* meanRstd; const float* gammaBeta; float eps;
  uint32_t thr, seed, stream; float keepScale;
  // backward
  const float* rIn; const float* dy; const float* mr; double* sums; float* dr; const float* maskSrc; float* dmask; float maskScale;
  uint32_t ithr, iseed, istream; float ikeepScale;   // dropout applied to the IMAGES of dr only
  // both
  int groups; int inner;
  w2l_bf16_image_sink im;
  int abl;   // probe build, timing only: 1 no transposed image, 2 no row image, 4 no LDS tile, 8 no fp32 result stores; 16 plain block -> row-block mapping
};

// the rounded tile -> transposed image: thread = a pair of adjacent columns, 16 rows each -> two 32-byte runs
__device__ __forceinline__ void li_write_transposed(const uint32_t* tile, int pitchDw, int inner, int g0, const w2l_bf16_image_sink& im) {
  if (!im.transposed) return;
  for (int cp = threadIdx.x; cp < inner / 2; cp += kLiThreads) {
    uint32_t v[kLiRows];
#pragma unroll
    for (int rr = 0; rr < kLiRows; ++rr) v[rr] = tile[rr * pitchDw + cp];
    uint32_t lo[kLiRows / 2], hi[kLiRows / 2];
#pragma unroll
    for (int q = 0; q < kLiRows / 2; ++q) {
      lo[q] = (v[2 * q] & 0xffffu) | (v[2 * q + 1] << 16);
      hi[q] = (v[2 * q] >> 16) | (v[2 * q + 1] & 0xffff0000u);
    }
    uint4* d0 = (uint4*)(im.transposed + (size_t)(2 * cp) * im.ldTrans + g0);
    uint4* d1 = (uint4*)(im.transposed + (size_t)(2 * cp + 1) * im.ldTrans + g0);
#pragma unroll
    for (int q = 0; q < kLiRows / 8; ++q) {
      d0[q] = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
      d1[q] = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
    }
  }
}

// NJ = 16-byte chunks per lane (the host picks the smallest instance that holds the row)
template <bool BWD, int NJ>
__global__ __launch_bounds__(kLiThreads) void ln_rows_images_k(LiP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* tile = (uint32_t*)smem;                   // [16][inner + 8] bf16
  const int inner = p.inner, n4 = inner >> 2;
  const int pitchDw = (inner + 8) >> 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Row block of this workgroup.  Consecutive workgroups go to different XCDs (round robin), each with its own L2, and FOUR
  // consecutive row blocks share every 128-byte line of the transposed image (16 rows x 2 bytes each): with the plain mapping the
  // four quarters of a line are written from four L2s and reach memory as partial lines (the transposed image cost 50 of the
  // forward kernel's 175 us at 11968 x 2160: profiles/r04_run38_ln_images_ablations.log).  Here XCD x owns a contiguous range
  // of row blocks, so the quarters of a line meet in one L2.
  const int nrb = (p.groups + kLiRows - 1) / kLiRows;
  const int perXcd = ((nrb + 7) / 8 + 3) / 4 * 4;
  const int rbk = (p.abl & 16) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * perXcd + (int)(blockIdx.x >> 3);
  if (rbk >= nrb) return;
  const int g0 = rbk * kLiRows;
#pragma unroll 1
  for (int q = 0; q < kLiRows / 16; ++q) {
    const int rr = wave + 16 * q, g = g0 + rr;
    uint32_t* trow = tile + rr * pitchDw;
    if (g >= p.groups) {   // past the last row: zeros in the tile (the transposed runs cover 16 rows)
      for (int i = lane; i < n4; i += 64) *(uint2*)(trow + 2 * i) = make_uint2(0u, 0u);
    } else {
    const size_t base = (size_t)g * inner;
    if constexpr (!BWD) {
      // every load of the row before the first use, none behind a lane predicate (a chunk past the row re-reads the row's last
      // chunk and is ignored): as `if (i < n4) { load; load; use; store }` per chunk hipcc waited for each chunk's loads in
      // turn -- one pair of 16-byte loads in flight per lane, 3.5 TB/s at 11968 x 2160 (profiles/r04_run24_ln_images_ab.log)
      float4 v[NJ], avs[NJ], xvs[NJ];
      const int lastc = n4 - 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) avs[j] = *(const float4*)(p.a + base + 4 * (size_t)min(lane + 64 * j, lastc));
      if (p.x) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) xvs[j] = *(const float4*)(p.x + base + 4 * (size_t)min(lane + 64 * j, lastc));
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) xvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      double s = 0, ss = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4) {
          const size_t e = base + 4 * (size_t)i;
          float4 av = avs[j];
          float4 rv = xvs[j];
          if (p.thr) {
            av.x = keep_elem(e, p.seed, p.stream, p.thr) ? av.x * p.keepScale : 0.f;
            av.y = keep_elem(e + 1, p.seed, p.stream, p.thr) ? av.y * p.keepScale : 0.f;
            av.z = keep_elem(e + 2, p.seed, p.stream, p.thr) ? av.z * p.keepScale : 0.f;
            av.w = keep_elem(e + 3, p.seed, p.stream, p.thr) ? av.w * p.keepScale : 0.f;
            if (p.r != p.a || !p.x) *(float4*)(p.a + e) = av;
          }
          rv.x += av.x; rv.y += av.y; rv.z += av.z; rv.w += av.w;
          if (p.r != p.a || p.x) *(float4*)(p.r + e) = rv;
          v[j] = rv;
          s += (double)((rv.x + rv.y) + (rv.z + rv.w));
          ss += (double)((rv.x * rv.x + rv.y * rv.y) + (rv.z * rv.z + rv.w * rv.w));
        }
      }
      s = wave_sum_f64(s);
      ss = wave_sum_f64(ss);
      const double mu = s / (double)inner;
      double var = ss / (double)inner - mu * mu;
      if (var < 0) var = 0;
      const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
      const float muf = (float)mu;
      if (lane == 0) { p.meanRstd[2 * g] = muf; p.meanRstd[2 * g + 1] = rstd; }
      const float gam = p.gammaBeta[0] * rstd, bet = p.gammaBeta[1];
      uint16_t* irow = p.im.rowMajor ? p.im.rowMajor + (size_t)g * p.im.ldRows : nullptr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        if (i < n4) {
          float4 o = v[j];
          o.x = (o.x - muf) * gam + bet; o.y = (o.y - muf) * gam + bet;
          o.z = (o.z - muf) * gam + bet; o.w = (o.w - muf) * gam + bet;
          if (!(p.abl & 8)) *(float4*)(p.y + base + 4 * (size_t)i) = o;
          const uint2 b = make_uint2(li_pack2(o.x, o.y), li_pack2(o.z, o.w));
          if (irow && !(p.abl & 2)) *(uint2*)(irow + 4 * i) = b;
          if (!(p.abl & 4)) *(uint2*)(trow + 2 * i) = b;
        }
      }
    } else {
      const float mu = p.mr[2 * g], rstd = p.mr[2 * g + 1];
      float4 xh[NJ], dv[NJ], mvs[NJ];
      const int lastc = n4 - 1;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const size_t e = base + 4 * (size_t)min(lane + 64 * j, lastc);
        xh[j] = *(const float4*)(p.rIn + e);
        dv[j] = *(const float4*)(p.dy + e);
      }
      if (p.dmask) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) mvs[j] = *(const float4*)(p.maskSrc + base + 4 * (size_t)min(lane + 64 * j, lastc));
      } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) mvs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        const float4 rv = xh[j];
        xh[j] = make_float4((rv.x - mu) * rstd, (rv.y - mu) * rstd, (rv.z - mu) * rstd, (rv.w - mu) * rstd);
        if (i < n4) {
          s1 += (double)((dv[j].x + dv[j].y) + (dv[j].z + dv[j].w));
          s2 += (double)((dv[j].x * xh[j].x + dv[j].y * xh[j].y) + (dv[j].z * xh[j].z + dv[j].w * xh[j].w));
        }
      }
      s1 = wave_sum_f64(s1);
      s2 = wave_sum_f64(s2);
      if (lane == 0) { p.sums[2 * g] = s1; p.sums[2 * g + 1] = s2; }
      const float c1 = (float)(s1 / (double)inner), c2 = (float)(s2 / (double)inner);
      const float gr = p.gammaBeta[0] * rstd;
      uint16_t* irow = p.im.rowMajor ? p.im.rowMajor + (size_t)g * p.im.ldRows : nullptr;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int i = lane + 64 * j;
        if (i < n4) {
          const size_t e = base + 4 * (size_t)i;
          float4 o;
          o.x = gr * (dv[j].x - c1 - xh[j].x * c2);
          o.y = gr * (dv[j].y - c1 - xh[j].y * c2);
          o.z = gr * (dv[j].z - c1 - xh[j].z * c2);
          o.w = gr * (dv[j].w - c1 - xh[j].w * c2);
          if (!(p.abl & 8)) *(float4*)(p.dr + e) = o;
          if (p.dmask) {
            const float4 mv = mvs[j];
            float4 d2;
            d2.x = mv.x > 0.f ? o.x * p.maskScale : 0.f;
            d2.y = mv.y > 0.f ? o.y * p.maskScale : 0.f;
            d2.z = mv.z > 0.f ? o.z * p.maskScale : 0.f;
            d2.w = mv.w > 0.f ? o.w * p.maskScale : 0.f;
            *(float4*)(p.dmask + e) = d2;
          }
          if (p.ithr) {   // the images are those of dropout(dr): the masked gradient is only ever a GEMM operand
            o.x = keep_elem(e, p.iseed, p.istream, p.ithr) ? o.x * p.ikeepScale : 0.f;
            o.y = keep_elem(e + 1, p.iseed, p.istream, p.ithr) ? o.y * p.ikeepScale : 0.f;
            o.z = keep_elem(e + 2, p.iseed, p.istream, p.ithr) ? o.z * p.ikeepScale : 0.f;
            o.w = keep_elem(e + 3, p.iseed, p.istream, p.ithr) ? o.w * p.ikeepScale : 0.f;
          }
          const uint2 b = make_uint2(li_pack2(o.x, o.y), li_pack2(o.z, o.w));
          if (irow && !(p.abl & 2)) *(uint2*)(irow + 4 * i) = b;
          if (!(p.abl & 4)) *(uint2*)(trow + 2 * i) = b;
        }
      }
    }
    }
  }
  __syncthreads();
  if (!(p.abl & 1)) li_write_transposed(tile, pitchDw, inner, g0, p.im);
}

static bool li_sink_ok(const w2l_bf16_image_sink* s, int groups, size_t inner) {
  if (!s || (!s->rowMajor && !s->transposed)) return false;
  if (s->rowMajor && (s->ldRows < inner || (s->ldRows & 3) || (((uintptr_t)s->rowMajor) & 7))) return false;
  // the transposed runs cover 16 rows at a time: the pitch must hold the rows rounded up to 16
  if (s->transposed && (s->ldTrans < (size_t)((groups + kLiRows - 1) / kLiRows * kLiRows) || (s->ldTrans & 7) || (((uintptr_t)s->transposed) & 15)))
    return false;
  return true;
}

template <bool BWD>
static int li_launch(LiP p, hipStream_t s) {
  { const char* e = tune_env("W2L_LI_ABL"); p.abl = e ? atoi(e) : 0; }
  // 69 - 74 KB of LDS at the TDS widths: more than the 64 KB a kernel gets without the opt-in below.  gfx950 has 160 KB per CU;
  // where the opt-in fails (a part with a smaller LDS) the call returns W2L_EUNSUPPORTED and host/net.cpp's lnForward /
  // lnBackward run LayerNorm + conversion instead (round-4 advice)
  const size_t shmem = (size_t)kLiRows * (p.inner + 8) * 2;
  const int nrb = (p.groups + kLiRows - 1) / kLiRows;
  const int perXcd = ((nrb + 7) / 8 + 3) / 4 * 4;
  const int nj = (p.inner / 4 + 63) / 64;
#define W2L_LI_LAUNCH(NJ_)                                                                                                       \
  do {                                                                                                                           \
    static const bool attr = hipFuncSetAttribute((const void*)ln_rows_images_k<BWD, NJ_>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                                 (int)(kLiRows * (kLiMaxInner + 8) * 2)) == hipSuccess;                          \
    if (!attr) { (void)hipGetLastError(); return W2L_EUNSUPPORTED; }   /* less LDS than the tile needs: the caller runs the plain kernels */ \
    hipLaunchKernelGGL((ln_rows_images_k<BWD, NJ_>), dim3((unsigned)(8 * perXcd)), dim3(kLiThreads), shmem, s, p);               \
  } while (0)
  if (nj <= 4) W2L_LI_LAUNCH(4);            // 1024 floats (the Transformer recipe)
  else if (nj <= 5) W2L_LI_LAUNCH(5);       // 1200
  else if (nj <= 6) W2L_LI_LAUNCH(6);       // 1520
  else if (nj <= 8) W2L_LI_LAUNCH(8);       // 1840
  else W2L_LI_LAUNCH(kLiMaxV);              // 2160
#undef W2L_LI_LAUNCH
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

// w2l_residual_layernorm_forward + the bf16 images of y.  W2L_EUNSUPPORTED for rows the kernel does not hold (inner > 2304 or not
// a multiple of 4): run the plain call and w2l_bf16_convert then.
W2L_API int w2l_residual_layernorm_forward_images(int groups, size_t inner, float* a, const float* x, float* r, float* y,
                                                  const float* gammaBeta, float eps, double p, uint32_t seed, uint32_t rngStream,
                                                  float* meanRstd, const w2l_bf16_image_sink* yImages, w2l_stream_t stream) {
  if (groups <= 0 || inner == 0 || !a || !r || !y || !gammaBeta || !meanRstd) return W2L_EINVAL;
  if (p < 0.0 || p >= 1.0) return W2L_EINVAL;
  if ((inner & 3) || inner > kLiMaxInner) return W2L_EUNSUPPORTED;
  if (tune_env("W2L_LN_IMG_OFF")) return W2L_EUNSUPPORTED;   // probe build: A/B against LayerNorm + conversion pass
  if (!li_sink_ok(yImages, groups, inner)) return W2L_EINVAL;
  if ((((uintptr_t)a) | ((uintptr_t)x) | ((uintptr_t)r) | ((uintptr_t)y)) & 15) return W2L_EINVAL;
  LiP q{};
  q.a = a; q.x = x; q.r = r; q.y = y; q.meanRstd = meanRstd; q.gammaBeta = gammaBeta; q.eps = eps;
  q.thr = dropout_threshold(p); q.seed = seed; q.stream = rngStream; q.keepScale = (float)(1.0 / (1.0 - p));
  q.groups = groups; q.inner = (int)inner; q.im = *yImages;
  return li_launch<false>(q, (hipStream_t)stream);
}

// w2l_layernorm_backward + the bf16 images of dr -- of dropout(dr) when imageDropP > 0 (the hash of w2l_dropout_copy over the
// flat index with imageDropSeed / imageDropStream; dr itself stays unmasked)
W2L_API int w2l_layernorm_backward_images(int groups, size_t inner, const float* r, const float* dy, const float* gammaBeta,
                                          const float* meanRstd, float* dr, float* dGammaBeta, const float* maskSrc, float* dmask,
                                          float maskScale, double* sums, const w2l_bf16_image_sink* drImages, double imageDropP,
                                          uint32_t imageDropSeed, uint32_t imageDropStream, w2l_stream_t stream) {
  if (groups <= 0 || inner == 0 || !r || !dy || !gammaBeta || !meanRstd || !dr || !sums) return W2L_EINVAL;
  if (imageDropP < 0.0 || imageDropP >= 1.0 || (maskSrc && !dmask)) return W2L_EINVAL;
  if ((inner & 3) || inner > kLiMaxInner) return W2L_EUNSUPPORTED;
  if (tune_env("W2L_LN_IMG_OFF")) return W2L_EUNSUPPORTED;
  if (!li_sink_ok(drImages, groups, inner)) return W2L_EINVAL;
  if ((((uintptr_t)r) | ((uintptr_t)dy) | ((uintptr_t)dr) | ((uintptr_t)maskSrc) | ((uintptr_t)dmask)) & 15) return W2L_EINVAL;
  LiP q{};
  q.rIn = r; q.dy = dy; q.mr = meanRstd; q.sums = sums; q.dr = dr; q.gammaBeta = gammaBeta;
  q.maskSrc = maskSrc; q.dmask = maskSrc ? dmask : nullptr; q.maskScale = maskScale;
  q.ithr = dropout_threshold(imageDropP); q.iseed = imageDropSeed; q.istream = imageDropStream;
  q.ikeepScale = (float)(1.0 / (1.0 - imageDropP));
  q.groups = groups; q.inner = (int)inner; q.im = *drImages;
  const int st = li_launch<true>(q, (hipStream_t)stream);
  if (st != W2L_OK) return st;
  if (dGammaBeta) return ln_param_grad(sums, groups, dGammaBeta, (hipStream_t)stream);
  return W2L_OK;
}
