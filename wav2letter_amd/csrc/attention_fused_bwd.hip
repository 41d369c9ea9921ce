// attention_fused_bwd.hip -- the attention core of fl::Transformer's BACKWARD pass for the mixed-precision mode (BASELINE
// config 5, "bf16 MFMA attention"; block semantics recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151 -- the
// gradient of what attention_fused.hip computes):
//
//   dP[i][j]  = dctx_i . v_j                      dPm = dropout mask of the forward (the same stateless hash) applied to dP
//   dS[i][j]  = scale * P[i][j] * (dPm[i][j] - sum_k P[i][k] dPm[i][k])
//   dq_i      = sum_j dS[i][j] (k_j + E[j - i + n0])        dk_j = sum_i dS[i][j] q_i        dv_j = sum_i Pd[i][j] dctx_i
//   dE[w]     = sum_(b, h, i) dS[i][i + w - n0] q_i
//
// The unfused path (attention.hip, host/net.cpp) runs eleven launches for this (four batched GEMMs, dropout, softmax backward,
// two banded position products, a fill and a two-launch column sum) and moves the [B][H][T][T] score gradient five times
// through HBM in fp32.  Here:
//
//   attn_bwd_prep_k      E^T as a bf16 image [d][64 NT] (zero outside the table window), so that the position product reads MFMA
//                        A operands with two 8-byte loads
//   attn_fused_bwd_q_k   the QUERY side, one wave per 32 queries, the mirror image of attn_fused_fwd_k: dP^T = V dctx^T in the
//                        32x32 MFMA C layout (a lane holds one query, 16 keys per tile), so the softmax backward is a register
//                        reduction + one half-wave exchange; dS^T in that layout IS the B operand of dq^T = K^T dS^T; the skewed
//                        copy dR^T[w][i] = dS^T[w + i - off][i] goes through the wave's 8 KiB LDS scratch (the forward's skew, run
//                        backwards) and multiplies E^T.  Leaves three bf16 images for the key side: dS^T [key][query],
//                        Pd^T [key][query], dR^T [window row][query] -- the roundings the unfused bf16 GEMMs apply to these operands.
//   attn_fused_bwd_kv_k  the KEY side, everything that sums over QUERIES: one workgroup per (role, head, utterance) builds the
//                        transposed bf16 image of q (or dctx) in LDS; a wave owns 32 rows of an image (keys, or table-window rows),
//                        reads its B fragments -- 8 consecutive queries of its row, 16 bytes -- straight from global memory and
//                        multiplies: dk = dS^T q, dv = Pd^T dctx, dE partial = dR^T q (only the query blocks inside the band).
//   attn_bwd_de_reduce_k the [B H] partials of dE summed in a fixed order into the table gradient (zero outside the window)
//
// Operands are rounded to bf16 (nearest even) exactly where the unfused bf16 path rounds them (dctx, v, Pd, dS, k, q, E);
// accumulation and the softmax backward are fp32: the two paths agree to fp32 summation order.
#include "common.hpp"

namespace w2l {

typedef __bf16 ab_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ab_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ab_f32x2 __attribute__((ext_vector_type(2)));
typedef float ab_f32x4 __attribute__((ext_vector_type(4)));
typedef float ab_f32x16 __attribute__((ext_vector_type(16)));

struct AbP {
  const float *q, *k, *v;   // [B][T][ld], head h at columns h d .. (h + 1) d
  const float* E;           // position table rows [2 csz - 1][d] (internal layout), or null
  const float* P;           // [B][H][T][T] of the forward pass
  const float* dctx;        // [B][T][ldc]
  float *dq, *dk, *dv;      // [B][T][ld] (each may be null when its images are asked for)
  w2l_bf16_image_sink dqI, dkI, dvI;   // bf16 images of dq / dk / dv written in place of a conversion pass (null pointers: none)
  float* dE;                // [2 n0 + 1][d] or null
  uint16_t *dSt, *Pdt;      // bf16 [B H][TP][TP]: row = key, column = query
  uint16_t* dRt;            // bf16 [B H][GW][TP]: row = window row (global numbering), column = query
  uint16_t* Et;             // bf16 [d][GW]
  float* dEp;               // [B H][GW][d]
  int B, H, T, ld, ldc;
  int W, n0, rlo;
  float scale;
  uint32_t thr, seed, stream;
  float keepScale;
};

__device__ __forceinline__ uint32_t ab_pack2(float a, float b) {
  const ab_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ab_bf16x2));
}
__device__ __forceinline__ uint16_t ab_bf16(float a) { return (uint16_t)ab_pack2(a, 0.f); }
__device__ __forceinline__ ab_bf16x8 ab_pack8(ab_f32x4 a, ab_f32x4 b) {
  const uint4 u = make_uint4(ab_pack2(a[0], a[1]), ab_pack2(a[2], a[3]), ab_pack2(b[0], b[1]), ab_pack2(b[2], b[3]));
  return __builtin_bit_cast(ab_bf16x8, u);
}

// One 32 x 32 output tile in the MFMA C layout -- this lane holds result row `row`, columns col0 + (r & 3) + 8 (r >> 2) + 4 lh -- as
// fp32 and / or as bf16 images: the row-major image takes 8-byte stores, the transposed one 2-byte stores that are contiguous
// over the 32 lanes of a half-wave (64-byte runs)
__device__ __forceinline__ void ab_store_tile(const ab_f32x16& o, bool ok, size_t row, int col0, int lh, float* f, size_t ldf,
                                              const w2l_bf16_image_sink& im) {
  if (!ok) return;
  if (f) {
    float* dst = f + row * ldf + col0 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) *(ab_f32x4*)(dst + 8 * g) = ab_f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
  }
  if (im.rowMajor) {
    uint16_t* dst = im.rowMajor + row * im.ldRows + col0 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) *(uint2*)(dst + 8 * g) = make_uint2(ab_pack2(o[4 * g], o[4 * g + 1]), ab_pack2(o[4 * g + 2], o[4 * g + 3]));
  }
  if (im.transposed) {
    uint16_t* dst = im.transposed + (size_t)(col0 + 4 * lh) * im.ldTrans + row;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * im.ldTrans] = ab_bf16(o[r]);
  }
}

// Window rows are numbered globally per (utterance, head): wg = w - wOrg, wOrg = (n0 - rlo) - 31 - 32 (NT - 1), so that query block
// qb's table block e (rows -32 qb - 31 + n0 - rlo + 32 e ..) is global block g = e - qb + NT - 1 for every qb: 2 NT blocks of 32.
template <int NT>
__device__ __forceinline__ int ab_w_origin(int n0, int rlo) { return n0 - rlo - 31 - 32 * (NT - 1); }

template <int NT, int D>
__global__ __launch_bounds__(256) void attn_bwd_prep_k(AbP p) {
  constexpr int GW = 64 * NT;
  const int idx = blockIdx.x * 256 + threadIdx.x;   // (channel quad, window row): consecutive threads = consecutive rows of one quad
  if (idx >= GW * (D / 4)) return;
  const int wg = idx % GW, c4 = idx / GW;
  const int w = wg + ab_w_origin<NT>(p.n0, p.rlo);
  const bool ok = w >= 0 && w < p.W;
  ab_f32x4 e = *(const ab_f32x4*)(p.E + (size_t)(p.rlo + (ok ? w : 0)) * D + 4 * c4);
#pragma unroll
  for (int x = 0; x < 4; ++x) p.Et[(size_t)(4 * c4 + x) * GW + wg] = ok ? ab_bf16(e[x]) : (uint16_t)0;
}

// NT: tiles of 32 over the frames (T <= 32 NT); D: head width (multiple of 32)
template <int NT, int D>
__global__ __launch_bounds__(256, 1) void attn_fused_bwd_q_k(AbP p, int blocksPerWg) {
  constexpr int KP = D + 8;              // bf16 pitch of a row image: 16 bytes x odd -> conflict-free ds_read_b128
  constexpr int VP = 32 * NT + 4;        // bf16 pitch of a transposed image: (16 NT + 2) dwords -> conflict-free ds_read_b64
  constexpr int KS = D / 16;
  constexpr int TP = 32 * NT, GW = 64 * NT;
  constexpr int kImgBytes = (32 * NT * KP > D * VP ? 32 * NT * KP : D * VP) * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = (uint16_t*)smem;                       // V image [32 NT][KP], later K^T image [D][VP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  float* rs = (float*)(smem + kImgBytes) + wave * (32 * 65);   // this wave's skew scratch [32 queries][65]
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int T = p.T;
  const size_t rowBase = (size_t)b * T;
  const int hc = h * D;
  const size_t bh = (size_t)b * p.H + h;

  // ---- V image: rows j < T rounded to bf16, rows up to 32 NT zero (batched unconditional loads: attention_fused.hip)
  constexpr int kU = 8;
  for (int base = 0; base < 32 * NT * (D / 8); base += kU * nthr) {
    ab_f32x4 a[kU], c[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int j = idx / (D / 8), c8 = idx - j * (D / 8);
      const float* src = p.v + (rowBase + (j < T ? j : T - 1)) * p.ld + hc + 8 * (c8 < D / 8 ? c8 : 0);
      a[u] = *(const ab_f32x4*)src;
      c[u] = *(const ab_f32x4*)(src + 4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int j = idx / (D / 8), c8 = idx - j * (D / 8);
      if (idx < 32 * NT * (D / 8)) {
        uint4 v = make_uint4(ab_pack2(a[u][0], a[u][1]), ab_pack2(a[u][2], a[u][3]), ab_pack2(c[u][0], c[u][1]), ab_pack2(c[u][2], c[u][3]));
        if (j >= T) v = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(img + j * KP + 8 * c8) = v;
      }
    }
  }

  // ---- this wave's 32 queries: B-operand fragments of dctx
  const int qb = blockIdx.x * blocksPerWg + wave;        // query block; waves past the last block only help with the images
  const bool active = wave < blocksPerWg && 32 * qb < T;
  const int i0 = 32 * qb, iq = i0 + li;
  const int iqc = iq < T ? iq : T - 1;
  ab_bf16x8 df[KS];
  if (active) {
    const float* src = p.dctx + (rowBase + iqc) * p.ldc + hc + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) df[s] = ab_pack8(*(const ab_f32x4*)(src + 16 * s), *(const ab_f32x4*)(src + 16 * s + 4));
  }
  __syncthreads();

  ab_bf16x8 sf[NT][2];        // dS^T as B-operand fragments (k slots = the layout's permutation of the keys)
  ab_bf16x8 rf[NT + 1][2];    // dR^T blocks likewise (k slots = window rows)
  if (active) {
    ab_f32x16 acc[NT];
    // ---- dP^T tiles: rows = keys 32 t + ..., columns = queries
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const ab_bf16x8 va = *(const ab_bf16x8*)(img + (32 * t + li) * KP + 16 * s + 8 * lh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, df[s], acc[t], 0, 0, 0);
      }
    }
    // ---- P in the same layout, the forward's dropout mask, the softmax backward
    const size_t prow = ((bh * T) + iqc) * T;
    const bool vec = (T & 3) == 0;
    uint16_t* dStb = p.dSt + bh * TP * TP + i0 + li;
    uint16_t* Pdtb = p.Pdt + bh * TP * TP + i0 + li;
    ab_f32x16 pr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 32 * t + 8 * g + 4 * lh;
        const bool in = iq < T && j < T;
        const int jc = j < T ? j : 0;
        ab_f32x4 pv;
        if (vec) {
          pv = *(const ab_f32x4*)(p.P + prow + jc);
        } else {
#pragma unroll
          for (int x = 0; x < 4; ++x) pv[x] = p.P[prow + (jc + x < T ? jc + x : T - 1)];
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) pr[t][4 * g + x] = (in && j + x < T) ? pv[x] : 0.f;
      }
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float pd = pr[t][r], dp = acc[t][r];
        if (p.thr) {
          const bool keep = keep_elem(prow + j, p.seed, p.stream, p.thr);
          pd = keep ? pd * p.keepScale : 0.f;
          dp = keep ? dp * p.keepScale : 0.f;
        }
        Pdtb[(size_t)j * TP] = ab_bf16(pd);
        acc[t][r] = dp;
        dot += pr[t][r] * dp;
      }
    dot += __shfl_xor(dot, 32);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float ds = p.scale * pr[t][r] * (acc[t][r] - dot);
        acc[t][r] = ds;
        dStb[(size_t)j * TP] = ab_bf16(ds);
      }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const ab_f32x4 a = {acc[t][8 * u], acc[t][8 * u + 1], acc[t][8 * u + 2], acc[t][8 * u + 3]};
        const ab_f32x4 c = {acc[t][8 * u + 4], acc[t][8 * u + 5], acc[t][8 * u + 6], acc[t][8 * u + 7]};
        sf[t][u] = ab_pack8(a, c);
      }
    // ---- the skewed copy.  Tile t (keys 32 t ..) holds the window rows w' = jj - li + 31 of blocks e = t (w' < 32) and e = t + 1
    // (w' >= 32): scratch[query][w'] <- the tile, read back in the C layout of a window block (row = w', column = query); an entry
    // of block e comes from tile e where wr + li >= 31 and from tile e - 1 below that
    if (p.E) {
      uint16_t* dRtb = p.dRt + bh * GW * TP + i0 + li;
      ab_f32x16 prevUp;
#pragma unroll
      for (int r = 0; r < 16; ++r) prevUp[r] = 0.f;
#pragma unroll
      for (int e = 0; e <= NT; ++e) {
        ab_f32x16 blk, up;
        if (e < NT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * lh;
            rs[li * 65 + jj - li + 31] = acc[e][r];
          }
          __builtin_amdgcn_wave_barrier();   // same wave wrote and reads: its LDS operations complete in order
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int wr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float lo = rs[li * 65 + wr];
            up[r] = rs[li * 65 + 32 + wr];
            blk[r] = wr + li >= 31 ? lo : prevUp[r];
          }
          __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int wr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            blk[r] = wr + li >= 31 ? 0.f : prevUp[r];
            up[r] = 0.f;
          }
        }
        prevUp = up;
        const int g = e - qb + NT - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int wr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          dRtb[(size_t)(32 * g + wr) * TP] = ab_bf16(blk[r]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const ab_f32x4 a = {blk[8 * u], blk[8 * u + 1], blk[8 * u + 2], blk[8 * u + 3]};
          const ab_f32x4 c = {blk[8 * u + 4], blk[8 * u + 5], blk[8 * u + 6], blk[8 * u + 7]};
          rf[e][u] = ab_pack8(a, c);
        }
      }
    }
  }
  __syncthreads();   // every wave is done with the V image

  // ---- K^T image [channel][key] bf16 (thread = (key pair, 4 channels); attention_fused.hip's V^T staging)
  for (int base = 0; base < 16 * NT * (D / 4); base += kU * nthr) {
    constexpr int nCq = D / 4 / 8;                      // channel-quad groups of 8
    ab_f32x4 a[kU], c[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      int idx = base + u * nthr + tid;
      if (idx >= 16 * NT * (D / 4)) idx = 0;
      const int blk = idx >> 6, l6 = idx & 63;
      const int jg = blk / nCq, cg = blk - jg * nCq;
      const int j = 2 * (jg * 8 + (l6 & 7)), c4 = cg * 8 + (l6 >> 3);
      a[u] = *(const ab_f32x4*)(p.k + (rowBase + (j < T ? j : T - 1)) * p.ld + hc + 4 * c4);
      c[u] = *(const ab_f32x4*)(p.k + (rowBase + (j + 1 < T ? j + 1 : T - 1)) * p.ld + hc + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int blk = idx >> 6, l6 = idx & 63;
      const int jg = blk / nCq, cg = blk - jg * nCq;
      const int j = 2 * (jg * 8 + (l6 & 7)), c4 = cg * 8 + (l6 >> 3);
      if (idx < 16 * NT * (D / 4)) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          *(uint32_t*)(img + (4 * c4 + x) * VP + j) = ab_pack2(j < T ? a[u][x] : 0.f, j + 1 < T ? c[u][x] : 0.f);
      }
    }
  }
  __syncthreads();

  if (active) {
    // ---- dq^T = K^T dS^T + E^T dR^T
    const uint16_t* etb = p.Et + 32 * (NT - 1 - qb) + 4 * lh;   // + c GW + 32 e + 16 u (+ 8)
#pragma unroll 1
    for (int ct = 0; ct < D / 32; ++ct) {
      ab_f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      const uint16_t* krow = img + (32 * ct + li) * VP + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint2 lo = *(const uint2*)(krow + 32 * t + 16 * u), hi = *(const uint2*)(krow + 32 * t + 16 * u + 8);
          const ab_bf16x8 ka = __builtin_bit_cast(ab_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, sf[t][u], o, 0, 0, 0);
        }
      if (p.E) {
        const uint16_t* erow = etb + (size_t)(32 * ct + li) * GW;
        uint2 el[NT + 1][2], eh[NT + 1][2];
#pragma unroll
        for (int e = 0; e <= NT; ++e)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            el[e][u] = *(const uint2*)(erow + 32 * e + 16 * u);
            eh[e][u] = *(const uint2*)(erow + 32 * e + 16 * u + 8);
          }
#pragma unroll
        for (int e = 0; e <= NT; ++e)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const ab_bf16x8 ea = __builtin_bit_cast(ab_bf16x8, make_uint4(el[e][u].x, el[e][u].y, eh[e][u].x, eh[e][u].y));
            o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ea, rf[e][u], o, 0, 0, 0);
          }
      }
      ab_store_tile(o, iq < T, rowBase + iqc, hc + 32 * ct, lh, p.dq, (size_t)p.ld, p.dqI);
    }
  }
}

// The key side.  blockIdx.x = role: 0 dk = dS^T q, 1 dv = Pd^T dctx, 2 / 3 dE partial = dR^T q for window blocks 0 .. NT - 1 /
// NT .. 2 NT - 1.  NT waves; wave w owns image rows 32 w .. 32 w + 31 of its role.
template <int NT, int D>
__global__ __launch_bounds__(64 * NT, 1) void attn_fused_bwd_kv_k(AbP p) {
  constexpr int TP = 32 * NT, GW = 64 * NT;
  constexpr int VP = TP + 8;             // bf16 pitch of the transposed image: 16 bytes x odd -> conflict-free ds_read_b128
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = (uint16_t*)smem;       // X^T image [D][VP], X = q or dctx
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nthr = 64 * NT;
  const int li = lane & 31, lh = lane >> 5;
  const int role = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int T = p.T;
  const size_t rowBase = (size_t)b * T;
  const int hc = h * D;
  const size_t bh = (size_t)b * p.H + h;
  const float* X = role == 1 ? p.dctx : p.q;
  const int ldx = role == 1 ? p.ldc : p.ld;

  constexpr int kU = 8;
  for (int base = 0; base < 16 * NT * (D / 4); base += kU * nthr) {
    constexpr int nCq = D / 4 / 8;
    ab_f32x4 a[kU], c[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      int idx = base + u * nthr + tid;
      if (idx >= 16 * NT * (D / 4)) idx = 0;
      const int blk = idx >> 6, l6 = idx & 63;
      const int jg = blk / nCq, cg = blk - jg * nCq;
      const int j = 2 * (jg * 8 + (l6 & 7)), c4 = cg * 8 + (l6 >> 3);
      a[u] = *(const ab_f32x4*)(X + (rowBase + (j < T ? j : T - 1)) * ldx + hc + 4 * c4);
      c[u] = *(const ab_f32x4*)(X + (rowBase + (j + 1 < T ? j + 1 : T - 1)) * ldx + hc + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int blk = idx >> 6, l6 = idx & 63;
      const int jg = blk / nCq, cg = blk - jg * nCq;
      const int j = 2 * (jg * 8 + (l6 & 7)), c4 = cg * 8 + (l6 >> 3);
      if (idx < 16 * NT * (D / 4)) {
#pragma unroll
        for (int x = 0; x < 4; ++x)
          *(uint32_t*)(img + (4 * c4 + x) * VP + j) = ab_pack2(j < T ? a[u][x] : 0.f, j + 1 < T ? c[u][x] : 0.f);
      }
    }
  }

  // ---- this wave's 32 rows: B fragments = 8 consecutive queries of the row per k-step, straight from the global image
  const int g = role >= 2 ? (role - 2) * NT + wave : wave;   // row block
  const uint16_t* rowp = role == 0   ? p.dSt + bh * TP * TP + (size_t)(32 * g + li) * TP
                         : role == 1 ? p.Pdt + bh * TP * TP + (size_t)(32 * g + li) * TP
                                     : p.dRt + bh * GW * TP + (size_t)(32 * g + li) * TP;
  ab_bf16x8 bf[2 * NT];
#pragma unroll
  for (int s = 0; s < 2 * NT; ++s) {
    // window block g holds query block qb's table block e = g + qb - (NT - 1): written (and non-zero) only for 0 <= e <= NT
    const int e = g + (s >> 1) - (NT - 1);
    const bool valid = 32 * (s >> 1) < T && (role < 2 || (e >= 0 && e <= NT));   // (columns of query blocks past T are never written)
    uint4 ld = *(const uint4*)(rowp + 16 * s + 8 * lh);   // unconditional (inside the workspace either way), zero selected afterwards
    if (!valid) ld = make_uint4(0u, 0u, 0u, 0u);
    bf[s] = __builtin_bit_cast(ab_bf16x8, ld);
  }
  __syncthreads();

#pragma unroll 1
  for (int ct = 0; ct < D / 32; ++ct) {
    ab_f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const uint16_t* xrow = img + (32 * ct + li) * VP + 8 * lh;
#pragma unroll
    for (int s = 0; s < 2 * NT; ++s) {
      const ab_bf16x8 xa = *(const ab_bf16x8*)(xrow + 16 * s);
      o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, bf[s], o, 0, 0, 0);
    }
    const int row = 32 * g + li;
    const w2l_bf16_image_sink none{nullptr, 0, nullptr, 0};
    if (role == 0) ab_store_tile(o, row < T, rowBase + (row < T ? row : 0), hc + 32 * ct, lh, p.dk, (size_t)p.ld, p.dkI);
    else if (role == 1) ab_store_tile(o, row < T, rowBase + (row < T ? row : 0), hc + 32 * ct, lh, p.dv, (size_t)p.ld, p.dvI);
    else ab_store_tile(o, true, bh * GW + row, 32 * ct, lh, p.dEp, (size_t)D, none);
  }
}

// dE[row][c] = sum over (b, h) of the partials in a fixed order; rows outside the window zero
template <int NT, int D>
__global__ __launch_bounds__(256) void attn_bwd_de_reduce_k(AbP p) {
  constexpr int GW = 64 * NT;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int rows = 2 * p.n0 + 1;
  if (idx >= rows * (D / 4)) return;
  const int row = idx / (D / 4), c4 = idx - row * (D / 4);
  const int w = row - p.rlo;
  ab_f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int wg = w - ab_w_origin<NT>(p.n0, p.rlo);
  if (w >= 0 && w < p.W && wg >= 0 && wg < GW) {   // (a window wider than the rows T frames reach: no (i, j) pair lands on the rest)
    const float* src = p.dEp + (size_t)wg * D + 4 * c4;
    const int n = p.B * p.H;
    int i = 0;
    for (; i + 8 <= n; i += 8) {   // eight loads in flight; the order of the additions is fixed
      ab_f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const ab_f32x4*)(src + (size_t)(i + u) * GW * D);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; s[3] += v[u][3]; }
    }
    for (; i < n; ++i) {
      const ab_f32x4 v = *(const ab_f32x4*)(src + (size_t)i * GW * D);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
  }
  *(ab_f32x4*)(p.dE + (size_t)row * D + 4 * c4) = s;
}

template <int NT, int D>
struct AbGeom {
  static constexpr size_t TP = 32 * NT, GW = 64 * NT;
  static size_t align(size_t x) { return (x + 255) / 256 * 256; }
  static size_t imgBytes(size_t BH) { return align(BH * TP * TP * 2); }
  static size_t dRtBytes(size_t BH) { return align(BH * GW * TP * 2); }
  static size_t etBytes() { return align((size_t)D * GW * 2); }
  static size_t dEpBytes(size_t BH) { return align(BH * GW * D * 4); }
  static size_t total(size_t BH, bool table) {
    return 2 * imgBytes(BH) + (table ? dRtBytes(BH) + etBytes() + dEpBytes(BH) : 0);
  }
};

template <int NT, int D>
static int ab_launch(AbP p, void* ws, size_t wsBytes, hipStream_t s) {
  typedef AbGeom<NT, D> G;
  const size_t BH = (size_t)p.B * p.H;
  const bool table = p.E != nullptr;
  if (wsBytes < G::total(BH, table)) return W2L_EINVAL;
  unsigned char* w = (unsigned char*)ws;
  p.dSt = (uint16_t*)w; w += G::imgBytes(BH);
  p.Pdt = (uint16_t*)w; w += G::imgBytes(BH);
  if (table) {
    p.dRt = (uint16_t*)w; w += G::dRtBytes(BH);
    p.Et = (uint16_t*)w; w += G::etBytes();
    p.dEp = (float*)w;
  } else {
    p.dRt = p.Et = nullptr; p.dEp = nullptr;
  }
  constexpr int KP = D + 8, VP = 32 * NT + 4;
  constexpr int imgBytes = (32 * NT * KP > D * VP ? 32 * NT * KP : D * VP) * 2;
  const int blocks = (p.T + 31) / 32;
  int bpw = 4;   // as the forward kernel: as few query blocks per workgroup as keeps >= 128 workgroups in flight
  while (bpw > 1 && (long long)((blocks + bpw - 2) / (bpw - 1)) * p.H * p.B <= 256) --bpw;
  const char* be = tune_env("W2L_AB_BPW");
  if (be) bpw = atoi(be);
  if (bpw > blocks) bpw = blocks;
  if (bpw < 1 || bpw > 4) bpw = 1;
  const size_t shmemQ = (size_t)imgBytes + 4 * 32 * 65 * sizeof(float);
  const size_t shmemK = (size_t)D * (32 * NT + 8) * 2;
  static const bool attr =
      hipFuncSetAttribute((const void*)attn_fused_bwd_q_k<NT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmemQ) == hipSuccess &&
      hipFuncSetAttribute((const void*)attn_fused_bwd_kv_k<NT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmemK) == hipSuccess;
  if (!attr) return W2L_EHIP;
  if (table) {
    hipLaunchKernelGGL((attn_bwd_prep_k<NT, D>), dim3((unsigned)((64 * NT * (D / 4) + 255) / 256)), dim3(256), 0, s, p);
    W2L_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((attn_fused_bwd_q_k<NT, D>), dim3((unsigned)((blocks + bpw - 1) / bpw), (unsigned)p.H, (unsigned)p.B), dim3(256), shmemQ, s,
                     p, bpw);
  W2L_LAUNCH_CHECK();
  hipLaunchKernelGGL((attn_fused_bwd_kv_k<NT, D>), dim3(table ? 4u : 2u, (unsigned)p.H, (unsigned)p.B), dim3(64 * NT), shmemK, s, p);
  W2L_LAUNCH_CHECK();
  if (table) {
    const int rows = 2 * p.n0 + 1;
    hipLaunchKernelGGL((attn_bwd_de_reduce_k<NT, D>), dim3((unsigned)((rows * (D / 4) + 255) / 256)), dim3(256), 0, s, p);
    W2L_LAUNCH_CHECK();
  }
  return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

static bool ab_supported(const w2l_attn_fused_desc* d, int& nt) {
  if (!d || d->B <= 0 || d->H <= 0 || d->T <= 0 || d->d <= 0 || d->B > 65535 || d->H > 65535) return false;
  nt = (d->T + 31) / 32;
  return (d->d == 32 || d->d == 256) && nt <= 6;
}

W2L_API size_t w2l_attn_fused_backward_workspace(const w2l_attn_fused_desc* d, int withPosTable) {
  int nt;
  if (!ab_supported(d, nt)) return 0;
  const size_t BH = (size_t)d->B * d->H;
  const bool t = withPosTable != 0;
#define W2L_AB(NTv, Dv) if (nt <= NTv && d->d == Dv) return AbGeom<NTv, Dv>::total(BH, t);
  W2L_AB(2, 32) W2L_AB(4, 32) W2L_AB(6, 32)
  W2L_AB(2, 256) W2L_AB(4, 256) W2L_AB(6, 256)
#undef W2L_AB
  return 0;
}

// returns W2L_EUNSUPPORTED for a geometry without a fused kernel (the caller runs the unfused sequence then)
static bool ab_sink_ok(const w2l_bf16_image_sink* s, const float* f) {
  if (!s || (!s->rowMajor && !s->transposed)) return f != nullptr;     // no images: the fp32 result is required
  if (s->rowMajor && ((s->ldRows & 3) || (((uintptr_t)s->rowMajor) & 7))) return false;
  return true;
}

W2L_API int w2l_attn_fused_backward_images(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v,
                                           const float* posTable, const float* P, const float* dctx, float* dq, float* dk, float* dv,
                                           const w2l_bf16_image_sink* dqImages, const w2l_bf16_image_sink* dkImages,
                                           const w2l_bf16_image_sink* dvImages, float* dPosTable, void* workspace,
                                           size_t workspaceBytes, w2l_stream_t stream) {
  if (!d || !q || !k || !v || !P || !dctx || !workspace) return W2L_EINVAL;
  if (!ab_sink_ok(dqImages, dq) || !ab_sink_ok(dkImages, dk) || !ab_sink_ok(dvImages, dv)) return W2L_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->T <= 0 || d->d <= 0 || d->B > 65535 || d->H > 65535) return W2L_EINVAL;
  if (d->dropP < 0.0 || d->dropP >= 1.0) return W2L_EINVAL;
  if (posTable && (d->W <= 0 || d->rlo < 0 || d->n0 < 0 || !dPosTable)) return W2L_EINVAL;
  int nt;
  if (!ab_supported(d, nt)) return W2L_EUNSUPPORTED;
  if (tune_env("W2L_AB_OFF")) return W2L_EUNSUPPORTED;   // probe build: A/B against the unfused sequence
  if ((d->ld & 3) || (d->ldc & 3) ||
      ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dctx) | ((uintptr_t)posTable) | ((uintptr_t)P) | ((uintptr_t)dq) |
        ((uintptr_t)dk) | ((uintptr_t)dv) | ((uintptr_t)dPosTable)) & 15) || (((uintptr_t)workspace) & 255))
    return W2L_EUNSUPPORTED;
  AbP p{};
  p.q = q; p.k = k; p.v = v; p.E = posTable; p.P = P; p.dctx = dctx; p.dq = dq; p.dk = dk; p.dv = dv; p.dE = dPosTable;
  const w2l_bf16_image_sink none{nullptr, 0, nullptr, 0};
  p.dqI = dqImages ? *dqImages : none; p.dkI = dkImages ? *dkImages : none; p.dvI = dvImages ? *dvImages : none;
  p.B = d->B; p.H = d->H; p.T = d->T; p.ld = d->ld; p.ldc = d->ldc; p.W = d->W; p.n0 = d->n0; p.rlo = d->rlo; p.scale = d->scale;
  p.thr = dropout_threshold(d->dropP);
  p.seed = d->dropSeed; p.stream = d->dropStream;
  p.keepScale = (float)(1.0 / (1.0 - d->dropP));
  hipStream_t s = (hipStream_t)stream;
#define W2L_AB(NTv, Dv) if (nt <= NTv && d->d == Dv) return ab_launch<NTv, Dv>(p, workspace, workspaceBytes, s);
  W2L_AB(2, 32) W2L_AB(4, 32) W2L_AB(6, 32)
  W2L_AB(2, 256) W2L_AB(4, 256) W2L_AB(6, 256)
#undef W2L_AB
  return W2L_EUNSUPPORTED;
}

W2L_API int w2l_attn_fused_backward(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                                    const float* P, const float* dctx, float* dq, float* dk, float* dv, float* dPosTable,
                                    void* workspace, size_t workspaceBytes, w2l_stream_t stream) {
  if (!dq || !dk || !dv) return W2L_EINVAL;
  return w2l_attn_fused_backward_images(d, q, k, v, posTable, P, dctx, dq, dk, dv, nullptr, nullptr, nullptr, dPosTable, workspace,
                                        workspaceBytes, stream);
}
