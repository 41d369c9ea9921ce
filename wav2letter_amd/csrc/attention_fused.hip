// attention_fused.hip -- the attention core of fl::Transformer's forward pass in ONE launch for the mixed-precision mode
// (BASELINE config 5, "bf16 MFMA attention"; block semantics recipes/joint_training_vox_populi/cpc/TransformerCPC.cpp:117-151):
//
//   S[i][j]  = (q_i . k_j + q_i . E[rlo + j - i + n0 - rlo]) * scale     the relative-position term exists where the row is in the table
//   P        = softmax_j(S + log padMask)                                  keys j >= keyLen[b] get probability 0
//   Pd       = dropout(P)                                                  the library's stateless hash over the [B][H][T][T] index
//   ctx_i    = sum_j Pd[i][j] v_j
//
// The unfused path (attention.hip) runs five launches per block for this and moves S twice, the R = Q E^T matrix (18 MB a
// block at the recipe's 1024 / 4 heads / 188 frames / 919-row table) twice and P three times through HBM.  Here one
// workgroup owns (utterance, head, a group of 32-query blocks), one WAVE owns one 32-query block, and scores never leave
// the registers: only P (and Pd when dropout is on) are written -- the backward pass reads them -- plus ctx.
//
// Layout trick (the CDNA attention idiom): the wave computes S^T = K Q^T, not Q K^T.  In the 32x32 MFMA C layout a lane
// then holds ONE query column (lane & 31) and 16 key rows per tile, (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
//   * the softmax over keys is a register reduction plus one exchange between the two half-waves;
//   * P^T in that layout IS an MFMA B operand (column = query, k = key in registers) for ctx^T = V^T P^T, with the k slots of
//     k-step (t, u) standing for the keys 32 t + 16 u + 8 (e >> 2) + 4 (lane >> 5) + (e & 3): a permutation of the summation
//     index, harmless as long as the A operand (V^T, read from a transposed bf16 image in LDS) uses the same one.
// Relative positions: R^T[w][i] = E[w] . q_i is computed for 32-row blocks of the table window; the (key, query) tile t needs
// the entries w = j - i + const, a skew no register layout can supply, so two consecutive blocks go through an 8 KiB LDS
// scratch of the wave ([query][64 w], pitch 65) and come back gathered.
//
// Operands are rounded to bf16 (nearest even) exactly where the unfused bf16 path rounds them (q, k, E, P, v), accumulation,
// scores and softmax are fp32: the two paths agree to fp32 summation order.
#include "common.hpp"

namespace w2l {

typedef __bf16 af_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 af_bf16x2 __attribute__((ext_vector_type(2)));
typedef float af_f32x2 __attribute__((ext_vector_type(2)));
typedef float af_f32x4 __attribute__((ext_vector_type(4)));
typedef float af_f32x16 __attribute__((ext_vector_type(16)));

struct AfP {
  const float *q, *k, *v;   // [B][T][ld], head h at columns h d .. (h + 1) d
  const float* E;           // position table rows [2 csz - 1][d] (internal layout), or null
  const int* keyLen;
  float *P, *Pd, *ctx;      // P, Pd [B][H][T][T]; ctx [B][T][ldc] (may be null when its images are asked for)
  w2l_bf16_image_sink ctxI; // bf16 images of ctx written in place of a conversion pass (null pointers: none)
  int B, H, T, ld, ldc;
  int W, n0, rlo;           // table rows rlo .. rlo + W are the ones an utterance of T frames reaches
  float scale;
  uint32_t thr, seed, stream;
  float keepScale;
};

__device__ __forceinline__ uint32_t af_pack2(float a, float b) {
  const af_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, af_bf16x2));
}
__device__ __forceinline__ af_bf16x8 af_pack8(af_f32x4 a, af_f32x4 b) {
  const uint4 u = make_uint4(af_pack2(a[0], a[1]), af_pack2(a[2], a[3]), af_pack2(b[0], b[1]), af_pack2(b[2], b[3]));
  return __builtin_bit_cast(af_bf16x8, u);
}

// NT: key tiles of 32 (T <= 32 NT); D: head width (multiple of 16)
template <int NT, int D>
__global__ __launch_bounds__(256, 1) void attn_fused_fwd_k(AfP p, int blocksPerWg, int abl) {
  constexpr int KP = D + 8;              // bf16 pitch of a K-image row: 16 bytes x odd -> conflict-free ds_read_b128
  constexpr int VP = 32 * NT + 4;        // bf16 pitch of a V^T-image row: (16 NT + 2) dwords, = 2 mod 4 -> conflict-free ds_read_b64
  constexpr int KS = D / 16;             // k-steps over the head width
  constexpr int kImgBytes = (32 * NT * KP > D * VP ? 32 * NT * KP : D * VP) * 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint16_t* img = (uint16_t*)smem;                       // K image [32 NT][KP], later V^T image [D][VP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  float* rs = (float*)(smem + kImgBytes) + wave * (32 * 65);   // this wave's skew scratch [32 queries][65]
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int T = p.T;
  const size_t rowBase = (size_t)b * T;
  const int hc = h * D;

  // ---- K image: rows j < T rounded to bf16, rows up to 32 NT zero.  Batches of kU items per thread: all loads of a batch
  // are issued (unconditionally, from a clamped row; zero selected afterwards) before the first conversion -- a load inside
  // the bounds branch costs a full memory round trip per item (the first build: 82 us a launch, 90 % of it here)
  constexpr int kU = 8;
  for (int base = (abl & 2) ? (1 << 30) : 0; base < 32 * NT * (D / 8); base += kU * nthr) {
    af_f32x4 a[kU], c[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int j = idx / (D / 8), c8 = idx - j * (D / 8);
      const float* src = p.k + (rowBase + (j < T ? j : T - 1)) * p.ld + hc + 8 * c8;
      a[u] = *(const af_f32x4*)src;
      c[u] = *(const af_f32x4*)(src + 4);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int j = idx / (D / 8), c8 = idx - j * (D / 8);
      if (idx < 32 * NT * (D / 8)) {
        uint4 v = make_uint4(af_pack2(a[u][0], a[u][1]), af_pack2(a[u][2], a[u][3]), af_pack2(c[u][0], c[u][1]), af_pack2(c[u][2], c[u][3]));
        if (j >= T) v = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(img + j * KP + 8 * c8) = v;
      }
    }
  }

  // ---- this wave's 32 queries: B-operand fragments of q, kept in registers for all three products that read them
  const int qb = blockIdx.x * blocksPerWg + wave;        // query block; waves past the last block only help with the images
  const bool active = wave < blocksPerWg && 32 * qb < T;
  const int i0 = 32 * qb, iq = i0 + li;                  // this lane's query (as B-operand column and as C-layout column)
  const int iqc = iq < T ? iq : T - 1;
  af_bf16x8 qf[KS];
  if (active) {
    const float* src = p.q + (rowBase + iqc) * p.ld + hc + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = af_pack8(*(const af_f32x4*)(src + 16 * s), *(const af_f32x4*)(src + 16 * s + 4));
  }
  __syncthreads();

  af_f32x16 acc[NT];
  if (active) {
    // ---- S^T tiles: rows = keys 32 t + ..., columns = queries
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const af_bf16x8 ka = *(const af_bf16x8*)(img + (32 * t + li) * KP + 16 * s + 8 * lh);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], acc[t], 0, 0, 0);
      }
    }
    // ---- relative positions.  Tile t (keys 32 t ..) and this query block need w = j - i + n0 - rlo for 32 x 32 (j, i) pairs:
    // the 63 values wb(t) .. wb(t) + 62, wb(t) = 32 t - i0 - 31 + n0 - rlo.  Table block e = rows wb(0) + 32 e .. + 32 is shared by
    // tiles e - 1 (upper half of its window) and e (lower half): NT + 1 blocks, each multiplied once.
    if (p.E && !(abl & 1)) {
      const int wb0 = -i0 - 31 + p.n0 - p.rlo;
      af_f32x16 prev;
#pragma unroll
      for (int e = 0; e <= NT; ++e) {
        af_f32x16 cur;
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] = 0.f;
        const int w = wb0 + 32 * e + li;                  // the table-window row this lane supplies as A-operand row
        const bool ok = w >= 0 && w < p.W;
        const float* src = p.E + (size_t)(p.rlo + (ok ? w : 0)) * D + 8 * lh;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          // unconditional loads from a clamped row, zero selected afterwards (a load inside a branch serialises)
          af_f32x4 a = *(const af_f32x4*)(src + 16 * s), c = *(const af_f32x4*)(src + 16 * s + 4);
          if (!ok) { a = af_f32x4{0.f, 0.f, 0.f, 0.f}; c = a; }
          cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_pack8(a, c), qf[s], cur, 0, 0, 0);
        }
        if (e > 0) {
          // scratch[query][0 .. 64) <- window of tile t = e - 1: prev = w - wb(t) in 0 .. 32, cur = 32 .. 64
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int wr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            rs[li * 65 + wr] = prev[r];
            rs[li * 65 + 32 + wr] = cur[r];
          }
          __builtin_amdgcn_wave_barrier();   // same wave wrote and reads: its LDS operations complete in order
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jj = (r & 3) + 8 * (r >> 2) + 4 * lh;   // key within the tile; w - wb = jj - li + 31
            acc[e - 1][r] += rs[li * 65 + jj - li + 31];
          }
          __builtin_amdgcn_wave_barrier();
        }
        prev = cur;
      }
    }
    // ---- softmax over the keys of this lane's query: registers, then the other half-wave
    const int kl = p.keyLen ? min(p.keyLen[b], T) : T;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = j < kl ? acc[t][r] * p.scale : -INFINITY;
        acc[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float ev = j < kl ? expf(acc[t][r] - mx) : 0.f;
        acc[t][r] = ev;
        sum += ev;
      }
    sum += __shfl_xor(sum, 32);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;   // an utterance with no valid key: zeros (as the unfused kernel)
    // ---- P (and Pd) to memory for the backward pass; the registers keep what P V multiplies
    const size_t prow = (((size_t)b * p.H + h) * T + iqc) * T;
    const bool vec = (T & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 32 * t + 8 * g + 4 * lh;
        af_f32x4 pv, dv;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const float pr = acc[t][4 * g + x] * inv;
          pv[x] = pr;
          float dr = pr;
          if (p.thr) dr = keep_elem(prow + j + x, p.seed, p.stream, p.thr) ? pr * p.keepScale : 0.f;
          dv[x] = dr;
          acc[t][4 * g + x] = dr;
        }
        if (iq < T && j < T && !(abl & 8)) {
          if (vec) {
            *(af_f32x4*)(p.P + prow + j) = pv;
            if (p.thr) *(af_f32x4*)(p.Pd + prow + j) = dv;
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
              if (j + x < T) {
                p.P[prow + j + x] = pv[x];
                if (p.thr) p.Pd[prow + j + x] = dv[x];
              }
          }
        }
      }
  }
  __syncthreads();   // every wave is done with the K image

  // ---- V^T image [channel][key] bf16: thread = (key pair, 4 channels); 8 key pairs x 8 channel quads per wave-instruction:
  // 128-byte runs on the memory side, 64 distinct banks on the LDS side
  for (int base = (abl & 4) ? (1 << 30) : 0; base < 16 * NT * (D / 4); base += kU * nthr) {
    constexpr int nCq = D / 4 / 8;                      // channel-quad groups of 8
    af_f32x4 a[kU], c[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int idx = base + u * nthr + tid;
      const int blk = idx >> 6, l6 = idx & 63;
      const int jg = blk / nCq, cg = blk - jg * nCq;
      const int j = 2 * (jg * 8 + (l6 & 7)), c4 = cg * 8 + (l6 >> 3);
      a[u] = *(const af_f32x4*)(p.v + (rowBase + (j < T ? j : T - 1)) * p.ld + hc + 4 * c4);
      c[u] = *(const af_f32x4*)(p.v + (rowBase + (j + 1 < T ? j + 1 : T - 1)) * p.ld + hc + 4 * c4);
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
          *(uint32_t*)(img + (4 * c4 + x) * VP + j) = af_pack2(j < T ? a[u][x] : 0.f, j + 1 < T ? c[u][x] : 0.f);
      }
    }
  }
  __syncthreads();

  if (active && !(abl & 16)) {
    // ---- ctx^T = V^T Pd^T: B fragments from the probability registers (k slot e of half lh <-> acc[t][8 u + e])
    af_bf16x8 pf[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const af_f32x4 a = {acc[t][8 * u], acc[t][8 * u + 1], acc[t][8 * u + 2], acc[t][8 * u + 3]};
        const af_f32x4 c = {acc[t][8 * u + 4], acc[t][8 * u + 5], acc[t][8 * u + 6], acc[t][8 * u + 7]};
        pf[t][u] = af_pack8(a, c);
      }
#pragma unroll
    for (int ct = 0; ct < D / 32; ++ct) {
      af_f32x16 o;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = 0.f;
      const uint16_t* vrow = img + (32 * ct + li) * VP + 4 * lh;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          // keys 32 t + 16 u + 4 lh + {0..3} and + 8 + {0..3}: two 8-byte reads
          const uint2 lo = *(const uint2*)(vrow + 32 * t + 16 * u), hi = *(const uint2*)(vrow + 32 * t + 16 * u + 8);
          const af_bf16x8 va = __builtin_bit_cast(af_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[t][u], o, 0, 0, 0);
        }
      if (iq < T) {
        if (p.ctx) {
          float* dst = p.ctx + (rowBase + iq) * p.ldc + hc + 32 * ct + 4 * lh;
#pragma unroll
          for (int g = 0; g < 4; ++g) *(af_f32x4*)(dst + 8 * g) = af_f32x4{o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]};
        }
        // the bf16 images the next product reads (attention_fused_bwd.hip::ab_store_tile): 8-byte stores into the row-major image,
        // 2-byte stores contiguous over the lanes into the transposed one
        if (p.ctxI.rowMajor) {
          uint16_t* dst = p.ctxI.rowMajor + (rowBase + iq) * p.ctxI.ldRows + hc + 32 * ct + 4 * lh;
#pragma unroll
          for (int g = 0; g < 4; ++g) *(uint2*)(dst + 8 * g) = make_uint2(af_pack2(o[4 * g], o[4 * g + 1]), af_pack2(o[4 * g + 2], o[4 * g + 3]));
        }
        if (p.ctxI.transposed) {
          uint16_t* dst = p.ctxI.transposed + (size_t)(hc + 32 * ct + 4 * lh) * p.ctxI.ldTrans + rowBase + iq;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * p.ctxI.ldTrans] = (uint16_t)af_pack2(o[r], 0.f);
        }
      }
    }
  }
}

template <int NT, int D>
static int af_launch(const AfP& p, hipStream_t s) {
  constexpr int KP = D + 8, VP = 32 * NT + 4;
  constexpr int imgBytes = (32 * NT * KP > D * VP ? 32 * NT * KP : D * VP) * 2;
  const int blocks = (p.T + 31) / 32;
  // timing-only ablations (probe build): 1 no position term, 2 no K staging, 4 no V staging, 8 no P / Pd stores, 16 no P V
  const char* ae = tune_env("W2L_AF_ABL");
  const int abl = ae ? atoi(ae) : 0;
  const char* be = tune_env("W2L_AF_BPW");
  // waves per workgroup: as few as keeps >= 128 workgroups in flight, at most 4 (the LDS scratch is sized for 4)
  int bpw = 4;
  while (bpw > 1 && (long long)((blocks + bpw - 2) / (bpw - 1)) * p.H * p.B <= 256) --bpw;   // (config 5: 2 blocks a workgroup, 192 workgroups)
  if (be) bpw = atoi(be);
  if (bpw > blocks) bpw = blocks;
  if (bpw < 1 || bpw > 4) bpw = 1;
  const int waves = 4;                       // the waves past the workgroup's query blocks only help staging the images
  const size_t shmem = (size_t)imgBytes + 4 * 32 * 65 * sizeof(float);
  static const bool attr =
      hipFuncSetAttribute((const void*)attn_fused_fwd_k<NT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) == hipSuccess;
  if (!attr) return W2L_EHIP;
  hipLaunchKernelGGL((attn_fused_fwd_k<NT, D>), dim3((unsigned)((blocks + bpw - 1) / bpw), (unsigned)p.H, (unsigned)p.B),
                     dim3(64 * waves), shmem, s, p, bpw, abl);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

}  // namespace w2l

using namespace w2l;

// returns W2L_EUNSUPPORTED for a geometry without a fused kernel (the caller runs the unfused sequence then)
W2L_API int w2l_attn_fused_forward_images(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v,
                                          const float* posTable, const int* keyLen, float* P, float* Pd, float* ctx,
                                          const w2l_bf16_image_sink* ctxImages, w2l_stream_t stream) {
  if (!d || !q || !k || !v || !P) return W2L_EINVAL;
  const bool images = ctxImages && (ctxImages->rowMajor || ctxImages->transposed);
  if (!ctx && !images) return W2L_EINVAL;
  if (images && ctxImages->rowMajor && ((ctxImages->ldRows & 3) || (((uintptr_t)ctxImages->rowMajor) & 7))) return W2L_EINVAL;
  if (d->B <= 0 || d->H <= 0 || d->T <= 0 || d->d <= 0 || d->B > 65535 || d->H > 65535) return W2L_EINVAL;
  if (d->dropP < 0.0 || d->dropP >= 1.0 || (d->dropP > 0.0 && !Pd)) return W2L_EINVAL;
  if (posTable && (d->W <= 0 || d->rlo < 0)) return W2L_EINVAL;
  // float4 access: head slices, rows and the table 16-byte aligned
  if ((d->ld & 3) || (d->ldc & 3) || ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)ctx) | ((uintptr_t)posTable) |
                                       ((uintptr_t)P) | ((uintptr_t)Pd)) & 15))
    return W2L_EUNSUPPORTED;
  AfP p;
  p.q = q; p.k = k; p.v = v; p.E = posTable; p.keyLen = keyLen; p.P = P; p.Pd = Pd; p.ctx = ctx;
  p.ctxI = images ? *ctxImages : w2l_bf16_image_sink{nullptr, 0, nullptr, 0};
  p.B = d->B; p.H = d->H; p.T = d->T; p.ld = d->ld; p.ldc = d->ldc; p.W = d->W; p.n0 = d->n0; p.rlo = d->rlo; p.scale = d->scale;
  p.thr = dropout_threshold(d->dropP);
  p.seed = d->dropSeed; p.stream = d->dropStream;
  p.keepScale = (float)(1.0 / (1.0 - d->dropP));
  const int nt = (d->T + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
#define W2L_AF(NTv, Dv) if (nt <= NTv && d->d == Dv) return af_launch<NTv, Dv>(p, s);
  W2L_AF(2, 32) W2L_AF(4, 32) W2L_AF(6, 32)
  W2L_AF(2, 256) W2L_AF(4, 256) W2L_AF(6, 256)
#undef W2L_AF
  return W2L_EUNSUPPORTED;
}

W2L_API int w2l_attn_fused_forward(const w2l_attn_fused_desc* d, const float* q, const float* k, const float* v, const float* posTable,
                                   const int* keyLen, float* P, float* Pd, float* ctx, w2l_stream_t stream) {
  if (!ctx) return W2L_EINVAL;
  return w2l_attn_fused_forward_images(d, q, k, v, posTable, keyLen, P, Pd, ctx, nullptr, stream);
}
