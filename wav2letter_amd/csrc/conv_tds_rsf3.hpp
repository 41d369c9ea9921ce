// conv_tds_rsf3.hpp -- the role-swapped TDS filter gradient (see conv_tds_rs.hip: tap = ga*GB + gb, both operands
// time-shifted, K = time) with the machine shape of conv_tds_rs3.hpp: ONE workgroup per CU, wave roles split.
//   * waves 0..HH-1 CONSUMERS: each owns one mel row of the tile and does nothing but fragment reads (two K steps per
//     ds_read2_b32, next pair of steps already in registers) and NRT x NCT independent MFMA accumulators that live in
//     registers over ALL tiles of the workgroup;
//   * 4 MOVERS: raw buffer loads (frames outside the utterance arrive as zeros: no selects) -> registers -> the two
//     time-fastest slabs of tile r, the loads of tile r+1 in flight, while the consumers are on tile r-1; the slabs are
//     double-buffered, one LDS-only barrier per tile.
// What this buys over tds_conv_rsf_k: there every workgroup staged (global latency exposed), multiplied and staged again,
// and the 2-3 co-resident workgroups ran in lockstep; here staging costs the SIMD ~70 instructions per tile beside
// 144-384 MFMAs per consumer wave (measured rule, tools/micro: every non-MFMA instruction a SIMD issues costs the
// matrix pipe ~7-10 cycles, nothing overlaps for free -- so count instructions).
//     C = 10: GA = 3, GB = 7, 8 mel rows, strips of 96 frames: 1 x 3 tiles of 32x32 per wave, two waves per SIMD
//     C = 18: GA = 7, GB = 3, 4 mel rows, strips of 96 frames: 4 x 2 tiles per wave, one wave per SIMD
// The partial-sum layout and the final reduction are tds_rsf_reduce_k's (deterministic: wave order, then workgroup order).
#pragma once

namespace w2l {

struct TdsRsf3P {
  const float* x;   // [B][Tin][H][C]
  const float* dy;  // [B][Tout][H][C]
  int B, Tin, Tout, H, kw, padl;
  int nStrips, hBlocks, nTiles;
};

template <int C, int GA, int GB, int HH, int TS>
struct Rsf3Cfg {
  static constexpr int ROWS = HH * C;
  static constexpr int Q = ROWS / 4;
  static constexpr int FSTEP = 256 / Q;
  static constexpr int NRT = (GA * C + 1 + 31) / 32;
  static constexpr int NCT = (GB * C + 31) / 32;
  static constexpr int XF = TS + (GA - 1) * GB;   // x slab frames
  static constexpr int DF = TS + GB - 1;          // dy slab frames
  static constexpr int FT = XF | 1;
  static constexpr int DT = DF | 1;
  static constexpr int ZT = (TS + 2) | 1;         // zero / ones rows
  static constexpr int XV = (XF + FSTEP - 1) / FSTEP, DV = (DF + FSTEP - 1) / FSTEP;
  static constexpr int XOFF = 0, DOFF = ROWS * FT, ZOFF = DOFF + ROWS * DT, OOFF = ZOFF + ZT, BUFF = (OOFF + ZT + 3) / 4 * 4;
  static constexpr int ACCF = NRT * NCT * 16 * 64;   // floats of one wave's accumulators
  static constexpr int LDSF = 2 * BUFF > HH * ACCF ? 2 * BUFF : HH * ACCF;   // (the slabs; at the end one accumulator image per consumer wave)
  static constexpr size_t LDS = (size_t)LDSF * sizeof(float);
  static constexpr int WAVES = HH + 4;
  static_assert(TS % 4 == 0 && C % 2 == 0 && ROWS % 4 == 0, "strip length / channel count");
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int C, int GA, int GB, int HH, int TS>
__global__ __launch_bounds__((HH + 4) * 64) void tds_conv_rsf3_k(TdsRsf3P p, float* __restrict__ partial) {
  using Cfg = Rsf3Cfg<C, GA, GB, HH, TS>;
  constexpr int NRT = Cfg::NRT, NCT = Cfg::NCT, FT = Cfg::FT, DT = Cfg::DT, Q = Cfg::Q, FSTEP = Cfg::FSTEP, XV = Cfg::XV, DV = Cfg::DV,
                BUFF = Cfg::BUFF, NT = (HH + 4) * 64;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // constant rows of both buffers
  for (int e = tid; e < 2 * Cfg::ZT; e += NT) {
    const int b = e / Cfg::ZT, i = e - b * Cfg::ZT;
    lds[b * BUFF + Cfg::ZOFF + i] = 0.f;
    lds[b * BUFF + Cfg::OOFF + i] = 1.f;
  }
  // tiles of this workgroup: blockIdx.x, blockIdx.x + gridDim.x, ...
  const int n = p.nTiles > (int)blockIdx.x ? (p.nTiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;

  if (wave < HH) {
    // ================================================================================================ consumers
    int ab[NRT], bb[NCT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const int m = 32 * rt + r, ga = m / C, ci = m - ga * C;
      ab[rt] = (m < GA * C ? Cfg::XOFF + (wave * C + ci) * FT + ga * GB : m == GA * C ? Cfg::OOFF : Cfg::ZOFF) + hf;
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
      const int nn = 32 * ct + r, gb = nn / C, co = nn - gb * C;
      bb[ct] = (gb < GB ? Cfg::DOFF + (wave * C + co) * DT + (GB - 1 - gb) : Cfg::ZOFF) + hf;
    }
    f32x16 acc[NRT][NCT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[rt][ct][q] = 0.f;

    rs3_barrier();
    for (int rd = 0; rd < n + 1; ++rd) {
      if (rd >= 1) {
        const float* buf = lds + ((rd - 1) & 1) * BUFF;
        // K loop over the strip: step s covers t' = s0 + 2s (lanes 0-31) and s0 + 2s + 1 (lanes 32-63); two steps per
        // fragment read, the next pair in registers before this pair's MFMAs
        float a[2][2][NRT], bv[2][2][NCT];
        auto load = [&](int slot, int s) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) { a[slot][0][rt] = buf[ab[rt] + 2 * s]; a[slot][1][rt] = buf[ab[rt] + 2 * s + 2]; }
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) { bv[slot][0][ct] = buf[bb[ct] + 2 * s]; bv[slot][1][ct] = buf[bb[ct] + 2 * s + 2]; }
        };
        load(0, 0);
#pragma unroll
        for (int s = 0; s < TS / 2; s += 2) {
          const int slot = (s / 2) & 1;
          if (s + 2 < TS / 2) load(slot ^ 1, s + 2);
          __builtin_amdgcn_sched_barrier(0);   // (pinned: hipcc otherwise sinks the reads to their first use and waits on every one)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
              for (int ct = 0; ct < NCT; ++ct)
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[slot][h][rt], bv[slot][h][ct], acc[rt][ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      rs3_barrier();
    }
    // ---- every wave's accumulators to its own LDS region (register layout kept: [tile][q][lane]); summed below
    float* mine = lds + wave * Cfg::ACCF;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int q = 0; q < 16; ++q) mine[((rt * NCT + ct) * 16 + q) * 64 + lane] = acc[rt][ct][q];
    rs3_barrier();
  } else {
    // ================================================================================================ movers
    __builtin_amdgcn_s_setprio(3);
    const int mt = tid - HH * 64;
    const int f0 = mt / Q, q4 = 4 * (mt - f0 * Q);
    const int fc = f0 < FSTEP ? f0 : FSTEP - 1;   // the spare threads of the last wave repeat chunk FSTEP-1 (same data, same addresses)
    const int HC = p.H * C;
    const int vbase = (fc * HC + q4) * 4;
    float4 xr[XV], dr[DV];
#pragma unroll
    for (int v = 0; v < XV; ++v) xr[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int v = 0; v < DV; ++v) dr[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int tile) {
      const int hb = tile % p.hBlocks, st = (tile / p.hBlocks) % p.nStrips, b = tile / (p.hBlocks * p.nStrips);
      const int s0 = st * TS;   // first t' of the strip
      // one buffer per utterance: frames before 0 wrap to offsets >= 2^31, frames past the end to offsets >= the size: zeros
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)b * p.Tin * HC), 0, p.Tin * HC * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + (size_t)b * p.Tout * HC), 0, p.Tout * HC * 4, 0x00020000);
      const int ox = vbase + ((s0 - p.padl) * HC + hb * HH * C) * 4;        // x slab frame f <-> input frame s0 - padl + f
      const int od = vbase + ((s0 - (GB - 1)) * HC + hb * HH * C) * 4;      // dy slab frame f <-> output frame s0 - (GB - 1) + f
#pragma unroll
      for (int v = 0; v < XV; ++v) xr[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, ox + v * (FSTEP * HC * 4), 0, 0));
#pragma unroll
      for (int v = 0; v < DV; ++v) dr[v] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rd, od + v * (FSTEP * HC * 4), 0, 0));
    };
    auto stage = [&](float* buf) {
      float* dx = buf + Cfg::XOFF + q4 * FT + fc;
#pragma unroll
      for (int v = 0; v < XV; ++v)
        if (FSTEP * (v + 1) <= FT || fc + FSTEP * v < FT) {
          dx[FSTEP * v] = xr[v].x; dx[FT + FSTEP * v] = xr[v].y; dx[2 * FT + FSTEP * v] = xr[v].z; dx[3 * FT + FSTEP * v] = xr[v].w;
        }
      float* dd = buf + Cfg::DOFF + q4 * DT + fc;
#pragma unroll
      for (int v = 0; v < DV; ++v)
        if (FSTEP * (v + 1) <= DT || fc + FSTEP * v < DT) {
          dd[FSTEP * v] = dr[v].x; dd[DT + FSTEP * v] = dr[v].y; dd[2 * DT + FSTEP * v] = dr[v].z; dd[3 * DT + FSTEP * v] = dr[v].w;
        }
    };
    if (n > 0) fetch(blockIdx.x);
    rs3_barrier();
    for (int rd = 0; rd < n + 1; ++rd) {
      if (rd < n) stage(lds + (rd & 1) * BUFF);   // tile rd-2's readers passed the last barrier
      if (rd + 1 < n) fetch(blockIdx.x + (rd + 1) * gridDim.x);
      rs3_barrier();
    }
    rs3_barrier();
  }
  // the mel rows of the workgroup, added in wave order
  float* dst = partial + (size_t)blockIdx.x * Cfg::ACCF;
  for (int e = tid; e < Cfg::ACCF; e += NT) {
    float t = lds[e];
#pragma unroll
    for (int w = 1; w < HH; ++w) t += lds[w * Cfg::ACCF + e];
    dst[e] = t;
  }
}

}  // namespace w2l
