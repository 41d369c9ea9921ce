// common.hpp -- shared device/host helpers for libw2l_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <math.h>

#include "../../include/w2l_hip.h"

#define W2L_API extern "C" __attribute__((visibility("default")))

namespace w2l {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return W2L_EHIP;
}

#define W2L_HIP_CHECK(expr)                         \
  do {                                              \
    hipError_t _e = (expr);                         \
    if (_e != hipSuccess) return ::w2l::hip_fail(_e); \
  } while (0)

#define W2L_LAUNCH_CHECK() W2L_HIP_CHECK(hipGetLastError())

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kWave = 64;

// ---- scale modes (CriterionUtils::computeScale) ---------------------------
__host__ __device__ inline float scale_of(int mode, int T, int L) {
  switch (mode) {
    case W2L_SCALE_INPUT_SZ: return T > 0 ? 1.0f / (float)T : 1.0f;
    case W2L_SCALE_INPUT_SZ_SQRT: return T > 0 ? sqrtf(1.0f / (float)T) : 1.0f;
    case W2L_SCALE_TARGET_SZ: return L > 0 ? 1.0f / (float)L : 1.0f;
    case W2L_SCALE_TARGET_SZ_SQRT: return L > 0 ? sqrtf(1.0f / (float)L) : 1.0f;
    default: return 1.0f;
  }
}

// natural log / exp through the hardware base-2 transcendentals (v_log_f32 / v_exp_f32, 1 ulp): the serial criterion
// scans pay for every instruction of their per-frame dependency chain, and `__logf` under -fno-fast-math expands to
// ocml's extended-precision sequence (~12 instructions).  Arguments are normal, positive (log) / <= ~0 (exp) here.
__device__ __forceinline__ float fast_logf(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
__device__ __forceinline__ float fast_expf(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// MaxDynamicSharedMemorySize opt-ins are a property of a function ON a device: `static bool seen[64] = {};` at the call site,
// true the first time the CURRENT device passes (a second device in the same process would otherwise launch without the opt-in;
// two threads racing here set the attribute twice, which is harmless)
__host__ inline bool first_on_device(bool (&seen)[64]) {
  int d = 0;
  (void)hipGetDevice(&d);
  if (d < 0 || d >= 64) d = 0;
  if (seen[d]) return false;
  seen[d] = true;
  return true;
}

// ---- wavefront helpers (wave = 64 lanes) ----------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// DPP row rotate-right by n inside each row of 16 lanes (dpp_ctrl 0x120 + n)
template <int N>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}

__device__ __forceinline__ float readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// max over all 64 lanes, result uniform. 4 DPP rotations give every lane its
// 16-lane row maximum; 4 v_readlane + 3 max combine the rows. No LDS.
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_row_ror<1>(v));
  v = fmaxf(v, dpp_row_ror<2>(v));
  v = fmaxf(v, dpp_row_ror<4>(v));
  v = fmaxf(v, dpp_row_ror<8>(v));
  float a = readlane(v, 0), b = readlane(v, 16), c = readlane(v, 32), d = readlane(v, 48);
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// max over the first ROWS rows of 16 lanes (ROWS = 2: lanes 0..31), result uniform.  The reduction inside a row is
// four `v_max_f32_dpp` written in place (one instruction per step where the builtin form costs a v_mov_dpp, a
// canonicalising v_max and the v_max proper); a DPP source written by the previous VALU instruction needs two wait
// states (s_nop 1).  Row-scoped DPP reads and writes the same 16 lanes in one pass, so in-place is safe.
template <int ROWS>
__device__ __forceinline__ float wave_max_rows(float v) {
  asm volatile(
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  float m = readlane(v, 0);
  if (ROWS > 1) m = fmaxf(m, readlane(v, 16));
  if (ROWS > 2) m = fmaxf(m, readlane(v, 32));
  if (ROWS > 3) m = fmaxf(m, readlane(v, 48));
  return m;
}

__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_row_ror<1>(v);
  v += dpp_row_ror<2>(v);
  v += dpp_row_ror<4>(v);
  v += dpp_row_ror<8>(v);
  float a = readlane(v, 0), b = readlane(v, 16), c = readlane(v, 32), d = readlane(v, 48);
  return (a + b) + (c + d);
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ float lane_shift_up(float v, float fill) {
  float r = __shfl_up(v, 1);
  return lane_id() == 0 ? fill : r;
}
__device__ __forceinline__ double lane_shift_up(double v, double fill) {
  double r = __shfl_up(v, 1);
  return lane_id() == 0 ? fill : r;
}
// value of lane-1 through DPP wave_shr:1 (no LDS round trip: ds_bpermute costs ~100 cycles in a serial scan)
__device__ __forceinline__ double lane_shift_up_dpp(double v, double fill) {
  const long long bits = __double_as_longlong(v), fb = __double_as_longlong(fill);
  const int lo = __builtin_amdgcn_update_dpp((int)fb, (int)bits, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(fb >> 32), (int)(bits >> 32), 0x138, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// value of lane+1 through DPP wave_shl:1 (lane 63 receives `fill`)
__device__ __forceinline__ double lane_shift_down_dpp(double v, double fill) {
  const long long bits = __double_as_longlong(v), fb = __double_as_longlong(fill);
  const int lo = __builtin_amdgcn_update_dpp((int)fb, (int)bits, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(fb >> 32), (int)(bits >> 32), 0x130, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float lane_shift_down(float v, float fill) {
  float r = __shfl_down(v, 1);
  return lane_id() == 63 ? fill : r;
}
__device__ __forceinline__ double lane_shift_down(double v, double fill) {
  double r = __shfl_down(v, 1);
  return lane_id() == 63 ? fill : r;
}

// ---- kernel-variant switches (A/B work, ablations): the PRODUCT library never reads the environment -- an
// inherited variable must not be able to change (let alone corrupt) training.  Only the probe build
// (-DW2L_PROBE -> libw2l_hip_probe.so, loaded by tools/ and by the variant tests) honours W2L_* variables.
#ifdef W2L_PROBE
inline const char* tune_env(const char* name) { return std::getenv(name); }
#else
inline const char* tune_env(const char*) { return nullptr; }
#endif

// A CU to itself: dynamic LDS nobody reads, sized so that no second workgroup of this kind fits beside the first (2 x 84 KiB > the
// CU's 160 KiB).  For the meet-in-the-middle ASG scans: 2 B workgroups of FullConnectionCriterion and 2 B of ForceAlignmentCriterion
// run side by side on two streams, each a chain of T / 2 dependent frames on one lone-wave pipeline, and where the dispatcher puts
// two of them on one CU both slow down (fcc_mitm_bwd 128 -> 140 ... 231 us, profiles/r06_run49_*).  With the request every one of
// the 4 B <= 256 workgroups gets a CU of its own by construction.  0 when they would not all fit (the request would serialise them).
constexpr unsigned kExclusiveLdsBytes = 84 * 1024;
inline unsigned exclusive_cu_lds(int B) {
  static const bool off = tune_env("W2L_ASG_SHARE_CUS") != nullptr;   // probe library: the plain placement (A/B runs)
  return (!off && 4 * B <= 256) ? kExclusiveLdsBytes : 0u;
}

// dynamic LDS request that gives a scan workgroup a CU of its own (exclusive_cu_lds), with the > 64 KiB opt-in of the function
inline unsigned mitm_excl(int B, const void* fn) {
  const unsigned bytes = exclusive_cu_lds(B);
  if (bytes) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return bytes;
}

// ---- stateless dropout hash: must stay bit-identical to oracle/nn_oracle.c --
__host__ __device__ inline uint32_t hash32(uint32_t idx, uint32_t seed, uint32_t stream) {
  uint32_t h = idx * 0x9E3779B1u + seed;
  h ^= stream * 0x85EBCA77u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__host__ __device__ inline bool keep_elem(uint64_t idx, uint32_t seed, uint32_t stream, uint32_t thr) {
  uint32_t h = hash32((uint32_t)idx ^ (uint32_t)(idx >> 32) * 0x27D4EB2Fu, seed, stream);
  return (h >> 8) >= thr;
}
inline uint32_t dropout_threshold(double p) {
  double t = p * 16777216.0;
  if (t < 0) t = 0;
  if (t > 16777216.0) t = 16777216.0;
  return (uint32_t)t;
}

}  // namespace w2l
