// runtime.hip -- library-level entry points (version, error state, self tests).
#include "common.hpp"

namespace w2l {
thread_local int g_last_hip_error = 0;

// exercises the DPP wave reductions and lane shifts against plain loops
__global__ void selftest_wave_ops(const float* in, float* out) {
  const int lane = threadIdx.x;
  float v = in[lane];
  out[lane] = wave_max(v);
  out[64 + lane] = wave_sum(v);
  out[128 + lane] = lane_shift_up(v, -1.f);
  out[192 + lane] = lane_shift_down(v, -2.f);
  out[256 + lane] = (float)lane_shift_up((double)v * 3.0, -3.0);
  out[320 + lane] = readlane(v, 17);
}

// a stand-in for a co-resident communication kernel: `blocks` workgroups hold their CU slots (threads, LDS) and spin
// on the wall clock for `micros` -- used by tools/gemm_corun.py to measure how the persistent GEMMs behave when an
// RCCL all-reduce of the overlapped gradient reduction shares the chip
__global__ void selftest_spin_k(long long ticks, float* sink) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) acc += lds[(threadIdx.x * 7) & 255];
  if (acc == -1.f) sink[0] = acc;
}
}  // namespace w2l

using namespace w2l;

W2L_API int w2l_selftest_spin(int blocks, int threads, int ldsBytes, int micros, float* sink, w2l_stream_t stream) {
  if (blocks <= 0 || threads < 256 || threads > 1024 || ldsBytes < 1024 || ldsBytes > 64 * 1024 || !sink) return W2L_EINVAL;
  hipLaunchKernelGGL(selftest_spin_k, dim3((unsigned)blocks), dim3((unsigned)threads), (size_t)ldsBytes, (hipStream_t)stream,
                     (long long)micros * 100, sink);  // wall_clock64 ticks at 100 MHz
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}

W2L_API const char* w2l_version(void) { return "w2l_hip 0.1 (gfx950)"; }
W2L_API int w2l_last_hip_error(void) { return g_last_hip_error; }

// in [64], out [384] device pointers
W2L_API int w2l_selftest_wave_ops(const float* in, float* out, w2l_stream_t stream) {
  if (!in || !out) return W2L_EINVAL;
  hipLaunchKernelGGL(selftest_wave_ops, dim3(1), dim3(64), 0, (hipStream_t)stream, in, out);
  W2L_LAUNCH_CHECK();
  return W2L_OK;
}
