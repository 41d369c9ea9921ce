// experiment (not product): what limits fcc_big_gemm?  variants of the same access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#pragma clang diagnostic ignored "-Wunused-value"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int U = 4;
template <int MODE>  // 0: loads+mfma (as product), 1: loads only, 2: mfma only, 3: loads only pack (no op operand)
__global__ __launch_bounds__(256) void k(const float4* __restrict__ pack, const float4* __restrict__ op, float* __restrict__ out,
                                         int NC, int SW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x / SW, sw = blockIdx.x - g * SW;
  const int nStages = NC / U;
  const int ss = sw * 4 + wave, nss = 4 * SW;
  const int s0 = (int)((long long)nStages * ss / nss), s1 = (int)((long long)nStages * (ss + 1) / nss);
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0; acc1[r] = 0; }
  const float4* pa0 = pack + ((size_t)(2 * g) * NC) * 64 + lane;
  const float4* pa1 = pack + ((size_t)(2 * g + 1) * NC) * 64 + lane;
  const float4* pe = op + lane;
  float4 A0[U], A1[U], E[U], B0[U], B1[U], F[U];
  float sink = 0.f;
  auto load = [&](float4 (&a0)[U], float4 (&a1)[U], float4 (&e)[U], int s) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = (size_t)s * U + u;
      if (MODE != 2) { a0[u] = pa0[c * 64]; a1[u] = pa1[c * 64]; }
      else { a0[u] = make_float4(1, 1, 1, 1); a1[u] = a0[u]; }
      if (MODE == 0 || MODE == 1) e[u] = pe[c * 64]; else e[u] = make_float4(1, 1, 1, 1);
    }
  };
  auto comp = [&](float4 (&a0)[U], float4 (&a1)[U], float4 (&e)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 1 || MODE == 3) {
        sink += a0[u].x + a0[u].y + a0[u].z + a0[u].w + a1[u].x + a1[u].y + a1[u].z + a1[u].w + e[u].x + e[u].y + e[u].z + e[u].w;
      } else {
        const float x0[4] = {a0[u].x, a0[u].y, a0[u].z, a0[u].w};
        const float x1[4] = {a1[u].x, a1[u].y, a1[u].z, a1[u].w};
        const float ev[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], x0[q], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ev[q], x1[q], acc1, 0, 0, 0);
        }
      }
    }
  };
  if (MODE == 4) {
    float4 C0[U], C1[U], H[U];
    auto load4 = [&](float4 (&a0)[U], float4 (&a1)[U], float4 (&e)[U], int s) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t c = (size_t)s * U + u;
        a0[u] = pa0[c * 64]; a1[u] = pa1[c * 64]; e[u] = pe[c * 64];
      }
    };
    if (s0 < s1) load4(A0, A1, E, s0);
    if (s0 + 1 < s1) load4(B0, B1, F, s0 + 1);
    for (int s = s0; s < s1; s += 3) {
      if (s + 2 < s1) load4(C0, C1, H, s + 2);
      comp(A0, A1, E);
      if (s + 3 < s1) load4(A0, A1, E, s + 3);
      if (s + 1 < s1) comp(B0, B1, F);
      if (s + 4 < s1) load4(B0, B1, F, s + 4);
      if (s + 2 < s1) comp(C0, C1, H);
    }
  } else {
  if (s0 < s1) load(A0, A1, E, s0);
  for (int s = s0; s < s1; s += 2) {
    if (s + 1 < s1) load(B0, B1, F, s + 1);
    comp(A0, A1, E);
    if (s + 2 < s1) load(A0, A1, E, s + 2);
    if (s + 1 < s1) comp(B0, B1, F);
  }
  }
  float v = sink;
  for (int r = 0; r < 16; ++r) v += acc0[r] + acc1[r];
  if (v == 12345.678f) out[blockIdx.x] = v;
}

template <int MODE>
float run(const float4* pack, const float4* op, float* out, int G, int SW, int NC) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(G * SW), dim3(256), 0, 0, pack, op, out, NC, SW);
  hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<MODE>, dim3(G * SW), dim3(256), 0, 0, pack, op, out, NC, SW);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}

int main() {
  const int Np = 10048, Kp = 10016, NC = Kp / 8, G = Np / 64;
  float4 *pack, *op; float* out;
  CK(hipMalloc(&pack, (size_t)Np * Kp * 4));
  CK(hipMalloc(&op, (size_t)32 * Kp * 4));
  CK(hipMalloc(&out, 1 << 20));
  CK(hipMemset(pack, 0x3c, (size_t)Np * Kp * 4));
  CK(hipMemset(op, 0x3c, (size_t)32 * Kp * 4));
  const double bytes = (double)Np * Kp * 4;
  for (int SW : {3, 6, 8, 13}) {
    float t0 = run<0>(pack, op, out, G, SW, NC), t1 = run<1>(pack, op, out, G, SW, NC), t2 = run<2>(pack, op, out, G, SW, NC),
          t3 = run<3>(pack, op, out, G, SW, NC), t4 = run<4>(pack, op, out, G, SW, NC);
    printf("   3-set pipeline: %.1f us (%.2f TB/s)\n", t4, bytes / t4 / 1e6);
    printf("SW=%2d WGs=%4d: full %.1f us (%.2f TB/s) | loads only %.1f us (%.2f TB/s) | pack-only loads %.1f us (%.2f TB/s) | mfma only %.1f us\n",
           SW, G * SW, t0, bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t3, bytes / t3 / 1e6, t2);
  }
  return 0;
}
