// clock_probe.hip -- what clock do the one-wave-per-utterance criterion scans run at, and what does ONE wave issue per cycle?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o tools/micro/clock_probe && tools/micro/clock_probe
// Each kernel runs `blocks` workgroups of 64 threads (like the scans: 64 utterances -> 64 waves on 64 CUs) and reports
// shader cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) around a loop of N iterations of a given instruction mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void probe(int n, float* out, long long* cyc, long long* wall) {
  float u = threadIdx.x * 0.001f + 1.0f, e0 = 1.0001f, e1 = 0.9999f;
  float a0 = 0.f, a1 = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {          // 16 dependent-in-pairs v_fmac_f32_dpp (the FCC frame's rotations)
      asm volatile("s_nop 1\n\t"
                   "v_mul_f32_e32 %0, %2, %3\n\t"
                   "v_mul_f32_dpp %1, %2, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %0, %2, %3 row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
                   "v_fmac_f32_dpp %1, %2, %4 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                   : "=&v"(a0), "=&v"(a1) : "v"(u), "v"(e0), "v"(e1));
      u = (a0 + a1) * 0.0624f;
    } else if (MODE == 1) {   // 16 plain dependent-in-pairs v_fmac_f32 + the same tail
#pragma unroll
      for (int k = 0; k < 8; ++k) { a0 = fmaf(u, e0, a0); a1 = fmaf(u, e1, a1); }
      asm volatile("" : "+v"(a0), "+v"(a1));
      u = (a0 + a1) * 0.0624f; a0 = 0.f; a1 = 0.f;
    } else if (MODE == 2) {   // 16 INDEPENDENT v_fma_f32 (issue rate of one wave)
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = fmaf(u, e0, (float)k);
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(t[k]));
      u = t[0] * 0.5f + t[15] * 1e-9f;
    } else if (MODE == 3) {   // 16 independent v_fma_f64
      double t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = fma((double)u, 1.0001, (double)k);
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(t[k]));
      u = (float)(t[0] * 0.5 + t[15] * 1e-9);
    } else if (MODE == 4) {   // the permlane combine: mov + swap + add, twice (16 and 32)
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(u), __float_as_int(u), false, false);
      float s = __int_as_float(r[0]) + __int_as_float(r[1]);
      auto q = __builtin_amdgcn_permlane16_swap(__float_as_int(s), __float_as_int(s), false, false);
      u = (__int_as_float(q[0]) + __int_as_float(q[1])) * 0.25f;
    } else if (MODE == 5) {   // LDS write -> s_barrier -> LDS read round trip (one wave: barrier is a no-op wait)
      __shared__ float sh[64];
      sh[threadIdx.x] = u;
      __syncthreads();
      u = sh[(threadIdx.x + 1) & 63] * 0.999f;
      __syncthreads();
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 64 + threadIdx.x] = u + a0 + a1;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}

// several waves per workgroup running the SAME loop (no synchronisation): what does a wave get when 2, 4, 5, 8 waves share the CU?
template <int MODE>
__global__ void probe_mw(int n, float* out, long long* cyc) {
  float u = threadIdx.x * 0.001f + 1.0f, e0 = 1.0001f;
  const long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
    if (MODE == 0) {
      float t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = fmaf(u, e0, (float)k);
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(t[k]));
      u = t[0] * 0.5f + t[15] * 1e-9f;
    } else {
      double t[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) t[k] = fma((double)u, 1.0001, (double)k);
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(t[k]));
      u = (float)(t[0] * 0.5 + t[15] * 1e-9);
    }
  }
  const long long c1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = u;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}
template <int MODE> int run_mw(const char* what, int waves, int n) {
  const int blocks = 64;
  float* out; long long* cyc;
  CK(hipMalloc(&out, blocks * waves * 64 * sizeof(float))); CK(hipMalloc(&cyc, blocks * 16 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe_mw<MODE>, dim3(blocks), dim3(64 * waves), 0, 0, n, out, cyc);
    CK(hipDeviceSynchronize());
  }
  std::vector<long long> hc(blocks * 16);
  CK(hipMemcpy(hc.data(), cyc, blocks * 16 * 8, hipMemcpyDeviceToHost));
  printf("%-28s %d waves per workgroup: cycles / iteration per wave:", what, waves);
  for (int w = 0; w < waves; ++w) printf(" %.0f", (double)hc[w] / n);
  printf("\n");
  CK(hipFree(out)); CK(hipFree(cyc));
  return 0;
}

template <int MODE> int run(const char* what, int blocks, int n, int instrs) {
  float* out; long long *cyc, *wall;
  CK(hipMalloc(&out, blocks * 64 * sizeof(float))); CK(hipMalloc(&cyc, blocks * 8)); CK(hipMalloc(&wall, blocks * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(64), 0, 0, n, out, cyc, wall);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> hc(blocks), hw(blocks);
    CK(hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hw.data(), wall, blocks * 8, hipMemcpyDeviceToHost));
    if (rep == 2)
      printf("%-52s blocks %4d: %8.3f ms, %7.1f cycles / iteration (%.2f per instruction), shader clock %.0f MHz (s_memtime / 100 MHz wall)\n",
             what, blocks, ms, (double)hc[0] / n, (double)hc[0] / n / instrs, (double)hc[0] / (double)hw[0] * 100.0);
  }
  CK(hipFree(out)); CK(hipFree(cyc)); CK(hipFree(wall));
  return 0;
}

int main() {
  for (int blocks : {64, 1024}) {
    run<0>("16 v_fmac_f32_dpp (2 chains) + add + mul", blocks, 200000, 19);
    run<1>("16 v_fmac_f32 (2 chains) + add + mul", blocks, 200000, 18);
    run<2>("16 independent v_fma_f32 + 2", blocks, 200000, 18);
    run<3>("16 independent v_fma_f64 + cvt", blocks, 200000, 20);
    run<4>("permlane32_swap + permlane16_swap combine", blocks, 200000, 8);
    run<5>("ds_write, barrier, ds_read, barrier", blocks, 200000, 6);
  }
  for (int waves : {1, 2, 4, 5, 6, 8}) run_mw<0>("18 fp32 VALU per iteration,", waves, 100000);
  for (int waves : {1, 4, 5, 8}) run_mw<1>("20 fp64 VALU per iteration,", waves, 100000);
  return 0;
}
