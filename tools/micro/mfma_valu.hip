// mfma_valu.hip -- does a wave that issues DEPENDENT v_mfma_f32_32x32x2_f32 chains starve the other waves of its SIMD?
// 12 waves per CU: waves 0-7 run MFMA chains (one accumulator: every MFMA waits for the previous one; or two
// accumulators alternating), waves 8-11 run a fixed amount of "mover" work (VALU selects + ds_write2_b32 + final
// lgkmcnt(0)) and report the cycles it took.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/micro/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MOVER, int PACE = 0>   // MOVER 0: VALU + LDS writes, 1: VALU only, 2: LDS writes only; PACE: s_nop 15 per MFMA
__global__ __launch_bounds__(768) void k(float* out, long long* clk, int iters, int mfmaWaves) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long cstart = clock64();
  if (wave < 8) {
    if (wave >= mfmaWaves) return;
    float a[36], b[36];
#pragma unroll
    for (int s = 0; s < 36; ++s) { a[s] = 1.0f + tid * 1e-6f + s; b[s] = 0.5f + s * 1e-3f; }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 36; ++s) {
        acc[s % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[s % NACC], 0, 0, 0);
        // PACE wait states (4 cycles each) of s_nop after every MFMA
#pragma unroll
        for (int z = 0; z < PACE / 16; ++z) asm volatile("s_nop 15");
        if (PACE % 16 == 1) asm volatile("s_nop 0"); if (PACE % 16 == 2) asm volatile("s_nop 1"); if (PACE % 16 == 3) asm volatile("s_nop 2");
        if (PACE % 16 == 4) asm volatile("s_nop 3"); if (PACE % 16 == 5) asm volatile("s_nop 4"); if (PACE % 16 == 6) asm volatile("s_nop 5");
        if (PACE % 16 == 7) asm volatile("s_nop 6"); if (PACE % 16 == 8) asm volatile("s_nop 7"); if (PACE % 16 == 9) asm volatile("s_nop 8");
        if (PACE % 16 == 10) asm volatile("s_nop 9"); if (PACE % 16 == 11) asm volatile("s_nop 10"); if (PACE % 16 == 12) asm volatile("s_nop 11");
        if (PACE % 16 == 13) asm volatile("s_nop 12"); if (PACE % 16 == 14) asm volatile("s_nop 13"); if (PACE % 16 == 15) asm volatile("s_nop 14");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) t += acc[i][q];
    out[blockIdx.x * blockDim.x + tid] = t;
    if (tid == 0) clk[256 + blockIdx.x] = clock64() - cstart;
  } else {
    float v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = tid * 0.5f + i;
    float* d = lds + (tid - 512) * 3 + 64;
    const long long c0 = clock64();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int i = 0; i < 24; i += 2) {
        float x0 = v[i], x1 = v[i + 1];
        if (MOVER != 2) { x0 = (lane + it) & 1 ? x0 : 0.f; x1 = (lane + it) & 2 ? x1 : 0.f; }
        if (MOVER != 1) { d[i * 71] = x0; d[i * 71 + 71] = x1; } else { v[i] = x0 + 1.f; v[i + 1] = x1 + 1.f; }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long c1 = clock64();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 24; ++i) t += v[i];
    out[blockIdx.x * blockDim.x + tid] = t;
    if (tid == 512) clk[blockIdx.x] = (c1 - c0) / 64;
  }
}

template <int NACC, int MOVER, int PACE = 0>
void run(const char* name, int mfmaWaves) {
  const int blocks = 256, threads = 768, iters = 200;
  float* out; long long* clk;
  (void)hipMalloc(&out, sizeof(float) * blocks * threads);
  (void)hipMalloc(&clk, sizeof(long long) * 2 * blocks);
  (void)hipMemset(clk, 0, sizeof(long long) * 2 * blocks);
  auto kern = k<NACC, MOVER, PACE>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, out, clk, iters, mfmaWaves);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(2 * blocks);
  (void)hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double m = 0, mm = 0;
  for (int i = 0; i < blocks; ++i) { m += h[i]; mm += h[blocks + i]; }
  const double perMfma = mfmaWaves ? mm / blocks / (36.0 * iters * (mfmaWaves / 4)) : 0;
  printf("%-64s MFMA waves/CU %d: mover pass = %7.0f cycles;  cycles per MFMA per SIMD %.1f (64 = peak)\n", name, mfmaWaves, m / blocks, perMfma);
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  run<1, 0>("no MFMA waves", 0);
  run<1, 0>("one accumulator (dependent chain), VALU + LDS mover", 4);
  run<1, 0>("one accumulator (dependent chain), VALU + LDS mover", 8);
  run<2, 0>("two accumulators alternating, VALU + LDS mover", 8);
  run<4, 0>("four accumulators alternating, VALU + LDS mover", 8);
  run<1, 1>("one accumulator, VALU-only mover", 8);
  run<1, 2>("one accumulator, LDS-only mover", 8);
  run<2, 1>("two accumulators, VALU-only mover", 8);
  run<2, 2>("two accumulators, LDS-only mover", 8);
  run<1, 0, 12>("one accumulator + 12 wait states per MFMA", 4);
  run<1, 0, 13>("one accumulator + 13 wait states per MFMA", 4);
  run<1, 0, 14>("one accumulator + 14 wait states per MFMA", 4);
  run<1, 0, 15>("one accumulator + 15 wait states per MFMA", 4);
  run<1, 0, 16>("one accumulator + 16 wait states per MFMA", 4);
  run<1, 0, 26>("one accumulator + 26 wait states per MFMA", 8);
  run<1, 0, 28>("one accumulator + 28 wait states per MFMA", 8);
  run<1, 0, 29>("one accumulator + 29 wait states per MFMA", 8);
  run<1, 0, 30>("one accumulator + 30 wait states per MFMA", 8);
  run<1, 0, 31>("one accumulator + 31 wait states per MFMA", 8);
  run<1, 0, 32>("one accumulator + 32 wait states per MFMA", 8);
  return 0;
}
