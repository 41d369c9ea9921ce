// mfma_rate.hip -- what does v_mfma_f32_32x32x2_f32 sustain on MI355X in the shapes the TDS kernels use?
// Chains of NK dependent MFMAs (one accumulator), W waves per SIMD, optionally with LDS operations slotted between
// the MFMAs (the overlap-add / fragment traffic of conv_tds_rs3.hpp), accumulator resets and workgroup barriers.
// Reports TFLOP/s, % of the 157.3 TFLOP/s peak, and the shader clock (clock64 / wall_clock64).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/micro/mfma_rate.hip && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NK, int LDSOPS, int RESET, int BARRIER>
__global__ __launch_bounds__(768) void k(float* out, long long* clk, int iters) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  float a[NK], b[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) { a[s] = 1.0f + tid * 1e-6f + s; b[s] = 0.5f + s * 1e-3f; }
  for (int e = tid; e < 8192; e += blockDim.x) lds[e] = 0.f;
  __syncthreads();
  float* o = lds + (tid >> 6) * 600 + (lane & 31) * 15 + (lane >> 5) * 4;
  const long long c0 = clock64(), w0 = wall_clock64();
  f32x16 acc, accP;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q] = 0.f; accP[q] = 0.f; }
  for (int it = 0; it < iters; ++it) {
    float old[16];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
      if (LDSOPS >= 1 && LDSOPS < 10 && s < 32) {
        if (s % 2 == 0) old[s / 2] = o[s / 2];
        else o[s / 2] = old[s / 2] + accP[s / 2];
      }
      if (LDSOPS >= 2 && LDSOPS < 10) a[s] += lds[4096 + ((tid + s * 67) & 2047)] * 1e-30f;
      if (LDSOPS == 10) asm volatile("s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1" ::: "s40", "s41", "s42", "s43");
      if (LDSOPS == 11) asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %1, %1, 1\n\tv_add_u32 %2, %2, 1\n\tv_add_u32 %3, %3, 1" : "+v"(b[(s + 5) % NK]), "+v"(b[(s + 9) % NK]), "+v"(b[(s + 13) % NK]), "+v"(b[(s + 17) % NK]));
      if (LDSOPS == 12) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt lgkmcnt(0)");
      __builtin_amdgcn_sched_barrier(0);
    }
    if (RESET) {
      accP = acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    }
    if (BARRIER && (it & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) t += acc[q] + accP[q];
  out[blockIdx.x * blockDim.x + tid] = t;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int NK, int LDSOPS, int RESET, int BARRIER>
void run(const char* name, int wavesPerCu) {
  const int blocks = 256, threads = 64 * wavesPerCu, iters = 400;
  float* out; long long* clk;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&clk, sizeof(long long) * 2 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto kern = k<NK, LDSOPS, RESET, BARRIER>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 100 * 1024, 0, out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<long long> h(2 * blocks);
  hipMemcpy(h.data(), clk, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
  const double flop = 4096.0 * NK * iters * wavesPerCu * blocks;
  const double mhz = 100.0 * (double)h[0] / (double)h[1];
  printf("%-58s %2d waves/CU: %8.1f us  %6.1f TF/s = %5.1f %% of 157.3   shader clock %.0f MHz   cycles per MFMA per SIMD %.1f\n", name, wavesPerCu,
         ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, mhz, (double)h[0] / ((double)NK * iters * wavesPerCu / 4.0));
  hipFree(out); hipFree(clk);
}

int main() {
  for (int w : {4, 8, 12}) run<35, 0, 0, 0>("chain of 35 dependent MFMAs, nothing else", w);
  for (int w : {4, 8, 12}) run<35, 0, 1, 0>("+ accumulator hand-over and reset per chain", w);
  for (int w : {4, 8, 12}) run<35, 1, 1, 0>("+ 16 ds_read + 16 ds_write between the MFMAs", w);
  for (int w : {4, 8, 12}) run<35, 2, 1, 0>("+ one fragment ds_read per MFMA as well", w);
  for (int w : {8, 12}) run<35, 2, 1, 1>("+ workgroup barrier every 70 MFMAs", w);
  for (int w : {8}) run<27, 2, 1, 1>("the same with chains of 27", w);
  for (int w : {4, 8}) run<35, 10, 1, 0>("chains + 4 SALU instructions per MFMA", w);
  for (int w : {4, 8}) run<35, 11, 1, 0>("chains + 4 VALU instructions per MFMA", w);
  for (int w : {4, 8}) run<35, 12, 1, 0>("chains + 4 s_waitcnt per MFMA", w);
  return 0;
}
