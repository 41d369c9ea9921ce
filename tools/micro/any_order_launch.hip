// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two kernels of ONE stream run side by side on gfx950?
// two spin kernels of ~100 us (64 workgroups each) back to back: ~100 us if concurrent, ~200 us if serial.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/any_order_launch.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(long long ticks, int* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
int main() {
  int* d; hipMalloc(&d, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const long long ticks = 10000;  // 100 MHz wall clock: 100 us
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, ticks, d);
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, ticks, d + 1);
      else hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, mode == 1 ? hipExtAnyOrderLaunch : 0, ticks, d + 1);
      hipEventRecord(e1, s);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "plain launches" : mode == 1 ? "second launch any-order" : "hipExtLaunchKernelGGL, flags 0", ms * 1e3);
    }
  }
  return 0;
}
