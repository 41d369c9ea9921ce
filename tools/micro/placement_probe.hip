// placement_probe.hip -- where does the workgroup dispatcher put the workgroups of the criterion scans?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/placement_probe.hip -o /tmp/placement_probe && /tmp/placement_probe
// Every wave records HW_REG_HW_ID (SIMD, CU, SH, SE) and HW_REG_XCC_ID, then spins ~100 us so that the whole grid is resident at once.
// Reported per launch shape: distinct (XCC, SE, SH, CU) in use, the largest number of workgroups and of waves on one CU, the largest
// number of waves on one SIMD.  Shapes: the round-5 scans (64 x 128 threads), the meet-in-the-middle pairs (grid (64, 2)), FCC and FAC
// side by side on two streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void probe(unsigned* out, int spin, int ldsWords) {
  extern __shared__ unsigned lds[];
  const int wg = blockIdx.y * gridDim.x + blockIdx.x, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (ldsWords > 0 && threadIdx.x == 0) lds[ldsWords - 1] = 1;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    out[2 * (wg * nw + wave)] = hw;
    out[2 * (wg * nw + wave) + 1] = xcc;
  }
}

struct Shape { int gx, gy, threads, lds; const char* name; };

static void report(const char* name, const std::vector<unsigned>& h, int wgs, int nw) {
  std::map<std::tuple<int, int, int, int>, std::set<int>> cuWgs;
  std::map<std::tuple<int, int, int, int>, int> cuWaves;
  std::map<std::tuple<int, int, int, int, int>, int> simdWaves;
  for (int w = 0; w < wgs; ++w)
    for (int v = 0; v < nw; ++v) {
      const unsigned hw = h[2 * (w * nw + v)], xc = h[2 * (w * nw + v) + 1] & 0xf;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      auto key = std::make_tuple((int)xc, se, sh, cu);
      cuWgs[key].insert(w);
      cuWaves[key]++;
      simdWaves[std::make_tuple((int)xc, se, sh, cu, simd)]++;
    }
  int maxWg = 0, maxWv = 0, maxSimd = 0;
  std::map<int, int> histWg;
  for (auto& kv : cuWgs) { maxWg = std::max(maxWg, (int)kv.second.size()); histWg[(int)kv.second.size()]++; }
  for (auto& kv : cuWaves) maxWv = std::max(maxWv, kv.second);
  std::map<int, int> histSimd;
  for (auto& kv : simdWaves) { maxSimd = std::max(maxSimd, kv.second); histSimd[kv.second]++; }
  printf("%-46s %4d workgroups x %d waves: %3zu CUs in use, max %d workgroups / %d waves per CU, max %d waves per SIMD;  CUs by workgroup count:",
         name, wgs, nw, cuWgs.size(), maxWg, maxWv, maxSimd);
  for (auto& kv : histWg) printf(" %dx%d", kv.second, kv.first);
  printf(";  SIMDs by wave count:");
  for (auto& kv : histSimd) printf(" %dx%d", kv.second, kv.first);
  printf("\n");
}

int main() {
  unsigned *d0, *d1;
  CK(hipMalloc(&d0, 1 << 20));
  CK(hipMalloc(&d1, 1 << 20));
  const int spin = 10000;   // 100 MHz ticks = 100 us
  const Shape shapes[] = {{64, 1, 128, 0, "round-5 FCC (64 x 2 waves)"}, {64, 1, 320, 0, "round-5 FAC (64 x 5 waves)"},
                          {64, 2, 128, 26248, "MITM FCC (64 x 2 workgroups x 2 waves)"}, {64, 2, 320, 5200, "MITM FAC (64 x 2 workgroups x 5 waves)"},
                          {128, 1, 128, 26248, "128 x 2 waves, 1-D grid"}, {64, 1, 256, 26248, "64 x 4 waves"}, {64, 1, 640, 5200, "64 x 10 waves"}};
  for (const Shape& s : shapes) {
    const int wgs = s.gx * s.gy, nw = s.threads / 64;
    hipLaunchKernelGGL(probe, dim3(s.gx, s.gy), dim3(s.threads), s.lds, 0, d0, spin, s.lds / 4);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * wgs * nw);
    CK(hipMemcpy(h.data(), d0, h.size() * 4, hipMemcpyDeviceToHost));
    report(s.name, h, wgs, nw);
  }
  // FCC and FAC side by side on two streams (the ASG criterion's launch pattern)
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (int mitm = 0; mitm < 2; ++mitm) {
    const int gy = mitm ? 2 : 1;
    hipLaunchKernelGGL(probe, dim3(64, gy), dim3(128), 26248, s0, d0, spin, 26248 / 4);
    hipLaunchKernelGGL(probe, dim3(64, gy), dim3(320), 5200, s1, d1, spin, 5200 / 4);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h0(2 * 64 * gy * 2), h1(2 * 64 * gy * 5), all;
    CK(hipMemcpy(h0.data(), d0, h0.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), d1, h1.size() * 4, hipMemcpyDeviceToHost));
    // merged view: treat every wave as its own "workgroup" slot of one wave
    all = h0;
    all.insert(all.end(), h1.begin(), h1.end());
    report(mitm ? "two streams, MITM FCC + FAC (per wave)" : "two streams, round-5 FCC + FAC (per wave)", all, (int)all.size() / 2, 1);
  }
  return 0;
}
