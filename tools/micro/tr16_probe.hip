// ds_read_b64_tr_b16 semantics probe (gfx950): which LDS element lands in (lane, j) when every lane supplies its own 8-byte
// address.  Model under test (cdna_hip_programming.md, LDS section): inside each 16-lane group the 16 lanes' 8-byte chunks form
// a 4 x 16 matrix (row r = chunks of lanes 4r .. 4r+3 of the group), lane l receives column (l & 15):
//   result[l][j] = chunk[16 (l >> 4) + 4 j + ((l & 15) >> 2)][l & 3]
//   hipcc --offload-arch=gfx950 -O3 tools/micro/tr16_probe.hip -o tools/micro/tr16_probe && tools/micro/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* chunkOf, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + 4 * chunkOf[l]));
  for (int j = 0; j < 4; ++j) out[4 * l + j] = (unsigned short)r[j];
}
int main() {
  int h[64], *d; unsigned short o[256], *od;
  hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
  for (int variant = 0; variant < 3; ++variant) {
    for (int l = 0; l < 64; ++l) h[l] = variant == 0 ? l : variant == 1 ? (l * 7 + 3) % 64 : 3 * l + 5;
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int want = 4 * h[16 * (l >> 4) + 4 * j + ((l & 15) >> 2)] + (l & 3);
        if (o[4 * l + j] != want) ++bad;
      }
    printf("variant %d: %d mismatches against the model\n", variant, bad);
    if (bad || variant == 0)
      for (int l = 0; l < 64; l += (bad ? 1 : 17)) printf("  lane %2d (chunk %3d): %4d %4d %4d %4d\n", l, h[l], o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  }
  return 0;
}
