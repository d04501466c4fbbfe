#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
// where do the four wavefronts of a 256-thread workgroup land?  HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe(uint32_t *out) {
  __shared__ char big[41344];
  big[threadIdx.x] = 1;
  uint32_t hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  // keep the workgroup alive for a while so that the machine fills up
  uint64_t t0 = clock64(); while (clock64() - t0 < 200000) {}
  if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc + big[threadIdx.x]; }
}
int main() {
  int const nb = 2048;
  uint32_t *d; hipMalloc(&d, nb * 4 * 2 * 4);
  probe<<<nb, 256>>>(d);
  hipDeviceSynchronize();
  uint32_t *h = (uint32_t *)malloc(nb * 4 * 2 * 4);
  hipMemcpy(h, d, nb * 4 * 2 * 4, hipMemcpyDeviceToHost);
  int same = 0, distinct4 = 0;
  for (int b = 0; b < nb; ++b) {
    int simd[4], cu[4];
    for (int w = 0; w < 4; ++w) { uint32_t hw = h[(b * 4 + w) * 2]; simd[w] = (hw >> 4) & 3; cu[w] = (hw >> 8) & 15; }
    int mask = 0; for (int w = 0; w < 4; ++w) mask |= 1 << simd[w];
    distinct4 += mask == 15;
    if (b < 12) printf("wg %d: simd %d %d %d %d  cu %d %d %d %d  hw %08x\n", b, simd[0], simd[1], simd[2], simd[3], cu[0], cu[1], cu[2], cu[3], h[b * 8]);
  }
  printf("workgroups whose 4 wavefronts sit on 4 different SIMDs: %d of %d\n", distinct4, nb);
  return 0;
}
