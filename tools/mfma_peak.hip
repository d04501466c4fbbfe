// Achievable FP64 matrix-core rate of this GPU: 16 independent 16x16x4 accumulators per wave,
// no memory traffic, 2 waves per SIMD (the occupancy of bfStageKernelC128Mfma).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double bf_d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void peak(double *out, int iters, double a0, double b0) {
  bf_d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (bf_d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  int const blocks = 256 * 2 * 8, iters = 2000;
  double *d;
  hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  peak<<<blocks, 256>>>(d, 10, 1e-3, 1e-3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  peak<<<blocks, 256>>>(d, iters, 1e-3, 1e-3);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 32 * 2048.0;
  printf("{\"fp64_mfma_16x16x4_tflops\": %.2f, \"ms\": %.3f}\n", flops / ms / 1e9, ms);
  return 0;
}
