// Layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction) -- could the ragged last slab of an item
// (1 - 12 of 16 rows are padding) run on it?  Feeds one-hot operands and reports which (block, i, k) / (block, k, j) / (block, i, j)
// every lane holds, plus the instruction's rate next to the 16 x 16 x 4 one.
//   hipcc -O2 --offload-arch=gfx950 tools/mfma_4x4x4_probe.hip -o tools/mfma_4x4x4_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void onehot(double *out, int la, int lb) {      // A = 1 in lane la only, B = 1 in lane lb only; D per lane
  int l = threadIdx.x;
  double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0, c = 0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  out[l] = d;
}
__global__ void rate4(double *out, int reps) {
  double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  for (int r = 0; r < reps; ++r) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void rate16(double *out, int reps) {
  double a = threadIdx.x * 0.5, b = 1.0 + threadIdx.x;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int r = 0; r < reps; ++r) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * 64 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  double *d; hipMalloc(&d, 256 * 8 * 64 * 8);
  std::vector<double> h(64);
  // for every pair (la, lb) find the lanes of D that light up: D lane ld = 1 iff (block, i) of la, (block, j) of lb match and k matches
  // decode by scanning: for each la, the set of lb that give any output, and where
  int firstD[64][64];
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      onehot<<<1, 64>>>(d, la, lb);
      hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
      int ld = -1, cnt = 0;
      for (int l = 0; l < 64; ++l) if (h[l] != 0.0) { ld = l; ++cnt; }
      firstD[la][lb] = cnt == 1 ? ld : (cnt == 0 ? -1 : -2);
    }
  printf("{\n\"d_lane_for_a_lane_x_b_lane\": [\n");
  for (int la = 0; la < 64; ++la) { printf(" ["); for (int lb = 0; lb < 64; ++lb) printf("%d%s", firstD[la][lb], lb < 63 ? "," : ""); printf("]%s\n", la < 63 ? "," : ""); }
  printf("],\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int const reps = 20000; float ms4, ms16;
  rate4<<<2048, 64>>>(d, 100); hipDeviceSynchronize();
  hipEventRecord(e0); rate4<<<2048, 64>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms4, e0, e1);
  rate16<<<2048, 64>>>(d, 100); hipDeviceSynchronize();
  hipEventRecord(e0); rate16<<<2048, 64>>>(d, reps); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms16, e0, e1);
  double const n = 2048.0 * reps * 4;
  printf("\"mfma_4x4x4_4b\": {\"ms\": %.2f, \"tflops\": %.2f, \"instr_per_us_per_cu\": %.2f},\n", ms4, n * 512 / ms4 / 1e9, n / ms4 / 1e3 / 256);
  printf("\"mfma_16x16x4\": {\"ms\": %.2f, \"tflops\": %.2f, \"instr_per_us_per_cu\": %.2f}\n}\n", ms16, n * 2048 / ms16 / 1e9, n / ms16 / 1e3 / 256);
  return 0;
}
