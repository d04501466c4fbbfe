// What bounds the k-loop of the 64-RHS stage kernel?  The PRODUCT's own loop (bfMfmaSegment<4, 2> of
// butterfly_amd/csrc/bfhip_stage_mfma.h: Gauss's three multiplications, two A sets requested one k-step ahead, one X set, hand-placed
// waits) run on the bare machine without items, tables being rebuilt, epilogues or launch tails: every wavefront calls it `reps`
// times on its own window of A and on a table of X rows, ~1 s per case.
//   a: hbm   every wavefront streams its own 896 KiB of A, 2048 wavefronts -> 1.8 GB footprint: HBM
//      l2    all wavefronts of a group of 8 read the SAME 896 KiB, 256 groups -> 230 MB ... still beyond L2; `tiny`: every
//            wavefront reads one 64 KiB window over and over (L2 / TCP hits): the loop with the HBM latency taken out
//   x: rows of a 2 MB window (L2 hits) or of a 64 MB window (Infinity Cache / HBM)
// If the loop is bound by the latency of its A stream (requested ONE k-step = 24 MFMAs = 1536 cycles of a SIMD shared by two
// wavefronts ahead), `a: tiny` is much faster than `a: hbm`; if it is bound by the matrix pipe all cases agree.
//   hipcc -O3 --offload-arch=gfx950 -Ibutterfly_amd/csrc tools/mfma_loop_probe.hip -o tools/mfma_loop_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>
#include "../butterfly_amd/csrc/bfhip_stage_c128.h"

#include "../butterfly_amd/csrc/bfhip_stage_mfma.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define PROBE_COLS 1792u
// DMA: the loop that prefetches its fragments through the wavefront's LDS ring (bfMfmaSegmentDma) instead of bfMfmaSegment
template <int NT, int MS, int WPS, bool DMA>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void probeLoop(double *out, char const *A, char const *X, uint64_t aStride, uint32_t aBytes,
                                                                                             uint32_t xRows, uint32_t reps, uint32_t cols, uint32_t aWaves) {
  __shared__ uint32_t tab[PROBE_COLS + BF_MF_TABPAD];
  __shared__ __attribute__((aligned(16))) char ringMem[DMA ? BF_MF_DMA_RING : 16];
  int const lane = threadIdx.x;
  uint32_t const li = lane & 15, lk = lane >> 4;
  uint32_t const wave = blockIdx.x % aWaves;           // (the A allocation holds aWaves windows)
  // X rows: a run of consecutive rows like a piece's (1 KiB each at 64 RHS), groups of 8 neighbouring wavefronts walk the same rows
  uint32_t const row0 = ((wave >> 3) * 2654435761u) % xRows;
  for (uint32_t j = lane; j < cols + BF_MF_TABPAD; j += 64) tab[j] = j < cols ? ((row0 + j) % xRows) * 1024u : (row0 % xRows) * 1024u;
  waveSync();
  bf_d4 acc[3][2][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[g][m][t] = (bf_d4){0, 0, 0, 0};
  uint32_t const mr = 16u * MS;
  BfMfSeg sg;
  sg.stepA = 4u * mr * 16u;
  sg.ksteps = (cols + 3u) / 4u;
  sg.ra = bfMakeRsrc(A + (uint64_t)wave * aStride, aBytes);
  sg.rx = bfMakeRsrc(X, xRows * 1024u);
  sg.voffA = (lk * mr + li) * 16u;
  sg.cX = li * 16u;
  uint32_t const ring = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ringMem);
  for (uint32_t r = 0; r < reps; ++r) {
    if (DMA) bfMfmaSegmentDma<NT, MS>(acc, sg, tab, lk, ring, ring + (uint32_t)lane * 16u);
    else bfMfmaSegment<NT, MS>(acc, sg, tab, lk);
  }
  double sum = 0;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) sum += acc[g][m][t][v];
  out[(uint64_t)blockIdx.x * 64 + threadIdx.x] = sum;
}

template <int NT, int MS, int WPS, bool DMA>
static void run(char const *name, double *out, char const *A, char const *X, uint64_t aStride, uint32_t aBytes, uint32_t xRows, uint32_t reps) {
  int const grid = 256 * 4 * WPS;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  probeLoop<NT, MS, WPS, DMA><<<grid, 64>>>(out, A, X, aStride, aBytes, xRows, 8, PROBE_COLS, 2048);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probeLoop<NT, MS, WPS, DMA><<<grid, 64>>>(out, A, X, aStride, aBytes, xRows, reps, PROBE_COLS, 2048);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double const ksteps = (double)grid * reps * (PROBE_COLS / 4);
  double const mfmaFlops = ksteps * 3 * MS * NT * 2048.0;            // issued
  printf("\"%s\": {\"mfma_tflops_issued\": %.2f, \"frac_of_78.6\": %.3f, \"algorithmic_frac_x4_3\": %.3f, \"ms\": %.1f, \"ksteps_per_us\": %.1f, \"a_tbs\": %.2f, \"x_tbs\": %.2f, \"waves_per_simd\": %d},\n", name,
         mfmaFlops / ms / 1e9, mfmaFlops / ms / 1e9 / 78.6, mfmaFlops / ms / 1e9 / 78.6 * 4.0 / 3.0, ms, ksteps / ms / 1e3, ksteps * MS * 1024.0 / ms / 1e9, ksteps * NT * 1024.0 / ms / 1e9, WPS);
  fflush(stdout);
}

// out-of-range lanes of an LDS-DMA: zeros, or nothing written?  One wavefront: LDS slot pre-set to 7.0, a 1 KiB DMA through a
// descriptor of 512 bytes (lanes >= 32 are past its end), read back.
__global__ void dmaOobProbe(double *out, char const *src) {
  __shared__ __attribute__((aligned(16))) double slot[128];
  slot[threadIdx.x] = 7.0; slot[64 + threadIdx.x] = 7.0;
  __syncthreads();
  uint32_t const lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)slot);
  bf_i4 const r = bfMakeRsrc(src, 512u);
  bfDmaLoad<0>(lds, threadIdx.x * 16u, r, bfSgprConst<0>());
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = slot[threadIdx.x]; out[64 + threadIdx.x] = slot[64 + threadIdx.x];
}

// the two loops on the same operands must agree bit for bit (same MFMAs, same order per accumulator): ragged segment (cols % 4 != 0,
// an odd number of k-steps), the leaf descriptor ending exactly at the segment's end
template <int NT, int MS>
static int compareLoops(double *out, char const *A, char const *X, uint32_t cols) {
  uint32_t const aBytes = 16u * MS * cols * 16u;
  size_t const nOut = (size_t)64 * 64;
  std::vector<double> h0(nOut), h1(nOut);
  probeLoop<NT, MS, 2, false><<<64, 64>>>(out, A, X, 917504, aBytes, 2048, 2, cols, 2048);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h0.data(), out, nOut * 8, hipMemcpyDeviceToHost));
  probeLoop<NT, MS, 2, true><<<64, 64>>>(out, A, X, 917504, aBytes, 2048, 2, cols, 2048);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h1.data(), out, nOut * 8, hipMemcpyDeviceToHost));
  size_t bad = 0;
  double amax = 0;
  for (size_t i = 0; i < nOut; ++i) { bad += memcmp(&h0[i], &h1[i], 8) != 0; amax = fabs(h0[i]) > amax ? fabs(h0[i]) : amax; }
  printf("\"compare_nt%d_ms%d_cols%u\": {\"values\": %zu, \"different\": %zu, \"max_abs\": %.3e},\n", NT, MS, cols, nOut, bad, amax);
  return bad != 0;
}

int main(int argc, char **argv) {
  uint32_t const reps = argc > 1 ? (uint32_t)atoi(argv[1]) : 1200;
  uint32_t const aBytes = (PROBE_COLS / 4) * 2048u;          // one segment of A for MS = 2: 448 k-steps x 2 KiB
  size_t const aTotal = (size_t)aBytes * 2048;
  double *out; char *A, *X;
  CHECK(hipMalloc(&out, 256 * 4 * 3 * 64 * 8));
  CHECK(hipMalloc(&A, aTotal + 4096));
  CHECK(hipMalloc(&X, 64u << 20));
  char *X1g;
  CHECK(hipMalloc(&X1g, (size_t)1 << 30));
  {
    std::vector<double> h((size_t)(64u << 20) / 8);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)s / 9.3e18; }
    for (size_t off = 0; off < aTotal; off += h.size() * 8) CHECK(hipMemcpy(A + off, h.data(), (aTotal - off < h.size() * 8 ? aTotal - off : h.size() * 8), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(X, h.data(), 64u << 20, hipMemcpyHostToDevice));
    for (size_t off = 0; off < ((size_t)1 << 30); off += (size_t)64 << 20) CHECK(hipMemcpy(X1g + off, h.data(), (size_t)64 << 20, hipMemcpyHostToDevice));
  }
  printf("{\n");
  {
    double h[128];
    dmaOobProbe<<<1, 64>>>(out, X);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    int inRange = 1, zeros = 1, untouched = 1;
    std::vector<double> hx(128);
    CHECK(hipMemcpy(hx.data(), X, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 64; ++i) inRange &= h[i] == hx[i];
    for (int i = 64; i < 128; ++i) { zeros &= h[i] == 0.0; untouched &= h[i] == 7.0; }
    printf("\"dma_oob\": {\"in_range_lanes_copied\": %d, \"out_of_range_lanes_write_zeros\": %d, \"out_of_range_lanes_untouched\": %d, \"sample\": [%.3g, %.3g, %.3g]},\n", inRange, zeros, untouched, h[0], h[64], h[127]);
  }
  int bad = 0;
  bad |= compareLoops<4, 2>(out, A, X, 1792);
  bad |= compareLoops<4, 2>(out, A, X, 1787);      // 447 k-steps (odd), 3 columns in the last
  bad |= compareLoops<4, 1>(out, A, X, 333);
  bad |= compareLoops<3, 2>(out, A, X, 61);
  bad |= compareLoops<4, 2>(out, A, X, 5);
  bad |= compareLoops<4, 2>(out, A, X, 1);
  run<4, 2, 2, false>("register_loop_a_hbm_x_2mb", out, A, X, aBytes, aBytes, 2048, reps);
  run<4, 2, 2, true>("dma_loop_a_hbm_x_2mb", out, A, X, aBytes, aBytes, 2048, reps);
  run<4, 2, 2, false>("register_loop_a_hbm_x_64mb", out, A, X, aBytes, aBytes, 65536, reps);
  run<4, 2, 2, true>("dma_loop_a_hbm_x_64mb", out, A, X, aBytes, aBytes, 65536, reps);
  run<4, 2, 2, false>("register_loop_a_hbm_x_1gb", out, A, X1g, aBytes, aBytes, 1048576, reps);      // X rows from beyond the Infinity Cache
  run<4, 2, 2, true>("dma_loop_a_hbm_x_1gb", out, A, X1g, aBytes, aBytes, 1048576, reps);
  run<4, 2, 2, false>("register_loop_a_same_for_all_x_2mb", out, A, X, 0, aBytes, 2048, reps);       // every wavefront reads the same 896 KiB: L2 hits after the first
  run<4, 2, 2, true>("dma_loop_a_same_for_all_x_2mb", out, A, X, 0, aBytes, 2048, reps);
  printf("\"loops_differ\": %d\n}\n", bad);
  return bad;
}
