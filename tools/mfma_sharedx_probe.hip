// Would the 64-RHS k-loop gain from sharing its X tiles between wavefronts?  (round 5; companion of tools/mfma_loop_probe.hip)
//
// The product's loop (bfMfmaSegment<4, 2>) has every wavefront fetch its own 4 KiB of X per k-step; the row chunks of a row group
// and the sibling groups of a radix-4 stage walk the SAME X rows, and on the real operand a third of those fetches miss the L2
// (5 GB of 16.5 GB per launch), which costs clock (mfma_loop_probe: 2.23 -> 2.04 GHz).  This probe runs the alternative on the
// bare machine: a workgroup of FOUR wavefronts with equal trip counts, each streaming its own leaf rows (registers, as in the
// product), the X tiles of a k-step fetched ONCE per workgroup -- wavefront w brings tile w as one LDS-DMA, two k-steps ahead, into
// a three-slot ring -- and read by all four with ds_read_b128 after a workgroup barrier per k-step.  Same MFMAs per wavefront.
// Cases: X rows private to a workgroup, taken from a 1 GB window (every X byte comes from HBM once per workgroup), against the
// register loop with the same rows private to a wavefront (once per wavefront) and shared by 4 list neighbours without any pacing.
//   hipcc -O3 --offload-arch=gfx950 -Ibutterfly_amd/csrc tools/mfma_sharedx_probe.hip -o tools/mfma_sharedx_probe.bin
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
#define PROBE_PAD 48u

template <int WPS, int MS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void probeSharedX(double *out, char const *A, char const *X, uint64_t aStride, uint32_t aBytes,
                                                                                                 uint32_t xRows, uint32_t reps, uint32_t cols, uint32_t aWaves, uint32_t cnt) {
  __shared__ uint32_t tab[PROBE_COLS + PROBE_PAD];
  __shared__ __attribute__((aligned(16))) char ringMem[BF_MF_SX_RING];
  uint32_t const lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t const li = lane & 15, lk = lane >> 4;
  uint32_t const wave = (blockIdx.x * 4u + w) % aWaves;
  uint32_t const row0 = (blockIdx.x * 2654435761u) % xRows;           // rows private to the workgroup
  for (uint32_t j = threadIdx.x; j < cols + PROBE_PAD; j += 256) tab[j] = j < cols ? ((row0 + j) % xRows) * 1024u : (row0 % xRows) * 1024u;
  __syncthreads();
  if (w >= cnt) return;                  // a bundle of fewer than four: the barriers wait for the surviving wavefronts only
  uint32_t mine = 0;
  for (uint32_t t = 0; t < 4u; ++t) if (t % cnt == w) mine |= 1u << t;
  mine = __builtin_amdgcn_readfirstlane(mine);
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
  for (uint32_t r = 0; r < reps; ++r) bfSxSegment<MS, true>(acc, sg, tab, lk, ring, lane, mine);      // the product's loop (bfhip_stage_mfma.h)
  double sum = 0;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) sum += acc[g][m][t][v];
  out[(uint64_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

// the product's register loop, X rows shared by `xGroup` list neighbours (one-wavefront workgroups, no pacing)
template <int WPS, int MS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS))) void probeRegister(double *out, char const *A, char const *X, uint64_t aStride, uint32_t aBytes,
                                                                                                 uint32_t xRows, uint32_t reps, uint32_t cols, uint32_t aWaves, uint32_t xGroup) {
  __shared__ uint32_t tab[PROBE_COLS + PROBE_PAD];
  uint32_t const lane = threadIdx.x;
  uint32_t const li = lane & 15, lk = lane >> 4;
  uint32_t const wave = blockIdx.x % aWaves;
  uint32_t const row0 = ((blockIdx.x / xGroup) * 2654435761u) % xRows;
  for (uint32_t j = lane; j < cols + PROBE_PAD; j += 64) tab[j] = j < cols ? ((row0 + j) % xRows) * 1024u : (row0 % xRows) * 1024u;
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
  for (uint32_t r = 0; r < reps; ++r) bfMfmaSegment<4, MS>(acc, sg, tab, lk);
  double sum = 0;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) sum += acc[g][m][t][v];
  out[(uint64_t)blockIdx.x * 64 + threadIdx.x] = sum;
}

static void report(char const *name, float ms, double waves, uint32_t reps) {
  double const ksteps = waves * reps * (PROBE_COLS / 4);
  double const mfmaFlops = ksteps * 3 * 2 * 4 * 2048.0;
  printf("\"%s\": {\"mfma_tflops_issued\": %.2f, \"frac_of_78.6\": %.3f, \"algorithmic_frac_x4_3\": %.3f, \"ms\": %.1f},\n", name, mfmaFlops / ms / 1e9, mfmaFlops / ms / 1e9 / 78.6,
         mfmaFlops / ms / 1e9 / 78.6 * 4.0 / 3.0, ms);
  fflush(stdout);
}

int main(int argc, char **argv) {
  uint32_t const reps = argc > 1 ? (uint32_t)atoi(argv[1]) : 600;
  uint32_t const aBytes = (PROBE_COLS / 4) * 2048u;
  size_t const aTotal = (size_t)aBytes * 2048;
  double *out; char *A, *X1g;
  CHECK(hipMalloc(&out, 2048 * 64 * 8));
  CHECK(hipMalloc(&A, aTotal + 4096));
  CHECK(hipMalloc(&X1g, (size_t)1 << 30));
  {
    std::vector<double> h((size_t)(64u << 20) / 8);
    uint64_t s = 88172645463325252ull;
    for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(int64_t)s / 9.3e18; }
    for (size_t off = 0; off < aTotal; off += h.size() * 8) CHECK(hipMemcpy(A + off, h.data(), (aTotal - off < h.size() * 8 ? aTotal - off : h.size() * 8), hipMemcpyHostToDevice));
    for (size_t off = 0; off < ((size_t)1 << 30); off += (size_t)64 << 20) CHECK(hipMemcpy(X1g + off, h.data(), (size_t)64 << 20, hipMemcpyHostToDevice));
  }
  printf("{\n");
  // the two loops must agree bit for bit: same rows (xGroup = 4 <-> one workgroup of 4), same leaf windows, ragged column count
  int bad = 0;
  // (a bundle of cnt < 4: the register loop's wavefronts 4 b + w, w >= cnt, have no counterpart and are not compared)
#define COMPARE(MSV) \
  for (uint32_t cnt : {4u, 3u, 2u}) \
    for (uint32_t cols : {1792u, 1787u, 61u, 5u}) { \
      size_t const nOut = (size_t)64 * 64; \
      std::vector<double> h0(nOut), h1(nOut, 0.0); \
      uint32_t const ab = 16u * MSV * cols * 16u; \
      probeRegister<2, MSV><<<64, 64>>>(out, A, X1g, aBytes, ab, 1048576, 2, cols, 2048, 4); \
      CHECK(hipDeviceSynchronize()); \
      CHECK(hipMemcpy(h0.data(), out, nOut * 8, hipMemcpyDeviceToHost)); \
      CHECK(hipMemset(out, 0, nOut * 8)); \
      probeSharedX<2, MSV><<<16, 256>>>(out, A, X1g, aBytes, ab, 1048576, 2, cols, 2048, cnt); \
      CHECK(hipDeviceSynchronize()); \
      CHECK(hipMemcpy(h1.data(), out, nOut * 8, hipMemcpyDeviceToHost)); \
      size_t diff = 0; \
      for (size_t i = 0; i < nOut; ++i) if ((i / 64) % 4 < cnt) diff += memcmp(&h0[i], &h1[i], 8) != 0; \
      printf("\"compare_ms%d_cnt%u_cols%u\": {\"different\": %zu},\n", MSV, cnt, cols, diff); \
      bad |= diff != 0; \
    }
  COMPARE(2)
  COMPARE(1)
#undef COMPARE
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms;
  int const waves = 256 * 4 * 2;
  for (uint32_t xg : {1u, 4u}) {
    probeRegister<2, 2><<<waves, 64>>>(out, A, X1g, aBytes, aBytes, 1048576, 8, PROBE_COLS, 2048, xg);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probeRegister<2, 2><<<waves, 64>>>(out, A, X1g, aBytes, aBytes, 1048576, reps, PROBE_COLS, 2048, xg);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    report(xg == 1 ? "register_loop_x_private_to_a_wavefront_1gb" : "register_loop_x_shared_by_4_neighbours_1gb", ms, waves, reps);
  }
  probeSharedX<2, 2><<<waves / 4, 256>>>(out, A, X1g, aBytes, aBytes, 1048576, 8, PROBE_COLS, 2048, 4);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  probeSharedX<2, 2><<<waves / 4, 256>>>(out, A, X1g, aBytes, aBytes, 1048576, reps, PROBE_COLS, 2048, 4);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  report("shared_x_loop_4_wavefronts_per_workgroup_1gb", ms, waves, reps);
  printf("\"loops_differ\": %d\n}\n", bad);
  return bad;
}
