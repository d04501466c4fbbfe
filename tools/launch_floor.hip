// What a launch costs on top of its bytes: chains of back-to-back launches of a plain streaming kernel (every wavefront reads
// its own contiguous `itemBytes` with 16-byte non-temporal loads, 8 in flight per lane -- the stage kernels' pattern with
// nothing else in the way) at the sizes of the short stage kernels.  Fitting t = bytes / BW + c over the sizes gives the cost
// c of a kernel boundary that no staged executor can avoid (DESIGN_EXPERIMENTS.md section 15).
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void readStream(u4 const *src, uint64_t unitsPerWave, unsigned *sink) {
  uint64_t const wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  u4 const *p = src + wave * unitsPerWave + (threadIdx.x & 63);
  unsigned acc = 0;
  for (uint64_t i = 0; i < unitsPerWave; i += 64 * 8) {
    u4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + 64 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

int main() {
  uint64_t const total = 48ull << 30;
  void *a = nullptr; unsigned *sink = nullptr;
  if (hipMalloc(&a, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc((void **)&sink, 4);
  hipMemset(a, 0, total);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("{\"rows\": [");
  bool first = true;
  for (uint64_t itemKiB : {32ull, 64ull, 128ull}) {
    for (double gb : {0.125, 0.25, 0.5, 1.0, 2.0, 4.0, 10.0}) {
      uint64_t const unitsPerWave = itemKiB * 1024 / 16;
      uint64_t const waves = (uint64_t)(gb * 1e9 / (itemKiB * 1024)) / 4 * 4;
      uint64_t const bytes = waves * itemKiB * 1024;
      int const chain = (int)(total / bytes) < 64 ? (int)(total / bytes) : 64;       // every launch of a chain reads its own memory
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < chain; ++k) readStream<<<(unsigned)(waves / 4), 256>>>((u4 const *)((char *)a + (uint64_t)k * bytes), unitsPerWave, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / chain < best) best = ms / chain;
      }
      printf("%s{\"item_kib\": %llu, \"launch_gb\": %.4f, \"waves\": %llu, \"chain\": %d, \"us_per_launch\": %.2f, \"gbs\": %.1f}", first ? "" : ", ",
             (unsigned long long)itemKiB, bytes / 1e9, (unsigned long long)waves, chain, best * 1e3, bytes / (best * 1e-3) / 1e9);
      first = false;
    }
  }
  printf("]}\n");
  return 0;
}
