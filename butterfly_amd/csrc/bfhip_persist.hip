// Persistent launch of the complex128 stage kernel (experimental, BFHIP_PERSISTENT=1), for a stage with more items than
// the chip has wavefront slots.
//
// With one item per wavefront a slot stands empty for 3 - 5 us between the end of an item and the start of the workgroup
// that replaces it (per-item timelines, tools/timeline.py: 5 - 10 % of the slots through the second half of a short
// launch), and the stage kernel's bandwidth is proportional to the wavefronts that stream.  Here the grid is the chip's
// slots (one-wavefront workgroups); wavefront w takes item w, then draws tickets for the items past the grid.
//
// Tickets.  One counter would be the flow launch's mistake again (same-address atomics retire at ~19 ns device-wide):
// there are BF_TICKET_POOLS counters, 256 bytes apart (packed into two cache lines they behaved like one counter: the
// first items of a launch took 25 % longer), pool p owning items grid + p, grid + p + POOLS, ... (the list is sorted
// big-first, so every pool sees the same sizes) and drawn from by the wavefronts (w >> 3) % POOLS == p -- eight consecutive
// workgroups, one per XCD.  Nothing depends on which workgroups are resident.  A launch draws exactly valid(p) +
// wavefronts(p) tickets from pool p (every wavefront ends on one failed draw); whoever draws the last one puts the counter
// back to zero for the next launch: no host state, no memset, safe to replay from a captured graph.
//
// The draw must not be waited for where it is issued.  hipcc's atomic optimizer rewrites a wave-uniform atomic as a wave
// reduction + one atomic + a broadcast that needs the result at once (first version: the ticket's round trip simply took
// the place of the dispatch gap, 1.135 ms against 1.08 ms plain at N = 65536).  This file is therefore compiled with the
// optimizer off (Makefile: -mllvm -amdgpu-atomic-optimizer-strategy=None) and lane 0 alone issues the atomic under an
// ordinary divergent branch: one request, tracked by the compiler like any load and waited for where its result is read --
// after the item's first dense piece has streamed, when the atomic (older than that piece's loads) has long returned.  The
// next item's record is requested there and arrives under the rest of the item.  Loop control only ever sees wave-uniform
// values (readfirstlane), so the lane branch cannot split the loop (the flow launch's first lesson).
// Same items, same arithmetic: bit-identical to the plain launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"
#include "bfhip_stage_c128.h"

template <bool TL>
__device__ __forceinline__ void bfStageBodyC128P(StageParams const &p, double2 *xs, uint32_t *tickets, uint64_t *timeline) {
  int const lane = threadIdx.x & 63;
  uint32_t const w = blockIdx.x, G = gridDim.x;           // G: a multiple of 8 * BF_TICKET_POOLS, < numItems
  uint32_t const pool = (w >> 3) % BF_TICKET_POOLS;
  uint32_t const dyn = p.numItems - G;
  uint32_t const valid = dyn / BF_TICKET_POOLS + (pool < dyn % BF_TICKET_POOLS ? 1u : 0u);
  uint32_t const draws = valid + G / BF_TICKET_POOLS;      // of this pool, per launch
  uint32_t *const counter = tickets + pool * BF_TICKET_STRIDE;
  uint32_t item = w;
  BfDevItem it = bfConstLoad(p.items + item);
  uint32_t t = 0;
  for (;;) {
    uint32_t tk = 0;
    if (lane == 0) tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // in flight
    uint32_t next = 0;
    BfDevItem itN = it;
    auto hook = [&]() {
      t = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
      if (t < valid) {
        next = G + t * BF_TICKET_POOLS + pool;
        itN = bfConstLoad(p.items + next);
      }
    };
    bfItemC128<TL, true>(p, it, item, xs, lane, timeline, hook);
    if (t >= valid) break;
    item = next;
    it = itN;
  }
  if (t + 1 == draws) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the pool's last draw of this launch
}

#ifndef BF_C128_WAVES
#define BF_C128_WAVES 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BF_C128_WAVES, BF_C128_WAVES))) void bfStageKernelC128P(StageParams p, uint32_t *tickets) {
  __shared__ __attribute__((aligned(16))) double2 lds[BF_XCAP];
  bfStageBodyC128P<false>(p, lds, tickets, nullptr);
}
__global__ __launch_bounds__(64) void bfStageKernelC128PTimeline(StageParams p, uint32_t *tickets, uint64_t *timeline) {
  __shared__ __attribute__((aligned(16))) double2 lds[BF_XCAP];
  bfStageBodyC128P<true>(p, lds, tickets, timeline);
}

// wavefront slots of the device for this kernel (5 per SIMD), rounded down to what the ticket pools need; 0 unless
// BFHIP_PERSISTENT=1
extern "C" uint32_t bfdevPersistentGrid(void) {
  static int enabled = -1;                         // the environment is read once
  static int cached[16];                           // per device ordinal: CU counts may differ (0: not asked yet)
  if (enabled < 0) { char const *e = getenv("BFHIP_PERSISTENT"); enabled = e && e[0] == '1'; }
  if (!enabled) return 0;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  if (!cached[dev]) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bfStageKernelC128P, 64, 0) != hipSuccess || occ <= 0) occ = 20;
    uint64_t const slots = (uint64_t)cus * (uint64_t)occ;
    cached[dev] = (int)(slots / (8u * BF_TICKET_POOLS) * (8u * BF_TICKET_POOLS));
  }
  return (uint32_t)cached[dev];
}

// stageParams: the StageParams of bfdevLaunchStage; grid from bfdevPersistentGrid, < numItems; timeline NULL or
// 8 x numItems uint64 (64 bytes per item: bfItemC128<TL> stamps timeline[8 * item + 0 .. 5])
extern "C" int bfdevLaunchPersistC128(void const *stageParams, uint32_t grid, void *tickets, void *timeline, void *stream) {
  StageParams const &p = *(StageParams const *)stageParams;
  if (!grid || grid >= p.numItems || grid % (8u * BF_TICKET_POOLS) || !tickets) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "persistent launch: bad grid");
  if (timeline) hipLaunchKernelGGL(bfStageKernelC128PTimeline, dim3(grid), dim3(64), 0, (hipStream_t)stream, p, (uint32_t *)tickets, (uint64_t *)timeline);
  else hipLaunchKernelGGL(bfStageKernelC128P, dim3(grid), dim3(64), 0, (hipStream_t)stream, p, (uint32_t *)tickets);
  hipError_t const e = hipGetLastError();
  return e == hipSuccess ? 0 : bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "persistent stage launch: %s", hipGetErrorString(e));
}
