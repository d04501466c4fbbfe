// Device code shared by bfhip_device.hip and bfhip_persist.hip (the persistent launch of the complex128 stage kernel is
// its own translation unit: it is compiled without hipcc's atomic optimizer, see the Makefile).  Not a public header.
#ifndef BFHIP_STAGE_C128_H
#define BFHIP_STAGE_C128_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfhip_internal.h"

#define BF_XCAP 256            /* must equal BfPlan.xcap */

struct StageParams {
  void const *arena;
  BfDevItem const *items;
  BfDevPiece const *pieces;
  uint32_t numItems;
  uint32_t nrhs;
  uint32_t coopItems;    // transposed: items [0, coopItems) get a workgroup each, its 4 wavefronts share the pieces
  uint32_t numBundles;   // forward complex128, 64-RHS kernel: workgroups (bundles[numBundles + 1] = first item of each, then numItems)
  void const *x;
  void *y;
  void *temp;
  void const *zero;   // >= 1 KiB of zeros (X fragments of out-of-range columns)
  uint32_t const *bundles;
};

// Leaf data is read exactly once per apply: stream it with the non-temporal
// policy so it does not displace the (re-read) vectors from L2 / Infinity Cache.
#ifndef BF_STREAM_NT
#define BF_STREAM_NT 1
#endif
typedef double bf_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 bfLoadStream(double2 const *p) {
#if BF_STREAM_NT
  bf_d2 v = __builtin_nontemporal_load((bf_d2 const *)p);
  return make_double2(v.x, v.y);
#else
  return *p;
#endif
}


__device__ __forceinline__ void waveSync() {
  // LDS traffic of one wave is issued in order; this only stops the compiler
  // from moving LDS accesses across the hand-off between lanes.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


#ifndef BF_C128_UNROLL
#define BF_C128_UNROLL 12       /* loads of a contraction issued together.  8 until late in round 4 (78 VGPRs, 6 wavefronts per SIMD once the
                                 * ragged end was batched); 10 / 12 / 14 / 16 (87 - 120 VGPRs, 5 or 4 wavefronts) alternated with it on two
                                 * boxes: headline -0.2 %, N = 65536 -1.2 ... -2.4 %, slowest 1/8 shard -0.7 ... -1.5 %, no clear order among them */
#endif
// Wavefronts per workgroup of this kernel.  A wavefront slot is refilled only when a whole workgroup's worth of slots is
// free on its CU: with 4-wavefront workgroups of unequal items, per-item timelines (tools/timeline.py) show 10 - 15 % of
// the slots empty through the second half of a short launch -- 3 - 5 us between the end of an item and the start of its
// replacement -- and this kernel's bandwidth is proportional to the wavefronts that stream.
#ifndef BF_C128_WG_WAVES
#define BF_C128_WG_WAVES 1
#endif
// (TL: the diagnostic instantiation records when every item started and ended, BFHIP_TIMELINE_FILE in bfdevLaunchStage)
// Index tables are read-only for the whole launch: loaded through the constant address space so that a wave-uniform
// load stays a scalar load (s_load, scalar cache) even in a kernel that stores and draws tickets before it -- after a
// store that may alias, hipcc otherwise turns every such load into a vector load + readfirstlane, a trip through the
// vector memory pipe that is busy streaming.  (Only there: in the one-item kernel the loads are scalar anyway, and with
// this form hipcc schedules its column loop for 54 VGPRs instead of 88 -- 8 wavefronts per SIMD with fewer loads in
// flight each, 20 - 60 % slower.)
template <typename T>
__device__ __forceinline__ T bfConstLoad(T const *q) {
  static_assert(sizeof(T) % 4 == 0, "dword-sized records");
  typedef uint32_t const __attribute__((address_space(4))) *CP;
  CP const w = (CP)(uintptr_t)q;
  union { T v; uint32_t u[sizeof(T) / 4]; } r;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) r.u[k] = w[k];
  return r.v;
}

typedef unsigned int bf_u4 __attribute__((ext_vector_type(4)));

// a window of 64 piece descriptors held one per lane: a single vector load replaces one
// dependent scalar load (an L2/HBM round trip) per piece; fields are broadcast by readlane
struct BfPieceWin { uint32_t w[6]; };
static_assert(sizeof(BfDevPiece) == 24, "BfDevPiece is 6 dwords");
__device__ __forceinline__ BfPieceWin bfPieceWinLoad(BfDevPiece const *pieces, uint32_t n, int lane) {
  BfPieceWin win;
  uint2 const *src = (uint2 const *)(pieces + (lane < (int)n ? lane : 0));
  uint2 a = src[0], b = src[1], c = src[2];
  win.w[0] = a.x; win.w[1] = a.y; win.w[2] = b.x; win.w[3] = b.y; win.w[4] = c.x; win.w[5] = c.y;
  return win;
}
__device__ __forceinline__ BfDevPiece bfPieceWinGet(BfPieceWin const &win, uint32_t i) {
  BfDevPiece pc;
  pc.dataOff = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)win.w[0], (int)i) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)win.w[1], (int)i) << 32);
  pc.inOff = (uint32_t)__builtin_amdgcn_readlane((int)win.w[2], (int)i);
  pc.ncols = (uint32_t)__builtin_amdgcn_readlane((int)win.w[3], (int)i);
  pc.flags = (uint32_t)__builtin_amdgcn_readlane((int)win.w[4], (int)i);
  pc.ld = (uint32_t)__builtin_amdgcn_readlane((int)win.w[5], (int)i);
  return pc;
}

struct BfNoHook { __device__ __forceinline__ void operator()() const {} };

// One item: `it` is its record (index `item`: only for the timeline).  `midHook` runs once, after the first dense piece has
// streamed (or at the end of an item without one): the persistent launch reads its ticket and requests the next record there.
template <bool TL, bool PERSIST, typename Hook>
__device__ __forceinline__ void bfItemC128(StageParams const &p, BfDevItem const &it, uint32_t const item, double2 *xs, int const lane, uint64_t *timeline, Hook &&midHook) {
  // diagnostic timestamps (TL only): 0 start, 1 first descriptor here, 2 first x gathered, 3 first piece streamed, 4 all pieces done, 5 end
  uint64_t tl[6] = {0, 0, 0, 0, 0, 0};
  if (TL) tl[0] = wall_clock64();
  bool hooked = false;
  uint32_t const mr = it.mrFlags & 0xffffu;
  uint32_t const g = 64u / mr;
  uint32_t const G = g * mr;
  bool const active = (uint32_t)lane < G;
  uint32_t const lc = active ? (uint32_t)lane : G - 1;   // clamped lane: inactive lanes recompute the last slot
  uint32_t const c = lc / mr;
  uint32_t const r = lc - c * mr;
  double2 const *arena = (double2 const *)p.arena;
  uint32_t const nrhs = p.nrhs;
  double2 *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (double2 *)p.y : (double2 *)p.temp;

  // (Measured and set aside, round 3: the descriptor of piece pi + 1 requested while piece pi streams, and x gathered
  // first with the first 3 - 4 loads of the piece already in flight behind it -- ISA as intended, same times at N = 65536,
  // on a 1/8 shard and at the headline size: the round trips at the head of a piece are not what the short launches lose.)
  for (uint32_t q = 0; q < nrhs; ++q) {
    double accr = 0.0, acci = 0.0;
    for (uint32_t pi = 0; pi < it.numPieces; ++pi) {
      BfDevPiece const pc = PERSIST ? bfConstLoad(p.pieces + it.pieceBegin + pi) : p.pieces[it.pieceBegin + pi];
      double2 const *xin = (pc.flags & BF_PIECE_IN_X) ? (double2 const *)p.x : (double2 const *)p.temp;
      xin += (uint64_t)pc.inOff * nrhs + q;
      uint32_t const n = pc.ncols;
      if (TL && !tl[1]) { __builtin_amdgcn_s_waitcnt(0); tl[1] = wall_clock64() + (n == 0xffffffffu); }
      if (pc.flags & BF_PIECE_IDENTITY) {
        if (c == 0 && active) {
          double2 v = xin[(uint64_t)r * nrhs];
          accr += v.x; acci += v.y;
        }
        continue;
      }
      // gather the input sub-vector into LDS
      waveSync();   // previous piece's reads are done before overwriting
      for (uint32_t j = lane; j < n; j += 64) xs[j] = xin[(uint64_t)j * nrhs];
      waveSync();
      if (TL && !tl[2]) { __builtin_amdgcn_s_waitcnt(0); tl[2] = wall_clock64(); }
      double2 const *ap = arena + pc.dataOff + lc;
      uint32_t const nfull = n / g;
      uint32_t j = c;
      uint32_t s = 0;
      // whole groups of BF_C128_UNROLL steps: their loads are issued together ...
      double2 a[BF_C128_UNROLL];
#pragma unroll 1
      for (; s + BF_C128_UNROLL <= nfull; s += BF_C128_UNROLL) {
#pragma unroll
        for (int k = 0; k < BF_C128_UNROLL; ++k) a[k] = bfLoadStream(ap + (uint64_t)(s + k) * G);
#pragma unroll
        for (int k = 0; k < BF_C128_UNROLL; ++k) {
          double2 const xv = xs[j];
          accr = fma(a[k].x, xv.x, accr); accr = fma(-a[k].y, xv.y, accr);
          acci = fma(a[k].x, xv.y, acci); acci = fma(a[k].y, xv.x, acci);
          j += g;
        }
      }
      // ... and so are those of the ragged end (left to the compiler's remainder loop every one of its up to 7 steps waited
      // for its own load: a memory round trip each, on pieces of 60 - 250 columns -- N = 65536, row shards -- a tenth of an item)
      if (s < nfull) {                                           // wave-uniform
        uint32_t const left = nfull - s;                         // 1 .. BF_C128_UNROLL - 1
#pragma unroll
        for (int k = 0; k < BF_C128_UNROLL - 1; ++k)
          if ((uint32_t)k < left) a[k] = bfLoadStream(ap + (uint64_t)(s + k) * G);
#pragma unroll
        for (int k = 0; k < BF_C128_UNROLL - 1; ++k)
          if ((uint32_t)k < left) {
            double2 const xv = xs[j];
            accr = fma(a[k].x, xv.x, accr); accr = fma(-a[k].y, xv.y, accr);
            acci = fma(a[k].x, xv.y, acci); acci = fma(a[k].y, xv.x, acci);
            j += g;
          }
      }
      uint32_t const rem = n - nfull * g;
      if (active && c < rem) {
        double2 a = bfLoadStream(ap + (uint64_t)nfull * G);
        double2 xv = xs[j];
        accr = fma(a.x, xv.x, accr); accr = fma(-a.y, xv.y, accr);
        acci = fma(a.x, xv.y, acci); acci = fma(a.y, xv.x, acci);
      }
      if (TL && !tl[3]) { __builtin_amdgcn_s_waitcnt(0); tl[3] = wall_clock64(); }
      if (PERSIST && !hooked) { midHook(); hooked = true; }      // wave-uniform
    }
    if (TL && !tl[4]) { __builtin_amdgcn_s_waitcnt(0); tl[4] = wall_clock64(); }
    // combine the g column groups (fixed order) and store
    waveSync();
    xs[lane] = make_double2(accr, acci);
    waveSync();
    if ((uint32_t)lane < mr) {
      double sr = 0.0, si = 0.0;
      for (uint32_t cc = 0; cc < g; ++cc) { double2 v = xs[cc * mr + lane]; sr += v.x; si += v.y; }
      out[((uint64_t)it.outOff + lane) * nrhs + q] = make_double2(sr, si);
    }
  }
  if (PERSIST && !hooked) midHook();
  if (TL) {
    __builtin_amdgcn_s_waitcnt(0);       // the item's stores have been issued and its loads have returned
    tl[5] = wall_clock64();
    uint64_t const v = lane == 0 ? tl[0] : lane == 1 ? tl[1] : lane == 2 ? tl[2] : lane == 3 ? tl[3] : lane == 4 ? tl[4] : tl[5];
    if (lane < 6) timeline[8 * (uint64_t)item + lane] = v;
  }
}

#endif
