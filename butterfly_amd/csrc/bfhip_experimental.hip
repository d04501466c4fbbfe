// bfhip_experimental.hip -- executors of the complex128 plan that were built, measured and set aside
// (DESIGN_EXPERIMENTS.md section 15).  NOT part of the default library: `make experimental` builds libbfhip_exp.so from the same sources plus this
// file and bfhip_persist.hip with -DBFHIP_EXPERIMENTAL; the default libbfhip.so holds no spin-waiting kernel and no
// environment-switched executor.  Both executors are tested bit-identical to the staged launches
// (tests/experimental_checks.py, run by the GPU suite against the flagged build).
//   * bfFlowKernelC128: the whole plan as ONE dependency-driven persistent launch (BFHIP_FLAG_FLOW / BFHIP_FLOW=1)
//   * bfStageKernelC128Timeline + bfTimelineLaunch: per-item timestamps of a stage launch (BFHIP_TIMELINE_FILE)
//   * bfdevLaunchStageExperimental: the hook of bfdevLaunchStage that sends a complex128 stage to the timeline launch or
//     to the persistent ticket launch of bfhip_persist.hip (BFHIP_PERSISTENT=1)
#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bfhip_internal.h"
#include "bfhip_stage_c128.h"
#include "../../include/bfhip_abi.h"

#define BF_WAVES_PER_WG 4

static int hipFail(hipError_t e, char const *what) {
  if (e == hipSuccess) return 0;
  int code = (e == hipErrorOutOfMemory) ? BFABI_ERROR_MEMORY_ERROR : BFABI_ERROR_RUNTIME_ERROR;
  return bfhipFail(code, "%s: %s", what, hipGetErrorString(e));
}

// the stage kernel's body with per-item timestamps (tools/timeline.py)
__global__ __launch_bounds__(BF_C128_WG_WAVES * 64) void bfStageKernelC128Timeline(StageParams p, uint64_t *timeline) {
  __shared__ __attribute__((aligned(16))) double2 lds[BF_C128_WG_WAVES][BF_XCAP];
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  uint32_t const item = __builtin_amdgcn_readfirstlane(blockIdx.x * BF_C128_WG_WAVES + wave);
  if (item >= p.numItems) return;
  BfDevItem const it = p.items[item];
  bfItemC128<true, false>(p, it, item, lds[wave], lane, timeline, BfNoHook());
}

// ---------------------------------------------------------------------------
// Dependency-driven launch of a whole complex128 plan ("flow"): ONE persistent
// launch instead of one per stage.
//
// Stage launches synchronise far more than the butterfly needs: an item of stage
// s + 1 reads the intermediate vectors of ITS product only (one per factor boundary,
// src/mat_product.c:225-238), yet a kernel boundary makes it wait for every item of
// stage s -- each launch pays its own ramp-up and its own tail (the last big items
// run with the chip mostly empty).  At 10 GB per launch that is 2 %; at N = 65536 or
// on a 1/8 shard the kernels are 0.13 - 0.5 ms and it is 10 %.
//
// Here the items of all stages form one list in stage order (big first inside a
// stage); resident wavefronts take the next item from an atomic ticket.  Every
// intermediate vector has a counter of the items that have written it: a piece that
// reads vector b waits until counter[b] has reached the number of b's writers, an
// item that has stored its rows releases them (agent-scope release: its stores reach
// memory before the counter moves) and bumps the counter of the vector it wrote.
// A ticket holder only ever waits for items with smaller tickets, which are held by
// wavefronts already running and waiting, in turn, only for smaller ones: the item
// with the smallest unfinished ticket never waits, so the list drains -- whatever
// the order in which workgroups are dispatched.  Counters are not reset between
// applies: apply number e waits for e * writers (the host resets them long before
// 32 bits wrap).  The arithmetic of an item is that of bfStageKernelC128, operation
// for operation: results are bit-identical to the staged launches.
// ---------------------------------------------------------------------------
struct FlowParams {
  void const *arena;
  BfDevItem const *items;        // all stages, pieceBegin global
  BfDevPiece const *pieces;      // all stages; `ld` = id of the vector the piece reads (0: x)
  uint32_t const *itemOut;       // per item: id of the intermediate it writes (0: y / a private slot: nobody waits)
  uint32_t const *writers;       // per vector id: number of items that write it
  uint32_t *counters;            // per vector id: items that have written it, summed over applies
  uint32_t *queue;               // ticket counter (counters[-1] in memory)
  uint32_t *error;               // set if a wait gave up (never, by construction)
  uint32_t numItems, nrhs;
  uint32_t epoch;                // 1-based apply number since the counters were last cleared
  uint32_t queueBase;            // ticket value of item 0 in this apply
  uint32_t spinLimit;            // a wait gives up after this many polls
  uint32_t debugMode;            // experiments only (BFHIP_FLOW_DEBUGMODE): 1 plain loads / stores for the intermediates, 2 never wait (results then wrong)
  void const *x;
  void *y;
  void *temp;
};

// -DBF_FLOW_STATIC=1: no ticket counter, wavefront slot w takes items w, w + S, ... (round 4).  Measured on one box
// (N = 65536 / rows shard 3 of 8 / headline): static 1.109 / 1.90 / 10.70 ms, tickets two at a time 1.51 / 2.66 / 11.18,
// staged launches 1.098 / 1.54 / 10.60; with the inner loop unrolled 8x at 4 wavefronts per SIMD 1.122 / 1.85 / 10.59.
// Without its tickets the one-launch executor EQUALS the staged launches where items are small and loses where a few
// 1 MiB items live as long as a launch: the stage boundaries it removes were not what short launches lose.
#ifndef BF_FLOW_STATIC
#define BF_FLOW_STATIC 0
#endif
#ifndef BF_FLOW_BATCH
#define BF_FLOW_BATCH 2u      /* measured at N = 65536: 1 -> 1.56 ms, 2 -> 1.36, 4 -> 1.64, 8 -> 2.26 (staged launches: 1.09) */
#endif
#define BF_FLOW_SPIN_LIMIT (1u << 21)     // x ~0.5 us: about a second (a legitimate wait is micro- to milliseconds), then the wait gives up and raises `error` instead of hanging the GPU

// Nothing in the launch branches on "lane == 0": a lane-invariant condition inside the item loop invites the compiler to
// give lane 0 and the other lanes loops of their own (the first version did exactly that: lanes 1..63 went round again
// with ticket 0 while lane 0 drew the next one -- the wavefront never came back together and the launch never ended).
// Tickets and counter updates are single-lane atomics issued with the exec mask narrowed inside one asm block; waits are
// polled by all lanes at once (one request: same address) and decided on the broadcast value.
__device__ __forceinline__ uint32_t bfFlowTicket(uint32_t *queue) {
  uint32_t ret;
  uint64_t saved;
  asm volatile("s_mov_b64 %1, exec\n\t"
               "s_mov_b64 exec, 1\n\t"
               "global_atomic_add %0, %2, %3, %4 sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, %1"
               : "=&v"(ret), "=&s"(saved)
               : "v"(0u), "v"(1u), "s"(queue)
               : "memory");
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)ret);      // lane 0 is active: the loop top is reached by whole wavefronts
}
__device__ __forceinline__ void bfFlowBump(uint32_t *counter) {
  uint64_t saved;
  // Ordering: there is NO fence before this call.  The item's rows were written with agent-scope (sc1) stores -- written
  // through this XCD's L2 -- and vmcnt counts them until they are acknowledged there; the counter moves only after
  // vmcnt has drained (the wait below), and readers use agent-scope loads.  That is the whole argument.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
               "s_mov_b64 %0, exec\n\t"
               "s_mov_b64 exec, 1\n\t"
               "global_atomic_add %1, %2, %3\n\t"
               "s_mov_b64 exec, %0"
               : "=&s"(saved)
               : "v"(0u), "v"(1u), "s"(counter)
               : "memory");
}

// Intermediate vectors are the only data that crosses between wavefronts inside the launch.  They are written with
// agent-scope stores (sc1: written through this XCD's L2) and read with agent-scope loads (sc1: never served from a
// stale line), so no cache-wide writeback / invalidate is needed around an item -- the first version fenced with
// buffer_wbl2 / buffer_inv per item and per wait, and an operator whose last-stage items read a dozen vectors each ran
// 2.6x slower than the staged launches.  Leaf data, x and the index tables are read-only: plain (streamed) loads.
__device__ __forceinline__ double2 bfLoadCoherent(double2 const *p) {
  unsigned long long const *q = (unsigned long long const *)p;
  unsigned long long const a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long const b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_double2(__longlong_as_double((long long)a), __longlong_as_double((long long)b));
}
__device__ __forceinline__ void bfStoreCoherent(double2 *p, double2 v) {
  unsigned long long *q = (unsigned long long *)p;
  __hip_atomic_store(q, (unsigned long long)__double_as_longlong(v.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, (unsigned long long)__double_as_longlong(v.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// poll until vector `dep` has all its writers (a legitimate wait is micro- to milliseconds; after spinLimit polls it
// gives up and raises `error` instead of hanging the GPU)
__device__ __forceinline__ void bfFlowWait(FlowParams const &p, uint32_t dep, uint32_t want) {
  uint32_t spins = 0;
  for (;;) {
    uint32_t const seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p.counters + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (seen >= want) break;
    __builtin_amdgcn_s_sleep(8);
    if (++spins > p.spinLimit) { __hip_atomic_store(p.error, dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }      // every lane stores the same word
  }
}

// The launch is software-pipelined one item ahead.  What an item costs on top of its streaming is a chain of dependent
// round trips -- ticket, item record, piece descriptors, counter polls -- that the staged kernels pay once per launch
// (as ramp) and a naive persistent kernel pays per item (measured: 1.56 ms against 1.09 ms at N = 65536).  Here the
// chain of item k + 1 runs underneath item k:
//   top of item k         the ticket drawn one item ago becomes item k + 1; its record is requested; the ticket of
//                         item k + 2 is drawn
//   piece 0 is gathered   the first 64 piece descriptors of item k + 1 are requested (one vector load, 24 bytes a lane)
//   piece 0 has streamed  they go to the wavefront's second LDS window; the counters of the vectors they read are polled
//   item k is stored      the poll answers say which of them still have to be waited for (none, in the steady state)
// Piece descriptors then come from LDS (broadcast reads), not from a scalar load per piece.  Tickets are drawn with a
// data-dependent increment (lane 0 adds 1, the others 0: the compiler's atomic optimizer turns that into one
// single-lane atomic) so that nothing in the loop branches on the lane number.
struct BfFlowWin { uint2 a[64], b[64], c[64]; };     // 64 piece descriptors, 8-byte thirds apart (conflict-free lane-wise writes)

__device__ __forceinline__ BfDevPiece bfFlowWinGet(BfFlowWin const *w, uint32_t i) {
  uint2 const a = w->a[i], b = w->b[i], c = w->c[i];            // same address in every lane: broadcast
  BfDevPiece pc;
  pc.dataOff = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.x) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.y) << 32);
  pc.inOff = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.x);
  pc.ncols = (uint32_t)__builtin_amdgcn_readfirstlane((int)b.y);
  pc.flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.x);
  pc.ld = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.y);
  return pc;
}

#ifndef BF_FLOW_UNROLL
#define BF_FLOW_UNROLL 4
#endif
#ifndef BF_FLOW_WAVES
#define BF_FLOW_WAVES 5
#endif
// (the path is latency x concurrency bound: at 4 wavefronts per SIMD the stage kernel itself runs 1.40 ms instead of 1.06 at
// N = 65536, while 5, 6 or 8 wavefronts with a shorter unroll measure the same -- so the pipelined state is paid for with unroll
// depth, not with occupancy)
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) __attribute__((amdgpu_waves_per_eu(BF_FLOW_WAVES, BF_FLOW_WAVES))) void bfFlowKernelC128(FlowParams p) {
  __shared__ __attribute__((aligned(16))) double2 lds[BF_WAVES_PER_WG][BF_XCAP];
  __shared__ __attribute__((aligned(16))) BfFlowWin wins[BF_WAVES_PER_WG][2];
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  double2 *xs = lds[wave];
  double2 const *arena = (double2 const *)p.arena;
  uint32_t const nrhs = p.nrhs;
  // Tickets are drawn BF_FLOW_BATCH items at a time: every wavefront of the chip draws from ONE counter, and same-address
  // atomics retire at ~19 ns each device-wide -- with one ticket per item the launch takes items x 19 ns whatever else is
  // hidden (81 k items at N = 65536: 1.5 ms).  Larger batches trade that for imbalance (consecutive items of the big-first
  // list are of similar size).  A wavefront works through its batch in order: it still only ever waits for smaller items.
  uint32_t const one = lane == 0 ? BF_FLOW_BATCH : 0u;

  // ---- prologue: the first item, fetched the slow way
#if BF_FLOW_STATIC
  // static dealing: wavefront slot w of the resident grid takes items w, w + S, w + 2 S, ... (S = slots) of the stage-major,
  // big-first list -- no ticket counter at all (same-address atomics retire at ~19 ns each device-wide: 81 k items = 1.5 ms).
  // Still deadlock-free: every slot walks its items in list order, so the smallest unfinished item never waits.
  (void)one;
  uint32_t const slotStride = gridDim.x * BF_WAVES_PER_WG;
  uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * BF_WAVES_PER_WG + wave));
  if (cur >= p.numItems) return;
  uint32_t batchEnd = 0, tkNext = 0;
  (void)batchEnd; (void)tkNext;
#else
  uint32_t tk = __hip_atomic_fetch_add(p.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
  if (cur >= p.numItems) return;
  uint32_t batchEnd = cur + BF_FLOW_BATCH;
  uint32_t tkNext = __hip_atomic_fetch_add(p.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next batch, in flight
#endif
  BfDevItem it = p.items[cur];
  uint32_t od = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.itemOut[cur]);
  uint32_t cw = 0;
  uint64_t pending = 0;            // bit i: piece i of the current window reads a vector that was incomplete when polled
  {
    BfFlowWin *w = &wins[wave][0];
    uint32_t const np = it.numPieces < 64u ? it.numPieces : 64u;
    uint32_t seen = ~0u, want = 0;
    if ((uint32_t)lane < np) {
      uint2 const *src = (uint2 const *)(p.pieces + it.pieceBegin + lane);
      uint2 const a = src[0], b = src[1], c = src[2];
      w->a[lane] = a; w->b[lane] = b; w->c[lane] = c;
      if (c.y) { want = (c.x >> 8) * p.epoch; seen = __hip_atomic_load(p.counters + c.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    pending = __ballot(seen < want);
    waveSync();
  }

  for (;;) {
    // ---- top of item `cur`: the next item is the next of this batch, or the first of the batch drawn a batch ago
#if BF_FLOW_STATIC
    bool const lastOfBatch = false;
    uint32_t const nxt = cur + slotStride < cur ? 0xffffffffu : cur + slotStride;
#else
    bool const lastOfBatch = cur + 1 == batchEnd;
    uint32_t const nxt = lastOfBatch ? (uint32_t)__builtin_amdgcn_readfirstlane((int)tkNext) : cur + 1;
#endif
    bool const haveNext = nxt < p.numItems;
    BfDevItem itN = it;
    uint32_t odN = 0;
    if (haveNext) {
      itN = p.items[nxt];
      odN = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.itemOut[nxt]);
      if (lastOfBatch) tkNext = __hip_atomic_fetch_add(p.queue, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the batch after the next
    }
    uint2 wa = make_uint2(0, 0), wb = wa, wc = wa;       // next item's piece descriptors on their way to LDS
    uint32_t seenN = ~0u, wantN = 0;
    int stageN = haveNext ? 0 : 3;                       // 0: nothing requested, 1: descriptors requested, 2: in LDS + polled, 3: done / none

    BfFlowWin const *w = &wins[wave][cw];
    uint32_t const mr = it.mrFlags & 0xffffu;
    uint32_t const g = 64u / mr;
    uint32_t const G = g * mr;
    bool const active = (uint32_t)lane < G;
    uint32_t const lc = active ? (uint32_t)lane : G - 1;
    uint32_t const c = lc / mr;
    uint32_t const r = lc - c * mr;
    double2 *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (double2 *)p.y : (double2 *)p.temp;
    for (uint32_t q = 0; q < nrhs; ++q) {
      double accr = 0.0, acci = 0.0;
      for (uint32_t pi = 0; pi < it.numPieces; ++pi) {
        BfDevPiece pc;
        bool wait;
        if (pi < 64u) { pc = bfFlowWinGet(w, pi); wait = (pending >> pi) & 1u; }
        else { pc = p.pieces[it.pieceBegin + pi]; wait = pc.ld != 0; }       // (items of more than 64 pieces: the rest the slow way)
        uint32_t const dep = pc.ld;
        double2 const *xin = (pc.flags & BF_PIECE_IN_X) ? (double2 const *)p.x : (double2 const *)p.temp;
        xin += (uint64_t)pc.inOff * nrhs + q;
        uint32_t const n = pc.ncols;
        if (wait && !(p.debugMode & 2u)) bfFlowWait(p, dep, (pc.flags >> 8) * p.epoch);              // seen complete BEFORE the vector is asked for
        if (pc.flags & BF_PIECE_IDENTITY) {
          if (c == 0 && active) {
            double2 v = dep ? bfLoadCoherent(xin + (uint64_t)r * nrhs) : xin[(uint64_t)r * nrhs];
            accr += v.x; acci += v.y;
          }
          continue;
        }
        waveSync();   // previous piece's reads are done before overwriting
        if (dep && !(p.debugMode & 1u)) { for (uint32_t j = lane; j < n; j += 64) xs[j] = bfLoadCoherent(xin + (uint64_t)j * nrhs); }
        else { for (uint32_t j = lane; j < n; j += 64) xs[j] = xin[(uint64_t)j * nrhs]; }
        if (stageN == 0) {       // under this piece's stream: the next item's piece descriptors
          uint32_t const np = itN.numPieces < 64u ? itN.numPieces : 64u;
          if ((uint32_t)lane < np) { uint2 const *src = (uint2 const *)(p.pieces + itN.pieceBegin + lane); wa = src[0]; wb = src[1]; wc = src[2]; }
          stageN = 1;
        }
        waveSync();
        double2 const *ap = arena + pc.dataOff + lc;
        uint32_t const nfull = n / g;
        uint32_t j = c;
        uint32_t s = 0;
#pragma unroll BF_FLOW_UNROLL
        for (; s < nfull; ++s) {
          double2 a = bfLoadStream(ap + (uint64_t)s * G);
          double2 xv = xs[j];
          accr = fma(a.x, xv.x, accr); accr = fma(-a.y, xv.y, accr);
          acci = fma(a.x, xv.y, acci); acci = fma(a.y, xv.x, acci);
          j += g;
        }
        uint32_t const rem = n - nfull * g;
        if (active && c < rem) {
          double2 a = bfLoadStream(ap + (uint64_t)nfull * G);
          double2 xv = xs[j];
          accr = fma(a.x, xv.x, accr); accr = fma(-a.y, xv.y, accr);
          acci = fma(a.x, xv.y, acci); acci = fma(a.y, xv.x, acci);
        }
        if (stageN == 1) {       // the descriptors have arrived long ago: to the other window, and poll what they read
          BfFlowWin *wn = &wins[wave][cw ^ 1u];
          uint32_t const np = itN.numPieces < 64u ? itN.numPieces : 64u;
          if ((uint32_t)lane < np) {
            wn->a[lane] = wa; wn->b[lane] = wb; wn->c[lane] = wc;
            if (wc.y) { wantN = (wc.x >> 8) * p.epoch; seenN = __hip_atomic_load(p.counters + wc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          }
          stageN = 2;
        }
      }
      waveSync();
      xs[lane] = make_double2(accr, acci);
      waveSync();
      if ((uint32_t)lane < mr) {
        double sr = 0.0, si = 0.0;
        for (uint32_t cc = 0; cc < g; ++cc) { double2 v = xs[cc * mr + lane]; sr += v.x; si += v.y; }
        double2 *dst = out + ((uint64_t)it.outOff + lane) * nrhs + q;
        if (od && !(p.debugMode & 1u)) bfStoreCoherent(dst, make_double2(sr, si)); else *dst = make_double2(sr, si);      // later items read it / a later kernel does
      }
    }
    if (od) bfFlowBump(p.counters + od);      // waits for this wavefront's stores (vmcnt) before the counter moves
    if (!haveNext) return;
    // ---- items without a dense piece never reached the hooks: catch up
    if (stageN < 2) {
      BfFlowWin *wn = &wins[wave][cw ^ 1u];
      uint32_t const np = itN.numPieces < 64u ? itN.numPieces : 64u;
      if ((uint32_t)lane < np) {
        if (stageN == 0) { uint2 const *src = (uint2 const *)(p.pieces + itN.pieceBegin + lane); wa = src[0]; wb = src[1]; wc = src[2]; }
        wn->a[lane] = wa; wn->b[lane] = wb; wn->c[lane] = wc;
        if (wc.y) { wantN = (wc.x >> 8) * p.epoch; seenN = __hip_atomic_load(p.counters + wc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
    }
    pending = __ballot(seenN < wantN);
    waveSync();       // the other window is written: every lane may read it
    if (lastOfBatch) batchEnd = nxt + BF_FLOW_BATCH;
    it = itN; od = odN; cur = nxt; cw ^= 1u;
  }
}


extern "C" {
// Diagnostic (BFHIP_TIMELINE_FILE=path, complex128 stages with nrhs < 3): the launch is synchronous and appends, per launch,
// a header line "launch <numItems> <numPieces>" and per item "<start> <end> <rows> <columns summed over its dense pieces>
// <pieces> <first descriptor here> <first x gathered> <first piece streamed> <all pieces done>"
// (wall_clock64 ticks: 100 MHz) to the file -- where inside a launch the time goes (tools/timeline.py).
static int bfTimelineLaunch(BfLaunchArgs const *a, StageParams const &p, uint32_t grid, uint32_t pgrid, hipStream_t s) {
  uint64_t const n = a->numItems;
  uint64_t *dev = nullptr;
  if (hipMalloc((void **)&dev, (n ? n : 1) * 64) != hipSuccess) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "timeline buffer");
  (void)hipMemsetAsync(dev, 0, (n ? n : 1) * 64, s);
  if (pgrid) (void)bfdevLaunchPersistC128(&p, pgrid, a->tickets, dev, s);
  else hipLaunchKernelGGL(bfStageKernelC128Timeline, dim3(grid), dim3(BF_C128_WG_WAVES * 64), 0, s, p, dev);
  int rc = hipFail(hipStreamSynchronize(s), "timeline launch");
  if (!rc && n) {
    std::vector<uint64_t> t(8 * n);
    std::vector<BfDevItem> items(n);
    (void)hipMemcpy(t.data(), dev, n * 64, hipMemcpyDeviceToHost);
    (void)hipMemcpy(items.data(), a->items, n * sizeof(BfDevItem), hipMemcpyDeviceToHost);
    uint64_t np = 0;
    for (uint64_t i = 0; i < n; ++i) if ((uint64_t)items[i].pieceBegin + items[i].numPieces > np) np = (uint64_t)items[i].pieceBegin + items[i].numPieces;
    std::vector<BfDevPiece> pieces(np ? np : 1);
    if (np) (void)hipMemcpy(pieces.data(), a->pieces, np * sizeof(BfDevPiece), hipMemcpyDeviceToHost);
    FILE *f = fopen(getenv("BFHIP_TIMELINE_FILE"), "a");
    if (f) {
      fprintf(f, "launch %llu %llu\n", (unsigned long long)n, (unsigned long long)np);
      for (uint64_t i = 0; i < n; ++i) {
        uint64_t cols = 0;
        for (uint32_t k = 0; k < items[i].numPieces; ++k) if (!(pieces[items[i].pieceBegin + k].flags & BF_PIECE_IDENTITY)) cols += pieces[items[i].pieceBegin + k].ncols;
        fprintf(f, "%llu %llu %u %llu %u %llu %llu %llu %llu\n", (unsigned long long)t[8 * i], (unsigned long long)t[8 * i + 5], items[i].mrFlags & 0xffffu, (unsigned long long)cols, items[i].numPieces,
                (unsigned long long)t[8 * i + 1], (unsigned long long)t[8 * i + 2], (unsigned long long)t[8 * i + 3], (unsigned long long)t[8 * i + 4]);
      }
      fclose(f);
    }
  }
  (void)hipFree(dev);
  return rc;
}


// What bfdevLaunchStage asks first in an EXPERIMENTAL build: *handled = 1 if the stage was launched here.
int bfdevLaunchStageExperimental(BfLaunchArgs const *a, void const *stageParams, uint32_t grid, void *stream, int *handled) {
  StageParams const &p = *(StageParams const *)stageParams;
  hipStream_t s = (hipStream_t)stream;
  // (read per launch on purpose: tools/timeline.py switches the diagnostic on for ONE apply after its warm-up)
  char const *tlEnv = getenv("BFHIP_TIMELINE_FILE");
  int const timelineOn = tlEnv && tlEnv[0];
  uint32_t const slots = bfdevPersistentGrid();
  uint32_t const pgrid = (a->tickets && slots && a->numItems > slots) ? slots : 0;        /* 0: every item has a slot of its own anyway */
  *handled = 1;
  if (timelineOn) return bfTimelineLaunch(a, p, grid, pgrid, s);
  if (pgrid) return bfdevLaunchPersistC128(&p, pgrid, a->tickets, nullptr, s);
  *handled = 0;
  return 0;
}
int bfdevLaunchFlow(BfFlowArgs const *a, void *stream) {
  if (!a->numItems) return 0;
  FlowParams p;
  p.arena = a->arena;
  p.items = (BfDevItem const *)a->items;
  p.pieces = (BfDevPiece const *)a->pieces;
  p.itemOut = (uint32_t const *)a->itemOut;
  p.writers = (uint32_t const *)a->writers;
  p.counters = (uint32_t *)a->counters;
  p.queue = (uint32_t *)a->counters;            // slot 0: vector id 0 (x) has no counter of its own
  p.error = (uint32_t *)a->counters + 1;        // slot 1: vector id 1 (y) neither
  p.numItems = a->numItems; p.nrhs = a->nrhs; p.epoch = a->epoch; p.queueBase = 0;
  static uint32_t spinLimit = 0;
  if (!spinLimit) { char const *e = getenv("BFHIP_FLOW_SPIN"); spinLimit = e ? (uint32_t)strtoul(e, NULL, 10) : BF_FLOW_SPIN_LIMIT; if (!spinLimit) spinLimit = BF_FLOW_SPIN_LIMIT; }
  p.spinLimit = spinLimit;
  static int debugMode = -1;                     // read once, like the spin limit
  if (debugMode < 0) { char const *e = getenv("BFHIP_FLOW_DEBUGMODE"); debugMode = e ? (int)strtoul(e, NULL, 10) : 0; }
  p.debugMode = (uint32_t)debugMode;
  p.x = a->x; p.y = a->y; p.temp = a->temp;
  // tickets are drawn one item ahead, so how many an apply draws is not fixed: the queue starts from zero every time
  hipError_t e = hipMemsetAsync(a->counters, 0, 4, (hipStream_t)stream);
  if (e != hipSuccess) return hipFail(e, "flow queue reset");
  hipLaunchKernelGGL(bfFlowKernelC128, dim3(a->gridWorkgroups), dim3(BF_WAVES_PER_WG * 64), 0, (hipStream_t)stream, p);
  return hipFail(hipGetLastError(), "flow launch");
}

// workgroups of the persistent launch: what the device holds at once (occupancy x CUs), never more than the items need
int bfdevFlowGrid(uint64_t numItems, uint32_t *grid) {
  int dev = 0, cus = 0, perCu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, bfFlowKernelC128, BF_WAVES_PER_WG * 64, 0);
  if (e != hipSuccess) return hipFail(e, "flow occupancy");
  uint64_t g = (uint64_t)cus * (uint64_t)(perCu > 0 ? perCu : 1);
  uint64_t const need = (numItems + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG;
  if (g > need) g = need;
  *grid = (uint32_t)(g ? g : 1);
  return 0;
}
}
