// bfhip_device.hip -- gfx950 (MI355X, CDNA4) device layer of the butterfly
// apply engine: the stage kernel (variable-size batched small GEMV over a
// CSR-of-blocks level), the deterministic reduce kernel, the on-device
// operand synthesizer, and thin wrappers over the HIP runtime so that the C
// host never includes HIP headers.
//
// Stage kernel (what replaces one level of the reference's recursion, i.e.
// every cblas_zgemm / cblas_dgemv call of that level -- reference
// src/mat_dense_complex.c:1754, src/mat_dense_real.c:1358 -- plus the
// AddInplace / SetRowRange scatter passes around them,
// src/mat_block_coo.c:404-418, src/mat_block_diag.c:387-399):
//
//   * one 64-lane wavefront per work item = <= 64 row slots of one output row
//     group, looping over the item's pieces (the blocks of that block row);
//   * a piece is stored column-major (rows x cols), so with lanes owning rows
//     a wave load of 16 B/lane is one contiguous <= 1 KiB segment: HBM-bound
//     streaming with no cross-lane reduction in the inner loop;
//   * when the item has fewer than 64 row slots, g = 64 / slots column groups
//     share the wave (lane = group * slots + slot), so narrow blocks (rank
//     16..20 inner factors) still issue near-full-width loads; groups are
//     combined once per item through LDS;
//   * the piece's input sub-vector is gathered into LDS once and read back
//     with broadcast ds_reads (all lanes of a group read the same address);
//   * every output row has exactly one owner: plain stores, no atomics, and a
//     fixed summation order (results are run-to-run reproducible).
//
// Roofline: nrhs = 1 does 8 flops per 16 bytes of leaf data (0.5 flop/B), far
// below the ~10 flop/B ridge of MI355X FP64: the kernel is HBM-bound and is
// judged on leaf bytes / time against 8 TB/s (MI355X_MICROARCH.md).

#include <hip/hip_runtime.h>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "bfhip_internal.h"
#include "bfhip_stage_c128.h"
#include "../../include/bfhip_abi.h"
#include "../../include/bfhip_synth.h"

#define BF_WAVES_PER_WG 4
#define BF_WAVE_LDS_BYTES (BF_XCAP * 16)

static int hipFail(hipError_t e, char const *what) {
  if (e == hipSuccess) return 0;
  int code = (e == hipErrorOutOfMemory) ? BFABI_ERROR_MEMORY_ERROR : BFABI_ERROR_RUNTIME_ERROR;
  return bfhipFail(code, "%s: %s", what, hipGetErrorString(e));
}

// ---------------------------------------------------------------------------
// element traits
// ---------------------------------------------------------------------------
template <int DT> struct Traits;
template <> struct Traits<BFHIP_C128> { using S = double; static constexpr int EPL = 1; static constexpr bool CPLX = true; };
template <> struct Traits<BFHIP_F64> { using S = double; static constexpr int EPL = 2; static constexpr bool CPLX = false; };
template <> struct Traits<BFHIP_F32> { using S = float; static constexpr int EPL = 4; static constexpr bool CPLX = false; };

template <typename V> __device__ __forceinline__ V bfLoadStreamV(V const *p) {
  static_assert(sizeof(V) == 16, "one 16-byte lane load");
#if BF_STREAM_NT
  bf_u4 raw = __builtin_nontemporal_load((bf_u4 const *)p);
  V v;
  __builtin_memcpy(&v, &raw, 16);
  return v;
#else
  return *p;
#endif
}

// ---------------------------------------------------------------------------
// complex128 stage kernel
// ---------------------------------------------------------------------------
template <bool TL>
__device__ __forceinline__ void bfStageBodyC128(StageParams const &p, double2 (*lds)[BF_XCAP], uint64_t *timeline) {
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  uint32_t const item = __builtin_amdgcn_readfirstlane(blockIdx.x * BF_C128_WG_WAVES + wave);
  if (item >= p.numItems) return;
  BfDevItem const it = p.items[item];
  bfItemC128<TL, false>(p, it, item, lds[wave], lane, timeline, BfNoHook());
}

#ifdef BF_C128_WAVES
#define BF_C128_ATTR __attribute__((amdgpu_waves_per_eu(BF_C128_WAVES, BF_C128_WAVES)))
#else
#define BF_C128_ATTR
#endif
__global__ __launch_bounds__(BF_C128_WG_WAVES * 64) BF_C128_ATTR void bfStageKernelC128(StageParams p) {
  __shared__ __attribute__((aligned(16))) double2 lds[BF_C128_WG_WAVES][BF_XCAP];
  bfStageBodyC128<false>(p, lds, nullptr);
}

#include "bfhip_stage_mfma.h"

// One row-major piece of MR rows (see the row-major branch of bfStageKernelReal): lane owns 16-byte units u = lane,
// lane + 64, ... of every row; UNR units are in flight at once, so (MR + 1) * UNR independent loads per lane.  All
// addresses are a wave-uniform base plus a 32-bit lane offset (a piece is < 64 KiB per row set).
template <int DT, int MR>
__device__ __forceinline__ void bfRowMajorPiece(void const *rowpV, uint32_t upr, typename Traits<DT>::S const *xin, uint32_t n,
                                                uint32_t nrhs, int lane, typename Traits<DT>::S *racc) {
  using S = typename Traits<DT>::S;
  constexpr int EPL = Traits<DT>::EPL;
  struct __attribute__((aligned(16))) V { S v[EPL]; };
  struct __attribute__((packed, aligned(sizeof(S)))) VU { S v[EPL]; };     // x: element-aligned 16 bytes
  char const *rowp = (char const *)rowpV;
  constexpr int UNR = MR <= 2 ? 4 : 2;
  uint32_t const rowBytes = upr * 16u;
  uint32_t const full = nrhs == 1 ? n / EPL : 0;        // units whose x is one contiguous in-range 16-byte load
  uint32_t u = (uint32_t)lane;
  if (UNR > 1) {
#pragma unroll 1
    for (; u + 64u * (UNR - 1) < full; u += 64u * UNR) {
      V a[UNR][MR];
      VU xv[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
#pragma unroll
        for (int r = 0; r < MR; ++r) a[k][r] = bfLoadStreamV((V const *)(rowp + ((uint32_t)r * rowBytes + (u + 64u * k) * 16u)));
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) xv[k] = *(VU const *)((char const *)xin + (u + 64u * k) * 16u);
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) racc[r] = fma(a[k][r].v[e], xv[k].v[e], racc[r]);
        }
      }
    }
  }
#pragma unroll 1
  for (; u < full; u += 64u) {
    V a[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) a[r] = bfLoadStreamV((V const *)(rowp + ((uint32_t)r * rowBytes + u * 16u)));
    VU const xv = *(VU const *)((char const *)xin + u * 16u);
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) racc[r] = fma(a[r].v[e], xv.v[e], racc[r]);
    }
  }
  // what is left: the ragged last unit (nrhs == 1), or everything with strided x (nrhs > 1); x past column n is
  // taken as zero so that the zero row padding never meets a NaN
#pragma unroll 1
  for (; u < upr; u += 64u) {
    V a[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) a[r] = bfLoadStreamV((V const *)(rowp + ((uint32_t)r * rowBytes + u * 16u)));
    S xv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) xv[e] = u * EPL + e < n ? xin[(uint64_t)(u * EPL + e) * nrhs] : (S)0;
#pragma unroll
    for (int r = 0; r < MR; ++r) {
#pragma unroll
      for (int e = 0; e < EPL; ++e) racc[r] = fma(a[r].v[e], xv[e], racc[r]);
    }
  }
}

// Four small items per wavefront, one per group of 16 lanes (BF_ITEM_SMALL: at most 2 lane granules of rows, at most
// BF_SMALL_PIECES pieces, fewer than BF_SMALL_COLS dense columns in all).  A streamed butterfly's inner factors are
// hundreds of thousands of such items (row nodes of ~5 rows: an identity term plus a leaf of ~50 columns); one per
// wavefront they cost a launch slot and four dependent memory round trips each for ~1 KB of data.  Their dense pieces
// are stored ROW-major like the wide few-row leaves (rows end on the lane granule, nothing else is padded: 5 rows take
// 5/8 of what the column-major layout reads): group lane gl owns 16-byte column units gl and gl + 16 of every row, x
// comes straight from global memory, the rows are summed inside the 16 lanes by a fixed butterfly.  No LDS.
// (like the other stage kernels the body is a function of the block index: bfStageKernelRealBoth runs it next to the
// one-item-per-wavefront body in one launch)
template <int DT>
__device__ __forceinline__ void bfStageBodySmall(StageParams const &p, uint32_t const bid) {
  using S = typename Traits<DT>::S;
  constexpr int EPL = Traits<DT>::EPL;
  constexpr int RMAX = 2 * EPL;
  struct __attribute__((aligned(16))) V { S v[EPL]; };
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  uint32_t const item0 = __builtin_amdgcn_readfirstlane((bid * BF_WAVES_PER_WG + wave) * 4u);
  if (item0 >= p.numItems) return;
  uint32_t const gid = (uint32_t)lane >> 4, gl = (uint32_t)lane & 15u;
  uint32_t const idx = item0 + gid;
  bool const valid = idx < p.numItems;
  BfDevItem it = p.items[valid ? idx : p.numItems - 1];
  uint32_t const mr = valid ? (it.mrFlags & 0xffffu) : 0u;
  uint32_t const np = valid ? it.numPieces : 0u;
  // piece descriptors: group lane k holds piece k
  BfDevPiece my;
  my.dataOff = 0; my.inOff = 0; my.ncols = 0; my.flags = 0; my.ld = 0;
  if (gl < np) my = p.pieces[it.pieceBegin + gl];
  uint32_t npMax = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t const v = (uint32_t)__builtin_amdgcn_readlane((int)np, 16 * k);
    npMax = v > npMax ? v : npMax;
  }
  V const *arena = (V const *)p.arena;
  uint32_t const nrhs = p.nrhs;
  S *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (S *)p.y : (S *)p.temp;
  for (uint32_t q = 0; q < nrhs; ++q) {
    S racc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) racc[r] = 0;
    S ident = 0;                                       // group lane r: identity contributions to row r
    for (uint32_t k = 0; k < npMax; ++k) {
      int const from = (int)(gid * 16u + k);
      uint32_t const flags = (uint32_t)__shfl((int)my.flags, from, 64);
      uint32_t const inOff = (uint32_t)__shfl((int)my.inOff, from, 64);
      uint32_t const ncols = (uint32_t)__shfl((int)my.ncols, from, 64);
      uint32_t const ld = (uint32_t)__shfl((int)my.ld, from, 64);
      uint32_t const dLo = (uint32_t)__shfl((int)(uint32_t)my.dataOff, from, 64);
      uint32_t const dHi = (uint32_t)__shfl((int)(uint32_t)(my.dataOff >> 32), from, 64);
      if (k >= np) continue;
      S const *xin = ((flags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp) + ((uint64_t)inOff * nrhs + q);
      if (flags & BF_PIECE_IDENTITY) {
        if (gl < mr) ident += xin[(uint64_t)gl * nrhs];
        continue;
      }
      uint32_t const upr = ld / EPL;                   // 16-byte units per row: <= BF_SMALL_COLS / EPL
      V const *base = arena + (((uint64_t)dHi << 32) | dLo) / EPL;
#pragma unroll 1
      for (uint32_t u = gl; u < upr; u += 16u) {
        V a[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) if ((uint32_t)r < mr) a[r] = bfLoadStreamV(base + (uint64_t)r * upr + u);
        S xv[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) xv[e] = u * EPL + e < ncols ? xin[(uint64_t)(u * EPL + e) * nrhs] : (S)0;   // the zero row padding never meets a NaN
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
          if ((uint32_t)r < mr) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) racc[r] = fma(a[r].v[e], xv[e], racc[r]);
          }
      }
    }
    // sum over the group's 16 lanes: fixed butterfly (deterministic); group lane r keeps row r
    S mine = ident;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      S t = racc[r];
      t += __shfl_xor(t, 8, 64);
      t += __shfl_xor(t, 4, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 1, 64);
      if (gl == (uint32_t)r) mine += t;
    }
    if (gl < mr) out[((uint64_t)it.outOff + gl) * nrhs + q] = mine;
  }
}

template <int DT>
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void bfStageKernelSmall(StageParams p) {
  bfStageBodySmall<DT>(p, blockIdx.x);
}

// ---------------------------------------------------------------------------
// real stage kernel (f64: 2 rows per lane, f32: 4 rows per lane)
// ---------------------------------------------------------------------------
#ifndef BF_REAL_UNROLL
#define BF_REAL_UNROLL 8
#endif
template <int DT>
__device__ __forceinline__ void bfStageBodyReal(StageParams const &p, uint32_t const bid, unsigned char (*ldsRaw)[BF_WAVE_LDS_BYTES]) {
  using S = typename Traits<DT>::S;
  constexpr int EPL = Traits<DT>::EPL;
  struct __attribute__((aligned(16))) V { S v[EPL]; };
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  uint32_t item = __builtin_amdgcn_readfirstlane(bid * BF_WAVES_PER_WG + wave);
  if (item >= p.numItems) return;
  S *xs = (S *)ldsRaw[wave];
  V const *arena = (V const *)p.arena;
  uint32_t const nrhs = p.nrhs;
  BfDevItem const it = p.items[item];
  uint32_t const mr = it.mrFlags & 0xffffu;
  uint32_t const ms = (mr + EPL - 1) / EPL;        // row slots
  uint32_t const mrPad = ms * EPL;
  uint32_t const g = 64u / ms;
  uint32_t const G = g * ms;
  bool const active = (uint32_t)lane < G;
  uint32_t const lc = active ? (uint32_t)lane : G - 1;
  uint32_t const c = lc / ms;
  uint32_t const rs = lc - c * ms;
  S *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (S *)p.y : (S *)p.temp;

  if (it.mrFlags & BF_ITEM_ROWMAJOR) {
    // Few-row wide leaves (mr <= 2 EPL rows, stored row-major, rows padded with zeros to the lane granule): a lane
    // owns 16 bytes of consecutive columns, loads them from every row of the piece and multiplies with the matching
    // 16 bytes of x, read straight from global memory (no LDS, no hand-off between lanes, so a piece may be as wide
    // as a task); rows are summed over the wave at the end by a fixed butterfly.  No row padding is read and every
    // lane is busy whatever mr is.  Piece descriptors come 64 at a time, one per lane.
    constexpr int RMAX = 2 * EPL;
    for (uint32_t q = 0; q < nrhs; ++q) {
      S racc[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) racc[r] = 0;
      S ident = 0;                                   // lane r < mr: identity contributions to row r
      for (uint32_t p0 = 0; p0 < it.numPieces; p0 += 64) {
        uint32_t const np = it.numPieces - p0 < 64u ? it.numPieces - p0 : 64u;
        BfPieceWin const win = bfPieceWinLoad(p.pieces + it.pieceBegin + p0, np, lane);
        for (uint32_t pi = 0; pi < np; ++pi) {
          BfDevPiece const pc = bfPieceWinGet(win, pi);
          S const *xin = (pc.flags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp;
          xin += (uint64_t)pc.inOff * nrhs + q;
          if (pc.flags & BF_PIECE_IDENTITY) {
            if ((uint32_t)lane < mr) ident += xin[(uint64_t)lane * nrhs];
            continue;
          }
          V const *rowp = arena + pc.dataOff / EPL;
          uint32_t const upr = pc.ld / EPL;
          switch (mr) {                                // wave-uniform
            case 1: bfRowMajorPiece<DT, 1>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 2: bfRowMajorPiece<DT, 2>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 3: bfRowMajorPiece<DT, 3>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 4: bfRowMajorPiece<DT, 4>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 5: bfRowMajorPiece<DT, (RMAX > 4 ? 5 : 4)>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 6: bfRowMajorPiece<DT, (RMAX > 4 ? 6 : 4)>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            case 7: bfRowMajorPiece<DT, (RMAX > 4 ? 7 : 4)>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
            default: bfRowMajorPiece<DT, RMAX>(rowp, upr, xin, pc.ncols, nrhs, lane, racc); break;
          }
        }
      }
      S mine = ident;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        S t = racc[r];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);      // fixed order: deterministic
        if ((uint32_t)lane == (uint32_t)r) mine += t;
      }
      if ((uint32_t)lane < mr) out[((uint64_t)it.outOff + lane) * nrhs + q] = mine;
    }
    return;
  }

  // The pieces of an item are packed back to back, so a run of consecutive narrow pieces is ONE contiguous mrPad x n
  // block (n <= BF_MERGE_COLS): lane l gathers x for columns l, l + 64, ... of the block from whichever piece holds
  // them -- all pieces' loads in flight together -- and the block is contracted in one go.  A narrow item (what a
  // streamed butterfly's inner factors are made of) costs three dependent memory round trips instead of three per
  // piece; an item of hundreds of 5-column pieces (the transposes of few-row leaves in a packed adjoint plan) is a few
  // such runs instead of hundreds of two-step contractions.  Items the planner flags BF_ITEM_MERGED are a single run.
  // A piece wider than BF_MERGE_COLS is contracted on its own, its x handed over through LDS in one loop.
  for (uint32_t q = 0; q < nrhs; ++q) {
    S acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0;
    for (uint32_t p0 = 0; p0 < it.numPieces; p0 += 64) {
      uint32_t const np = it.numPieces - p0 < 64u ? it.numPieces - p0 : 64u;
      BfPieceWin const win = bfPieceWinLoad(p.pieces + it.pieceBegin + p0, np, lane);
      uint32_t k = 0;
      while (k < np) {
        constexpr int XS = BF_MERGE_COLS / 64;
        uint32_t src[XS];                 // input element index of the lane's column t, ~0 = none
        uint32_t srcX = 0;                // bit t: that index is into x (else the vector arena)
#pragma unroll
        for (int t = 0; t < XS; ++t) src[t] = ~0u;
        uint32_t base = 0, wideCols = 0, wideIn = 0, wideFlags = 0;
        uint64_t data0 = 0, nextOff = 0;
        for (; k < np; ++k) {
          BfDevPiece const pc = bfPieceWinGet(win, k);
          if (pc.flags & BF_PIECE_IDENTITY) {
            S const *xin = (pc.flags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp;
            xin += (uint64_t)pc.inOff * nrhs + q;
            if (c == 0 && active) {
#pragma unroll
              for (int e = 0; e < EPL; ++e) {
                uint32_t row = rs * EPL + e;
                if (row < mr) acc[e] += xin[(uint64_t)row * nrhs];
              }
            }
            continue;
          }
          if (pc.ncols > BF_MERGE_COLS) {                   // a wide piece: on its own, after the run before it
            if (base) break;
            wideCols = pc.ncols; wideIn = pc.inOff; wideFlags = pc.flags; data0 = pc.dataOff;
            ++k;
            break;
          }
          if (base && (pc.dataOff != nextOff || base + pc.ncols > BF_MERGE_COLS)) break;
          if (base == 0) data0 = pc.dataOff;
          nextOff = pc.dataOff + (uint64_t)mrPad * pc.ncols;
#pragma unroll
          for (int t = 0; t < XS; ++t) {
            uint32_t const J = (uint32_t)lane + 64u * t - base;        // wraps below base
            if (J < pc.ncols) { src[t] = pc.inOff + J; srcX = (pc.flags & BF_PIECE_IN_X) ? srcX | (1u << t) : srcX & ~(1u << t); }
          }
          base += pc.ncols;
        }
        uint32_t n;
        V const *ap = arena + data0 / EPL + lc;               // dataOff is in elements; a lane load is EPL elements
        if (wideCols) {
          S const *xin = (wideFlags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp;
          xin += (uint64_t)wideIn * nrhs + q;
          n = wideCols;
          waveSync();
          for (uint32_t j = lane; j < n; j += 64) xs[j] = xin[(uint64_t)j * nrhs];
          waveSync();
        } else if (base) {
          n = base;
          // a block of at most 2 wave loads is fetched before x is handed over, so both trips overlap
          bool const tiny = n <= 2 * g;
          V a2[2];
          if (tiny) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
              for (int e = 0; e < EPL; ++e) a2[kk].v[e] = 0;
              if (active && c + kk * g < n) a2[kk] = bfLoadStreamV(ap + (uint64_t)kk * G);
            }
          }
          S xg[XS];
#pragma unroll
          for (int t = 0; t < XS; ++t)
            xg[t] = src[t] != ~0u ? (((srcX >> t) & 1u) ? (S const *)p.x : (S const *)p.temp)[(uint64_t)src[t] * nrhs + q] : (S)0;
          waveSync();
#pragma unroll
          for (int t = 0; t < XS; ++t) if ((uint32_t)lane + 64u * t < n) xs[lane + 64 * t] = xg[t];
          waveSync();
          if (tiny) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              S xv = (active && c + kk * g < n) ? xs[c + kk * g] : (S)0;
#pragma unroll
              for (int e = 0; e < EPL; ++e) acc[e] = fma(a2[kk].v[e], xv, acc[e]);
            }
            continue;
          }
        } else {
          continue;                                          // identity pieces only
        }
        uint32_t const nfull = n / g;
        uint32_t j = c;
        uint32_t s = 0;
        // whole groups of 8 steps, then the ragged end with ITS loads issued together as well (the compiler's remainder
        // loop waits for every load on its own: up to 7 memory round trips per run; bfhip_stage_c128.h)
        V a8[BF_REAL_UNROLL];
#pragma unroll 1
        for (; s + BF_REAL_UNROLL <= nfull; s += BF_REAL_UNROLL) {
#pragma unroll
          for (int k = 0; k < BF_REAL_UNROLL; ++k) a8[k] = bfLoadStreamV(ap + (uint64_t)(s + k) * G);
#pragma unroll
          for (int k = 0; k < BF_REAL_UNROLL; ++k) {
            S const xv = xs[j];
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = fma(a8[k].v[e], xv, acc[e]);
            j += g;
          }
        }
        if (s < nfull) {                                         // wave-uniform
          uint32_t const left = nfull - s;
#pragma unroll
          for (int k = 0; k < BF_REAL_UNROLL - 1; ++k)
            if ((uint32_t)k < left) a8[k] = bfLoadStreamV(ap + (uint64_t)(s + k) * G);
#pragma unroll
          for (int k = 0; k < BF_REAL_UNROLL - 1; ++k)
            if ((uint32_t)k < left) {
              S const xv = xs[j];
#pragma unroll
              for (int e = 0; e < EPL; ++e) acc[e] = fma(a8[k].v[e], xv, acc[e]);
              j += g;
            }
        }
        uint32_t const rem = n - nfull * g;
        if (active && c < rem) {
          V a = bfLoadStreamV(ap + (uint64_t)nfull * G);
          S xv = xs[j];
#pragma unroll
          for (int e = 0; e < EPL; ++e) acc[e] = fma(a.v[e], xv, acc[e]);
        }
      }
    }
    waveSync();
#pragma unroll
    for (int e = 0; e < EPL; ++e) xs[lane * EPL + e] = acc[e];   // index = c*mrPad + row for active lanes
    waveSync();
    for (uint32_t row = lane; row < mr; row += 64) {
      S sum = 0;
      for (uint32_t cc = 0; cc < g; ++cc) sum += xs[cc * mrPad + row];
      out[((uint64_t)it.outOff + row) * nrhs + q] = sum;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) void bfStageKernelReal(StageParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[BF_WAVES_PER_WG][BF_WAVE_LDS_BYTES];
  bfStageBodyReal<DT>(p, blockIdx.x, ldsRaw);
}

// The two item families of a real stage in ONE launch: workgroups [0, gridReal) take one item per wavefront, the rest
// four small items per wavefront.  The two write disjoint rows.  As two launches every inner stage of a streamed
// butterfly ended with a 55 - 85 us kernel of ~10^4 short workgroups that ran at a fraction of the memory rate behind the
// tail of the first; in one launch the small items fill that tail.  Both bodies fit 4 wavefronts per SIMD already.
template <int DT>
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) __attribute__((amdgpu_waves_per_eu(4, 8))) void bfStageKernelRealBoth(StageParams pr, StageParams ps, uint32_t gridReal) {
  __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[BF_WAVES_PER_WG][BF_WAVE_LDS_BYTES];
  if (blockIdx.x < gridReal) bfStageBodyReal<DT>(pr, blockIdx.x, ldsRaw);
  else bfStageBodySmall<DT>(ps, blockIdx.x - gridReal);
}


// ---------------------------------------------------------------------------
// transposed stage kernel: y = A^T x on the forward plan's packed pieces
// (reference: RmulVec of every container, see include/bfhip.h).
//
// An item is <= 16 columns of A (16 outputs); a piece is a forward piece's
// sub-block of those columns: column-major, `ld` elements per column, hence
// ONE contiguous run of 16*ld elements that is read exactly once.  Lane =
// (c4 = lane / 16, r = lane % 16): one load instruction fetches, for each of 4
// columns (c4 + 4*cq), 16 consecutive 16-byte units (256 contiguous bytes per
// 16-lane row), so HBM sees full lines and the same streaming pattern as the
// forward kernel.  A lane keeps one accumulator per column quad cq: 4 of them,
// whatever the piece height, so nothing is staged in LDS and the kernel runs at
// full occupancy; the 16 row-lanes of a column are combined once per item
// with a fixed xor-butterfly inside the 16-lane row (deterministic).
// ---------------------------------------------------------------------------
#define BF_T_COLS 16
// With 4 row lanes a load instruction takes 64 bytes of each column, half a 128-byte line; the other half is asked
// for by the next block step.  Streamed (non-temporal) lines are not kept for it, so the wide kernel loads its
// column-major pieces with the default policy (row-major pieces are whole 128-byte lines read once: streamed) (measured on the streamed operand: every transposed stage 10 - 35 % shorter, DESIGN_EXPERIMENTS.md section 10).
#ifndef BF_T_WIDE_NT
#define BF_T_WIDE_NT 0
#endif
// Measured on the streamed N = 1M fp32 operand (transposed apply 8.63 ms, DESIGN_EXPERIMENTS.md section 10): two row blocks requested
// before the first is used (8 KB per wavefront in flight, 126 VGPRs, still 4 wavefronts per SIMD) -> 8.82 ms; 8 row lanes
// x 8 loads (BF_T_WIDE_R 8: 155 VGPRs, 3 wavefronts per SIMD) -> 9.65 ms; the same loads at consecutive addresses (wrong
// results, timing only) -> no change.  Neither bytes in flight nor the 64-byte-per-column pattern bounds this kernel: its
// wavefronts issue 12 VALU instructions per vector load (SQ counters, profiles/), so it is the instruction count per
// byte and the resident wavefronts that pay -- hence the single-right-hand-side instantiations below (fp32: 111 -> 92
// VGPRs, 5 wavefronts per SIMD, 8.63 -> 8.34 ms).  Forcing 6 or 7 wavefronts with amdgpu_waves_per_eu spills 12 / 24
// VGPRs to scratch: 9.02 / 10.47 ms.
// row lanes of the 64-column tiling (= its loads in flight: R row lanes x 64 / R columns per load, R loads per item width)
#ifndef BF_T_WIDE_R
#define BF_T_WIDE_R 4
#endif

// R = row lanes per column, C = 64 / R columns per load instruction, NQ loads in flight per block: an
// item is up to NQ * C columns of A.
//   R = 16, NQ = 4 (16-column items): tall pieces (fac_helm2: 16 - 64 row chunks of complex128 = 1 - 4 blocks).
//   R = 4, NQ = 4 (64-column items): operands made of short leaves (fac_streamer: 20 - 60 rows = 5 - 15
//     16-byte units): with 16 row lanes a 10-unit piece keeps 10 of 16 row lanes busy and a 36-column
//     remainder 9 of 16 quads; with 4 row lanes a block covers 4 units x 16 columns, the row waste drops
//     to the last 4-unit step and items are 4x larger.
// (the body is a function of the block index so that one launch can run two tilings side by side: bfStageKernelTBoth)
// ONE: a single right-hand side, known at compile time -- the 64-bit row * nrhs products of every x address (quarter-rate
// v_mad_u64_u32, four per block step of the fp32 tiling) fold away.
template <int DT, int R, int NQ, bool COOP, bool ONE, typename CoopBuf>
__device__ __forceinline__ void bfStageBodyT(StageParams const &p, uint32_t const bid, CoopBuf &coopBuf) {
  constexpr int C = 64 / R;                          // columns per load instruction
  using S = typename Traits<DT>::S;
  constexpr int EPL = Traits<DT>::EPL;              // rows per 16-byte unit (complex: 1)
  constexpr int NC = Traits<DT>::CPLX ? 2 : 1;      // scalars per element
  constexpr int UNIT = EPL * NC;                    // scalars per 16-byte unit
  struct __attribute__((aligned(16))) U { S v[UNIT]; };
  int const wave = threadIdx.x >> 6;
  int const lane = threadIdx.x & 63;
  // the first coopItems items (the big ones: the list is sorted) get a workgroup each: wavefront w takes pieces
  // w, w + 4, ... and the four partial results are added in LDS in a fixed order; the rest run one per wavefront
  // (COOP = false: the stage has no such items and the kernel is the plain one-item-per-wavefront one)
  bool const coop = COOP && bid < p.coopItems;
  uint32_t item = __builtin_amdgcn_readfirstlane(coop ? bid : p.coopItems + (bid - p.coopItems) * BF_WAVES_PER_WG + wave);
  if (item >= p.numItems) return;
  uint32_t const wsel = coop ? (uint32_t)wave : 0u, wmask = coop ? BF_WAVES_PER_WG - 1u : 0u;
  BfDevItem const it = p.items[item];
  uint32_t const mr = it.mrFlags & 0xffffu;          // columns of A in this item (<= NQ * C)
  uint32_t const c4 = lane / R, r = lane % R;
  uint32_t jcol[NQ];                                 // this lane's NQ columns, clamped into the item:
#pragma unroll                                       // the duplicates are summed but never stored
  for (int cq = 0; cq < NQ; ++cq) jcol[cq] = c4 + C * cq < mr ? c4 + C * cq : mr - 1;
  U const *arena = (U const *)p.arena;               // 16-byte units
  uint32_t const nrhs = ONE ? 1u : p.nrhs;
  S *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (S *)p.y : (S *)p.temp;
  uint32_t const nq = (mr + C - 1) / C;              // load instructions that touch a column of this item

  // lane mapping for row-major pieces: CG_RM groups of EPL columns across the item's width, RL_RM row lanes
  constexpr int CG_RM = (NQ * C) / EPL > 64 ? 64 : (NQ * C) / EPL;
  constexpr int RL_RM = 64 / CG_RM;
  uint32_t const cgRm = lane % CG_RM, rlRm = lane / CG_RM;

  for (uint32_t q = 0; q < nrhs; ++q) {
    S acc[NQ][NC];
#pragma unroll
    for (int cq = 0; cq < NQ; ++cq)
#pragma unroll
      for (int k = 0; k < NC; ++k) acc[cq][k] = 0;
    S racc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) racc[e] = 0;
    bool anyRowMajor = false;
    for (uint32_t wbase = 0; wbase < it.numPieces; wbase += 64) {
      uint32_t const wn = it.numPieces - wbase < 64 ? it.numPieces - wbase : 64;
      BfPieceWin const win = bfPieceWinLoad(p.pieces + it.pieceBegin + wbase, wn, lane);
      uint32_t const pstep = wmask + 1u;               // this wavefront's pieces: wsel, wsel + pstep, ... (wbase is a multiple of 64)
      for (uint32_t pi = wsel; pi < wn; pi += pstep) {
        BfDevPiece const pc = bfPieceWinGet(win, pi);
        S const *xin = (pc.flags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp;
        xin += ((uint64_t)pc.inOff * nrhs + q) * NC;
        if (pc.flags & BF_PIECE_IDENTITY) {
          if (r == 0) {
#pragma unroll
            for (int cq = 0; cq < NQ; ++cq)
#pragma unroll
              for (int k = 0; k < NC; ++k) acc[cq][k] += xin[(uint64_t)jcol[cq] * nrhs * NC + k];
          }
          continue;
        }
        if constexpr (!Traits<DT>::CPLX) {
          if (pc.flags & BF_PIECE_ROWMAJOR) {
            // rows of a row-major forward piece (few-row leaves): element (row s, column j) = arena[dataOff + s * ld + j].
            // Lane = (column group cg of EPL consecutive columns, row lane rl): one 16-byte load per row, no reduction
            // inside the piece; the RL row lanes are combined once per item below.
            anyRowMajor = true;
            uint32_t const n = pc.ncols;                          // rows taken (<= 2 EPL)
            uint32_t const cgc = cgRm < (mr + EPL - 1) / EPL ? cgRm : (mr - 1) / EPL;   // clamp into the item's columns
            U const *src = arena + pc.dataOff / EPL + cgc;
            uint32_t const rowUnits = pc.ld / EPL;
            // every row this lane owns is requested before the first is used (rows past n: clamped address, x = 0).
            // A block column of a streamed butterfly is a chain of hundreds of such ~1 KB pieces: when the wavefront's
            // next piece is row-major too, both are requested before either is used (two pieces in flight for 10 VGPRs)
            constexpr int ITERS = (2 * EPL + RL_RM - 1) / RL_RM;
            bool dual = pi + pstep < wn;
            BfDevPiece pc2 = pc;
            if (dual) {
              pc2 = bfPieceWinGet(win, pi + pstep);
              dual = (pc2.flags & (BF_PIECE_ROWMAJOR | BF_PIECE_IDENTITY)) == BF_PIECE_ROWMAJOR;
            }
            U a[ITERS], a2[ITERS];
            S xv[ITERS], xv2[ITERS];
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
              uint32_t const srow = rlRm + (uint32_t)i * RL_RM, sc = srow < n ? srow : n - 1;
              a[i] = bfLoadStreamV(src + (uint64_t)sc * rowUnits);
              S const xr = xin[(uint64_t)sc * nrhs];
              xv[i] = srow < n ? xr : (S)0;
            }
            if (dual) {                                             // wave-uniform
              S const *xin2 = ((pc2.flags & BF_PIECE_IN_X) ? (S const *)p.x : (S const *)p.temp) + ((uint64_t)pc2.inOff * nrhs + q);
              uint32_t const n2 = pc2.ncols;
              U const *src2 = arena + pc2.dataOff / EPL + cgc;
              uint32_t const rowUnits2 = pc2.ld / EPL;
#pragma unroll
              for (int i = 0; i < ITERS; ++i) {
                uint32_t const srow = rlRm + (uint32_t)i * RL_RM, sc = srow < n2 ? srow : n2 - 1;
                a2[i] = bfLoadStreamV(src2 + (uint64_t)sc * rowUnits2);
                S const xr = xin2[(uint64_t)sc * nrhs];
                xv2[i] = srow < n2 ? xr : (S)0;
              }
            }
#pragma unroll
            for (int i = 0; i < ITERS; ++i)
#pragma unroll
              for (int e = 0; e < EPL; ++e) racc[e] = fma(a[i].v[e], xv[i], racc[e]);
            if (dual) {
#pragma unroll
              for (int i = 0; i < ITERS; ++i)
#pragma unroll
                for (int e = 0; e < EPL; ++e) racc[e] = fma(a2[i].v[e], xv2[i], racc[e]);
              pi += pstep;                                          // the second piece is done
            }
            continue;
          }
        }
        uint32_t const stride = pc.ld / EPL;         // 16-byte units between columns (forward mrPad / EPL)
        uint32_t const n = pc.ncols;                 // rows of the forward piece taken (a piece may be entered part-way)
        uint32_t const units = (n + EPL - 1) / EPL;  // 16-byte units per column that hold them
        U const *src = arena + pc.dataOff / EPL;
        for (uint32_t rb = 0; rb < units; rb += R) {
          // the last R-unit block of a column may be short: clamp the unit index into the column
          // (the rows it stands for are >= n, so their x is zero below) -- never read past the piece
          uint32_t const ru = rb + r < units ? rb + r : units - 1;
          U a[NQ];
#pragma unroll
          for (int cq = 0; cq < NQ; ++cq) if ((uint32_t)cq < nq) a[cq] = (R >= 16 || BF_T_WIDE_NT) ? bfLoadStreamV(src + jcol[cq] * stride + ru) : src[jcol[cq] * stride + ru];   // nq: wave-uniform
          S xv[UNIT];
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            // unconditional loads at a clamped row, zeroed by a select: a load under `row < n ? ... : 0`
            // becomes a branch with its own wait, and the 2 - 4 x loads of a block then run one after another
            // (one 16-byte load for the blocks whose rows all exist was tried for nrhs = 1: no change)
            uint32_t const row = (rb + r) * EPL + e;
            uint32_t const rowc = row < n ? row : n - 1;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
              S const v = xin[(uint64_t)rowc * nrhs * NC + k];
              xv[e * NC + k] = row < n ? v : (S)0;
            }
          }
#pragma unroll
          for (int cq = 0; cq < NQ; ++cq) {
            if ((uint32_t)cq >= nq) continue;
            if (Traits<DT>::CPLX) {
              acc[cq][0] = fma(a[cq].v[0], xv[0], acc[cq][0]); acc[cq][0] = fma(-a[cq].v[1], xv[NC - 1], acc[cq][0]);
              acc[cq][NC - 1] = fma(a[cq].v[0], xv[NC - 1], acc[cq][NC - 1]); acc[cq][NC - 1] = fma(a[cq].v[1], xv[0], acc[cq][NC - 1]);
            } else {
#pragma unroll
              for (int e = 0; e < EPL; ++e) acc[cq][0] = fma(a[cq].v[e], xv[e], acc[cq][0]);
            }
          }
        }
      }
    }
    // combine the R row lanes of every column (fixed butterfly order) and store
#pragma unroll
    for (int m = 1; m < R; m <<= 1)
#pragma unroll
      for (int cq = 0; cq < NQ; ++cq)
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[cq][k] += __shfl_xor(acc[cq][k], m, R);
    if constexpr (!Traits<DT>::CPLX) {
      if (anyRowMajor) {      // wave-uniform: same pieces for every lane
#pragma unroll
        for (int e = 0; e < EPL; ++e)
#pragma unroll
          for (int m = CG_RM; m < 64; m <<= 1) racc[e] += __shfl_xor(racc[e], m, 64);     // over the row lanes, fixed order
        // column col's total sits in lane col / EPL, element col % EPL: hand it to the lane that stores col
#pragma unroll
        for (int cq = 0; cq < NQ; ++cq) {
          uint32_t const col = c4 + C * cq < mr ? c4 + C * cq : 0;
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            S const t = __shfl(racc[e], (int)(col / EPL), 64);
            if ((int)(col % EPL) == e) acc[cq][0] += t;
          }
        }
      }
    }
    if (COOP && coop) {
      if (r == 0) {
#pragma unroll
        for (int cq = 0; cq < NQ; ++cq)
#pragma unroll
          for (int k = 0; k < NC; ++k) coopBuf[wave][c4 + C * cq][k] = acc[cq][k];
      }
      __syncthreads();
      if (wave == 0 && r == 0) {
#pragma unroll
        for (int cq = 0; cq < NQ; ++cq)
#pragma unroll
          for (int k = 0; k < NC; ++k) {
            S t = coopBuf[0][c4 + C * cq][k];
#pragma unroll
            for (int w = 1; w < BF_WAVES_PER_WG; ++w) t += coopBuf[w][c4 + C * cq][k];
            acc[cq][k] = t;
          }
      }
      __syncthreads();
      if (wave != 0) continue;
    }
    if (r == 0) {
#pragma unroll
      for (int cq = 0; cq < NQ; ++cq)
        if (c4 + C * cq < mr)
#pragma unroll
          for (int k = 0; k < NC; ++k) out[(((uint64_t)it.outOff + c4 + C * cq) * nrhs + q) * NC + k] = acc[cq][k];
    }
  }
}

template <int DT, int R, int NQ, bool COOP, bool ONE>
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) void bfStageKernelT(StageParams p) {
  using S = typename Traits<DT>::S;
  constexpr int NC = Traits<DT>::CPLX ? 2 : 1;
  __shared__ S coopBuf[COOP ? BF_WAVES_PER_WG : 1][COOP ? NQ * (64 / R) : 1][NC];
  bfStageBodyT<DT, R, NQ, COOP, ONE>(p, blockIdx.x, coopBuf);
}

// Both tilings of a transposed stage in ONE launch (real operands): workgroups [0, gridNarrow) run the 16-row-lane
// tiling on the 16-column items of tall leaves, the rest the 4-row-lane tiling on the 64-column items.  As two launches a
// stage of a streamed butterfly paid the ramp and the tail of a 0.15 - 0.6 ms kernel twice and the two item families could
// not fill each other's tails; both kernels already run at 4 wavefronts per SIMD with the shared-item code, so nothing is
// lost by giving them one register allocation.
template <int DT, bool ONE>
__global__ __launch_bounds__(BF_WAVES_PER_WG * 64) void bfStageKernelTBoth(StageParams pn, StageParams pw, uint32_t gridNarrow) {
  using S = typename Traits<DT>::S;
  __shared__ S coopBuf[BF_WAVES_PER_WG][64][1];
  if (blockIdx.x < gridNarrow) bfStageBodyT<DT, 16, 4, true, ONE>(pn, blockIdx.x, coopBuf);
  else bfStageBodyT<DT, BF_T_WIDE_R, BF_T_WIDE_R, true, ONE>(pw, blockIdx.x - gridNarrow, coopBuf);
}

// ---------------------------------------------------------------------------
// deterministic reduce: dest[row] = sum over the row's interval sources, in
// list order, of temp[srcBias + row]
// ---------------------------------------------------------------------------
// All reduces of a stage run in one launch (they are a few microseconds each: 12 separate
// launches cost 5 % of an apply at N = 65536): the descriptors travel by value in the kernel
// arguments, a workgroup finds its reduce by scanning <= BF_REDUCE_BATCH block offsets.
#define BF_REDUCE_BATCH 16
struct ReduceBatch {
  uint32_t count, nrhs;
  void const *temp;
  struct Entry {
    uint32_t const *rowInterval, *ivBegin;
    int64_t const *srcBias;
    uint64_t numRows;
    void *dest;
    uint32_t blockBegin, pad;
  } e[BF_REDUCE_BATCH];
};

template <typename T, int NC, bool LONG = false>   // NC = scalar components per element; LONG: some row has >= 64 partial sums
__global__ __launch_bounds__(256) void bfReduceKernel(ReduceBatch const B) {
  uint32_t k = 0;
  while (k + 1 < B.count && B.e[k + 1].blockBegin <= blockIdx.x) ++k;
  ReduceBatch::Entry const &E = B.e[k];
  uint32_t const nrhs = B.nrhs;
  uint64_t idx = (uint64_t)(blockIdx.x - E.blockBegin) * 256 + threadIdx.x;
  uint64_t total = E.numRows * nrhs;
  if (idx >= total) return;
  uint64_t row = idx / nrhs;
  uint32_t q = (uint32_t)(idx - row * nrhs);
  uint32_t iv = E.rowInterval[row];
  if (iv == 0xffffffffu) return;              // BF_REDUCE_SKIP: the one group that owns this row wrote it where it belongs
  uint32_t b = E.ivBegin[iv], e = E.ivBegin[iv + 1];
  T const *temp = (T const *)B.temp;
  T *dest = (T *)E.dest;
  T acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = 0;
  uint32_t s = b;
  if (LONG) {
    // a long list (a block column cut into hundreds of groups: a finely cut stage of a packed adjoint plan has rows with ~1000
    // partial sums) 32 loads at a time: with 8 in flight its few wavefronts were the whole launch (86 us; 59 so).  Its own
    // instantiation: in the common one the 32 registers cost occupancy (12 -> 20 us on a 1.3 M-row reduce).
    for (; s + 32 <= e; s += 32) {
      T v[32][NC];
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        T const *src = temp + ((uint64_t)(E.srcBias[s + k] + (int64_t)row) * nrhs + q) * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[k][c] = src[c];
      }
#pragma unroll
      for (int k = 0; k < 32; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] += v[k][c];
    }
  }
#pragma unroll 8
  for (; s < e; ++s) {          // same order as before; the unrolled loads are issued together
    T const *src = temp + ((uint64_t)(E.srcBias[s] + (int64_t)row) * nrhs + q) * NC;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] += src[c];
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) dest[idx * NC + c] = acc[c];
}

// dst[perm ? perm[i] : i] = src[i] * (scale ? scale[i]^power : 1): the vector plumbing around the two applies of a
// covariance product (bfVecRealPermute scatters: out[perm[i]] = in[i], src/vec_real.c:312-329; a BfMatDiagReal applied
// once or twice, examples/covariance/lbo_cov.c:36-60)
template <typename S>
__global__ __launch_bounds__(256) void bfScalePermuteKernel(S *dst, S const *src, S const *scale, int power, uint64_t const *perm, uint64_t n) {
  uint64_t const i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  S v = src[i];
  if (scale) { S const g = scale[i]; v *= power == 2 ? g * g : g; }
  uint64_t const j = perm ? perm[i] : i;
  if (j < n) dst[j] = v;          // an index outside the vector (not a permutation) is dropped, never written
}

// ---------------------------------------------------------------------------
// synthetic operand fill: one workgroup per piece
// ---------------------------------------------------------------------------
template <int DT>
__global__ __launch_bounds__(256) void bfSynthKernel(void *arenaV, BfSynthPiece const *pieces, uint64_t seed) {
  using S = typename Traits<DT>::S;
  constexpr bool CPLX = Traits<DT>::CPLX;
  BfSynthPiece const pc = pieces[blockIdx.x];
  uint64_t total = pc.rowMajor ? (uint64_t)pc.mr * pc.ldr : (uint64_t)pc.mrPad * pc.ncols;
  S *dst = (S *)arenaV + pc.dataOff * (CPLX ? 2 : 1);
  for (uint64_t e = threadIdx.x; e < total; e += 256) {
    uint32_t col, r;
    if (pc.rowMajor) { r = (uint32_t)(e / pc.ldr); col = (uint32_t)(e - (uint64_t)r * pc.ldr); }
    else { col = (uint32_t)(e / pc.mrPad); r = (uint32_t)(e - (uint64_t)col * pc.mrPad); }
    S re = 0, im = 0;
    if (r < pc.mr && col < pc.ncols) {
      uint64_t idx = pc.vbase + (uint64_t)(pc.row0 + r) * pc.strideR + (uint64_t)(pc.col0 + col) * pc.strideC;
      re = (S)(bfhip_synth_value(seed, idx, 0) * pc.scale);
      if (CPLX) im = (S)(bfhip_synth_value(seed, idx, 1) * pc.scale);
    }
    if (CPLX) { dst[2 * e] = re; dst[2 * e + 1] = im; }
    else dst[e] = re;
  }
}


// ---------------------------------------------------------------------------
// device-resident GMRES building blocks (reference caller of the apply path:
// bfSolveGMRES, src/linalg.c:47-317: residual :127-131, column norms :139,
// modified Gram-Schmidt :174-184, normalisation :197-198, solution :245-285)
// ---------------------------------------------------------------------------
#define BF_GM_THREADS 256

__device__ __forceinline__ double2 bfBlockReduce2(double2 v, double2 *sh) {
  // fixed-order tree over the 256 threads of the block
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = BF_GM_THREADS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { sh[threadIdx.x].x += sh[threadIdx.x + s].x; sh[threadIdx.x].y += sh[threadIdx.x + s].y; }
    __syncthreads();
  }
  double2 r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ void bfRowRange(uint64_t n, uint32_t nb, uint64_t &r0, uint64_t &r1) {
  uint64_t per = (n + nb - 1) / nb;
  r0 = (uint64_t)blockIdx.x * per;
  r1 = r0 + per < n ? r0 + per : n;
  if (r0 > n) r0 = n;
}

// W = B - AX0 (AX0 may be null); partialOut[q*nb + bx] = sum |W|^2 over the block's rows
__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresResidualKernel(double2 const *B, double2 const *AX0, double2 *W,
                                                                      double2 *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const q = blockIdx.y;
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  double acc = 0.0;
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 v = B[r * nrhs + q];
    if (AX0) { double2 y = AX0[r * nrhs + q]; v.x -= y.x; v.y -= y.y; }
    W[r * nrhs + q] = v;
    acc += v.x * v.x + v.y * v.y;
  }
  double2 t = bfBlockReduce2(make_double2(acc, 0.0), sh);
  if (threadIdx.x == 0) partialOut[(uint64_t)q * nb + blockIdx.x] = t;
}

__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresDotKernel(double2 const *Vi, double2 const *W, double2 *partialOut,
                                                                 uint64_t n, uint32_t nrhs, uint32_t nb) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const q = blockIdx.y;
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  double ar = 0.0, ai = 0.0;
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 v = Vi[r * nrhs + q], w = W[r * nrhs + q];
    ar += v.x * w.x + v.y * w.y;      // conj(v) * w
    ai += v.x * w.y - v.y * w.x;
  }
  double2 t = bfBlockReduce2(make_double2(ar, ai), sh);
  if (threadIdx.x == 0) partialOut[(uint64_t)q * nb + blockIdx.x] = t;
}

__device__ __forceinline__ double2 bfSumPartials(double2 const *partial, uint32_t q, uint32_t nb, double2 *sh) {
  double2 a = make_double2(0.0, 0.0);
  for (uint32_t b = threadIdx.x; b < nb; b += BF_GM_THREADS) { double2 v = partial[(uint64_t)q * nb + b]; a.x += v.x; a.y += v.y; }
  return bfBlockReduce2(a, sh);
}

__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresMgsKernel(double2 const *Vi, double2 const *Vnext, double2 *W,
                                                                 double2 const *partialIn, double2 *partialOut, double2 *hOut,
                                                                 uint64_t n, uint32_t nrhs, uint32_t nb) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const q = blockIdx.y;
  double2 const h = bfSumPartials(partialIn, q, nb, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0) hOut[q] = h;
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  double ar = 0.0, ai = 0.0;
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 v = Vi[r * nrhs + q], w = W[r * nrhs + q];
    w.x -= h.x * v.x - h.y * v.y;
    w.y -= h.x * v.y + h.y * v.x;
    W[r * nrhs + q] = w;
    if (Vnext) {
      double2 u = Vnext[r * nrhs + q];
      ar += u.x * w.x + u.y * w.y;
      ai += u.x * w.y - u.y * w.x;
    } else {
      ar += w.x * w.x + w.y * w.y;
    }
  }
  double2 t = bfBlockReduce2(make_double2(ar, ai), sh);
  if (threadIdx.x == 0) partialOut[(uint64_t)q * nb + blockIdx.x] = t;
}

__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresFinishKernel(double2 const *W, double2 const *partialIn, double2 *Vout,
                                                                    double2 *hOut, uint64_t n, uint32_t nrhs, uint32_t nb) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const q = blockIdx.y;
  double2 const s = bfSumPartials(partialIn, q, nb, sh);
  double const nrm = sqrt(s.x);
  if (blockIdx.x == 0 && threadIdx.x == 0) hOut[q] = make_double2(nrm, 0.0);
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 w = W[r * nrhs + q];
    Vout[r * nrhs + q] = make_double2(w.x / nrm, w.y / nrm);
  }
}

__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresUpdateKernel(double2 const *X0, double2 const *V, double2 const *y, uint32_t j,
                                                                    double2 *X, uint64_t n, uint32_t nrhs) {
  uint64_t e = (uint64_t)blockIdx.x * BF_GM_THREADS + threadIdx.x;
  uint64_t total = n * nrhs;
  if (e >= total) return;
  uint32_t q = (uint32_t)(e % nrhs);
  double2 x = X0 ? X0[e] : make_double2(0.0, 0.0);
  for (uint32_t i = 0; i < j; ++i) {
    double2 v = V[(uint64_t)i * total + e], c = y[(uint64_t)i * nrhs + q];
    x.x += v.x * c.x - v.y * c.y;
    x.y += v.x * c.y + v.y * c.x;
  }
  X[e] = x;
}

// ---- batched Gram-Schmidt (CGS2): all projections of an iteration in one launch --------------------
// The reference orthogonalises W against V_0..V_j one vector at a time (modified Gram-Schmidt,
// src/linalg.c:174-184): j + 1 dependent BLAS-1 passes, each a few microseconds of work behind a launch.
// Here one pass is three launches whatever j is -- all dots h_i = V_i^H W (W read once per group of 8
// basis vectors), their reduction over row blocks, and W -= sum_i h_i V_i -- and the pass is run twice
// (classical Gram-Schmidt with reorthogonalisation, as stable as MGS); H[:, j] = h(pass 1) + h(pass 2).
#define BF_GM_GROUP 8

__device__ __forceinline__ double bfWaveSum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);      // fixed butterfly: deterministic
  return v;
}

// partial[((q * numVec + i) * nb) + bx] = sum over the block's rows of conj(V_i) * W   (i in this group of 8)
__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresDotsKernel(double2 const *V, double2 const *W, double2 *partial, uint64_t n,
                                                                  uint32_t nrhs, uint32_t nb, uint32_t numVec) {
  __shared__ double2 sh[BF_GM_THREADS / 64][BF_GM_GROUP];
  uint32_t const q = blockIdx.y, i0 = blockIdx.z * BF_GM_GROUP;
  uint32_t const cnt = numVec - i0 < BF_GM_GROUP ? numVec - i0 : BF_GM_GROUP;
  uint64_t const vecLen = n * nrhs;
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  double ar[BF_GM_GROUP], ai[BF_GM_GROUP];
#pragma unroll
  for (int k = 0; k < BF_GM_GROUP; ++k) ar[k] = ai[k] = 0.0;
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 const w = W[r * nrhs + q];
#pragma unroll
    for (int k = 0; k < BF_GM_GROUP; ++k)
      if ((uint32_t)k < cnt) {
        double2 const v = V[(uint64_t)(i0 + k) * vecLen + r * nrhs + q];
        ar[k] += v.x * w.x + v.y * w.y;
        ai[k] += v.x * w.y - v.y * w.x;
      }
  }
  int const wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < BF_GM_GROUP; ++k) {
    double const sr = bfWaveSum(ar[k]), si = bfWaveSum(ai[k]);
    if (lane == 0) sh[wave][k] = make_double2(sr, si);
  }
  __syncthreads();
  if (threadIdx.x < cnt) {
    double2 t = sh[0][threadIdx.x];
    for (int w2 = 1; w2 < BF_GM_THREADS / 64; ++w2) { t.x += sh[w2][threadIdx.x].x; t.y += sh[w2][threadIdx.x].y; }
    partial[((uint64_t)q * numVec + i0 + threadIdx.x) * nb + blockIdx.x] = t;
  }
}

// h[i * nrhs + q] = sum_b partial; hSum[i * nrhs + q] = h + (hPrev ? hPrev[i * nrhs + q] : 0)
__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresDotsFinishKernel(double2 const *partial, double2 const *hPrev, double2 *h, double2 *hSum,
                                                                        uint32_t nrhs, uint32_t nb, uint32_t numVec) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const i = blockIdx.x, q = blockIdx.y;
  double2 a = make_double2(0.0, 0.0);
  for (uint32_t b = threadIdx.x; b < nb; b += BF_GM_THREADS) { double2 v = partial[((uint64_t)q * numVec + i) * nb + b]; a.x += v.x; a.y += v.y; }
  double2 const t = bfBlockReduce2(a, sh);
  if (threadIdx.x == 0) {
    h[(uint64_t)i * nrhs + q] = t;
    if (hSum) { double2 p = hPrev ? hPrev[(uint64_t)i * nrhs + q] : make_double2(0.0, 0.0); hSum[(uint64_t)i * nrhs + q] = make_double2(t.x + p.x, t.y + p.y); }
  }
}

// W -= sum_i h_i V_i; partialOut (optional) = per-block sum |W|^2 of the result
__global__ __launch_bounds__(BF_GM_THREADS) void bfGmresProjectKernel(double2 const *V, double2 *W, double2 const *h, double2 *partialOut, uint64_t n,
                                                                     uint32_t nrhs, uint32_t nb, uint32_t numVec) {
  __shared__ double2 sh[BF_GM_THREADS];
  uint32_t const q = blockIdx.y;
  uint64_t const vecLen = n * nrhs;
  uint64_t r0, r1;
  bfRowRange(n, nb, r0, r1);
  double acc = 0.0;
  for (uint64_t r = r0 + threadIdx.x; r < r1; r += BF_GM_THREADS) {
    double2 w = W[r * nrhs + q];
    for (uint32_t i = 0; i < numVec; ++i) {
      double2 const v = V[(uint64_t)i * vecLen + r * nrhs + q], c = h[(uint64_t)i * nrhs + q];   // h: uniform, cached
      w.x -= c.x * v.x - c.y * v.y;
      w.y -= c.x * v.y + c.y * v.x;
    }
    W[r * nrhs + q] = w;
    acc += w.x * w.x + w.y * w.y;
  }
  if (partialOut) {
    double2 const t = bfBlockReduce2(make_double2(acc, 0.0), sh);
    if (threadIdx.x == 0) partialOut[(uint64_t)q * nb + blockIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------
// host-callable wrappers
// ---------------------------------------------------------------------------
extern "C" {

int bfdevSetDevice(int device) {
  if (device < 0) return 0;
  return hipFail(hipSetDevice(device), "hipSetDevice");
}
int bfdevGetDevice(int *device) { return hipFail(hipGetDevice(device), "hipGetDevice"); }
int bfdevMalloc(void **p, size_t bytes) {
  *p = NULL;
  if (!bytes) bytes = 16;
  return hipFail(hipMalloc(p, bytes), "hipMalloc");
}
void bfdevFree(void *p) { if (p) (void)hipFree(p); }
int bfdevMemcpyH2D(void *dst, void const *src, size_t bytes) { return bytes ? hipFail(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D") : 0; }
int bfdevMemcpyD2H(void *dst, void const *src, size_t bytes) { return bytes ? hipFail(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H") : 0; }
int bfdevMemset(void *dst, int value, size_t bytes) { return bytes ? hipFail(hipMemset(dst, value, bytes), "hipMemset") : 0; }
int bfdevSync(void *stream) { return hipFail(hipStreamSynchronize((hipStream_t)stream), "hipStreamSynchronize"); }
// What a caller's pointer is: 0 plain (pageable / unknown) host memory, 1 device memory of the CURRENT device, 2 host memory the
// runtime can DMA from / to directly (hipHostMalloc'd or hipHostRegister'ed), 3 device memory of another device.
int bfdevPointerKind(void const *p) {
  hipPointerAttribute_t a;
  hipError_t const e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return 0; }      // older runtimes: an error for unregistered memory
  if (a.type == hipMemoryTypeDevice) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    return a.device == cur ? 1 : 3;
  }
  if (a.type == hipMemoryTypeHost) return 2;
  return 0;      // unregistered, managed: staged like pageable memory
}
int bfdevHostRegister(void *p, size_t bytes) { return hipFail(hipHostRegister(p, bytes, hipHostRegisterDefault), "hipHostRegister"); }
int bfdevHostUnregister(void *p) { return hipFail(hipHostUnregister(p), "hipHostUnregister"); }
int bfdevMemcpyAnyAsync(void *dst, void const *src, size_t bytes, void *stream) { return bytes ? hipFail(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream), "hipMemcpyAsync") : 0; }
int bfdevHostAllocPinned(void **p, size_t bytes) { return hipFail(hipHostMalloc(p, bytes, hipHostMallocDefault), "hipHostMalloc"); }
void bfdevHostFreePinned(void *p) { if (p) (void)hipHostFree(p); }

int bfdevSynthFill(void *arena, uint32_t dtype, BfSynthPiece const *hostPieces, uint64_t count, uint64_t seed) {
  if (!count) return 0;
  BfSynthPiece *d = NULL;
  int rc = hipFail(hipMalloc((void **)&d, count * sizeof(BfSynthPiece)), "hipMalloc(synth pieces)");
  if (rc) return rc;
  rc = hipFail(hipMemcpy(d, hostPieces, count * sizeof(BfSynthPiece), hipMemcpyHostToDevice), "hipMemcpy(synth pieces)");
  if (!rc) {
    // a launch's global size (blocks x 256 threads) is a 32-bit count of work-items: at most 2^24 - 1 blocks of 256, or the
    // launch silently covers a truncated grid (found at 18 M pieces: the packed adjoint of the N = 1M streamed operand)
    uint64_t done = 0;
    while (done < count && !rc) {
      uint32_t n = (uint32_t)((count - done) > (1u << 23) ? (1u << 23) : (count - done));
      if (dtype == BFHIP_C128) hipLaunchKernelGGL(bfSynthKernel<BFHIP_C128>, dim3(n), dim3(256), 0, 0, arena, d + done, seed);
      else if (dtype == BFHIP_F64) hipLaunchKernelGGL(bfSynthKernel<BFHIP_F64>, dim3(n), dim3(256), 0, 0, arena, d + done, seed);
      else hipLaunchKernelGGL(bfSynthKernel<BFHIP_F32>, dim3(n), dim3(256), 0, 0, arena, d + done, seed);
      rc = hipFail(hipGetLastError(), "synth fill launch");
      done += n;
    }
    if (!rc) rc = hipFail(hipDeviceSynchronize(), "synth fill");
  }
  (void)hipFree(d);
  return rc;
}

int bfdevLaunchStage(BfLaunchArgs const *a, void *stream) {
  if (!a->numItems) return 0;
  /* a launch's global size is a 32-bit count of work-items; the widest mapping here is one 64-lane wavefront per item */
  if (a->numItems >= (1ull << 32) / 256) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "stage of %llu items exceeds the 32-bit global size of a launch", (unsigned long long)a->numItems);
  StageParams p;
  p.arena = a->arena;
  p.items = (BfDevItem const *)a->items;
  p.pieces = (BfDevPiece const *)a->pieces;
  p.numItems = (uint32_t)a->numItems;
  p.nrhs = a->nrhs;
  p.coopItems = 0;
  p.numBundles = (uint32_t)a->numBundles;
  p.bundles = (uint32_t const *)a->bundles;
  p.x = a->x;
  p.y = a->y;
  p.temp = a->temp;
  p.zero = a->zero;
  uint32_t grid = (uint32_t)((a->numItems + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG);
  hipStream_t s = (hipStream_t)stream;
  if (a->transposed) {
    // two ranges of items, two launches: [0, numNarrow) are <= 16 columns of tall leaves (16 row lanes, whole 256-byte
    // runs, streamed loads), the rest up to 64 columns of short pieces (4 row lanes) unless the stage has none wider
    // than 16.  In either range the leading big items of many pieces get a workgroup each -- except on the complex
    // 16-column kernel, which needs 98 VGPRs with the shared-item code (4 wavefronts per SIMD instead of 5) and loses
    // 5 % on fac_helm2's adjoint.
    uint64_t const numNarrow = a->numNarrow < a->numItems ? a->numNarrow : a->numItems;
    if (a->dtype != BFHIP_C128 && numNarrow && numNarrow < a->numItems && a->maxRowsRest > 16) {
      // real operands with both item families: one launch (bfStageKernelTBoth)
      StageParams pn = p, pw = p;
      uint64_t const cntW = a->numItems - numNarrow;
      uint64_t ncN = a->numCoopNarrow < numNarrow ? a->numCoopNarrow : numNarrow, ncW = a->numCoop < cntW ? a->numCoop : cntW;
      pn.items = (BfDevItem const *)a->items; pn.numItems = (uint32_t)numNarrow; pn.coopItems = (uint32_t)ncN;
      pw.items = (BfDevItem const *)a->items + numNarrow; pw.numItems = (uint32_t)cntW; pw.coopItems = (uint32_t)ncW;
      uint32_t const gridN = (uint32_t)(ncN + (numNarrow - ncN + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG);
      uint32_t const gridW = (uint32_t)(ncW + (cntW - ncW + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG);
#define BF_LAUNCH_TBOTH(DT) do { if (a->nrhs == 1) hipLaunchKernelGGL((bfStageKernelTBoth<DT, true>), dim3(gridN + gridW), dim3(BF_WAVES_PER_WG * 64), 0, s, pn, pw, gridN); \
                                 else hipLaunchKernelGGL((bfStageKernelTBoth<DT, false>), dim3(gridN + gridW), dim3(BF_WAVES_PER_WG * 64), 0, s, pn, pw, gridN); } while (0)
      if (a->dtype == BFHIP_F64) BF_LAUNCH_TBOTH(BFHIP_F64);
      else if (a->dtype == BFHIP_F32) BF_LAUNCH_TBOTH(BFHIP_F32);
      else return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unknown dtype %u", a->dtype);
#undef BF_LAUNCH_TBOTH
      return hipFail(hipGetLastError(), "transposed stage launch");
    }
    for (int range = 0; range < 2; ++range) {
      uint64_t const first = range ? numNarrow : 0, count = range ? a->numItems - numNarrow : numNarrow;
      if (!count) continue;
      bool const wide = range == 1 && a->maxRowsRest > 16;
      uint64_t nc = range ? a->numCoop : a->numCoopNarrow;
      if (nc > count) nc = count;
      if (a->dtype == BFHIP_C128 && !wide) nc = 0;
      p.items = (BfDevItem const *)a->items + first;
      p.numItems = (uint32_t)count;
      p.coopItems = (uint32_t)nc;
      grid = (uint32_t)(nc + (count - nc + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG);
#define BF_LAUNCH_T1(DT, ONE) do { if (wide && nc) hipLaunchKernelGGL((bfStageKernelT<DT, BF_T_WIDE_R, BF_T_WIDE_R, true, ONE>), dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p); \
                             else if (wide) hipLaunchKernelGGL((bfStageKernelT<DT, BF_T_WIDE_R, BF_T_WIDE_R, false, ONE>), dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p); \
                             else if (nc) hipLaunchKernelGGL((bfStageKernelT<DT, 16, 4, true, ONE>), dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p); \
                             else hipLaunchKernelGGL((bfStageKernelT<DT, 16, 4, false, ONE>), dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p); } while (0)
#define BF_LAUNCH_T(DT) do { if (a->nrhs == 1) BF_LAUNCH_T1(DT, true); else BF_LAUNCH_T1(DT, false); } while (0)
      if (a->dtype == BFHIP_C128) BF_LAUNCH_T(BFHIP_C128);
      else if (a->dtype == BFHIP_F64) BF_LAUNCH_T(BFHIP_F64);
      else if (a->dtype == BFHIP_F32) BF_LAUNCH_T(BFHIP_F32);
      else return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unknown dtype %u", a->dtype);
#undef BF_LAUNCH_T
#undef BF_LAUNCH_T1
    }
    return hipFail(hipGetLastError(), "transposed stage launch");
  }
  if (a->dtype == BFHIP_C128 && a->nrhs >= BF_MFMA_MIN_RHS) {
    dim3 const g((uint32_t)((a->numItems + BF_MF_WG_WAVES - 1) / BF_MF_WG_WAVES)), b(64 * BF_MF_WG_WAVES);
    /* more than 32 right-hand sides: one workgroup of four wavefronts per BUNDLE of items that read the same X rows */
    if (a->nrhs > 32 && BF_MF_BUNDLES && (!a->bundles || !a->numBundles)) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: a forward complex128 stage without its bundle table");
    dim3 const gB(BF_MF_BUNDLES ? (uint32_t)a->numBundles : g.x), bB(BF_MF_BUNDLES ? 256u : b.x);
    if (a->exactComplex) {                                                                   /* BFHIP_FLAG_EXACT_COMPLEX: four real products per complex one */
      if (a->nrhs <= 16) hipLaunchKernelGGL(bfStageKernelC128Mfma1Exact, g, b, 0, s, p);
      else if (a->nrhs <= 32) hipLaunchKernelGGL(bfStageKernelC128Mfma2Exact, g, b, 0, s, p);
      else hipLaunchKernelGGL(bfStageKernelC128MfmaExact, gB, bB, 0, s, p);
    }
    else if (a->nrhs <= 16) hipLaunchKernelGGL(bfStageKernelC128Mfma1, g, b, 0, s, p);       /* one RHS tile: 5 wavefronts per SIMD */
    else if (a->nrhs <= 32) hipLaunchKernelGGL(bfStageKernelC128Mfma2, g, b, 0, s, p);       /* two: 3 */
    else hipLaunchKernelGGL(bfStageKernelC128Mfma, gB, bB, 0, s, p);                         /* up to four per pass: 2 */
  }
  else if (a->dtype == BFHIP_C128) {
    grid = (uint32_t)((a->numItems + BF_C128_WG_WAVES - 1) / BF_C128_WG_WAVES);
#ifdef BFHIP_EXPERIMENTAL
    { int handled = 0; int const rcx = bfdevLaunchStageExperimental(a, &p, grid, stream, &handled); if (handled) return rcx; }     /* EXPERIMENTAL=1 builds only */
#endif
    hipLaunchKernelGGL(bfStageKernelC128, dim3(grid), dim3(BF_C128_WG_WAVES * 64), 0, s, p);
  }
  else if (a->dtype == BFHIP_F64 || a->dtype == BFHIP_F32) {
    // items [firstSmall, numItems) are small (BF_ITEM_SMALL): their own launch, four to a wavefront; the two launches
    // write disjoint rows
    uint64_t const firstSmall = a->firstSmall < a->numItems ? a->firstSmall : a->numItems;
    uint64_t const numSmall = a->numItems - firstSmall;
    p.numItems = (uint32_t)firstSmall;
    grid = (uint32_t)((firstSmall + BF_WAVES_PER_WG - 1) / BF_WAVES_PER_WG);
    if (grid && numSmall) {
      StageParams ps = p;
      ps.items = (BfDevItem const *)a->items + firstSmall;
      ps.numItems = (uint32_t)numSmall;
      uint32_t const gridSmall = (uint32_t)((numSmall + 4 * BF_WAVES_PER_WG - 1) / (4 * BF_WAVES_PER_WG));
      if (a->dtype == BFHIP_F64) hipLaunchKernelGGL(bfStageKernelRealBoth<BFHIP_F64>, dim3(grid + gridSmall), dim3(BF_WAVES_PER_WG * 64), 0, s, p, ps, grid);
      else hipLaunchKernelGGL(bfStageKernelRealBoth<BFHIP_F32>, dim3(grid + gridSmall), dim3(BF_WAVES_PER_WG * 64), 0, s, p, ps, grid);
      return hipFail(hipGetLastError(), "stage launch");
    }
    if (grid) {
      if (a->dtype == BFHIP_F64) hipLaunchKernelGGL(bfStageKernelReal<BFHIP_F64>, dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p);
      else hipLaunchKernelGGL(bfStageKernelReal<BFHIP_F32>, dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p);
    }
    if (numSmall) {
      p.items = (BfDevItem const *)a->items + firstSmall;
      p.numItems = (uint32_t)numSmall;
      grid = (uint32_t)((numSmall + 4 * BF_WAVES_PER_WG - 1) / (4 * BF_WAVES_PER_WG));
      if (a->dtype == BFHIP_F64) hipLaunchKernelGGL(bfStageKernelSmall<BFHIP_F64>, dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p);
      else hipLaunchKernelGGL(bfStageKernelSmall<BFHIP_F32>, dim3(grid), dim3(BF_WAVES_PER_WG * 64), 0, s, p);
    }
  }
  else return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unknown dtype %u", a->dtype);
  return hipFail(hipGetLastError(), "stage launch");
}


int bfdevMemsetAsync(void *dst, int value, size_t bytes, void *stream) { return bytes ? hipFail(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), "hipMemsetAsync") : 0; }

int bfdevScalePermute(void *dst, void const *src, void const *scale, int power, uint64_t const *perm, uint64_t n, uint32_t dtype, void *stream) {
  if (!n) return 0;
  uint32_t const grid = (uint32_t)((n + 255) / 256);
  if (dtype == BFHIP_F64) hipLaunchKernelGGL(bfScalePermuteKernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)dst, (double const *)src, (double const *)scale, power, perm, n);
  else if (dtype == BFHIP_F32) hipLaunchKernelGGL(bfScalePermuteKernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)dst, (float const *)src, (float const *)scale, power, perm, n);
  else return bfhipFail(BFABI_ERROR_TYPE_ERROR, "scale/permute: real element types only");
  return hipFail(hipGetLastError(), "scale/permute launch");
}

int bfdevLaunchReduce(BfReduceArgs const *a, uint32_t count, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  for (uint32_t base = 0; base < count; base += BF_REDUCE_BATCH) {
    ReduceBatch B;
    B.count = count - base < BF_REDUCE_BATCH ? count - base : BF_REDUCE_BATCH;
    B.nrhs = a[base].nrhs;
    B.temp = a[base].temp;
    uint32_t blocks = 0;
    for (uint32_t k = 0; k < B.count; ++k) {
      BfReduceArgs const *r = &a[base + k];
      B.e[k].rowInterval = (uint32_t const *)r->rowInterval; B.e[k].ivBegin = (uint32_t const *)r->ivBegin;
      B.e[k].srcBias = (int64_t const *)r->srcBias; B.e[k].numRows = r->numRows; B.e[k].dest = r->dest;
      B.e[k].blockBegin = blocks; B.e[k].pad = 0;
      blocks += (uint32_t)((r->numRows * r->nrhs + 255) / 256);
    }
    if (!blocks) continue;
    uint32_t const dtype = a[base].dtype;
    bool longLists = false;
    for (uint32_t k = 0; k < B.count; ++k) longLists = longLists || a[base + k].longLists;
    if (dtype == BFHIP_C128) hipLaunchKernelGGL((bfReduceKernel<double, 2>), dim3(blocks), dim3(256), 0, s, B);
    else if (dtype == BFHIP_F64 && longLists) hipLaunchKernelGGL((bfReduceKernel<double, 1, true>), dim3(blocks), dim3(256), 0, s, B);
    else if (dtype == BFHIP_F64) hipLaunchKernelGGL((bfReduceKernel<double, 1>), dim3(blocks), dim3(256), 0, s, B);
    else if (longLists) hipLaunchKernelGGL((bfReduceKernel<float, 1, true>), dim3(blocks), dim3(256), 0, s, B);
    else hipLaunchKernelGGL((bfReduceKernel<float, 1>), dim3(blocks), dim3(256), 0, s, B);
    int rc = hipFail(hipGetLastError(), "reduce launch");
    if (rc) return rc;
  }
  return 0;
}

int bfdevGmresResidual(void const *B, void const *AX0, void *W, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream) {
  hipLaunchKernelGGL(bfGmresResidualKernel, dim3(nb, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)B, (double2 const *)AX0, (double2 *)W, (double2 *)partialOut, n, nrhs, nb);
  return hipFail(hipGetLastError(), "gmres residual launch");
}
int bfdevGmresDot(void const *Vi, void const *W, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream) {
  hipLaunchKernelGGL(bfGmresDotKernel, dim3(nb, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)Vi, (double2 const *)W, (double2 *)partialOut, n, nrhs, nb);
  return hipFail(hipGetLastError(), "gmres dot launch");
}
int bfdevGmresMgsStep(void const *Vi, void const *Vnext, void *W, void const *partialIn, void *partialOut, void *hOut,
                      uint64_t n, uint32_t nrhs, uint32_t nb, void *stream) {
  hipLaunchKernelGGL(bfGmresMgsKernel, dim3(nb, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)Vi, (double2 const *)Vnext, (double2 *)W, (double2 const *)partialIn, (double2 *)partialOut, (double2 *)hOut, n, nrhs, nb);
  return hipFail(hipGetLastError(), "gmres mgs launch");
}
int bfdevGmresFinish(void const *W, void const *partialIn, void *Vout, void *hOut, uint64_t n, uint32_t nrhs, uint32_t nb, void *stream) {
  hipLaunchKernelGGL(bfGmresFinishKernel, dim3(nb, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)W, (double2 const *)partialIn, (double2 *)Vout, (double2 *)hOut, n, nrhs, nb);
  return hipFail(hipGetLastError(), "gmres finish launch");
}
int bfdevGmresDots(void const *V, void const *W, void *partial, uint64_t n, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream) {
  hipLaunchKernelGGL(bfGmresDotsKernel, dim3(nb, nrhs, (numVec + BF_GM_GROUP - 1) / BF_GM_GROUP), dim3(BF_GM_THREADS), 0, (hipStream_t)stream,
                     (double2 const *)V, (double2 const *)W, (double2 *)partial, n, nrhs, nb, numVec);
  return hipFail(hipGetLastError(), "gmres dots launch");
}
int bfdevGmresDotsFinish(void const *partial, void const *hPrev, void *h, void *hSum, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream) {
  hipLaunchKernelGGL(bfGmresDotsFinishKernel, dim3(numVec, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)partial,
                     (double2 const *)hPrev, (double2 *)h, (double2 *)hSum, nrhs, nb, numVec);
  return hipFail(hipGetLastError(), "gmres dots-finish launch");
}
int bfdevGmresProject(void const *V, void *W, void const *h, void *partialOut, uint64_t n, uint32_t nrhs, uint32_t nb, uint32_t numVec, void *stream) {
  hipLaunchKernelGGL(bfGmresProjectKernel, dim3(nb, nrhs), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)V, (double2 *)W,
                     (double2 const *)h, (double2 *)partialOut, n, nrhs, nb, numVec);
  return hipFail(hipGetLastError(), "gmres project launch");
}
int bfdevGmresUpdate(void const *X0, void const *V, void const *y, uint32_t j, void *X, uint64_t n, uint32_t nrhs, void *stream) {
  uint64_t total = n * nrhs;
  hipLaunchKernelGGL(bfGmresUpdateKernel, dim3((uint32_t)((total + BF_GM_THREADS - 1) / BF_GM_THREADS)), dim3(BF_GM_THREADS), 0, (hipStream_t)stream, (double2 const *)X0, (double2 const *)V, (double2 const *)y, j, (double2 *)X, n, nrhs);
  return hipFail(hipGetLastError(), "gmres update launch");
}
int bfdevMemcpyD2HAsync(void *dst, void const *src, size_t bytes, void *stream) { return bytes ? hipFail(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), "hipMemcpyAsync D2H") : 0; }
int bfdevMemcpyH2DAsync(void *dst, void const *src, size_t bytes, void *stream) { return bytes ? hipFail(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), "hipMemcpyAsync H2D") : 0; }
int bfdevMemcpyD2DAsync(void *dst, void const *src, size_t bytes, void *stream) { return bytes ? hipFail(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream), "hipMemcpyAsync D2D") : 0; }

int bfdevEventCreate(void **ev) { return hipFail(hipEventCreate((hipEvent_t *)ev), "hipEventCreate"); }
void bfdevEventDestroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int bfdevEventRecord(void *ev, void *stream) { return hipFail(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream), "hipEventRecord"); }
int bfdevEventSync(void *ev) { return hipFail(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize"); }
int bfdevEventElapsed(void *start, void *stop, float *ms) {
  hipError_t e = hipEventSynchronize((hipEvent_t)stop);
  if (e != hipSuccess) return hipFail(e, "hipEventSynchronize");
  return hipFail(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
}

}  // extern "C"
