// bfhip_stage_mfma.h -- complex128 stage kernel for blocks of right-hand sides (nrhs >= 2): the same items and the
// same packed pieces as bfStageKernelC128, contracted on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).  It replaces
// the cblas_zgemm of every leaf of a level (reference src/mat_dense_complex.c:1704-1765) and the view / accumulate
// passes around it (src/mat_block_coo.c:404-418, src/mat_block_diag.c:387-399).  Included by bfhip_device.hip only.
//
// Shape of the work.  An item is <= 64 rows of one output row group; its dense pieces are stored back to back,
// column-major with the item's row count as the column stride, so the item's leaf data is ONE column-major
// mr x C matrix (C = the sum of the pieces' widths) -- a *segment* -- whose column c multiplies one row of the
// input vector.  The kernel writes tab[] (that row's byte offset, 32 bits per column) into LDS once per segment and then runs
// one flat k-loop over the segment: piece boundaries do not exist inside the loop.  (A segment ends where the next
// piece is not contiguous, reads another vector (x / intermediates), or the table is full; fac_helm2 items are one
// segment, two in the last stage.)
//
// One pass = MS 16-row slabs x NT 16-RHS tiles: 4 * MS * NT MFMAs per k-step (4 leaf columns).  Fragment maps
// (cdna_hip_programming.md section 3): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D reg v of lane l =
// D[i = 4 v + (l >> 4)][j = l & 15]; a complex multiply-accumulate is 4 real MFMAs, the -Ai*Bi term through the NEG
// bit of the instruction (blgp bit 0 negates A on the f64 MFMA: tools/mfma_probe.hip).
//
// What the inner loop is made of, and why (tools/mfma_probe.hip, profiles/r4_mfma_probe.json):
//   * ONE wavefront cannot keep the FP64 matrix pipe busy (0.75 of peak with 16 independent accumulators), two or more
//     can (0.99): every cycle a wavefront spends on anything else costs a quarter of the pipe, so the loop holds no
//     VALU address arithmetic, no selects and no register copies.  Fragments come through buffer loads -- a loop
//     invariant VGPR offset, the k-step advance in an SGPR, columns / rows past the end of the segment returned as
//     zeros by the range check (which includes the SGPR offset) instead of clamps + selects.
//   * ONE set of fragment registers: a fragment is requested again right after the last MFMA that reads it (A of slab
//     0 after the first half of the k-step, X tile t inside the second half, A of slab 1 at the end), half a k-step
//     ahead of its next use; 128 accumulator + 24 fragment registers leave room for THREE wavefronts per SIMD, so a
//     wavefront between items (table, stores) leaves two on the pipe.
//   * The loads are asm statements and the waits are placed by hand: hipcc's wait insertion treats every load pending
//     at a loop header as one lump (s_waitcnt vmcnt(0) at the top: the fragment requested last would be waited for
//     first).  Loads return in order, so vmcnt(n) = "all but the n youngest have arrived".
#ifndef BFHIP_STAGE_MFMA_H
#define BFHIP_STAGE_MFMA_H

#define BF_MF_TABCAP 1792u          /* columns of one segment: 7.1 KiB of LDS per wavefront -- 20 wavefronts per CU (the 1-tile instantiation) fit 160 KiB */
#define BF_MF_TABPAD 24u
#define BF_MF_SPAN_BYTES (1u << 31) /* a segment's input rows span less than this many bytes (32-bit buffer offsets) */
#ifndef BF_MFMA_WAVES_PER_SIMD
#define BF_MFMA_WAVES_PER_SIMD 2
#endif
#ifndef BF_MF_XCD_RUN
#define BF_MF_XCD_RUN 32u           /* workgroups */
#endif
#ifndef BF_MF_WG_WAVES
#define BF_MF_WG_WAVES 1u
#endif
#ifndef BF_MFMA_MIN_RHS
#define BF_MFMA_MIN_RHS 2
#endif

typedef double bf_d4 __attribute__((ext_vector_type(4)));
typedef int bf_i4 __attribute__((ext_vector_type(4)));
union BfFrag { bf_u4 u; double d[2]; };

// a raw buffer descriptor in SGPRs: 48-bit base, stride 0, num_records in bytes
__device__ __forceinline__ bf_i4 bfMakeRsrc(void const *base, uint32_t bytes) {
  uint64_t const b = (uint64_t)base;
  bf_i4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}
template <int STREAM, int OFF> __device__ __forceinline__ void bfFragLoad(BfFrag &f, uint32_t voff, bf_i4 rsrc, uint32_t soff) {
  if (STREAM) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 nt" : "=v"(f.u) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(f.u) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
}
template <int N> __device__ __forceinline__ void bfFragWait(BfFrag &f) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(f.u) : "n"(N)); }
template <int N> __device__ __forceinline__ void bfFragWait(BfFrag &f, BfFrag &g) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f.u), "+v"(g.u) : "n"(N)); }

// What a segment's k-loop needs besides the table.
struct BfMfSeg {
  bf_i4 ra, rx;          // leaf matrix (mr x cols, column-major), input rows [minRow, maxRow] x the pass's RHS
  uint32_t voffA;        // lane: (column lk of a k-step, row s0 + li); slab 1 is +256 bytes
  uint32_t cX;           // lane: li * 16; + tab[column] = byte offset of the lane's X fragment
  uint32_t stepA;        // 4 columns = 4 * mr * 16 bytes
  uint32_t ksteps;
};

// The k-loop of one segment: MS slabs x NT tiles.
//
// Complex products by Gauss's three multiplications: with T1 = Ar Xr, T2 = Ai Xi, T3 = (Ar + Ai)(Xr + Xi) summed over
// the segment, Re = T1 - T2 and Im = T3 - T1 - T2 -- 3 real MFMAs per complex multiply-accumulate instead of 4, paid
// for with MS + NT v_add_f64 per k-step (the fragment sums) and a third accumulator per tile (24 x 8 = 192 VGPRs: two
// wavefronts per SIMD still fit).  The kernel is power-bound, not issue-bound (MFMA-busy x clock is
// constant across every variant of this loop), so a quarter fewer MFMAs is a quarter less time.  Normwise as accurate as
// the four-multiplication form (the imaginary part loses relative accuracy only where it is small next to |A||X|).
//
// Registers and requests.  A fragments (streamed from HBM, the long latency) have TWO sets: the A of k-step ks + 1 is
// requested at the top of k-step ks.  X fragments (L2) have ONE set: tile t is requested again right after the 3 MS
// MFMAs that read it, 3 MS (NT - 1) MFMAs before its next use (tile-outer, slab-inner order).  Requests return in order,
// so in front of every tile "all but the NT - 1 + MS youngest have arrived" is the wait (s_waitcnt vmcnt).
// An odd number of k-steps ends with a k-step of zeros (the leaf fragments past the end of the segment).
template <int MS, int SET>
__device__ __forceinline__ void bfMfmaRequestA(BfFrag (&a)[2][2], BfMfSeg const &sg, uint32_t soffA) {
  bfFragLoad<1, 0>(a[SET][0], sg.voffA, sg.ra, soffA);
  if (MS > 1) bfFragLoad<1, 256>(a[SET][1], sg.voffA, sg.ra, soffA);
}
template <int T>
__device__ __forceinline__ void bfMfmaRequestX(BfFrag (&x)[4], BfMfSeg const &sg, uint32_t voffX) {
  bfFragLoad<0, 256 * T>(x[T], voffX, sg.rx, 0);
}
template <int NT, int MS, int SET, int T>
__device__ __forceinline__ void bfMfmaTile(bf_d4 (&acc)[3][2][4], BfFrag (&a)[2][2], BfFrag (&x)[4], double (&as)[2], BfMfSeg const &sg, uint32_t voffXnext) {
  if (T < NT) {
    constexpr int TT = T < NT ? T : 0;
    constexpr int pending = NT - 1 + MS;
    if (T == 0 && MS > 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[SET][0].u), "+v"(a[SET][1].u), "+v"(x[0].u) : "n"(pending));
    else if (T == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[SET][0].u), "+v"(x[0].u) : "n"(pending));
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[TT].u) : "n"(pending));
    if (T == 0) {
      as[0] = a[SET][0].d[0] + a[SET][0].d[1];
      if (MS > 1) as[1] = a[SET][1].d[0] + a[SET][1].d[1];
    }
    double const xs = x[TT].d[0] + x[TT].d[1];
#pragma unroll
    for (int m = 0; m < MS; ++m) {
      acc[0][m][TT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][m].d[0], x[TT].d[0], acc[0][m][TT], 0, 0, 0);
      acc[1][m][TT] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][m].d[1], x[TT].d[1], acc[1][m][TT], 0, 0, 0);
      acc[2][m][TT] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], xs, acc[2][m][TT], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaRequestX<TT>(x, sg, voffXnext);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NT, int MS>
__device__ __forceinline__ void bfMfmaSegment(bf_d4 (&acc)[3][2][4], BfMfSeg const &sg, uint32_t const *tab, uint32_t lk) {
  BfFrag a[2][2], x[4];
  double as[2];
  uint32_t ti = lk;
  uint32_t soffA = 0;
  {
    uint32_t const v0 = tab[ti] + sg.cX;
    bfMfmaRequestA<MS, 0>(a, sg, soffA);
    bfMfmaRequestX<0>(x, sg, v0);
    if (NT > 1) bfMfmaRequestX<1>(x, sg, v0);
    if (NT > 2) bfMfmaRequestX<2>(x, sg, v0);
    if (NT > 3) bfMfmaRequestX<3>(x, sg, v0);
  }
  uint32_t t1 = tab[ti + 4], t2 = tab[ti + 8];      // read an iteration ahead of their use
  ti += 12;
  for (uint32_t ks = 0; ks < sg.ksteps; ks += 2) {
    uint32_t const v1 = t1 + sg.cX, v2 = t2 + sg.cX;
    t1 = tab[ti];                                    // the table is padded past the last k-step (BF_MF_TABPAD)
    t2 = tab[ti + 4];
    ti += 8;
    soffA += sg.stepA;
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaRequestA<MS, 1>(a, sg, soffA);
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaTile<NT, MS, 0, 0>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 1>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 2>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 3>(acc, a, x, as, sg, v1);
    soffA += sg.stepA;
    bfMfmaRequestA<MS, 0>(a, sg, soffA);
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaTile<NT, MS, 1, 0>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 1>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 2>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 3>(acc, a, x, as, sg, v2);
  }
  // the requests of the k-steps past the end (zeros from the range check / a padded table row) must land before the
  // registers are used again
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0][0].u), "+v"(a[0][MS > 1 ? 1 : 0].u), "+v"(x[0].u), "+v"(x[NT > 1 ? 1 : 0].u), "+v"(x[NT > 2 ? 2 : 0].u), "+v"(x[NT > 3 ? 3 : 0].u));
}

// One pass = rows [s0, s0 + 16 MS) x RHS [q0, q0 + 16 NT) of one item, over all its segments.
template <int NT, int MS>
__device__ __forceinline__ void bfMfmaPass(StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, uint32_t *tab, int lane) {
  uint32_t const nrhs = p.nrhs;
  uint32_t const li = lane & 15, lk = lane >> 4;
  bf_d4 acc[3][2][4];            // T1 = sum Ar Xr, T2 = sum Ai Xi, T3 = sum (Ar + Ai)(Xr + Xi)
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[g][m][t] = (bf_d4){0, 0, 0, 0};
  bool hasIdentity = false;
  uint32_t const np = it.numPieces;
  uint32_t const spanRows = BF_MF_SPAN_BYTES / (nrhs * 16u);
  uint32_t pi = 0;
  while (pi < np) {
    // ---- the next segment: its row table into LDS, its extent into scalars.  (The descriptor window is loaded
    // again for every segment: six registers that must not stay live across the k-loop.)
    uint32_t cols = 0, minRow = 0, maxRow = 0, inX = 0;
    uint64_t aOff = 0, expect = 0;
    bool started = false;
    BfPieceWin win;
    uint32_t wbase = 0xffffff00u;
    while (pi < np) {
      if (pi - wbase >= 64u) {
        wbase = pi;
        win = bfPieceWinLoad(p.pieces + it.pieceBegin + wbase, np - wbase < 64u ? np - wbase : 64u, lane);
      }
      BfDevPiece const pc = bfPieceWinGet(win, pi - wbase);
      if (pc.flags & BF_PIECE_IDENTITY) { hasIdentity = true; ++pi; continue; }
      uint32_t const px = pc.flags & BF_PIECE_IN_X, last = pc.inOff + pc.ncols - 1;
      uint32_t lo = pc.inOff, hi = last;
      if (started) {
        if (pc.dataOff != expect || px != inX || cols + pc.ncols > BF_MF_TABCAP) break;
        lo = minRow < lo ? minRow : lo;
        hi = maxRow > hi ? maxRow : hi;
        if (hi - lo >= spanRows) break;
      } else {
        started = true;
        aOff = pc.dataOff;
        inX = px;
      }
      minRow = lo;
      maxRow = hi;
      for (uint32_t j = (uint32_t)lane; j < pc.ncols; j += 64u) tab[cols + j] = pc.inOff + j;
      cols += pc.ncols;
      expect = pc.dataOff + (uint64_t)mr * pc.ncols;
      ++pi;
    }
    if (!cols) break;                      // identity pieces only
    // columns past the end: the leaf fragment is zero there (range check), any row of the segment will do
    if ((uint32_t)lane < BF_MF_TABPAD) tab[cols + lane] = minRow;
    waveSync();
    // rows -> byte offsets from the segment's first row (fits 32 bits: spanRows)
    for (uint32_t j = (uint32_t)lane; j < cols + BF_MF_TABPAD; j += 64u) tab[j] = (tab[j] - minRow) * (nrhs * 16u);
    waveSync();
    BfMfSeg sg;
    sg.stepA = 4u * mr * 16u;
    sg.ksteps = (cols + 3u) / 4u;
    sg.ra = bfMakeRsrc((double2 const *)p.arena + aOff, mr * cols * 16u);
    char const *xin = inX ? (char const *)p.x : (char const *)p.temp;
    // the last row ends with this pass's RHS (lanes past nrhs read zeros there, the next row's values elsewhere:
    // columns of the product that are never stored)
    sg.rx = bfMakeRsrc(xin + ((uint64_t)minRow * nrhs + q0) * 16u, ((maxRow - minRow) * nrhs + (nrhs - q0)) * 16u);
    sg.voffA = (lk * mr + s0 + li) * 16u;          // rows past the item's end alias the next column: rows of the product that are never stored
    sg.cX = li * 16u;
    bfMfmaSegment<NT, MS>(acc, sg, tab, lk);
    waveSync();                                      // the table is rewritten by the next segment
  }
  double2 *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (double2 *)p.y : (double2 *)p.temp;
  uint32_t const qmax = nrhs - 1 - q0;
  // (the lane's coordinates are derived again from an opaque copy: hipcc otherwise computes the store addresses
  // before the k-loop and carries them through it, which costs the third wavefront per SIMD)
  uint32_t lane2 = (uint32_t)lane;
  asm volatile("" : "+v"(lane2));
  uint32_t const li2 = lane2 & 15u, lk2 = lane2 >> 4;
#pragma unroll
  for (int m = 0; m < MS; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        uint32_t const row = s0 + 16u * m + 4 * v + lk2;
        if (row < mr && 16u * t + li2 <= qmax) {
          double re = acc[0][m][t][v] - acc[1][m][t][v], im = acc[2][m][t][v] - acc[0][m][t][v] - acc[1][m][t][v];
          if (hasIdentity) {      // rare (real-operand zoo; complex operands have none)
            for (uint32_t k = 0; k < np; ++k) {
              BfDevPiece const pc = p.pieces[it.pieceBegin + k];
              if (!(pc.flags & BF_PIECE_IDENTITY)) continue;
              double2 const *xin = (pc.flags & BF_PIECE_IN_X) ? (double2 const *)p.x : (double2 const *)p.temp;
              double2 xv = xin[((uint64_t)pc.inOff + row) * nrhs + q0 + 16 * t + li2];
              re += xv.x; im += xv.y;
            }
          }
          out[((uint64_t)it.outOff + row) * nrhs + q0 + 16 * t + li2] = make_double2(re, im);
        }
      }
  // stores and fragment requests share vmcnt: nothing of this pass may be pending when the next one counts
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int MS, int MAXNT>
__device__ __forceinline__ void bfMfmaDispatch(uint32_t nt, StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, uint32_t *tab, int lane) {
  if (MAXNT >= 4 && nt == 4) bfMfmaPass<4, MS>(p, it, mr, s0, q0, tab, lane);
  else if (MAXNT >= 3 && nt == 3) bfMfmaPass<3, MS>(p, it, mr, s0, q0, tab, lane);
  else if (MAXNT >= 2 && nt == 2) bfMfmaPass<2, MS>(p, it, mr, s0, q0, tab, lane);
  else bfMfmaPass<1, MS>(p, it, mr, s0, q0, tab, lane);
}

// MAXNT = the widest pass the launch needs (RHS tiles of 16): the accumulators of 4 tiles x 2 slabs x 3 products leave two
// wavefronts per SIMD, which is what the matrix pipe needs at 64 RHS -- but with 2 - 32 RHS the kernel is bound by the leaf
// stream, not by the pipe, and then it is wavefronts (bytes in flight) that count: the 1- and 2-tile instantiations need
// a third / half of the registers and run WAVES = 5 / 3 wavefronts per SIMD (N = 262144: 2 - 16 RHS 14.3 - 15.9 -> see
// DESIGN.md section 4).
template <int MAXNT, int WAVES>
__device__ __forceinline__ void bfStageBodyC128Mfma(StageParams const &p, uint32_t (*tabs)[BF_MF_TABCAP + BF_MF_TABPAD]) {
  int const lane = threadIdx.x & 63;
  uint32_t const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *tab = tabs[wave];
  // The wavefronts of a workgroup take neighbours of the item list: row chunks of one row group and of its sibling
  // groups (same cost, same input rows: the planner's order for RHS-block operators).  Workgroups are dealt to the 8 XCDs
  // round robin: runs of BF_MF_XCD_RUN workgroups that are neighbours in the list go to ONE XCD, so that a row of X is
  // fetched into that L2 once for all of them.
  uint32_t wg = blockIdx.x;
  uint32_t const numWg = (p.numItems + BF_MF_WG_WAVES - 1u) / BF_MF_WG_WAVES;
  if (BF_MF_XCD_RUN > 1) {
    uint32_t const blk = 8u * BF_MF_XCD_RUN;
    if (wg < numWg / blk * blk) { uint32_t const r = wg % blk; wg = wg - r + (r % 8u) * BF_MF_XCD_RUN + r / 8u; }
  }
  uint32_t const item = wg * BF_MF_WG_WAVES + wave;
  if (item >= p.numItems) return;
  BfDevItem const it = p.items[item];
  uint32_t const mr = it.mrFlags & 0xffffu;
  uint32_t const nrhs = p.nrhs;
  for (uint32_t q0 = 0; q0 < nrhs; q0 += 64) {
    uint32_t const nt = (nrhs - q0 >= 64) ? 4u : (nrhs - q0 + 15u) / 16u;
    uint32_t s0 = 0;
    while (s0 < mr) {
      if (mr - s0 > 16) { bfMfmaDispatch<2, MAXNT>(nt, p, it, mr, s0, q0, tab, lane); s0 += 32; }
      else { bfMfmaDispatch<1, MAXNT>(nt, p, it, mr, s0, q0, tab, lane); s0 += 16; }
    }
  }
}

__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(BF_MFMA_WAVES_PER_SIMD, BF_MFMA_WAVES_PER_SIMD))) void bfStageKernelC128Mfma(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<4, BF_MFMA_WAVES_PER_SIMD>(p, tabs);
}
// <= 32 right-hand sides (2 tiles): 3 wavefronts per SIMD
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void bfStageKernelC128Mfma2(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<2, 3>(p, tabs);
}
// <= 16 right-hand sides (1 tile): 5 wavefronts per SIMD
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void bfStageKernelC128Mfma1(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<1, 5>(p, tabs);
}
#endif
