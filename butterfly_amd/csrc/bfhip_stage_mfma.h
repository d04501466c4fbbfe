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
// One pass = MS 16-row slabs x NT 16-RHS tiles.  Fragment maps (cdna_hip_programming.md section 3): A[i = l & 15][k = l >> 4],
// B[k = l >> 4][j = l & 15], D reg v of lane l = D[i = 4 v + (l >> 4)][j = l & 15].  A complex multiply-accumulate is
//   * by default THREE real MFMAs (Gauss: T1 = Ar Xr, T2 = Ai Xi, T3 = (Ar + Ai)(Xr + Xi); Re = T1 - T2, Im = T3 - T1 - T2):
//     3 MS NT MFMAs per k-step (4 leaf columns), three accumulators per tile.  Normwise as accurate as zgemm's four
//     multiplications; componentwise the imaginary part carries an error of eps * (|Ar| + |Ai|)(|Xr| + |Xi|) summed over the
//     contraction, whatever its own size;
//   * with BFHIP_FLAG_EXACT_COMPLEX the FOUR real MFMAs of the textbook product (GAUSS = false below: Re += Ar Xr - Ai Xi,
//     Im += Ar Xi + Ai Xr, the -Ai Xi term through the NEG bit of the instruction -- blgp bit 0 negates A on the f64 MFMA:
//     tools/mfma_probe.hip), two accumulators per tile: componentwise what cblas_zgemm's own recurrence gives
//     (src/mat_dense_complex.c:1704-1765), a third more matrix-pipe work.
//
// The ragged last slab of an item (1 - 12 of its 16 rows real) runs as QUARTER slabs on v_mfma_f64_4x4x4 in the 64-RHS kernel
// (bfMfmaSegmentQ below): the X fragment is that instruction's B operand as it stands.
//
// What the inner loops are made of, and why (tools/mfma_probe.hip, tools/mfma_loop_probe.hip, profiles/r4_mfma_probe.json):
//   * ONE wavefront cannot keep the FP64 matrix pipe busy (0.75 of peak with 16 independent accumulators), two or more
//     can (0.99): every cycle a wavefront spends on anything else costs a quarter of the pipe, so the loop holds no
//     VALU address arithmetic, no selects and no register copies.  Fragments come through buffer loads -- a loop
//     invariant VGPR offset, the k-step advance in an SGPR, columns / rows past the end of the segment returned as
//     zeros by the range check (which includes the SGPR offset) instead of clamps + selects.
//   * The loads are asm statements and the waits are placed by hand: hipcc's wait insertion treats every load pending
//     at a loop header as one lump (s_waitcnt vmcnt(0) at the top: the fragment requested last would be waited for
//     first).  Loads return in order, so vmcnt(n) = "all but the n youngest have arrived".
//   * bfMfmaSegment (all three kernels): fragments land in registers, two sets for the leaf stream (requested one k-step ahead),
//     one for the X tiles (requested again right after the MFMAs that read them).
//   * bfMfmaSegmentDma (A/B builds only, BF_MF_DMA = 1): fragments land in a two-slot LDS ring per wavefront, two k-steps ahead of
//     their use -- built and measured in round 5, bit-identical, 2 - 4 % slower (below).
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
#ifndef BF_MF_QUARTER
#define BF_MF_QUARTER 1            /* the 64-RHS kernel runs the ragged last slab of an item on v_mfma_f64_4x4x4 (A/B builds: 0) */
#endif
// (Round 5, measured and removed: with BF_MF_WG_WAVES = 2 / 4 / 8 an s_barrier every pair of k-steps, so that list neighbours -- which
// walk the same X rows -- pace each other and their X requests reach the L2 together: 31.38 / 31.67 / 34.07 ms against 30.91 with
// one-wavefront workgroups; 4 wavefronts without the barrier 31.51.  DESIGN.md section 9.)

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
// cache policy of the leaf stream (read once per apply): "nt" = non-temporal; A/B builds try others (-DBF_MF_A_POLICY='"sc0 sc1 nt"')
#ifndef BF_MF_A_POLICY
#define BF_MF_A_POLICY "nt"
#endif
template <int STREAM, int OFF> __device__ __forceinline__ void bfFragLoad(BfFrag &f, uint32_t voff, bf_i4 rsrc, uint32_t soff) {
  if (STREAM) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 " BF_MF_A_POLICY : "=v"(f.u) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
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
// wavefronts per SIMD still fit).  The kernel is clock-bound, not issue-bound (the pipe stays 93 % busy and
// the clock follows the traffic beyond L2 in every variant of this loop), so a quarter fewer MFMAs is a quarter less time.  Normwise as accurate as
// the four-multiplication form (the imaginary part loses relative accuracy only where it is small next to |A||X|).
//
// Registers and requests.  A fragments (streamed from HBM, the long latency) have TWO sets: the A of k-step ks + 1 is
// requested at the top of k-step ks.  X fragments (L2) have ONE set: tile t is requested again right after the 3 MS
// MFMAs that read it, 3 MS (NT - 1) MFMAs before its next use (tile-outer, slab-inner order).  Requests return in order,
// so in front of every tile "all but the NT - 1 + MS youngest have arrived" is the wait (s_waitcnt vmcnt).
// An odd number of k-steps ends with a k-step of zeros (the leaf fragments past the end of the segment).
// the four real products of a complex multiply-accumulate (BFHIP_FLAG_EXACT_COMPLEX): Re += Ar Xr - Ai Xi, Im += Ar Xi + Ai Xr
__device__ __forceinline__ void bfMfmaExact(bf_d4 &re, bf_d4 &im, BfFrag const &a, BfFrag const &x) {
  re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.d[0], x.d[0], re, 0, 0, 0);
  re = __builtin_amdgcn_mfma_f64_16x16x4f64(a.d[1], x.d[1], re, 0, 0, 1);      // blgp bit 0: -A
  im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.d[0], x.d[1], im, 0, 0, 0);
  im = __builtin_amdgcn_mfma_f64_16x16x4f64(a.d[1], x.d[0], im, 0, 0, 0);
}
template <int MS, int SET>
__device__ __forceinline__ void bfMfmaRequestA(BfFrag (&a)[2][2], BfMfSeg const &sg, uint32_t soffA) {
  bfFragLoad<1, 0>(a[SET][0], sg.voffA, sg.ra, soffA);
  if (MS > 1) bfFragLoad<1, 256>(a[SET][1], sg.voffA, sg.ra, soffA);
}
template <int T>
__device__ __forceinline__ void bfMfmaRequestX(BfFrag (&x)[4], BfMfSeg const &sg, uint32_t voffX) {
  bfFragLoad<0, 256 * T>(x[T], voffX, sg.rx, 0);
}
template <int NT, int MS, int SET, int T, bool GAUSS>
__device__ __forceinline__ void bfMfmaTile(bf_d4 (&acc)[3][2][4], BfFrag (&a)[2][2], BfFrag (&x)[4], double (&as)[2], BfMfSeg const &sg, uint32_t voffXnext) {
  if (T < NT) {
    constexpr int TT = T < NT ? T : 0;
    constexpr int pending = NT - 1 + MS;
    if (T == 0 && MS > 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[SET][0].u), "+v"(a[SET][1].u), "+v"(x[0].u) : "n"(pending));
    else if (T == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[SET][0].u), "+v"(x[0].u) : "n"(pending));
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[TT].u) : "n"(pending));
    if (GAUSS) {
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
    } else {
#pragma unroll
      for (int m = 0; m < MS; ++m) bfMfmaExact(acc[0][m][TT], acc[1][m][TT], a[SET][m], x[TT]);
    }
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaRequestX<TT>(x, sg, voffXnext);
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NT, int MS, bool GAUSS = true>
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
    bfMfmaTile<NT, MS, 0, 0, GAUSS>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 1, GAUSS>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 2, GAUSS>(acc, a, x, as, sg, v1);
    bfMfmaTile<NT, MS, 0, 3, GAUSS>(acc, a, x, as, sg, v1);
    soffA += sg.stepA;
    bfMfmaRequestA<MS, 0>(a, sg, soffA);
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaTile<NT, MS, 1, 0, GAUSS>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 1, GAUSS>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 2, GAUSS>(acc, a, x, as, sg, v2);
    bfMfmaTile<NT, MS, 1, 3, GAUSS>(acc, a, x, as, sg, v2);
  }
  // the requests of the k-steps past the end (zeros from the range check / a padded table row) must land before the
  // registers are used again
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0][0].u), "+v"(a[0][MS > 1 ? 1 : 0].u), "+v"(x[0].u), "+v"(x[NT > 1 ? 1 : 0].u), "+v"(x[NT > 2 ? 2 : 0].u), "+v"(x[NT > 3 ? 3 : 0].u));
}

// ---- the ragged last slab of a pass on v_mfma_f64_4x4x4 (round 5) ----------------------------------------------------------------------
// Row groups are ranks of (target, source) pairs -- 17 - 24 rows are common -- and an item's last 16-row slab is mostly padding (N =
// 262144: 4.4 % of all MFMA time, N = 65536: 10 %).  v_mfma_f64_4x4x4 computes four independent 4 x 4 x 4 blocks in a quarter of the
// big instruction's time (tools/mfma_4x4x4_probe.hip: A[i][k] in lane 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane
// 16 i + 4 b + j for block b; 0.88 of the big instruction's flop rate).  Its B operand is the 16 x 16 x 4 X FRAGMENT AS IT STANDS -- block
// b = right-hand sides 4 b ... 4 b + 3 -- and the SAME four leaf rows go to every block: a quarter fragment is one more leaf load whose
// lanes take row (lane & 3) instead of row (lane & 15).  A tail of t = 1 ... 12 rows runs as ceil(t / 4) quarter slabs (QT), next to MS = 0
// or 1 full slabs; 13 ... 16 rows stay a full slab.  Accumulators: one double per lane, tile and product.  Same waits as bfMfmaSegment
// with MS + QT leaf requests per k-step.  (The sums of a 4 x 4 x 4 block and of a 16 x 16 x 4 tile may round differently: results are not
// bit-identical to the all-full-slab kernel, equally accurate.)
template <int MS, int QT, int SET>
__device__ __forceinline__ void bfMfmaRequestAQ(BfFrag (&a)[2], BfFrag (&aq)[2][3], BfMfSeg const &sg, uint32_t voffAq, uint32_t soffA) {
  if (MS > 0) bfFragLoad<1, 0>(a[SET], sg.voffA, sg.ra, soffA);
  bfFragLoad<1, 256 * MS>(aq[SET][0], voffAq, sg.ra, soffA);
  if (QT > 1) bfFragLoad<1, 256 * MS + 64>(aq[SET][1], voffAq, sg.ra, soffA);
  if (QT > 2) bfFragLoad<1, 256 * MS + 128>(aq[SET][2], voffAq, sg.ra, soffA);
}
template <int MS, int QT, int SET, int T>
__device__ __forceinline__ void bfMfmaTileQ(bf_d4 (&acc)[3][4], double (&accq)[3][3][4], BfFrag (&a)[2], BfFrag (&aq)[2][3], BfFrag (&x)[4], double &as, double (&asq)[3],
                                            BfMfSeg const &sg, uint32_t voffXnext) {
  constexpr int pending = 4 - 1 + MS + QT;
  if (T == 0) {      // this k-step's leaf fragments and tile 0 (never the same variable twice in one statement)
    if (MS > 0 && QT == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(a[SET].u), "+v"(aq[SET][0].u), "+v"(x[0].u) : "n"(pending));
    else if (MS > 0 && QT == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[SET].u), "+v"(aq[SET][0].u), "+v"(aq[SET][1].u), "+v"(x[0].u) : "n"(pending));
    else if (MS > 0) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a[SET].u), "+v"(aq[SET][0].u), "+v"(aq[SET][1].u), "+v"(aq[SET][2].u), "+v"(x[0].u) : "n"(pending));
    else if (QT == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(aq[SET][0].u), "+v"(x[0].u) : "n"(pending));
    else if (QT == 2) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(aq[SET][0].u), "+v"(aq[SET][1].u), "+v"(x[0].u) : "n"(pending));
    else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(aq[SET][0].u), "+v"(aq[SET][1].u), "+v"(aq[SET][2].u), "+v"(x[0].u) : "n"(pending));
    if (MS > 0) as = a[SET].d[0] + a[SET].d[1];
#pragma unroll
    for (int q = 0; q < QT; ++q) asq[q] = aq[SET][q].d[0] + aq[SET][q].d[1];
  } else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x[T].u) : "n"(pending));
  double const xs = x[T].d[0] + x[T].d[1];
  if (MS > 0) {
    acc[0][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET].d[0], x[T].d[0], acc[0][T], 0, 0, 0);
    acc[1][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET].d[1], x[T].d[1], acc[1][T], 0, 0, 0);
    acc[2][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(as, xs, acc[2][T], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < QT; ++q) {
    accq[0][q][T] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[SET][q].d[0], x[T].d[0], accq[0][q][T], 0, 0, 0);
    accq[1][q][T] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[SET][q].d[1], x[T].d[1], accq[1][q][T], 0, 0, 0);
    accq[2][q][T] = __builtin_amdgcn_mfma_f64_4x4x4f64(asq[q], xs, accq[2][q][T], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  bfMfmaRequestX<T>(x, sg, voffXnext);
  __builtin_amdgcn_sched_barrier(0);
}
// the k-loop of one segment: MS (0 or 1) full slabs + QT (1 ... 3) quarter slabs x 4 tiles, Gauss's three multiplications
template <int MS, int QT>
__device__ __forceinline__ void bfMfmaSegmentQ(bf_d4 (&acc)[3][4], double (&accq)[3][3][4], BfMfSeg const &sg, uint32_t voffAq, uint32_t const *tab, uint32_t lk) {
  BfFrag a[2], aq[2][3], x[4];
  double as = 0, asq[3];
  uint32_t ti = lk;
  uint32_t soffA = 0;
  {
    uint32_t const v0 = tab[ti] + sg.cX;
    bfMfmaRequestAQ<MS, QT, 0>(a, aq, sg, voffAq, soffA);
    bfMfmaRequestX<0>(x, sg, v0);
    bfMfmaRequestX<1>(x, sg, v0);
    bfMfmaRequestX<2>(x, sg, v0);
    bfMfmaRequestX<3>(x, sg, v0);
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
    bfMfmaRequestAQ<MS, QT, 1>(a, aq, sg, voffAq, soffA);
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaTileQ<MS, QT, 0, 0>(acc, accq, a, aq, x, as, asq, sg, v1);
    bfMfmaTileQ<MS, QT, 0, 1>(acc, accq, a, aq, x, as, asq, sg, v1);
    bfMfmaTileQ<MS, QT, 0, 2>(acc, accq, a, aq, x, as, asq, sg, v1);
    bfMfmaTileQ<MS, QT, 0, 3>(acc, accq, a, aq, x, as, asq, sg, v1);
    soffA += sg.stepA;
    bfMfmaRequestAQ<MS, QT, 0>(a, aq, sg, voffAq, soffA);
    __builtin_amdgcn_sched_barrier(0);
    bfMfmaTileQ<MS, QT, 1, 0>(acc, accq, a, aq, x, as, asq, sg, v2);
    bfMfmaTileQ<MS, QT, 1, 1>(acc, accq, a, aq, x, as, asq, sg, v2);
    bfMfmaTileQ<MS, QT, 1, 2>(acc, accq, a, aq, x, as, asq, sg, v2);
    bfMfmaTileQ<MS, QT, 1, 3>(acc, accq, a, aq, x, as, asq, sg, v2);
  }
  // the requests of the k-steps past the end (set 0 of the leaf fragments, every X tile) must land before the registers are used again
  if (MS > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0].u) :: "memory");
  if (QT == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(aq[0][0].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  else if (QT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(aq[0][0].u), "+v"(aq[0][1].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(aq[0][0].u), "+v"(aq[0][1].u), "+v"(aq[0][2].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
}
// the rows of the quarter slabs out of their accumulators: lane (i = lane >> 4, j = lane & 15) holds D[i][j] of every quarter and tile
template <int MS, int QT>
__device__ __forceinline__ void bfMfmaStoreQ(StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, double (&accq)[3][3][4], bool hasIdentity, int lane) {
  uint32_t const nrhs = p.nrhs, np = it.numPieces;
  double2 *out = (it.mrFlags & BF_ITEM_OUT_Y) ? (double2 *)p.y : (double2 *)p.temp;
  uint32_t const qmax = nrhs - 1 - q0;
  uint32_t lane2 = (uint32_t)lane;
  asm volatile("" : "+v"(lane2));
  uint32_t const li2 = lane2 & 15u, lk2 = lane2 >> 4;
#pragma unroll
  for (int q = 0; q < QT; ++q)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      uint32_t const row = s0 + 16u * MS + 4u * q + lk2;
      if (row < mr && 16u * t + li2 <= qmax) {
        double re = accq[0][q][t] - accq[1][q][t];
        double im = accq[2][q][t] - accq[0][q][t] - accq[1][q][t];
        if (hasIdentity) {
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
}

// ---- the same k-loop with its fragments prefetched through LDS (round 5: built, measured, NOT the product's loop: BF_MF_DMA) --------
// The question it answers (tools/mfma_loop_probe.hip: the loops on the bare machine, no items, no tables rebuilt, no tails): the
// register loop reaches 0.88 of the FP64 matrix peak with the leaf fragments streamed from HBM, 0.80 when the X rows miss L2 and
// 0.93 when the leaf fragments come from L2 -- is that the latency of requests made only one k-step (24 MFMAs = 1536 cycles of a
// SIMD that two wavefronts share, ~1.3 us) ahead?  Requests return in order (one counter), so a deeper prefetch needs a place to
// land that is not a register: every fragment of k-step ks + 2 is requested, during k-step
// ks, as an LDS-DMA (buffer_load ... lds: no VGPR destination) into a ring of two 6 KiB slots per wavefront; at the top of a k-step
// its slot is complete (s_waitcnt vmcnt(F): only the F requests of the next k-step may be pending), the fragments are read with six
// ds_read_b128 into ONE register set, the slot is handed to the requests of k-step ks + 2 right away, and the MFMAs run from
// registers.  Same MFMAs in the same order: bit-identical to bfMfmaSegment (the probe compares them on ragged segments).  2 x 6 KiB
// + the 7.1 KiB table per one-wavefront workgroup: eight of them (two per SIMD) fit a CU's 160 KiB.
// The answer is NO: with twice the prefetch distance the leaf stream from HBM and the X rows from beyond L2 cost exactly what they
// cost the register loop (counters: the same clocks, see BF_MF_DMA below), and the ds_read bubble costs 4 % on top.
//   * LDS-DMA writes M0 + lane * 16 (M0 is written in the statement that uses it: the compiler does not keep it); every instruction
//     uses offset:0, tile / slab offsets travel in the SGPR offset.
//   * Columns / rows past the end of the segment: out-of-range lanes of an LDS-DMA deliver zeros like a register load (probed:
//     tools/mfma_loop_probe.hip `dma_oob`), the padded table rows are rows of the segment.
#define BF_MF_DMA_SLOT 6144u            /* MS + NT <= 6 fragments of 1 KiB */
#define BF_MF_DMA_RING (2u * BF_MF_DMA_SLOT)
template <int STREAM> __device__ __forceinline__ void bfDmaLoad(uint32_t ldsDst, uint32_t voff, bf_i4 rsrc, uint32_t soff) {
  if (STREAM) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(ldsDst), "v"(voff), "s"(rsrc), "s"(soff));
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(ldsDst), "v"(voff), "s"(rsrc), "s"(soff));
}
// All R = MS + NT fragments of a slot into registers: the wait for the slot's LDS-DMAs, the ds_reads and the wait for THEIR data are
// ONE asm statement with early-clobber outputs -- hipcc treats an asm output as written when the statement ends, and with the
// wait in a statement of its own it is free to copy a fragment register before the data has landed (it did: v_mov of the A fragment
// between the ds_read and the s_waitcnt in the instantiations that needed a copy; cdna_hip_programming.md section 5.7 item 1).
template <int R, int BASE> __device__ __forceinline__ void bfLdsReadSlot(BfFrag (&f)[6], uint32_t laneLds) {
  static_assert(R >= 2 && R <= 6, "one to two slabs, one to four tiles");
  if constexpr (R == 6)
    asm volatile("s_waitcnt vmcnt(6)\n\tds_read_b128 %0, %6 offset:%7\n\tds_read_b128 %1, %6 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\tds_read_b128 %3, %6 offset:%10\n\t"
                 "ds_read_b128 %4, %6 offset:%11\n\tds_read_b128 %5, %6 offset:%12\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0].u), "=&v"(f[1].u), "=&v"(f[2].u), "=&v"(f[3].u), "=&v"(f[4].u), "=&v"(f[5].u)
                 : "v"(laneLds), "n"(BASE), "n"(BASE + 1024), "n"(BASE + 2048), "n"(BASE + 3072), "n"(BASE + 4096), "n"(BASE + 5120));
  else if constexpr (R == 5)
    asm volatile("s_waitcnt vmcnt(5)\n\tds_read_b128 %0, %5 offset:%6\n\tds_read_b128 %1, %5 offset:%7\n\tds_read_b128 %2, %5 offset:%8\n\tds_read_b128 %3, %5 offset:%9\n\t"
                 "ds_read_b128 %4, %5 offset:%10\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0].u), "=&v"(f[1].u), "=&v"(f[2].u), "=&v"(f[3].u), "=&v"(f[4].u)
                 : "v"(laneLds), "n"(BASE), "n"(BASE + 1024), "n"(BASE + 2048), "n"(BASE + 3072), "n"(BASE + 4096));
  else if constexpr (R == 4)
    asm volatile("s_waitcnt vmcnt(4)\n\tds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0].u), "=&v"(f[1].u), "=&v"(f[2].u), "=&v"(f[3].u)
                 : "v"(laneLds), "n"(BASE), "n"(BASE + 1024), "n"(BASE + 2048), "n"(BASE + 3072));
  else if constexpr (R == 3)
    asm volatile("s_waitcnt vmcnt(3)\n\tds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0].u), "=&v"(f[1].u), "=&v"(f[2].u)
                 : "v"(laneLds), "n"(BASE), "n"(BASE + 1024), "n"(BASE + 2048));
  else
    asm volatile("s_waitcnt vmcnt(2)\n\tds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f[0].u), "=&v"(f[1].u)
                 : "v"(laneLds), "n"(BASE), "n"(BASE + 1024));
}
// (a buffer instruction's scalar offset is an SGPR or an inline constant, never a literal: the tile offsets are made opaque to hipcc)
template <uint32_t V> __device__ __forceinline__ uint32_t bfSgprConst() { uint32_t v; asm("s_movk_i32 %0, %1" : "=s"(v) : "n"(V)); return v; }
template <int NT, int MS, int SLOT>
__device__ __forceinline__ void bfMfmaDmaRequest(BfMfSeg const &sg, uint32_t ring, uint32_t soffA, uint32_t voffX) {
  constexpr uint32_t base = SLOT * BF_MF_DMA_SLOT;
  bfDmaLoad<1>(ring + base, sg.voffA, sg.ra, soffA);
  if (MS > 1) bfDmaLoad<1>(ring + base + 1024u, sg.voffA, sg.ra, soffA + 256u);
  bfDmaLoad<0>(ring + base + MS * 1024u, voffX, sg.rx, bfSgprConst<0>());
  if (NT > 1) bfDmaLoad<0>(ring + base + (MS + 1) * 1024u, voffX, sg.rx, bfSgprConst<256>());
  if (NT > 2) bfDmaLoad<0>(ring + base + (MS + 2) * 1024u, voffX, sg.rx, bfSgprConst<512>());
  if (NT > 3) bfDmaLoad<0>(ring + base + (MS + 3) * 1024u, voffX, sg.rx, bfSgprConst<768>());
}
// one k-step out of slot SLOT; soffA / voffX: what is requested into the slot once it has been read (k-step + 2)
template <int NT, int MS, int SLOT, bool GAUSS>
__device__ __forceinline__ void bfMfmaDmaStep(bf_d4 (&acc)[3][2][4], BfMfSeg const &sg, uint32_t ring, uint32_t laneLds, uint32_t soffA, uint32_t voffX) {
  constexpr uint32_t base = SLOT * BF_MF_DMA_SLOT;
  // this slot's MS + NT requests are the oldest (complete once only the other slot's may be pending): wait, read, wait -- one statement
  BfFrag f[6];
  bfLdsReadSlot<MS + NT, base>(f, laneLds);
  BfFrag (&a)[6] = f;
  BfFrag *const x = f + MS;
  __builtin_amdgcn_sched_barrier(0);
  bfMfmaDmaRequest<NT, MS, SLOT>(sg, ring, soffA, voffX);        // the slot is free: everything of it is in registers
  __builtin_amdgcn_sched_barrier(0);
  if (GAUSS) {
    double as[2];
    as[0] = a[0].d[0] + a[0].d[1];
    if (MS > 1) as[1] = a[1].d[0] + a[1].d[1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double const xs = x[t].d[0] + x[t].d[1];
#pragma unroll
      for (int m = 0; m < MS; ++m) {
        acc[0][m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[0], x[t].d[0], acc[0][m][t], 0, 0, 0);
        acc[1][m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m].d[1], x[t].d[1], acc[1][m][t], 0, 0, 0);
        acc[2][m][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], xs, acc[2][m][t], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < MS; ++m) bfMfmaExact(acc[0][m][t], acc[1][m][t], a[m], x[t]);
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int NT, int MS, bool GAUSS = true>
__device__ __forceinline__ void bfMfmaSegmentDma(bf_d4 (&acc)[3][2][4], BfMfSeg const &sg, uint32_t const *tab, uint32_t lk, uint32_t ring, uint32_t laneLds) {
  uint32_t ti = lk;
  uint32_t soffA = 0;
  bfMfmaDmaRequest<NT, MS, 0>(sg, ring, soffA, tab[ti] + sg.cX);
  soffA += sg.stepA;
  bfMfmaDmaRequest<NT, MS, 1>(sg, ring, soffA, tab[ti + 4] + sg.cX);
  soffA += sg.stepA;
  uint32_t t2 = tab[ti + 8], t3 = tab[ti + 12];      // the rows of k-steps 2 and 3, read an iteration ahead of their use
  ti += 16;
  for (uint32_t ks = 0; ks < sg.ksteps; ks += 2) {
    uint32_t const v2 = t2 + sg.cX, v3 = t3 + sg.cX;
    t2 = tab[ti];                                     // the table is padded past the last k-step (BF_MF_TABPAD)
    t3 = tab[ti + 4];
    ti += 8;
    bfMfmaDmaStep<NT, MS, 0, GAUSS>(acc, sg, ring, laneLds, soffA, v2);
    soffA += sg.stepA;
    bfMfmaDmaStep<NT, MS, 1, GAUSS>(acc, sg, ring, laneLds, soffA, v3);
    soffA += sg.stepA;
  }
  // the requests of the k-steps past the end must land before the ring (and the table) are used again
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// The pass's rows x right-hand sides out of the accumulators (Gauss: Re = T1 - T2, Im = T3 - T1 - T2) into y / the intermediate.
template <int NT, int MS, bool GAUSS>
__device__ __forceinline__ void bfMfmaStore(StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, bf_d4 (&acc)[3][2][4], bool hasIdentity, int lane) {
  uint32_t const nrhs = p.nrhs, np = it.numPieces;
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
          double re = GAUSS ? acc[0][m][t][v] - acc[1][m][t][v] : acc[0][m][t][v];
          double im = GAUSS ? acc[2][m][t][v] - acc[0][m][t][v] - acc[1][m][t][v] : acc[1][m][t][v];
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
  // (stores and fragment requests share vmcnt: the NEXT pass of the wavefront, if there is one, waits for these stores before it
  //  requests anything -- bfMfmaPass; the last pass ends with its stores in flight and frees its slot that much earlier)
}

// One pass = rows [s0, s0 + 16 MS) x RHS [q0, q0 + 16 NT) of one item, over all its segments.
// (DMA: the k-loop with its fragments prefetched through the wavefront's LDS ring -- the 4-tile kernel; `ring` = its LDS byte address)
// (QT > 0: MS = 0 or 1 full slabs and QT quarter slabs on v_mfma_f64_4x4x4, four tiles, Gauss -- bfMfmaSegmentQ)
template <int NT, int MS, bool DMA, bool GAUSS, int QT = 0>
__device__ __forceinline__ void bfMfmaPass(StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, uint32_t *tab, int lane, uint32_t ring) {
  static_assert(QT == 0 || (NT == 4 && MS <= 1 && GAUSS && !DMA), "quarter slabs: the 4-tile Gauss register loop");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores of the wavefront's previous pass: nothing may be pending when the k-loop counts
  uint32_t const nrhs = p.nrhs;
  uint32_t const li = lane & 15, lk = lane >> 4;
  bf_d4 acc[3][2][4];            // T1 = sum Ar Xr, T2 = sum Ai Xi, T3 = sum (Ar + Ai)(Xr + Xi)
  bf_d4 accs[3][4];              // QT > 0: the one full slab
  double accq[3][3][4];          // QT > 0: the quarter slabs
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[g][m][t] = (bf_d4){0, 0, 0, 0};
  if (QT > 0) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        accs[g][t] = (bf_d4){0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 3; ++q) accq[g][q][t] = 0.0;
      }
  }
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
    if constexpr (QT > 0) bfMfmaSegmentQ<MS, QT>(accs, accq, sg, (lk * mr + s0 + (li & 3u)) * 16u, tab, lk);      // a quarter fragment: row (lane & 3) of its four
    else if (DMA) bfMfmaSegmentDma<NT, MS, GAUSS>(acc, sg, tab, lk, ring, ring + (uint32_t)lane * 16u);
    else bfMfmaSegment<NT, MS, GAUSS>(acc, sg, tab, lk);
    waveSync();                                      // the table is rewritten by the next segment
  }
  if constexpr (QT > 0) {
    bfMfmaStoreQ<MS, QT>(p, it, mr, s0, q0, accq, hasIdentity, lane);
    if (MS > 0) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[g][0][t] = accs[g][t];
    }
  }
  bfMfmaStore<NT, MS, GAUSS>(p, it, mr, s0, q0, acc, hasIdentity, lane);
}

template <int MS, int MAXNT, bool DMA, bool GAUSS>
__device__ __forceinline__ void bfMfmaDispatch(uint32_t nt, StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t s0, uint32_t q0, uint32_t *tab, int lane, uint32_t ring) {
  if (MAXNT >= 4 && nt == 4) bfMfmaPass<4, MS, DMA, GAUSS>(p, it, mr, s0, q0, tab, lane, ring);
  else if (MAXNT >= 3 && nt == 3) bfMfmaPass<3, MS, DMA, GAUSS>(p, it, mr, s0, q0, tab, lane, ring);
  else if (MAXNT >= 2 && nt == 2) bfMfmaPass<2, MS, DMA, GAUSS>(p, it, mr, s0, q0, tab, lane, ring);
  else bfMfmaPass<1, MS, DMA, GAUSS>(p, it, mr, s0, q0, tab, lane, ring);
}

// ---- bundles (round 5: built, measured, NOT the product's kernel: BF_MF_BUNDLES) -------------------------------------------------------
// A workgroup of four wavefronts whose items read the SAME X rows.  The row chunks of a row group and the sibling groups of a radix-4
// stage (reference src/fac_helm2.c:277-318) multiply different leaf rows by the same rows of the input; the planner keeps them
// together in the list and bfPlanBundles marks runs of four with equal inputs, <= 32 rows and the same number of slabs (SHARED
// bundles: ONE pass each, the same number of k-steps); everything else goes four unrelated items to a workgroup (MIXED bundles, the
// one-wavefront passes above -- a workgroup must keep its four wavefronts alive: a new workgroup needs a free slot on EVERY SIMD
// (tools/wave_placement_probe.hip: its four wavefronts always land on four different SIMDs), so survivors of workgroups whose other
// wavefronts had exited block the CU as soon as one SIMD holds two of them -- such workgroups ran at a fifth of the rate).  In a shared bundle every wavefront streams its own leaf fragments into registers exactly as bfMfmaSegment does; the
// four X tiles of a k-step are fetched ONCE per workgroup -- wavefront w brings tile w as an LDS-DMA, two k-steps ahead, into a ring of
// three 4 KiB slots -- and every wavefront reads them with ds_read_b128 right after the MFMAs that used the previous k-step's copy.
// One workgroup barrier per k-step: behind it the next k-step's slot is complete and nobody still reads the slot before the current
// one, which is the one requested into next.  Same MFMAs in the same order per accumulator: bit-identical to the register loop
// (tools/mfma_sharedx_probe.hip compares them; the GPU suite passes with either kernel).
// What it does (N = 262144, 64 RHS, counters over the shared bundles alone, 83 - 93 % of a stage's work): HBM traffic 1.08x the
// algorithmic bytes instead of 1.4x, clock +6 % -- and the matrix pipe busy 0.80 of the cycles instead of 0.87: four wavefronts
// that wait for each other at every k-step and at every segment's first requests (and, in the first version, for the ONE table that
// wavefront 0 wrote: every wavefront writes its own copy now, 1 % of the apply).
// On the bare machine (the probe: no items, no tables) the loop is 9 - 17 % faster than the register loop; in the kernel the shared
// bundles are 2 - 5 % faster per unit of work, the mixed ones 2 % slower than one-wavefront workgroups, the whole apply 30.65 - 30.88 ms
// against 30.13 - 30.22 for the one-wavefront kernel, same box.  Not the barrier itself (without it, wrong results: the same time) and
// not the order of the bundles (shuffled inside their cost buckets: the same time).  DESIGN.md section 9.
#define BF_MF_SX_SLOT 4096u
#define BF_MF_SX_RING (3u * BF_MF_SX_SLOT)
#ifndef BF_MF_BUNDLES
#define BF_MF_BUNDLES 0           /* 1: the 64-RHS kernels run bundles (A/B builds: make variant V=bundles DEFS=-DBF_MF_BUNDLES=1) */
#endif
#ifndef BF_MF_XCD_RUN_B
#define BF_MF_XCD_RUN_B 8u           /* bundles (workgroups of four wavefronts) that are list neighbours and go to one XCD */
#endif

template <int MS, int SET, bool GAUSS>
__device__ __forceinline__ void bfSxStep(bf_d4 (&acc)[3][2][4], BfFrag (&a)[2][2], BfFrag (&x)[4], BfMfSeg const &sg, uint32_t &soffA, uint32_t voffXnext2,
                                         uint32_t slotNext, uint32_t slotFree, uint32_t laneLds, uint32_t mine) {
  // everything this wavefront asked for has arrived: the leaf fragments of this k-step, its tiles of the NEXT k-step's X (LDS-DMA),
  // the X fragments of this k-step (ds_read) ...
  if (MS > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a[SET][0].u), "+v"(a[SET][1].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a[SET][0].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  // ... and everybody else's
  asm volatile("s_barrier" ::: "memory");
  soffA += sg.stepA;
  bfMfmaRequestA<MS, SET ^ 1>(a, sg, soffA);
  if (mine & 1u) bfDmaLoad<0>(slotFree, voffXnext2, sg.rx, bfSgprConst<0>());
  if (mine & 2u) bfDmaLoad<0>(slotFree + 1024u, voffXnext2, sg.rx, bfSgprConst<256>());
  if (mine & 4u) bfDmaLoad<0>(slotFree + 2048u, voffXnext2, sg.rx, bfSgprConst<512>());
  if (mine & 8u) bfDmaLoad<0>(slotFree + 3072u, voffXnext2, sg.rx, bfSgprConst<768>());
  uint32_t const vaddr = laneLds + slotNext;
  double as[2];
  if (GAUSS) {
    as[0] = a[SET][0].d[0] + a[SET][0].d[1];
    if (MS > 1) as[1] = a[SET][1].d[0] + a[SET][1].d[1];
  }
  __builtin_amdgcn_sched_barrier(0);
#define BF_SX_TILE(T) do { \
    if (GAUSS) { \
      double const xs = x[T].d[0] + x[T].d[1]; \
      _Pragma("unroll") for (int m = 0; m < MS; ++m) { \
        acc[0][m][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][m].d[0], x[T].d[0], acc[0][m][T], 0, 0, 0); \
        acc[1][m][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[SET][m].d[1], x[T].d[1], acc[1][m][T], 0, 0, 0); \
        acc[2][m][T] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[m], xs, acc[2][m][T], 0, 0, 0); \
      } \
    } else { \
      _Pragma("unroll") for (int m = 0; m < MS; ++m) bfMfmaExact(acc[0][m][T], acc[1][m][T], a[SET][m], x[T]); \
    } \
    __builtin_amdgcn_sched_barrier(0); \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x[T].u) : "v"(vaddr), "n"(1024 * T)); \
    __builtin_amdgcn_sched_barrier(0); \
  } while (0)
  BF_SX_TILE(0); BF_SX_TILE(1); BF_SX_TILE(2); BF_SX_TILE(3);
#undef BF_SX_TILE
}

// the k-loop of one segment for one wavefront of a bundle (4 tiles, MS slabs); `mine`: bit t = this wavefront fetches tile t
template <int MS, bool GAUSS>
__device__ __forceinline__ void bfSxSegment(bf_d4 (&acc)[3][2][4], BfMfSeg const &sg, uint32_t const *tab, uint32_t lk, uint32_t ring, uint32_t lane, uint32_t mine) {
  BfFrag a[2][2], x[4];
  uint32_t const laneLds = lane * 16u;
  uint32_t ti = lk, soffA = 0;
  uint32_t s0 = ring, s1 = ring + BF_MF_SX_SLOT, s2 = ring + 2u * BF_MF_SX_SLOT;      // slots of k-steps ks, ks + 1, ks + 2 (wave-uniform)
  {
    uint32_t const v0 = tab[ti] + sg.cX, v1 = tab[ti + 4] + sg.cX;
    if (mine & 1u) { bfDmaLoad<0>(s0, v0, sg.rx, bfSgprConst<0>()); bfDmaLoad<0>(s1, v1, sg.rx, bfSgprConst<0>()); }
    if (mine & 2u) { bfDmaLoad<0>(s0 + 1024u, v0, sg.rx, bfSgprConst<256>()); bfDmaLoad<0>(s1 + 1024u, v1, sg.rx, bfSgprConst<256>()); }
    if (mine & 4u) { bfDmaLoad<0>(s0 + 2048u, v0, sg.rx, bfSgprConst<512>()); bfDmaLoad<0>(s1 + 2048u, v1, sg.rx, bfSgprConst<512>()); }
    if (mine & 8u) { bfDmaLoad<0>(s0 + 3072u, v0, sg.rx, bfSgprConst<768>()); bfDmaLoad<0>(s1 + 3072u, v1, sg.rx, bfSgprConst<768>()); }
  }
  bfMfmaRequestA<MS, 0>(a, sg, soffA);
  uint32_t t2 = tab[ti + 8], t3 = tab[ti + 12];      // the rows of k-steps 2 and 3 (the table is padded past the last k-step: BF_MF_TABPAD)
  ti += 16;
  // (never the same variable twice in one statement: hipcc then copies it BEFORE the wait and may keep the copy)
  if (MS > 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0][0].u), "+v"(a[0][1].u));
  else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0][0].u));
  asm volatile("s_barrier" ::: "memory");
  {
    uint32_t const vaddr = laneLds + s0;
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(x[0].u), "=&v"(x[1].u), "=&v"(x[2].u), "=&v"(x[3].u) : "v"(vaddr));
  }
  for (uint32_t ks = 0; ks < sg.ksteps; ks += 2) {
    uint32_t const v2 = t2 + sg.cX, v3 = t3 + sg.cX;
    t2 = tab[ti];
    t3 = tab[ti + 4];
    ti += 8;
    bfSxStep<MS, 0, GAUSS>(acc, a, x, sg, soffA, v2, s1, s2, laneLds, mine);
    bfSxStep<MS, 1, GAUSS>(acc, a, x, sg, soffA, v3, s2, s0, laneLds, mine);
    uint32_t const o0 = s0, o1 = s1;
    s0 = s2; s1 = o0; s2 = o1;                       // two k-steps on
  }
  // the requests of the k-steps past the end (zeros from the range check / padded table rows) must land before the registers, the
  // ring and the table are used again, by anybody
  if (MS > 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a[0][0].u), "+v"(a[0][1].u), "+v"(a[1][0].u), "+v"(a[1][1].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a[0][0].u), "+v"(a[1][0].u), "+v"(x[0].u), "+v"(x[1].u), "+v"(x[2].u), "+v"(x[3].u));
  asm volatile("s_barrier" ::: "memory");
}

// One pass of one wavefront of a bundle: rows [0, mr <= 16 MS) x RHS [q0, q0 + 64) of ITS item.  Every wavefront writes its own copy
// of the segment table (the same rows for all of them): with ONE table, written by wavefront 0, the other three waited for it at
// every segment.
template <int MS, bool GAUSS>
__device__ __forceinline__ void bfMfmaPassBundle(StageParams const &p, BfDevItem const &it, uint32_t mr, uint32_t q0, uint32_t *tab, int lane, uint32_t ring, uint32_t wave, uint32_t mine) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores of the previous pass
  uint32_t const nrhs = p.nrhs;
  uint32_t const li = lane & 15, lk = lane >> 4;
  bf_d4 acc[3][2][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < MS; ++m)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[g][m][t] = (bf_d4){0, 0, 0, 0};
  bool hasIdentity = false;
  uint32_t const np = it.numPieces;
  uint32_t const spanRows = BF_MF_SPAN_BYTES / (nrhs * 16u);
  uint32_t pi = 0;
  while (pi < np) {
    // every wavefront walks ITS item's descriptors (same input rows and widths for all of them, its own leaf offsets) and finds the
    // same segment
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
    if (!cols) break;                      // identity pieces only (the same for every wavefront of the bundle)
    if ((uint32_t)lane < BF_MF_TABPAD) tab[cols + lane] = minRow;
    waveSync();
    for (uint32_t j = (uint32_t)lane; j < cols + BF_MF_TABPAD; j += 64u) tab[j] = (tab[j] - minRow) * (nrhs * 16u);
    waveSync();
    BfMfSeg sg;
    sg.stepA = 4u * mr * 16u;
    sg.ksteps = (cols + 3u) / 4u;
    sg.ra = bfMakeRsrc((double2 const *)p.arena + aOff, mr * cols * 16u);
    char const *xin = inX ? (char const *)p.x : (char const *)p.temp;
    sg.rx = bfMakeRsrc(xin + ((uint64_t)minRow * nrhs + q0) * 16u, ((maxRow - minRow) * nrhs + (nrhs - q0)) * 16u);
    sg.voffA = (lk * mr + li) * 16u;
    sg.cX = li * 16u;
    bfSxSegment<MS, GAUSS>(acc, sg, tab, lk, ring, (uint32_t)lane, mine);      // ends with a barrier: table and ring are free again
  }
  bfMfmaStore<4, MS, GAUSS>(p, it, mr, 0u, q0, acc, hasIdentity, lane);
}

// The 64-RHS kernel's body: workgroup = bundle of four items.  SHARED bundles run the loop above; MIXED ones (items that found nobody
// to share with, taller ones, zero fills) the one-wavefront passes, four unrelated items side by side -- a workgroup always keeps
// four live wavefronts (measured: workgroups that kept one ran at a fifth of the rate; see the note on bundles above).
template <bool GAUSS>
__device__ __forceinline__ void bfStageBodyC128MfmaBundles(StageParams const &p, uint32_t (*tabs)[BF_MF_TABCAP + BF_MF_TABPAD], char *ringMem) {
  int const lane = threadIdx.x & 63;
  uint32_t const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t wg = blockIdx.x;
  uint32_t const numWg = p.numBundles;
  if (BF_MF_XCD_RUN_B > 1) {      // runs of list neighbours on ONE XCD (workgroups are dealt to the 8 XCDs round robin)
    uint32_t const blk = 8u * BF_MF_XCD_RUN_B;
    if (wg < numWg / blk * blk) { uint32_t const r = wg % blk; wg = wg - r + (r % 8u) * BF_MF_XCD_RUN_B + r / 8u; }
  }
  if (wg >= numWg) return;
  uint32_t const entry = __builtin_amdgcn_readfirstlane(p.bundles[wg]), first = entry & 0x7fffffffu;
  uint32_t const cnt = (__builtin_amdgcn_readfirstlane(p.bundles[wg + 1]) & 0x7fffffffu) - first;
  bool const mixed = (entry >> 31) != 0;   // four unrelated items: one-wavefront passes, every wavefront its own table
  if (wave >= cnt) return;                 // (the last bundle of a list; a barrier waits for the surviving wavefronts of a workgroup only)
  BfDevItem const it = p.items[first + wave];
  uint32_t const mr = it.mrFlags & 0xffffu;
  uint32_t const nrhs = p.nrhs;
  uint32_t const ring = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ringMem);
  uint32_t mine = 0;
  for (uint32_t t = 0; t < 4u; ++t) if (t % cnt == wave) mine |= 1u << t;
  mine = __builtin_amdgcn_readfirstlane(mine);
  uint32_t *tab = tabs[wave];
  for (uint32_t q0 = 0; q0 < nrhs; q0 += 64) {
    if (!mixed) {                          // (bfPlanBundles: <= 32 rows each, the same number of slabs)
      // (the last, partial 64-RHS block of a wider right-hand side runs all four tiles: the columns past nrhs are whatever follows
      //  them in memory -- zeros past the end of the last row -- and are never stored)
      if (mr > 16) bfMfmaPassBundle<2, GAUSS>(p, it, mr, q0, tab, lane, ring, wave, mine);
      else bfMfmaPassBundle<1, GAUSS>(p, it, mr, q0, tab, lane, ring, wave, mine);
    } else {
      uint32_t const nt = (nrhs - q0 >= 64) ? 4u : (nrhs - q0 + 15u) / 16u;
      uint32_t s0 = 0;
      while (s0 < mr) {
        if (mr - s0 > 16) { bfMfmaDispatch<2, 4, false, GAUSS>(nt, p, it, mr, s0, q0, tab, lane, 0u); s0 += 32; }
        else { bfMfmaDispatch<1, 4, false, GAUSS>(nt, p, it, mr, s0, q0, tab, lane, 0u); s0 += 16; }
      }
    }
  }
}

// MAXNT = the widest pass the launch needs (RHS tiles of 16): the accumulators of 4 tiles x 2 slabs x 3 products leave two
// wavefronts per SIMD, which is what the matrix pipe needs at 64 RHS -- but with 2 - 32 RHS the kernel is bound by the leaf
// stream, not by the pipe, and then it is wavefronts (bytes in flight) that count: the 1- and 2-tile instantiations need
// a third / half of the registers and run WAVES = 5 / 3 wavefronts per SIMD (N = 262144: 2 - 16 RHS 14.3 - 15.9 -> see
// DESIGN_EXPERIMENTS.md section 4).
template <int MAXNT, int WAVES, bool DMA, bool GAUSS = true>
__device__ __forceinline__ void bfStageBodyC128Mfma(StageParams const &p, uint32_t (*tabs)[BF_MF_TABCAP + BF_MF_TABPAD], char *rings) {
  int const lane = threadIdx.x & 63;
  uint32_t const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *tab = tabs[wave];
  // (the low 32 bits of a flat LDS address are the offset inside the workgroup's allocation: what M0 and ds_read take)
  uint32_t const ring = DMA ? __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(rings + wave * BF_MF_DMA_RING)) : 0u;
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
      uint32_t const left = mr - s0;
      if (BF_MF_QUARTER && MAXNT == 4 && GAUSS && !DMA && nt == 4 && left < 32 && (left & 15u) && (left & 15u) <= 12u) {
        // the last pass of the item ends in 1 ... 12 rows: quarter slabs (v_mfma_f64_4x4x4) instead of a mostly padded full one
        if constexpr (MAXNT == 4 && GAUSS && !DMA) {
          uint32_t const qt = ((left & 15u) + 3u) / 4u;
          if (left > 16) {
            if (qt == 1) bfMfmaPass<4, 1, false, true, 1>(p, it, mr, s0, q0, tab, lane, ring);
            else if (qt == 2) bfMfmaPass<4, 1, false, true, 2>(p, it, mr, s0, q0, tab, lane, ring);
            else bfMfmaPass<4, 1, false, true, 3>(p, it, mr, s0, q0, tab, lane, ring);
          } else {
            if (qt == 1) bfMfmaPass<4, 0, false, true, 1>(p, it, mr, s0, q0, tab, lane, ring);
            else if (qt == 2) bfMfmaPass<4, 0, false, true, 2>(p, it, mr, s0, q0, tab, lane, ring);
            else bfMfmaPass<4, 0, false, true, 3>(p, it, mr, s0, q0, tab, lane, ring);
          }
        }
        s0 = mr;
      }
      else if (left > 16) { bfMfmaDispatch<2, MAXNT, DMA, GAUSS>(nt, p, it, mr, s0, q0, tab, lane, ring); s0 += 32; }
      else { bfMfmaDispatch<1, MAXNT, DMA, GAUSS>(nt, p, it, mr, s0, q0, tab, lane, ring); s0 += 16; }
    }
  }
}

// Which k-loop the 4-tile kernel runs.  Measured (tools/mfma_loop_probe.hip with --pmc, profiles/r5_mfma_loop_probe*.json): the register
// loop keeps the matrix pipe busy 0.93 - 0.94 of the cycles WHEREVER its operands come from -- it is not bound by their latency --
// and what the traffic beyond L2 costs is CLOCK: 2.37 GHz with the leaf stream and the X rows in L2, 2.23 with the leaf stream from HBM,
// 2.06 / 2.04 with the X rows from the Infinity Cache / HBM as well.  The LDS-ring loop holds the same clocks and loses 4 % of the
// pipe to the ds_read bubble at the top of each k-step (busy 0.89): prefetching deeper buys nothing here.  0 = register loop (product),
// 1 = LDS ring (A/B builds: make variant V=dma DEFS=-DBF_MF_DMA=1).
#ifndef BF_MF_DMA
#define BF_MF_DMA 0
#endif
#if BF_MF_BUNDLES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BF_MFMA_WAVES_PER_SIMD, BF_MFMA_WAVES_PER_SIMD))) void bfStageKernelC128Mfma(StageParams p) {
  __shared__ uint32_t tabs[4][BF_MF_TABCAP + BF_MF_TABPAD];
  __shared__ __attribute__((aligned(16))) char ring[BF_MF_SX_RING];
  bfStageBodyC128MfmaBundles<true>(p, tabs, ring);
}
#else
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(BF_MFMA_WAVES_PER_SIMD, BF_MFMA_WAVES_PER_SIMD))) void bfStageKernelC128Mfma(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
#if BF_MF_DMA
  __shared__ __attribute__((aligned(16))) char rings[BF_MF_WG_WAVES * BF_MF_DMA_RING];
  bfStageBodyC128Mfma<4, BF_MFMA_WAVES_PER_SIMD, true>(p, tabs, rings);
#else
  bfStageBodyC128Mfma<4, BF_MFMA_WAVES_PER_SIMD, false>(p, tabs, nullptr);
#endif
}
#endif
// <= 32 right-hand sides (2 tiles): 3 wavefronts per SIMD
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void bfStageKernelC128Mfma2(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<2, 3, false>(p, tabs, nullptr);
}
// <= 16 right-hand sides (1 tile): 5 wavefronts per SIMD
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void bfStageKernelC128Mfma1(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<1, 5, false>(p, tabs, nullptr);
}
// BFHIP_FLAG_EXACT_COMPLEX: the same three kernels with the four real products of every complex one (componentwise zgemm's recurrence)
#if BF_MF_BUNDLES
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BF_MFMA_WAVES_PER_SIMD, BF_MFMA_WAVES_PER_SIMD))) void bfStageKernelC128MfmaExact(StageParams p) {
  __shared__ uint32_t tabs[4][BF_MF_TABCAP + BF_MF_TABPAD];
  __shared__ __attribute__((aligned(16))) char ring[BF_MF_SX_RING];
  bfStageBodyC128MfmaBundles<false>(p, tabs, ring);
}
#else
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(BF_MFMA_WAVES_PER_SIMD, BF_MFMA_WAVES_PER_SIMD))) void bfStageKernelC128MfmaExact(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
#if BF_MF_DMA
  __shared__ __attribute__((aligned(16))) char rings[BF_MF_WG_WAVES * BF_MF_DMA_RING];
  bfStageBodyC128Mfma<4, BF_MFMA_WAVES_PER_SIMD, true, false>(p, tabs, rings);
#else
  bfStageBodyC128Mfma<4, BF_MFMA_WAVES_PER_SIMD, false, false>(p, tabs, nullptr);
#endif
}
#endif
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void bfStageKernelC128Mfma2Exact(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<2, 3, false, false>(p, tabs, nullptr);
}
__global__ __launch_bounds__(64 * BF_MF_WG_WAVES) __attribute__((amdgpu_waves_per_eu(5, 5))) void bfStageKernelC128Mfma1Exact(StageParams p) {
  __shared__ uint32_t tabs[BF_MF_WG_WAVES][BF_MF_TABCAP + BF_MF_TABPAD];
  bfStageBodyC128Mfma<1, 5, false, false>(p, tabs, nullptr);
}
#endif
