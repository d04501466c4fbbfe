// bfhip_build.hip -- gfx950 device layer of the fac_helm2 value builder
// (include/bfhip_build.h).
//
// What runs here replaces, for every dense leaf of a Helmholtz butterfly,
//   * bfHelm2GetKernelMatrix (single layer)        reference src/helm2.c:93-125
//       -> bfEvalKernel: one Hankel evaluation per matrix element, all kernel
//          matrices of a batch in one launch (flat tile list, binary search);
//   * bfHelm2GetReexpansionMatrix                   src/helm2.c:321-365
//     = bfMatDenseComplexDenseComplexLstSq          src/mat_dense_complex.c:1767-1849
//       (LAPACK zgesvd + truncation + two zgemm)
//       -> bfJacobiKernel: one-sided (Hestenes) Jacobi SVD, one workgroup per
//          problem, column pairs of a round-robin step spread over lane groups;
//          bfQrcpKernel ahead of it for the problems that do not stay in LDS
//          (Householder QR with column pivoting; Jacobi then works on R^H);
//          bfGemmKernel: T = diag(1/sigma^2) (U Sigma)^H Z_orig, X = V T;
//   * the leaf -> packed-arena copy that bfhip_api.c does on the host for
//     host-valued operands -> bfPackKernel.
// Plus bfHelm2DenseKernel, the matrix-free N x N kernel matvec used as the
// acceptance check (examples/simple/bf_all_blocks.c:132-153).
//
// None of these kernels is on the apply path; they are compute-bound FP64 VALU
// work (Bessel functions, rotations) and are written for clarity first: fixed
// summation orders, no atomics in any result.

#include <hip/hip_runtime.h>
#include <time.h>
#include <stdint.h>
#include <stdlib.h>

#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"
#include "../../include/bfhip_build.h"

static int hipFailB(hipError_t e, char const *what) {
  if (e == hipSuccess) return 0;
  int code = (e == hipErrorOutOfMemory) ? BFABI_ERROR_MEMORY_ERROR : BFABI_ERROR_RUNTIME_ERROR;
  return bfhipFail(code, "%s: %s", what, hipGetErrorString(e));
}

template <typename T> static int uploadArrayB(T **d, T const *h, uint64_t count, char const *what) {
  *d = NULL;
  int rc = hipFailB(hipMalloc((void **)d, count * sizeof(T)), what);
  if (rc) return rc;
  return hipFailB(hipMemcpy(*d, h, count * sizeof(T), hipMemcpyHostToDevice), what);
}

// ---------------------------------------------------------------------------
// points and the kernel
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bfPoint(BfBuildPts const &ps, uint32_t i, double const *pts, double const *tpts, double &x, double &y) {
  if (ps.kind != BFHIP_PTS_CIRCLE) {
    double const *base = ps.kind == BFHIP_PTS_TREE ? pts : tpts;
    x = base[2 * (ps.first + i)];
    y = base[2 * (ps.first + i) + 1];
  } else {
    // bfCircle2SamplePoints, src/circle.c:12-35
    double const theta = (6.283185307179586 / (double)ps.count) * (double)i;
    x = ps.r * cos(theta) + ps.cx;
    y = ps.r * sin(theta) + ps.cy;
  }
}

// S : (i/4) H0^(1)(k r)                           = (-Y0 + i J0)(kr) / 4            src/helm2.c:111-117
// S': (i/4) k H1^(1)(k r)/r  n_tgt.(x_tgt - x_src)  = (-Y1 + i J1)(kr) k dot / (4 r)  src/helm2.c:155-163
// D : the same with the SOURCE normal                                                src/helm2.c:200-208
// combined field: alpha S + beta D                                                   src/helm2.c:253-266
// all 0 at r == 0.  (dx, dy) = x_tgt - x_src; (nx, ny) = the normal the potential uses.
__device__ __forceinline__ double2 bfHelm2G(double k, double dx, double dy) {
  double const r = hypot(dx, dy);
  if (r == 0.0) return make_double2(0.0, 0.0);
  double const kr = k * r;
  return make_double2(-0.25 * y0(kr), 0.25 * j0(kr));
}
__device__ __forceinline__ double2 bfHelm2Sp(double k, double dx, double dy, double nx, double ny) {
  double const r = hypot(dx, dy);
  if (r == 0.0) return make_double2(0.0, 0.0);
  double const kr = k * r;
  double const sc = 0.25 * k * (nx * dx + ny * dy) / r;
  return make_double2(-sc * y1(kr), sc * j1(kr));
}

// Kapur-Rokhlin end-point corrections of the punctured trapezoid rule for a log-singular kernel
// (Kapur & Rokhlin, SIAM J. Numer. Anal. 34, 1997); the values are the tables the reference applies
// (src/quadrature.c:12-41, order 10 as printed there).
__constant__ double bfKrWeights[18] = {
    1.825748064736159, -1.325748064736159,
    4.967362978287758, -16.20501504859126, 25.85153761832639, -22.22599466791883, 9.930104998037539, -1.817995878141594,
    7.832432020568779, -4.565161670374749, 1.452168846354677, -2.901348302886379, 3.870862162579900, -3.523821383570681,
    2.172421547519342, -8.707796087382991, 2.053584266072635, -2.166984103403823};

struct EvalEnvDev {
  double const *pts, *normals, *colWeights, *tpts, *tnormals;
  uint64_t const *orig;
  double k, selfRe, selfIm;
  double2 alpha, beta;
  uint64_t n;
  uint32_t krOrder, krBase;          // krBase: offset of the order's table in bfKrWeights
  unsigned long long *krHits;
};

// 1 + w_KR[d-1] if original indices a, b are d = 1..order apart on the closed curve, else 1
__device__ __forceinline__ double bfKrFactor(EvalEnvDev const &E, uint64_t a, uint64_t b, bool &hit) {
  uint64_t const fwd = a >= b ? a - b : a + E.n - b;        // (a - b) mod n
  uint64_t const d = fwd <= E.n - fwd ? fwd : E.n - fwd;
  hit = d >= 1 && d <= E.krOrder;
  return hit ? 1.0 + bfKrWeights[E.krBase + d - 1] : 1.0;
}

// kernel value for potential `pot`: (sx, sy) source normal, (tx, ty) target normal (used as needed)
__device__ __forceinline__ double2 bfKernelValue(EvalEnvDev const &E, uint32_t pot, double dx, double dy, double snx, double sny,
                                                  double tnx, double tny) {
  switch (pot) {
    case 1: return bfHelm2Sp(E.k, dx, dy, tnx, tny);
    case 2: return bfHelm2Sp(E.k, dx, dy, snx, sny);
    case 3: {
      double2 const S = bfHelm2G(E.k, dx, dy), D = bfHelm2Sp(E.k, dx, dy, snx, sny);
      return make_double2(E.alpha.x * S.x - E.alpha.y * S.y + E.beta.x * D.x - E.beta.y * D.y,
                          E.alpha.x * S.y + E.alpha.y * S.x + E.beta.x * D.y + E.beta.y * D.x);
    }
    default: return bfHelm2G(E.k, dx, dy);
  }
}

// unit normal at point i of a point set: stored normals for tree points, the radial direction for a
// sampled circle (bfCircle2SampleUnitNormals, src/circle.c:36-58)
__device__ __forceinline__ void bfNormal(EvalEnvDev const &E, BfBuildPts const &ps, uint32_t i, double &nx, double &ny) {
  if (ps.kind != BFHIP_PTS_CIRCLE) {
    double const *base = ps.kind == BFHIP_PTS_TREE ? E.normals : E.tnormals;
    nx = base[2 * (ps.first + i)];
    ny = base[2 * (ps.first + i) + 1];
  } else {
    double const theta = (6.283185307179586 / (double)ps.count) * (double)i;
    nx = cos(theta);
    ny = sin(theta);
  }
}

// one entry of a kernel matrix: target i of `tgt`, source j of `src`
__device__ __forceinline__ double2 bfKernelEntry(EvalEnvDev const &E, BfBuildPts const &src, BfBuildPts const &tgt, uint32_t i, uint32_t j,
                                                  uint32_t pot, uint32_t decorate) {
  double tx, ty, sx, sy;
  bfPoint(tgt, i, E.pts, E.tpts, tx, ty);
  bfPoint(src, j, E.pts, E.tpts, sx, sy);
  bool const bothTree = src.kind == BFHIP_PTS_TREE && tgt.kind == BFHIP_PTS_TREE;
  if (decorate && bothTree && tgt.first + i == src.first + j) return make_double2(E.selfRe, E.selfIm);
  double snx = 0, sny = 0, tnx = 0, tny = 0;
  if (pot == 1) bfNormal(E, tgt, i, tnx, tny);       // S' leaves have tree targets (checked on the host)
  if (pot >= 2) bfNormal(E, src, j, snx, sny);
  double2 g = bfKernelValue(E, pot, tx - sx, ty - sy, snx, sny, tnx, tny);
  if (decorate && E.krOrder && bothTree) {
    bool hit;
    double const f = bfKrFactor(E, E.orig[tgt.first + i], E.orig[src.first + j], hit);
    g.x *= f; g.y *= f;
    if (hit && E.krHits) atomicAdd(E.krHits, 1ull);          // bookkeeping only: every pair must be met exactly once
  }
  if (decorate && E.colWeights && src.kind == BFHIP_PTS_TREE) {
    double const w = E.colWeights[src.first + j];
    g.x *= w; g.y *= w;
  }
  return g;
}

__global__ __launch_bounds__(256) void bfEvalKernel(BfEvalMat const *mats, uint64_t const *prefix, uint32_t numMats, EvalEnvDev const E,
                                                    uint64_t tileBase) {
  uint64_t const tile = tileBase + blockIdx.x;
  uint32_t lo = 0, hi = numMats;                   // prefix[lo] <= tile < prefix[hi]
  while (hi - lo > 1) {
    uint32_t const mid = (lo + hi) >> 1;
    if (prefix[mid] <= tile) lo = mid; else hi = mid;
  }
  BfEvalMat const M = mats[lo];
  uint64_t const total = (uint64_t)M.tgt.count * M.src.count;
  uint64_t const e0 = (tile - prefix[lo]) * BF_EVAL_TILE;
  double2 *dst = (double2 *)M.dst;
#pragma unroll
  for (int q = 0; q < (int)(BF_EVAL_TILE / 256); ++q) {
    uint64_t const e = e0 + (uint64_t)q * 256 + threadIdx.x;
    if (e >= total) break;
    uint32_t const i = (uint32_t)(e % M.tgt.count), j = (uint32_t)(e / M.tgt.count);
    dst[e] = bfKernelEntry(E, M.src, M.tgt, i, j, M.pot, M.decorate);
  }
}

static EvalEnvDev toDev(BfEvalEnv const *env) {
  EvalEnvDev E;
  E.pts = (double const *)env->dPoints; E.normals = (double const *)env->dNormals; E.colWeights = (double const *)env->dColWeights;
  E.tpts = (double const *)env->dTgtPoints; E.tnormals = (double const *)env->dTgtNormals;
  E.k = env->wavenumber; E.selfRe = env->selfRe; E.selfIm = env->selfIm;
  E.alpha = make_double2(env->alphaRe, env->alphaIm); E.beta = make_double2(env->betaRe, env->betaIm);
  E.orig = (uint64_t const *)env->dOrigIndex; E.n = env->numPoints;
  E.krOrder = env->dOrigIndex ? env->krOrder : 0;
  E.krBase = env->krOrder == 2 ? 0 : env->krOrder == 6 ? 2 : 8;
  E.krHits = env->dKrHits;
  return E;
}

int bfdevBuildEval(BfEvalMat const *hostMats, uint64_t const *hostTilePrefix, uint64_t numMats, BfEvalEnv const *env) {
  if (!numMats) return 0;
  BfEvalMat *dM = NULL;
  uint64_t *dP = NULL;
  int rc = uploadArrayB(&dM, hostMats, numMats, "eval matrices");
  if (!rc) rc = uploadArrayB(&dP, hostTilePrefix, numMats + 1, "eval tile prefix");
  uint64_t const tiles = hostTilePrefix[numMats];
  for (uint64_t done = 0; done < tiles && !rc;) {
    uint32_t const n = (uint32_t)(tiles - done > (1u << 30) ? (1u << 30) : tiles - done);
    hipLaunchKernelGGL(bfEvalKernel, dim3(n), dim3(256), 0, 0, dM, dP, (uint32_t)numMats, toDev(env), done);
    rc = hipFailB(hipGetLastError(), "kernel-matrix evaluation launch");
    done += n;
  }
  if (!rc) rc = hipFailB(hipDeviceSynchronize(), "kernel-matrix evaluation");
  (void)hipFree(dM);
  (void)hipFree(dP);
  return rc;
}

// ---------------------------------------------------------------------------
// one-sided (Hestenes) Jacobi SVD, block form.  One workgroup per problem.
//
// The stacked matrix S = [A; V] (mt + me rows) carries the right singular
// vectors along: a rotation J = [[c, s], [-s e^{-i phi}, c e^{-i phi}]] of columns
// (p, q) -- phi = arg(a_p^H a_q), tan(theta) the small root of t^2 + 2 zeta t - 1,
// zeta = (|a_q|^2 - |a_p|^2) / (2 |a_p^H a_q|) -- is applied to all of S, the inner
// products use the A rows only.
//
// Traffic is what bounds this kernel: a plain sweep streams every column pair
// from memory (~80 m^3 bytes per sweep, ~30 sweeps).  So the columns are cut in
// blocks of b; a pair of blocks (2b stacked columns) is staged in LDS, swept
// completely there (round-robin over the 2b columns: b disjoint pairs per step
// on the workgroup's lane groups), and written back; block pairs follow the same
// round-robin ordering.  Memory traffic per sweep drops by ~2.5 b, and when all
// columns fit (me <= 2b, the bulk of the problems) the matrix is loaded once
// and never leaves LDS until it has converged.
// ---------------------------------------------------------------------------
#define BF_JACOBI_MAX_SWEEPS 40
#define BF_JACOBI_LDS_MAX (144u << 10)

template <int W> __device__ __forceinline__ double bfGroupSum(double v) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) v += __shfl_xor(v, m, W);
  return v;
}

// pair kk of step s of a round-robin tournament over M (even) players
__device__ __forceinline__ void bfRoundRobin(uint32_t M, uint32_t s, uint32_t kk, uint32_t &p, uint32_t &q) {
  p = kk == 0 ? M - 1 : (s + kk) % (M - 1);
  q = kk == 0 ? s : (s + (M - 1) - kk) % (M - 1);
  if (p > q) { uint32_t const t = p; p = q; q = t; }
}

// singular values, the reference's truncation rule (src/mat_dense_complex.c:1800-1812) and the
// statistics, once the columns of A are orthogonal
template <int W>
__device__ void bfJacobiFinish(BfSvdProb const &P, BfSvdStats *stats, int sweep, bool converged, double *sigMaxShared) {
  uint32_t const mt = P.mt, me = P.me;
  double2 const *A = (double2 const *)P.a;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;
  for (uint32_t j = g; j < me; j += groups) {
    double2 const *aj = A + (uint64_t)j * mt;
    double s2 = 0;
    for (uint32_t r = l; r < mt; r += W) { double2 const a = aj[r]; s2 = fma(a.x, a.x, fma(a.y, a.y, s2)); }
    s2 = bfGroupSum<W>(s2);
    if (l == 0) P.scale[j] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    double mx = 0;
    for (uint32_t j = 0; j < me; ++j) mx = fmax(mx, P.scale[j]);
    *sigMaxShared = sqrt(mx);
  }
  __syncthreads();
  double const eps = 2.220446049250313e-16;
  double const tol = (double)P.dim * eps * *sigMaxShared + eps;
  unsigned long long dropped = 0;
  for (uint32_t j = tid; j < me; j += nthreads) {
    double const s2 = P.scale[j];
    bool const keep = sqrt(s2) >= tol;
    P.scale[j] = keep ? 1.0 / s2 : 0.0;
    dropped += keep ? 0 : 1;
  }
  // statistics only (not part of any result)
  if (dropped) atomicAdd(&stats->truncated, dropped);
  if (tid == 0) {
    atomicMax(&stats->maxSweeps, (unsigned long long)(sweep + (converged ? 1 : 0)));
    atomicAdd(&stats->sumSweeps, (unsigned long long)(sweep + (converged ? 1 : 0)));
    if (!converged) atomicAdd(&stats->notConverged, 1ull);
  }
}

// largest squared column norm of A (a lower bound of sigma_max^2), order-independent
template <int W>
__device__ double bfJacobiMaxNorm2(BfSvdProb const &P, unsigned long long *maxBitsShared) {
  uint32_t const mt = P.mt, me = P.me;
  double2 const *A = (double2 const *)P.a;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;
  if (tid == 0) *maxBitsShared = 0;
  __syncthreads();
  for (uint32_t j = g; j < me; j += groups) {
    double2 const *aj = A + (uint64_t)j * mt;
    double s2 = 0;
    for (uint32_t r = l; r < mt; r += W) { double2 const a = aj[r]; s2 = fma(a.x, a.x, fma(a.y, a.y, s2)); }
    s2 = bfGroupSum<W>(s2);
    if (l == 0) atomicMax(maxBitsShared, (unsigned long long)__double_as_longlong(s2));
  }
  __syncthreads();
  return __longlong_as_double((long long)*maxBitsShared);
}

// rotation of a column pair from its inner products (alpha, beta, gamma); false: below the threshold
__device__ __forceinline__ bool bfJacobiAngle(double alpha, double beta, double gr, double gi, double tol2, double dead2,
                                               double &c, double &sn, double &er, double &ei) {
  double const g2 = gr * gr + gi * gi;
  if (alpha < dead2 || beta < dead2) return false;
  if (!(g2 > tol2 * alpha * beta) || g2 == 0.0) return false;
  double const gabs = sqrt(g2);
  double const zeta = (beta - alpha) / (2.0 * gabs);
  double const t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
  c = 1.0 / sqrt(1.0 + t * t); sn = c * t;
  er = gr / gabs; ei = -gi / gabs;                // e^{-i phi}
  return true;
}

#define BF_JACOBI_MAX_COLS 2304          /* 2 (rows + cols) 16 B <= the LDS tile  =>  cols <= 2300 */

template <int W>
__global__ __launch_bounds__(1024) void bfJacobiKernel(BfSvdProb const *probs, uint32_t const *list, BfSvdStats *stats, uint32_t ldsBytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bfJacobiLds[];
  double2 *tile = (double2 *)bfJacobiLds;
  __shared__ int rotated;
  __shared__ double sigMax;
  __shared__ unsigned long long maxNormBits;
  __shared__ uint32_t numLive;
  __shared__ uint16_t live[BF_JACOBI_MAX_COLS];     // the columns still above the threshold, in column order
  BfSvdProb const P = probs[list[blockIdx.x]];
  uint32_t const mt = P.mt, me = P.me;
  if (me == 0) return;                              // rank 0 after the QR preconditioner: nothing to orthogonalise
  double2 *A = (double2 *)P.a, *V = (double2 *)P.v;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;
  uint32_t const R = mt + me, Rp = R | 1u;          // odd stride (in 16-byte units): groups of a wave spread over the banks
  uint32_t C = (ldsBytes / 16u) / Rp;               // stacked columns the tile holds
  C = C < 2 ? 2 : C & ~1u;
  uint32_t const b = C / 2 < (me + 1) / 2 ? C / 2 : (me + 1) / 2;
  bool const resident = (me + b - 1) / b <= 2;      // everything fits: loaded once, slot = column
  double const tol2 = (double)mt * 2.220446049250313e-16 * 2.220446049250313e-16;   // xGESVJ: sqrt(mt) eps

  // Columns below the truncation threshold (relative to the largest column norm, a lower bound of
  // sigma_max) are never rotated: they will be dropped, and what they carry is below the rounding
  // error of the matrix.  Left alone they would keep the sweeps busy orthogonalising noise.  A column
  // that is never rotated never changes, so it stays below: every sweep starts by listing the columns
  // still above the threshold and visits the pairs of THAT list only (with half of the columns
  // frozen, a quarter of the pairs).
  double const deadRel = (double)P.dim * 2.220446049250313e-16;
  double const dead2 = deadRel * deadRel * bfJacobiMaxNorm2<W>(P, &maxNormBits);

  // slot of the staged block pair (I, J) of the live list -> global column; none: 0xffffffff
  uint32_t nLive = me;
  auto slotCol = [&](uint32_t I, uint32_t J, uint32_t slot) -> uint32_t {
    uint32_t const blk = slot < b ? I : J, off = slot < b ? slot : slot - b;
    uint32_t const a = blk * b + off;
    return a < nLive ? (uint32_t)live[a] : 0xffffffffu;
  };
  auto loadPair = [&](uint32_t I, uint32_t J) {
    for (uint32_t e = tid; e < 2 * b * R; e += nthreads) {
      uint32_t const slot = e / R, r = e - slot * R;
      uint32_t const c = slotCol(I, J, slot);
      if (c == 0xffffffffu) continue;
      tile[slot * Rp + r] = r < mt ? A[(uint64_t)c * mt + r] : V[(uint64_t)c * me + (r - mt)];
    }
  };
  auto storePair = [&](uint32_t I, uint32_t J) {
    for (uint32_t e = tid; e < 2 * b * R; e += nthreads) {
      uint32_t const slot = e / R, r = e - slot * R;
      uint32_t const c = slotCol(I, J, slot);
      if (c == 0xffffffffu) continue;
      double2 const v = tile[slot * Rp + r];
      if (r < mt) A[(uint64_t)c * mt + r] = v; else V[(uint64_t)c * me + (r - mt)] = v;
    }
  };
  // one column pair held in the tile at slots (p, q): inner products over the A rows, rotation of all R rows
  auto rotatePair = [&](uint32_t p, uint32_t q) {
    double2 *sp = tile + p * Rp, *sq = tile + q * Rp;
    double alpha = 0, beta = 0, gr = 0, gi = 0;
    for (uint32_t r = l; r < mt; r += W) {
      double2 const x = sp[r], y = sq[r];
      alpha = fma(x.x, x.x, fma(x.y, x.y, alpha));
      beta = fma(y.x, y.x, fma(y.y, y.y, beta));
      gr = fma(x.x, y.x, fma(x.y, y.y, gr));     // conj(x) * y
      gi = fma(x.x, y.y, fma(-x.y, y.x, gi));
    }
    alpha = bfGroupSum<W>(alpha); beta = bfGroupSum<W>(beta);
    gr = bfGroupSum<W>(gr); gi = bfGroupSum<W>(gi);
    double c, sn, er, ei;
    if (!bfJacobiAngle(alpha, beta, gr, gi, tol2, dead2, c, sn, er, ei)) return;
    for (uint32_t r = l; r < R; r += W) {
      double2 const x = sp[r], y = sq[r];
      double2 const yt = make_double2(er * y.x - ei * y.y, er * y.y + ei * y.x);
      sp[r] = make_double2(c * x.x - sn * yt.x, c * x.y - sn * yt.y);
      sq[r] = make_double2(sn * x.x + c * yt.x, sn * x.y + c * yt.y);
    }
    if (l == 0) rotated = 1;
  };

  // V starts as the identity
  if (resident) {
    for (uint32_t e = tid; e < me * R; e += nthreads) {
      uint32_t const c = e / R, r = e - c * R;
      tile[c * Rp + r] = r < mt ? A[(uint64_t)c * mt + r] : make_double2(r - mt == c ? 1.0 : 0.0, 0.0);
    }
  } else {
    for (uint64_t e = tid; e < (uint64_t)me * me; e += nthreads) V[e] = make_double2((e % me == e / me) ? 1.0 : 0.0, 0.0);
  }
  int sweep = 0;
  bool converged = false;
  __syncthreads();
  for (; sweep < BF_JACOBI_MAX_SWEEPS; ++sweep) {
    // the live list of this sweep (P.scale is free until the end: scratch for the squared norms)
    for (uint32_t j = g; j < me; j += groups) {
      double2 const *aj = resident ? tile + j * Rp : A + (uint64_t)j * mt;
      double s2 = 0;
      for (uint32_t r = l; r < mt; r += W) { double2 const a = aj[r]; s2 = fma(a.x, a.x, fma(a.y, a.y, s2)); }
      s2 = bfGroupSum<W>(s2);
      if (l == 0) P.scale[j] = s2;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t n = 0;
      for (uint32_t j = 0; j < me; ++j)
        if (P.scale[j] >= dead2) live[n++] = (uint16_t)j;
      numLive = n;
      rotated = 0;
    }
    __syncthreads();
    nLive = numLive;
    if (nLive < 2) { converged = true; break; }
    if (resident) {
      uint32_t const M = nLive + (nLive & 1u);
      for (uint32_t s = 0; s + 1 < M; ++s) {
        for (uint32_t kk = g; kk < M / 2; kk += groups) {
          uint32_t p, q;
          bfRoundRobin(M, s, kk, p, q);
          if (q >= nLive) continue;                  // the dummy player of an odd count
          rotatePair(live[p], live[q]);
        }
        __syncthreads();
      }
    } else {
      uint32_t nb = (nLive + b - 1) / b;
      nb = nb < 2 ? 2 : nb;
      uint32_t const NB = nb + (nb & 1u);
      for (uint32_t S = 0; S + 1 < NB; ++S) {
        for (uint32_t KK = 0; KK < NB / 2; ++KK) {
          uint32_t I, J;
          bfRoundRobin(NB, S, KK, I, J);
          if (J >= nb) continue;                       // the dummy block of an odd block count
          loadPair(I, J);
          __syncthreads();
          // one complete sweep over the 2b staged columns
          for (uint32_t s = 0; s + 1 < 2 * b; ++s) {
            for (uint32_t kk = g; kk < b; kk += groups) {
              uint32_t p, q;
              bfRoundRobin(2 * b, s, kk, p, q);
              if (slotCol(I, J, p) == 0xffffffffu || slotCol(I, J, q) == 0xffffffffu) continue;
              rotatePair(p, q);
            }
            __syncthreads();
          }
          storePair(I, J);
          __syncthreads();
        }
      }
    }
    converged = rotated == 0;
    __syncthreads();
    if (converged) break;
  }
  if (resident) {
    for (uint32_t e = tid; e < me * R; e += nthreads) {
      uint32_t const c = e / R, r = e - c * R;
      double2 const v = tile[c * Rp + r];
      if (r < mt) A[(uint64_t)c * mt + r] = v; else V[(uint64_t)c * me + (r - mt)] = v;
    }
  }
  __syncthreads();
  bfJacobiFinish<W>(P, stats, sweep, converged, &sigMax);
}

// ---------------------------------------------------------------------------
// Block form for the problems whose stacked columns are long for the LDS tile (rows + columns >= 512: only 2 - 8
// stacked columns of a block fit, an inner step keeps 2 - 5 of 16 wavefronts busy and every block pair is a round trip
// of its columns for a handful of rotations).  Here a block is 16 columns whatever their length.  For a pair of blocks
// (32 columns Xp):
//   A. G = Xp^H Xp, 32 x 32, one pass over the rows (32-row chunks staged in LDS, 4 x 4 register blocks, 8 row slices;
//      accumulating only the upper triangle -- 36 blocks x 14 slices -- measured no faster);
//   B. two-sided Jacobi on G in LDS (16 disjoint rotations per round-robin step, four to a wavefront on each of the four
//      SIMDs, 16 lanes each; the same rotation formula and the same thresholds as the scalar kernel -- the inner products it would
//      compute are the entries of G), one sweep, the rotations accumulated in Q (32 x 32);
//   C. [Xp; Vp] <- [Xp; Vp] Q, one thread per row, the row's 32 values in registers, Q broadcast from LDS.
// The next outer sweep forms every G afresh from the columns, so what an inner solve leaves undone (its later rotations
// use updated, not recomputed, inner products) is met again; a sweep in which no fresh G holds a pair above the
// threshold ends the iteration, exactly the scalar kernel's criterion.  (CPU prototype on re-expansion matrices of
// 160 - 300 columns: 6 - 9 outer sweeps against 9 scalar ones, V orthogonal to 1e-13.)  Traffic per sweep falls by
// the block size over the tile's 2 - 5 columns, and the arithmetic is two GEMM-shaped passes all wavefronts share.
// ---------------------------------------------------------------------------
#define BF_GRAM_NB 16
#define BF_GRAM_P 32
#define BF_GRAM_THREADS 512
#define BF_GRAM_RC 32
#define BF_GRAM_LD 33
#define BF_GRAM_MAX_COLS 4096
#define BF_GRAM_NONE 0xffffu
#define BF_GRAM_INNER 1                  /* sweeps of the inner solve per visit of a block pair (2: 10.2 instead of 10.5 outer sweeps on
                                          * average, but the solve -- then by one wavefront -- was half of a visit's time: 4.66 against 3.76 s per batch) */


__global__ __launch_bounds__(BF_GRAM_THREADS) void bfJacobiGramKernel(BfSvdProb const *probs, uint32_t const *list, BfSvdStats *stats) {
  constexpr int W = 64;
  __shared__ __attribute__((aligned(16))) double2 G[BF_GRAM_P * BF_GRAM_LD];
  __shared__ __attribute__((aligned(16))) double2 Q[BF_GRAM_P * BF_GRAM_LD];
  __shared__ __attribute__((aligned(16))) double2 tile[BF_GRAM_P * BF_GRAM_LD];      // [column][row of the chunk]
  __shared__ uint16_t live[BF_GRAM_MAX_COLS];
  __shared__ uint16_t pcol[BF_GRAM_P];
  __shared__ int rotated, pairRot;
  __shared__ double sigMax;
  __shared__ unsigned long long maxNormBits;
  __shared__ uint32_t numLive;
  BfSvdProb const P = probs[list[blockIdx.x]];
  uint32_t const mt = P.mt, me = P.me;
  if (me == 0) return;
  double2 *A = (double2 *)P.a, *V = (double2 *)P.v;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;
  double const tol2 = (double)mt * 2.220446049250313e-16 * 2.220446049250313e-16;
  double const deadRel = (double)P.dim * 2.220446049250313e-16;
  double const dead2 = deadRel * deadRel * bfJacobiMaxNorm2<W>(P, &maxNormBits);
  for (uint64_t e = tid; e < (uint64_t)me * me; e += nthreads) V[e] = make_double2((e % me == e / me) ? 1.0 : 0.0, 0.0);
  int sweep = 0;
  bool converged = false;
  __syncthreads();
  for (; sweep < BF_JACOBI_MAX_SWEEPS; ++sweep) {
    for (uint32_t j = g; j < me; j += groups) {
      double2 const *aj = A + (uint64_t)j * mt;
      double s2 = 0;
      for (uint32_t r = l; r < mt; r += W) { double2 const a = aj[r]; s2 = fma(a.x, a.x, fma(a.y, a.y, s2)); }
      s2 = bfGroupSum<W>(s2);
      if (l == 0) P.scale[j] = s2;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t n = 0;
      for (uint32_t j = 0; j < me; ++j)
        if (P.scale[j] >= dead2) live[n++] = (uint16_t)j;
      numLive = n;
      rotated = 0;
    }
    __syncthreads();
    uint32_t const nLive = numLive;
    if (nLive < 2) { converged = true; break; }
    uint32_t nb = (nLive + BF_GRAM_NB - 1) / BF_GRAM_NB;
    nb = nb < 2 ? 2 : nb;
    uint32_t const NBk = nb + (nb & 1u);
    for (uint32_t S = 0; S + 1 < NBk; ++S) {
      for (uint32_t KK = 0; KK < NBk / 2; ++KK) {
        uint32_t I, J;
        bfRoundRobin(NBk, S, KK, I, J);
        if (J >= nb) continue;
        if (tid < BF_GRAM_P) {
          uint32_t const a = (tid < BF_GRAM_NB ? I : J) * BF_GRAM_NB + (tid % BF_GRAM_NB);
          pcol[tid] = a < nLive ? live[a] : (uint16_t)BF_GRAM_NONE;
        }
        for (uint32_t e = tid; e < BF_GRAM_P * BF_GRAM_LD; e += nthreads) {
          G[e] = make_double2(0.0, 0.0);
          uint32_t const i = e / BF_GRAM_LD, j = e - i * BF_GRAM_LD;
          Q[e] = make_double2(i == j ? 1.0 : 0.0, 0.0);
        }
        if (tid == 0) pairRot = 0;
        __syncthreads();
        // ---- A: the Gram matrix of the pair's columns
        {
          uint32_t const slice = tid / 64, w = tid % 64, bp = w / 8, bq = w % 8;
          double ar[4][4], ai[4][4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { ar[i][j] = 0; ai[i][j] = 0; }
          for (uint32_t r0 = 0; r0 < mt; r0 += BF_GRAM_RC) {
#pragma unroll
            for (int k = 0; k < (BF_GRAM_P * BF_GRAM_RC) / BF_GRAM_THREADS; ++k) {
              uint32_t const e = tid + BF_GRAM_THREADS * k, c = e / BF_GRAM_RC, r = e % BF_GRAM_RC;
              uint32_t const col = pcol[c];
              double2 v = make_double2(0.0, 0.0);
              if (col != BF_GRAM_NONE && r0 + r < mt) v = A[(uint64_t)col * mt + r0 + r];
              tile[c * BF_GRAM_LD + r] = v;
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              uint32_t const r = slice * 4 + rr;
              double2 a[4], b[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) { a[i] = tile[(4 * bp + i) * BF_GRAM_LD + r]; b[i] = tile[(4 * bq + i) * BF_GRAM_LD + r]; }
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {            // conj(a) * b
                  ar[i][j] = fma(a[i].x, b[j].x, fma(a[i].y, b[j].y, ar[i][j]));
                  ai[i][j] = fma(a[i].x, b[j].y, fma(-a[i].y, b[j].x, ai[i][j]));
                }
            }
            __syncthreads();
          }
          for (uint32_t sl = 0; sl < BF_GRAM_THREADS / 64; ++sl) {      // the row slices added in a fixed order
            if (slice == sl) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  double2 &gg = G[(4 * bp + i) * BF_GRAM_LD + 4 * bq + j];
                  gg = make_double2(gg.x + ar[i][j], gg.y + ai[i][j]);
                }
            }
            __syncthreads();
          }
        }
        // ---- B: two-sided Jacobi on G; Q accumulates the rotations.  Wavefronts 0 - 3 (one per SIMD) take four of the 16
        // disjoint rotations of a round-robin step each, 16 lanes per rotation; the others only keep the barriers.
        {
          bool const worker = tid < 256;
          uint32_t const kk = tid / 16, sub = tid % 16;
          for (int inner = 0; inner < BF_GRAM_INNER; ++inner) {
            for (uint32_t s = 0; s + 1 < BF_GRAM_P; ++s) {
              uint32_t p = 0, q = 1;
              double c = 1, sn = 0, er = 1, ei = 0;
              bool rot = false;
              if (worker) {
                bfRoundRobin(BF_GRAM_P, s, kk, p, q);
                double2 const gpq = G[p * BF_GRAM_LD + q];
                rot = bfJacobiAngle(G[p * BF_GRAM_LD + p].x, G[q * BF_GRAM_LD + q].x, gpq.x, gpq.y, tol2, dead2, c, sn, er, ei);
              }
              __syncthreads();                                       // every rotation of the step has read its inputs
              if (rot) {
                if (inner == 0 && sub == 0) { pairRot = 1; rotated = 1; }
#pragma unroll
                for (int t = 0; t < BF_GRAM_P / 16; ++t) {                // columns p, q of G and of Q
                  uint32_t const i = sub + 16 * t;
                  double2 x = G[i * BF_GRAM_LD + p], y = G[i * BF_GRAM_LD + q];
                  double2 yt = make_double2(er * y.x - ei * y.y, er * y.y + ei * y.x);
                  G[i * BF_GRAM_LD + p] = make_double2(c * x.x - sn * yt.x, c * x.y - sn * yt.y);
                  G[i * BF_GRAM_LD + q] = make_double2(sn * x.x + c * yt.x, sn * x.y + c * yt.y);
                  x = Q[i * BF_GRAM_LD + p]; y = Q[i * BF_GRAM_LD + q];
                  yt = make_double2(er * y.x - ei * y.y, er * y.y + ei * y.x);
                  Q[i * BF_GRAM_LD + p] = make_double2(c * x.x - sn * yt.x, c * x.y - sn * yt.y);
                  Q[i * BF_GRAM_LD + q] = make_double2(sn * x.x + c * yt.x, sn * x.y + c * yt.y);
                }
              }
              __syncthreads();
              if (rot) {
#pragma unroll
                for (int t = 0; t < BF_GRAM_P / 16; ++t) {                // rows p, q of G: J^H G, conj(e) on row q
                  uint32_t const j = sub + 16 * t;
                  double2 const x = G[p * BF_GRAM_LD + j], y = G[q * BF_GRAM_LD + j];
                  double2 const yt = make_double2(er * y.x + ei * y.y, er * y.y - ei * y.x);
                  G[p * BF_GRAM_LD + j] = make_double2(c * x.x - sn * yt.x, c * x.y - sn * yt.y);
                  G[q * BF_GRAM_LD + j] = make_double2(sn * x.x + c * yt.x, sn * x.y + c * yt.y);
                }
              }
              __syncthreads();
            }
          }
        }
        __syncthreads();
        // ---- C: the pair's columns of [X; V] times Q
        if (pairRot) {
          uint32_t const R = mt + me;
          for (uint32_t row0 = 0; row0 < R; row0 += nthreads) {
            uint32_t const row = row0 + tid;
            if (row < R) {
              double2 *base = row < mt ? A + row : V + (row - mt);
              uint64_t const ld = row < mt ? mt : me;
              double2 x[BF_GRAM_P];
#pragma unroll
              for (int k = 0; k < BF_GRAM_P; ++k) {
                uint32_t const col = pcol[k];
                x[k] = col != BF_GRAM_NONE ? base[(uint64_t)col * ld] : make_double2(0.0, 0.0);
              }
#pragma unroll 4
              for (int c = 0; c < BF_GRAM_P; ++c) {
                uint32_t const col = pcol[c];
                if (col == BF_GRAM_NONE) continue;
                double yr = 0, yi = 0;
#pragma unroll
                for (int k = 0; k < BF_GRAM_P; ++k) {
                  double2 const qk = Q[k * BF_GRAM_LD + c];
                  yr = fma(x[k].x, qk.x, fma(-x[k].y, qk.y, yr));
                  yi = fma(x[k].x, qk.y, fma(x[k].y, qk.x, yi));
                }
                base[(uint64_t)col * ld] = make_double2(yr, yi);
              }
            }
          }
        }
        __syncthreads();
      }
    }
    converged = rotated == 0;
    __syncthreads();
    if (converged) break;
  }
  __syncthreads();
  bfJacobiFinish<W>(P, stats, sweep, converged, &sigMax);
}

// Fallback for problems whose stacked column pair does not fit the LDS tile (check points +
// equivalent sources > ~4600): the same sweeps straight out of global memory, one column pair per
// 64-lane group of a 1024-thread workgroup.  Slow (every rotation streams its columns), but total.
__global__ __launch_bounds__(1024) void bfJacobiGlobalKernel(BfSvdProb const *probs, uint32_t const *list, BfSvdStats *stats) {
  constexpr int W = 64;
  __shared__ int rotated;
  __shared__ double sigMax;
  __shared__ unsigned long long maxNormBits;
  BfSvdProb const P = probs[list[blockIdx.x]];
  uint32_t const mt = P.mt, me = P.me;
  double2 *A = (double2 *)P.a, *V = (double2 *)P.v;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;
  double const tol2 = (double)mt * 2.220446049250313e-16 * 2.220446049250313e-16;
  double const deadRel = (double)P.dim * 2.220446049250313e-16;
  double const dead2 = deadRel * deadRel * bfJacobiMaxNorm2<W>(P, &maxNormBits);
  for (uint64_t e = tid; e < (uint64_t)me * me; e += nthreads) V[e] = make_double2((e % me == e / me) ? 1.0 : 0.0, 0.0);
  uint32_t const M = me + (me & 1u);
  int sweep = 0;
  bool converged = false;
  __syncthreads();
  for (; sweep < BF_JACOBI_MAX_SWEEPS; ++sweep) {
    if (tid == 0) rotated = 0;
    __syncthreads();
    for (uint32_t s = 0; s + 1 < M; ++s) {
      for (uint32_t kk = g; kk < M / 2; kk += groups) {
        uint32_t p, q;
        bfRoundRobin(M, s, kk, p, q);
        if (q >= me) continue;                       // the dummy column of an odd me
        double2 *ap = A + (uint64_t)p * mt, *aq = A + (uint64_t)q * mt;
        double alpha = 0, beta = 0, gr = 0, gi = 0;
        for (uint32_t r = l; r < mt; r += W) {
          double2 const x = ap[r], y = aq[r];
          alpha = fma(x.x, x.x, fma(x.y, x.y, alpha));
          beta = fma(y.x, y.x, fma(y.y, y.y, beta));
          gr = fma(x.x, y.x, fma(x.y, y.y, gr));
          gi = fma(x.x, y.y, fma(-x.y, y.x, gi));
        }
        alpha = bfGroupSum<W>(alpha); beta = bfGroupSum<W>(beta);
        gr = bfGroupSum<W>(gr); gi = bfGroupSum<W>(gi);
        double c, sn, er, ei;
        if (!bfJacobiAngle(alpha, beta, gr, gi, tol2, dead2, c, sn, er, ei)) continue;
        double2 *vp = V + (uint64_t)p * me, *vq = V + (uint64_t)q * me;
        for (uint32_t r = l; r < mt + me; r += W) {
          double2 *xp = r < mt ? ap + r : vp + (r - mt), *yp = r < mt ? aq + r : vq + (r - mt);
          double2 const x = *xp, y = *yp;
          double2 const yt = make_double2(er * y.x - ei * y.y, er * y.y + ei * y.x);
          *xp = make_double2(c * x.x - sn * yt.x, c * x.y - sn * yt.y);
          *yp = make_double2(sn * x.x + c * yt.x, sn * x.y + c * yt.y);
        }
        if (l == 0) rotated = 1;
      }
      __syncthreads();
    }
    converged = rotated == 0;
    __syncthreads();
    if (converged) break;
  }
  bfJacobiFinish<W>(P, stats, sweep, converged, &sigMax);
}

// ---------------------------------------------------------------------------
// Preconditioner of the Jacobi SVD for the problems that do not fit LDS (Drmac & Veselic's
// scheme, first stage): Householder QR with column pivoting  A P = Q R,  stopped at the first step
// whose largest remaining column is below the threshold under which the Jacobi kernel never
// rotates a column (r steps).  The right-hand side B rides along as extra, never pivoted
// columns and leaves as Q^H B.  What the Jacobi kernel then orthogonalises is
//   X = (R[0:r, :] P^T)^H      (me x r, rows in the ORIGINAL column order of A),
// whose columns are already graded and few: X V1 = W (orthogonal columns, norms sigma)  gives
//   A ~ (Q[:, 0:r] V1) (W^H)  and  pinv(A) B = W diag(1/sigma^2) V1^H (Q^H B)[0:r, :].
// One workgroup per problem; a column of the trailing matrix per 64-lane group (its segment in
// registers between the dot product and the update when it has <= 1024 rows); the squared norm
// of every updated column is accumulated from the updated values themselves, so the pivot
// choice never sees a downdated norm.  The reflector lives in LDS.  All orders are fixed.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void bfQrcpKernel(BfQrProb const *probs, uint32_t *ranks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bfQrLds[];
  constexpr int W = 64;
  BfQrProb const P = probs[blockIdx.x];
  uint32_t const mt = P.mt, me = P.me, n = P.n;
  double2 *A = (double2 *)P.a, *B = (double2 *)P.b, *X = (double2 *)P.x;
  double2 *u = (double2 *)bfQrLds;                       // [mt]
  double *cn = (double *)(u + mt);                       // [me] squared norms of rows j.. of the columns
  uint32_t *perm = (uint32_t *)(cn + me);                // [me]
  __shared__ double redVal[16];
  __shared__ uint32_t redIdx[16];
  __shared__ double sCoef, sDead2;
  __shared__ uint32_t sPivot;
  uint32_t const nthreads = blockDim.x, tid = threadIdx.x;
  uint32_t const groups = nthreads / W, g = tid / W, l = tid % W;

  // largest cn[c], c >= from (lowest index among equals) -> sPivot; every thread gets the value
  auto argmax = [&](uint32_t from) -> double {
    double bv = -1.0; uint32_t bi = 0xffffffffu;
    for (uint32_t c = from + tid; c < me; c += nthreads) { double const v = cn[c]; if (v > bv) { bv = v; bi = c; } }
#pragma unroll
    for (int m = 1; m < W; m <<= 1) {
      double const ov = __shfl_xor(bv, m, W); uint32_t const oi = __shfl_xor(bi, m, W);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (l == 0) { redVal[g] = bv; redIdx[g] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (uint32_t k = 1; k < groups; ++k)
        if (redVal[k] > bv || (redVal[k] == bv && redIdx[k] < bi)) { bv = redVal[k]; bi = redIdx[k]; }
      redVal[0] = bv; sPivot = bi;
    }
    __syncthreads();
    return redVal[0];
  };

  for (uint32_t c = g; c < me; c += groups) {
    double2 const *ac = A + (uint64_t)c * mt;
    double s2 = 0;
    for (uint32_t r = l; r < mt; r += W) { double2 const a = ac[r]; s2 = fma(a.x, a.x, fma(a.y, a.y, s2)); }
    s2 = bfGroupSum<W>(s2);
    if (l == 0) { cn[c] = s2; perm[c] = c; }
  }
  __syncthreads();
  {
    double const mx = argmax(0);
    double const deadRel = (double)P.dim * 2.220446049250313e-16;
    if (tid == 0) sDead2 = deadRel * deadRel * mx;
    __syncthreads();
  }
  uint32_t const steps = mt < me ? mt : me;
  uint32_t j = 0;
  for (; j < steps; ++j) {
    double const best = argmax(j);
    if (!(best >= sDead2) || best <= 0.0) break;       // everything left is below the threshold (or not a number)
    uint32_t const p = sPivot;
    if (p != j) {
      double2 *aj = A + (uint64_t)j * mt, *ap = A + (uint64_t)p * mt;
      for (uint32_t r = tid; r < mt; r += nthreads) { double2 const t = aj[r]; aj[r] = ap[r]; ap[r] = t; }
      if (tid == 0) { double const t = cn[j]; cn[j] = cn[p]; cn[p] = t; uint32_t const q = perm[j]; perm[j] = perm[p]; perm[p] = q; }
    }
    __syncthreads();
    // reflector H = I - coef u u^H with u = x + e^{i arg x0} |x| e_1:  H x = -e^{i arg x0} |x| e_1
    uint32_t const L = mt - j;
    double2 *aj = A + (uint64_t)j * mt + j;
    for (uint32_t i = tid; i < L; i += nthreads) u[i] = aj[i];
    __syncthreads();
    if (tid == 0) {
      double2 const x0 = u[0];
      double const nx = sqrt(best), a0 = hypot(x0.x, x0.y);
      double const pr = a0 > 0.0 ? x0.x / a0 : 1.0, pi = a0 > 0.0 ? x0.y / a0 : 0.0;
      u[0] = make_double2(x0.x + pr * nx, x0.y + pi * nx);
      sCoef = 1.0 / (nx * (nx + a0));                  // 2 / (u^H u)
      aj[0] = make_double2(-pr * nx, -pi * nx);          // R[j][j]
    }
    __syncthreads();
    double const coef = sCoef;
    uint32_t const trailing = me - j - 1, cols = trailing + n;
    for (uint32_t c = g; c < cols; c += groups) {
      double2 *col = c < trailing ? A + (uint64_t)(j + 1 + c) * mt + j : B + (uint64_t)(c - trailing) * mt + j;
      double sr = 0, si = 0, nn = 0;
      if (L <= 16 * W) {
        double2 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          uint32_t const i = l + W * t;
          v[t] = i < L ? col[i] : make_double2(0.0, 0.0);
          double2 const ui = i < L ? u[i] : make_double2(0.0, 0.0);
          sr = fma(ui.x, v[t].x, fma(ui.y, v[t].y, sr));     // conj(u) * v
          si = fma(ui.x, v[t].y, fma(-ui.y, v[t].x, si));
        }
        sr = bfGroupSum<W>(sr) * coef; si = bfGroupSum<W>(si) * coef;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          uint32_t const i = l + W * t;
          if (i < L) {
            double2 const ui = u[i];
            double2 const w = make_double2(v[t].x - (sr * ui.x - si * ui.y), v[t].y - (sr * ui.y + si * ui.x));
            col[i] = w;
            if (i) nn = fma(w.x, w.x, fma(w.y, w.y, nn));
          }
        }
      } else {
        for (uint32_t i = l; i < L; i += W) {
          double2 const ui = u[i], x = col[i];
          sr = fma(ui.x, x.x, fma(ui.y, x.y, sr));
          si = fma(ui.x, x.y, fma(-ui.y, x.x, si));
        }
        sr = bfGroupSum<W>(sr) * coef; si = bfGroupSum<W>(si) * coef;
        for (uint32_t i = l; i < L; i += W) {
          double2 const ui = u[i], x = col[i];
          double2 const w = make_double2(x.x - (sr * ui.x - si * ui.y), x.y - (sr * ui.y + si * ui.x));
          col[i] = w;
          if (i) nn = fma(w.x, w.x, fma(w.y, w.y, nn));
        }
      }
      nn = bfGroupSum<W>(nn);
      if (l == 0 && c < trailing) cn[j + 1 + c] = nn;
    }
    __syncthreads();
  }
  uint32_t const r = j;
  if (tid == 0) ranks[blockIdx.x] = r;
  // X[perm[c] + i me] = conj(R[i][c]), i <= c; zero below the diagonal of R
  for (uint32_t c = g; c < me; c += groups) {
    double2 const *rc = A + (uint64_t)c * mt;
    uint32_t const row = perm[c];
    for (uint32_t i = l; i < r; i += W) {
      double2 v = make_double2(0.0, 0.0);
      if (i <= c) { v = rc[i]; v.y = -v.y; }
      X[(uint64_t)i * me + row] = v;
    }
  }
}

// LDS a problem of the QR kernel needs; 0: does not fit (the caller keeps such a problem on the plain path)
static uint32_t qrcpLds(uint32_t mt, uint32_t me) {
  uint64_t const need = (uint64_t)mt * 16 + (uint64_t)me * 12 + 64;
  return need <= (150u << 10) ? (uint32_t)need : 0;
}

int bfdevQrcpFits(uint32_t mt, uint32_t me) { return qrcpLds(mt, me) != 0; }

typedef struct QrOrder { double cost; uint32_t idx; int cls; } QrOrder;
static int qrOrderCmp(void const *pa, void const *pb) {
  QrOrder const *a = (QrOrder const *)pa, *b = (QrOrder const *)pb;
  if (a->cls != b->cls) return a->cls < b->cls ? -1 : 1;
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  return a->idx < b->idx ? -1 : 1;
}

// one launch per LDS class (several small problems then share a CU)
int bfdevBuildQrcp(BfQrProb const *hostProbs, uint64_t numProbs, uint32_t *hostRanks) {
  if (!numProbs) return 0;
  enum { NC = 6 };                                       /* <= 8, 16, 32, 64, 128, 150 KiB */
  static uint32_t const cap[NC] = {8u << 10, 16u << 10, 32u << 10, 64u << 10, 128u << 10, 150u << 10};
  uint32_t *order = (uint32_t *)malloc(numProbs * sizeof(uint32_t));
  BfQrProb *sorted = (BfQrProb *)malloc(numProbs * sizeof(BfQrProb));
  uint32_t *ranks = (uint32_t *)malloc(numProbs * sizeof(uint32_t));
  uint64_t count[NC + 1] = {0};
  int rc = order && sorted && ranks ? 0 : bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  QrOrder *ord = (QrOrder *)malloc(numProbs * sizeof(QrOrder));
  if (!rc && !ord) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  for (uint64_t i = 0; i < numProbs && !rc; ++i) {
    uint32_t const need = qrcpLds(hostProbs[i].mt, hostProbs[i].me);
    if (!need) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "QR preconditioner: problem %llu does not fit LDS", (unsigned long long)i); break; }
    int c = 0;
    while (cap[c] < need) ++c;
    ord[i].cls = c; ord[i].idx = (uint32_t)i;
    ord[i].cost = (double)hostProbs[i].mt * hostProbs[i].me * (hostProbs[i].me + hostProbs[i].n);
    count[c + 1] += 1;
  }
  for (int c = 0; c < NC; ++c) count[c + 1] += count[c];
  if (!rc) {
    // by LDS class, the longest first within a class (dispatch order: the tail of a launch is made of cheap problems)
    qsort(ord, numProbs, sizeof(QrOrder), qrOrderCmp);
    for (uint64_t at = 0; at < numProbs; ++at) { sorted[at] = hostProbs[ord[at].idx]; order[ord[at].idx] = (uint32_t)at; }
  }
  free(ord);
  BfQrProb *dP = NULL;
  uint32_t *dR = NULL;
  if (!rc) rc = uploadArrayB(&dP, sorted, numProbs, "qr problems");
  if (!rc) rc = hipFailB(hipMalloc((void **)&dR, numProbs * sizeof(uint32_t)), "qr ranks");
  if (!rc) rc = hipFailB(hipFuncSetAttribute((void const *)bfQrcpKernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap[NC - 1]), "hipFuncSetAttribute(QR LDS)");
  for (int c = NC - 1; c >= 0 && !rc; --c) {              /* the big ones first */
    uint64_t const nc = count[c + 1] - count[c];
    if (!nc) continue;
    uint32_t const threads = c >= 2 ? 512 : 256;          /* LDS class ~ rows: short columns need few lane groups */
    hipLaunchKernelGGL(bfQrcpKernel, dim3((uint32_t)nc), dim3(threads), cap[c], 0, dP + count[c], dR + count[c]);
    rc = hipFailB(hipGetLastError(), "QR preconditioner launch");
  }
  if (!rc) rc = hipFailB(hipDeviceSynchronize(), "QR preconditioner");
  if (!rc) rc = hipFailB(hipMemcpy(ranks, dR, numProbs * sizeof(uint32_t), hipMemcpyDeviceToHost), "qr ranks");
  for (uint64_t i = 0; i < numProbs && !rc; ++i) hostRanks[i] = ranks[order[i]];
  (void)hipFree(dP); (void)hipFree(dR);
  free(order); free(sorted); free(ranks);
  return rc;
}

// Launch classes.  Workgroup: 256 threads for <= 64 columns, 1024 above.  LDS: the smallest of
// 16/32/64/144 KiB that holds the whole stacked matrix (several small problems then share a
// CU), else 144 KiB and block sweeps.  Lane-group width W: the largest power of two that still
// gives each of the b column pairs of an inner step its own group, at most the column length.
static uint32_t const kJacobiLds[4] = {16u << 10, 32u << 10, 64u << 10, BF_JACOBI_LDS_MAX};

static int jacobiClass(BfSvdProb const *p, int forceGlobal, int *wlog, int *big, int *ldsClass) {
  uint64_t const R = (uint64_t)p->mt + p->me, Rp = R | 1u;
  if (2 * Rp * 16 > BF_JACOBI_LDS_MAX || p->me > BF_JACOBI_MAX_COLS || forceGlobal) { *ldsClass = -1; *big = 1; *wlog = 4; return 0; }   /* global-memory fallback */
  uint64_t const meEven = p->me + (p->me & 1u);
  int lc = 3;
  for (int c = 0; c < 3; ++c)
    if (meEven * Rp * 16 <= kJacobiLds[c]) { lc = c; break; }
  uint32_t C = (uint32_t)((kJacobiLds[lc] / 16u) / Rp);
  C = C < 2 ? 2 : C & ~1u;
  uint32_t const b = C / 2 < (p->me + 1) / 2 ? C / 2 : (p->me + 1) / 2;
  *ldsClass = lc;
  *big = p->me > 64;
  uint32_t const threads = *big ? 1024 : 256;
  uint32_t w = 64;
  while (w > 4 && (threads / w < b || w / 2 >= p->mt)) w >>= 1;
  int l = 0;
  while ((4u << l) < w) ++l;
  *wlog = l;                                       // W = 4 << l, l = 0..4
  return 0;
}

template <int W> static int jacobiLaunch(uint32_t count, uint32_t threads, uint32_t lds, BfSvdProb const *dP, uint32_t const *dL, BfSvdStats *dS) {
  int rc = hipFailB(hipFuncSetAttribute((void const *)bfJacobiKernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BF_JACOBI_LDS_MAX),
                    "hipFuncSetAttribute(Jacobi LDS)");
  if (rc) return rc;
  hipLaunchKernelGGL(bfJacobiKernel<W>, dim3(count), dim3(threads), lds, 0, dP, dL, dS, lds);
  return hipFailB(hipGetLastError(), "Jacobi SVD launch");
}

typedef struct JacobiOrder { double cost; uint32_t idx; } JacobiOrder;
static int jacobiCostDescending(void const *pa, void const *pb) {
  JacobiOrder const *a = (JacobiOrder const *)pa, *b = (JacobiOrder const *)pb;
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  return a->idx < b->idx ? -1 : 1;
}

int bfdevBuildJacobi(BfSvdProb const *hostProbs, uint64_t numProbs, BfSvdStats *stats) {
  if (!numProbs) return 0;
  enum { NW = 5, NL = 4, NCLS = 2 * NW * NL };
  uint32_t *lists[NCLS] = {0};
  uint64_t counts[NCLS] = {0};
  uint8_t *cls = (uint8_t *)malloc(numProbs);
  int rc = cls ? 0 : bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  // BFHIP_JACOBI_GLOBAL=1 sends every problem through the fallback kernel (test hook for a path that
  // otherwise needs problems of > 2300 columns)
  char const *env = getenv("BFHIP_JACOBI_GLOBAL");
  int const forceGlobal = env && env[0] == '1';
  uint32_t *globalList = (uint32_t *)malloc(numProbs * sizeof(uint32_t));
  uint64_t numGlobal = 0;
  if (!globalList) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  // Problems of rows + columns >= 512 (fewer than nine stacked columns of a block would fit the LDS tile) run the block form
  // on Gram matrices (bfJacobiGramKernel), as do those too long for the tile at all when they have <= 4096 columns.
  // (N = 262144 build with the limit at 384 / 512 / 768 / 1024: 21.5 / 22.0 / 22.0 - 23.6 / 27.5 s.)
  // BFHIP_JACOBI_GRAM_MIN moves the limit (0: every problem -- a test hook; a huge value: none).
  char const *genv = getenv("BFHIP_JACOBI_GRAM_MIN");
  uint64_t const gramMin = genv && genv[0] ? strtoull(genv, NULL, 10) : 512;
  JacobiOrder *gramList = (JacobiOrder *)malloc((numProbs ? numProbs : 1) * sizeof(JacobiOrder));
  uint64_t numGram = 0;
  if (!gramList) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  for (uint64_t i = 0; i < numProbs && !rc; ++i) {
    int wlog, big, lc;
    rc = jacobiClass(&hostProbs[i], forceGlobal, &wlog, &big, &lc);
    if (!forceGlobal && hostProbs[i].me <= BF_GRAM_MAX_COLS && (lc < 0 || (uint64_t)hostProbs[i].mt + hostProbs[i].me >= gramMin)) {
      cls[i] = 0xff;
      gramList[numGram].idx = (uint32_t)i;
      gramList[numGram++].cost = (double)hostProbs[i].me * hostProbs[i].me * (hostProbs[i].mt + hostProbs[i].me);
      continue;
    }
    if (lc < 0) { cls[i] = 0xff; globalList[numGlobal++] = (uint32_t)i; continue; }
    cls[i] = (uint8_t)((wlog * 2 + big) * NL + lc);
    counts[cls[i]] += 1;
  }
  for (int c = 0; c < NCLS && !rc; ++c) {
    if (!counts[c]) continue;
    lists[c] = (uint32_t *)malloc(counts[c] * sizeof(uint32_t));
    if (!lists[c]) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    counts[c] = 0;
  }
  for (uint64_t i = 0; i < numProbs && !rc; ++i)
    if (cls[i] != 0xff) lists[cls[i]][counts[cls[i]]++] = (uint32_t)i;
  // longest first within a class (workgroups are dispatched in list order): the tail of a launch is then made of
  // the cheap problems, not of one 890-column problem started last
  for (int c = 0; c < NCLS && !rc; ++c) {
    if (counts[c] < 2) continue;
    JacobiOrder *ord = (JacobiOrder *)malloc(counts[c] * sizeof(JacobiOrder));
    if (!ord) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); break; }
    for (uint64_t i = 0; i < counts[c]; ++i) {
      BfSvdProb const *q = &hostProbs[lists[c][i]];
      ord[i].cost = (double)q->me * q->me * (q->mt + q->me); ord[i].idx = lists[c][i];
    }
    qsort(ord, counts[c], sizeof(JacobiOrder), jacobiCostDescending);
    for (uint64_t i = 0; i < counts[c]; ++i) lists[c][i] = ord[i].idx;
    free(ord);
  }
  BfSvdProb *dP = NULL;
  BfSvdStats *dS = NULL;
  if (!rc) rc = uploadArrayB(&dP, hostProbs, numProbs, "svd problems");
  BfSvdStats zero = {0, 0, 0, 0};
  if (!rc) rc = uploadArrayB(&dS, &zero, 1, "svd stats");
  uint32_t *dL[NCLS] = {0};
  // BFHIP_JACOBI_PROFILE=1: time every class launch on its own (a synchronisation per class) and print it
  char const *penv = getenv("BFHIP_JACOBI_PROFILE");
  int const profile = penv && penv[0] == '1';
  // the block-form problems go first, on a stream of their own: they are the longest, and the class launches below (null
  // stream; a non-blocking stream does not order with it) fill the CUs their tail leaves idle
  uint32_t *dGram = NULL;
  hipStream_t sGram = NULL;
  if (!rc && numGram) rc = hipFailB(hipStreamCreateWithFlags(&sGram, hipStreamNonBlocking), "hipStreamCreate");
  if (!rc && numGram) {
    qsort(gramList, numGram, sizeof(JacobiOrder), jacobiCostDescending);
    uint32_t *ids = (uint32_t *)malloc(numGram * sizeof(uint32_t));
    if (!ids) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    for (uint64_t i = 0; i < numGram && !rc; ++i) ids[i] = gramList[i].idx;
    if (!rc) rc = uploadArrayB(&dGram, ids, numGram, "svd block-form list");
    free(ids);
    struct timespec t0, t1;
    BfSvdStats before = {0, 0, 0, 0};
    if (profile && !rc) { (void)hipDeviceSynchronize(); (void)hipMemcpy(&before, dS, sizeof before, hipMemcpyDeviceToHost); clock_gettime(CLOCK_MONOTONIC, &t0); }
    if (!rc) {
      hipLaunchKernelGGL(bfJacobiGramKernel, dim3((uint32_t)numGram), dim3(BF_GRAM_THREADS), 0, sGram, dP, dGram, dS);
      rc = hipFailB(hipGetLastError(), "Jacobi SVD (block form) launch");
    }
    if (profile && !rc) {
      (void)hipDeviceSynchronize();
      clock_gettime(CLOCK_MONOTONIC, &t1);
      BfSvdStats after;
      (void)hipMemcpy(&after, dS, sizeof after, hipMemcpyDeviceToHost);
      fprintf(stderr, "[jacobi] block form (Gram) problems=%llu sweeps=%.1f  %.3f s\n", (unsigned long long)numGram,
              (double)(after.sumSweeps - before.sumSweeps) / (double)numGram, (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    }
  }
  // classes are launched back to back (largest tiles first: they run longest) and synchronised once
  for (int lc = NL - 1; lc >= 0 && !rc; --lc)
    for (int wb = 0; wb < 2 * NW && !rc; ++wb) {
      int const c = wb * NL + lc;
      if (!counts[c]) continue;
      rc = uploadArrayB(&dL[c], lists[c], counts[c], "svd class list");
      if (rc) break;
      uint32_t const threads = wb & 1 ? 1024 : 256, lds = kJacobiLds[lc], n = (uint32_t)counts[c];
      struct timespec t0, t1;
      BfSvdStats before = {0, 0, 0, 0};
      if (profile) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&before, dS, sizeof before, hipMemcpyDeviceToHost);
        clock_gettime(CLOCK_MONOTONIC, &t0);
      }
      switch (wb >> 1) {
        case 0: rc = jacobiLaunch<4>(n, threads, lds, dP, dL[c], dS); break;
        case 1: rc = jacobiLaunch<8>(n, threads, lds, dP, dL[c], dS); break;
        case 2: rc = jacobiLaunch<16>(n, threads, lds, dP, dL[c], dS); break;
        case 3: rc = jacobiLaunch<32>(n, threads, lds, dP, dL[c], dS); break;
        default: rc = jacobiLaunch<64>(n, threads, lds, dP, dL[c], dS); break;
      }
      if (profile && !rc) {
        (void)hipDeviceSynchronize();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        BfSvdStats after;
        (void)hipMemcpy(&after, dS, sizeof after, hipMemcpyDeviceToHost);
        uint32_t lo = 0xffffffffu, hi = 0;
        double cube = 0;
        for (uint64_t i = 0; i < counts[c]; ++i) {
          BfSvdProb const *q = &hostProbs[lists[c][i]];
          lo = q->me < lo ? q->me : lo; hi = q->me > hi ? q->me : hi;
          cube += (double)q->me * q->me * (q->mt + q->me);
        }
        fprintf(stderr, "[jacobi] W=%d threads=%u lds=%uK problems=%u me=%u..%u sum(me^2 (mt+me))=%.3g sweeps=%.1f  %.3f s\n", 4 << (wb >> 1), threads,
                lds >> 10, n, lo, hi, cube, (double)(after.sumSweeps - before.sumSweeps) / n,
                (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
      }
    }
  uint32_t *dG = NULL;
  if (!rc && numGlobal) {
    rc = uploadArrayB(&dG, globalList, numGlobal, "svd fallback list");
    if (!rc) {
      hipLaunchKernelGGL(bfJacobiGlobalKernel, dim3((uint32_t)numGlobal), dim3(1024), 0, 0, dP, dG, dS);
      rc = hipFailB(hipGetLastError(), "Jacobi SVD (global-memory fallback) launch");
    }
  }
  if (!rc) rc = hipFailB(hipDeviceSynchronize(), "Jacobi SVD");
  (void)hipFree(dG); (void)hipFree(dGram);
  if (sGram) (void)hipStreamDestroy(sGram);
  free(globalList); free(gramList);
  for (int c = 0; c < NCLS; ++c) (void)hipFree(dL[c]);
  if (!rc && stats) {
    BfSvdStats got;
    rc = hipFailB(hipMemcpy(&got, dS, sizeof got, hipMemcpyDeviceToHost), "svd stats");
    if (!rc) {
      if (got.maxSweeps > stats->maxSweeps) stats->maxSweeps = got.maxSweeps;
      stats->notConverged += got.notConverged;
      stats->truncated += got.truncated;
      stats->sumSweeps += got.sumSweeps;
    }
  }
  (void)hipFree(dP);
  (void)hipFree(dS);
  for (int c = 0; c < NCLS; ++c) free(lists[c]);
  free(cls);
  return rc;
}

// ---------------------------------------------------------------------------
// batched small complex GEMM, 32 x 32 output tile per workgroup, K in steps of
// 16 through LDS, 2 x 2 outputs per thread (fixed k order: deterministic)
// ---------------------------------------------------------------------------
#define BF_GT 32
#define BF_GK 16

__global__ __launch_bounds__(256) void bfGemmKernel(BfGemmJob const *jobs, uint64_t const *prefix, uint32_t numJobs, uint64_t tileBase) {
  __shared__ double2 As[BF_GK][BF_GT + 1];
  __shared__ double2 Bs[BF_GK][BF_GT + 1];
  uint64_t const tile = tileBase + blockIdx.x;
  uint32_t lo = 0, hi = numJobs;
  while (hi - lo > 1) {
    uint32_t const mid = (lo + hi) >> 1;
    if (prefix[mid] <= tile) lo = mid; else hi = mid;
  }
  BfGemmJob const J = jobs[lo];
  uint32_t const tilesM = (J.M + BF_GT - 1) / BF_GT;
  uint32_t const tl = (uint32_t)(tile - prefix[lo]);
  uint32_t const i0 = (tl % tilesM) * BF_GT, j0 = (tl / tilesM) * BF_GT;
  double2 const *A = (double2 const *)J.a, *B = (double2 const *)J.b;
  uint32_t const tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  double cr[2][2] = {{0, 0}, {0, 0}}, ci[2][2] = {{0, 0}, {0, 0}};
  for (uint32_t k0 = 0; k0 < J.K; k0 += BF_GK) {
    // stage op(A)[i0.., k0..] and B[k0.., j0..]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double2 a = make_double2(0.0, 0.0);
      if (J.transA) {
        uint32_t const kk = tid & 15, ii = (tid >> 4) + 16 * h;
        if (k0 + kk < J.K && i0 + ii < J.M) { a = A[(uint64_t)(i0 + ii) * J.lda + k0 + kk]; a.y = -a.y; }
        As[kk][ii] = a;
      } else {
        uint32_t const ii = tid & 31, kk = (tid >> 5) + 8 * h;
        if (k0 + kk < J.K && i0 + ii < J.M) a = A[(uint64_t)(k0 + kk) * J.lda + i0 + ii];
        As[kk][ii] = a;
      }
      uint32_t const kb = tid & 15, jj = (tid >> 4) + 16 * h;
      double2 b = make_double2(0.0, 0.0);
      if (k0 + kb < J.K && j0 + jj < J.N) b = B[(uint64_t)(j0 + jj) * J.ldb + k0 + kb];
      Bs[kb][jj] = b;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BF_GK; ++kk) {
      double2 const a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
      double2 const av[2] = {a0, a1}, bv[2] = {b0, b1};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          cr[u][v] = fma(av[u].x, bv[v].x, cr[u][v]); cr[u][v] = fma(-av[u].y, bv[v].y, cr[u][v]);
          ci[u][v] = fma(av[u].x, bv[v].y, ci[u][v]); ci[u][v] = fma(av[u].y, bv[v].x, ci[u][v]);
        }
    }
    __syncthreads();
  }
  double2 *C = (double2 *)J.c;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    uint32_t const i = i0 + ty + 16 * u;
    if (i >= J.M) continue;
    double const sc = J.scale ? J.scale[i] : 1.0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      uint32_t const j = j0 + tx + 16 * v;
      if (j < J.N) C[(uint64_t)j * J.ldc + i] = make_double2(cr[u][v] * sc, ci[u][v] * sc);
    }
  }
}

int bfdevBuildGemm(BfGemmJob const *hostJobs, uint64_t numJobs) {
  if (!numJobs) return 0;
  uint64_t *prefix = (uint64_t *)malloc((numJobs + 1) * sizeof(uint64_t));
  if (!prefix) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  prefix[0] = 0;
  for (uint64_t i = 0; i < numJobs; ++i)
    prefix[i + 1] = prefix[i] + (uint64_t)((hostJobs[i].M + BF_GT - 1) / BF_GT) * ((hostJobs[i].N + BF_GT - 1) / BF_GT);
  BfGemmJob *dJ = NULL;
  uint64_t *dP = NULL;
  int rc = uploadArrayB(&dJ, hostJobs, numJobs, "gemm jobs");
  if (!rc) rc = uploadArrayB(&dP, prefix, numJobs + 1, "gemm tile prefix");
  uint64_t const tiles = prefix[numJobs];
  for (uint64_t done = 0; done < tiles && !rc;) {
    uint32_t const n = (uint32_t)(tiles - done > (1u << 30) ? (1u << 30) : tiles - done);
    hipLaunchKernelGGL(bfGemmKernel, dim3(n), dim3(256), 0, 0, dJ, dP, (uint32_t)numJobs, done);
    rc = hipFailB(hipGetLastError(), "builder gemm launch");
    done += n;
  }
  if (!rc) rc = hipFailB(hipDeviceSynchronize(), "builder gemm");
  (void)hipFree(dJ);
  (void)hipFree(dP);
  free(prefix);
  return rc;
}

// ---------------------------------------------------------------------------
// leaf store (column-major leaves of one batch) -> packed arena pieces
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bfPackKernel(double2 *arena, double2 const *store, BfPackPiece const *pieces, uint64_t base) {
  BfPackPiece const pc = pieces[base + blockIdx.x];
  uint32_t const total = pc.mrPad * pc.ncols;
  double2 *dst = arena + pc.dataOff;
  double2 const *src = store + pc.srcOff;
  for (uint32_t e = threadIdx.x; e < total; e += 256) {
    uint32_t const r = e % pc.mrPad, c = e / pc.mrPad;
    dst[e] = r < pc.mr ? src[(uint64_t)c * pc.srcLd + r] : make_double2(0.0, 0.0);
  }
}

int bfdevBuildPack(void *arena, void const *store, BfPackPiece const *hostPieces, uint64_t count) {
  if (!count) return 0;
  BfPackPiece *d = NULL;
  int rc = uploadArrayB(&d, hostPieces, count, "pack pieces");
  for (uint64_t done = 0; done < count && !rc;) {
    uint32_t const n = (uint32_t)(count - done > (1u << 30) ? (1u << 30) : count - done);
    hipLaunchKernelGGL(bfPackKernel, dim3(n), dim3(256), 0, 0, (double2 *)arena, (double2 const *)store, d, done);
    rc = hipFailB(hipGetLastError(), "pack launch");
    done += n;
  }
  if (!rc) rc = hipFailB(hipDeviceSynchronize(), "pack");
  (void)hipFree(d);
  return rc;
}

// ---------------------------------------------------------------------------
// matrix-free dense apply: y_i = sum_j G(p_i, p_j) x_j.  A workgroup owns 256
// targets and one slice of the sources (staged through LDS, 256 at a time);
// slices are summed afterwards in fixed order.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bfHelm2DenseKernel(EvalEnvDev const E, uint32_t pot, uint64_t n, uint64_t m, bool square,
                                                          double2 const *x, double2 *partial, uint64_t sliceLen) {
  __shared__ double sx[256], sy[256], sw[256];
  __shared__ double2 xv[256];
  __shared__ uint64_t so[256];
  __shared__ double snx[256], sny[256];
  uint64_t const i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  uint64_t const j0 = (uint64_t)blockIdx.y * sliceLen;
  uint64_t const j1 = j0 + sliceLen < n ? j0 + sliceLen : n;
  double tx = 0, ty = 0, nx = 0, ny = 0;
  uint64_t oi = 0;
  double const *tp = square ? E.pts : E.tpts, *tn = square ? E.normals : E.tnormals;
  bool const kr = square && E.krOrder;
  if (i < m) {
    tx = tp[2 * i]; ty = tp[2 * i + 1];
    if (pot == 1) { nx = tn[2 * i]; ny = tn[2 * i + 1]; }
    if (kr) oi = E.orig[i];
  }
  double ar = 0, ai = 0;
  for (uint64_t jb = j0; jb < j1; jb += 256) {
    uint64_t const j = jb + threadIdx.x;
    __syncthreads();
    if (j < j1) {
      sx[threadIdx.x] = E.pts[2 * j]; sy[threadIdx.x] = E.pts[2 * j + 1]; xv[threadIdx.x] = x[j];
      sw[threadIdx.x] = E.colWeights ? E.colWeights[j] : 1.0;
      so[threadIdx.x] = kr ? E.orig[j] : 0;
      snx[threadIdx.x] = pot >= 2 ? E.normals[2 * j] : 0.0;
      sny[threadIdx.x] = pot >= 2 ? E.normals[2 * j + 1] : 0.0;
    }
    __syncthreads();
    uint32_t const cnt = (uint32_t)(j1 - jb < 256 ? j1 - jb : 256);
    if (i < m)
      for (uint32_t t = 0; t < cnt; ++t) {
        double2 g;
        if (square && jb + t == i) g = make_double2(E.selfRe, E.selfIm);
        else {
          g = bfKernelValue(E, pot, tx - sx[t], ty - sy[t], snx[t], sny[t], nx, ny);
          bool hit;
          double const f = sw[t] * (kr ? bfKrFactor(E, oi, so[t], hit) : 1.0);
          g.x *= f; g.y *= f;
        }
        double2 const v = xv[t];
        ar = fma(g.x, v.x, ar); ar = fma(-g.y, v.y, ar);
        ai = fma(g.x, v.y, ai); ai = fma(g.y, v.x, ai);
      }
  }
  if (i < m) partial[(uint64_t)blockIdx.y * m + i] = make_double2(ar, ai);
}

__global__ __launch_bounds__(256) void bfSliceSumKernel(double2 const *partial, uint64_t n, uint32_t slices, double2 *y) {
  uint64_t const i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double sr = 0, si = 0;
  for (uint32_t s = 0; s < slices; ++s) { double2 const v = partial[(uint64_t)s * n + i]; sr += v.x; si += v.y; }
  y[i] = make_double2(sr, si);
}

int bfdevMemFree(uint64_t *freeBytes) {
  size_t f = 0, t = 0;
  int rc = hipFailB(hipMemGetInfo(&f, &t), "hipMemGetInfo");
  *freeBytes = f;
  return rc;
}

int bfdevHelm2Dense(BfEvalEnv const *env, uint32_t pot, uint64_t n, uint64_t numTgt, void const *dX, void *dY, void *stream) {
  if (!n) return 0;
  hipStream_t const s = (hipStream_t)stream;
  bool const square = numTgt == 0;
  uint64_t const m = square ? n : numTgt;
  uint32_t const tb = (uint32_t)((m + 255) / 256);
  // enough workgroups to fill 256 CUs several times over, slices of >= 256 sources
  uint32_t slices = tb >= 4096 ? 1 : (4096 + tb - 1) / tb;
  if ((uint64_t)slices * 256 > n) slices = (uint32_t)((n + 255) / 256);
  uint64_t const sliceLen = ((n + slices - 1) / slices + 255) / 256 * 256;
  slices = (uint32_t)((n + sliceLen - 1) / sliceLen);
  double2 *partial = NULL;
  int rc = hipFailB(hipMalloc((void **)&partial, (size_t)slices * m * sizeof(double2)), "hipMalloc(dense apply partials)");
  if (rc) return rc;
  hipLaunchKernelGGL(bfHelm2DenseKernel, dim3(tb, slices), dim3(256), 0, s, toDev(env), pot, n, m, square, (double2 const *)dX, partial, sliceLen);
  rc = hipFailB(hipGetLastError(), "dense apply launch");
  if (!rc) {
    hipLaunchKernelGGL(bfSliceSumKernel, dim3(tb), dim3(256), 0, s, partial, m, slices, (double2 *)dY);
    rc = hipFailB(hipGetLastError(), "dense apply sum launch");
  }
  if (!rc) rc = hipFailB(hipStreamSynchronize(s), "dense apply");
  (void)hipFree(partial);
  return rc;
}
