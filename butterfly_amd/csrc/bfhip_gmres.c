/* bfhip_gmres.c -- device-resident GMRES around the butterfly apply (SURVEY.md
 * section 8(f) row 1).  Mirrors the reference's bfSolveGMRES
 * (src/linalg.c:47-317) step for step -- unrestarted, modified Gram-Schmidt,
 * one Givens rotation per column and right-hand side, convergence on
 * max_p |s_{j+1,p}| / max_p ||r_p|| -- but keeps x0, b, the Krylov basis V and
 * the work vector on the GPU: per iteration the host receives only the new
 * Hessenberg column ((j+2) * nrhs complex numbers, into pinned memory).
 *
 * Orthogonalisation: the reference runs modified Gram-Schmidt, one basis vector at
 * a time with a heap allocation per BLAS-1 call (`bfMatCopy`, `bfMatScaleCols`,
 * `bfMatSubInplace`, src/linalg.c:174-184): j + 1 dependent passes per iteration.  On a
 * GPU that is j + 1 launches of a few microseconds of work each; by default this solver
 * runs classical Gram-Schmidt twice (CGS2) instead -- all j + 1 dots in one launch, one
 * reduction launch, one projection launch, then the same again -- 7 launches per iteration
 * whatever j is.  CGS2 is as stable as MGS; H[:, j] is the sum of the two passes'
 * coefficients, and iteration counts stay within +-1 of the reference's order
 * (tests/test_gmres.py).  BFHIP_GMRES_MGS=1 in the environment selects the reference's
 * own order (one fused kernel per basis vector), which follows oracle/linalg_ref.py
 * iteration for iteration.
 *
 * Reference quirk kept for parity: when the residual test passes at iteration
 * j the loop breaks before j is incremented, so the solution uses j (not j+1)
 * basis vectors and numIter reports j (src/linalg.c:235-243,245-287).
 */
#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"

#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

/* bfVecComplexGetGivensRotation, src/vec_complex.c:275-286 */
static void givens(cplx a, cplx b, cplx *c, cplx *s) {
  if (cabs(b) == 0) { *c = 1; *s = 0; }
  else if (cabs(b) > cabs(a)) { cplx t = -a / b; *s = 1 / sqrt(1 + pow(cabs(t), 2)); *c = t * *s; }
  else { cplx t = -b / a; *c = 1 / sqrt(1 + pow(cabs(t), 2)); *s = t * *c; }
}
/* mulInplace_givensComplex, src/vec_complex.c:155-168 */
static void applyGivens(cplx *z0p, cplx *z1p, cplx c, cplx s) {
  cplx z0 = *z0p, z1 = *z1p;
  *z0p = conj(c) * z0 + -s * z1;
  *z1p = s * z0 + c * z1;
}

static int solveDevice(BfGmresApplyFn apply, void *ctx, uint64_t n, BfhipOperator *precond, int orth, void const *dB, size_t nrhs, void const *dX0, double tol,
                       size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream);

/* the Krylov basis and every staging buffer live on the OPERATOR's device, whatever device is
 * current in the caller; the caller's device is restored on every path */
int bfhipSolveGMRESDevice(BfhipOperator *op, void const *dB, size_t nrhs, void const *dX0, double tol,
                          size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream) {
  return bfhipSolveGMRESPrecondDevice(op, NULL, dB, nrhs, dX0, tol, maxNumIter, numIter, residual, dX, stream);
}

/* Left-preconditioned GMRES (the reference's M argument, src/linalg.c:90-97,131,159): `solveM` applies what
 * bfMatSolve(M, .) computes -- the action of M^{-1} -- as a device operator of its own (an approximate
 * inverse: block-diagonal, a coarser butterfly, ...).  The residual is then the PRECONDITIONED residual, as in
 * the reference ("the residual will be determined from the preconditioned residual vectors", :41-43). */
int bfhipSolveGMRESPrecondDevice(BfhipOperator *op, BfhipOperator *solveM, void const *dB, size_t nrhs, void const *dX0, double tol,
                                 size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream) {
  BfhipGmresOptions o;
  memset(&o, 0, sizeof o);
  o.structSize = sizeof o; o.orthogonalization = BFHIP_GMRES_ORTH_DEFAULT; o.tol = tol; o.maxNumIter = maxNumIter; o.solveM = solveM;
  return bfhipSolveGMRESOptsDevice(op, &o, dB, nrhs, dX0, numIter, residual, dX, stream);
}

/* The solver with its choices spelled out per call (BfhipGmresOptions, include/bfhip.h): orthogonalisation order,
 * tolerance, iteration cap, left preconditioner. */
static int applyOperator(void *ctx, void const *dX, size_t nrhs, void *dY, void *stream) { return bfhipApplyDevice(ctx, dX, nrhs, dY, stream); }

int bfhipSolveGMRESOptsDevice(BfhipOperator *op, BfhipGmresOptions const *opt, void const *dB, size_t nrhs, void const *dX0,
                              size_t *numIter, double *residual, void *dX, void *stream) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  BfhipStats st;
  memset(&st, 0, sizeof st);
  st.structSize = sizeof st;
  int rc = bfhipGetStats(op, &st);
  if (rc) return rc;
  if (st.dtype != BFHIP_C128) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "GMRES is implemented for complex operators");
  if (st.numRows != st.numCols) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "GMRES needs a square operator (linalg.c:85-87)");
  int const dev = bfhipOperatorDevice(op);
  if (dev < 0) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator has no device (compiled with BFHIP_FLAG_PLAN_ONLY)");
  return bfGmresSolve(applyOperator, op, st.numRows, dev, opt, dB, nrhs, dX0, numIter, residual, dX, stream);
}

/* The solver around any device matvec of order n on device `device` (the sharded step of bfhip_shard.hip is the other caller). */
int bfGmresSolve(BfGmresApplyFn apply, void *ctx, uint64_t n, int device, BfhipGmresOptions const *opt, void const *dB, size_t nrhs,
                 void const *dX0, size_t *numIter, double *residual, void *dX, void *stream) {
  if (!apply || !opt || opt->structSize < sizeof(BfhipGmresOptions)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument / BfhipGmresOptions.structSize too small");
  BfhipOperator *solveM = opt->solveM;
  double const tol = opt->tol;
  size_t const maxNumIter = opt->maxNumIter;
  if (opt->orthogonalization > BFHIP_GMRES_ORTH_MGS) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "unknown orthogonalization %u", opt->orthogonalization);
  int orth = (int)opt->orthogonalization;
  if (orth == BFHIP_GMRES_ORTH_DEFAULT) {       /* the environment decides only where the caller did not */
    char const *mgsEnv = getenv("BFHIP_GMRES_MGS");
    orth = mgsEnv && mgsEnv[0] == '1' ? BFHIP_GMRES_ORTH_MGS : BFHIP_GMRES_ORTH_CGS2;
  }
  if (solveM && (bfhipOperatorDevice(solveM) != device || bfhipGetNumRows(solveM) != n || bfhipGetNumCols(solveM) != n))
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the preconditioner must be an n x n operator on the operator's device (linalg.c:92-97)");
  if (solveM) {
    /* the Krylov vectors are complex128: a real preconditioner would be applied to them as if they were real n-vectors */
    BfhipStats ms;
    memset(&ms, 0, sizeof ms);
    ms.structSize = sizeof ms;
    int rcs = bfhipGetStats(solveM, &ms);
    if (rcs) return rcs;
    if (ms.dtype != BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "the preconditioner must be a complex128 operator like the system it preconditions");
  }
  int prev = -1, dev = device;
  bfdevGetDevice(&prev);
  int rc = prev != dev ? bfdevSetDevice(dev) : 0;
  if (rc) return rc;
  rc = solveDevice(apply, ctx, n, solveM, orth, dB, nrhs, dX0, tol, maxNumIter, numIter, residual, dX, stream);
  if (prev >= 0 && prev != dev) bfdevSetDevice(prev);
  return rc;
}

static int solveDevice(BfGmresApplyFn apply, void *ctx, uint64_t n, BfhipOperator *precond, int orth, void const *dB, size_t nrhs, void const *dX0, double tol,
                       size_t maxNumIter, size_t *numIter, double *residual, void *dX, void *stream) {
  if (!dB || !dX) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (maxNumIter == 0) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "maxNumIter must be positive (linalg.c:81-82)");
  if (nrhs == 0 || nrhs > 0xffffu) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "nrhs out of range");
  int rc = 0;
  size_t const m = maxNumIter;
  size_t const vecBytes = (size_t)n * nrhs * 16;
  uint32_t nb = (uint32_t)((n + 255) / 256);          /* row blocks = partial sums per RHS and dot */
  if (nb > 1024) nb = 1024;
  if (nb == 0) nb = 1;
  int const useMgs = orth == BFHIP_GMRES_ORTH_MGS;

  void *dV = NULL, *dW = NULL, *dPartA = NULL, *dPartB = NULL, *dH = NULL, *dY = NULL, *dAX0 = NULL;
  void *dPartAll = NULL, *dH1 = NULL, *dH2 = NULL;     /* CGS2: partials of all dots, coefficients of the two passes */
  void *dPre = NULL;                                   /* left preconditioner: the vector it is applied to */
  void *hHpinned = NULL;
  void *evCol[2] = {NULL, NULL};      /* "column j is in host memory", two in flight */
  cplx *hHslot[2] = {NULL, NULL};
  cplx *hH = NULL, *H = NULL, *S = NULL, *Jc = NULL, *Js = NULL, *y = NULL;
  double *rnorm = NULL;
  size_t j = 0;
  int converged = 0;
  double lastResidual = INFINITY;

#define CHECK(expr) do { rc = (expr); if (rc) goto done; } while (0)
  CHECK(bfdevMalloc(&dV, (m + 1) * vecBytes));
  CHECK(bfdevMalloc(&dW, vecBytes));
  CHECK(bfdevMalloc(&dPartA, (size_t)nb * nrhs * 16));
  CHECK(bfdevMalloc(&dPartB, (size_t)nb * nrhs * 16));
  CHECK(bfdevMalloc(&dH, (m + 2) * nrhs * 16));
  CHECK(bfdevMalloc(&dY, (m + 1) * nrhs * 16));
  if (!useMgs) {
    CHECK(bfdevMalloc(&dPartAll, (size_t)nb * nrhs * (m + 1) * 16));
    CHECK(bfdevMalloc(&dH1, (m + 1) * nrhs * 16));
    CHECK(bfdevMalloc(&dH2, (m + 1) * nrhs * 16));
  }
  CHECK(bfdevHostAllocPinned(&hHpinned, 2 * (m + 2) * nrhs * sizeof(cplx)));
  hH = hHpinned;
  hHslot[0] = hH; hHslot[1] = hH + (m + 2) * nrhs;
  CHECK(bfdevEventCreate(&evCol[0]));
  CHECK(bfdevEventCreate(&evCol[1]));
  H = calloc((m + 2) * m * nrhs, sizeof(cplx));       /* H[(j*(m+2) + i)*nrhs + p] */
  S = calloc((m + 1) * nrhs, sizeof(cplx));
  Jc = malloc(m * nrhs * sizeof(cplx));
  Js = malloc(m * nrhs * sizeof(cplx));
  y = malloc((m + 1) * nrhs * sizeof(cplx));
  rnorm = malloc(nrhs * sizeof(double));
  if (!H || !S || !Jc || !Js || !y || !rnorm) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }

  /* R = B - A X0 (linalg.c:127-131); X0 == NULL means zeros (:120-123) */
  if (dX0) {
    CHECK(bfdevMalloc(&dAX0, vecBytes));
    CHECK(apply(ctx, dX0, nrhs, dAX0, stream));
  }
  if (precond) {
    /* R = M^{-1} (B - A X0) (:127-135): the difference goes to a scratch vector, the preconditioner writes W,
     * and the norm partials are taken from W */
    CHECK(bfdevMalloc(&dPre, vecBytes));
    CHECK(bfdevGmresResidual(dB, dAX0, dPre, dPartA, n, (uint32_t)nrhs, nb, stream));
    CHECK(bfhipApplyDevice(precond, dPre, nrhs, dW, stream));
    CHECK(bfdevGmresResidual(dW, NULL, dW, dPartA, n, (uint32_t)nrhs, nb, stream));
  } else
  CHECK(bfdevGmresResidual(dB, dAX0, dW, dPartA, n, (uint32_t)nrhs, nb, stream));
  /* V[0] = R / ||R|| per column; S[0] = ||R|| (:139-151) */
  CHECK(bfdevGmresFinish(dW, dPartA, dV, dH, n, (uint32_t)nrhs, nb, stream));
  CHECK(bfdevMemcpyD2HAsync(hH, dH, nrhs * 16, stream));
  CHECK(bfdevSync(stream));
  double beta = 0;
  for (size_t p = 0; p < nrhs; ++p) { rnorm[p] = creal(hH[p]); S[p] = rnorm[p]; if (rnorm[p] > beta) beta = rnorm[p]; }
  if (!(beta > 0)) {
    /* zero residual: x0 already solves the system; the reference would divide by zero here */
    if (dX0) CHECK(bfdevMemcpyD2DAsync(dX, dX0, vecBytes, stream));
    else CHECK(bfdevMemset(dX, 0, vecBytes));
    j = 0;
    lastResidual = 0;
    goto finish;
  }

  /* One iteration's device work is enqueued by enqueueIteration(); the host needs its Hessenberg column only
   * for the Givens rotations and the convergence test, so iteration jj + 1 is enqueued BEFORE the host waits
   * for column jj (V[jj+1] is already on the device): the GPU never idles on the host round trip.  At most
   * one speculative iteration is thrown away when the test passes. */
#define ENQUEUE(J) do { \
    size_t const j_ = (J); \
    char *Vj = (char *)dV + j_ * vecBytes; \
    if (precond) { \
      CHECK(apply(ctx, Vj, nrhs, dPre, stream));                   /* W = M^{-1} (A V[j])  (:155-163) */ \
      CHECK(bfhipApplyDevice(precond, dPre, nrhs, dW, stream)); \
    } else \
    CHECK(apply(ctx, Vj, nrhs, dW, stream));                       /* W = A V[j]  (:157) */ \
    void *pin = dPartA, *pout = dPartB; \
    if (useMgs) { \
      /* modified Gram-Schmidt (:174-184): dot with V_0, then for each i subtract and start the next dot */ \
      CHECK(bfdevGmresDot(dV, dW, dPartA, n, (uint32_t)nrhs, nb, stream)); \
      for (size_t i = 0; i <= j_; ++i) { \
        char *Vi = (char *)dV + i * vecBytes; \
        char *Vn = i < j_ ? (char *)dV + (i + 1) * vecBytes : NULL; \
        CHECK(bfdevGmresMgsStep(Vi, Vn, dW, pin, pout, (char *)dH + i * nrhs * 16, n, (uint32_t)nrhs, nb, stream)); \
        void *t = pin; pin = pout; pout = t; \
      } \
    } else { \
      /* CGS2: h1 = V^H W, W -= V h1; h2 = V^H W, W -= V h2 (its |W|^2 partials feed the normalisation); H[:, j] = h1 + h2 */ \
      uint32_t const nv = (uint32_t)(j_ + 1); \
      CHECK(bfdevGmresDots(dV, dW, dPartAll, n, (uint32_t)nrhs, nb, nv, stream)); \
      CHECK(bfdevGmresDotsFinish(dPartAll, NULL, dH1, NULL, (uint32_t)nrhs, nb, nv, stream)); \
      CHECK(bfdevGmresProject(dV, dW, dH1, NULL, n, (uint32_t)nrhs, nb, nv, stream)); \
      CHECK(bfdevGmresDots(dV, dW, dPartAll, n, (uint32_t)nrhs, nb, nv, stream)); \
      CHECK(bfdevGmresDotsFinish(dPartAll, dH1, dH2, dH, (uint32_t)nrhs, nb, nv, stream)); \
      CHECK(bfdevGmresProject(dV, dW, dH2, dPartA, n, (uint32_t)nrhs, nb, nv, stream)); \
    } \
    /* H[j][j+1] = ||W||, V[j+1] = W / ||W||  (:186-198) */ \
    CHECK(bfdevGmresFinish(dW, pin, (char *)dV + (j_ + 1) * vecBytes, (char *)dH + (j_ + 1) * nrhs * 16, n, (uint32_t)nrhs, nb, stream)); \
    CHECK(bfdevMemcpyD2HAsync(hHslot[j_ & 1], dH, (j_ + 2) * nrhs * 16, stream)); \
    CHECK(bfdevEventRecord(evCol[j_ & 1], stream)); \
  } while (0)

  ENQUEUE(0);
  for (j = 0; j < m; ++j) {
    if (j + 1 < m) ENQUEUE(j + 1);
    CHECK(bfdevEventSync(evCol[j & 1]));
    cplx const *hCol = hHslot[j & 1];

    double resmax = 0;
    for (size_t p = 0; p < nrhs; ++p) {
      cplx *col = H + (j * (m + 2)) * nrhs;        /* column j, entries i = 0..j+1 at col[i*nrhs + p] */
      for (size_t i = 0; i < j + 2; ++i) col[i * nrhs + p] = hCol[i * nrhs + p];
      for (size_t i = 0; i < j; ++i)               /* earlier rotations (:206-212) */
        applyGivens(&col[i * nrhs + p], &col[(i + 1) * nrhs + p], Jc[i * nrhs + p], Js[i * nrhs + p]);
      givens(col[j * nrhs + p], col[(j + 1) * nrhs + p], &Jc[j * nrhs + p], &Js[j * nrhs + p]);   /* (:214-219) */
      applyGivens(&col[j * nrhs + p], &col[(j + 1) * nrhs + p], Jc[j * nrhs + p], Js[j * nrhs + p]);
      applyGivens(&S[j * nrhs + p], &S[(j + 1) * nrhs + p], Jc[j * nrhs + p], Js[j * nrhs + p]);  /* (:222-228) */
      double r = cabs(S[(j + 1) * nrhs + p]);
      if (r > resmax) resmax = r;
    }
    lastResidual = resmax / beta;                  /* (:230-231) */
    if (lastResidual < tol) { converged = 1; break; }   /* j is NOT incremented (:235-241) */
  }
#undef ENQUEUE
  if (!converged) j = m;

  /* back substitution per RHS on the j x j triangle (:245-285) */
  for (size_t p = 0; p < nrhs; ++p) {
    for (size_t r = j; r-- > 0;) {
      cplx acc = S[r * nrhs + p];
      for (size_t c = r + 1; c < j; ++c) acc -= H[(c * (m + 2) + r) * nrhs + p] * y[c * nrhs + p];
      y[r * nrhs + p] = acc / H[(r * (m + 2) + r) * nrhs + p];
    }
  }
  CHECK(bfdevMemcpyH2DAsync(dY, y, j * nrhs * 16, stream));
  CHECK(bfdevGmresUpdate(dX0, dV, dY, (uint32_t)j, dX, n, (uint32_t)nrhs, stream));
  CHECK(bfdevSync(stream));

finish:
  if (numIter) *numIter = j;
  if (residual) *residual = lastResidual;
done:
  (void)bfdevSync(stream);   /* a speculative iteration may still be in flight: drain before its buffers go */
  bfdevFree(dV); bfdevFree(dW); bfdevFree(dPartA); bfdevFree(dPartB); bfdevFree(dH); bfdevFree(dY); bfdevFree(dAX0);
  bfdevFree(dPartAll); bfdevFree(dH1); bfdevFree(dH2); bfdevFree(dPre);
  bfdevEventDestroy(evCol[0]); bfdevEventDestroy(evCol[1]);
  bfdevHostFreePinned(hHpinned);
  free(H); free(S); free(Jc); free(Js); free(y); free(rnorm);
  return rc;
#undef CHECK
}

int bfhipSolveGMRES(BfhipOperator *op, void const *B, size_t ldb, size_t nrhs, void const *X0, size_t ldx0, double tol,
                    size_t maxNumIter, size_t *numIter, double *residual, void *X, size_t ldx) {
  if (!op || !B || !X) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (nrhs == 0 || ldb < nrhs || ldx < nrhs || (X0 && ldx0 < nrhs)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad nrhs / leading dimension");
  uint64_t n = bfhipGetNumRows(op);
  size_t const vecBytes = (size_t)n * nrhs * 16;
  void *dB = NULL, *dX0 = NULL, *dX = NULL;
  int prev = -1, dev = bfhipOperatorDevice(op);
  if (dev < 0) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator has no device (compiled with BFHIP_FLAG_PLAN_ONLY)");
  char *pack = malloc(vecBytes ? vecBytes : 1);
  int rc = 0;
  if (!pack) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  bfdevGetDevice(&prev);
  if (prev != dev && (rc = bfdevSetDevice(dev))) { free(pack); return rc; }
  if ((rc = bfdevMalloc(&dB, vecBytes))) goto done;
  if ((rc = bfdevMalloc(&dX, vecBytes))) goto done;
  for (uint64_t i = 0; i < n; ++i) memcpy(pack + i * nrhs * 16, (char const *)B + i * ldb * 16, nrhs * 16);
  if ((rc = bfdevMemcpyH2D(dB, pack, vecBytes))) goto done;
  if (X0) {
    if ((rc = bfdevMalloc(&dX0, vecBytes))) goto done;
    for (uint64_t i = 0; i < n; ++i) memcpy(pack + i * nrhs * 16, (char const *)X0 + i * ldx0 * 16, nrhs * 16);
    if ((rc = bfdevMemcpyH2D(dX0, pack, vecBytes))) goto done;
  }
  if ((rc = bfhipSolveGMRESDevice(op, dB, nrhs, dX0, tol, maxNumIter, numIter, residual, dX, NULL))) goto done;
  if ((rc = bfdevMemcpyD2H(pack, dX, vecBytes))) goto done;
  for (uint64_t i = 0; i < n; ++i) memcpy((char *)X + i * ldx * 16, pack + i * nrhs * 16, nrhs * 16);
done:
  bfdevFree(dB); bfdevFree(dX0); bfdevFree(dX);
  free(pack);
  if (prev >= 0 && prev != dev) bfdevSetDevice(prev);
  return rc;
}
