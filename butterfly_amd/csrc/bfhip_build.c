/* bfhip_build.c -- host side of the fac_helm2 value builder (include/bfhip_build.h).
 *
 * The reference builds one block at a time on the CPU (src/fac_helm2.c:42-509:
 * kernel matrix, SVD-based least squares, store the block).  Here the dense
 * leaves of the already laid-out operand are processed in batches that fit a
 * device workspace: every kernel matrix of a batch in one launch, every
 * least-squares problem of a batch in one Jacobi launch per size class, two
 * batched GEMM launches, then one launch that copies the batch's leaves into
 * the packed arena the apply kernels read.  Nothing but the recipes and the
 * point coordinates crosses PCIe.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"
#include "../../include/bfhip_build.h"

#define DEFAULT_WORKSPACE ((uint64_t)8 << 30)

static double nowSeconds(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static BfBuildPts toPts(BfhipPointSet const *p) {
  BfBuildPts q;
  q.kind = p->kind; q.count = p->count; q.first = p->first; q.cx = p->cx; q.cy = p->cy; q.r = p->r;
  return q;
}

static int checkPts(BfhipPointSet const *p, BfhipHelm2Problem const *prob, char const *what, uint64_t idx) {
  if (p->count == 0) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: empty %s point set", (unsigned long long)idx, what);
  if (p->kind == BFHIP_PTS_TREE || p->kind == BFHIP_PTS_TREE_TGT) {
    if (p->kind == BFHIP_PTS_TREE_TGT && !prob->tgtPoints)
      return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: %s points refer to a target tree but tgtPoints is NULL", (unsigned long long)idx, what);
    uint64_t const numPoints = p->kind == BFHIP_PTS_TREE ? prob->numPoints : prob->numTgtPoints;
    if (p->first + p->count > numPoints)
      return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: %s points [%llu, %llu) exceed numPoints", (unsigned long long)idx, what,
                       (unsigned long long)p->first, (unsigned long long)(p->first + p->count));
  } else if (p->kind != BFHIP_PTS_CIRCLE) {
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: unknown point-set kind %u", (unsigned long long)idx, p->kind);
  }
  return 0;
}

static uint32_t leafRows(BfhipHelm2Recipe const *r) { return r->kind == BFHIP_LEAF_KERNEL ? r->tgt.count : r->equiv.count; }

static int checkProblem(BfhipHelm2Problem const *prob) {
  if (!prob) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL problem");
  if (prob->structSize < sizeof(BfhipHelm2Problem)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipHelm2Problem.structSize too small");
  if (prob->layerPot != BFHIP_LAYER_POTENTIAL_SINGLE && prob->layerPot != BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE &&
      prob->layerPot != BFHIP_LAYER_POTENTIAL_PV_DOUBLE && prob->layerPot != BFHIP_LAYER_POTENTIAL_COMBINED_FIELD)
    return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "layer potential %u is not built on the device (S = 1, D = 2, S' = 3, combined = 5 are)",
                     prob->layerPot);                 /* as bfHelm2GetKernelMatrix, src/helm2.c:296-316 */
  if (prob->layerPot != BFHIP_LAYER_POTENTIAL_SINGLE && !prob->normals)
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "this layer potential needs the unit normals at the points");
  if (prob->krOrder != 0 && prob->krOrder != 2 && prob->krOrder != 6 && prob->krOrder != 10)
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "Kapur-Rokhlin order must be 0, 2, 6 or 10");    /* src/quadrature.c:106 */
  if (prob->krOrder && !prob->origIndex) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the KR correction needs origIndex");
  if (prob->tgtPoints && (prob->krOrder || prob->selfValue[0] != 0 || prob->selfValue[1] != 0))
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "selfValue / KR correction apply to square operators (no separate target tree)");
  if (prob->krOrder && prob->numPoints < 2 * (uint64_t)prob->krOrder + 1)
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "too few points for the KR correction");          /* src/quadrature.c:115 */
  if (!prob->points || (!prob->recipes && prob->numRecipes)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL points / recipes");
  if (!(prob->wavenumber > 0)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "wavenumber must be positive");
  return 0;
}

static int checkRecipe(BfhipHelm2Problem const *prob, uint64_t i) {
  BfhipHelm2Recipe const *r = &prob->recipes[i];
  int rc;
  if ((rc = checkPts(&r->src, prob, "source", i))) return rc;
  if ((rc = checkPts(&r->tgt, prob, "target", i))) return rc;
  if (r->kind == BFHIP_LEAF_REEXP) {
    if ((rc = checkPts(&r->equiv, prob, "equivalent-source", i))) return rc;
    if (r->tgt.count < r->equiv.count)
      return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "recipe %llu: fewer check points (%u) than equivalent sources (%u)",
                       (unsigned long long)i, r->tgt.count, r->equiv.count);
  } else if (r->kind != BFHIP_LEAF_KERNEL) {
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: unknown kind %u", (unsigned long long)i, r->kind);
  } else if (prob->layerPot == BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE &&
             (r->tgt.kind == BFHIP_PTS_CIRCLE || (r->tgt.kind == BFHIP_PTS_TREE_TGT && !prob->tgtNormals))) {
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: an S' kernel leaf needs target normals, i.e. tree-point targets", (unsigned long long)i);
  }
  return 0;
}

/* potential codes of the device layer: what the operator's own leaves use, and what the two
 * matrices of a re-expansion use (BF_PROXY_LAYER_POT, include/bf/layer_pot.h:63-69) */
static uint32_t leafPotCode(uint32_t layerPot) {
  return layerPot == BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE ? 1 : layerPot == BFHIP_LAYER_POTENTIAL_PV_DOUBLE ? 2
       : layerPot == BFHIP_LAYER_POTENTIAL_COMBINED_FIELD ? 3 : 0;
}
static uint32_t proxyPotCode(uint32_t layerPot) {
  return layerPot == BFHIP_LAYER_POTENTIAL_PV_DOUBLE ? 2 : layerPot == BFHIP_LAYER_POTENTIAL_COMBINED_FIELD ? 3 : 0;
}

/* device copies of what every kernel evaluation reads */
typedef struct DevEnv { BfEvalEnv env; void *dPoints, *dNormals, *dWeights, *dOrig, *dHits, *dTgtPoints, *dTgtNormals; } DevEnv;

static void envFree(DevEnv *e) {
  bfdevFree(e->dPoints); bfdevFree(e->dNormals); bfdevFree(e->dWeights); bfdevFree(e->dOrig); bfdevFree(e->dHits);
  bfdevFree(e->dTgtPoints); bfdevFree(e->dTgtNormals);
  memset(e, 0, sizeof *e);
}

static int envUpload(BfhipHelm2Problem const *prob, DevEnv *e) {
  memset(e, 0, sizeof *e);
  size_t const n = (size_t)(prob->numPoints ? prob->numPoints : 1);
  int rc = bfdevMalloc(&e->dPoints, n * 16);
  if (!rc) rc = bfdevMemcpyH2D(e->dPoints, prob->points, (size_t)prob->numPoints * 16);
  if (!rc && prob->normals) {
    rc = bfdevMalloc(&e->dNormals, n * 16);
    if (!rc) rc = bfdevMemcpyH2D(e->dNormals, prob->normals, (size_t)prob->numPoints * 16);
  }
  if (!rc && prob->colWeights) {
    rc = bfdevMalloc(&e->dWeights, n * 8);
    if (!rc) rc = bfdevMemcpyH2D(e->dWeights, prob->colWeights, (size_t)prob->numPoints * 8);
  }
  if (!rc && prob->tgtPoints) {
    size_t const m = (size_t)(prob->numTgtPoints ? prob->numTgtPoints : 1);
    rc = bfdevMalloc(&e->dTgtPoints, m * 16);
    if (!rc) rc = bfdevMemcpyH2D(e->dTgtPoints, prob->tgtPoints, (size_t)prob->numTgtPoints * 16);
    if (!rc && prob->tgtNormals) {
      rc = bfdevMalloc(&e->dTgtNormals, m * 16);
      if (!rc) rc = bfdevMemcpyH2D(e->dTgtNormals, prob->tgtNormals, (size_t)prob->numTgtPoints * 16);
    }
  }
  if (!rc && prob->krOrder) {
    rc = bfdevMalloc(&e->dOrig, n * 8);
    if (!rc) rc = bfdevMemcpyH2D(e->dOrig, prob->origIndex, (size_t)prob->numPoints * 8);
    if (!rc) rc = bfdevMalloc(&e->dHits, 8);
    if (!rc) rc = bfdevMemset(e->dHits, 0, 8);
  }
  if (rc) { envFree(e); return rc; }
  e->env.dOrigIndex = e->dOrig; e->env.numPoints = prob->numPoints; e->env.krOrder = prob->krOrder;
  e->env.dKrHits = (unsigned long long *)e->dHits;
  e->env.dTgtPoints = e->dTgtPoints; e->env.dTgtNormals = e->dTgtNormals;
  e->env.dPoints = e->dPoints; e->env.dNormals = e->dNormals; e->env.dColWeights = e->dWeights;
  e->env.wavenumber = prob->wavenumber; e->env.selfRe = prob->selfValue[0]; e->env.selfIm = prob->selfValue[1];
  e->env.alphaRe = prob->alpha[0]; e->env.alphaIm = prob->alpha[1]; e->env.betaRe = prob->beta[0]; e->env.betaIm = prob->beta[1];
  return 0;
}

/* workspace of one re-expansion problem, in complex elements */
typedef struct ReexpWs { uint64_t zeq, v, x, zor, t, scale, total; } ReexpWs;

/* Problems with at least this many equivalent sources go through the QR preconditioner (bfQrcpKernel) before the
 * Jacobi kernel: those are the ones whose stacked matrix does not stay in LDS.  BFHIP_JACOBI_QR_MIN overrides the
 * default (0: every problem, a test hook; a huge value: none). */
static uint32_t qrMinCols(void) {
  char const *env = getenv("BFHIP_JACOBI_QR_MIN");
  if (env && env[0]) return (uint32_t)strtoul(env, NULL, 10);
  return 65;
}
static int usesQr(BfhipHelm2Recipe const *r, uint32_t qrMin) {
  return r->equiv.count >= qrMin && bfdevQrcpFits(r->tgt.count, r->equiv.count);
}

static ReexpWs reexpWs(BfhipHelm2Recipe const *r, uint32_t qrMin) {
  uint64_t const mt = r->tgt.count, me = r->equiv.count, n = r->src.count;
  ReexpWs w;
  w.zeq = 0;
  w.v = w.zeq + mt * me;
  w.x = w.v + me * me;
  w.zor = w.x + (usesQr(r, qrMin) ? me * me : 0);
  w.t = w.zor + mt * n;
  w.scale = w.t + me * n;
  w.total = w.scale + (me + 1) / 2;
  return w;
}

static uint64_t recipeCost(BfhipHelm2Recipe const *r) {
  uint64_t c = (uint64_t)leafRows(r) * r->src.count;
  if (r->kind == BFHIP_LEAF_REEXP) c += reexpWs(r, qrMinCols()).total;
  return c;
}

/* Device buffers of the batches, kept from one batch to the next (allocating and freeing ~100 GB per batch costs seconds). */
typedef struct BatchBufs { void *store, *ws; uint64_t storeElems, wsElems; } BatchBufs;
static void batchBufsFree(BatchBufs *b) { bfdevFree(b->store); bfdevFree(b->ws); memset(b, 0, sizeof *b); }
static int batchBufsReserve(BatchBufs *b, uint64_t storeElems, uint64_t wsElems) {
  int rc = 0;
  if (storeElems > b->storeElems || !b->store) {
    bfdevFree(b->store); b->store = NULL; b->storeElems = 0;
    if ((rc = bfdevMalloc(&b->store, (size_t)(storeElems ? storeElems : 1) * 16))) return rc;
    b->storeElems = storeElems;
  }
  if (wsElems > b->wsElems) {
    bfdevFree(b->ws); b->ws = NULL; b->wsElems = 0;
    if ((rc = bfdevMalloc(&b->ws, (size_t)wsElems * 16))) return rc;
    b->wsElems = wsElems;
  }
  return 0;
}

/* leaf-store and workspace elements of a batch */
static void batchSizes(BfhipHelm2Problem const *prob, uint64_t const *idx, uint64_t count, uint32_t qrMin, uint64_t *storeElems, uint64_t *wsElems);

/* Compute recipes idx[0..count) on the current device.  *dStore receives a
 * device buffer holding the leaves column-major at storeOff[i] (elements). */
static int buildBatch(BfhipHelm2Problem const *prob, uint64_t const *idx, uint64_t count, BfEvalEnv const *env,
                      BatchBufs *bufs, void **dStore, uint64_t *storeOff, BfhipBuildStats *st) {
  int rc = 0;
  *dStore = NULL;
  void *dWs = NULL;
  uint64_t storeElems = 0, wsElems = 0, numReexp = 0;
  uint64_t *wsOff = malloc((count + 1) * sizeof *wsOff);
  BfEvalMat *mats = malloc(2 * count * sizeof *mats);
  uint64_t *prefix = malloc((2 * count + 1) * sizeof *prefix);
  BfSvdProb *probs = malloc((count + 1) * sizeof *probs);
  BfGemmJob *g1 = malloc((count + 1) * sizeof *g1), *g2 = malloc((count + 1) * sizeof *g2);
  BfQrProb *qr = malloc((count + 1) * sizeof *qr);
  uint64_t *qrOf = malloc((count + 1) * sizeof *qrOf);          /* least-squares problem of the k-th QR problem */
  uint32_t *qrRank = malloc((count + 1) * sizeof *qrRank);
  uint32_t const qrMin = qrMinCols();
  uint64_t nq = 0;
  if (!wsOff || !mats || !prefix || !probs || !g1 || !g2 || !qr || !qrOf || !qrRank) {
    rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder batch)");
    goto done;
  }
  for (uint64_t i = 0; i < count; ++i) {
    BfhipHelm2Recipe const *r = &prob->recipes[idx[i]];
    storeOff[i] = storeElems;
    storeElems += (uint64_t)leafRows(r) * r->src.count;
    wsOff[i] = wsElems;
    if (r->kind == BFHIP_LEAF_REEXP) { wsElems += reexpWs(r, qrMin).total; ++numReexp; }
  }
  double const tAlloc = nowSeconds();
  if ((rc = batchBufsReserve(bufs, storeElems, wsElems))) goto done;
  *dStore = bufs->store; dWs = bufs->ws;
  char *store = (char *)*dStore, *ws = (char *)dWs;
  uint64_t nm = 0, np = 0;
  prefix[0] = 0;
  uint32_t const leafPot = leafPotCode(prob->layerPot), proxyPot = proxyPotCode(prob->layerPot);
  for (uint64_t i = 0; i < count; ++i) {
    BfhipHelm2Recipe const *r = &prob->recipes[idx[i]];
    if (r->kind == BFHIP_LEAF_KERNEL) {
      mats[nm].src = toPts(&r->src); mats[nm].tgt = toPts(&r->tgt); mats[nm].dst = store + storeOff[i] * 16;
      mats[nm].pot = leafPot; mats[nm].decorate = 1;
      prefix[nm + 1] = prefix[nm] + ((uint64_t)r->tgt.count * r->src.count + BF_EVAL_TILE - 1) / BF_EVAL_TILE;
      ++nm;
      st->kernelLeaves += 1;
      st->kernelEvals += (uint64_t)r->tgt.count * r->src.count;
      continue;
    }
    ReexpWs const w = reexpWs(r, qrMin);
    char *base = ws + wsOff[i] * 16;
    uint32_t const mt = r->tgt.count, me = r->equiv.count, n = r->src.count;
    /* re-expansions use the proxy potential; the column weights of the operator scale Z_orig */
    mats[nm].src = toPts(&r->equiv); mats[nm].tgt = toPts(&r->tgt); mats[nm].dst = base + w.zeq * 16;
    mats[nm].pot = proxyPot; mats[nm].decorate = 0;
    prefix[nm + 1] = prefix[nm] + ((uint64_t)mt * me + BF_EVAL_TILE - 1) / BF_EVAL_TILE;
    ++nm;
    mats[nm].src = toPts(&r->src); mats[nm].tgt = toPts(&r->tgt); mats[nm].dst = base + w.zor * 16;
    mats[nm].pot = proxyPot; mats[nm].decorate = 1;
    prefix[nm + 1] = prefix[nm] + ((uint64_t)mt * n + BF_EVAL_TILE - 1) / BF_EVAL_TILE;
    ++nm;
    probs[np].a = base + w.zeq * 16; probs[np].v = base + w.v * 16; probs[np].scale = (double *)(base + w.scale * 16);
    probs[np].mt = mt; probs[np].me = me; probs[np].dim = mt > me ? mt : me; probs[np].pad = 0;
    if (usesQr(r, qrMin)) {
      qr[nq].a = probs[np].a; qr[nq].b = base + w.zor * 16; qr[nq].x = base + w.x * 16;
      qr[nq].mt = mt; qr[nq].me = me; qr[nq].n = n; qr[nq].dim = probs[np].dim;
      qrOf[nq++] = np;
    }
    /* T = diag(1/sigma^2) (U Sigma)^H Z_orig */
    memset(&g1[np], 0, sizeof g1[np]);
    g1[np].a = probs[np].a; g1[np].b = base + w.zor * 16; g1[np].c = base + w.t * 16; g1[np].scale = probs[np].scale;
    g1[np].M = me; g1[np].N = n; g1[np].K = mt; g1[np].lda = mt; g1[np].ldb = mt; g1[np].ldc = me; g1[np].transA = 1;
    /* X = V T */
    memset(&g2[np], 0, sizeof g2[np]);
    g2[np].a = probs[np].v; g2[np].b = base + w.t * 16; g2[np].c = store + storeOff[i] * 16; g2[np].scale = NULL;
    g2[np].M = me; g2[np].N = n; g2[np].K = me; g2[np].lda = me; g2[np].ldb = me; g2[np].ldc = me; g2[np].transA = 0;
    ++np;
    st->reexpLeaves += 1;
    st->kernelEvals += (uint64_t)mt * me + (uint64_t)mt * n;
  }
  char const *penv = getenv("BFHIP_JACOBI_PROFILE");
  int const profile = penv && penv[0] == '1';
  double tp = tAlloc;
#define BF_PHASE(name) do { if (profile) { double const t = nowSeconds(); fprintf(stderr, "[build] %-12s %.3f s\n", name, t - tp); tp = t; } } while (0)
  BF_PHASE("alloc + lists");
  if ((rc = bfdevBuildEval(mats, prefix, nm, env))) goto done;
  BF_PHASE("kernel eval");
  /* QR-preconditioned problems: A P = Q R in place, Z_orig <- Q^H Z_orig, and the Jacobi kernel gets X = (R[0:r] P^T)^H
   * (me x r) instead of A:  X V1 = W  =>  pinv(A) Z_orig = W diag(1/sigma^2) V1^H (Q^H Z_orig)[0:r]  (bfhip_build.hip) */
  if ((rc = bfdevBuildQrcp(qr, nq, qrRank))) goto done;
  BF_PHASE("qr");
  for (uint64_t k = 0; k < nq; ++k) {
    uint64_t const q = qrOf[k];
    uint32_t const me = qr[k].me, rk = qrRank[k];
    probs[q].a = qr[k].x; probs[q].mt = me; probs[q].me = rk;                     /* V1: rk x rk in the v block */
    g1[q].a = probs[q].v; g1[q].M = rk; g1[q].K = rk; g1[q].lda = rk;             /* T[0:rk] = diag(1/sigma^2) V1^H (Q^H Z_orig)[0:rk] */
    g2[q].a = qr[k].x; g2[q].K = rk; g2[q].lda = me;                              /* X = W T[0:rk] */
    st->qrProblems += 1; st->qrColumns += me; st->qrRank += rk;
    /* rank 0: the matrix is zero or not finite (the pivot loop stops on `!(best >= threshold)`) -- nothing for the Jacobi kernel to flag.
     * Count it like an SVD that did not converge (the build then fails unless BFHIP_ALLOW_UNCONVERGED_SVD=1); with K = 0 the second
     * GEMM writes an exactly zero leaf instead of leaving what the reused store held from the batch before. */
    if (!rk) st->notConverged += 1;
  }
  BfSvdStats ss = {st->maxSweeps, 0, 0, 0};
  if ((rc = bfdevBuildJacobi(probs, np, &ss))) goto done;
  st->maxSweeps = ss.maxSweeps; st->notConverged += ss.notConverged; st->truncated += ss.truncated; st->sumSweeps += ss.sumSweeps;
  BF_PHASE("jacobi");
  if ((rc = bfdevBuildGemm(g1, np))) goto done;
  if ((rc = bfdevBuildGemm(g2, np))) goto done;
  BF_PHASE("gemm");
#undef BF_PHASE
  st->numBatches += 1;
done:
  if (rc) *dStore = NULL;
  free(wsOff); free(mats); free(prefix); free(probs); free(g1); free(g2); free(qr); free(qrOf); free(qrRank);
  return rc;
}

static void batchSizes(BfhipHelm2Problem const *prob, uint64_t const *idx, uint64_t count, uint32_t qrMin, uint64_t *storeElems, uint64_t *wsElems) {
  uint64_t se = 0, we = 0;
  for (uint64_t i = 0; i < count; ++i) {
    BfhipHelm2Recipe const *r = &prob->recipes[idx[i]];
    se += (uint64_t)leafRows(r) * r->src.count;
    if (r->kind == BFHIP_LEAF_REEXP) we += reexpWs(r, qrMin).total;
  }
  *storeElems = se; *wsElems = we;
}

/* ---- arena fill: called by the compile step instead of packing host values ---- */
typedef struct BuildCtx {
  BfhipHelm2Problem const *prob;
  BfhipBuildStats *stats;
} BuildCtx;

typedef struct PieceRec { uint64_t rec, dataOff; uint32_t row0, col0, mr, mrPad, ncols; } PieceRec;

static int fillArena(BfPlan const *pl, BfIr const *ir, void *dArena, void *vctx) {
  BuildCtx *ctx = vctx;
  BfhipHelm2Problem const *prob = ctx->prob;
  BfhipBuildStats *st = ctx->stats;
  int rc = 0;
  if (pl->dtype != BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "the Helmholtz builder fills complex128 operands");
  uint64_t const R = prob->numRecipes;
  int64_t *recOf = malloc((ir->numNodes + 1) * sizeof *recOf);
  uint64_t *pieceBegin = calloc(R + 2, sizeof *pieceBegin);
  PieceRec *pieces = NULL, *sorted = NULL;
  uint64_t *batch = NULL, *storeOff = NULL;
  BfPackPiece *pack = NULL;
  DevEnv dev;
  memset(&dev, 0, sizeof dev);
  BatchBufs bufs;
  memset(&bufs, 0, sizeof bufs);
  if (!recOf || !pieceBegin) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder)"); goto done; }
  for (uint64_t i = 0; i < ir->numNodes; ++i) recOf[i] = -1;
  for (uint64_t i = 0; i < R && !rc; ++i) {
    BfhipHelm2Recipe const *r = &prob->recipes[i];
    if (r->node >= ir->numNodes || ir->kind[r->node] != BFHIP_NODE_DENSE)
      rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: node %llu is not a dense leaf", (unsigned long long)i, (unsigned long long)r->node);
    else if (recOf[r->node] >= 0)
      rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: leaf %llu already has a recipe", (unsigned long long)i, (unsigned long long)r->node);
    else if ((rc = checkRecipe(prob, i)) == 0) {
      if (ir->rows[r->node] != leafRows(r) || ir->cols[r->node] != r->src.count)
        rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "recipe %llu: leaf %llu is %llu x %llu, recipe yields %u x %u", (unsigned long long)i,
                       (unsigned long long)r->node, (unsigned long long)ir->rows[r->node], (unsigned long long)ir->cols[r->node],
                       leafRows(r), r->src.count);
      recOf[r->node] = (int64_t)i;
    }
  }
  if (rc) goto done;

  /* the plan's pieces, grouped by recipe (counting sort) */
  uint64_t numPieces = 0;
  for (uint64_t s = 0; s < pl->numStages; ++s) numPieces += pl->stages[s].numPieces;
  pieces = malloc((numPieces + 1) * sizeof *pieces);
  sorted = malloc((numPieces + 1) * sizeof *sorted);
  if (!pieces || !sorted) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder pieces)"); goto done; }
  uint64_t np = 0;
  for (uint64_t s = 0; s < pl->numStages && !rc; ++s) {
    BfStage const *stg = &pl->stages[s];
    for (uint64_t i = 0; i < stg->numItems && !rc; ++i) {
      BfDevItem const *it = &stg->items[i];
      uint32_t const mr = it->mrFlags & 0xffffu;
      uint32_t const mrPad = (mr + pl->epl - 1) / pl->epl * pl->epl;
      for (uint32_t k = 0; k < it->numPieces; ++k) {
        BfDevPiece const *pc = &stg->pieces[it->pieceBegin + k];
        BfPieceSrc const *src = &stg->pieceSrc[it->pieceBegin + k];
        if (pc->flags & BF_PIECE_IDENTITY) continue;
        int64_t const rec = recOf[src->node];
        if (rec < 0) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "dense leaf %llu has no recipe", (unsigned long long)src->node); break; }
        PieceRec *p = &pieces[np++];
        p->rec = (uint64_t)rec; p->dataOff = pc->dataOff; p->row0 = src->row0; p->col0 = src->col0;
        p->mr = mr; p->mrPad = mrPad; p->ncols = pc->ncols;
        pieceBegin[rec + 1] += 1;
      }
    }
  }
  if (rc) goto done;
  for (uint64_t i = 0; i < R; ++i) pieceBegin[i + 1] += pieceBegin[i];
  {
    uint64_t *cursor = malloc((R + 1) * sizeof *cursor);
    if (!cursor) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder)"); goto done; }
    memcpy(cursor, pieceBegin, (R + 1) * sizeof *cursor);
    for (uint64_t i = 0; i < np; ++i) sorted[cursor[pieces[i].rec]++] = pieces[i];
    free(cursor);
  }

  char const *penv = getenv("BFHIP_JACOBI_PROFILE");
  int const profile = penv && penv[0] == '1';
  double tph = nowSeconds();
  if ((rc = envUpload(prob, &dev))) goto done;

  /* default workspace: half of what is free once the arena is allocated, at least 8 GiB asked for --
   * large batches keep all CUs busy through the tail of the biggest least-squares problems */
  uint64_t wsBytes = prob->workspaceBytes;
  if (!wsBytes) {
    uint64_t freeBytes = 0;
    if ((rc = bfdevMemFree(&freeBytes))) goto done;
    wsBytes = freeBytes / 2 > DEFAULT_WORKSPACE ? freeBytes / 2 : DEFAULT_WORKSPACE;
  }
  uint64_t const budget = wsBytes / 16;
  batch = malloc((R + 1) * sizeof *batch);
  storeOff = malloc((R + 1) * sizeof *storeOff);
  if (!batch || !storeOff) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder)"); goto done; }
  /* the largest batch sizes the buffers once */
  {
    uint64_t maxStore = 0, maxWs = 0;
    for (uint64_t j = 0; j < R;) {
      uint64_t nb = 0, cost = 0;
      for (; j < R; ++j) {
        if (pieceBegin[j + 1] == pieceBegin[j]) continue;
        uint64_t const c = recipeCost(&prob->recipes[j]);
        if (nb && cost + c > budget) break;
        batch[nb++] = j; cost += c;
      }
      if (!nb) break;
      uint64_t se, we;
      batchSizes(prob, batch, nb, qrMinCols(), &se, &we);
      maxStore = se > maxStore ? se : maxStore; maxWs = we > maxWs ? we : maxWs;
    }
    if ((rc = batchBufsReserve(&bufs, maxStore, maxWs))) goto done;
  }
  uint64_t i = 0;
  while (i < R && !rc) {
    /* next batch: recipes whose leaves survive in this plan, until the workspace is full */
    uint64_t nb = 0, cost = 0, packCount = 0;
    for (; i < R; ++i) {
      if (pieceBegin[i + 1] == pieceBegin[i]) continue;          /* sharded away */
      uint64_t const c = recipeCost(&prob->recipes[i]);
      if (nb && cost + c > budget) break;
      batch[nb++] = i; cost += c; packCount += pieceBegin[i + 1] - pieceBegin[i];
    }
    if (!nb) break;
    void *dStore = NULL;
    if (profile) { double const t = nowSeconds(); fprintf(stderr, "[build] host before batch %.3f s\n", t - tph); tph = t; }
    if ((rc = buildBatch(prob, batch, nb, &dev.env, &bufs, &dStore, storeOff, st))) break;
    if (profile) { double const t = nowSeconds(); fprintf(stderr, "[build] batch total %.3f s (%llu recipes)\n", t - tph, (unsigned long long)nb); tph = t; }
    pack = malloc((packCount + 1) * sizeof *pack);
    if (!pack) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (builder pack list)");
    uint64_t q = 0;
    for (uint64_t b = 0; b < nb && !rc; ++b) {
      uint64_t const rec = batch[b];
      uint32_t const ld = leafRows(&prob->recipes[rec]);
      for (uint64_t k = pieceBegin[rec]; k < pieceBegin[rec + 1]; ++k) {
        PieceRec const *p = &sorted[k];
        pack[q].dataOff = p->dataOff;
        pack[q].srcOff = storeOff[b] + (uint64_t)p->col0 * ld + p->row0;
        pack[q].srcLd = ld; pack[q].mr = p->mr; pack[q].mrPad = p->mrPad; pack[q].ncols = p->ncols;
        ++q;
      }
    }
    if (!rc) rc = bfdevBuildPack(dArena, dStore, pack, q);
    free(pack); pack = NULL;
    if (profile) { double const t = nowSeconds(); fprintf(stderr, "[build] pack + free  %.3f s\n", t - tph); tph = t; }
  }
  /* every KR pair must have been met exactly once by a dense near-field leaf; a pair inside a
   * butterflied block cannot be corrected through the values */
  if (!rc && prob->krOrder && !prob->tgtPoints && pl->numRows == prob->numPoints && pl->numCols == prob->numPoints) {
    unsigned long long hits = 0;
    rc = bfdevMemcpyD2H(&hits, dev.dHits, 8);
    if (!rc && hits != 2ull * prob->krOrder * prob->numPoints)
      rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "KR correction: %llu of %llu near-diagonal entries lie in dense leaves",
                     hits, 2ull * prob->krOrder * prob->numPoints);
  }
done:
  batchBufsFree(&bufs);
  envFree(&dev);
  free(recOf); free(pieceBegin); free(pieces); free(sorted); free(batch); free(storeOff);
  return rc;
}

int bfhipBuildHelm2(BfhipDesc const *desc, BfhipHelm2Problem const *prob, BfhipOptions const *opts, BfhipOperator **out,
                    BfhipBuildStats *stats) {
  if (!desc || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  int rc = checkProblem(prob);
  if (rc) return rc;
  if (desc->dtype != BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "the Helmholtz builder fills complex128 operands");
  if (opts && (opts->flags & BFHIP_FLAG_PLAN_ONLY)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BFHIP_FLAG_PLAN_ONLY has no device to build on");
  BfhipBuildStats local;
  memset(&local, 0, sizeof local);
  local.structSize = sizeof local;
  BuildCtx ctx = {prob, &local};
  BfIr ir;
  if ((rc = bfIrFromDesc(desc, &ir))) return rc;
  double const t0 = nowSeconds();
  rc = bfhipCompileIrFill(&ir, opts, fillArena, &ctx, out);
  local.seconds = nowSeconds() - t0;
  if (stats && stats->structSize >= sizeof local) *stats = local;     /* also on failure: the counts say why */
  if (!rc && stats && stats->structSize < sizeof local) { bfhipFree(out); return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipBuildStats.structSize too small"); }
  if (!rc && local.notConverged && !getenv("BFHIP_ALLOW_UNCONVERGED_SVD")) {
    /* a least-squares leaf from an SVD that hit the sweep cap is a silently wrong operator: refuse
     * it (BFHIP_ALLOW_UNCONVERGED_SVD=1 in the environment keeps the operator, for diagnosis) */
    bfhipFree(out);
    return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "%llu Jacobi SVD problem(s) did not converge within the sweep cap; operator discarded",
                     (unsigned long long)local.notConverged);
  }
  return rc;
}

int bfhipHelm2BuildLeaf(BfhipHelm2Problem const *prob, uint64_t recipeIndex, int device, void *out) {
  int rc = checkProblem(prob);
  if (rc) return rc;
  if (!out || recipeIndex >= prob->numRecipes) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad recipe index / NULL output");
  if ((rc = checkRecipe(prob, recipeIndex))) return rc;
  int prev = -1;
  bfdevGetDevice(&prev);
  if ((rc = bfdevSetDevice(device))) return rc;
  BfhipHelm2Recipe const *r = &prob->recipes[recipeIndex];
  uint64_t const m = leafRows(r), n = r->src.count;
  void *dStore = NULL;
  DevEnv dev;
  memset(&dev, 0, sizeof dev);
  double *tmp = malloc((size_t)m * n * 16);
  BfhipBuildStats st;
  memset(&st, 0, sizeof st);
  uint64_t off = 0;
  if (!tmp) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  if (!rc) rc = envUpload(prob, &dev);
  BatchBufs bufs;
  memset(&bufs, 0, sizeof bufs);
  if (!rc) rc = buildBatch(prob, &recipeIndex, 1, &dev.env, &bufs, &dStore, &off, &st);
  if (!rc) rc = bfdevMemcpyD2H(tmp, dStore, (size_t)m * n * 16);
  if (!rc) {
    double *o = out;                                   /* column-major store -> row-major result */
    for (uint64_t i = 0; i < m; ++i)
      for (uint64_t j = 0; j < n; ++j) { o[2 * (i * n + j)] = tmp[2 * (j * m + i)]; o[2 * (i * n + j) + 1] = tmp[2 * (j * m + i) + 1]; }
  }
  batchBufsFree(&bufs); envFree(&dev); free(tmp);
  if (prev >= 0) bfdevSetDevice(prev);
  return rc;
}

int bfhipHelm2DenseApplyDevice(BfhipHelm2Problem const *prob, int device, void const *dX, void *dY, void *stream) {
  int rc = checkProblem(prob);
  if (rc) return rc;
  if (!dX || !dY) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  int prev = -1;
  bfdevGetDevice(&prev);
  if ((rc = bfdevSetDevice(device))) return rc;
  DevEnv dev;
  rc = envUpload(prob, &dev);
  if (!rc && prob->tgtPoints && prob->layerPot == BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE && !prob->tgtNormals)
    rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "S' on a separate target tree needs tgtNormals");
  if (!rc) rc = bfdevHelm2Dense(&dev.env, leafPotCode(prob->layerPot), prob->numPoints, prob->tgtPoints ? prob->numTgtPoints : 0, dX, dY, stream);
  envFree(&dev);
  if (prev >= 0) bfdevSetDevice(prev);
  return rc;
}

int bfhipHelm2DenseApply(BfhipHelm2Problem const *prob, int device, void const *X, void *Y) {
  int rc = checkProblem(prob);
  if (rc) return rc;
  if (!X || !Y) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  int prev = -1;
  bfdevGetDevice(&prev);
  if ((rc = bfdevSetDevice(device))) return rc;
  void *dX = NULL, *dY = NULL;
  uint64_t const m = prob->tgtPoints ? prob->numTgtPoints : prob->numPoints;
  rc = bfdevMalloc(&dX, (size_t)(prob->numPoints ? prob->numPoints : 1) * 16);
  if (!rc) rc = bfdevMalloc(&dY, (size_t)(m ? m : 1) * 16);
  if (!rc) rc = bfdevMemcpyH2D(dX, X, (size_t)prob->numPoints * 16);
  if (!rc) rc = bfhipHelm2DenseApplyDevice(prob, -1, dX, dY, NULL);
  if (!rc) rc = bfdevMemcpyD2H(Y, dY, (size_t)m * 16);
  bfdevFree(dX); bfdevFree(dY);
  if (prev >= 0) bfdevSetDevice(prev);
  return rc;
}
