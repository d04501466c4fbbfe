/* Host-side code of libbfhip (IR walker, planner, layout, plan inspection) under AddressSanitizer /
 * UBSan, without a GPU: the device layer is replaced by stubs that fail, and only paths that never
 * reach it are driven (BFHIP_FLAG_PLAN_ONLY).  Built and run by tests/test_host_asan.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bfhip_build.h"
#include "../../oracle/bfref.h"      /* byte-compatible BfMat graphs for the graph walker (bfhipCompile) */

#define CHECK(c) do { int rc_ = (c); if (rc_) { fprintf(stderr, "%s -> %d (%s)\n", #c, rc_, bfhipLastErrorMessage()); return 1; } } while (0)

static int drive(double const *pts, uint64_t n, double const *tgt, uint64_t m, double k) {
  BfhipHelm2Layout *lay = NULL;
  CHECK(bfhipHelm2LayoutCreate2(pts, n, tgt, m, k, &lay));
  BfhipDesc const *d = bfhipHelm2LayoutGetDesc(lay);
  uint64_t nrec = 0;
  BfhipHelm2Recipe const *rec = bfhipHelm2LayoutGetRecipes(lay, &nrec);
  if (!d || !rec || !nrec || !bfhipHelm2LayoutGetPerm(lay) || !bfhipHelm2LayoutGetTreePoints(lay)) return 2;
  for (unsigned flags = BFHIP_FLAG_PLAN_ONLY; flags <= (BFHIP_FLAG_PLAN_ONLY | BFHIP_FLAG_ADJOINT); flags += BFHIP_FLAG_ADJOINT) {
    for (unsigned maxRhs = 1; maxRhs <= 64; maxRhs += 63) {
      BfhipOptions o;
      memset(&o, 0, sizeof o);
      o.structSize = sizeof o; o.device = -1; o.flags = flags; o.maxRhs = maxRhs;
      BfhipOperator *op = NULL;
      CHECK(bfhipCompileDesc(d, &o, &op));
      BfhipPlanInfo info;
      memset(&info, 0, sizeof info);
      info.structSize = sizeof info;
      CHECK(bfhipPlanGetInfo(op, &info));
      uint64_t const stages = info.numStages + info.numStagesT;
      uint64_t items = 0;
      for (uint64_t s = 0; s < stages; ++s) {
        BfhipStageView v;
        memset(&v, 0, sizeof v);
        v.structSize = sizeof v;
        CHECK(bfhipPlanGetStage(op, s, &v));
        items += v.numItems;
        /* the bundle table of forward complex stages: a partition of the item list into runs of <= 4 */
        if (v.bundleBegin) {
          uint32_t prev = 0;
          for (uint64_t b_ = 0; b_ <= v.numBundles; ++b_) {
            uint32_t const f = v.bundleBegin[b_] & 0x7fffffffu;
            if ((b_ == 0 && f != 0) || (b_ > 0 && (f <= prev || f - prev > 4)) || f > v.numItems) return 7;
            prev = f;
          }
          if (prev != v.numItems) return 7;
        }
        for (uint64_t r = 0; r < v.numReduce; ++r) {
          BfhipReduceView rv;
          memset(&rv, 0, sizeof rv);
          rv.structSize = sizeof rv;
          CHECK(bfhipPlanGetReduce(op, s, r, &rv));
        }
      }
      BfhipStats st;
      memset(&st, 0, sizeof st);
      st.structSize = sizeof st;
      CHECK(bfhipGetStats(op, &st));
      if (!items || st.numRows != (m ? m : n) || st.numCols != n) return 3;
      /* a row shard through the options (square operators only carry topRowBlock for it) */
      bfhipFree(&op);
      if (!m) {
        /* (round 5: shards carry the plan of their adjoint too -- the transposed task list pruned by reachability from the shard's
         *  rows; the packed flag falls back to it on a shard) */
        o.rowBlockBegin = 1; o.rowBlockEnd = 3; o.flags = BFHIP_FLAG_PLAN_ONLY | (flags & BFHIP_FLAG_ADJOINT ? BFHIP_FLAG_ADJOINT_PACKED : 0);
        CHECK(bfhipCompileDesc(d, &o, &op));
        if (flags & BFHIP_FLAG_ADJOINT) {
          memset(&info, 0, sizeof info);
          info.structSize = sizeof info;
          CHECK(bfhipPlanGetInfo(op, &info));
          if (!info.numStagesT || info.reserved) return 6;      /* a shared-leaf adjoint plan, not a packed one */
        }
        bfhipFree(&op);
        o.flags = flags;
        /* row ranges: the library's own cuts for 3 ranks, each compiled (liveness pruning), plus an unclean range */
        uint64_t cuts[4], loads[3];
        CHECK(bfhipRowPartition(d, 3, cuts, loads));
        if (cuts[0] != 0 || cuts[3] != n || !loads[0]) return 4;
        o.rowBlockBegin = o.rowBlockEnd = 0;
        for (int r = 0; r < 3; ++r) {
          o.rowBegin = cuts[r]; o.rowEnd = cuts[r + 1];
          CHECK(bfhipCompileDesc(d, &o, &op));
          if (bfhipGetNumRows(op) != cuts[r + 1] - cuts[r]) return 5;
          bfhipFree(&op);
        }
        o.rowBegin = 3; o.rowEnd = n - 5;
        CHECK(bfhipCompileDesc(d, &o, &op));
        bfhipFree(&op);
        o.rowBegin = o.rowEnd = 0;
      }
    }
  }
  bfhipHelm2LayoutFree(&lay);
  return 0;
}

/* the native layout of a streamed butterfly (rank model), compiled plan-only, forward and transposed */
static int driveStreamerLayout(void) {
  enum { N = 3000 };
  double *pts = malloc(3 * N * sizeof *pts);
  for (int i = 0; i < N; ++i) {           /* Fibonacci sphere */
    double const x = 1 - 2.0 * (i + 0.5) / N, r = sqrt(1 - x * x), th = 3.141592653589793 * (sqrt(5.0) - 1) * i;
    pts[3 * i] = x; pts[3 * i + 1] = r * cos(th); pts[3 * i + 2] = r * sin(th);
  }
  uint32_t depth = 0;
  CHECK(bfhipStreamerOctreeDepth(pts, N, &depth));
  uint64_t bands[8] = {30, 64, 90, 110, 12, 140, 160, 182};     /* (a band narrower than minNumCols takes the skinny path, src/fac.c:649-676) */
  BfhipStreamerSpec spec;
  memset(&spec, 0, sizeof spec);
  spec.structSize = sizeof spec; spec.colDepth = 3; spec.wmax = 28.0; spec.bandColumns = bands;
  for (uint64_t maxCols = 0; maxCols <= 300; maxCols += 300) {
    spec.maxCols = maxCols;
    BfhipStreamerLayout *lay = NULL;
    CHECK(bfhipStreamerLayoutCreate(pts, N, &spec, &lay));
    BfhipStreamerStats st;
    memset(&st, 0, sizeof st);
    st.structSize = sizeof st;
    CHECK(bfhipStreamerLayoutGetStats(lay, &st));
    if (st.numRows != N || !st.denseReal || !bfhipStreamerLayoutGetPerm(lay)) return 20;
    BfhipOptions o;
    memset(&o, 0, sizeof o);
    o.structSize = sizeof o; o.device = -1; o.flags = BFHIP_FLAG_PLAN_ONLY | BFHIP_FLAG_ADJOINT; o.demoteToF32 = 1;
    BfhipOperator *op = NULL;
    CHECK(bfhipCompileDesc(bfhipStreamerLayoutGetDesc(lay), &o, &op));
    if (bfhipGetNumRows(op) != N || bfhipGetNumCols(op) != st.numCols) return 21;
    bfhipFree(&op);
    uint64_t *sub = malloc(bfhipStreamerLayoutGetDesc(lay)->numNodes * 8);
    CHECK(bfhipDescSubtreeLeafElems(bfhipStreamerLayoutGetDesc(lay), sub));
    if (sub[bfhipStreamerLayoutGetDesc(lay)->root] * 8 != st.leafBytes) return 22;
    free(sub);
    bfhipStreamerLayoutFree(&lay);
  }
  free(pts);
  return 0;
}

/* a small reference-style object graph with every node type the walker accepts */
static double *randv(size_t count) {
  double *v = malloc(count * sizeof *v);
  for (size_t i = 0; i < count; ++i) v[i] = (double)rand() / RAND_MAX - 0.5;
  return v;
}
static BfMat *dense(size_t m, size_t n) { return bfMatDenseComplexNewFromPtr(m, n, randv(2 * m * n), 2); }

static int driveGraph(void) {
  BfMat *d1[2] = {dense(5, 7), dense(6, 4)};
  BfMat *f1 = bfMatBlockDiagNewFromBlocks(2, d1);                                   /* 11 x 11 */
  size_t const ro[3] = {0, 4, 9}, co[3] = {0, 5, 11}, ri[3] = {0, 1, 1}, ci[3] = {0, 0, 1};
  BfMat *c0[3] = {dense(4, 5), dense(5, 5), dense(5, 6)};
  BfMat *f0 = bfMatBlockCooNewFromArrays(2, 2, 3, ro, co, ri, ci, c0);              /* 9 x 11 */
  BfMat *fs[2] = {f0, f1};
  BfMat *prod = bfMatProductNewFromFactors(2, fs);                                  /* 9 x 11 */
  size_t const cr[2] = {0, 3}, cc[2] = {1, 5};
  double const cv[4] = {0.5, -1.0, 2.0, 0.25};
  BfMat *terms[2] = {dense(6, 6), bfMatCooComplexNewFromArrays(6, 6, 2, cr, cc, cv)};
  BfMat *sum = bfMatSumNewFromTerms(2, terms);
  BfMat *grid[4] = {prod, dense(9, 6), dense(6, 11), sum};
  size_t const gro[3] = {0, 9, 15}, gco[3] = {0, 11, 17};
  BfMat *A = bfMatBlockDenseNewFromBlocks(2, 2, gro, gco, grid);                    /* 15 x 17 */
  if (!A) return 10;
  for (unsigned flags = BFHIP_FLAG_PLAN_ONLY; flags <= (BFHIP_FLAG_PLAN_ONLY | BFHIP_FLAG_ADJOINT); flags += BFHIP_FLAG_ADJOINT) {
    BfhipOptions o;
    memset(&o, 0, sizeof o);
    o.structSize = sizeof o; o.device = -1; o.flags = flags; o.maxRhs = 3;
    BfhipOperator *op = NULL;
    CHECK(bfhipCompile(A, &o, &op));
    if (bfhipGetNumRows(op) != 15 || bfhipGetNumCols(op) != 17) return 11;
    double *packed = malloc(bfhipNumBytes(op) * 4 + 4096);
    BfhipPlanInfo info;
    memset(&info, 0, sizeof info);
    info.structSize = sizeof info;
    CHECK(bfhipPlanGetInfo(op, &info));
    free(packed);
    packed = malloc((size_t)info.arenaElems * 16 + 16);
    CHECK(bfhipPlanPackArena(op, packed));
    free(packed);
    bfhipFree(&op);
  }
  /* refusals: NULL graph, a node with a NULL vtable */
  BfhipOperator *op = NULL;
  if (bfhipCompile(NULL, NULL, &op) == 0) return 12;
  BfAbiMat bogus;
  memset(&bogus, 0, sizeof bogus);
  if (bfhipCompile(&bogus, NULL, &op) == 0) return 13;
  bfMatDelete(&A);
  return 0;
}

/* malformed public descriptors must be refused, not walked (DFS stack sized by nodes
 * while a child can be listed many times; childBegin never validated) */
static int driveBadDescs(void) {
  BfhipOptions o;
  memset(&o, 0, sizeof o);
  o.structSize = sizeof o; o.device = -1; o.flags = BFHIP_FLAG_PLAN_ONLY;
  BfhipOperator *op = NULL;
  /* (1) legal but nasty: a BLOCK root listing the same 1 x 1 leaf 40 times */
  {
    enum { K = 40 };
    uint8_t kind[2] = {BFHIP_NODE_BLOCK, BFHIP_NODE_DENSE};
    uint64_t rows[2] = {K, 1}, cols[2] = {K, 1}, cb[3] = {0, K, K}, cn[K], r0[K], c0[K];
    for (int i = 0; i < K; ++i) { cn[i] = 1; r0[i] = (uint64_t)i; c0[i] = (uint64_t)i; }
    BfhipDesc d;
    memset(&d, 0, sizeof d);
    d.structSize = sizeof d; d.dtype = BFHIP_F64; d.numNodes = 2; d.root = 0;
    d.kind = kind; d.rows = rows; d.cols = cols; d.childBegin = cb; d.childNode = cn; d.childRow0 = r0; d.childCol0 = c0;
    CHECK(bfhipCompileDesc(&d, &o, &op));
    if (bfhipGetNumRows(op) != K) return 20;
    bfhipFree(&op);
    /* (2) childBegin not monotone */
    uint64_t cbBad[3] = {0, K, 3};
    d.childBegin = cbBad;
    if (bfhipCompileDesc(&d, &o, &op) != 1 /* BF_ERROR_INVALID_ARGUMENTS */) return 21;
    /* (3) childBegin[0] != 0 */
    uint64_t cbBad2[3] = {2, K, K};
    d.childBegin = cbBad2;
    if (bfhipCompileDesc(&d, &o, &op) != 1) return 22;
    /* (4) an interior entry beyond the child arrays */
    uint64_t cbBad3[3] = {0, K + 5, K};
    d.childBegin = cbBad3;
    if (bfhipCompileDesc(&d, &o, &op) != 1) return 23;
  }
  return 0;
}

int main(void) {
  uint64_t const n = 6000, m = 2500;
  double *pts = malloc(n * 16), *tgt = malloc(m * 16), *rnd = malloc(3000 * 16);
  for (uint64_t i = 0; i < n; ++i) { double t = 2 * M_PI * i / n; pts[2 * i] = cos(t); pts[2 * i + 1] = 0.4 * sin(t); }
  for (uint64_t i = 0; i < m; ++i) { double t = 2 * M_PI * i / m; tgt[2 * i] = 1.9 + 0.8 * cos(t); tgt[2 * i + 1] = 0.3 + 0.6 * sin(t); }
  unsigned long long s = 88172645463325252ull;
  for (uint64_t i = 0; i < 6000; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; rnd[i] = (double)(s >> 11) / 9007199254740992.0; }
  int rc = drive(pts, n, NULL, 0, 150.0);
  if (!rc) rc = drive(pts, n, tgt, m, 90.0);
  if (!rc) rc = drive(rnd, 3000, NULL, 0, 60.0);
  if (!rc) rc = driveGraph();
  if (!rc) rc = driveStreamerLayout();
  if (!rc) rc = driveBadDescs();
  /* error paths */
  BfhipHelm2Layout *lay = NULL;
  double zeros[16] = {0};
  if (!rc && bfhipHelm2LayoutCreate(zeros, 8, 1.0, &lay) == 0) rc = 4;
  if (!rc && bfhipHelm2LayoutCreate(pts, 1, 1.0, &lay) == 0) rc = 5;
  free(pts); free(tgt); free(rnd);
  printf(rc ? "FAILED %d\n" : "host code clean\n", rc);
  return rc;
}

/* ---- device layer stand-ins: plan-only paths must never reach them ------------------------------- */
#define STUB(name) int name() { fprintf(stderr, "device layer reached: %s\n", #name); abort(); }
/* releasing nothing is fine (bfhipFree on a plan-only operator); releasing something is not */
void bfdevFree(void *p) { if (p) { fprintf(stderr, "device layer reached: bfdevFree(non-NULL)\n"); abort(); } }
void bfdevEventDestroy(void *e) { if (e) { fprintf(stderr, "device layer reached: bfdevEventDestroy(non-NULL)\n"); abort(); } }
STUB(bfdevBuildEval)
STUB(bfdevBuildGemm)
STUB(bfdevBuildJacobi)
STUB(bfdevBuildPack)
STUB(bfdevBuildQrcp)
STUB(bfdevQrcpFits)
STUB(bfdevEventCreate)
STUB(bfdevEventElapsed)
STUB(bfdevEventRecord)
STUB(bfdevEventSync)
STUB(bfdevScalePermute)
STUB(bfdevGetDevice)
STUB(bfdevHostAllocPinned)
void bfdevHostFreePinned(void *p) { if (p) { fprintf(stderr, "device layer reached: bfdevHostFreePinned(non-NULL)\n"); abort(); } }
STUB(bfdevGmresDot)
STUB(bfdevGmresDots)
STUB(bfdevGmresDotsFinish)
STUB(bfdevGmresProject)
STUB(bfdevGmresFinish)
STUB(bfdevGmresMgsStep)
STUB(bfdevGmresResidual)
STUB(bfdevGmresUpdate)
STUB(bfdevHelm2Dense)
STUB(bfdevLaunchReduce)
STUB(bfdevLaunchFlow)
STUB(bfdevFlowGrid)
STUB(bfdevMemsetAsync)
STUB(bfdevLaunchStage)
STUB(bfdevMalloc)
STUB(bfdevMemFree)
STUB(bfdevMemcpyD2DAsync)
STUB(bfdevMemcpyD2H)
STUB(bfdevMemcpyD2HAsync)
STUB(bfdevMemcpyH2D)
STUB(bfdevMemcpyH2DAsync)
STUB(bfdevMemset)
STUB(bfdevSetDevice)
STUB(bfdevPointerKind)
STUB(bfdevMemcpyAnyAsync)
STUB(bfdevHostRegister)
STUB(bfdevHostUnregister)
/* the sharded step lives in bfhip_shard.hip (device layer): the shim's sharded branch is not reached on a plan-only operator */
size_t bfhipShardedGetNumRows(const BfhipSharded *sh) { (void)sh; fprintf(stderr, "device layer reached: bfhipShardedGetNumRows\n"); abort(); }
size_t bfhipShardedGetNumCols(const BfhipSharded *sh) { (void)sh; fprintf(stderr, "device layer reached: bfhipShardedGetNumCols\n"); abort(); }
STUB(bfhipShardedApplyHost)
STUB(bfhipShardedOperator)
void bfhipShardedFree(BfhipSharded **p) { if (p && *p) { fprintf(stderr, "device layer reached: bfhipShardedFree(non-NULL)\n"); abort(); } }
STUB(bfdevSync)
STUB(bfdevSynthFill)
