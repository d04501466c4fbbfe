/* examples/sharded_apply.c -- the multi-GPU step of include/bfhip.h used from plain C.
 *
 * One process per GPU.  With N ranks the id of rank 0's communicator must reach the others; this
 * example ships it through a file (argv[3]) so that it runs under any launcher:
 *
 *     for r in 0 1 2 3; do ./sharded_apply 4 $r /tmp/bfhip.id & done; wait
 *
 * With one rank (`./sharded_apply 1 0`) it exercises the same code path -- the RCCL communicator, the
 * in-place all-gather / all-reduce, the segment reordering -- on a single GPU, which is what the GPU
 * test-suite runs (tests/test_gpu_parity.py::test_sharded_apply_from_plain_c).
 *
 * The operand is a small block matrix of dense leaves with seeded synthetic values (leafData == NULL);
 * every rank compiles only the block rows it owns (BfhipOptions.rowBlockBegin/End wants a contiguous
 * run, so ownership here is by runs of block rows) and the result is checked against the unsharded
 * operator on rank 0's GPU; a third pass deals the ranks contiguous row RANGES (bfhipRowPartition +
 * BfhipOptions.rowBegin/rowEnd).  Every pass also runs the ADJOINT of the step (bfhipShardedApplyTransposeDevice: the rank's
 * A_r^T on its entries of v, one all-reduce) against the unsharded A^T v, and a last pass solves a square system with
 * bfhipShardedSolveGMRESDevice next to bfhipSolveGMRESOptsDevice.  Exit code 0 iff the three forward passes reproduce the
 * unsharded apply bit for bit, the adjoints agree to rounding and the two solves take the same iterations. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "bfhip.h"

#define CHECK(c) do { int rc_ = (c); if (rc_) { fprintf(stderr, "%s -> %s (%s)\n", #c, bfhipErrorString(rc_), bfhipLastErrorMessage()); return 1; } } while (0)

/* the few HIP runtime calls a host needs, without the HIP headers */
extern int hipMalloc(void **p, size_t bytes);
extern int hipFree(void *p);
extern int hipMemcpy(void *dst, void const *src, size_t bytes, int kind);   /* 1: H2D, 2: D2H */
extern int hipDeviceSynchronize(void);
extern int hipSetDevice(int device);
extern int hipGetDeviceCount(int *count);

enum { NBR = 6, NBC = 3 };

int main(int argc, char **argv) {
  int const nranks = argc > 1 ? atoi(argv[1]) : 1, rank = argc > 2 ? atoi(argv[2]) : 0;
  char const *idPath = argc > 3 ? argv[3] : NULL;
  if (nranks < 1 || nranks > NBR || rank < 0 || rank >= nranks || (nranks > 1 && !idPath)) { fprintf(stderr, "usage: %s nranks rank [idfile]\n", argv[0]); return 2; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) || ndev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
  int const device = rank % ndev;
  if (hipSetDevice(device)) return 2;

  /* descriptor: NBR x NBC grid of dense complex leaves, ragged sizes */
  uint64_t const rowsOf[NBR] = {37, 64, 5, 130, 48, 71}, colsOf[NBC] = {90, 33, 120};
  uint64_t n = 0, m = 0, ro[NBR + 1] = {0}, co[NBC + 1] = {0};
  for (int i = 0; i < NBR; ++i) { ro[i + 1] = ro[i] + rowsOf[i]; }
  for (int j = 0; j < NBC; ++j) { co[j + 1] = co[j] + colsOf[j]; }
  m = ro[NBR]; n = co[NBC];
  enum { NN = NBR * NBC + 1 };
  uint8_t kind[NN]; uint64_t rows[NN], cols[NN], cb[NN + 1], cn[NBR * NBC], r0[NBR * NBC], c0[NBR * NBC], trb[NBR * NBC];
  for (int i = 0; i < NBR; ++i) for (int j = 0; j < NBC; ++j) {
    int const k = i * NBC + j;
    kind[k] = BFHIP_NODE_DENSE; rows[k] = rowsOf[i]; cols[k] = colsOf[j]; cb[k] = 0;
    cn[k] = (uint64_t)k; r0[k] = ro[i]; c0[k] = co[j]; trb[k] = (uint64_t)i;
  }
  kind[NN - 1] = BFHIP_NODE_BLOCK; rows[NN - 1] = m; cols[NN - 1] = n; cb[NN - 1] = 0; cb[NN] = NBR * NBC;
  BfhipDesc d;
  memset(&d, 0, sizeof d);
  d.structSize = sizeof d; d.dtype = BFHIP_C128; d.numNodes = NN; d.root = NN - 1;
  d.kind = kind; d.rows = rows; d.cols = cols; d.childBegin = cb; d.childNode = cn; d.childRow0 = r0; d.childCol0 = c0; d.topRowBlock = trb;

  /* ownership: contiguous runs of block rows, as even as they come */
  uint32_t owner[NBR];
  for (int i = 0; i < NBR; ++i) owner[i] = (uint32_t)(i * nranks / NBR);
  uint64_t myBegin = NBR, myEnd = 0;
  for (int i = 0; i < NBR; ++i) if ((int)owner[i] == rank) { if ((uint64_t)i < myBegin) myBegin = (uint64_t)i; myEnd = (uint64_t)i + 1; }

  BfhipOptions o;
  memset(&o, 0, sizeof o);
  o.structSize = sizeof o; o.device = device; o.seed = 42; o.maxRhs = 2;
  o.flags = BFHIP_FLAG_ADJOINT;          /* every operator below also carries the plan of its transpose */
  BfhipOperator *full = NULL, *mine = NULL;
  CHECK(bfhipCompileDesc(&d, &o, &full));
  o.rowBlockBegin = myBegin; o.rowBlockEnd = myEnd;
  CHECK(bfhipCompileDesc(&d, &o, &mine));

  /* communicator */
  char id[128];
  if (rank == 0) {
    CHECK(bfhipCommGetUniqueId(id));
    if (idPath) { FILE *fp = fopen(idPath, "wb"); if (!fp || fwrite(id, 1, 128, fp) != 128) return 3; fclose(fp); }
  } else {
    FILE *fp = NULL;
    for (int t = 0; t < 600 && !fp; ++t) { fp = fopen(idPath, "rb"); if (!fp) usleep(100000); }
    if (!fp) { fprintf(stderr, "no communicator id at %s\n", idPath); return 3; }
    usleep(200000);
    if (fread(id, 1, 128, fp) != 128) return 3;
    fclose(fp);
  }
  BfhipComm *comm = NULL;
  CHECK(bfhipCommInitRank(id, nranks, rank, device, &comm));

  size_t const nrhs = 2;
  double *x = malloc(n * nrhs * 16), *yRef = malloc(m * nrhs * 16), *y = malloc(m * nrhs * 16);
  for (uint64_t i = 0; i < 2 * n * nrhs; ++i) x[i] = sin(0.37 * (double)i) + 0.01 * (double)(i % 7);
  void *dX = NULL, *dY = NULL;
  if (hipMalloc(&dX, n * nrhs * 16) || hipMalloc(&dY, m * nrhs * 16) || hipMemcpy(dX, x, n * nrhs * 16, 1)) return 4;
  CHECK(bfhipApplyDevice(full, dX, nrhs, dY, NULL));
  if (hipDeviceSynchronize() || hipMemcpy(yRef, dY, m * nrhs * 16, 2)) return 4;

  /* the adjoint's reference: z = A^T v on the whole operator */
  double *v = malloc(m * nrhs * 16), *zRef = malloc(n * nrhs * 16), *z = malloc(n * nrhs * 16);
  for (uint64_t i = 0; i < 2 * m * nrhs; ++i) v[i] = cos(0.11 * (double)i) - 0.02 * (double)(i % 5);
  void *dV = NULL, *dZ = NULL;
  if (hipMalloc(&dV, m * nrhs * 16) || hipMalloc(&dZ, n * nrhs * 16) || hipMemcpy(dV, v, m * nrhs * 16, 1)) return 4;
  CHECK(bfhipApplyTransposeDevice(full, dV, nrhs, dZ, NULL));
  if (hipDeviceSynchronize() || hipMemcpy(zRef, dZ, n * nrhs * 16, 2)) return 4;
  int badT = 0;
#define CHECK_ADJOINT(what) do { \
    CHECK(bfhipShardedApplyTransposeDevice(sh, dV, nrhs, dZ, NULL)); \
    if (hipDeviceSynchronize() || hipMemcpy(z, dZ, n * nrhs * 16, 2)) return 4; \
    double num_ = 0, den_ = 0; \
    for (uint64_t i_ = 0; i_ < 2 * n * nrhs; ++i_) { num_ += (z[i_] - zRef[i_]) * (z[i_] - zRef[i_]); den_ += zRef[i_] * zRef[i_]; } \
    int const ok_ = sqrt(num_ / den_) <= 1e-13; \
    if (!ok_) badT = 1; \
    printf("rank %d/%d %s adjoint: rel. diff. to the unsharded A^T v %.1e, %s\n", rank, nranks, what, sqrt(num_ / den_), ok_ ? "adjoint-equal" : "MISMATCH"); \
  } while (0)

  int bad = 0;
  BfhipShardSpec spec;
  memset(&spec, 0, sizeof spec);
  spec.structSize = sizeof spec; spec.mode = BFHIP_SHARD_ROWS; spec.numRowsGlobal = m;
  spec.numSegments = NBR; spec.segRows = rowsOf; spec.segOwner = owner;
  BfhipSharded *sh = NULL;
  CHECK(bfhipShardedCreate(mine, comm, &spec, (uint32_t)nrhs, &sh));
  for (int rep = 0; rep < 2; ++rep) CHECK(bfhipShardedApplyDevice(sh, dX, nrhs, dY, NULL));
  double localMs = 0, collMs = 0;
  CHECK(bfhipShardedLastTimes(sh, &localMs, &collMs));
  if (hipMemcpy(y, dY, m * nrhs * 16, 2)) return 4;
  if (memcmp(y, yRef, m * nrhs * 16) != 0) { fprintf(stderr, "rank %d: rows mode differs from the unsharded apply\n", rank); bad = 1; }
  printf("rank %d/%d rows mode: local %.3f ms, collective %.3f ms, %s\n", rank, nranks, localMs, collMs, bad ? "MISMATCH" : "bit-identical");
  CHECK_ADJOINT("rows mode");
  bfhipShardedFree(&sh);

  /* blocks mode: a full-length partial y per rank; here simply "my block rows at their original
   * offsets, zeros elsewhere" (any split of the top-level blocks works the same way) */
  uint64_t cb2[NN + 1], cn2[NBR * NBC], r02[NBR * NBC], c02[NBR * NBC], cnt = 0;
  for (int i = 0; i < NBR; ++i) if ((int)owner[i] == rank) for (int j = 0; j < NBC; ++j) { cn2[cnt] = (uint64_t)(i * NBC + j); r02[cnt] = ro[i]; c02[cnt] = co[j]; ++cnt; }
  memset(cb2, 0, sizeof cb2);
  cb2[NN] = cnt;
  BfhipDesc d2 = d;
  d2.childBegin = cb2; d2.childNode = cn2; d2.childRow0 = r02; d2.childCol0 = c02; d2.topRowBlock = NULL;
  o.rowBlockBegin = o.rowBlockEnd = 0;
  BfhipOperator *part = NULL;
  CHECK(bfhipCompileDesc(&d2, &o, &part));
  spec.mode = BFHIP_SHARD_BLOCKS; spec.numSegments = 0; spec.segRows = NULL; spec.segOwner = NULL;
  CHECK(bfhipShardedCreate(part, comm, &spec, (uint32_t)nrhs, &sh));
  CHECK(bfhipShardedApplyDevice(sh, dX, nrhs, dY, NULL));
  CHECK(bfhipShardedLastTimes(sh, &localMs, &collMs));
  if (hipMemcpy(y, dY, m * nrhs * 16, 2)) return 4;
  int bad2 = memcmp(y, yRef, m * nrhs * 16) != 0;
  if (bad2) fprintf(stderr, "rank %d: blocks mode differs from the unsharded apply\n", rank);
  printf("rank %d/%d blocks mode: local %.3f ms, collective %.3f ms, %s\n", rank, nranks, localMs, collMs, bad2 ? "MISMATCH" : "bit-identical");
  CHECK_ADJOINT("blocks mode");
  bfhipShardedFree(&sh);

  /* row RANGES (round 3): balanced cuts from the library, one contiguous range per rank; the rank's operator keeps
   * exactly what its rows depend on (BfhipOptions.rowBegin / rowEnd) and produces them bit for bit */
  uint64_t cuts[NBR + 1], segRows2[NBR];
  uint32_t segOwner2[NBR];
  CHECK(bfhipRowPartition(&d, (uint32_t)nranks, cuts, NULL));
  for (int r = 0; r < nranks; ++r) { segRows2[r] = cuts[r + 1] - cuts[r]; segOwner2[r] = (uint32_t)r; }
  o.rowBlockBegin = o.rowBlockEnd = 0;
  o.rowBegin = cuts[rank]; o.rowEnd = cuts[rank + 1];
  BfhipOperator *range = NULL;
  CHECK(bfhipCompileDesc(&d, &o, &range));
  o.rowBegin = o.rowEnd = 0;
  spec.mode = BFHIP_SHARD_ROWS; spec.numSegments = (uint32_t)nranks; spec.segRows = segRows2; spec.segOwner = segOwner2;
  CHECK(bfhipShardedCreate(range, comm, &spec, (uint32_t)nrhs, &sh));
  CHECK(bfhipShardedApplyDevice(sh, dX, nrhs, dY, NULL));
  CHECK(bfhipShardedLastTimes(sh, &localMs, &collMs));
  if (hipMemcpy(y, dY, m * nrhs * 16, 2)) return 4;
  int bad3 = memcmp(y, yRef, m * nrhs * 16) != 0;
  if (bad3) fprintf(stderr, "rank %d: row-range mode differs from the unsharded apply\n", rank);
  printf("rank %d/%d row ranges [%llu, %llu): local %.3f ms, collective %.3f ms, %s\n", rank, nranks, (unsigned long long)cuts[rank],
         (unsigned long long)cuts[rank + 1], localMs, collMs, bad3 ? "MISMATCH" : "bit-identical");
  CHECK_ADJOINT("row ranges");
  bfhipShardedFree(&sh);
  bfhipFree(&range);
  bad |= bad3;

  /* GMRES over the sharded step (bfhipShardedSolveGMRESDevice) next to the one-GPU solver on the same square operand:
   * 4 x 4 blocks of 60 x 60, identities on the diagonal plus seeded dense leaves scaled to a contraction -- built as a
   * product-free block matrix whose diagonal blocks are BLOCK nodes {Identity, Dense}.  Ranks own runs of block rows. */
  int badG = 0;
  {
    enum { Q = 4, B = 60, NG = Q * Q + Q + 1, NCH = Q * (Q + 1) };
    uint8_t gk[NG]; uint64_t gr[NG], gc[NG], gcb[NG + 1], gcn[NCH], gr0[NCH], gc0[NCH], gtrb[NCH];
    uint64_t nc = 0;
    /* nodes 0 .. Q*Q-1: dense leaves (i, j); Q*Q .. Q*Q+Q-1: the identities of the diagonal; the root last */
    for (int i = 0; i < NG - 1; ++i) { gk[i] = i < Q * Q ? BFHIP_NODE_DENSE : BFHIP_NODE_IDENTITY; gr[i] = B; gc[i] = B; gcb[i] = 0; }
    int const rootG = NG - 1;
    gk[rootG] = BFHIP_NODE_BLOCK; gr[rootG] = Q * B; gc[rootG] = Q * B; gcb[rootG] = 0;
    for (int i = 0; i < Q; ++i) {
      for (int j = 0; j < Q; ++j) { gcn[nc] = (uint64_t)(i * Q + j); gr0[nc] = (uint64_t)i * B; gc0[nc] = (uint64_t)j * B; gtrb[nc] = (uint64_t)i; ++nc; }
      gcn[nc] = (uint64_t)(Q * Q + i); gr0[nc] = (uint64_t)i * B; gc0[nc] = (uint64_t)i * B; gtrb[nc] = (uint64_t)i; ++nc;
    }
    gcb[NG] = nc;
    BfhipDesc dg;
    memset(&dg, 0, sizeof dg);
    dg.structSize = sizeof dg; dg.dtype = BFHIP_C128; dg.numNodes = NG; dg.root = (uint64_t)rootG;
    dg.kind = gk; dg.rows = gr; dg.cols = gc; dg.childBegin = gcb; dg.childNode = gcn; dg.childRow0 = gr0; dg.childCol0 = gc0; dg.topRowBlock = gtrb;
    uint64_t const ng = (uint64_t)Q * B;
    uint32_t ownerG[Q];
    uint64_t rowsG[Q], gb = Q, ge = 0;
    for (int i = 0; i < Q; ++i) { ownerG[i] = (uint32_t)(i * (nranks < Q ? nranks : Q) / Q); rowsG[i] = B; if ((int)ownerG[i] == rank) { if ((uint64_t)i < gb) gb = (uint64_t)i; ge = (uint64_t)i + 1; } }
    BfhipOptions og = o;
    og.flags = 0; og.rowBegin = og.rowEnd = 0; og.rowBlockBegin = og.rowBlockEnd = 0;
    BfhipOperator *gfull = NULL, *gmine = NULL;
    CHECK(bfhipCompileDesc(&dg, &og, &gfull));
    if (gb < ge) { og.rowBlockBegin = gb; og.rowBlockEnd = ge; CHECK(bfhipCompileDesc(&dg, &og, &gmine)); }
    if (gmine) {
      memset(&spec, 0, sizeof spec);
      spec.structSize = sizeof spec; spec.mode = BFHIP_SHARD_ROWS; spec.numRowsGlobal = ng; spec.numSegments = Q; spec.segRows = rowsG; spec.segOwner = ownerG;
      CHECK(bfhipShardedCreate(gmine, comm, &spec, 1, &sh));
      double *b = malloc(ng * 16), *xa = malloc(ng * 16), *xb = malloc(ng * 16);
      for (uint64_t i = 0; i < 2 * ng; ++i) b[i] = sin(0.3 * (double)i + 1.0);
      void *dB = NULL, *dXa = NULL, *dXb = NULL;
      if (hipMalloc(&dB, ng * 16) || hipMalloc(&dXa, ng * 16) || hipMalloc(&dXb, ng * 16) || hipMemcpy(dB, b, ng * 16, 1)) return 4;
      BfhipGmresOptions go;
      memset(&go, 0, sizeof go);
      go.structSize = sizeof go; go.orthogonalization = BFHIP_GMRES_ORTH_MGS; go.tol = 1e-10; go.maxNumIter = 60;
      size_t ita = 0, itb = 0; double ra = 0, rb = 0;
      CHECK(bfhipSolveGMRESOptsDevice(gfull, &go, dB, 1, NULL, &ita, &ra, dXa, NULL));
      CHECK(bfhipShardedSolveGMRESDevice(sh, &go, dB, 1, NULL, &itb, &rb, dXb, NULL));
      if (hipMemcpy(xa, dXa, ng * 16, 2) || hipMemcpy(xb, dXb, ng * 16, 2)) return 4;
      double num = 0, den = 0;
      for (uint64_t i = 0; i < 2 * ng; ++i) { num += (xa[i] - xb[i]) * (xa[i] - xb[i]); den += xa[i] * xa[i]; }
      badG = ita != itb || sqrt(num / den) > 1e-10;
      printf("rank %d/%d gmres: %zu iterations (one GPU: %zu), residual %.2e, solutions differ by %.1e, %s\n", rank, nranks, itb, ita, rb, sqrt(num / den), badG ? "MISMATCH" : "gmres-equal");
      hipFree(dB); hipFree(dXa); hipFree(dXb); free(b); free(xa); free(xb);
      bfhipShardedFree(&sh);
    }
    bfhipFree(&gmine); bfhipFree(&gfull);
  }

  /* error behaviour: a spec that does not cover the operator is refused, nothing aborts */
  spec.mode = BFHIP_SHARD_ROWS; spec.numSegments = NBR - 1; spec.segRows = rowsOf; spec.segOwner = owner;
  if (bfhipShardedCreate(mine, comm, &spec, 1, &sh) == 0) { fprintf(stderr, "short segment list was accepted\n"); bad = 1; }

  bfhipCommDestroy(&comm);
  bfhipFree(&part); bfhipFree(&mine); bfhipFree(&full);
  hipFree(dX); hipFree(dY); hipFree(dV); hipFree(dZ);
  free(x); free(y); free(yRef); free(v); free(z); free(zRef);
  return bad || bad2 || badT || badG;
}
