/* bfhip_plan.c -- the "compile" step: expression IR -> per-stage
 * CSR-of-blocks layout resident in HBM.
 *
 * What the reference does on every bfMatMul call -- recurse through
 * Product / BlockDense / BlockDiag / BlockCoo, allocating a view per row
 * range and a fresh result per leaf -- is done here
 * once, ahead of time:
 *
 * 1. Scheduling (ALAP).  Every leaf product becomes a task
 *    (stage, leaf, input segment, output segment).  A Block contributes its
 *    children in the stage its result is due; a Product F0*...*F_{L-1} gives
 *    F0 the due stage and each later factor an earlier one, through one
 *    intermediate vector per factor boundary (mat_product.c:225-238 keeps the
 *    same intermediates, heap-allocated per call).  With as-late-as-possible
 *    placement all contributions to one vector land in the same stage, so
 *    stages are the only synchronization (kernel boundaries).
 *
 * 2. Row groups.  Tasks of a stage that write exactly the same rows of the
 *    same vector (the <= 4 blocks of a BlockCoo block row,
 *    fac_helm2.c:277-318; a BlockDiag block on its own) form a group; a group
 *    is cut into items of <= 64 row slots, one wavefront each, and the item
 *    accumulates all its blocks in registers -- the reference's AddInplace
 *    pass (mat_block_coo.c:413) disappears and each output row has one owner.
 *
 * 3. Overlapping groups.  Where differently shaped groups hit the same rows
 *    (the final accumulation into y: evaluation factors of many products at
 *    different tree depths plus dense near-field leaves,
 *    mat_block_dense.c:541-563) each group writes a private slot and one
 *    deterministic reduce pass sums the slots per row in a fixed order.
 *
 * 4. Packing.  Each (item, block) "piece" is stored column-major (rows x
 *    cols, rows padded to the 16-byte lane granule) and pieces are laid out in
 *    the order the kernel consumes them, so a wavefront streams its bytes
 *    linearly with 16-byte-per-lane coalesced loads.
 */
#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"

#include <stdlib.h>
#include <string.h>

typedef struct Task {
  uint32_t stage;
  uint32_t outBuf, inBuf;
  uint64_t leaf;
  uint64_t outOff, inOff;
  uint64_t rows, cols;
  uint64_t sub0;       /* offset of this task inside its leaf along the contracted dimension (long leaves are cut) */
  uint64_t rsub0;      /* first row of the leaf this task produces (row-range shards may trim a leaf that straddles a cut) */
  uint64_t seq;        /* emission order: fixes the summation order */
  uint32_t cls, pad;   /* transposed plans: 1 = rows of a tall leaf (>= 16 lane units: the 16-column kernel reads whole 256-byte runs) */
} Task;

/* A leaf is contracted in sub-tasks of at most this many columns (forward) / rows (transposed):
 * granularity for splitting one row group over several items when a single item would stream too
 * much.  Forward: a multiple of the piece width (xcap), so the pieces are what they would have been. */
#define BF_TASK_SPAN 1024u

typedef struct Buf {
  uint64_t len;
  uint64_t arenaOff;   /* element offset in the vector arena (temps only) */
  int32_t stage;       /* stage in which it is written (-1: X) */
} Buf;

typedef struct Builder {
  BfIr const *ir;
  Task *tasks; uint64_t numTasks, capTasks;
  Buf *bufs; uint64_t numBufs, capBufs;
  int err;
} Builder;

static int pushTask(Builder *b, Task const *t) {
  if (b->numTasks == b->capTasks) {
    uint64_t cap = b->capTasks ? b->capTasks * 2 : 4096;
    Task *p = realloc(b->tasks, cap * sizeof(Task));
    if (!p) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (tasks)");
    b->tasks = p; b->capTasks = cap;
  }
  b->tasks[b->numTasks] = *t;
  b->tasks[b->numTasks].seq = b->numTasks;
  ++b->numTasks;
  return 0;
}
static int newBuf(Builder *b, uint64_t len, int32_t stage, uint32_t *id) {
  if (b->numBufs == b->capBufs) {
    uint64_t cap = b->capBufs ? b->capBufs * 2 : 256;
    Buf *p = realloc(b->bufs, cap * sizeof(Buf));
    if (!p) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (buffers)");
    b->bufs = p; b->capBufs = cap;
  }
  if (b->numBufs >= 0xffffffffu) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "too many intermediate vectors");
  b->bufs[b->numBufs].len = len;
  b->bufs[b->numBufs].arenaOff = 0;
  b->bufs[b->numBufs].stage = stage;
  *id = (uint32_t)b->numBufs++;
  return 0;
}

static int emit(Builder *b, uint64_t node, uint32_t inBuf, uint64_t inOff, uint32_t outBuf, uint64_t outOff, int32_t stageEnd) {
  BfIr const *ir = b->ir;
  int rc;
  switch (ir->kind[node]) {
  case BFHIP_NODE_DENSE:
  case BFHIP_NODE_IDENTITY: {
    Task t;
    memset(&t, 0, sizeof t);
    t.stage = (uint32_t)stageEnd; t.leaf = node;
    t.inBuf = inBuf; t.outBuf = outBuf; t.outOff = outOff;
    t.rows = ir->rows[node];
    uint64_t const span = ir->kind[node] == BFHIP_NODE_IDENTITY ? ir->cols[node] : BF_TASK_SPAN;
    for (uint64_t c0 = 0; c0 < ir->cols[node]; c0 += span) {
      t.sub0 = c0; t.inOff = inOff + c0;
      t.cols = ir->cols[node] - c0 < span ? ir->cols[node] - c0 : span;
      if ((rc = pushTask(b, &t))) return rc;
    }
    return 0;
  }
  case BFHIP_NODE_BLOCK:
    for (uint64_t c = ir->childBegin[node]; c < ir->childBegin[node + 1]; ++c)
      if ((rc = emit(b, ir->childNode[c], inBuf, inOff + ir->childCol0[c], outBuf, outOff + ir->childRow0[c], stageEnd))) return rc;
    return 0;
  case BFHIP_NODE_PRODUCT: {
    uint64_t cb = ir->childBegin[node], ce = ir->childBegin[node + 1];
    uint32_t curOut = outBuf; uint64_t curOutOff = outOff;
    int32_t se = stageEnd;
    for (uint64_t c = cb; c < ce; ++c) {
      uint64_t f = ir->childNode[c];
      uint32_t in = inBuf; uint64_t inO = inOff;
      int32_t seNext = se - (int32_t)ir->depth[f];
      if (c + 1 < ce) {
        if ((rc = newBuf(b, ir->cols[f], seNext, &in))) return rc;
        inO = 0;
      }
      if ((rc = emit(b, f, in, inO, curOut, curOutOff, se))) return rc;
      se = seNext; curOut = in; curOutOff = 0;
    }
    return 0;
  }
  }
  return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unknown node kind");
}

/* A^T: rows and columns trade places.  A Block child at (r0, c0) reads the
 * input at +r0 and writes the output at +c0; (F0 ... F_{L-1})^T = F_{L-1}^T ...
 * F0^T, so F_{L-1}^T is applied last (mat_product.c:314-345 walks the factors
 * in the same order for RmulVec). */
static int emitT(Builder *b, uint64_t node, uint32_t inBuf, uint64_t inOff, uint32_t outBuf, uint64_t outOff, int32_t stageEnd) {
  BfIr const *ir = b->ir;
  int rc;
  switch (ir->kind[node]) {
  case BFHIP_NODE_DENSE:
  case BFHIP_NODE_IDENTITY: {
    Task t;
    memset(&t, 0, sizeof t);
    t.stage = (uint32_t)stageEnd; t.leaf = node;
    t.inBuf = inBuf; t.outBuf = outBuf; t.outOff = outOff;
    t.rows = ir->cols[node];
    uint64_t const span = ir->kind[node] == BFHIP_NODE_IDENTITY ? ir->rows[node] : BF_TASK_SPAN;
    for (uint64_t r0 = 0; r0 < ir->rows[node]; r0 += span) {      /* contracted dimension of A^T = rows of A */
      t.sub0 = r0; t.inOff = inOff + r0;
      t.cols = ir->rows[node] - r0 < span ? ir->rows[node] - r0 : span;
      if ((rc = pushTask(b, &t))) return rc;
    }
    return 0;
  }
  case BFHIP_NODE_BLOCK:
    for (uint64_t c = ir->childBegin[node]; c < ir->childBegin[node + 1]; ++c)
      if ((rc = emitT(b, ir->childNode[c], inBuf, inOff + ir->childRow0[c], outBuf, outOff + ir->childCol0[c], stageEnd))) return rc;
    return 0;
  case BFHIP_NODE_PRODUCT: {
    uint64_t cb = ir->childBegin[node], ce = ir->childBegin[node + 1];
    uint32_t curOut = outBuf; uint64_t curOutOff = outOff;
    int32_t se = stageEnd;
    for (uint64_t c = ce; c-- > cb;) {          /* F_{L-1}^T is due last */
      uint64_t f = ir->childNode[c];
      uint32_t in = inBuf; uint64_t inO = inOff;
      int32_t seNext = se - (int32_t)ir->depth[f];
      if (c > cb) {
        if ((rc = newBuf(b, ir->rows[f], seNext, &in))) return rc;   /* input of F_c^T has rows(F_c) entries */
        inO = 0;
      }
      if ((rc = emitT(b, f, in, inO, curOut, curOutOff, se))) return rc;
      se = seNext; curOut = in; curOutOff = 0;
    }
    return 0;
  }
  }
  return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unknown node kind");
}


/* ---- row-range shards: backward liveness over the task list --------------------------------------
 * A rank of a multi-GPU job owns the output rows [rowBegin, rowEnd).  Every top-level block row of the
 * reference is computed independently from the full x (bfMatBlockDenseMul, src/mat_block_dense.c:534-566), a
 * nested BlockDense row block likewise (src/fac_helm2.c:814-858), and inside a product the evaluation factor is
 * block diagonal over target nodes (src/fac_helm2.c:403-509) fed by radix-4 BlockCoo stages (:277-318): which
 * leaves a set of output rows needs follows from the task list alone.  Walk the stages backwards: a task is live
 * if any row it writes is live; a live task makes the input elements it reads live.  Dead tasks are dropped
 * before grouping, so a shard holds exactly the leaves its rows depend on (the factors near the source end of a
 * butterfly are needed by every target row and are replicated), and every surviving row group is the one the
 * whole operator would have: the rows a rank produces are bit for bit those of the one-GPU plan. */
static int cmpTask(void const *pa, void const *pb);

typedef struct LiveMap {
  uint64_t *bits;
  uint64_t *base;        /* per buffer: first bit */
  uint64_t numBits;
} LiveMap;

static void liveFree(LiveMap *lm) { free(lm->bits); free(lm->base); memset(lm, 0, sizeof *lm); }

static int liveInit(LiveMap *lm, Buf const *bufs, uint64_t numBufs) {
  memset(lm, 0, sizeof *lm);
  lm->base = malloc((numBufs + 1) * 8);
  if (!lm->base) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (liveness)");
  uint64_t acc = 0;
  for (uint64_t i = 0; i < numBufs; ++i) { lm->base[i] = acc; acc += (bufs[i].len + 63) / 64 * 64; }
  lm->base[numBufs] = acc;
  lm->numBits = acc;
  lm->bits = calloc(acc / 64 + 1, 8);
  if (!lm->bits) { liveFree(lm); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (liveness)"); }
  return 0;
}
static void liveMark(LiveMap *lm, uint32_t buf, uint64_t off, uint64_t len) {
  if (!len) return;
  uint64_t a = lm->base[buf] + off, b = a + len;       /* [a, b) */
  uint64_t wa = a / 64, wb = (b - 1) / 64;
  uint64_t const ma = ~0ull << (a % 64), mb = ~0ull >> (63 - (b - 1) % 64);
  if (wa == wb) { lm->bits[wa] |= ma & mb; return; }
  lm->bits[wa] |= ma;
  for (uint64_t w = wa + 1; w < wb; ++w) lm->bits[w] = ~0ull;
  lm->bits[wb] |= mb;
}
static int liveAny(LiveMap const *lm, uint32_t buf, uint64_t off, uint64_t len) {
  if (!len) return 0;
  uint64_t a = lm->base[buf] + off, b = a + len;
  uint64_t wa = a / 64, wb = (b - 1) / 64;
  uint64_t const ma = ~0ull << (a % 64), mb = ~0ull >> (63 - (b - 1) % 64);
  if (wa == wb) return (lm->bits[wa] & ma & mb) != 0;
  if (lm->bits[wa] & ma) return 1;
  for (uint64_t w = wa + 1; w < wb; ++w) if (lm->bits[w]) return 1;
  return (lm->bits[wb] & mb) != 0;
}
static int liveBit(LiveMap const *lm, uint32_t buf, uint64_t off) {
  uint64_t a = lm->base[buf] + off;
  return (int)((lm->bits[a / 64] >> (a % 64)) & 1u);
}

/* Tasks sorted by stage (ascending).  Marks liveness from rows [rowBegin, rowEnd) of buffer `by`, trims the tasks
 * that write y to that range, drops dead tasks (the array is compacted in place, order kept) and shifts y to start at
 * rowBegin.  Buffers nothing live writes get length 0. */
static int pruneToRowRange(Builder *b, uint32_t bx, uint32_t by, int32_t S, uint64_t rowBegin, uint64_t rowEnd, LiveMap *lm) {
  int rc = liveInit(lm, b->bufs, b->numBufs);
  if (rc) return rc;
  liveMark(lm, by, rowBegin, rowEnd - rowBegin);
  uint8_t *keep = calloc(b->numTasks ? b->numTasks : 1, 1);
  uint8_t *written = calloc(b->numBufs, 1);
  if (!keep || !written) { free(keep); free(written); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (liveness)"); }
  for (uint64_t t = b->numTasks; t-- > 0;) {        /* stages descending: every reader of a buffer comes before its writers */
    Task *tk = &b->tasks[t];
    if (!liveAny(lm, tk->outBuf, tk->outOff, tk->rows)) continue;
    keep[t] = 1;
    written[tk->outBuf] = 1;
    if (tk->outBuf == by) {
      /* trim to the range: the leaf's rows [rsub0, rsub0 + rows) */
      uint64_t lo = tk->outOff > rowBegin ? tk->outOff : rowBegin;
      uint64_t hi = tk->outOff + tk->rows < rowEnd ? tk->outOff + tk->rows : rowEnd;
      tk->rsub0 += lo - tk->outOff;
      if (b->ir->kind[tk->leaf] == BFHIP_NODE_IDENTITY) { tk->inOff += lo - tk->outOff; tk->cols = hi - lo; }   /* an identity copies row i to row i */
      tk->outOff = lo - rowBegin;
      tk->rows = hi - lo;
    }
    if (tk->inBuf != bx) liveMark(lm, tk->inBuf, tk->inOff, tk->cols);
  }
  (void)S;
  uint64_t w = 0;
  for (uint64_t t = 0; t < b->numTasks; ++t) if (keep[t]) b->tasks[w++] = b->tasks[t];
  b->numTasks = w;
  for (uint64_t i = 2; i < b->numBufs; ++i) if (!written[i]) b->bufs[i].len = 0;
  /* y now holds the range only: shift its liveness bits to the front (gap fill below asks for them) */
  for (uint64_t r = 0; r < rowEnd - rowBegin; ++r) {
    uint64_t const a = lm->base[by] + r;
    lm->bits[a / 64] |= 1ull << (a % 64);
  }
  b->bufs[by].len = rowEnd - rowBegin;
  free(keep); free(written);
  return 0;
}


/* The same shard, transposed: z = A_r^T v_r, where A_r = rows [rowBegin, rowEnd) of A (what pruneToRowRange keeps) and v_r the
 * matching entries of v -- a rank's contribution to A^T v, full length; the ranks' contributions add up (ONE all-reduce,
 * bfhipShardedApplyTransposeDevice).  In the transposed task list data flows from v to z, so the forward plan's backward
 * liveness becomes forward reachability: walking the stages upwards, a task is kept if any element it reads has been written by
 * a kept task (or is an entry of v inside the range), and what it writes becomes reachable.  Tasks reading v are trimmed to the
 * range (the shard's input is compact: rows [rowBegin, rowEnd) only).  Every kept task's leaf is one the forward shard holds
 * (reachability from the range through a leaf <=> the forward liveness of that leaf), so the pieces are found in its arena.
 * On return `lm` marks what must hold a defined value: every element a kept task reads, and all of z -- the gap fill writes
 * zeros there where no kept task does. */
static int pruneToRowRangeT(Builder *b, uint32_t bx, uint32_t by, uint64_t rowBegin, uint64_t rowEnd, LiveMap *lm) {
  LiveMap reach;
  int rc = liveInit(&reach, b->bufs, b->numBufs);
  if (rc) return rc;
  if ((rc = liveInit(lm, b->bufs, b->numBufs))) { liveFree(&reach); return rc; }
  uint8_t *keep = calloc(b->numTasks ? b->numTasks : 1, 1);
  uint8_t *written = calloc(b->numBufs, 1);
  if (!keep || !written) { free(keep); free(written); liveFree(&reach); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (liveness)"); }
  for (uint64_t t = 0; t < b->numTasks; ++t) {       /* stages ascending: every writer of a buffer comes before its readers */
    Task *tk = &b->tasks[t];
    if (tk->inBuf == bx) {
      uint64_t const lo = tk->inOff > rowBegin ? tk->inOff : rowBegin;
      uint64_t const hi = tk->inOff + tk->cols < rowEnd ? tk->inOff + tk->cols : rowEnd;
      if (lo >= hi) continue;
      tk->sub0 += lo - tk->inOff;
      if (b->ir->kind[tk->leaf] == BFHIP_NODE_IDENTITY) { tk->outOff += lo - tk->inOff; tk->rows = hi - lo; }   /* an identity copies entry i to entry i */
      tk->inOff = lo - rowBegin;
      tk->cols = hi - lo;
    } else {
      if (!liveAny(&reach, tk->inBuf, tk->inOff, tk->cols)) continue;
      liveMark(lm, tk->inBuf, tk->inOff, tk->cols);
    }
    keep[t] = 1;
    written[tk->outBuf] = 1;
    liveMark(&reach, tk->outBuf, tk->outOff, tk->rows);
  }
  uint64_t w = 0;
  for (uint64_t t = 0; t < b->numTasks; ++t) if (keep[t]) b->tasks[w++] = b->tasks[t];
  b->numTasks = w;
  for (uint64_t i = 2; i < b->numBufs; ++i) if (!written[i]) b->bufs[i].len = 0;
  liveMark(lm, by, 0, b->bufs[by].len);
  b->bufs[bx].len = rowEnd - rowBegin;
  free(keep); free(written);
  liveFree(&reach);
  return 0;
}


/* ---- balanced row ranges for W ranks ---------------------------------------------------------------
 * cuts[0] = 0 < cuts[1] < ... < cuts[W] = rows: rank r owns rows [cuts[r], cuts[r + 1]).  A cut is only placed where
 * no leaf that writes y straddles it (a "clean" position: quadtree node boundaries in a fac_helm2 operand), so no leaf
 * is trimmed and every rank's rows are bit for bit the one-GPU result.  The load of a range is what pruneToRowRange
 * would keep for it: every task carries the hull [lo, hi) of the y rows that depend on it (propagated backwards through
 * the intermediates: min / max per element), and load(a, b) = sum of leaf elements of the tasks whose hull meets
 * [a, b) -- replication of the source-side factors included.  The largest load is minimised by bisection over a greedy
 * sweep (a contiguous partition). */
typedef struct PartTask { uint32_t lo, hi; uint64_t w; } PartTask;
static int cmpPartTask(void const *pa, void const *pb) {
  PartTask const *a = pa, *b = pb;
  if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
  return a->hi < b->hi ? -1 : (a->hi > b->hi);
}

/* largest clean cut index j > i0 (at most jMax) with load(cut[i0], cut[j]) <= L; 0 if even i0 + 1 exceeds L */
static uint64_t partAdvance(PartTask const *pt, uint64_t npt, uint64_t const *cut, uint64_t i0, uint64_t jMax, uint64_t L, uint64_t *loadOut) {
  uint64_t const a = cut[i0];
  uint64_t acc = 0, k = 0, best = 0, bestLoad = 0;
  for (uint64_t j = i0 + 1; j <= jMax; ++j) {
    uint64_t const bnd = cut[j];
    while (k < npt && pt[k].lo < bnd) { if (pt[k].hi > a) acc += pt[k].w; ++k; }
    if (acc > L) break;
    best = j; bestLoad = acc;
  }
  if (loadOut) *loadOut = bestLoad;
  return best;
}

int bfPlanRowPartition(BfIr const *ir, uint32_t world, uint64_t *cuts, uint64_t *loads) {
  if (!world) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "zero ranks");
  Builder b;
  memset(&b, 0, sizeof b);
  b.ir = ir;
  int rc = 0;
  uint32_t bx, by;
  uint64_t const root = ir->root, N = ir->rows[root];
  int32_t const S = (int32_t)ir->depth[root];
  uint32_t *lo = NULL, *hi = NULL;
  uint64_t *base = NULL, *cut = NULL;
  int32_t *interior = NULL;
  PartTask *pt = NULL;
  if (N >= 0xffffffffu) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator too tall for 32-bit row indices");
  if ((rc = newBuf(&b, ir->cols[root], -1, &bx))) goto done;
  if ((rc = newBuf(&b, N, S - 1, &by))) goto done;
  if ((rc = emit(&b, root, bx, 0, by, 0, S - 1))) goto done;
  qsort(b.tasks, b.numTasks, sizeof(Task), cmpTask);
  base = malloc((b.numBufs + 1) * 8);
  if (!base) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
  uint64_t tot = 0;
  for (uint64_t i = 0; i < b.numBufs; ++i) { base[i] = tot; if (i != bx) tot += b.bufs[i].len; }
  lo = malloc((tot + 1) * 4); hi = malloc((tot + 1) * 4);
  pt = malloc((b.numTasks + 1) * sizeof *pt);
  interior = calloc(N + 2, sizeof *interior);
  if (!lo || !hi || !pt || !interior) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (row partition)"); goto done; }
  for (uint64_t e = 0; e < tot; ++e) { lo[e] = 0xffffffffu; hi[e] = 0; }
  for (uint64_t r = 0; r < N; ++r) { lo[base[by] + r] = (uint32_t)r; hi[base[by] + r] = (uint32_t)r + 1; }
  uint64_t npt = 0, total = 0;
  for (uint64_t t = b.numTasks; t-- > 0;) {
    Task const *tk = &b.tasks[t];
    uint32_t L = 0xffffffffu, H = 0;
    uint64_t const o = base[tk->outBuf] + tk->outOff;
    for (uint64_t r = 0; r < tk->rows; ++r) { if (lo[o + r] < L) L = lo[o + r]; if (hi[o + r] > H) H = hi[o + r]; }
    if (L == 0xffffffffu) continue;                      /* nothing reads what it writes */
    if (tk->inBuf != bx) {
      uint64_t const i0 = base[tk->inBuf] + tk->inOff;
      for (uint64_t c = 0; c < tk->cols; ++c) { if (L < lo[i0 + c]) lo[i0 + c] = L; if (H > hi[i0 + c]) hi[i0 + c] = H; }
    }
    if (tk->outBuf == by && tk->rows > 1) { interior[tk->outOff + 1] += 1; interior[tk->outOff + tk->rows] -= 1; }
    if (ir->kind[tk->leaf] != BFHIP_NODE_DENSE) continue;
    pt[npt].lo = L; pt[npt].hi = H; pt[npt].w = tk->rows * tk->cols;
    total += pt[npt].w;
    ++npt;
  }
  qsort(pt, npt, sizeof *pt, cmpPartTask);
  /* clean cut positions, 0 and N included */
  uint64_t ncut = 0;
  cut = malloc((N + 2) * 8);
  if (!cut) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
  int32_t depthIn = 0;
  for (uint64_t p = 0; p <= N; ++p) { depthIn += interior[p]; if (p == 0 || p == N || depthIn == 0) cut[ncut++] = p; }
  if (ncut - 1 < world) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "the operator's %llu rows can be cut at %llu places only: not enough for %u ranks", (unsigned long long)N, (unsigned long long)(ncut - 2), world); goto done; }
  /* smallest L for which the greedy sweep covers all rows with `world` non-empty ranges */
  uint64_t Llo = total / world, Lhi = 0;
  for (uint64_t k = 0; k < npt; ++k) Lhi += pt[k].w;           /* = total: one rank holding everything */
  for (int iter = 0; iter < 48 && Lhi - Llo > total / 100000 + 1; ++iter) {
    uint64_t const L = Llo + (Lhi - Llo) / 2;
    uint64_t i = 0;
    int ok = 1;
    for (uint32_t r = 0; r < world && ok; ++r) {
      uint64_t const jMax = (ncut - 1) - (world - 1 - r);     /* leave a clean interval for every later rank */
      uint64_t const j = partAdvance(pt, npt, cut, i, jMax, L, NULL);
      if (!j) ok = 0; else i = j;
    }
    if (ok && i == ncut - 1) Lhi = L; else Llo = L + 1;
  }
  {
    uint64_t i = 0;
    cuts[0] = 0;
    for (uint32_t r = 0; r < world; ++r) {
      uint64_t const jMax = (ncut - 1) - (world - 1 - r);
      uint64_t ld = 0;
      uint64_t j = r + 1 == world ? ncut - 1 : partAdvance(pt, npt, cut, i, jMax, Lhi, NULL);
      if (!j) j = i + 1;                                         /* a single leaf heavier than the target: it still goes somewhere */
      (void)partAdvance(pt, npt, cut, i, j, UINT64_MAX, &ld);    /* the load of [cut[i], cut[j]) */
      cuts[r + 1] = cut[j];
      if (loads) loads[r] = ld;
      i = j;
    }
  }
done:
  free(lo); free(hi); free(base); free(cut); free(interior); free(pt);
  free(b.tasks); free(b.bufs);
  return rc;
}

static int cmpFwdPiece(void const *pa, void const *pb) {
  BfFwdPiece const *a = pa, *b = pb;
  if (a->node != b->node) return a->node < b->node ? -1 : 1;
  if (a->col0 != b->col0) return a->col0 < b->col0 ? -1 : 1;
  return a->row0 < b->row0 ? -1 : (a->row0 > b->row0);
}

int bfPlanFwdPieces(BfPlan const *plan, BfFwdPiece **out, uint64_t *count) {
  uint64_t n = 0;
  for (uint64_t s = 0; s < plan->numStages; ++s) n += plan->stages[s].numPieces;
  BfFwdPiece *t = malloc((n ? n : 1) * sizeof *t);
  if (!t) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (forward piece table)");
  uint64_t k = 0;
  for (uint64_t s = 0; s < plan->numStages; ++s) {
    BfStage const *st = &plan->stages[s];
    if (st->numPieces && (!st->items || !st->pieces || !st->pieceSrc)) { free(t); return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: forward plan mirrors already released"); }
    for (uint64_t i = 0; i < st->numItems; ++i) {
      BfDevItem const *it = &st->items[i];
      uint32_t mr = it->mrFlags & 0xffffu;
      for (uint32_t p = 0; p < it->numPieces; ++p) {
        BfDevPiece const *pc = &st->pieces[it->pieceBegin + p];
        if (pc->flags & BF_PIECE_IDENTITY) continue;
        BfPieceSrc const *src = &st->pieceSrc[it->pieceBegin + p];
        t[k].node = src->node; t[k].dataOff = pc->dataOff; t[k].row0 = src->row0; t[k].mr = mr;
        t[k].mrPad = (mr + plan->epl - 1) / plan->epl * plan->epl; t[k].col0 = src->col0; t[k].ncols = pc->ncols;
        t[k].rowMajor = (pc->flags & BF_PIECE_ROWMAJOR) != 0; t[k].ldr = pc->ld;
        ++k;
      }
    }
  }
  qsort(t, k, sizeof *t, cmpFwdPiece);
  *out = t;
  *count = k;
  return 0;
}

/* Transposed items are sorted big first; the leading ones that stream at least BF_COOP_BYTES get a workgroup each
 * when they average at least BF_COOP_PIECES pieces: a block column of a streamed butterfly is hundreds of few-row pieces,
 * one dependent load each.  Items of a few tall pieces (fac_helm2) are 3 - 6 % slower shared than alone. */
uint64_t bfPlanCountCoop(BfDevItem const *items, BfDevPiece const *pieces, uint64_t numItems, uint32_t elemSize) {
  uint64_t k = 0, np = 0;
  for (; k < numItems; ++k) {
    uint64_t cols = 0;
    for (uint32_t i = 0; i < items[k].numPieces; ++i)
      if (!(pieces[items[k].pieceBegin + i].flags & BF_PIECE_IDENTITY)) cols += pieces[items[k].pieceBegin + i].ncols;
    if ((items[k].mrFlags & 0xffffu) * cols * elemSize < BF_COOP_BYTES) break;
    np += items[k].numPieces;
  }
  return np >= BF_COOP_PIECES * k ? k : 0;
}

/* first table entry of (node, column piece containing col) */
static uint64_t findFwd(BfFwdPiece const *t, uint64_t n, uint64_t node, uint32_t colPiece0) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = (lo + hi) / 2;
    if (t[mid].node < node || (t[mid].node == node && t[mid].col0 < colPiece0)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

/* sort tasks by (stage, outBuf, outOff, rows, class, seq) */
static int cmpTask(void const *pa, void const *pb) {
  Task const *a = pa, *b = pb;
  if (a->stage != b->stage) return a->stage < b->stage ? -1 : 1;
  if (a->outBuf != b->outBuf) return a->outBuf < b->outBuf ? -1 : 1;
  if (a->outOff != b->outOff) return a->outOff < b->outOff ? -1 : 1;
  if (a->rows != b->rows) return a->rows < b->rows ? -1 : 1;
  if (a->cls != b->cls) return a->cls < b->cls ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq);
}

typedef struct Group {
  uint64_t taskBegin, taskEnd;   /* range in the sorted task array */
  uint64_t outOff, rows;
  uint32_t outBuf;
  uint64_t slotOff;              /* vector-arena offset of its private slot, if reduced */
  int reduced;
  uint64_t colsSum, piecesPerChunk;
  uint32_t cls;                  /* transposed: 1 = 16-column items (tall leaves) */
  uint32_t chunkRows;            /* rows per item of this group */
} Group;

typedef struct ItemTmp {
  uint64_t cost;
  uint64_t sortCost;             /* what the list is ordered by: cost, or its bucket when neighbours are to share their input (cmpItemCost) */
  uint64_t inKey;                /* (input buffer, first input element) of the group's first task; 0 unless sortCost is a bucket */
  uint64_t group;
  uint32_t rowBegin, rows;       /* chunk inside the group */
  uint32_t small, narrow;        /* small: runs four to a wavefront, sorted behind everything else; narrow: 16-column transposed item, sorted first */
  uint64_t sig;                  /* RHS-block operators: hash of what the item's group reads (bundles, below) */
  uint64_t bundle;               /* its bundle's place in the list */
} ItemTmp;

static int cmpItemCost(void const *pa, void const *pb) {
  ItemTmp const *a = pa, *b = pb;
  if (a->small != b->small) return a->small < b->small ? -1 : 1;
  if (a->narrow != b->narrow) return a->narrow > b->narrow ? -1 : 1;
  if (a->sortCost != b->sortCost) return a->sortCost > b->sortCost ? -1 : 1;      /* big first */
  if (a->inKey != b->inKey) return a->inKey < b->inKey ? -1 : 1;
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  if (a->group != b->group) return a->group < b->group ? -1 : 1;
  return a->rowBegin < b->rowBegin ? -1 : (a->rowBegin > b->rowBegin);
}
/* RHS-block operators (complex128): items that read exactly the same input rows -- the row chunks of a group and the sibling groups
 * of a radix-4 stage (src/fac_helm2.c:277-318) -- are brought together (cmpItemSig), cut into runs of up to BF_BUNDLE_ITEMS, and the
 * list is ordered run by run (cmpItemBundle), big first by cost bucket: list neighbours run side by side on one XCD (the 64-RHS
 * kernel's workgroup mapping) and walk the same X panel at the same time, so that a row of it is fetched into that L2 once.  Round 4
 * ordered by the FIRST input only: groups that share their first block but not the others sat in between.  Same box, N = 262144, 64
 * RHS: 30.3 - 30.6 ms against 30.8 - 31.0.  (The runs are also what the A/B kernel of bfhip_stage_mfma.h, BF_MF_BUNDLES, takes as
 * workgroups.) */
static int cmpItemSig(void const *pa, void const *pb) {
  ItemTmp const *a = pa, *b = pb;
  if (a->sig != b->sig) return a->sig < b->sig ? -1 : 1;
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  if (a->group != b->group) return a->group < b->group ? -1 : 1;
  return a->rowBegin < b->rowBegin ? -1 : (a->rowBegin > b->rowBegin);
}
static int cmpItemBundle(void const *pa, void const *pb) {
  ItemTmp const *a = pa, *b = pb;
  if (a->bundle != b->bundle) return a->bundle < b->bundle ? -1 : 1;
  if (a->bundle == UINT64_MAX) return cmpItemCost(pa, pb);      /* the items that found no full bundle: big first, by input inside a cost bucket */
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  if (a->group != b->group) return a->group < b->group ? -1 : 1;
  return a->rowBegin < b->rowBegin ? -1 : (a->rowBegin > b->rowBegin);
}
typedef struct BundleTmp { uint64_t sortCost, inKey, cost, first; } BundleTmp;
static int cmpBundle(void const *pa, void const *pb) {
  BundleTmp const *a = pa, *b = pb;
  if (a->sortCost != b->sortCost) return a->sortCost > b->sortCost ? -1 : 1;      /* big first, by cost bucket */
  if (a->inKey != b->inKey) return a->inKey < b->inKey ? -1 : 1;                  /* neighbours read neighbouring input */
  if (a->cost != b->cost) return a->cost > b->cost ? -1 : 1;
  return a->first < b->first ? -1 : (a->first > b->first);
}
static uint64_t costBucket(uint64_t c) {      /* quarter octaves */
  if (!c) c = 1;
  uint32_t const pbit = 63u - (uint32_t)__builtin_clzll(c);
  return (uint64_t)pbit * 4u + (pbit >= 2 ? (c >> (pbit - 2)) & 3u : 0u);
}
static int sameInputs(Task const *tasks, BfIr const *ir, uint64_t aBegin, uint64_t aEnd, uint64_t bBegin, uint64_t bEnd) {
  if (aEnd - aBegin != bEnd - bBegin) return 0;
  for (uint64_t i = 0; i < aEnd - aBegin; ++i) {
    Task const *a = &tasks[aBegin + i], *b = &tasks[bBegin + i];
    if (a->inBuf != b->inBuf || a->inOff != b->inOff || a->cols != b->cols ||
        (ir->kind[a->leaf] == BFHIP_NODE_IDENTITY) != (ir->kind[b->leaf] == BFHIP_NODE_IDENTITY)) return 0;
  }
  return 1;
}

static int cmpU64(void const *pa, void const *pb) {
  uint64_t a = *(uint64_t const *)pa, b = *(uint64_t const *)pb;
  return a < b ? -1 : a > b;
}
static uint64_t roundUp(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

/* Bundles of a forward complex128 stage, from its final item list: the workgroups of the 64-RHS matrix-core kernel, up to
 * BF_BUNDLE_ITEMS list neighbours each (bundleBegin[numBundles + 1]: first item, bit 31 = BF_BUNDLE_MIXED; the last entry = numItems).
 *   shared:  exactly BF_BUNDLE_ITEMS items of <= 32 rows and the same number of 16-row slabs whose pieces read the same input rows in
 *            the same order -- the wavefronts fetch every X tile once for all of them;
 *   mixed:   any other neighbours (taller items, zero fills, different inputs): four unrelated one-wavefront passes.
 * Derived from (items, pieces) alone, so that a loaded plan and a row shard get theirs the same way. */
static int sameBundleInputs(BfDevItem const *a, BfDevItem const *c, BfDevPiece const *pieces) {
  if (!a->numPieces || c->numPieces != a->numPieces || (a->mrFlags & 0xffffu) > 32u || (c->mrFlags & 0xffffu) > 32u ||
      ((c->mrFlags & 0xffffu) > 16u) != ((a->mrFlags & 0xffffu) > 16u)) return 0;
  for (uint32_t k = 0; k < a->numPieces; ++k) {
    BfDevPiece const *pa = &pieces[a->pieceBegin + k], *pc = &pieces[c->pieceBegin + k];
    if (pa->inOff != pc->inOff || pa->ncols != pc->ncols || ((pa->flags ^ pc->flags) & (BF_PIECE_IN_X | BF_PIECE_IDENTITY))) return 0;
  }
  return 1;
}
static int startsSharedBundle(BfDevItem const *items, BfDevPiece const *pieces, uint64_t numItems, uint64_t i) {
  if (BF_BUNDLE_ITEMS < 2 || i + BF_BUNDLE_ITEMS > numItems) return 0;
  for (uint64_t j = i + 1; j < i + BF_BUNDLE_ITEMS; ++j) if (!sameBundleInputs(&items[i], &items[j], pieces)) return 0;
  return 1;
}
int bfPlanBundles(BfDevItem const *items, BfDevPiece const *pieces, uint64_t numItems, uint32_t **out, uint64_t *numBundles) {
  if (numItems >= BF_BUNDLE_MIXED) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "stage too large for the bundle table");
  uint32_t *bb = malloc((numItems + 1) * 4);
  if (!bb) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (bundles)");
  uint64_t nb = 0;
  for (uint64_t i = 0; i < numItems;) {
    if (startsSharedBundle(items, pieces, numItems, i)) { bb[nb++] = (uint32_t)i; i += BF_BUNDLE_ITEMS; continue; }
    uint64_t j = i + 1;
    while (j < numItems && j - i < BF_BUNDLE_ITEMS && !startsSharedBundle(items, pieces, numItems, j)) ++j;      /* never into a shared bundle */
    bb[nb++] = (uint32_t)i | BF_BUNDLE_MIXED;
    i = j;
  }
  bb[nb] = (uint32_t)numItems;
  *out = bb; *numBundles = nb;
  return 0;
}

void bfPlanFree(BfPlan *plan) {
  if (!plan) return;
  for (uint64_t s = 0; s < plan->numStages && plan->stages; ++s) {
    BfStage *st = &plan->stages[s];
    free(st->items); free(st->pieces); free(st->pieceSrc); free(st->pieceBuf); free(st->itemBuf); free(st->bundleBegin);
    for (uint64_t r = 0; r < st->numReduce; ++r) {
      free(st->reduce[r].rowInterval); free(st->reduce[r].ivBegin); free(st->reduce[r].srcBias);
    }
    free(st->reduce);
  }
  free(plan->stages);
  free(plan->bufWriters);
  memset(plan, 0, sizeof *plan);
}

/* target upper bound on the leaf bytes one item (one wavefront) streams */
/* (measured round 3, A/B on one box: 16 / 32 / 64 / 128 KiB here, and 64 ... 1024 KiB for the contraction cap below, give
 * the same times at N = 65536 and on a 1/8 shard to 1 %; the headline is 3 - 5 % slower with caps below 1 MiB) */
#define BF_ITEM_BYTES (128u << 10)

/* vector-arena allocator: 4-element alignment keeps every segment 16-byte
 * aligned for all element types */
static uint64_t arenaAlloc(uint64_t *top, uint64_t len) {
  uint64_t off = roundUp(*top, 4);
  *top = off + len;
  return off;
}

int bfPlanBuild(BfIr const *ir, BfPlanOptions const *po, BfPlan *plan) {
  memset(plan, 0, sizeof *plan);
  Builder b;
  memset(&b, 0, sizeof b);
  b.ir = ir;
  int rc = 0;

  plan->dtype = po->storeDtype;
  plan->elemSize = plan->dtype == BFHIP_C128 ? 16 : (plan->dtype == BFHIP_F64 ? 8 : 4);
  plan->epl = 16 / plan->elemSize;
  plan->maxItemRows = 64 * plan->epl;
  int const T = po->fwdPieces != NULL;
  plan->transposed = T;
  uint32_t itemRows = po->itemRows ? po->itemRows : plan->maxItemRows;
  if (itemRows > plan->maxItemRows) itemRows = plan->maxItemRows;
  itemRows = (uint32_t)roundUp(itemRows, plan->epl);
  plan->xcap = po->xcap ? po->xcap : 256;
  if (T) {
    /* an item is 16 (or, for operands made of short leaves, 64) columns of A (one output each), whatever
     * the element size: the transposed kernel tiles a forward piece as 4 columns x 16 row units per load */
    itemRows = po->tCols > 16 ? po->tCols : 16;
    plan->maxItemRows = po->tCols ? itemRows : 64;      /* tCols == 0: chosen stage by stage below */
  }
  if (po->rowEnd > 0 && po->rowBlockEnd > 0) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "row range and row-block range are exclusive");
  LiveMap lm;
  memset(&lm, 0, sizeof lm);
  uint64_t *fullStageElems = NULL;

  uint32_t bx, by;
  uint64_t root = ir->root;
  int32_t S = (int32_t)ir->depth[root];
  uint64_t numRows = T ? ir->cols[root] : ir->rows[root];      /* rows of the operator this plan applies */
  uint64_t numColsOp = T ? ir->rows[root] : ir->cols[root];

  /* ---- 1. schedule ------------------------------------------------------ */
  if ((rc = newBuf(&b, numColsOp, -1, &bx))) goto fail;          /* buffer 0 = X */
  if (po->rowBlockEnd > 0) {
    /* row sharding: keep block rows [begin, end) of a BLOCK root */
    if (ir->kind[root] != BFHIP_NODE_BLOCK || !ir->topRowBlock) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "row sharding needs a block-matrix root"); goto fail; }
    uint64_t cb = ir->childBegin[root], ce = ir->childBegin[root + 1];
    uint64_t nb = 0;
    for (uint64_t c = cb; c < ce; ++c) if (ir->topRowBlock[c - cb] + 1 > nb) nb = ir->topRowBlock[c - cb] + 1;
    uint64_t *lo = malloc((nb + 1) * 8), *hi = calloc(nb + 1, 8), *base = calloc(nb + 1, 8);
    if (!lo || !hi || !base) { free(lo); free(hi); free(base); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
    for (uint64_t i = 0; i <= nb; ++i) lo[i] = UINT64_MAX;
    for (uint64_t c = cb; c < ce; ++c) {
      uint64_t rb = ir->topRowBlock[c - cb], ch = ir->childNode[c];
      if (ir->childRow0[c] < lo[rb]) lo[rb] = ir->childRow0[c];
      if (ir->childRow0[c] + ir->rows[ch] > hi[rb]) hi[rb] = ir->childRow0[c] + ir->rows[ch];
    }
    uint64_t acc = 0;
    for (uint64_t i = 0; i < nb; ++i) {
      base[i] = acc;
      if (i >= po->rowBlockBegin && i < po->rowBlockEnd && lo[i] != UINT64_MAX) acc += hi[i] - lo[i];
    }
    if (T) { numColsOp = acc; b.bufs[bx].len = acc; }      /* the shard's A^T takes the kept block rows' entries of v, compacted, and yields all of z */
    else numRows = acc;
    /* depth of the kept part */
    uint32_t dep = 1;
    for (uint64_t c = cb; c < ce; ++c) {
      uint64_t rb = ir->topRowBlock[c - cb];
      if (rb >= po->rowBlockBegin && rb < po->rowBlockEnd && ir->depth[ir->childNode[c]] > dep) dep = ir->depth[ir->childNode[c]];
    }
    S = (int32_t)dep;
    if ((rc = newBuf(&b, numRows, S - 1, &by))) { free(lo); free(hi); free(base); goto fail; }
    for (uint64_t c = cb; c < ce && !rc; ++c) {
      uint64_t rb = ir->topRowBlock[c - cb];
      if (rb < po->rowBlockBegin || rb >= po->rowBlockEnd) continue;
      if (T) rc = emitT(&b, ir->childNode[c], bx, ir->childRow0[c] - lo[rb] + base[rb], by, ir->childCol0[c], S - 1);
      else rc = emit(&b, ir->childNode[c], bx, ir->childCol0[c], by, ir->childRow0[c] - lo[rb] + base[rb], S - 1);
    }
    free(lo); free(hi); free(base);
    if (rc) goto fail;
  } else {
    if ((rc = newBuf(&b, numRows, S - 1, &by))) goto fail;        /* buffer 1 = Y */
    if ((rc = (T ? emitT : emit)(&b, root, bx, 0, by, 0, S - 1))) goto fail;
  }
  plan->numRows = numRows;
  plan->numCols = numColsOp;
  plan->numStages = (uint64_t)S;
  plan->stages = calloc((size_t)S, sizeof(BfStage));
  if (!plan->stages) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
  plan->numBufs = b.numBufs;
  plan->bufWriters = calloc(b.numBufs + 1, 4);
  if (!plan->bufWriters) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }

  /* invariant: every buffer is written in exactly one stage */
  for (uint64_t t = 0; t < b.numTasks; ++t) {
    Task const *tk = &b.tasks[t];
    if (b.bufs[tk->outBuf].stage != (int32_t)tk->stage || (int32_t)tk->stage >= S ||
        b.bufs[tk->inBuf].stage >= (int32_t)tk->stage) {
      rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: stage schedule inconsistent");
      goto fail;
    }
  }

  if (T && !po->tCols)     /* tall leaves get 16-column items of their own (below) */
    for (uint64_t t = 0; t < b.numTasks; ++t)
      b.tasks[t].cls = ir->kind[b.tasks[t].leaf] == BFHIP_NODE_DENSE && ir->rows[b.tasks[t].leaf] >= 16u * plan->epl;
  qsort(b.tasks, b.numTasks, sizeof(Task), cmpTask);

  /* how long contractions are cut (below) depends on the size of the stage: a shard cuts like the whole operator,
   * so that its rows are bit for bit the whole operator's */
  fullStageElems = calloc((size_t)S + 1, 8);
  if (!fullStageElems) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
  for (uint64_t t = 0; t < b.numTasks; ++t) fullStageElems[b.tasks[t].stage] += b.tasks[t].rows * b.tasks[t].cols;

  if (po->rowEnd > 0) {
    /* row-range shard: keep what rows [rowBegin, rowEnd) of y depend on (see pruneToRowRange) */
    uint64_t const opRows = T ? numColsOp : numRows;       /* rows of A */
    if (po->rowBegin >= po->rowEnd || po->rowEnd > opRows) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "row range [%llu, %llu) outside the operator's %llu rows", (unsigned long long)po->rowBegin, (unsigned long long)po->rowEnd, (unsigned long long)opRows); goto fail; }
    if (T) {
      if ((rc = pruneToRowRangeT(&b, bx, by, po->rowBegin, po->rowEnd, &lm))) goto fail;
      qsort(b.tasks, b.numTasks, sizeof(Task), cmpTask);
      numColsOp = po->rowEnd - po->rowBegin;
      plan->numCols = numColsOp;
    } else {
      if ((rc = pruneToRowRange(&b, bx, by, S, po->rowBegin, po->rowEnd, &lm))) goto fail;
      qsort(b.tasks, b.numTasks, sizeof(Task), cmpTask);      /* trimmed tasks may have moved */
      numRows = po->rowEnd - po->rowBegin;
      plan->numRows = numRows;
    }
  }

  /* vector arena: intermediates first */
  uint64_t top = 0;
  for (uint64_t i = 2; i < b.numBufs; ++i) b.bufs[i].arenaOff = arenaAlloc(&top, b.bufs[i].len);

  /* ---- 2..4 per stage --------------------------------------------------- */
  uint64_t arenaTop = 0;       /* leaf arena, elements */
  uint64_t tBegin = 0;
  uint8_t *bufRead = calloc(b.numBufs, 1);
  if (!bufRead) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
  for (int32_t s = 0; s < S; ++s) {
    BfStage *st = &plan->stages[s];
    uint64_t tEnd = tBegin;
    while (tEnd < b.numTasks && b.tasks[tEnd].stage == (uint32_t)s) ++tEnd;

    /* groups */
    uint64_t numGroups = 0, capGroups = 1024;
    Group *groups = malloc(capGroups * sizeof(Group));
    if (!groups) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); free(bufRead); goto fail; }
    /* One item (one wavefront) streams chunkRows x (sum of its group's contraction lengths) elements.
     * Row groups whose sum is long -- tall leaves in a transposed plan, a block column of hundreds of
     * leaves (fac_streamer's W factors), very wide leaves -- are cut into several groups over the same
     * output rows; the planner's overlap machinery below then gives each a private slot and one
     * deterministic reduce.  The cap is 1 MiB per item, less when the stage is too small to fill the GPU
     * otherwise (>= ~4096 items wanted), never below the regular item size. */
    /* The transposed kernel tiles a piece as (R row lanes x 64 / R columns) per load: R = 16 (16-column items) is
     * exact for pieces whose height is a multiple of 16 lane units (fac_helm2: 16 - 64 rows of complex128; the tall
     * leaves of a streamed butterfly's first factors) and reads whole 256-byte runs; pieces of 1 - 15 units (the
     * streamer's later factors) leave most of 16 row lanes idle, and R = 4 (64-column items) wastes only the last
     * 4-unit step.  Tall leaves (>= 16 units) always get 16-column items; for the others each stage picks by the lane
     * slots either tiling would spend on them.  A block column that holds both kinds becomes two groups over the same
     * outputs (slots + reduce, like any overlap) and the stage two launches (BfStage.numNarrow). */
    if (T && !po->tCols) {
      uint64_t slots16 = 0, slots4 = 0;
      for (uint64_t t = tBegin; t < tEnd; ++t) {
        Task const *tk = &b.tasks[t];
        if (ir->kind[tk->leaf] != BFHIP_NODE_DENSE || tk->sub0 || tk->cls) continue;
        for (uint64_t k = findFwd(po->fwdPieces, po->numFwdPieces, tk->leaf, 0); k < po->numFwdPieces && po->fwdPieces[k].node == tk->leaf; ++k) {
          uint64_t const u = (po->fwdPieces[k].mr + plan->epl - 1) / plan->epl, w = po->fwdPieces[k].ncols;
          slots16 += (u + 15) / 16 * 16 * w;
          slots4 += (u + 3) / 4 * 4 * w;
        }
      }
      itemRows = (10 * slots4 < 8 * slots16) ? 64 : 16;
    }
    uint64_t const stageElems = fullStageElems[s];
    uint64_t const itemBytes = BF_ITEM_BYTES;
    uint64_t capBytes = 1u << 20;
    uint64_t const itemsWanted = po->itemsWanted ? po->itemsWanted : 4096;
    if (stageElems * plan->elemSize / itemsWanted < capBytes) capBytes = stageElems * plan->elemSize / itemsWanted;
    if (capBytes < itemBytes) capBytes = itemBytes;
    uint64_t const floorRowsCap = T ? itemRows : (uint64_t)(po->minChunkRows ? po->minChunkRows : 16) * plan->epl;
    uint64_t capColsCls[2];
    capColsCls[0] = capBytes / (floorRowsCap * plan->elemSize);
    capColsCls[1] = capBytes / (16u * plan->elemSize);
    for (int c = 0; c < 2; ++c) if (capColsCls[c] < BF_TASK_SPAN) capColsCls[c] = BF_TASK_SPAN;
    for (uint64_t t = tBegin; t < tEnd;) {
      uint64_t u = t + 1;
      uint64_t span = b.tasks[t].cols;
      uint64_t const capCols = capColsCls[b.tasks[t].cls];
      while (u < tEnd && b.tasks[u].outBuf == b.tasks[t].outBuf && b.tasks[u].outOff == b.tasks[t].outOff && b.tasks[u].rows == b.tasks[t].rows &&
             b.tasks[u].cls == b.tasks[t].cls && span + b.tasks[u].cols <= capCols) { span += b.tasks[u].cols; ++u; }
      if (numGroups == capGroups) {
        capGroups *= 2;
        Group *p = realloc(groups, capGroups * sizeof(Group));
        if (!p) { free(groups); free(bufRead); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
        groups = p;
      }
      Group *g = &groups[numGroups++];
      g->taskBegin = t; g->taskEnd = u; g->outBuf = b.tasks[t].outBuf; g->outOff = b.tasks[t].outOff; g->rows = b.tasks[t].rows;
      g->slotOff = 0; g->reduced = 0; g->cls = b.tasks[t].cls;
      t = u;
    }

    /* which output vectors of this stage need a reduce pass, and the gaps of
     * those that do not.  Groups are sorted by (outBuf, outOff). */
    uint64_t numZeroItems = 0;
    uint64_t capRed = 4;
    st->reduce = calloc(capRed, sizeof(BfReduce));
    /* zero-fill chunks for direct buffers are appended as group-less items */
    typedef struct Gap { uint32_t buf; uint64_t off, len; } Gap;
    uint64_t numGaps = 0, capGaps = 16;
    Gap *gaps = malloc(capGaps * sizeof(Gap));
    if (!st->reduce || !gaps) { free(groups); free(gaps); free(bufRead); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto fail; }
#define PUSH_GAP1(B, O, L) do { if ((L) > 0) { if (numGaps == capGaps) { capGaps *= 2; Gap *p_ = realloc(gaps, capGaps * sizeof(Gap)); if (!p_) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; } gaps = p_; } gaps[numGaps].buf = (B); gaps[numGaps].off = (O); gaps[numGaps].len = (L); ++numGaps; } } while (0)
    /* a row-range shard zero-fills only what something reads (the live runs of the gap) */
#define PUSH_GAP(B, O, L) do { if (!lm.bits) PUSH_GAP1(B, O, L); else { uint64_t const o_ = (O), e_ = o_ + (L); uint64_t r_ = o_; \
      while (r_ < e_) { while (r_ < e_ && !liveBit(&lm, (B), r_)) ++r_; uint64_t const s_ = r_; while (r_ < e_ && liveBit(&lm, (B), r_)) ++r_; PUSH_GAP1(B, s_, r_ - s_); } } } while (0)

    {
      /* walk groups buffer by buffer */
      uint64_t gi = 0;
      /* mark buffers seen */
      uint8_t *seen = calloc(b.numBufs, 1);
      if (!seen) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
      while (gi < numGroups) {
        uint64_t gj = gi;
        uint32_t ob = groups[gi].outBuf;
        int overlap = 0;
        uint64_t prevEnd = 0;
        while (gj < numGroups && groups[gj].outBuf == ob) {
          if (gj > gi && groups[gj].outOff < prevEnd) overlap = 1;
          uint64_t e = groups[gj].outOff + groups[gj].rows;
          if (e > prevEnd) prevEnd = e;
          ++gj;
        }
        seen[ob] = 1;
        uint64_t blen = b.bufs[ob].len;
        if (!overlap) {
          uint64_t pos = 0;
          for (uint64_t g = gi; g < gj; ++g) {
            PUSH_GAP(ob, pos, groups[g].outOff - pos);
            pos = groups[g].outOff + groups[g].rows;
          }
          PUSH_GAP(ob, pos, blen - pos);
        } else {
          /* private slots + reduce */
          if (st->numReduce == capRed) {
            capRed *= 2;
            BfReduce *p = realloc(st->reduce, capRed * sizeof(BfReduce));
            if (!p) { free(seen); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
            memset(p + st->numReduce, 0, (capRed - st->numReduce) * sizeof(BfReduce));
            st->reduce = p;
          }
          BfReduce *rd = &st->reduce[st->numReduce++];
          memset(rd, 0, sizeof *rd);
          rd->destSpace = ob == by ? BF_SPACE_Y : BF_SPACE_TEMP;
          rd->destOff = ob == by ? 0 : b.bufs[ob].arenaOff;
          rd->numRows = blen;
          uint64_t ng = gj - gi;
          uint64_t *bp = malloc((2 * ng + 2) * 8);
          if (!bp) { free(seen); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
          uint64_t nbp = 0;
          bp[nbp++] = 0; bp[nbp++] = blen;
          for (uint64_t g = gi; g < gj; ++g) {
            bp[nbp++] = groups[g].outOff; bp[nbp++] = groups[g].outOff + groups[g].rows;
          }
          qsort(bp, nbp, 8, cmpU64);
          uint64_t w = 1;
          for (uint64_t i = 1; i < nbp; ++i) if (bp[i] != bp[w - 1]) bp[w++] = bp[i];
          nbp = w;
          uint64_t niv = nbp - 1;
          if (niv >= 0xffffffffu) { free(bp); free(seen); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "too many reduce intervals"); goto stage_fail; }
          rd->numIntervals = niv;
          rd->rowInterval = malloc((blen ? blen : 1) * 4);
          rd->ivBegin = calloc(niv + 2, 4);
          if (!rd->rowInterval || !rd->ivBegin) { free(bp); free(seen); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
          /* count sources per interval */
          uint64_t nsrc = 0;
          for (uint64_t g = gi; g < gj; ++g) {
            uint64_t lo = 0, hi = nbp;   /* first bp >= outOff */
            while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (bp[mid] < groups[g].outOff) lo = mid + 1; else hi = mid; }
            uint64_t endRow = groups[g].outOff + groups[g].rows;
            for (uint64_t i = lo; i < niv && bp[i] < endRow; ++i) { ++rd->ivBegin[i + 1]; ++nsrc; }
          }
          /* A group that meets no other group writes where it would without the overlap of its neighbours, and the reduce
           * leaves its rows alone (row interval BF_REDUCE_SKIP): the block columns of a packed adjoint plan of a streamed
           * butterfly are cut into several groups only where they are long, and 80 - 96 % of the rows of its stages have
           * exactly one source -- as slots they were written, read and copied once more for nothing.  (Real operands: the
           * overlaps of a complex fac_helm2 plan are whole block rows, every row has several sources.) */
          uint8_t *skip = calloc(niv + 1, 1);
          if (!skip) { free(bp); free(seen); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
          for (uint64_t g = gi; g < gj; ++g) {
            uint64_t lo = 0, hi = nbp;
            while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (bp[mid] < groups[g].outOff) lo = mid + 1; else hi = mid; }
            uint64_t endRow = groups[g].outOff + groups[g].rows;
            int alone = plan->dtype != BFHIP_C128;
            for (uint64_t i = lo; alone && i < niv && bp[i] < endRow; ++i) alone = rd->ivBegin[i + 1] == 1;
            if (alone) {
              for (uint64_t i = lo; i < niv && bp[i] < endRow; ++i) { skip[i] = 1; rd->ivBegin[i + 1] = 0; --nsrc; }
              groups[g].reduced = 0;
            } else {
              groups[g].reduced = 1;
              groups[g].slotOff = arenaAlloc(&top, groups[g].rows);
            }
          }
          for (uint64_t i = 0; i < niv; ++i)
            for (uint64_t r = bp[i]; r < bp[i + 1]; ++r) rd->rowInterval[r] = skip[i] ? BF_REDUCE_SKIP : (uint32_t)i;
          free(skip);
          rd->maxSrc = 0;
          for (uint64_t i = 0; i < niv; ++i) { if (rd->ivBegin[i + 1] > rd->maxSrc) rd->maxSrc = rd->ivBegin[i + 1]; }
          for (uint64_t i = 0; i < niv; ++i) rd->ivBegin[i + 1] += rd->ivBegin[i];
          rd->numSrc = nsrc;
          rd->srcBias = malloc((nsrc ? nsrc : 1) * 8);
          uint32_t *fill = calloc(niv + 1, 4);
          if (!rd->srcBias || !fill) { free(fill); free(bp); free(seen); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
          /* deterministic order: groups in (outOff, rows, emission) order */
          for (uint64_t g = gi; g < gj; ++g) {
            if (!groups[g].reduced) continue;
            uint64_t lo = 0, hi = nbp;
            while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (bp[mid] < groups[g].outOff) lo = mid + 1; else hi = mid; }
            uint64_t endRow = groups[g].outOff + groups[g].rows;
            for (uint64_t i = lo; i < niv && bp[i] < endRow; ++i)
              rd->srcBias[rd->ivBegin[i] + fill[i]++] = (int64_t)groups[g].slotOff - (int64_t)groups[g].outOff;
          }
          free(fill);
          free(bp);
        }
        gi = gj;
      }
      /* buffers due this stage that received nothing: all zeros */
      for (uint64_t bi = 1; bi < b.numBufs; ++bi)
        if (b.bufs[bi].stage == s && !seen[bi]) PUSH_GAP((uint32_t)bi, 0, b.bufs[bi].len);
      free(seen);
    }

    /* items.  A group of m rows is cut into items of `chunkRows` rows: 64 row
     * slots when the group's blocks are narrow, fewer when they are wide, so
     * that no item streams much more than BF_ITEM_BYTES -- the run time of a
     * launch is bounded below by its largest item (one wavefront), which is
     * what limits small launches (row-sharded operators, N <= 65536).  16-row
     * granularity keeps the MFMA kernel's slabs full; a short remainder still
     * fills the GEMV kernel's lanes through column groups. */
    uint64_t numItems = 0;
#ifndef BF_PLAN_BUNDLE_ORDER
#define BF_PLAN_BUNDLE_ORDER 1     /* 1 = runs of equal inputs kept together (product); A/B builds: 0 = items by cost bucket and first input (round 4), 2 = full bundles first, leftovers after (for -DBF_MF_BUNDLES=1 kernels) */
#endif
    int const bundled = BF_PLAN_BUNDLE_ORDER && po->groupByInput && !T && plan->dtype == BFHIP_C128;
    for (uint64_t g = 0; g < numGroups; ++g) {
      Group *gr = &groups[g];
      gr->colsSum = 0; gr->piecesPerChunk = 0;
      for (uint64_t t = gr->taskBegin; t < gr->taskEnd; ++t) {
        Task const *tk = &b.tasks[t];
        if (ir->kind[tk->leaf] == BFHIP_NODE_IDENTITY) { gr->piecesPerChunk += 1; gr->colsSum += 1; }
        else if (T) { gr->piecesPerChunk += (tk->cols + 7) / 8 + 2; gr->colsSum += tk->cols; }   /* upper bound: forward row chunks have >= 8 rows, one may straddle each end */
        else { gr->piecesPerChunk += (tk->cols + plan->xcap - 1) / plan->xcap; gr->colsSum += tk->cols; }
      }
      uint64_t chunk = gr->cls ? 16 : itemRows;
      if (bundled && chunk > 32) chunk = 32;        /* one two-slab pass per item: the wavefronts of a bundle run the same number of k-steps */
      if (!T && gr->colsSum) {
        uint64_t const floorRows = (po->minChunkRows ? po->minChunkRows : 16) * plan->epl;
        uint64_t const gran = floorRows < 16 * plan->epl ? floorRows : 16 * plan->epl;
        uint64_t want = (itemBytes / plan->elemSize) / gr->colsSum / gran * gran;
        if (want < floorRows) want = floorRows;
        if (want < chunk) chunk = want;
      }
      gr->chunkRows = (uint32_t)chunk;
      numItems += (gr->rows + chunk - 1) / chunk;
    }
    /* zero fills run in the stage's kernel: forward 64 row slots, transposed this stage's 16 or 64 columns (a 64-row fill
     * in a 16-column stage would push the whole stage onto the 64-column kernel) */
    uint64_t const zeroChunk = T ? itemRows : plan->maxItemRows;
    for (uint64_t z = 0; z < numGaps; ++z) numZeroItems += (gaps[z].len + zeroChunk - 1) / zeroChunk;
    ItemTmp *tmp = malloc((numItems + numZeroItems + 1) * sizeof(ItemTmp));
    if (!tmp) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
    uint64_t ni = 0;
    uint64_t numPieces = 0;
    for (uint64_t g = 0; g < numGroups; ++g) {
      uint64_t m = groups[g].rows;
      uint64_t chunk = groups[g].chunkRows;
      uint64_t colsSum = groups[g].colsSum, piecesPerChunk = groups[g].piecesPerChunk;
      for (uint64_t r0 = 0; r0 < m; r0 += chunk) {
        uint64_t rows = m - r0 < chunk ? m - r0 : chunk;
        tmp[ni].group = g; tmp[ni].rowBegin = (uint32_t)r0; tmp[ni].rows = (uint32_t)rows;
        tmp[ni].cost = rows * colsSum;
        tmp[ni].sortCost = tmp[ni].cost;
        tmp[ni].inKey = 0;
        tmp[ni].sig = 0; tmp[ni].bundle = 0;
        if (po->groupByInput && !T) {
          /* Items that read the same input rows -- the row chunks of a group, and the sibling groups of a radix-4
           * butterfly stage, which read the same column blocks (src/fac_helm2.c:277-318) -- become list neighbours, so
           * that the RHS-block kernel runs them side by side on one XCD and their X panel is fetched into ONE L2 once:
           * the list is big-first by cost BUCKET (quarter octaves), by input inside a bucket. */
          tmp[ni].sortCost = costBucket(tmp[ni].cost);
          Task const *t0 = &b.tasks[groups[g].taskBegin];
          tmp[ni].inKey = ((uint64_t)t0->inBuf << 40) | (t0->inOff & 0xffffffffffull);
          if (bundled) {
            uint64_t h = 1469598103934665603ull;
            for (uint64_t t = groups[g].taskBegin; t < groups[g].taskEnd; ++t) {
              Task const *tk = &b.tasks[t];
              uint64_t const w[3] = { ((uint64_t)tk->inBuf << 1) | (ir->kind[tk->leaf] == BFHIP_NODE_IDENTITY), tk->inOff, tk->cols };
              for (int q = 0; q < 3; ++q) { h ^= w[q]; h *= 1099511628211ull; h ^= h >> 29; }
            }
            tmp[ni].sig = h;
          }
        }
        /* colsSum < 128 also means "not row-major" */
        tmp[ni].small = !T && plan->dtype != BFHIP_C128 && rows <= 2 * plan->epl && colsSum < BF_SMALL_COLS && piecesPerChunk <= BF_SMALL_PIECES;
        tmp[ni].narrow = groups[g].cls;
        ++ni;
        numPieces += piecesPerChunk;
      }
    }
    numItems = ni;
    if (bundled && numItems) {
      /* same inputs together, big first inside a run; runs are cut into bundles; bundles are ordered like items were */
      qsort(tmp, numItems, sizeof(ItemTmp), cmpItemSig);
      BundleTmp *bt = malloc(numItems * sizeof(BundleTmp));
      if (!bt) { free(tmp); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
      /* a bundle is FULL: BF_BUNDLE_ITEMS items with the same inputs and the same number of 16-row slabs (its wavefronts meet at a
       * barrier every k-step).  What is left over -- runs shorter than that -- follows the full bundles, ordered as the items of an
       * unbundled list are: the kernel runs them four unrelated items to a workgroup. */
      uint64_t nb = 0;
      for (uint64_t i = 0; i < numItems;) {
        Group const *g0 = &groups[tmp[i].group];
        uint64_t j = i + 1;
        while (j < numItems && j - i < BF_BUNDLE_ITEMS && tmp[j].sig == tmp[i].sig && g0->colsSum && (tmp[j].rows > 16) == (tmp[i].rows > 16) &&
               (tmp[j].group == tmp[i].group || sameInputs(b.tasks, ir, g0->taskBegin, g0->taskEnd, groups[tmp[j].group].taskBegin, groups[tmp[j].group].taskEnd))) ++j;
        if ((j - i == BF_BUNDLE_ITEMS && BF_BUNDLE_ITEMS > 1) || BF_PLAN_BUNDLE_ORDER == 1) {
          bt[nb].cost = tmp[i].cost; bt[nb].sortCost = costBucket(tmp[i].cost); bt[nb].inKey = tmp[i].inKey; bt[nb].first = i;
          for (uint64_t k = i; k < j; ++k) tmp[k].bundle = nb;
          ++nb;
        } else
          for (uint64_t k = i; k < j; ++k) tmp[k].bundle = UINT64_MAX;
        i = j;
      }
      qsort(bt, nb, sizeof(BundleTmp), cmpBundle);
      uint64_t *place = malloc((nb ? nb : 1) * 8);
      if (!place) { free(bt); free(tmp); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
      for (uint64_t k = 0; k < nb; ++k) place[tmp[bt[k].first].bundle] = k;
      for (uint64_t i = 0; i < numItems; ++i) if (tmp[i].bundle != UINT64_MAX) tmp[i].bundle = place[tmp[i].bundle];
      free(place); free(bt);
      qsort(tmp, numItems, sizeof(ItemTmp), cmpItemBundle);
    } else
    qsort(tmp, numItems, sizeof(ItemTmp), cmpItemCost);
    uint64_t totalItems = numItems + numZeroItems;
    if (totalItems >= 0xffffffffu || numPieces >= 0xffffffffu) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "stage too large for 32-bit indices"); goto stage_fail; }
    st->items = malloc((totalItems ? totalItems : 1) * sizeof(BfDevItem));
    st->pieces = malloc((numPieces ? numPieces : 1) * sizeof(BfDevPiece));
    st->pieceSrc = malloc((numPieces ? numPieces : 1) * sizeof(BfPieceSrc));
    st->pieceBuf = calloc(numPieces ? numPieces : 1, 4);
    st->itemBuf = calloc(totalItems ? totalItems : 1, 4);
    if (!st->items || !st->pieces || !st->pieceSrc || !st->pieceBuf || !st->itemBuf) { free(tmp); rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto stage_fail; }
    uint64_t np = 0;
    uint64_t numBig = 0;
    while (numBig < numItems && !tmp[numBig].small) ++numBig;
    st->firstSmall = numBig + numZeroItems;        /* order: ordinary items (transposed: 16-column ones first; big first), zero fills, small items */
    st->numNarrow = 0;
    while (st->numNarrow < numItems && tmp[st->numNarrow].narrow) ++st->numNarrow;
    st->maxRowsRest = 0;
    for (uint64_t i = 0; i < numItems; ++i) {
      Group const *g = &groups[tmp[i].group];
      BfDevItem *it = &st->items[i < numBig ? i : i + numZeroItems];
      uint32_t mr = tmp[i].rows, r0 = tmp[i].rowBegin;
      uint32_t mrPad = (uint32_t)roundUp(mr, plan->epl);
      it->pieceBegin = (uint32_t)np;
      uint64_t outOff;
      uint32_t flags = 0;
      /* Few-row wide leaves of real operands (a streamed butterfly ends in row nodes of ~5 rows with leaves thousands
       * of columns wide) are stored ROW-major: column-major they pad 5 rows to the 4- (fp32) or 2-row (fp64) lane
       * granule (+60 % bytes) and give the transposed kernel 2 of its 16 row lanes; row-major every lane owns
       * 16 bytes of consecutive columns, forward and transposed, and nothing is padded but the row ends. */
      /* (the small items of a stage are row-major too, whatever their width: their kernel reads nothing else) */
      /* (not a chunk made of hundreds of pieces a few columns wide -- the transposes of few-row leaves in a packed adjoint plan:
       * row-major each piece is a dependent step that keeps two lanes busy; column-major the kernel contracts runs of them as
       * one block) */
      int const rowMajor = !T && plan->dtype != BFHIP_C128 && mr <= 2 * plan->epl &&
                           (tmp[i].small || (g->colsSum >= 128 && g->colsSum >= 16 * g->piecesPerChunk));
      if (rowMajor) flags |= BF_ITEM_ROWMAJOR;
      if (g->reduced) outOff = g->slotOff + r0;
      else if (g->outBuf == by) { outOff = g->outOff + r0; flags |= BF_ITEM_OUT_Y; }
      else { outOff = b.bufs[g->outBuf].arenaOff + g->outOff + r0; st->itemBuf[i < numBig ? i : i + numZeroItems] = g->outBuf; ++plan->bufWriters[g->outBuf]; }
      if (outOff >= 0xffffffffu) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "vector arena exceeds 32-bit offsets"); goto stage_fail; }
      it->outOff = (uint32_t)outOff;
      if (tmp[i].small) flags |= BF_ITEM_SMALL;
      if (tmp[i].narrow) flags |= BF_ITEM_TNARROW;
      else if (mr > st->maxRowsRest) st->maxRowsRest = mr;
      it->mrFlags = mr | flags;
      if (mr > st->maxRows) st->maxRows = mr;
      for (uint64_t t = g->taskBegin; t < g->taskEnd; ++t) {
        Task const *tk = &b.tasks[t];
        uint64_t inBase = tk->inBuf == bx ? tk->inOff : b.bufs[tk->inBuf].arenaOff + tk->inOff;
        uint32_t inFlag = tk->inBuf == bx ? BF_PIECE_IN_X : 0;
        if (!bufRead[tk->inBuf]) { bufRead[tk->inBuf] = 1; st->vecIn += b.bufs[tk->inBuf].len; }
        if (ir->kind[tk->leaf] == BFHIP_NODE_IDENTITY) {
          BfDevPiece *pc = &st->pieces[np];
          pc->dataOff = 0; pc->inOff = (uint32_t)(inBase + r0); pc->ncols = mr; pc->flags = inFlag | BF_PIECE_IDENTITY; pc->ld = 0;
          st->pieceSrc[np].node = tk->leaf; st->pieceSrc[np].row0 = (uint32_t)tk->rsub0 + r0; st->pieceSrc[np].col0 = 0;
          st->pieceBuf[np] = tk->inBuf == bx ? 0 : tk->inBuf;
          ++np;
          continue;
        }
        if (T) {
          /* every forward piece of this leaf whose column range holds this chunk of A's columns */
          uint64_t found = 0;
          /* this task contracts rows [sub0, sub0 + cols) of A; inBase already points at row sub0 of the input */
          uint64_t const ta = tk->sub0, tb = tk->sub0 + tk->cols;
          /* column-major forward pieces start at multiples of xcap, row-major ones at multiples of BF_TASK_SPAN
           * (row chunks of one leaf may be of either kind): look both up */
          for (int pass = 0; pass < 2; ++pass) {
          uint32_t const colPiece0 = pass ? r0 / BF_TASK_SPAN * BF_TASK_SPAN : r0 / plan->xcap * plan->xcap;
          uint64_t k = findFwd(po->fwdPieces, po->numFwdPieces, tk->leaf, colPiece0);
          for (; k < po->numFwdPieces && po->fwdPieces[k].node == tk->leaf && po->fwdPieces[k].col0 == colPiece0; ++k) {
            BfFwdPiece const *fp = &po->fwdPieces[k];
            if ((int)fp->rowMajor != pass || r0 + mr > fp->col0 + fp->ncols) continue;
            uint64_t const lo = fp->row0 > ta ? fp->row0 : ta, hi = (uint64_t)fp->row0 + fp->mr < tb ? (uint64_t)fp->row0 + fp->mr : tb;
            if (lo >= hi) continue;                          /* forward piece outside this task's rows */
            if (np >= numPieces) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: transposed piece count"); goto stage_fail; }
            BfDevPiece *pc = &st->pieces[np];
            if (inBase + (lo - ta) >= 0xffffffffu) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "vector arena exceeds 32-bit offsets"); goto stage_fail; }
            /* a forward piece that straddles the task boundary is entered at row lo: a multiple of the lane
             * granule, since task boundaries (multiples of BF_TASK_SPAN) and forward row chunks both are */
            if (!fp->rowMajor && (lo - fp->row0) % plan->epl) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: transposed piece not on a lane boundary"); goto stage_fail; }
            pc->inOff = (uint32_t)(inBase + (lo - ta));
            pc->ncols = (uint32_t)(hi - lo);   /* steps = rows of the forward piece taken */
            pc->flags = inFlag;
            if (fp->rowMajor) {                /* element (row s, column j) = arena[dataOff + s * ld + j] */
              pc->flags |= BF_PIECE_ROWMAJOR;
              pc->dataOff = fp->dataOff + (lo - fp->row0) * fp->ldr + (r0 - fp->col0);
              pc->ld = fp->ldr;
            } else {
              pc->dataOff = fp->dataOff + (uint64_t)(r0 - fp->col0) * fp->mrPad + (lo - fp->row0);
              pc->ld = fp->mrPad;
            }
            st->pieceSrc[np].node = tk->leaf; st->pieceSrc[np].row0 = (uint32_t)lo; st->pieceSrc[np].col0 = r0;
            st->pieceBuf[np] = tk->inBuf == bx ? 0 : tk->inBuf;
            ++np; ++found;
          }
          }
          if (!found) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: leaf missing from the forward plan"); goto stage_fail; }
          continue;
        }
        /* a column-major piece's x segment is staged in LDS (xcap columns); a row-major piece reads x from global
         * memory and spans the whole task (<= BF_TASK_SPAN columns, starting at a multiple of it) */
        uint64_t const pieceCols = rowMajor ? BF_TASK_SPAN : plan->xcap;
        for (uint64_t c0 = 0; c0 < tk->cols; c0 += pieceCols) {
          uint64_t nc = tk->cols - c0 < pieceCols ? tk->cols - c0 : pieceCols;
          BfDevPiece *pc = &st->pieces[np];
          if (inBase + c0 >= 0xffffffffu) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "vector arena exceeds 32-bit offsets"); goto stage_fail; }
          pc->dataOff = arenaTop; pc->inOff = (uint32_t)(inBase + c0); pc->ncols = (uint32_t)nc; pc->flags = inFlag; pc->ld = 0;
          st->pieceSrc[np].node = tk->leaf; st->pieceSrc[np].row0 = (uint32_t)tk->rsub0 + r0; st->pieceSrc[np].col0 = (uint32_t)(tk->sub0 + c0);
          st->pieceBuf[np] = tk->inBuf == bx ? 0 : tk->inBuf;
          if (rowMajor) {
            /* every row starts on a 128-byte line (rowAlignBytes; +0.3 % arena on the streamed benchmark operand): a
             * 64-column chunk of a row, what a transposed item reads, is then whole lines -- at 16-byte alignment it
             * touches three lines for two and the transposed stage fetches 1.4x its bytes (PMC, DESIGN_EXPERIMENTS.md section 10) --
             * and the forward kernel's 1 KiB row loads are whole lines too (6.35 -> 6.57 TB/s on the W0 stage) */
            /* (not the narrow pieces of small items: a 47-column row is 188 bytes) */
            uint64_t const al = po->rowAlignBytes > 16 && nc >= 128 ? po->rowAlignBytes / plan->elemSize : plan->epl;
            arenaTop = roundUp(arenaTop, al);
            pc->dataOff = arenaTop;
            pc->flags |= BF_PIECE_ROWMAJOR;
            pc->ld = (uint32_t)roundUp(nc, al);
            arenaTop += (uint64_t)mr * pc->ld;
          } else
          arenaTop += (uint64_t)mrPad * nc;
          ++np;
        }
      }
      it->numPieces = (uint32_t)(np - it->pieceBegin);
      /* the pieces of an item are packed back to back: when they are few and narrow the kernel takes them as one
       * block (one LDS hand-off and one dependent load chain per item instead of one per piece) */
      if (!T && !rowMajor && plan->dtype != BFHIP_C128 && it->numPieces <= 64) {
        uint64_t dense = 0;
        for (uint32_t k = 0; k < it->numPieces; ++k)
          if (!(st->pieces[it->pieceBegin + k].flags & BF_PIECE_IDENTITY)) dense += st->pieces[it->pieceBegin + k].ncols;
        if (dense && dense <= BF_MERGE_COLS) it->mrFlags |= BF_ITEM_MERGED;
      }
    }
    /* zero-fill items */
    uint64_t ii = numBig;
    for (uint64_t z = 0; z < numGaps; ++z) {
      for (uint64_t r0 = 0; r0 < gaps[z].len; r0 += zeroChunk) {
        uint64_t rows = gaps[z].len - r0 < zeroChunk ? gaps[z].len - r0 : zeroChunk;
        BfDevItem *it = &st->items[ii++];
        it->pieceBegin = (uint32_t)np; it->numPieces = 0;
        uint64_t outOff = gaps[z].buf == by ? gaps[z].off + r0 : b.bufs[gaps[z].buf].arenaOff + gaps[z].off + r0;
        if (outOff >= 0xffffffffu) { free(tmp); rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "vector arena exceeds 32-bit offsets"); goto stage_fail; }
        it->outOff = (uint32_t)outOff;
        it->mrFlags = (uint32_t)rows | (gaps[z].buf == by ? BF_ITEM_OUT_Y : 0);
        if (gaps[z].buf != by) { st->itemBuf[ii - 1] = gaps[z].buf; ++plan->bufWriters[gaps[z].buf]; }
        if (rows > st->maxRows) st->maxRows = (uint32_t)rows;
        if (rows > st->maxRowsRest) st->maxRowsRest = (uint32_t)rows;
      }
    }
    st->numItems = totalItems;
    st->numPieces = np;
    if (!T && plan->dtype == BFHIP_C128 && totalItems && (rc = bfPlanBundles(st->items, st->pieces, totalItems, &st->bundleBegin, &st->numBundles))) { free(tmp); goto stage_fail; }
    st->numCoopNarrow = T ? bfPlanCountCoop(st->items, st->pieces, st->numNarrow, plan->elemSize) : 0;
    st->numCoop = T ? bfPlanCountCoop(st->items + st->numNarrow, st->pieces, numItems - st->numNarrow, plan->elemSize) : 0;
    /* algorithmic counts */
    for (uint64_t t = tBegin; t < tEnd; ++t)
      if (ir->kind[b.tasks[t].leaf] == BFHIP_NODE_DENSE) { st->leafElems += b.tasks[t].rows * b.tasks[t].cols; plan->numLeaves += b.tasks[t].sub0 == 0; }
    for (uint64_t bi = 1; bi < b.numBufs; ++bi) if (b.bufs[bi].stage == s) st->vecOut += b.bufs[bi].len;
    plan->leafElems += st->leafElems;
    memset(bufRead, 0, b.numBufs);
    free(tmp);
    free(groups);
    free(gaps);
    tBegin = tEnd;
    continue;
  stage_fail:
    free(groups);
    free(gaps);
    free(bufRead);
    goto fail;
#undef PUSH_GAP
#undef PUSH_GAP1
  }
  free(bufRead);
  plan->flowOk = !T;
  for (int32_t s = 0; s < S; ++s)
    for (uint64_t r = 0; r < plan->stages[s].numReduce; ++r)
      if (plan->stages[s].reduce[r].destSpace != BF_SPACE_Y) plan->flowOk = 0;      /* a reduce into an intermediate sits between two stages */
  plan->arenaElems = arenaTop;
  plan->tempElems = roundUp(top, 4);
  free(b.tasks);
  free(b.bufs);
  free(fullStageElems);
  liveFree(&lm);
  return 0;

fail:
  free(b.tasks);
  free(b.bufs);
  free(fullStageElems);
  liveFree(&lm);
  bfPlanFree(plan);
  return rc ? rc : BFABI_ERROR_RUNTIME_ERROR;
}
