/* bfhip_api.c -- the C-ABI of include/bfhip.h: operator lifetime, compile
 * (IR -> plan -> HBM), apply, profiling, and the BfMat vtable shim that makes
 * the device operator a drop-in behind the reference's bfMatMul / bfMatMulVec
 * (reference src/mat.c:183-189; precedent for a foreign operator behind the
 * vtable: BfMatFunc, include/bf/mat_func.h:5-28, src/mat_func.c:59-82).
 */
#define _GNU_SOURCE
#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"
#include "../../include/bfhip_synth.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* ---- errors ----------------------------------------------------------------- */
static __thread char lastError[512];

int bfhipFail(int code, char const *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(lastError, sizeof lastError, fmt, ap);
  va_end(ap);
  return code;
}
char const *bfhipLastErrorMessage(void) { return lastError; }
char const *bfhipErrorString(int code) {
  static char const *const names[] = {"BF_ERROR_NONE", "BF_ERROR_INVALID_ARGUMENTS", "BF_ERROR_RUNTIME_ERROR",
                                      "BF_ERROR_NOT_IMPLEMENTED", "BF_ERROR_MEMORY_ERROR", "BF_ERROR_OUT_OF_RANGE",
                                      "BF_ERROR_FILE_ERROR", "BF_ERROR_TYPE_ERROR", "BF_ERROR_INCOMPATIBLE_SHAPES"};
  return (code >= 0 && code <= 8) ? names[code] : "BF_ERROR_UNKNOWN";
}

double bfhipSyntheticValue(uint64_t seed, uint64_t idx, int imag) { return bfhip_synth_value(seed, idx, imag); }

int bfhipSyntheticLeafBases(BfhipDesc const *desc, uint64_t *bases) {
  if (!desc || !bases) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  uint64_t acc = 0;
  for (uint64_t i = 0; i < desc->numNodes; ++i) {
    bases[i] = acc;
    if (desc->kind[i] == BFHIP_NODE_DENSE) acc += desc->rows[i] * desc->cols[i];
  }
  return 0;
}

/* ---- operator --------------------------------------------------------------- */
struct BfhipOperator {
  BfPlan plan;
  BfPlan tplan;               /* plan of A^T over the same leaf arena (BFHIP_FLAG_ADJOINT) */
  int hasTplan;
  uint32_t srcDtype;          /* dtype of the operand as given (C128 / F64) */
  int device;
  uint32_t flags;
  void *dArena;               /* leaf data */
  void *dArenaT;              /* BFHIP_FLAG_ADJOINT_PACKED: the leaves of A^T packed for tplan, a FORWARD plan of the transposed expression */
  void *dTemp;                /* vector arena: intermediates + partial slots, tempElems * maxRhs */
  void *dZero;                /* 4 KiB of zeros */
  uint32_t tempRhs;
  uint64_t metaBytes;
  uint64_t leafBytesAlgorithmic;
  /* staging for the host-pointer apply */
  void *dX, *dY;
  void *hX, *hY;              /* pinned host mirrors of dX / dY */
  void *evHost[4];            /* "piece i of the result is in hY" (created on first use) */
  uint32_t xyRhs;
  /* profiling */
  void **evStart, **evStop;   /* [BF_EV_POOL][numStages]: one set per apply in flight, so timing an apply never waits for the one before */
  double *stageMs;
  uint64_t *stageLaunches;
  uint32_t lastNrhs;
  uint64_t evIssued, evHarvested;   /* applies whose events were recorded / read back */
  uint64_t applyCount;              /* forward applies so far */
  void *dCov;                       /* scratch of the covariance products (2 vectors of the longer side) */
  uint32_t profEvery;               /* events around one apply in profEvery (0, 1: every apply) */
  /* BFHIP_FLAG_PLAN_ONLY: the IR is kept (borrowed leaf pointers!) for bfhipPlanPackArena; irT: its transposed view when the
   * adjoint plan has an arena of its own (BFHIP_FLAG_ADJOINT_PACKED), for bfhipPlanPackArenaT */
  BfIr *ir, *irT;
  int packedT;
  uint64_t seed;
  /* dependency-driven launch of the forward plan (complex128, one right-hand side): flat copies of the index tables */
  int flow;
  void *dFlowItems, *dFlowPieces, *dFlowItemOut, *dFlowWriters, *dFlowCounters;
  uint32_t flowNumItems, flowGrid, flowEpoch, flowQueueBase, flowMaxWriters;
  uint64_t flowNumBufs;
  uint8_t evFlow[64];               /* per event set: that apply ran as ONE launch (its time is recorded under stage 0) */
};

#define BF_ARENA_SLACK 256u
#define BF_HOST_PIECE_ROWS 8192u    /* host vectors of more than 4 x this many rows cross PCIe in 4 pieces, copy and DMA overlapped */
#define BF_EV_POOL 64u

static void freeDevicePlanOf(BfPlan *plan) {
  for (uint64_t s = 0; s < plan->numStages && plan->stages; ++s) {
    BfStage *st = &plan->stages[s];
    bfdevFree(st->dItems); st->dItems = NULL;
    bfdevFree(st->dPieces); st->dPieces = NULL;
    bfdevFree(st->dBundleBegin); st->dBundleBegin = NULL;
    bfdevFree(st->dTickets); st->dTickets = NULL;
    for (uint64_t r = 0; r < st->numReduce; ++r) {
      bfdevFree(st->reduce[r].dRowInterval); bfdevFree(st->reduce[r].dIvBegin); bfdevFree(st->reduce[r].dSrcBias);
      st->reduce[r].dRowInterval = st->reduce[r].dIvBegin = st->reduce[r].dSrcBias = NULL;
    }
  }
}
static void freeDevicePlan(BfhipOperator *op) {
  freeDevicePlanOf(&op->plan);
  freeDevicePlanOf(&op->tplan);
}

void bfhipFree(BfhipOperator **pop) {
  if (!pop || !*pop) return;
  BfhipOperator *op = *pop;
  int prev = -1;
  if (!(op->flags & BFHIP_FLAG_PLAN_ONLY)) {
    bfdevGetDevice(&prev);
    bfdevSetDevice(op->device);
  }
  if (op->evStart && op->evStop) for (uint64_t s = 0; s < BF_EV_POOL * op->plan.numStages; ++s) { bfdevEventDestroy(op->evStart[s]); bfdevEventDestroy(op->evStop[s]); }
  free(op->evStart); free(op->evStop); free(op->stageMs); free(op->stageLaunches);
  freeDevicePlan(op);
  bfdevFree(op->dArena);
  bfdevFree(op->dArenaT);
  bfdevFree(op->dTemp);
  bfdevFree(op->dZero);
  bfdevFree(op->dFlowItems); bfdevFree(op->dFlowPieces); bfdevFree(op->dFlowItemOut); bfdevFree(op->dFlowWriters); bfdevFree(op->dFlowCounters);
  bfdevFree(op->dX);
  bfdevFree(op->dCov);
  bfdevFree(op->dY);
  bfdevHostFreePinned(op->hX);
  bfdevHostFreePinned(op->hY);
  for (int i = 0; i < 4; ++i) bfdevEventDestroy(op->evHost[i]);
  bfPlanFree(&op->plan);
  bfPlanFree(&op->tplan);
  if (op->ir) { bfIrFree(op->ir); free(op->ir); }
  if (op->irT) { bfIrFree(op->irT); free(op->irT); }
  int const touchedDevice = !(op->flags & BFHIP_FLAG_PLAN_ONLY);
  free(op);
  *pop = NULL;
  if (touchedDevice && prev >= 0) bfdevSetDevice(prev);
}

static int uploadArray(void **d, void const *h, size_t bytes, uint64_t *meta) {
  int rc = bfdevMalloc(d, bytes);
  if (rc) return rc;
  *meta += bytes;
  return bfdevMemcpyH2D(*d, h, bytes);
}

/* write one piece (mr x ncols sub-block of a leaf, column-major, rows padded
 * to mrPad) at dst */
static void packPiece(BfPlan const *pl, BfIr const *ir, BfDevPiece const *pc, BfPieceSrc const *src,
                      uint32_t mr, uint32_t mrPad, unsigned char *dst, uint64_t seed) {
  uint64_t node = src->node;
  void const *data = ir->leafData[node];
  int const cplx = pl->dtype == BFHIP_C128;
  uint64_t ldr = ir->leafRowStride[node], ldc = ir->leafColStride[node];
  double const *A = (double const *)data;
  double scale = 0;
  /* synthetic leaves: element (i, j) of the ORIGINAL leaf is stream value base + i * n + j, n its column count; the leaves of a
   * transposed view (bfIrTransposed) hold element (j, i) of it */
  uint64_t vbase = ir->synthBase[node], n = ir->transposedView ? ir->rows[node] : ir->cols[node];
  uint64_t const sR = ir->transposedView ? 1 : n, sC = ir->transposedView ? n : 1;
  if (!data) scale = cplx ? sqrt(3.0 / (2.0 * (double)n)) : sqrt(3.0 / (double)n);
  int const rowMajor = (pc->flags & BF_PIECE_ROWMAJOR) != 0;      /* real dtypes only */
  uint32_t const rowsStored = rowMajor ? mr : mrPad;
  for (uint32_t c = 0; c < pc->ncols; ++c) {
    for (uint32_t r = 0; r < rowsStored; ++r) {
      double re = 0, im = 0;
      if (r < mr) {
        uint64_t i = src->row0 + r, j = src->col0 + c;
        if (data) {
          if (cplx && !(ir->leafReal[node] & BF_LEAF_REAL)) { double const *e = A + 2 * (i * ldr + j * ldc); re = e[0]; im = (ir->leafReal[node] & BF_LEAF_CONJ) ? -e[1] : e[1]; }
          else re = A[i * ldr + j * ldc];
        } else {
          re = bfhip_synth_value(seed, vbase + i * sR + j * sC, 0) * scale;
          if (cplx) im = bfhip_synth_value(seed, vbase + i * sR + j * sC, 1) * scale;
        }
      }
      uint64_t e = rowMajor ? (uint64_t)r * pc->ld + c : (uint64_t)c * mrPad + r;
      if (cplx) { ((double *)dst)[2 * e] = re; ((double *)dst)[2 * e + 1] = im; }
      else if (pl->dtype == BFHIP_F64) ((double *)dst)[e] = re;
      else ((float *)dst)[e] = (float)re;
    }
  }
  if (rowMajor)      /* the row ends are padded to the lane granule with zeros */
    for (uint32_t r = 0; r < mr; ++r)
      for (uint32_t c = pc->ncols; c < pc->ld; ++c) {
        uint64_t e = (uint64_t)r * pc->ld + c;
        if (pl->dtype == BFHIP_F64) ((double *)dst)[e] = 0; else ((float *)dst)[e] = 0;
      }
  /* sparse decorations folded into this leaf (bfhip_ir.c: tryFold): patches are sorted by (leaf, row, col) */
  if (ir->numPatches && data) {
    uint64_t lo = 0, hi = ir->numPatches;
    while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (ir->patches[mid].leaf < node || (ir->patches[mid].leaf == node && ir->patches[mid].row < src->row0)) lo = mid + 1; else hi = mid; }
    for (; lo < ir->numPatches && ir->patches[lo].leaf == node && ir->patches[lo].row < src->row0 + mr; ++lo) {
      BfIrPatch const *pt = &ir->patches[lo];
      if (pt->col < src->col0 || pt->col >= src->col0 + pc->ncols) continue;
      uint64_t e = rowMajor ? (uint64_t)(pt->row - src->row0) * pc->ld + (pt->col - src->col0) : (uint64_t)(pt->col - src->col0) * mrPad + (pt->row - src->row0);
      if (cplx) { ((double *)dst)[2 * e] += pt->re; ((double *)dst)[2 * e + 1] += pt->im; }
      else if (pl->dtype == BFHIP_F64) ((double *)dst)[e] += pt->re;
      else ((float *)dst)[e] = (float)((double)((float *)dst)[e] + pt->re);
    }
  }
}

/* pack leaves into the arena, stage by stage, in arena order.  hostDst != NULL:
 * write everything (synthetic leaves included) to host memory; else upload
 * host-valued leaves through a staging buffer and synthesize the rest on the
 * device. */
static int packLeavesPlan(BfPlan const *pl, void *dArena, BfIr const *ir, uint64_t seed, void *hostDst) {
  size_t const es = pl->elemSize;
  int const cplx = pl->dtype == BFHIP_C128;
  size_t const chunkBytes = (size_t)64 << 20;
  unsigned char *stage = NULL;
  BfSynthPiece *synth = NULL;
  uint64_t numSynth = 0, capSynth = 0;
  int rc = 0;
  uint64_t chunkBase = 0;    /* arena element offset of stage[0] */
  size_t fill = 0;           /* bytes used in stage */
  for (uint64_t s = 0; s < pl->numStages && !rc; ++s) {
    BfStage const *st = &pl->stages[s];
    for (uint64_t i = 0; i < st->numItems && !rc; ++i) {
      BfDevItem const *it = &st->items[i];
      uint32_t mr = it->mrFlags & 0xffffu;
      uint32_t mrPad = (mr + pl->epl - 1) / pl->epl * pl->epl;
      for (uint32_t k = 0; k < it->numPieces && !rc; ++k) {
        BfDevPiece const *pc = &st->pieces[it->pieceBegin + k];
        BfPieceSrc const *src = &st->pieceSrc[it->pieceBegin + k];
        if (pc->flags & BF_PIECE_IDENTITY) continue;
        uint64_t node = src->node;
        size_t bytes = ((pc->flags & BF_PIECE_ROWMAJOR) ? (size_t)mr * pc->ld : (size_t)mrPad * pc->ncols) * es;
        if (hostDst) {
          packPiece(pl, ir, pc, src, mr, mrPad, (unsigned char *)hostDst + pc->dataOff * es, seed);
          continue;
        }
        if (!ir->leafData[node]) {
          if (numSynth == capSynth) {
            capSynth = capSynth ? capSynth * 2 : 4096;
            BfSynthPiece *p = realloc(synth, capSynth * sizeof *synth);
            if (!p) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (synth pieces)"); break; }
            synth = p;
          }
          BfSynthPiece *sp = &synth[numSynth++];
          sp->dataOff = pc->dataOff;
          sp->vbase = ir->synthBase[node];
          uint64_t const ncol = ir->transposedView ? ir->rows[node] : ir->cols[node];      /* columns of the leaf the stream was laid over */
          sp->strideR = ir->transposedView ? 1u : (uint32_t)ncol;
          sp->strideC = ir->transposedView ? (uint32_t)ncol : 1u;
          sp->row0 = src->row0; sp->col0 = src->col0;
          sp->mr = mr; sp->mrPad = mrPad; sp->ncols = pc->ncols;
          sp->rowMajor = (pc->flags & BF_PIECE_ROWMAJOR) != 0; sp->ldr = pc->ld;
          sp->scale = cplx ? sqrt(3.0 / (2.0 * (double)ncol)) : sqrt(3.0 / (double)ncol);
          continue;
        }
        if (!stage) {
          stage = malloc(chunkBytes);
          if (!stage) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (staging)"); break; }
          chunkBase = pc->dataOff; fill = 0;
        }
        /* pieces are consecutive in the arena except across synthetic ones */
        if (fill && (pc->dataOff != chunkBase + fill / es || fill + bytes > chunkBytes)) {
          rc = bfdevMemcpyH2D((char *)dArena + chunkBase * es, stage, fill);
          fill = 0;
          if (rc) break;
        }
        if (!fill) chunkBase = pc->dataOff;
        if (bytes > chunkBytes) { rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "piece larger than staging chunk"); break; }
        packPiece(pl, ir, pc, src, mr, mrPad, stage + fill, seed);
        fill += bytes;
      }
    }
  }
  if (!rc && fill) rc = bfdevMemcpyH2D((char *)dArena + chunkBase * es, stage, fill);
  free(stage);
  if (!rc && numSynth) rc = bfdevSynthFill(dArena, pl->dtype, synth, numSynth, seed);
  free(synth);
  return rc;
}

static int packLeaves(BfhipOperator const *op, BfIr const *ir, uint64_t seed, void *hostDst) {
  return packLeavesPlan(&op->plan, op->dArena, ir, seed, hostDst);
}

static int uploadPlanMeta(BfhipOperator *op, BfPlan *plan) {
  int rc = 0;
  for (uint64_t s = 0; s < plan->numStages && !rc; ++s) {
    BfStage *st = &plan->stages[s];
    rc = uploadArray(&st->dItems, st->items, st->numItems * sizeof(BfDevItem), &op->metaBytes);
    if (!rc) rc = uploadArray(&st->dPieces, st->pieces, st->numPieces * sizeof(BfDevPiece), &op->metaBytes);
    if (!rc && st->bundleBegin) rc = uploadArray(&st->dBundleBegin, st->bundleBegin, (st->numBundles + 1) * 4, &op->metaBytes);      /* forward complex128 */
    for (uint64_t r = 0; r < st->numReduce && !rc; ++r) {
      BfReduce *rd = &st->reduce[r];
      rc = uploadArray(&rd->dRowInterval, rd->rowInterval, rd->numRows * 4, &op->metaBytes);
      if (!rc) rc = uploadArray(&rd->dIvBegin, rd->ivBegin, (rd->numIntervals + 1) * 4, &op->metaBytes);
      if (!rc) rc = uploadArray(&rd->dSrcBias, rd->srcBias, rd->numSrc * 8, &op->metaBytes);
    }
  }
  return rc;
}

static void dropPlanMirrors(BfPlan *plan) {
  for (uint64_t s = 0; s < plan->numStages && plan->stages; ++s) {
    BfStage *st = &plan->stages[s];
    free(st->pieceSrc); st->pieceSrc = NULL;
    free(st->pieces); st->pieces = NULL;
    free(st->items); st->items = NULL;
    free(st->bundleBegin); st->bundleBegin = NULL;
    free(st->pieceBuf); st->pieceBuf = NULL;
    free(st->itemBuf); st->itemBuf = NULL;
    for (uint64_t r = 0; r < st->numReduce; ++r) {
      free(st->reduce[r].rowInterval); st->reduce[r].rowInterval = NULL;
      free(st->reduce[r].ivBegin); st->reduce[r].ivBegin = NULL;
      free(st->reduce[r].srcBias); st->reduce[r].srcBias = NULL;
    }
  }
}


#ifdef BFHIP_EXPERIMENTAL      /* the dependency-driven launch lives in `make experimental` builds only (bfhip_experimental.hip) */
/* Flat index tables for the dependency-driven launch (bfFlowKernelC128): all stages' items in stage order with global
 * piece indices, pieces carrying the id of the vector they read, per item the vector it writes, per vector its number
 * of writers, and the counters.  Needs the host mirrors of the plan. */
static int buildFlow(BfhipOperator *op, int hostOnly) {
  BfPlan const *pl = &op->plan;
  uint64_t ni = 0, np = 0;
  for (uint64_t s = 0; s < pl->numStages; ++s) { ni += pl->stages[s].numItems; np += pl->stages[s].numPieces; }
  if (!ni || ni >= 0x7fffffffu || np >= 0xffffffffu) return 0;       /* nothing to run / too large for 32-bit tickets: staged launches */
  BfDevItem *items = malloc(ni * sizeof *items);
  BfDevPiece *pieces = malloc((np ? np : 1) * sizeof *pieces);
  uint32_t *itemOut = malloc(ni * 4);
  int rc = 0;
  if (!items || !pieces || !itemOut) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (flow tables)"); goto done; }
  uint64_t i0 = 0, p0 = 0;
  for (uint64_t s = 0; s < pl->numStages; ++s) {
    BfStage const *st = &pl->stages[s];
    for (uint64_t i = 0; i < st->numItems; ++i) {
      items[i0 + i] = st->items[i];
      items[i0 + i].pieceBegin = (uint32_t)(p0 + st->items[i].pieceBegin);
      itemOut[i0 + i] = st->itemBuf[i];
    }
    for (uint64_t k = 0; k < st->numPieces; ++k) {
      pieces[p0 + k] = st->pieces[k];
      pieces[p0 + k].ld = st->pieceBuf[k];          /* column-major complex pieces do not use `ld` */
      if (st->pieceBuf[k]) {                        /* the number of writers of that vector rides above the flag bits */
        if (pl->bufWriters[st->pieceBuf[k]] >= (1u << 24)) { rc = 0; goto done; }      /* (never: a vector has one writer per <= 64 of its rows) -> staged launches */
        pieces[p0 + k].flags |= pl->bufWriters[st->pieceBuf[k]] << 8;
      }
    }
    i0 += st->numItems; p0 += st->numPieces;
  }
  uint32_t maxW = 1;
  for (uint64_t b = 0; b < pl->numBufs; ++b) if (pl->bufWriters[b] > maxW) maxW = pl->bufWriters[b];
  {
    /* The launch drains iff the list, walked in ticket order by ONE worker, never waits: every vector a piece reads has
     * been written completely by items earlier in the list.  Checked here, once, on the host. */
    uint32_t *seen = calloc(pl->numBufs + 1, 4);
    if (!seen) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (flow tables)"); goto done; }
    for (uint64_t i = 0; i < ni && !rc; ++i) {
      for (uint32_t k = 0; k < items[i].numPieces; ++k) {
        uint32_t const dep = pieces[items[i].pieceBegin + k].ld;
        if (dep >= pl->numBufs || (dep && (dep < 2 || !pl->bufWriters[dep] || seen[dep] != pl->bufWriters[dep]))) {
          rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: item %llu reads vector %u before it is complete (%u of %u writers)", (unsigned long long)i, dep,
                         dep < pl->numBufs ? seen[dep] : 0, dep < pl->numBufs ? pl->bufWriters[dep] : 0);
          break;
        }
      }
      if (itemOut[i] >= pl->numBufs || itemOut[i] == 1) rc = bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "internal: item %llu writes vector %u", (unsigned long long)i, itemOut[i]);
      else if (itemOut[i]) ++seen[itemOut[i]];
    }
    free(seen);
    if (rc) goto done;
  }
  uint64_t const nb = pl->numBufs < 2 ? 2 : pl->numBufs;
  if (hostOnly) { op->flow = 1; goto done; }        /* plan-only operators: the tables are built and checked, nothing is uploaded */
  if ((rc = uploadArray(&op->dFlowItems, items, ni * sizeof *items, &op->metaBytes))) goto done;
  if ((rc = uploadArray(&op->dFlowPieces, pieces, (np ? np : 1) * sizeof *pieces, &op->metaBytes))) goto done;
  if ((rc = uploadArray(&op->dFlowItemOut, itemOut, ni * 4, &op->metaBytes))) goto done;
  {
    uint32_t *w = calloc(nb, 4);
    if (!w) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM (flow tables)"); goto done; }
    memcpy(w, pl->bufWriters, pl->numBufs * 4);
    rc = uploadArray(&op->dFlowWriters, w, nb * 4, &op->metaBytes);
    free(w);
    if (rc) goto done;
  }
  if ((rc = bfdevMalloc(&op->dFlowCounters, nb * 4))) goto done;
  if ((rc = bfdevMemset(op->dFlowCounters, 0, nb * 4))) goto done;
  if ((rc = bfdevFlowGrid(ni, &op->flowGrid))) goto done;
  op->flowNumItems = (uint32_t)ni; op->flowNumBufs = nb; op->flowMaxWriters = maxW;
  op->flowEpoch = 0; op->flowQueueBase = 0;
  op->flow = 1;
done:
  free(items); free(pieces); free(itemOut);
  return rc;
}
#endif

static int ensureTemp(BfhipOperator *op, uint32_t nrhs) {
  if (op->dTemp && op->tempRhs >= nrhs) return 0;
  bfdevFree(op->dTemp);
  op->dTemp = NULL;
  op->tempRhs = 0;
  uint64_t te = op->plan.tempElems > op->tplan.tempElems ? op->plan.tempElems : op->tplan.tempElems;
  int rc = bfdevMalloc(&op->dTemp, (size_t)te * nrhs * op->plan.elemSize);
  if (rc) return rc;
  op->tempRhs = nrhs;
  return 0;
}

/* The adjoint plan over the forward plan's packed leaves (BFHIP_FLAG_ADJOINT): index metadata only.  Needs the forward plan's
 * host mirrors.  Row-range / row-block shards: the transposed plan of the shard's own tasks (bfPlanBuild prunes the same way). */
static int buildSharedTplan(BfhipOperator *op, BfIr const *ir, BfPlanOptions const *po) {
  BfFwdPiece *fwd = NULL;
  uint64_t nf = 0;
  int rc = bfPlanFwdPieces(&op->plan, &fwd, &nf);
  if (rc) return rc;
  BfPlanOptions pt = *po;
  pt.itemsWanted = 0;
  pt.fwdPieces = fwd;
  pt.numFwdPieces = nf;
  pt.tCols = 0;      /* item width (16 or 64 columns of A) chosen stage by stage */
  rc = bfPlanBuild(ir, &pt, &op->tplan);
  free(fwd);
  if (!rc) { op->hasTplan = 1; op->packedT = 0; }
  return rc;
}

int bfhipCompileIrFill(BfIr *ir, BfhipOptions const *opts, BfFillFn fill, void *fillCtx, BfhipOperator **out) {
  BfhipOptions o;
  memset(&o, 0, sizeof o);
  o.device = -1;
  if (opts) {
    if (opts->structSize < BFHIP_OPTIONS_SIZE_V1) { bfIrFree(ir); return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipOptions.structSize too small"); }
    memcpy(&o, opts, opts->structSize < sizeof o ? opts->structSize : sizeof o);      /* fields a shorter (older) struct lacks stay 0 */
  }
  BfhipOperator *op = calloc(1, sizeof *op);
  if (!op) { bfIrFree(ir); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  int rc = 0;
  int prevDev = -1;
  int const planOnly = (o.flags & BFHIP_FLAG_PLAN_ONLY) != 0;
  BfIr irT;                      /* the transposed expression of a packed adjoint plan: released under `done` on every path */
  memset(&irT, 0, sizeof irT);
  op->flags = o.flags;
  op->seed = o.seed;
#ifndef BFHIP_EXPERIMENTAL
  if (o.flags & BFHIP_FLAG_FLOW) {
    rc = bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "BFHIP_FLAG_FLOW: this library was built without the experimental executors (make -C butterfly_amd/csrc experimental)");
    goto done;
  }
#endif
  if (!planOnly) {
    bfdevGetDevice(&prevDev);
    if ((rc = bfdevSetDevice(o.device))) goto done;
    if ((rc = bfdevGetDevice(&op->device))) goto done;
  }
  op->srcDtype = ir->dtype;
  BfPlanOptions po;
  memset(&po, 0, sizeof po);
  po.storeDtype = ir->dtype;
  if (o.demoteToF32) {
    if (ir->dtype != BFHIP_F64) { rc = bfhipFail(BFABI_ERROR_TYPE_ERROR, "demoteToF32 applies to real operands only"); goto done; }
    po.storeDtype = BFHIP_F32;
  }
  po.groupByInput = o.maxRhs >= 3;
#ifndef BF_MIN_CHUNK_ROWS
#define BF_MIN_CHUNK_ROWS 16     /* lower bound of the adaptive item height, in 16-byte row units (A/B builds: 8) */
#endif
  po.minChunkRows = o.maxRhs >= 3 ? 32 : BF_MIN_CHUNK_ROWS;   /* operators compiled for RHS blocks run on the matrix-core kernel */
  po.rowBlockBegin = o.rowBlockBegin;
  po.rowAlignBytes = 128;      /* rows of row-major pieces on 128-byte lines: the forward kernel gains 1 - 3 % on them, the transposed one 3 % */
  po.rowBlockEnd = o.rowBlockEnd;
  po.rowBegin = o.rowBegin;
  po.rowEnd = o.rowEnd;
  if ((rc = bfPlanBuild(ir, &po, &op->plan))) goto done;
  op->leafBytesAlgorithmic = op->plan.leafElems * op->plan.elemSize;
  /* The adjoint plan.  BFHIP_FLAG_ADJOINT_PACKED: a FORWARD plan of the transposed expression over its own packed copy of the
   * leaves (twice the leaf memory; A^T x then runs on the forward kernels at the forward rate).  Not with a caller-side
   * value builder (its values exist in the forward arena only) and not plan-only: those get the shared-leaf plan below. */
  /* (a row shard's adjoint is the shared-leaf plan as well: its input is the shard's rows of v, its transposed expression would have
   *  to be pruned by COLUMNS -- bfPlanBuild prunes the transposed task list instead, pruneToRowRangeT) */
  int packedT = (o.flags & BFHIP_FLAG_ADJOINT_PACKED) && !fill && !(po.rowEnd > 0 || po.rowBlockEnd > 0);
  if (packedT) {
    if ((rc = bfIrTransposed(ir, &irT))) goto done;
    BfPlanOptions pt = po;
    /* The block columns of a real (streamed) butterfly are long -- hundreds of leaves -- and the 1 MiB item cap leaves a stage of
     * the transposed expression with ~8000 items for 4096 wavefront slots: two uneven rounds.  Cutting them for >= 32768 items
     * per stage (N = 1M fp32: 8.66 -> 8.08 ms; 16384: 8.25, 65536: 8.15) costs nothing but a few more partial sums.  Complex
     * (fac_helm2) plans keep 4096: their stages are 10 GB, the cap binds either way and more items measured 1-2 % slower, as did
     * more items in the FORWARD plan of either operand (DESIGN_EXPERIMENTS.md section 10). */
    if (op->plan.dtype != BFHIP_C128) pt.itemsWanted = 32768;
    if ((rc = bfPlanBuild(&irT, &pt, &op->tplan))) goto done;
    op->hasTplan = 1;
    op->packedT = 1;
    if (planOnly) {          /* kept for bfhipPlanPackArenaT */
      op->irT = malloc(sizeof *op->irT);
      if (!op->irT) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
      *op->irT = irT;
      memset(&irT, 0, sizeof irT);
    }
  } else if (o.flags & (BFHIP_FLAG_ADJOINT | BFHIP_FLAG_ADJOINT_PACKED)) {
    if ((rc = buildSharedTplan(op, ir, &po))) goto done;
  }
  if (planOnly) {
#ifdef BFHIP_EXPERIMENTAL
    if (op->plan.dtype == BFHIP_C128 && op->plan.flowOk && (o.flags & BFHIP_FLAG_FLOW) && (rc = buildFlow(op, 1))) goto done;
#endif
    /* keep the IR (with its borrowed leaf pointers) for bfhipPlanPackArena */
    op->ir = malloc(sizeof *op->ir);
    if (!op->ir) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
    *op->ir = *ir;
    memset(ir, 0, sizeof *ir);
    *out = op;
    return 0;
  }

  /* + 256 B of slack: no kernel reads past the last piece by construction, the slack keeps a
   * future mistake there off the end of the mapping */
  if ((rc = bfdevMalloc(&op->dArena, (size_t)op->plan.arenaElems * op->plan.elemSize + BF_ARENA_SLACK))) goto done;
  if ((rc = uploadPlanMeta(op, &op->plan))) goto done;
  if (op->hasTplan && (rc = uploadPlanMeta(op, &op->tplan))) goto done;
  /* leaf values: computed on the device by the caller's builder, or packed / synthesized from the IR */
  if ((rc = fill ? fill(&op->plan, ir, op->dArena, fillCtx) : packLeaves(op, ir, o.seed, NULL))) goto done;
  if (packedT) {
    rc = bfdevMalloc(&op->dArenaT, (size_t)op->tplan.arenaElems * op->tplan.elemSize + BF_ARENA_SLACK);
    if (rc == BFABI_ERROR_MEMORY_ERROR) {
      /* no room for a second copy of the leaves (the flag doubles the leaf memory): the adjoint degrades to the shared-leaf plan
       * on the transposed kernels instead of failing the compile */
      op->dArenaT = NULL;
      freeDevicePlanOf(&op->tplan);
      bfPlanFree(&op->tplan);
      op->hasTplan = 0; op->packedT = 0; packedT = 0;
      if ((rc = buildSharedTplan(op, ir, &po))) goto done;
      if ((rc = uploadPlanMeta(op, &op->tplan))) goto done;
    } else {
      if (!rc) rc = packLeavesPlan(&op->tplan, op->dArenaT, &irT, o.seed, NULL);
      if (rc) goto done;
    }
  }
#ifdef BFHIP_EXPERIMENTAL
  {
    char const *envFlow = getenv("BFHIP_FLOW");         /* A/B switch for whole programs; BFHIP_FLAG_FLOW is the per-operator one */
    if (op->plan.dtype == BFHIP_C128 && op->plan.flowOk && ((o.flags & BFHIP_FLAG_FLOW) || (envFlow && envFlow[0] == '1')) && (rc = buildFlow(op, 0))) goto done;
  }
  /* the persistent ticket launch (BFHIP_PERSISTENT=1): every stage's counters exist before the first apply, so that an
   * apply never allocates or synchronises (a sharded step must not fail on ONE rank after its peers entered the collective) */
  if (op->plan.dtype == BFHIP_C128 && bfdevPersistentGrid()) {
    for (uint64_t s_ = 0; s_ < op->plan.numStages; ++s_) {
      BfStage *st_ = &op->plan.stages[s_];
      if ((rc = bfdevMalloc(&st_->dTickets, BF_TICKET_POOLS * BF_TICKET_STRIDE * 4))) goto done;
      if ((rc = bfdevMemset(st_->dTickets, 0, BF_TICKET_POOLS * BF_TICKET_STRIDE * 4))) goto done;
    }
    if ((rc = bfdevSync(NULL))) goto done;
  }
#endif
  /* host mirrors of the bulky per-piece arrays are no longer needed */
  dropPlanMirrors(&op->plan);
  dropPlanMirrors(&op->tplan);
  if ((rc = ensureTemp(op, o.maxRhs ? o.maxRhs : 1))) goto done;
  if ((rc = bfdevMalloc(&op->dZero, 4096))) goto done;
  if ((rc = bfdevMemset(op->dZero, 0, 4096))) goto done;
  if (op->flags & BFHIP_FLAG_PROFILE) {
    uint64_t S = op->plan.numStages;
    op->evStart = calloc(BF_EV_POOL * S, sizeof(void *));
    op->evStop = calloc(BF_EV_POOL * S, sizeof(void *));
    op->stageMs = calloc(S, sizeof(double));
    op->stageLaunches = calloc(S, sizeof(uint64_t));
    if (!op->evStart || !op->evStop || !op->stageMs || !op->stageLaunches) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
    for (uint64_t s = 0; s < BF_EV_POOL * S && !rc; ++s) {
      rc = bfdevEventCreate(&op->evStart[s]);
      if (!rc) rc = bfdevEventCreate(&op->evStop[s]);
    }
  }
done:
  bfIrFree(ir);
  bfIrFree(&irT);
  if (rc) { bfhipFree(&op); if (prevDev >= 0) bfdevSetDevice(prevDev); return rc; }
  if (prevDev >= 0 && o.device >= 0) bfdevSetDevice(prevDev);
  *out = op;
  return 0;
}

int bfhipCompile(void const *bfMat, BfhipOptions const *opts, BfhipOperator **out) {
  if (!bfMat || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  BfIr ir;
  int rc = bfIrFromBfMat(bfMat, &ir);
  if (rc) return rc;
  return bfhipCompileIrFill(&ir, opts, NULL, NULL, out);
}

int bfhipCompileDesc(BfhipDesc const *desc, BfhipOptions const *opts, BfhipOperator **out) {
  if (!desc || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  BfIr ir;
  int rc = bfIrFromDesc(desc, &ir);
  if (rc) return rc;
  return bfhipCompileIrFill(&ir, opts, NULL, NULL, out);
}

/* out[v] = leaf elements (sum of rows x cols over the dense leaves) under node v, for every node */
int bfhipDescSubtreeLeafElems(BfhipDesc const *desc, uint64_t *out) {
  if (!desc || !out || !desc->kind || !desc->rows || !desc->cols || !desc->childBegin) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  uint64_t const n = desc->numNodes;
  int ordered = 1;       /* children before parents (what the native layouts emit): one pass */
  for (uint64_t v = 0; v < n && ordered; ++v)
    for (uint64_t c = desc->childBegin[v]; c < desc->childBegin[v + 1]; ++c) if (desc->childNode[c] >= v) { ordered = 0; break; }
  if (ordered) {
    for (uint64_t v = 0; v < n; ++v) {
      uint64_t t = desc->kind[v] == BFHIP_NODE_DENSE ? desc->rows[v] * desc->cols[v] : 0;
      for (uint64_t c = desc->childBegin[v]; c < desc->childBegin[v + 1]; ++c) t += out[desc->childNode[c]];
      out[v] = t;
    }
    return 0;
  }
  /* any order: memoised depth-first walk with an explicit stack (node, next child) */
  uint8_t *done = calloc(n ? n : 1, 1);
  uint64_t *stack = malloc((2 * n + 2) * 8);
  if (!done || !stack) { free(done); free(stack); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  int rc = 0;
  for (uint64_t r = 0; r < n && !rc; ++r) {
    if (done[r]) continue;
    uint64_t sp = 0;
    stack[sp++] = r; stack[sp++] = desc->childBegin[r];
    out[r] = desc->kind[r] == BFHIP_NODE_DENSE ? desc->rows[r] * desc->cols[r] : 0;
    while (sp) {
      uint64_t const v = stack[sp - 2], c = stack[sp - 1];
      if (c == desc->childBegin[v + 1]) { done[v] = 1; sp -= 2; if (sp) out[stack[sp - 2]] += out[v]; continue; }
      stack[sp - 1] = c + 1;
      uint64_t const ch = desc->childNode[c];
      if (ch >= n) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "child index out of range"); break; }
      if (done[ch]) { out[v] += out[ch]; continue; }
      if (sp >= 2 * n) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "descriptor is not a tree (cycle)"); break; }
      out[ch] = desc->kind[ch] == BFHIP_NODE_DENSE ? desc->rows[ch] * desc->cols[ch] : 0;
      stack[sp++] = ch; stack[sp++] = desc->childBegin[ch];
    }
  }
  free(done); free(stack);
  return rc;
}

int bfhipRowPartition(BfhipDesc const *desc, uint32_t world, uint64_t *cuts, uint64_t *leafElems) {
  if (!desc || !cuts) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  BfIr ir;
  int rc = bfIrFromDesc(desc, &ir);
  if (rc) return rc;
  rc = bfPlanRowPartition(&ir, world, cuts, leafElems);
  bfIrFree(&ir);
  return rc;
}
int bfhipRowPartitionMat(void const *bfMat, uint32_t world, uint64_t *cuts, uint64_t *leafElems) {
  if (!bfMat || !cuts) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  BfIr ir;
  int rc = bfIrFromBfMat(bfMat, &ir);
  if (rc) return rc;
  rc = bfPlanRowPartition(&ir, world, cuts, leafElems);
  bfIrFree(&ir);
  return rc;
}

/* ---- profiling helpers ------------------------------------------------------ */
/* read back the events of applies [evHarvested, upTo): synchronizes on each (they complete in order) */
static int harvestEvents(BfhipOperator *op, uint64_t upTo) {
  uint64_t const S = op->plan.numStages;
  for (; op->evHarvested < upTo; ++op->evHarvested) {
    uint64_t const slot = op->evHarvested % BF_EV_POOL;
    for (uint64_t s = 0; s < (op->evFlow[slot] ? 1 : S); ++s) {
      float ms = 0;
      int rc = bfdevEventElapsed(op->evStart[slot * S + s], op->evStop[slot * S + s], &ms);
      if (rc) return rc;
      op->stageMs[s] += ms;
      op->stageLaunches[s] += 1;
    }
  }
  return 0;
}

static uint64_t stageBytes(BfhipOperator const *op, uint64_t s, uint32_t nrhs) {
  BfStage const *st = &op->plan.stages[s];
  return st->leafElems * op->plan.elemSize + (st->vecIn + st->vecOut) * (uint64_t)nrhs * op->plan.elemSize;
}

int bfhipGetStageProfile(BfhipOperator *op, double *ms, uint64_t *launches, uint64_t *bytes, int reset) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  if (!(op->flags & BFHIP_FLAG_PROFILE)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_PROFILE");
  int rc = harvestEvents(op, op->evIssued);
  if (rc) return rc;
  uint32_t const nr = op->lastNrhs ? op->lastNrhs : 1;
  int const oneLaunch = op->flow && nr < 2;        /* the whole apply is one launch: its time and bytes are reported under stage 0 */
  for (uint64_t s = 0; s < op->plan.numStages; ++s) {
    if (ms) ms[s] = op->stageMs[s];
    if (launches) launches[s] = op->stageLaunches[s];
    if (bytes) bytes[s] = oneLaunch ? 0 : stageBytes(op, s, nr);
    if (bytes && oneLaunch) bytes[0] += stageBytes(op, s, nr);
    if (reset) { op->stageMs[s] = 0; op->stageLaunches[s] = 0; }
  }
  return 0;
}

int bfhipSetProfileSampling(BfhipOperator *op, uint32_t every) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  if (!(op->flags & BFHIP_FLAG_PROFILE)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_PROFILE");
  op->profEvery = every;
  op->applyCount = 0;
  return 0;
}

/* ---- apply ------------------------------------------------------------------ */
static int runPlan(BfhipOperator *op, BfPlan *plan, void const *dX, size_t nrhs, void *dY, void *stream) {
  if (!op || !dX || !dY) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (nrhs == 0 || nrhs > 0xffffu) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "nrhs out of range");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator was compiled with BFHIP_FLAG_PLAN_ONLY: no device operator exists");
  int rc = 0;
  int prev = -1;
  bfdevGetDevice(&prev);
  if (prev != op->device && (rc = bfdevSetDevice(op->device))) return rc;
  if (op->tempRhs < nrhs) {
    /* growing the vector arena is not stream-ordered: drain first */
    if ((rc = bfdevSync(stream))) goto out;
    if ((rc = ensureTemp(op, (uint32_t)nrhs))) goto out;
  }
  int const prof = (op->flags & BFHIP_FLAG_PROFILE) != 0 && plan == &op->plan && (op->profEvery <= 1 || op->applyCount % op->profEvery == 0);
  if (plan == &op->plan) ++op->applyCount;
  /* timing never makes an apply wait for the previous one: only when all BF_EV_POOL event sets are in flight is the oldest read back */
  if (prof && op->evIssued - op->evHarvested >= BF_EV_POOL && (rc = harvestEvents(op, op->evIssued - BF_EV_POOL + 1))) goto out;
  uint64_t const evBase = prof ? (op->evIssued % BF_EV_POOL) * plan->numStages : 0;
  if (prof) op->evFlow[op->evIssued % BF_EV_POOL] = 0;
#ifdef BFHIP_EXPERIMENTAL
  if (op->flow && plan == &op->plan && nrhs < 2) {            /* (two and more right-hand sides: the matrix-core kernel, staged) */
    /* the whole plan as ONE dependency-driven launch (bfFlowKernelC128), then the reduce passes into y */
    uint32_t const perApply = op->flowNumItems + op->flowGrid * 4u;      /* tickets an apply consumes: every wavefront draws one past the end */
    if (op->flowEpoch >= 0x7fffffffu / op->flowMaxWriters - 1 || op->flowQueueBase >= 0xffffffffu - 2u * perApply) {
      if ((rc = bfdevMemsetAsync(op->dFlowCounters, 0, op->flowNumBufs * 4, stream))) goto out;      /* long before 32 bits wrap */
      op->flowEpoch = 0; op->flowQueueBase = 0;
    }
    BfFlowArgs fa;
    fa.arena = op->dArena; fa.items = op->dFlowItems; fa.pieces = op->dFlowPieces; fa.itemOut = op->dFlowItemOut; fa.writers = op->dFlowWriters;
    fa.counters = op->dFlowCounters; fa.numItems = op->flowNumItems; fa.nrhs = (uint32_t)nrhs; fa.epoch = ++op->flowEpoch;
    fa.queueBase = op->flowQueueBase; fa.gridWorkgroups = op->flowGrid; fa.x = dX; fa.y = dY; fa.temp = op->dTemp;
    op->flowQueueBase += perApply;
    if (prof) { op->evFlow[op->evIssued % BF_EV_POOL] = 1; if ((rc = bfdevEventRecord(op->evStart[evBase], stream))) goto out; }
    if ((rc = bfdevLaunchFlow(&fa, stream))) goto out;
    if (prof && (rc = bfdevEventRecord(op->evStop[evBase], stream))) goto out;
    {
      static int dbg = -1;
      if (dbg < 0) { char const *e = getenv("BFHIP_FLOW_DEBUG"); dbg = e && e[0] == '1'; }
      if (dbg) {      /* diagnostic: after the launch every vector's counter must stand at writers x epoch */
        if ((rc = bfdevSync(stream))) goto out;
        uint32_t *c = malloc(op->flowNumBufs * 4), *w = malloc(op->flowNumBufs * 4);
        if (c && w && !bfdevMemcpyD2H(c, op->dFlowCounters, op->flowNumBufs * 4) && !bfdevMemcpyD2H(w, op->dFlowWriters, op->flowNumBufs * 4)) {
          uint64_t bad = 0;
          for (uint64_t b = 2; b < op->flowNumBufs; ++b) if (c[b] != w[b] * op->flowEpoch) { if (bad++ < 8) fprintf(stderr, "bfhip flow: vector %llu counter %u, expected %u x %u\n", (unsigned long long)b, c[b], w[b], op->flowEpoch); }
          fprintf(stderr, "bfhip flow: epoch %u queue %u (base %u + %u items + %u waves) error %u, %llu of %llu counters off\n", op->flowEpoch, c[0], fa.queueBase, op->flowNumItems,
                  op->flowGrid * 4u, c[1], (unsigned long long)bad, (unsigned long long)op->flowNumBufs);
        }
        free(c); free(w);
      }
    }
    for (uint64_t s = 0; s < plan->numStages; ++s) {
      BfStage *st = &plan->stages[s];
      for (uint64_t r0 = 0; r0 < st->numReduce; r0 += 16) {
        BfReduceArgs ra[16];
        uint32_t const cnt = (uint32_t)(st->numReduce - r0 < 16 ? st->numReduce - r0 : 16);
        for (uint32_t r = 0; r < cnt; ++r) {
          BfReduce *rd = &st->reduce[r0 + r];
          ra[r].rowInterval = rd->dRowInterval; ra[r].ivBegin = rd->dIvBegin; ra[r].srcBias = rd->dSrcBias;
          ra[r].numRows = rd->numRows; ra[r].temp = op->dTemp; ra[r].nrhs = (uint32_t)nrhs; ra[r].dtype = plan->dtype;
          ra[r].longLists = rd->maxSrc >= 64; ra[r].pad = 0;
          ra[r].dest = dY;          /* flowOk: every reduce sums into y */
        }
        if ((rc = bfdevLaunchReduce(ra, cnt, stream))) goto out;
      }
    }
    if (prof) { ++op->evIssued; op->lastNrhs = (uint32_t)nrhs; }
    goto out;
  }
#endif
  for (uint64_t s = 0; s < plan->numStages; ++s) {
    BfStage *st = &plan->stages[s];
    BfLaunchArgs a;
    a.arena = (plan == &op->tplan && op->dArenaT) ? op->dArenaT : op->dArena; a.items = st->dItems; a.pieces = st->dPieces; a.numItems = st->numItems; a.firstSmall = st->firstSmall; a.numCoop = st->numCoop; a.numNarrow = st->numNarrow; a.numCoopNarrow = st->numCoopNarrow; a.maxRowsRest = st->maxRowsRest;
    a.x = dX; a.y = dY; a.temp = op->dTemp; a.zero = op->dZero; a.nrhs = (uint32_t)nrhs; a.dtype = plan->dtype; a.maxRows = st->maxRows;
    a.transposed = plan->transposed;
    a.tickets = NULL;
    a.exactComplex = (op->flags & BFHIP_FLAG_EXACT_COMPLEX) != 0; a.pad2 = 0;
    a.bundles = st->dBundleBegin; a.numBundles = st->numBundles;
    if (plan->dtype == BFHIP_C128 && !plan->transposed && nrhs < 2) a.tickets = st->dTickets;      /* NULL unless this is an EXPERIMENTAL build run with BFHIP_PERSISTENT=1 (allocated at compile time) */
    if (prof && (rc = bfdevEventRecord(op->evStart[evBase + s], stream))) goto out;
    if ((rc = bfdevLaunchStage(&a, stream))) goto out;
    if (prof && (rc = bfdevEventRecord(op->evStop[evBase + s], stream))) goto out;
    for (uint64_t r0 = 0; r0 < st->numReduce; r0 += 16) {
      BfReduceArgs ra[16];
      uint32_t const cnt = (uint32_t)(st->numReduce - r0 < 16 ? st->numReduce - r0 : 16);
      for (uint32_t r = 0; r < cnt; ++r) {
        BfReduce *rd = &st->reduce[r0 + r];
        ra[r].rowInterval = rd->dRowInterval; ra[r].ivBegin = rd->dIvBegin; ra[r].srcBias = rd->dSrcBias;
        ra[r].numRows = rd->numRows; ra[r].temp = op->dTemp; ra[r].nrhs = (uint32_t)nrhs; ra[r].dtype = plan->dtype;
        ra[r].longLists = rd->maxSrc >= 64; ra[r].pad = 0;
        ra[r].dest = rd->destSpace == BF_SPACE_Y ? dY : (void *)((char *)op->dTemp + rd->destOff * nrhs * plan->elemSize);
      }
      if ((rc = bfdevLaunchReduce(ra, cnt, stream))) goto out;
    }
  }
  if (prof) { ++op->evIssued; op->lastNrhs = (uint32_t)nrhs; }
out:
  /* every path hands the caller's device back */
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

/* grow the vector arena for `nrhs` right-hand sides now (not stream-ordered: synchronizes the device's default
 * stream), so that later applies of up to that many cannot fail on an allocation */
int bfhipOperatorReserveRhs(BfhipOperator *op, uint32_t nrhs) {
  if (!op || !nrhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator / zero nrhs");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator was compiled with BFHIP_FLAG_PLAN_ONLY: no device operator exists");
  if (op->tempRhs >= nrhs) return 0;
  int prev = -1, rc;
  bfdevGetDevice(&prev);
  if (prev != op->device && (rc = bfdevSetDevice(op->device))) return rc;
  rc = bfdevSync(NULL);
  if (!rc) rc = ensureTemp(op, nrhs);
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

/* Is the forward plan applied as ONE dependency-driven launch (one right-hand side), and has any of its waits ever
 * given up?  (They cannot, by construction; the flag exists so that a broken invariant shows up as an error instead
 * of a hung GPU.)  Synchronizes the device's default stream when the flag is read. */
int bfhipFlowStatus(BfhipOperator *op, uint32_t *enabled, uint32_t *waitGaveUp) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  if (enabled) *enabled = (uint32_t)op->flow;
  if (waitGaveUp) {
    *waitGaveUp = 0;
    if (op->flow && !(op->flags & BFHIP_FLAG_PLAN_ONLY)) {
      uint32_t two[2] = {0, 0};
      int prev = -1, rc;
      bfdevGetDevice(&prev);
      if (prev != op->device && (rc = bfdevSetDevice(op->device))) return rc;
      rc = bfdevMemcpyD2H(two, op->dFlowCounters, sizeof two);       /* [0] ticket queue, [1] error flag */
      if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
      if (rc) return rc;
      *waitGaveUp = two[1];
    }
  }
  return 0;
}

int bfhipApplyDevice(BfhipOperator *op, void const *dX, size_t nrhs, void *dY, void *stream) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  return runPlan(op, &op->plan, dX, nrhs, dY, stream);
}

int bfhipApplyTransposeDevice(BfhipOperator *op, void const *dX, size_t nrhs, void *dY, void *stream) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  if (!op->hasTplan) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_ADJOINT");
  return runPlan(op, &op->tplan, dX, nrhs, dY, stream);
}

/* ---- covariance products (the caller of the real path) ------------------------ */
static int covScratch(BfhipOperator *op) {
  if (op->dCov) return 0;
  /* two vectors of the longer side: [permuted input | A^T result] resp. [scaled input | A result] */
  uint64_t const big = op->plan.numRows > op->plan.numCols ? op->plan.numRows : op->plan.numCols;
  return bfdevMalloc(&op->dCov, 2 * big * op->plan.elemSize + 32);
}

int bfhipCovSampleDevice(BfhipOperator *op, void const *dGammaLam, uint64_t const *dRowPerm, void const *dW, void *dZ, void *stream) {
  if (!op || !dW || !dZ) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (op->plan.dtype == BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "covariance products are defined for real operators");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator was compiled with BFHIP_FLAG_PLAN_ONLY: no device operator exists");
  int rc, prev = -1;
  bfdevGetDevice(&prev);
  if (prev != op->device && (rc = bfdevSetDevice(op->device))) return rc;
  uint64_t const m = op->plan.numRows, n = op->plan.numCols, big = m > n ? m : n;
  size_t const es = op->plan.elemSize;
  if ((rc = covScratch(op))) goto out;
  char *t0 = op->dCov, *t1 = (char *)op->dCov + big * es;
  void const *xin = dW;
  if (dGammaLam) { if ((rc = bfdevScalePermute(t0, dW, dGammaLam, 1, NULL, n, op->plan.dtype, stream))) goto out; xin = t0; }
  if ((rc = runPlan(op, &op->plan, xin, 1, dRowPerm ? (void *)t1 : dZ, stream))) goto out;
  if (dRowPerm) rc = bfdevScalePermute(dZ, t1, NULL, 0, dRowPerm, m, op->plan.dtype, stream);
out:
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

int bfhipCovMatvecDevice(BfhipOperator *op, void const *dGammaLam, uint64_t const *dRowPerm, uint64_t const *dRevRowPerm, void const *dV, void *dZ, void *stream) {
  if (!op || !dV || !dZ) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (op->plan.dtype == BFHIP_C128) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "covariance products are defined for real operators");
  if (!op->hasTplan) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_ADJOINT");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator was compiled with BFHIP_FLAG_PLAN_ONLY: no device operator exists");
  int rc, prev = -1;
  bfdevGetDevice(&prev);
  if (prev != op->device && (rc = bfdevSetDevice(op->device))) return rc;
  uint64_t const m = op->plan.numRows, n = op->plan.numCols, big = m > n ? m : n;
  size_t const es = op->plan.elemSize;
  if ((rc = covScratch(op))) goto out;
  char *t0 = op->dCov, *t1 = (char *)op->dCov + big * es;
  void const *vin = dV;
  if (dRevRowPerm) { if ((rc = bfdevScalePermute(t0, dV, NULL, 0, dRevRowPerm, m, op->plan.dtype, stream))) goto out; vin = t0; }
  if ((rc = runPlan(op, &op->tplan, vin, 1, t1, stream))) goto out;                 /* tmp2 = Phi^T v */
  if (dGammaLam && (rc = bfdevScalePermute(t1, t1, dGammaLam, 2, NULL, n, op->plan.dtype, stream))) goto out;   /* GammaLam twice */
  if ((rc = runPlan(op, &op->plan, t1, 1, dRowPerm ? (void *)t0 : dZ, stream))) goto out;
  if (dRowPerm) rc = bfdevScalePermute(dZ, t0, NULL, 0, dRowPerm, m, op->plan.dtype, stream);
out:
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

/* The host-vector apply behind bfhipApply and every slot of the vtable shim (the path an unmodified reference caller takes:
 * bfSolveGMRES calls bfMatMul once per iteration, src/linalg.c:125,155).  What each of X and Y is decides what it costs:
 *   device memory of the operator's GPU, densely packed   used in place: no copy at all (the call is then bfhipApplyDevice + a wait);
 *   pinned / registered host memory, densely packed       DMA straight from / to the caller's buffer (hipHostMalloc'd, or registered with
 *                                                         bfhipHostRegister by a caller who knows its lifetime: a Krylov basis, say);
 *   anything else (pageable memory, ld != nrhs, fp32)     packed through the operator's own pinned staging buffer, then DMA.
 * The wait is on the apply's own stream, not on the device.  The library never registers a caller's buffer by itself: a cached
 * registration of memory the caller has since freed (glibc hands 4 MB vectors back to the kernel) would leave the GPU with a stale
 * mapping. */
static int applyHost(BfhipOperator *op, int transpose, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy) {
  if (!op || !X || !Y) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (transpose && !op->hasTplan) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_ADJOINT");
  if (nrhs == 0 || ldx < nrhs || ldy < nrhs) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad nrhs / leading dimension");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "operator was compiled with BFHIP_FLAG_PLAN_ONLY: no device operator exists");
  int rc;
  int prev = -1;
  bfdevGetDevice(&prev);
  if ((rc = bfdevSetDevice(op->device))) return rc;
  size_t es = op->plan.elemSize;
  size_t hostEs = op->srcDtype == BFHIP_C128 ? 16 : 8;      /* host side is always double precision */
  uint64_t n = transpose ? op->plan.numRows : op->plan.numCols, m = transpose ? op->plan.numCols : op->plan.numRows;
  uint64_t big = n > m ? n : m;
  int const same = es == hostEs;
  int const kx = same && ldx == nrhs ? bfdevPointerKind(X) : 0, ky = same && ldy == nrhs ? bfdevPointerKind(Y) : 0;
  void *hx = NULL, *hy = NULL;
  if (op->xyRhs < nrhs && (kx != 1 || ky != 1)) {
    bfdevFree(op->dX); bfdevFree(op->dY); op->dX = op->dY = NULL; op->xyRhs = 0;
    bfdevHostFreePinned(op->hX); bfdevHostFreePinned(op->hY); op->hX = op->hY = NULL;
    if ((rc = bfdevMalloc(&op->dX, big * nrhs * es))) goto out;
    if ((rc = bfdevMalloc(&op->dY, big * nrhs * es))) goto out;
    /* pinned staging: the copies run at PCIe rate instead of through the runtime's pageable bounce buffers */
    if ((rc = bfdevHostAllocPinned(&op->hX, (big * nrhs * es) != 0 ? big * nrhs * es : 16))) goto out;
    if ((rc = bfdevHostAllocPinned(&op->hY, (big * nrhs * es) != 0 ? big * nrhs * es : 16))) goto out;
    op->xyRhs = (uint32_t)nrhs;
  }
  hx = op->hX; hy = op->hY;
  void const *dXuse = op->dX;
  void *dYuse = ky == 1 ? Y : op->dY;
  if (kx == 1) dXuse = X;
  else if (kx == 2 || kx == 3) { if ((rc = bfdevMemcpyAnyAsync(op->dX, X, n * nrhs * es, NULL))) goto out; }
  else {
    /* pack to ld == nrhs (and demote if the operator computes in fp32) into the pinned buffer, in pieces: the DMA of a piece runs
     * while the CPU packs the next one (a 4 MB vector: ~0.1 ms of the ~0.25 ms its copy costs) */
    uint64_t const rowsPer = n > 4 * BF_HOST_PIECE_ROWS ? (n + 3) / 4 : n;
    for (uint64_t r0 = 0; r0 < n; r0 += rowsPer) {
      uint64_t const r1 = r0 + rowsPer < n ? r0 + rowsPer : n;
      if (same) {
        if (ldx == nrhs) memcpy((char *)hx + r0 * nrhs * es, (char const *)X + r0 * nrhs * es, (r1 - r0) * nrhs * es);
        else for (uint64_t i = r0; i < r1; ++i) memcpy((char *)hx + i * nrhs * es, (char const *)X + i * ldx * es, nrhs * es);
      } else {
        for (uint64_t i = r0; i < r1; ++i) for (size_t q = 0; q < nrhs; ++q) ((float *)hx)[i * nrhs + q] = (float)((double const *)X)[i * ldx + q];
      }
      if ((rc = bfdevMemcpyH2DAsync((char *)op->dX + r0 * nrhs * es, (char *)hx + r0 * nrhs * es, (r1 - r0) * nrhs * es, NULL))) goto out;
    }
  }
  if ((rc = runPlan(op, transpose ? &op->tplan : &op->plan, dXuse, nrhs, dYuse, NULL))) goto out;
  if (ky == 2 || ky == 3) { if ((rc = bfdevMemcpyAnyAsync(Y, op->dY, m * nrhs * es, NULL))) goto out; }
  if (ky != 0) { if ((rc = bfdevSync(NULL))) goto out; }
  else {
    /* the result comes back in pieces too: the CPU unpacks a piece while the DMA of the next is in flight (an event per piece) */
    uint64_t const rowsPer = m > 4 * BF_HOST_PIECE_ROWS ? (m + 3) / 4 : m;
    uint32_t np = 0;
    for (uint64_t r0 = 0; r0 < m; r0 += rowsPer, ++np) {
      uint64_t const r1 = r0 + rowsPer < m ? r0 + rowsPer : m;
      if ((rc = bfdevMemcpyD2HAsync((char *)hy + r0 * nrhs * es, (char *)op->dY + r0 * nrhs * es, (r1 - r0) * nrhs * es, NULL))) goto out;
      if (!op->evHost[np] && (rc = bfdevEventCreate(&op->evHost[np]))) goto out;
      if ((rc = bfdevEventRecord(op->evHost[np], NULL))) goto out;
    }
    np = 0;
    for (uint64_t r0 = 0; r0 < m; r0 += rowsPer, ++np) {
      uint64_t const r1 = r0 + rowsPer < m ? r0 + rowsPer : m;
      if ((rc = bfdevEventSync(op->evHost[np]))) goto out;
      if (same) {
        if (ldy == nrhs) memcpy((char *)Y + r0 * nrhs * es, (char *)hy + r0 * nrhs * es, (r1 - r0) * nrhs * es);
        else for (uint64_t i = r0; i < r1; ++i) memcpy((char *)Y + i * ldy * es, (char *)hy + i * nrhs * es, nrhs * es);
      } else {
        for (uint64_t i = r0; i < r1; ++i) for (size_t q = 0; q < nrhs; ++q) ((double *)Y)[i * ldy + q] = ((float *)hy)[i * nrhs + q];
      }
    }
    if (m == 0 && (rc = bfdevSync(NULL))) goto out;
  }
out:
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

/* A caller who knows the lifetime of its vectors (the Krylov basis of a solver, a right-hand side applied many times) registers
 * them once: bfhipApply / the shim's Mul then DMA straight from / to them instead of packing through the staging buffer. */
int bfhipHostRegister(void *p, size_t bytes) {
  if (!p || !bytes) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL pointer / zero bytes");
  return bfdevHostRegister(p, bytes);
}
int bfhipHostUnregister(void *p) {
  if (!p) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL pointer");
  return bfdevHostUnregister(p);
}

int bfhipApply(BfhipOperator *op, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy) {
  return applyHost(op, 0, X, ldx, nrhs, Y, ldy);
}
int bfhipApplyTranspose(BfhipOperator *op, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy) {
  return applyHost(op, 1, X, ldx, nrhs, Y, ldy);
}

/* ---- introspection ---------------------------------------------------------- */
int bfhipOperatorHasAdjoint(BfhipOperator const *op) { return op ? op->hasTplan : 0; }
uint32_t bfhipOperatorSrcDtype(BfhipOperator const *op) { return op ? op->srcDtype : BFHIP_C128; }
int bfhipOperatorDevice(BfhipOperator const *op) { return (!op || (op->flags & BFHIP_FLAG_PLAN_ONLY)) ? -1 : op->device; }
size_t bfhipGetNumRows(BfhipOperator const *op) { return op ? op->plan.numRows : 0; }
size_t bfhipGetNumCols(BfhipOperator const *op) { return op ? op->plan.numCols : 0; }
size_t bfhipNumBytes(BfhipOperator const *op) { return op ? op->plan.leafElems * (op->srcDtype == BFHIP_C128 ? 16 : 8) : 0; }

int bfhipGetStats(BfhipOperator const *op, BfhipStats *st) {
  if (!op || !st || st->structSize < sizeof(BfhipStats)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad stats struct");
  BfPlan const *pl = &op->plan;
  st->dtype = pl->dtype;
  st->numRows = pl->numRows; st->numCols = pl->numCols; st->numStages = pl->numStages;
  st->numLeaves = pl->numLeaves;
  st->numItems = 0; st->numPieces = 0; st->vecElemsRead = 0; st->vecElemsWritten = 0;
  for (uint64_t s = 0; s < pl->numStages; ++s) {
    st->numItems += pl->stages[s].numItems;
    st->numPieces += pl->stages[s].numPieces;
    st->vecElemsRead += pl->stages[s].vecIn;
    st->vecElemsWritten += pl->stages[s].vecOut;
  }
  st->leafElems = pl->leafElems;
  st->leafBytes = pl->leafElems * pl->elemSize;
  st->arenaBytes = pl->arenaElems * pl->elemSize + (op->dArenaT ? op->tplan.arenaElems * op->tplan.elemSize : 0);      /* both packed copies with BFHIP_FLAG_ADJOINT_PACKED */
  st->tempElems = pl->tempElems;
  st->metaBytes = op->metaBytes;
  return 0;
}

/* ---- plan inspection (BFHIP_FLAG_PLAN_ONLY) --------------------------------- */
static int needPlanOnly(BfhipOperator const *op) {
  if (!op) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator");
  if (!(op->flags & BFHIP_FLAG_PLAN_ONLY)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "plan inspection needs BFHIP_FLAG_PLAN_ONLY");
  return 0;
}
int bfhipPlanGetInfo(BfhipOperator const *op, BfhipPlanInfo *info) {
  int rc = needPlanOnly(op);
  if (rc) return rc;
  if (!info || info->structSize < sizeof *info) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad info struct");
  BfPlan const *pl = &op->plan;
  info->dtype = pl->dtype; info->elemSize = pl->elemSize; info->epl = pl->epl; info->xcap = pl->xcap;
  info->numRows = pl->numRows; info->numCols = pl->numCols; info->numStages = pl->numStages;
  info->arenaElems = pl->arenaElems; info->tempElems = pl->tempElems;
  info->numStagesT = op->hasTplan ? op->tplan.numStages : 0;
  info->tempElemsT = op->hasTplan ? op->tplan.tempElems : 0;
  info->reserved = op->packedT ? 1u : 0u;                     /* 1: the adjoint plan is a forward plan over its own arena ... */
  info->arenaElemsT = op->packedT ? op->tplan.arenaElems : 0;      /* ... of this many elements */
  return 0;
}
int bfhipPlanGetStage(BfhipOperator const *op, uint64_t stage, BfhipStageView *v) {
  int rc = needPlanOnly(op);
  if (rc) return rc;
  BfPlan const *pl = &op->plan;
  if (stage >= pl->numStages && op->hasTplan) { stage -= pl->numStages; pl = &op->tplan; }
  if (!v || v->structSize < offsetof(BfhipStageView, numBundles) || stage >= pl->numStages) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad stage view request");
  BfStage const *st = &pl->stages[stage];
  v->numItems = st->numItems; v->numPieces = st->numPieces; v->numReduce = st->numReduce;
  v->items = st->items; v->pieces = st->pieces;
  if (v->structSize >= sizeof *v) { v->numBundles = st->numBundles; v->bundleBegin = st->bundleBegin; }
  return 0;
}
int bfhipPlanGetReduce(BfhipOperator const *op, uint64_t stage, uint64_t index, BfhipReduceView *v) {
  int rc = needPlanOnly(op);
  if (rc) return rc;
  BfPlan const *pl = &op->plan;
  if (stage >= pl->numStages && op->hasTplan) { stage -= pl->numStages; pl = &op->tplan; }
  if (!v || v->structSize < sizeof *v || stage >= pl->numStages || index >= pl->stages[stage].numReduce)
    return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad reduce view request");
  BfReduce const *rd = &pl->stages[stage].reduce[index];
  v->destIsY = rd->destSpace == BF_SPACE_Y; v->destOff = rd->destOff; v->numRows = rd->numRows;
  v->numIntervals = rd->numIntervals; v->numSrc = rd->numSrc;
  v->rowInterval = rd->rowInterval; v->ivBegin = rd->ivBegin; v->srcBias = rd->srcBias;
  return 0;
}
int bfhipPlanPackArena(BfhipOperator const *op, void *dst) {
  int rc = needPlanOnly(op);
  if (rc) return rc;
  if (!dst) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL destination");
  return packLeaves(op, op->ir, op->seed, dst);
}
/* the second arena of a BFHIP_FLAG_ADJOINT_PACKED plan (arenaElemsT elements): the leaves of the transposed expression */
int bfhipPlanPackArenaT(BfhipOperator const *op, void *dst) {
  int rc = needPlanOnly(op);
  if (rc) return rc;
  if (!dst) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL destination");
  if (!op->packedT || !op->irT) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "operator was not compiled with BFHIP_FLAG_ADJOINT_PACKED");
  return packLeavesPlan(&op->tplan, NULL, op->irT, op->seed, dst);
}


/* =============================================================================
 * Serialization: the flattened device layout is the
 * natural on-disk form of a compiled operator.  The reference's bfMatDump is
 * write-only and lacks a complex dense payload (src/mat.c:67-73,
 * src/mat_dense_complex.c:173-222); here Save / Load round-trip the operator
 * exactly (bit-identical applies) without the BfMat graph or a rebuild.
 * File: "BFHIPOP1" | header | per plan: per stage {counts, items, pieces,
 * reduces} | leaf arena bytes.  Little-endian, same-architecture format.
 * ============================================================================= */
#define BFHIP_FILE_MAGIC "BFHIPOP1"

typedef struct FileHeader {
  char magic[8];
  uint32_t version, dtype, srcDtype, elemSize, epl, xcap, hasTplan, reserved;
  uint64_t numRows, numCols, arenaElems, leafElems, numLeaves, leafBytesAlgorithmic;
} FileHeader;

typedef struct FilePlanHeader { uint64_t numStages, tempElems, numRows, numCols; uint32_t maxItemRows, transposed; } FilePlanHeader;
typedef struct FileStageHeader { uint64_t numItems, numPieces, leafElems, vecIn, vecOut, numReduce; uint32_t maxRows, reserved; } FileStageHeader;
typedef struct FileReduceHeader { uint64_t destOff, numRows, numIntervals, numSrc; uint32_t destSpace, reserved; } FileReduceHeader;

static int writeAll(FILE *fp, void const *p, size_t n) { return n == 0 || fwrite(p, 1, n, fp) == n ? 0 : bfhipFail(BFABI_ERROR_FILE_ERROR, "short write"); }
static int readAll(FILE *fp, void *p, size_t n) { return n == 0 || fread(p, 1, n, fp) == n ? 0 : bfhipFail(BFABI_ERROR_FILE_ERROR, "short read / truncated file"); }

static int writeDeviceArray(FILE *fp, void const *d, size_t bytes) {
  if (!bytes) return 0;
  size_t const chunk = (size_t)64 << 20;
  void *h = malloc(bytes < chunk ? bytes : chunk);
  if (!h) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  int rc = 0;
  for (size_t off = 0; off < bytes && !rc; off += chunk) {
    size_t n = bytes - off < chunk ? bytes - off : chunk;
    rc = bfdevMemcpyD2H(h, (char const *)d + off, n);
    if (!rc) rc = writeAll(fp, h, n);
  }
  free(h);
  return rc;
}
static int readDeviceArray(FILE *fp, void **d, size_t bytes, uint64_t *meta) {
  int rc = bfdevMalloc(d, bytes + BF_ARENA_SLACK);
  if (rc || !bytes) return rc;
  if (meta) *meta += bytes;
  size_t const chunk = (size_t)64 << 20;
  void *h = malloc(bytes < chunk ? bytes : chunk);
  if (!h) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  for (size_t off = 0; off < bytes && !rc; off += chunk) {
    size_t n = bytes - off < chunk ? bytes - off : chunk;
    rc = readAll(fp, h, n);
    if (!rc) rc = bfdevMemcpyH2D((char *)*d + off, h, n);
  }
  free(h);
  return rc;
}

static int savePlan(FILE *fp, BfPlan const *pl) {
  FilePlanHeader ph = {pl->numStages, pl->tempElems, pl->numRows, pl->numCols, pl->maxItemRows, (uint32_t)pl->transposed};
  int rc = writeAll(fp, &ph, sizeof ph);
  for (uint64_t s = 0; s < pl->numStages && !rc; ++s) {
    BfStage const *st = &pl->stages[s];
    FileStageHeader sh = {st->numItems, st->numPieces, st->leafElems, st->vecIn, st->vecOut, st->numReduce, st->maxRows, 0};
    rc = writeAll(fp, &sh, sizeof sh);
    if (!rc) rc = writeDeviceArray(fp, st->dItems, st->numItems * sizeof(BfDevItem));
    if (!rc) rc = writeDeviceArray(fp, st->dPieces, st->numPieces * sizeof(BfDevPiece));
    for (uint64_t r = 0; r < st->numReduce && !rc; ++r) {
      BfReduce const *rd = &st->reduce[r];
      FileReduceHeader rh = {rd->destOff, rd->numRows, rd->numIntervals, rd->numSrc, rd->destSpace, 0};
      rc = writeAll(fp, &rh, sizeof rh);
      if (!rc) rc = writeDeviceArray(fp, rd->dRowInterval, rd->numRows * 4);
      if (!rc) rc = writeDeviceArray(fp, rd->dIvBegin, (rd->numIntervals + 1) * 4);
      if (!rc) rc = writeDeviceArray(fp, rd->dSrcBias, rd->numSrc * 8);
    }
  }
  return rc;
}

/* index tables are small (0.03 % of the operand): read whole, validate on the host, then upload */
static int readMetaArray(FILE *fp, void **d, void **h, size_t bytes, uint64_t *meta) {
  *h = NULL;
  int rc = bfdevMalloc(d, bytes);
  if (rc || !bytes) return rc;
  if (meta) *meta += bytes;
  *h = malloc(bytes);
  if (!*h) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  if ((rc = readAll(fp, *h, bytes))) return rc;
  return bfdevMemcpyH2D(*d, *h, bytes);
}

/* A file is untrusted input: every offset the kernels will dereference is checked against the
 * sizes in the header before the operator is accepted (a truncated or corrupt file must not turn
 * into device out-of-bounds accesses). */
/* offset + extent <= len without wrapping: offsets and extents come straight from the file as 64-bit values,
 * and `off + ext > len` accepts off = 2^64 - 16 (the sum wraps to a small value) */
static int fitsIn(uint64_t off, uint64_t ext, uint64_t len) { return off <= len && ext <= len - off; }
/* a * b + c, saturating at UINT64_MAX */
static uint64_t mulAddSat(uint64_t a, uint64_t b, uint64_t c) {
  uint64_t r;
  if (__builtin_mul_overflow(a, b, &r) || __builtin_add_overflow(r, c, &r)) return UINT64_MAX;
  return r;
}

static int validateStage(BfPlan const *pl, uint64_t arenaElems, BfStage const *st, BfDevItem const *items, BfDevPiece const *pieces) {
  uint64_t const inX = pl->numCols, outY = pl->numRows, temp = pl->tempElems;
  for (uint64_t i = 0; i < st->numItems; ++i) {
    BfDevItem const *it = &items[i];
    uint32_t const mr = it->mrFlags & 0xffffu;
    uint64_t const outLen = (it->mrFlags & BF_ITEM_OUT_Y) ? outY : temp;
    if (!mr || mr > pl->maxItemRows || mr > st->maxRows || (it->mrFlags & ~(0xffffu | BF_ITEM_OUT_Y | BF_ITEM_ROWMAJOR | BF_ITEM_MERGED | BF_ITEM_SMALL | BF_ITEM_TNARROW)) ||
        ((it->mrFlags & BF_ITEM_TNARROW) != 0) != (i < st->numNarrow) || ((it->mrFlags & BF_ITEM_TNARROW) && (!pl->transposed || mr > 16)) ||
        ((it->mrFlags & BF_ITEM_SMALL) != 0) != (i >= st->firstSmall) ||
        ((it->mrFlags & BF_ITEM_ROWMAJOR) && (pl->transposed || pl->dtype == BFHIP_C128 || mr > 2 * pl->epl)) ||
        (pl->transposed && mr > 64) ||       /* bfStageKernelT: at most 64 columns of A per item */
        !fitsIn(it->outOff, mr, outLen) || !fitsIn(it->pieceBegin, it->numPieces, st->numPieces))
      return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: item %llu out of bounds", (unsigned long long)i);
    uint32_t const mrPad = (mr + pl->epl - 1) / pl->epl * pl->epl;
    if (it->mrFlags & (BF_ITEM_MERGED | BF_ITEM_SMALL)) {     /* the kernel reads the dense pieces as one block from the first one's offset */
      int const sm = (it->mrFlags & BF_ITEM_SMALL) != 0;
      uint64_t next = 0, dense = 0;
      int const smRm = sm && (it->mrFlags & BF_ITEM_ROWMAJOR);      /* small items: row-major pieces, no contiguity promise */
      int badm = pl->transposed || pl->dtype == BFHIP_C128 || (!sm && (it->mrFlags & BF_ITEM_ROWMAJOR)) || it->numPieces > (sm ? BF_SMALL_PIECES : 64u) ||
                 (sm && (mr > 2 * pl->epl || !(it->mrFlags & BF_ITEM_ROWMAJOR)));
      for (uint32_t k = 0; k < it->numPieces && !badm; ++k) {
        BfDevPiece const *pc = &pieces[it->pieceBegin + k];
        if (pc->flags & BF_PIECE_IDENTITY) continue;
        if (!smRm && dense && pc->dataOff != next) badm = 1;
        next = mulAddSat(mrPad, pc->ncols, pc->dataOff); dense += pc->ncols;
      }
      if (badm || (!dense && !sm) || dense > (sm ? BF_SMALL_COLS : BF_MERGE_COLS))
        return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: merged item %llu is not one block", (unsigned long long)i);
    }
    for (uint32_t k = 0; k < it->numPieces; ++k) {
      BfDevPiece const *pc = &pieces[it->pieceBegin + k];
      uint64_t const inLen = (pc->flags & BF_PIECE_IN_X) ? inX : temp;
      int bad = (pc->flags & ~(BF_PIECE_IN_X | BF_PIECE_IDENTITY | BF_PIECE_ROWMAJOR)) != 0;
      int const rm = (pc->flags & BF_PIECE_ROWMAJOR) != 0;
      if (!(pc->flags & BF_PIECE_IDENTITY) && !pl->transposed && rm != ((it->mrFlags & BF_ITEM_ROWMAJOR) != 0)) bad = 1;
      if (pc->flags & BF_PIECE_IDENTITY) bad |= !fitsIn(pc->inOff, mr, inLen);
      else if (pl->transposed && rm)      /* rows of a row-major forward piece: ncols rows, mr columns from dataOff */
        bad |= !pc->ld || pc->ld % pl->epl || pc->dataOff % pl->epl || pl->dtype == BFHIP_C128 || !pc->ncols ||
               !fitsIn(pc->dataOff, mulAddSat(pc->ncols - 1, pc->ld, (mr + pl->epl - 1) / pl->epl * pl->epl), arenaElems) ||
               !fitsIn(pc->inOff, pc->ncols, inLen);
      else if (rm)
        bad |= !pc->ncols || pc->ld % pl->epl || pc->ld < pc->ncols || pc->dataOff % pl->epl ||   /* x is read from global memory: no xcap */
               !fitsIn(pc->dataOff, mulAddSat(mr, pc->ld, 0), arenaElems) || !fitsIn(pc->inOff, pc->ncols, inLen);
      else if (pl->transposed)
        bad |= !pc->ld || pc->ld % pl->epl || pc->ncols > pc->ld || pc->dataOff % pl->epl ||
               !fitsIn(pc->dataOff, mulAddSat(mr - 1, pc->ld, (pc->ncols + pl->epl - 1) / pl->epl * pl->epl), arenaElems) ||
               !fitsIn(pc->inOff, pc->ncols, inLen);
      else
        bad |= !pc->ncols || pc->ncols > pl->xcap || pc->dataOff % pl->epl ||
               !fitsIn(pc->dataOff, mulAddSat(mrPad, pc->ncols, 0), arenaElems) || !fitsIn(pc->inOff, pc->ncols, inLen);
      if (bad) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: piece %u of item %llu out of bounds", k, (unsigned long long)i);
    }
  }
  return 0;
}

static int validateReduce(BfPlan const *pl, BfReduce const *rd, uint32_t const *rowInterval, uint32_t const *ivBegin, int64_t const *srcBias) {
  uint64_t const destLen = rd->destSpace == BF_SPACE_Y ? pl->numRows : pl->tempElems;
  if ((rd->destSpace != BF_SPACE_Y && rd->destSpace != BF_SPACE_TEMP) || !fitsIn(rd->destOff, rd->numRows, destLen) || rd->numIntervals > rd->numRows + 1)
    return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: reduce destination out of bounds");
  if (ivBegin[0] != 0 || ivBegin[rd->numIntervals] > rd->numSrc) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: reduce interval table");
  for (uint64_t i = 0; i < rd->numIntervals; ++i)
    if (ivBegin[i + 1] < ivBegin[i]) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: reduce interval table not monotone");
  for (uint64_t r = 0; r < rd->numRows; ++r) {
    uint32_t const iv = rowInterval[r];
    if (iv == BF_REDUCE_SKIP) continue;
    if (iv >= rd->numIntervals) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: reduce row %llu", (unsigned long long)r);
    for (uint32_t k = ivBegin[iv]; k < ivBegin[iv + 1]; ++k) {
      int64_t src;
      if (__builtin_add_overflow(srcBias[k], (int64_t)r, &src) || src < 0 || (uint64_t)src >= pl->tempElems) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file: reduce source out of bounds");
    }
  }
  return 0;
}

static int loadPlan(FILE *fp, BfhipOperator *op, BfPlan *pl, FileHeader const *fh, uint64_t arenaElems) {
  FilePlanHeader ph;
  int rc = readAll(fp, &ph, sizeof ph);
  if (rc) return rc;
  if (ph.numStages > (1u << 20) || ph.tempElems >= 0xffffffffu) return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt plan header");
  pl->dtype = fh->dtype; pl->elemSize = fh->elemSize; pl->epl = fh->epl; pl->xcap = fh->xcap;
  pl->maxItemRows = ph.maxItemRows; pl->transposed = (int)ph.transposed;
  pl->numRows = ph.numRows; pl->numCols = ph.numCols; pl->numStages = ph.numStages; pl->tempElems = ph.tempElems;
  pl->arenaElems = pl->transposed ? 0 : arenaElems;
  pl->leafElems = fh->leafElems; pl->numLeaves = fh->numLeaves;
  if (pl->epl != 16 / pl->elemSize || pl->xcap != 256 || pl->maxItemRows > 64 * pl->epl || (pl->transposed && pl->maxItemRows > 128))
    return bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt plan header (lane granule / piece width / item height)");
  pl->stages = calloc(ph.numStages ? ph.numStages : 1, sizeof(BfStage));
  if (!pl->stages) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  for (uint64_t s = 0; s < pl->numStages && !rc; ++s) {
    BfStage *st = &pl->stages[s];
    FileStageHeader sh;
    if ((rc = readAll(fp, &sh, sizeof sh))) break;
    if (sh.numItems > 0xffffffffu || sh.numPieces > 0xffffffffu || sh.numReduce > (1u << 20)) { rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt stage header"); break; }
    st->numItems = sh.numItems; st->numPieces = sh.numPieces; st->leafElems = sh.leafElems; st->vecIn = sh.vecIn; st->vecOut = sh.vecOut;
    st->maxRows = sh.maxRows;
    void *hItems = NULL, *hPieces = NULL;
    rc = readMetaArray(fp, &st->dItems, &hItems, st->numItems * sizeof(BfDevItem), &op->metaBytes);
    if (!rc) rc = readMetaArray(fp, &st->dPieces, &hPieces, st->numPieces * sizeof(BfDevPiece), &op->metaBytes);
    if (!rc) {       /* small items are the tail of the list (validateStage checks that they are nowhere else) */
      st->firstSmall = st->numItems;
      while (st->firstSmall && (((BfDevItem const *)hItems)[st->firstSmall - 1].mrFlags & BF_ITEM_SMALL)) --st->firstSmall;
      st->numNarrow = 0;
      while (st->numNarrow < st->numItems && (((BfDevItem const *)hItems)[st->numNarrow].mrFlags & BF_ITEM_TNARROW)) ++st->numNarrow;
      rc = validateStage(pl, arenaElems, st, hItems, hPieces);
      if (!rc && pl->transposed) {
        st->maxRowsRest = 0;
        for (uint64_t i = st->numNarrow; i < st->numItems; ++i) {
          uint32_t const mr = ((BfDevItem const *)hItems)[i].mrFlags & 0xffffu;
          if (mr > st->maxRowsRest) st->maxRowsRest = mr;
        }
        st->numCoopNarrow = bfPlanCountCoop(hItems, hPieces, st->numNarrow, pl->elemSize);
        st->numCoop = bfPlanCountCoop((BfDevItem const *)hItems + st->numNarrow, hPieces, st->numItems - st->numNarrow, pl->elemSize);
      }
      if (!rc && pl->dtype == BFHIP_C128 && !pl->transposed && st->numItems) {
        uint32_t *bb = NULL;
        rc = bfPlanBundles(hItems, hPieces, st->numItems, &bb, &st->numBundles);
        if (!rc) rc = uploadArray(&st->dBundleBegin, bb, (st->numBundles + 1) * 4, &op->metaBytes);
        free(bb);
      }
    }
    free(hItems); free(hPieces);
    if (!rc && sh.numReduce) {
      st->reduce = calloc(sh.numReduce, sizeof(BfReduce));
      if (!st->reduce) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    }
    for (uint64_t r = 0; r < sh.numReduce && !rc; ++r) {
      BfReduce *rd = &st->reduce[r];
      FileReduceHeader rh;
      if ((rc = readAll(fp, &rh, sizeof rh))) break;
      st->numReduce = r + 1;
      rd->destOff = rh.destOff; rd->numRows = rh.numRows; rd->numIntervals = rh.numIntervals; rd->numSrc = rh.numSrc; rd->destSpace = rh.destSpace;
      if (rd->numRows >= 0xffffffffu || rd->numIntervals >= 0xffffffffu || rd->numSrc >= 0xffffffffu) { rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt reduce header"); break; }
      void *hRow = NULL, *hIv = NULL, *hBias = NULL;
      rc = readMetaArray(fp, &rd->dRowInterval, &hRow, rd->numRows * 4, &op->metaBytes);
      if (!rc) rc = readMetaArray(fp, &rd->dIvBegin, &hIv, (rd->numIntervals + 1) * 4, &op->metaBytes);
      if (!rc) rc = readMetaArray(fp, &rd->dSrcBias, &hBias, rd->numSrc * 8, &op->metaBytes);
      if (!rc) rc = validateReduce(pl, rd, hRow, hIv, hBias);
      if (!rc) {
        uint32_t const *iv = hIv;
        rd->maxSrc = 0;
        for (uint64_t i = 0; i < rd->numIntervals; ++i) if (iv[i + 1] - iv[i] > rd->maxSrc) rd->maxSrc = iv[i + 1] - iv[i];
      }
      free(hRow); free(hIv); free(hBias);
    }
  }
  return rc;
}

int bfhipSave(BfhipOperator *op, char const *path) {
  if (!op || !path) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  if (op->flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "a plan-only operator has no device data to save");
  int prev = -1;
  bfdevGetDevice(&prev);
  int rc = bfdevSetDevice(op->device);
  if (rc) return rc;
  if ((rc = bfdevSync(NULL))) return rc;
  FILE *fp = fopen(path, "wb");
  if (!fp) return bfhipFail(BFABI_ERROR_FILE_ERROR, "cannot open %s for writing", path);
  FileHeader fh;
  memset(&fh, 0, sizeof fh);
  memcpy(fh.magic, BFHIP_FILE_MAGIC, 8);
  fh.version = 1; fh.dtype = op->plan.dtype; fh.srcDtype = op->srcDtype; fh.elemSize = op->plan.elemSize; fh.epl = op->plan.epl;
  fh.xcap = op->plan.xcap; fh.hasTplan = (uint32_t)op->hasTplan;
  fh.reserved = op->dArenaT ? 1u : 0u;        /* bit 0: the adjoint plan is a forward plan of the transposed expression over a second arena (BFHIP_FLAG_ADJOINT_PACKED) */
  fh.numRows = op->plan.numRows; fh.numCols = op->plan.numCols; fh.arenaElems = op->plan.arenaElems;
  fh.leafElems = op->plan.leafElems; fh.numLeaves = op->plan.numLeaves; fh.leafBytesAlgorithmic = op->leafBytesAlgorithmic;
  rc = writeAll(fp, &fh, sizeof fh);
  uint64_t const arenaElemsT = op->dArenaT ? op->tplan.arenaElems : 0;
  if (!rc && op->dArenaT) rc = writeAll(fp, &arenaElemsT, sizeof arenaElemsT);
  if (!rc) rc = savePlan(fp, &op->plan);
  if (!rc && op->hasTplan) rc = savePlan(fp, &op->tplan);
  if (!rc) rc = writeDeviceArray(fp, op->dArena, (size_t)op->plan.arenaElems * op->plan.elemSize);
  if (!rc && op->dArenaT) rc = writeDeviceArray(fp, op->dArenaT, (size_t)arenaElemsT * op->plan.elemSize);
  if (fclose(fp) != 0 && !rc) rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "error closing %s", path);
  if (prev >= 0 && prev != op->device) bfdevSetDevice(prev);
  return rc;
}

int bfhipLoad(char const *path, BfhipOptions const *opts, BfhipOperator **out) {
  if (!path || !out) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL argument");
  *out = NULL;
  BfhipOptions o;
  memset(&o, 0, sizeof o);
  o.device = -1;
  if (opts) {
    if (opts->structSize < BFHIP_OPTIONS_SIZE_V1) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipOptions.structSize too small");
    memcpy(&o, opts, opts->structSize < sizeof o ? opts->structSize : sizeof o);
  }
  if (o.flags & BFHIP_FLAG_PLAN_ONLY) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "cannot load as plan-only");
  FILE *fp = fopen(path, "rb");
  if (!fp) return bfhipFail(BFABI_ERROR_FILE_ERROR, "cannot open %s", path);
  FileHeader fh;
  int rc = readAll(fp, &fh, sizeof fh);
  if (!rc && (memcmp(fh.magic, BFHIP_FILE_MAGIC, 8) != 0 || fh.version != 1 || fh.dtype > BFHIP_F32 ||
              fh.elemSize != (fh.dtype == BFHIP_C128 ? 16u : fh.dtype == BFHIP_F64 ? 8u : 4u)))
    rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "%s is not a bfhip operator file (bad magic / version / dtype)", path);
  if (rc) { fclose(fp); return rc; }
  BfhipOperator *op = calloc(1, sizeof *op);
  if (!op) { fclose(fp); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  int prevDev = -1;
  bfdevGetDevice(&prevDev);
  op->flags = o.flags & ~(uint32_t)(BFHIP_FLAG_ADJOINT | BFHIP_FLAG_ADJOINT_PACKED);
  op->srcDtype = fh.srcDtype;
  op->leafBytesAlgorithmic = fh.leafBytesAlgorithmic;
  if ((rc = bfdevSetDevice(o.device))) goto done;
  if ((rc = bfdevGetDevice(&op->device))) goto done;
  uint64_t arenaElemsT = 0;
  int const packedT = (fh.reserved & 1u) != 0;
  if (fh.reserved & ~1u) { rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file (header flags)"); goto done; }
  if (packedT && (!fh.hasTplan || (rc = readAll(fp, &arenaElemsT, sizeof arenaElemsT)))) { if (!rc) rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file (packed adjoint without a plan)"); goto done; }
  if ((rc = loadPlan(fp, op, &op->plan, &fh, fh.arenaElems))) goto done;
  if (fh.hasTplan) {
    if ((rc = loadPlan(fp, op, &op->tplan, &fh, packedT ? arenaElemsT : fh.arenaElems))) goto done;
    /* a packed adjoint plan is a FORWARD plan over its own arena, a shared one a transposed plan over the forward arena */
    if ((op->tplan.transposed != 0) == packedT) { rc = bfhipFail(BFABI_ERROR_FILE_ERROR, "corrupt operator file (adjoint plan kind)"); goto done; }
    op->hasTplan = 1;
    op->packedT = packedT;
    op->flags |= packedT ? BFHIP_FLAG_ADJOINT_PACKED : BFHIP_FLAG_ADJOINT;
  }
  if ((rc = readDeviceArray(fp, &op->dArena, (size_t)fh.arenaElems * fh.elemSize, NULL))) goto done;
  if (packedT && (rc = readDeviceArray(fp, &op->dArenaT, (size_t)arenaElemsT * fh.elemSize, NULL))) goto done;
  if ((rc = ensureTemp(op, o.maxRhs ? o.maxRhs : 1))) goto done;
  if ((rc = bfdevMalloc(&op->dZero, 4096))) goto done;
  if ((rc = bfdevMemset(op->dZero, 0, 4096))) goto done;
  if (op->flags & BFHIP_FLAG_PROFILE) {
    uint64_t S = op->plan.numStages;
    op->evStart = calloc(BF_EV_POOL * S, sizeof(void *)); op->evStop = calloc(BF_EV_POOL * S, sizeof(void *));
    op->stageMs = calloc(S, sizeof(double)); op->stageLaunches = calloc(S, sizeof(uint64_t));
    if (!op->evStart || !op->evStop || !op->stageMs || !op->stageLaunches) { rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); goto done; }
    for (uint64_t s = 0; s < BF_EV_POOL * S && !rc; ++s) { rc = bfdevEventCreate(&op->evStart[s]); if (!rc) rc = bfdevEventCreate(&op->evStop[s]); }
  }
done:
  fclose(fp);
  if (rc) { bfhipFree(&op); if (prevDev >= 0) bfdevSetDevice(prevDev); return rc; }
  if (prevDev >= 0 && o.device >= 0) bfdevSetDevice(prevDev);
  *out = op;
  return 0;
}

/* =============================================================================
 * BfMat vtable shim
 * ============================================================================= */
typedef struct BfhipMat {
  BfAbiMat super;             /* must be first: this IS a BfMat */
  BfhipOperator *op;          /* the operator; with `sh`: this rank's share of it (shape queries go to `sh`) */
  int ownsOperator;
  int transposed;             /* bfMatTranspose has been applied an odd number of times: Mul / MulVec run the adjoint plan */
  struct BfhipSharded *sh;    /* bfhipShardedMatNew: applies are the sharded step (every rank's host calls with the same vectors) */
  int ownsSharded;
} BfhipMat;

/* rows / columns of the operator the object stands for (untransposed), and the host-vector apply behind every slot */
static uint64_t shimOpRows(BfhipMat const *s) { return s->sh ? bfhipShardedGetNumRows(s->sh) : bfhipGetNumRows(s->op); }
static uint64_t shimOpCols(BfhipMat const *s) { return s->sh ? bfhipShardedGetNumCols(s->sh) : bfhipGetNumCols(s->op); }
static int applyHost(BfhipOperator *op, int transpose, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy);
static int shimApplyHost(BfhipMat const *s, int transpose, void const *X, size_t ldx, size_t nrhs, void *Y, size_t ldy) {
  return s->sh ? bfhipShardedApplyHost(s->sh, transpose, X, ldx, nrhs, Y, ldy) : applyHost(s->op, transpose, X, ldx, nrhs, Y, ldy);
}

/* Failures surface the way the reference's own Mul failures do: the global error code is set
 * (bfSetError, src/error.c:20-24) and NULL is returned (the RAISE_ERROR / BF_ERROR_END idiom, e.g.
 * src/mat_product.c:404-405).  libbfhip does not link the reference; when the host process has
 * it loaded its bfSetError is found at run time.  Note that bfSetError asserts on a non-zero code
 * (src/error.c:21), so in a reference build with assertions a failed Mul is as fatal as the
 * reference's own BF_DIE() paths; bfhipSetErrorForwarding(0) keeps failures to NULL +
 * bfhipLastErrorMessage(). */
#include <dlfcn.h>
static int forwardErrors = 1;
void bfhipSetErrorForwarding(int on) { forwardErrors = on; }
static void shimRaise(int code) {
  if (!forwardErrors || !code) return;
  static void (*setError)(int);
  static int looked;
  if (!looked) { looked = 1; setError = (void (*)(int))dlsym(RTLD_DEFAULT, "bfSetError"); }
  if (setError) setError(code);
}
#define SHIM_FAIL(code, ...) do { shimRaise(bfhipFail((code), __VA_ARGS__)); return NULL; } while (0)

/* shape of what the object currently stands for: A, or A^T after bfMatTranspose (the reference's transposed product
 * answers with its reversed, transposed factors' shapes: src/mat_product.c:146-192, 409-420) */
static size_t shimGetNumRows(BfAbiMat const *m) { BfhipMat const *s = (BfhipMat const *)m; return s->transposed ? shimOpCols(s) : shimOpRows(s); }
static size_t shimGetNumCols(BfAbiMat const *m) { BfhipMat const *s = (BfhipMat const *)m; return s->transposed ? shimOpRows(s) : shimOpCols(s); }
static int shimGetType(BfAbiMat const *m) { (void)m; return BFABI_TYPE_MAT_FUNC; }
static size_t shimNumBytes(BfAbiMat const *m) { return bfhipNumBytes(((BfhipMat const *)m)->op); }
static void shimDelete(BfAbiMat **m) {
  if (!m || !*m) return;
  BfhipMat *s = (BfhipMat *)*m;
  /* a view never owns the operator (bfMatDenseRealDeinit skips the payload of a view the same way,
   * src/mat_dense_real.c:1667-1672) */
  if (s->ownsSharded && s->sh && !(s->super.props & BFABI_MAT_PROPS_VIEW)) bfhipShardedFree(&s->sh);
  if (s->ownsOperator && !(s->super.props & BFABI_MAT_PROPS_VIEW)) bfhipFree(&s->op);
  free(s);
  *m = NULL;
}
/* GetView: a shallow copy flagged VIEW, what every reference type returns (e.g.
 * bfMatDenseRealGetView, src/mat_dense_real.c:67-85).  bfMatBlockDenseGetBlockConst calls it on
 * every sub-block of a BlockDense on each Mul (src/mat_block_dense.c:1043-1061, via bfMatGet
 * with BF_POLICY_VIEW), so a shim placed INSIDE a reference container needs it. */
static BfAbiMat *shimGetView(BfAbiMat *m) {
  BfhipMat *v = malloc(sizeof *v);
  if (!v) SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  *v = *(BfhipMat *)m;
  v->super.props |= BFABI_MAT_PROPS_VIEW;
  return &v->super;
}

/* Y = A X for a reference dense RHS; the result is allocated through the
 * RHS's own EmptyLike slot so the reference owns and frees it
 * (bfMatBlockCooMul does the same with ZerosLike, mat_block_coo.c:401). */
static void *shimMulImpl(void const *rhsV, BfhipMat const *self, int transpose) {
  BfAbiMat const *rhs = rhsV;
  BfhipOperator *op = self ? self->op : NULL;
  if (!op || !rhs || !rhs->vtbl) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operand");
  if (transpose && !op->hasTplan) SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "Mul on a transposed operator needs BFHIP_FLAG_ADJOINT");
  uint64_t const inLen = transpose ? shimOpRows(self) : shimOpCols(self), outLen = transpose ? shimOpCols(self) : shimOpRows(self);
  BfAbiGetTypeFn getType = (BfAbiGetTypeFn)rhs->vtbl->slot[BFABI_SLOT_GetType];
  if (!getType || getType(rhs) != BFABI_TYPE_MAT_DENSE_COMPLEX || op->srcDtype != BFHIP_C128)
    /* same restriction as bfMatDenseComplexMul's switch (mat_dense_complex.c:1036-1047) */
    SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "Mul needs a complex operator and a BfMatDenseComplex right-hand side");
  if (rhs->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "transposed right-hand side");
  BfAbiMatDenseComplex const *x = (BfAbiMatDenseComplex const *)rhs;
  if (rhs->numRows != inLen)
    SHIM_FAIL(BFABI_ERROR_INCOMPATIBLE_SHAPES, "operator has %llu columns, right-hand side %llu rows", (unsigned long long)inLen, (unsigned long long)rhs->numRows);
  BfAbiLikeFn emptyLike = (BfAbiLikeFn)rhs->vtbl->slot[BFABI_SLOT_EmptyLike];
  if (!emptyLike) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "right-hand side has no EmptyLike");
  /* a column-strided right-hand side (a view of every k-th column, a column range of a wider matrix:
   * bfMatDenseComplexGetColRange leaves colStride as it is, src/mat_dense_complex.c:648-672) is gathered into a packed
   * copy first -- cblas_zgemm in the reference cannot take it either (it passes ldb = rowStride and assumes unit column
   * stride, :1754), so this is more than the reference does, not less */
  /* A transposed COMPLEX object multiplies as its conjugate transpose, as in the reference: bfMatTranspose ends in
   * bfMatDenseComplexTranspose = bfMatConjTrans on every dense leaf (src/mat_dense_complex.c:1475-1478, src/mat.c:359-362) and
   * getCblasTranspose maps the flags to CblasConjTrans (:27-35).  A^H X = conj(A^T conj(X)): the right-hand side is
   * conjugated into the packed copy, the result in place. */
  void *packed = NULL;
  void const *xdata = x->data;
  size_t xld = x->rowStride;
  if (x->colStride != 1 || transpose) {
    if (x->colStride == 0) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "right-hand side with colStride 0");
    size_t const nr = rhs->numRows, nc = rhs->numCols;
    packed = malloc((nr && nc ? nr * nc : 1) * 16);
    if (!packed) SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "host OOM");
    for (size_t i = 0; i < nr; ++i)
      for (size_t q = 0; q < nc; ++q) {
        double const *e = (double const *)((char const *)x->data + (i * x->rowStride + q * x->colStride) * 16);
        double *d = (double *)((char *)packed + (i * nc + q) * 16);
        d[0] = e[0]; d[1] = transpose ? -e[1] : e[1];
      }
    xdata = packed; xld = nc;
  }
  BfAbiMat *res = emptyLike(rhs, outLen, rhs->numCols);
  if (!res) { free(packed); SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "EmptyLike failed"); }
  BfAbiMatDenseComplex *y = (BfAbiMatDenseComplex *)res;
  int rc;
  if (y->colStride != 1) rc = bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "EmptyLike returned a result with colStride != 1");
  else rc = shimApplyHost(self, transpose, xdata, xld, rhs->numCols, y->data, y->rowStride);
  if (!rc && transpose)
    for (size_t i = 0; i < outLen; ++i)
      for (size_t q = 0; q < rhs->numCols; ++q) ((double *)y->data)[2 * (i * y->rowStride + q) + 1] *= -1.0;
  free(packed);
  if (rc) {
    BfAbiDeleteFn del = (BfAbiDeleteFn)res->vtbl->slot[BFABI_SLOT_Delete];
    if (del) del(&res);
    shimRaise(rc);
    return NULL;
  }
  return res;
}

void *bfhipMatMulFunc(void const *rhsV, void *opV) {
  BfhipMat tmp;
  memset(&tmp, 0, sizeof tmp);
  tmp.op = opV;
  return shimMulImpl(rhsV, &tmp, 0);
}

static BfAbiMat *shimMul(BfAbiMat const *lhs, BfAbiMat const *rhs) {
  return shimMulImpl(rhs, (BfhipMat const *)lhs, ((BfhipMat const *)lhs)->transposed);
}

/* bfMatRmul(A_hip, X) = X A (slot 44, src/mat.c:195-197; bfMatProductRmul walks the factors in order, src/mat_product.c:282-310, down
 * to bfMatDenseComplexRmul's one zgemm, src/mat_dense_complex.c:1075-1133 -- a dense complex `otherMat` only, :1125-1133).
 * X A = (A^T X^T)^T: the adjoint plan applied to the rows of X as right-hand sides.  X^T is gathered into a packed copy (any row /
 * column stride of X), the result is scattered into a matrix allocated through X's EmptyLike.  After bfMatTranspose the object
 * stands for A^H (shimMulImpl): X A^H = conj(conj(X) A^T) = conj((A conj(X)^T)^T), the FORWARD plan between two conjugations. */
static BfAbiMat *shimRmul(BfAbiMat const *lhs, BfAbiMat const *other) {
  BfhipMat const *self = (BfhipMat const *)lhs;
  BfhipOperator *op = self ? self->op : NULL;
  if (!op || !other || !other->vtbl) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operand");
  int const conj = self->transposed, transpose = !self->transposed;          /* which plan runs */
  if (transpose && !op->hasTplan) SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "Rmul needs an operator compiled with BFHIP_FLAG_ADJOINT");
  BfAbiGetTypeFn getType = (BfAbiGetTypeFn)other->vtbl->slot[BFABI_SLOT_GetType];
  if (!getType || getType(other) != BFABI_TYPE_MAT_DENSE_COMPLEX || op->srcDtype != BFHIP_C128)
    SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "Rmul needs a complex operator and a BfMatDenseComplex left operand");
  if (other->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "transposed left operand");
  /* rows / columns of what the object stands for */
  uint64_t const rows = self->transposed ? shimOpCols(self) : shimOpRows(self), cols = self->transposed ? shimOpRows(self) : shimOpCols(self);
  size_t const m = other->numRows, k = other->numCols;
  if (k != rows) SHIM_FAIL(BFABI_ERROR_INCOMPATIBLE_SHAPES, "operator has %llu rows, left operand %llu columns", (unsigned long long)rows, (unsigned long long)k);
  BfAbiLikeFn emptyLike = (BfAbiLikeFn)other->vtbl->slot[BFABI_SLOT_EmptyLike];
  if (!emptyLike) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "left operand has no EmptyLike");
  BfAbiMatDenseComplex const *x = (BfAbiMatDenseComplex const *)other;
  double *xt = malloc((k && m ? k * m : 1) * 16), *zt = malloc((cols && m ? cols * m : 1) * 16);
  if (!xt || !zt) { free(xt); free(zt); SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  for (size_t q = 0; q < m; ++q)
    for (size_t i = 0; i < k; ++i) {
      double const *e = (double const *)((char const *)x->data + (q * x->rowStride + i * x->colStride) * 16);
      xt[2 * (i * m + q)] = e[0]; xt[2 * (i * m + q) + 1] = conj ? -e[1] : e[1];
    }
  int rc = shimApplyHost(self, transpose, xt, m, m, zt, m);          /* (k x m) -> (cols x m) */
  free(xt);
  if (rc) { free(zt); shimRaise(rc); return NULL; }
  BfAbiMat *res = emptyLike(other, m, cols);
  if (!res) { free(zt); SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "EmptyLike failed"); }
  BfAbiMatDenseComplex *y = (BfAbiMatDenseComplex *)res;
  for (size_t q = 0; q < m; ++q)
    for (size_t j = 0; j < cols; ++j) {
      double *d = (double *)((char *)y->data + (q * y->rowStride + j * y->colStride) * 16);
      d[0] = zt[2 * (j * m + q)]; d[1] = conj ? -zt[2 * (j * m + q) + 1] : zt[2 * (j * m + q) + 1];
    }
  free(zt);
  return res;
}

/* bfMatTranspose (slot 63, src/mat.c:271-273): in place, as bfMatProductTranspose reverses and transposes its factors
 * (src/mat_product.c:409-420).  The adjoint plan over the same packed leaves exists already (BFHIP_FLAG_ADJOINT), so the
 * object only changes which of its two plans Mul / MulVec / RmulVec run and what GetNumRows / GetNumCols answer; twice
 * is the identity.  For a REAL operator that is the transpose; a COMPLEX one multiplies as its conjugate transpose
 * afterwards, as the reference's does (its dense complex leaves transpose by bfMatConjTrans: shimMulImpl).  The slot returns nothing: without an
 * adjoint plan the reference's error state is raised (NOT_IMPLEMENTED) and the object is left as it was. */
static void shimTranspose(BfAbiMat *m) {
  BfhipMat *s = (BfhipMat *)m;
  if (!s->op->hasTplan) { shimRaise(bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "Transpose needs an operator compiled with BFHIP_FLAG_ADJOINT")); return; }
  s->transposed = !s->transposed;
  size_t const r = s->super.numRows;
  s->super.numRows = s->super.numCols;
  s->super.numCols = r;
}

/* y = A x (transpose == 0) or z = x^T A as a vector (bfMatRmulVec) for a reference BfVecReal; real
 * operators only: the block types reject complex vectors (mat_block_coo.c:438-444).  The result
 * is sized by the operator, as the reference's containers size theirs (bfVecRealNewWithValue(m, 0)
 * in src/mat_block_dense.c:574-590 and src/mat_block_coo.c:427-444; n for RmulVec, :696-712):
 * a malloc'd BfVecReal {vtbl, props NONE, size, stride 1, malloc'd data} carrying the ARGUMENT's
 * vtable, so that the reference's bfVecDelete -> bfVecRealDeinitAndDealloc frees data and struct with
 * free() (src/vec_real.c:661-676, src/mem.c:65-67).  Rectangular operators are the normal case:
 * cov_matvec applies the N x m operator Phi both ways (examples/covariance/lbo_cov.c:48-60). */
static BfAbiVec *shimApplyVec(BfAbiMat const *lhs, BfAbiVec const *vec, int rmul) {
  BfhipOperator *op = ((BfhipMat const *)lhs)->op;
  char const *const what = rmul ? "RmulVec" : "MulVec";
  int const transpose = rmul != ((BfhipMat const *)lhs)->transposed;          /* x^T (A^T) = (A x)^T */
  if (!vec || !vec->vtbl) SHIM_FAIL(BFABI_ERROR_INVALID_ARGUMENTS, "%s: NULL vector", what);
  BfAbiVecGetTypeFn getType = (BfAbiVecGetTypeFn)vec->vtbl->slot[BFABI_VSLOT_GetType];
  if (!getType || getType(vec) != BFABI_TYPE_VEC_REAL || op->srcDtype != BFHIP_F64)
    SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "%s needs a real operator and a BfVecReal", what);
  if (transpose && !op->hasTplan) SHIM_FAIL(BFABI_ERROR_NOT_IMPLEMENTED, "%s needs an operator compiled with BFHIP_FLAG_ADJOINT", what);
  BfhipMat const *self = (BfhipMat const *)lhs;
  uint64_t const inLen = transpose ? shimOpRows(self) : shimOpCols(self);
  uint64_t const outLen = transpose ? shimOpCols(self) : shimOpRows(self);
  if (vec->size != inLen)
    SHIM_FAIL(BFABI_ERROR_INCOMPATIBLE_SHAPES, "%s: operator expects %llu entries, vector has %llu", what, (unsigned long long)inLen, (unsigned long long)vec->size);
  BfAbiVecReal const *x = (BfAbiVecReal const *)vec;
  BfAbiVecReal *y = malloc(sizeof *y);
  double *data = malloc((outLen ? outLen : 1) * sizeof(double));
  if (!y || !data) { free(y); free(data); SHIM_FAIL(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  y->super.vtbl = vec->vtbl;
  y->super.props = BFABI_VEC_PROPS_NONE;
  y->super.size = outLen;
  y->stride = 1;
  y->data = data;
  int rc = shimApplyHost(self, transpose, x->data, x->stride, 1, y->data, 1);
  if (rc) { free(data); free(y); shimRaise(rc); return NULL; }
  return &y->super;
}
static BfAbiVec *shimMulVec(BfAbiMat const *lhs, BfAbiVec const *vec) { return shimApplyVec(lhs, vec, 0); }
static BfAbiVec *shimRmulVec(BfAbiMat const *lhs, BfAbiVec const *vec) { return shimApplyVec(lhs, vec, 1); }

static BfAbiMatVtable ShimVtable = {.slot = {
  [BFABI_SLOT_GetView] = (void *)shimGetView,
  [BFABI_SLOT_RmulVec] = (void *)shimRmulVec,
  [BFABI_SLOT_Delete] = (void *)shimDelete,
  [BFABI_SLOT_GetType] = (void *)shimGetType,
  [BFABI_SLOT_NumBytes] = (void *)shimNumBytes,
  [BFABI_SLOT_GetNumRows] = (void *)shimGetNumRows,
  [BFABI_SLOT_GetNumCols] = (void *)shimGetNumCols,
  [BFABI_SLOT_Mul] = (void *)shimMul,
  [BFABI_SLOT_Rmul] = (void *)shimRmul,
  [BFABI_SLOT_MulVec] = (void *)shimMulVec,
  [BFABI_SLOT_Transpose] = (void *)shimTranspose,
}};

void *bfhipMatNew(BfhipOperator *op, int ownsOperator) {
  if (!op) { bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL operator"); return NULL; }
  BfhipMat *m = calloc(1, sizeof *m);
  if (!m) { bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); return NULL; }
  m->super.vtbl = &ShimVtable;
  m->super.props = BFABI_MAT_PROPS_NONE;
  m->super.numRows = op->plan.numRows;
  m->super.numCols = op->plan.numCols;
  m->op = op;
  m->ownsOperator = ownsOperator;
  return m;
}

/* the same object over a sharded operator: shapes are the whole operator's, applies are the sharded step */
void *bfhipShardedMatNew(struct BfhipSharded *sh, int ownsSharded) {
  if (!sh) { bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL sharded operator"); return NULL; }
  BfhipMat *m = calloc(1, sizeof *m);
  if (!m) { bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); return NULL; }
  m->super.vtbl = &ShimVtable;
  m->super.props = BFABI_MAT_PROPS_NONE;
  m->super.numRows = bfhipShardedGetNumRows(sh);
  m->super.numCols = bfhipShardedGetNumCols(sh);
  m->op = bfhipShardedOperator(sh);
  m->sh = sh;
  m->ownsSharded = ownsSharded;
  return m;
}
