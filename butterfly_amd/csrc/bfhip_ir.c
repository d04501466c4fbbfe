/* bfhip_ir.c -- operand intake.  Either copies a flat BfhipDesc or walks a
 * reference BfMat object graph (layouts: include/bfhip_abi.h) into the same
 * owned IR.  Read-only with respect to the caller's objects; leaf value
 * pointers are borrowed until bfhipCompile* returns (everything is copied to
 * the device then), so the handle never retains pointers into A.
 *
 * The walker replaces the reference's per-call recursive dispatch
 * (src/mat.c:183 -> mat_product.c:211, mat_block_{dense,diag,coo}.c Mul) with
 * a one-time traversal; what each container means is taken from:
 *   BlockDiag  blocks k at (rowOffset[k], colOffset[k])      mat_block_diag.c:387-399
 *   BlockCoo   blocks k at (rowOffset[rowInd[k]], colOffset[colInd[k]])  mat_block_coo.c:404-418
 *   BlockDense block[i*numCols + j] at (rowOffset[i], colOffset[j])      mat_block_dense.c:534-566
 *   Product    factorArr[0..L-1], applied right-to-left       mat_product.c:211-245
 */
#include "bfhip_internal.h"
#include "../../include/bfhip_abi.h"

#include <stdlib.h>
#include <string.h>

void bfIrFree(BfIr *ir) {
  if (!ir) return;
  free(ir->kind); free(ir->rows); free(ir->cols); free(ir->childBegin);
  free(ir->childNode); free(ir->childRow0); free(ir->childCol0);
  free(ir->leafData); free(ir->leafRowStride); free(ir->leafColStride); free(ir->leafReal);
  free(ir->synthBase); free(ir->topRowBlock); free(ir->depth); free(ir->patches);
  memset(ir, 0, sizeof *ir);
}

static void *dupArray(void const *src, uint64_t count, size_t elt) {
  void *p = malloc((count ? count : 1) * elt);
  if (p && src && count) memcpy(p, src, count * elt);
  return p;
}

static int cmpPatch(void const *pa, void const *pb);

/* The expression of A^T over the same leaf values: every node's shape swapped, a block's children placed at (col0, row0), a
 * product's factors in reverse order, host-valued leaves read with their strides swapped, synthetic leaves marked
 * (transposedView) so that element (i, j) of a transposed leaf is the stream value of element (j, i) of the original.  What
 * bfMatProductTranspose does to a product in place (reference src/mat_product.c:409-420), as a second expression. */
int bfIrTransposed(BfIr const *src, BfIr *dst) {
  memset(dst, 0, sizeof *dst);
  uint64_t const n = src->numNodes, nc = src->numChildren;
  dst->dtype = src->dtype; dst->numNodes = n; dst->numChildren = nc; dst->root = src->root;
  dst->capNodes = n; dst->capChildren = nc;
  dst->transposedView = !src->transposedView;
  dst->kind = dupArray(src->kind, n, 1);
  dst->rows = dupArray(src->cols, n, 8);
  dst->cols = dupArray(src->rows, n, 8);
  dst->childBegin = dupArray(src->childBegin, n + 1, 8);
  dst->childNode = dupArray(src->childNode, nc, 8);
  dst->childRow0 = dupArray(src->childCol0, nc, 8);
  dst->childCol0 = dupArray(src->childRow0, nc, 8);
  dst->leafData = dupArray(src->leafData, n, sizeof(void *));
  dst->leafRowStride = dupArray(src->leafColStride, n, 8);
  dst->leafColStride = dupArray(src->leafRowStride, n, 8);
  dst->leafReal = dupArray(src->leafReal, n, 1);
  dst->synthBase = dupArray(src->synthBase, n, 8);
  dst->depth = dupArray(src->depth, n, 4);
  dst->patches = dupArray(src->patches, src->numPatches, sizeof(BfIrPatch));
  dst->numPatches = dst->capPatches = src->numPatches;
  if (!dst->kind || !dst->rows || !dst->cols || !dst->childBegin || !dst->childNode || !dst->childRow0 || !dst->childCol0 || !dst->leafData ||
      !dst->leafRowStride || !dst->leafColStride || !dst->leafReal || !dst->synthBase || !dst->depth || !dst->patches) {
    bfIrFree(dst);
    return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
  }
  for (uint64_t v = 0; v < n; ++v) {
    if (src->kind[v] != BFHIP_NODE_PRODUCT) continue;
    uint64_t const b = src->childBegin[v], e = src->childBegin[v + 1];
    for (uint64_t c = b; c < e; ++c) {          /* factors in reverse order; a product's children sit at (0, 0) */
      dst->childNode[c] = src->childNode[e - 1 - (c - b)];
      dst->childRow0[c] = src->childCol0[e - 1 - (c - b)];
      dst->childCol0[c] = src->childRow0[e - 1 - (c - b)];
    }
  }
  for (uint64_t k = 0; k < dst->numPatches; ++k) { uint32_t const r = dst->patches[k].row; dst->patches[k].row = dst->patches[k].col; dst->patches[k].col = r; }
  if (dst->numPatches) qsort(dst->patches, dst->numPatches, sizeof(BfIrPatch), cmpPatch);
  return 0;
}

int bfIrFromDesc(BfhipDesc const *d, BfIr *ir) {
  memset(ir, 0, sizeof *ir);
  if (!d || d->structSize < sizeof(BfhipDesc)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfhipDesc.structSize too small");
  if (d->dtype != BFHIP_C128 && d->dtype != BFHIP_F64) return bfhipFail(BFABI_ERROR_TYPE_ERROR, "descriptor dtype must be C128 or F64");
  if (!d->numNodes || d->root >= d->numNodes) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "bad root / empty descriptor");
  if (!d->kind || !d->rows || !d->cols || !d->childBegin) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "descriptor arrays missing");
  uint64_t n = d->numNodes, nc = d->childBegin[n];
  /* the CSR must be well formed before anything indexes the child arrays through it */
  if (d->childBegin[0] != 0) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "childBegin[0] must be 0");
  for (uint64_t i = 0; i < n; ++i)
    if (d->childBegin[i + 1] < d->childBegin[i] || d->childBegin[i + 1] > nc)
      return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "childBegin is not a non-decreasing prefix array (node %llu)", (unsigned long long)i);
  if (nc && (!d->childNode || !d->childRow0 || !d->childCol0)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "descriptor child arrays missing");
  ir->dtype = d->dtype;
  ir->numNodes = n;
  ir->numChildren = nc;
  ir->root = d->root;
  ir->kind = dupArray(d->kind, n, 1);
  ir->rows = dupArray(d->rows, n, 8);
  ir->cols = dupArray(d->cols, n, 8);
  ir->childBegin = dupArray(d->childBegin, n + 1, 8);
  ir->childNode = dupArray(d->childNode, nc, 8);
  ir->childRow0 = dupArray(d->childRow0, nc, 8);
  ir->childCol0 = dupArray(d->childCol0, nc, 8);
  ir->leafData = calloc(n, sizeof(void *));
  ir->leafRowStride = malloc(n * 8);
  ir->leafColStride = malloc(n * 8);
  ir->leafReal = calloc(n ? n : 1, 1);
  if (!ir->leafReal || !ir->kind || !ir->rows || !ir->cols || !ir->childBegin || !ir->childNode || !ir->childRow0 ||
      !ir->childCol0 || !ir->leafData || !ir->leafRowStride || !ir->leafColStride) {
    bfIrFree(ir);
    return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM copying descriptor");
  }
  for (uint64_t i = 0; i < n; ++i) {
    ir->leafData[i] = d->leafData ? d->leafData[i] : NULL;
    ir->leafRowStride[i] = (d->leafRowStride && ir->leafData[i]) ? d->leafRowStride[i] : ir->cols[i];
    ir->leafColStride[i] = 1;
  }
  if (d->topRowBlock) {
    uint64_t rc = ir->childBegin[ir->root + 1] - ir->childBegin[ir->root];
    ir->topRowBlock = dupArray(d->topRowBlock, rc, 8);
  }
  return bfIrFinalize(ir);
}

/* ---- growable IR for the BfMat walker ------------------------------------ */
static int irReserveNodes(BfIr *ir, uint64_t want) {
  if (want <= ir->capNodes) return 0;
  uint64_t cap = ir->capNodes ? ir->capNodes * 2 : 1024;
  while (cap < want) cap *= 2;
#define GROW(field, elt) do { void *p = realloc(ir->field, cap * (elt)); if (!p) return 1; ir->field = p; } while (0)
  GROW(kind, 1); GROW(rows, 8); GROW(cols, 8); GROW(leafData, sizeof(void *));
  GROW(leafRowStride, 8); GROW(leafColStride, 8); GROW(leafReal, 1);
  { void *p = realloc(ir->childBegin, (cap + 1) * 8); if (!p) return 1; ir->childBegin = p; }
#undef GROW
  ir->capNodes = cap;
  return 0;
}
static int irReserveChildren(BfIr *ir, uint64_t want) {
  if (want <= ir->capChildren) return 0;
  uint64_t cap = ir->capChildren ? ir->capChildren * 2 : 4096;
  while (cap < want) cap *= 2;
#define GROW(field) do { void *p = realloc(ir->field, cap * 8); if (!p) return 1; ir->field = p; } while (0)
  GROW(childNode); GROW(childRow0); GROW(childCol0);
#undef GROW
  ir->capChildren = cap;
  return 0;
}

/* The walker emits nodes in post-order into temporary per-node child lists,
 * then compacts to CSR. */
typedef struct WalkChild { uint64_t node, r0, c0; } WalkChild;
typedef struct Walk {
  BfIr *ir;
  WalkChild **lists;   /* per node */
  uint64_t *counts;
  uint64_t capLists;
  int sawComplex, sawReal;
  int spanMismatch;    /* some block is smaller than its (rowOffset, colOffset) span */
} Walk;

static int walkNewNode(Walk *w, uint8_t kind, uint64_t rows, uint64_t cols, uint64_t *id) {
  BfIr *ir = w->ir;
  if (irReserveNodes(ir, ir->numNodes + 1)) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
  if (ir->numNodes + 1 > w->capLists) {
    uint64_t cap = w->capLists ? w->capLists * 2 : 1024;
    void *p = realloc(w->lists, cap * sizeof(WalkChild *));
    void *q = realloc(w->counts, cap * sizeof(uint64_t));
    if (p) w->lists = p;
    if (q) w->counts = q;
    if (!p || !q) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
    w->capLists = cap;
  }
  uint64_t i = ir->numNodes++;
  ir->kind[i] = kind;
  ir->rows[i] = rows;
  ir->cols[i] = cols;
  ir->leafData[i] = NULL;
  ir->leafRowStride[i] = cols;
  ir->leafColStride[i] = 1;
  ir->leafReal[i] = 0;
  w->lists[i] = NULL;
  w->counts[i] = 0;
  *id = i;
  return 0;
}

static int checkOffsets(size_t const *off, size_t n, char const *what) {
  if (!off) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "%s is NULL", what);
  for (size_t i = 0; i < n; ++i)
    if (off[i + 1] < off[i]) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "%s not monotone at %zu", what, i);
  return 0;
}

static int walkMat(Walk *w, BfAbiMat const *mat, uint64_t *outId, int level);

static int walkBlockChildren(Walk *w, BfAbiMatBlock const *blk, uint64_t id, size_t numBlocks,
                             size_t const *rowIdx, size_t const *colIdx, int denseGrid, int level) {
  BfAbiMat const *m = &blk->super;
  size_t nbr = m->numRows, nbc = m->numCols;
  if (numBlocks && !blk->block) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "block array is NULL");
  WalkChild *list = malloc((numBlocks ? numBlocks : 1) * sizeof(WalkChild));
  if (!list) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
  uint64_t cnt = 0;
  uint64_t totalRows = blk->rowOffset[nbr], totalCols = blk->colOffset[nbc];
  for (size_t k = 0; k < numBlocks; ++k) {
    size_t bi = denseGrid ? k / nbc : (rowIdx ? rowIdx[k] : k);
    size_t bj = denseGrid ? k % nbc : (colIdx ? colIdx[k] : k);
    if (bi >= nbr || bj >= nbc) { free(list); return bfhipFail(BFABI_ERROR_OUT_OF_RANGE, "block index (%zu,%zu) outside %zux%zu grid", bi, bj, nbr, nbc); }
    BfAbiMat const *child = blk->block[k];
    if (!child) { free(list); return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "NULL block %zu", k); }
    uint64_t cid;
    int rc = walkMat(w, child, &cid, level + 1);
    if (rc) { free(list); return rc; }
    uint64_t r0 = blk->rowOffset[bi], c0 = blk->colOffset[bj];
    uint64_t spanR = blk->rowOffset[bi + 1] - r0, spanC = blk->colOffset[bj + 1] - c0;
    uint64_t cr = w->ir->rows[cid], cc = w->ir->cols[cid];
    /* the Mul path needs exact spans (AddInplace / SetRowRange shape checks,
     * mat_dense_complex.c:1571-1588, :1494-1503); the MulVec path places a
     * block by its own size (mat_block_coo.c:451-455) */
    if (cr != spanR || cc != spanC) w->spanMismatch = 1;
    if (cr > spanR || cc > spanC || r0 + cr > totalRows || c0 + cc > totalCols) {
      free(list);
      return bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "block %zu is %llux%llu but its span is %llux%llu", k,
                       (unsigned long long)cr, (unsigned long long)cc, (unsigned long long)spanR, (unsigned long long)spanC);
    }
    if (cr == 0 || cc == 0) continue;   /* degenerate rows/cols are skipped (mat_block_dense.c:601-615) */
    list[cnt].node = cid; list[cnt].r0 = r0; list[cnt].c0 = c0;
    ++cnt;
  }
  w->lists[id] = list;
  w->counts[id] = cnt;
  return 0;
}

/* Try to add `value` at (i, j) of `node` into a host-valued dense leaf that covers the position and is reached
 * through block nodes only (then the entry simply adds to that leaf's element: nothing else multiplies it).
 * Returns 1 when folded, 0 when no such leaf exists (products, identities, synthetic leaves), < 0 on error. */
static int tryFold(Walk *w, uint64_t node, uint64_t i, uint64_t j, double re, double im, int level) {
  BfIr *ir = w->ir;
  if (level > 64) return 0;
  if (ir->kind[node] == BFHIP_NODE_DENSE) {
    if (!ir->leafData[node] || (ir->leafReal[node] & BF_LEAF_REAL) || ir->rows[node] * ir->cols[node] == 1) return 0;
    if (ir->numPatches == ir->capPatches) {
      uint64_t cap = ir->capPatches ? ir->capPatches * 2 : 4096;
      BfIrPatch *p = realloc(ir->patches, cap * sizeof *p);
      if (!p) return -1;
      ir->patches = p; ir->capPatches = cap;
    }
    BfIrPatch *pt = &ir->patches[ir->numPatches++];
    pt->leaf = node; pt->row = (uint32_t)i; pt->col = (uint32_t)j; pt->re = re; pt->im = im;
    return 1;
  }
  if (ir->kind[node] != BFHIP_NODE_BLOCK) return 0;
  for (uint64_t c = 0; c < w->counts[node]; ++c) {
    WalkChild const *ch = &w->lists[node][c];
    if (i < ch->r0 || j < ch->c0 || i - ch->r0 >= ir->rows[ch->node] || j - ch->c0 >= ir->cols[ch->node]) continue;
    int rc = tryFold(w, ch->node, i - ch->r0, j - ch->c0, re, im, level + 1);
    if (rc) return rc;
  }
  return 0;
}

static int walkSparseTerm(Walk *w, BfAbiMat const *mat, int type, uint64_t const *foldInto, size_t numFoldInto, uint64_t *outId);

static int walkMat(Walk *w, BfAbiMat const *mat, uint64_t *outId, int level) {
  if (level > 256) return bfhipFail(BFABI_ERROR_RUNTIME_ERROR, "BfMat graph nested deeper than 256");
  if (!mat || !mat->vtbl) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfMat or its vtable is NULL");
  BfAbiGetTypeFn getType = (BfAbiGetTypeFn)mat->vtbl->slot[BFABI_SLOT_GetType];
  if (!getType) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BfMat vtable has no GetType");
  int type = getType(mat);
  int rc;
  switch (type) {
  case BFABI_TYPE_MAT_DENSE_COMPLEX: {
    BfAbiMatDenseComplex const *d = (BfAbiMatDenseComplex const *)mat;
    /* A leaf bfMatDenseComplexTranspose has flagged (= bfMatConjTrans: TRANS | CONJ, src/mat_dense_complex.c:1475-1478): the
     * reference multiplies by it through CblasConjTrans -- getCblasTranspose maps TRANS *or* CONJ to it (:27-35) -- with the
     * extents GetNumRows / GetNumCols report, swapped under TRANS (:503-511).  Here: the conjugate of the stored values read
     * with the strides swapped.  CONJ without TRANS would be ConjTrans with UNswapped extents; nothing in the reference
     * produces that state and it is refused. */
    int const flagged = (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) != 0;
    if (flagged && !(mat->props & BFABI_MAT_PROPS_TRANS))
      return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "dense complex leaf flagged CONJ without TRANS (mat_dense_complex.c:27-35 would pass zgemm ConjTrans with untransposed extents)");
    if (!d->data && mat->numRows && mat->numCols) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "dense complex leaf has NULL data");
    w->sawComplex = 1;
    if ((rc = walkNewNode(w, BFHIP_NODE_DENSE, flagged ? mat->numCols : mat->numRows, flagged ? mat->numRows : mat->numCols, outId))) return rc;
    w->ir->leafData[*outId] = d->data;
    w->ir->leafRowStride[*outId] = flagged ? d->colStride : d->rowStride;
    w->ir->leafColStride[*outId] = flagged ? d->rowStride : d->colStride;
    if (flagged) w->ir->leafReal[*outId] |= BF_LEAF_CONJ;
    return 0;
  }
  case BFABI_TYPE_MAT_DENSE_REAL: {
    BfAbiMatDenseReal const *d = (BfAbiMatDenseReal const *)mat;
    /* a real leaf flagged TRANS (src/mat_dense_real.c:20-28, :291-298): the transpose, read with the strides swapped */
    int const flagged = (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) != 0;
    if (flagged && !(mat->props & BFABI_MAT_PROPS_TRANS))
      return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "dense real leaf flagged CONJ without TRANS");
    if (!d->data && mat->numRows && mat->numCols) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "dense real leaf has NULL data");
    w->sawReal = 1;
    if ((rc = walkNewNode(w, BFHIP_NODE_DENSE, flagged ? mat->numCols : mat->numRows, flagged ? mat->numRows : mat->numCols, outId))) return rc;
    w->ir->leafData[*outId] = d->data;
    w->ir->leafRowStride[*outId] = flagged ? d->super.colStride : d->super.rowStride;
    w->ir->leafColStride[*outId] = flagged ? d->super.rowStride : d->super.colStride;
    return 0;
  }
  case BFABI_TYPE_MAT_IDENTITY:
    if (mat->numRows != mat->numCols) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "non-square identity (mat_identity.c:152)");
    return walkNewNode(w, BFHIP_NODE_IDENTITY, mat->numRows, mat->numCols, outId);
  case BFABI_TYPE_MAT_BLOCK_DIAG:
  case BFABI_TYPE_MAT_BLOCK_COO:
  case BFABI_TYPE_MAT_BLOCK_DENSE: {
    BfAbiMatBlock const *blk = (BfAbiMatBlock const *)mat;
    if (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "transposed block matrix");
    if ((rc = checkOffsets(blk->rowOffset, mat->numRows, "rowOffset"))) return rc;
    if ((rc = checkOffsets(blk->colOffset, mat->numCols, "colOffset"))) return rc;
    uint64_t id;
    /* children first would reorder ids; reserve the parent id after walking
     * children is fine because children lists are per node */
    if ((rc = walkNewNode(w, BFHIP_NODE_BLOCK, blk->rowOffset[mat->numRows], blk->colOffset[mat->numCols], &id))) return rc;
    if (type == BFABI_TYPE_MAT_BLOCK_DIAG) {
      size_t nb = mat->numRows < mat->numCols ? mat->numRows : mat->numCols;   /* mat_block_diag.c:635-638 */
      rc = walkBlockChildren(w, blk, id, nb, NULL, NULL, 0, level);
    } else if (type == BFABI_TYPE_MAT_BLOCK_COO) {
      BfAbiMatBlockCoo const *coo = (BfAbiMatBlockCoo const *)mat;
      if (coo->numBlocks && (!coo->rowInd || !coo->colInd)) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "BlockCoo index arrays are NULL");
      rc = walkBlockChildren(w, blk, id, coo->numBlocks, coo->rowInd, coo->colInd, 0, level);
    } else {
      rc = walkBlockChildren(w, blk, id, mat->numRows * mat->numCols, NULL, NULL, 1, level);
    }
    if (rc) return rc;
    *outId = id;
    return 0;
  }
  case BFABI_TYPE_MAT_PRODUCT: {
    BfAbiMatProduct const *p = (BfAbiMatProduct const *)mat;
    if (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "transposed product (mat_product.c:152)");
    size_t nf = p->factorArr.num_elts;
    if (!nf || !p->factorArr.data) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "product has no factors");
    uint64_t id;
    if ((rc = walkNewNode(w, BFHIP_NODE_PRODUCT, 0, 0, &id))) return rc;
    WalkChild *list = malloc(nf * sizeof(WalkChild));
    if (!list) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
    for (size_t i = 0; i < nf; ++i) {
      uint64_t cid;
      rc = walkMat(w, (BfAbiMat const *)p->factorArr.data[i], &cid, level + 1);
      if (rc) { free(list); return rc; }
      list[i].node = cid; list[i].r0 = 0; list[i].c0 = 0;
    }
    for (size_t i = 0; i + 1 < nf; ++i)
      if (w->ir->cols[list[i].node] != w->ir->rows[list[i + 1].node]) {
        free(list);
        return bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "product factors %zu and %zu do not chain", i, i + 1);
      }
    w->ir->rows[id] = w->ir->rows[list[0].node];        /* mat_product.c:146-166 */
    w->ir->cols[id] = w->ir->cols[list[nf - 1].node];   /* mat_product.c:168-192 */
    w->lists[id] = list;
    w->counts[id] = nf;
    *outId = id;
    return 0;
  }
  case BFABI_TYPE_MAT_SUM: {
    /* bfMatSumMul (src/mat_sum.c:54-83): zeros, then += every term's product.  What
     * bfMatBlockDenseAddInplace leaves behind when a correction lands on a butterfly
     * (src/mat_block_dense.c:486-497). */
    BfAbiMatSum const *sum = (BfAbiMatSum const *)mat;
    size_t nt = sum->termArr.num_elts;
    if (!nt || !sum->termArr.data) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "sum has no terms");
    uint64_t id;
    if ((rc = walkNewNode(w, BFHIP_NODE_BLOCK, 0, 0, &id))) return rc;
    WalkChild *list = malloc(nt * sizeof(WalkChild));
    uint64_t *plain = malloc(nt * sizeof(uint64_t));
    if (!list || !plain) { free(list); free(plain); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph"); }
    /* ordinary terms first; the sparse corrections (Kapur-Rokhlin entries, c I) are then added into whatever
     * dense leaf of those terms covers each entry, and only the entries over butterflied blocks stay terms */
    size_t numPlain = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (size_t i = 0; i < nt; ++i) {
        BfAbiMat const *term = (BfAbiMat const *)sum->termArr.data[i];
        if (!term || !term->vtbl || !term->vtbl->slot[BFABI_SLOT_GetType]) { free(list); free(plain); return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "sum term or its vtable is NULL"); }
        int const ttype = ((BfAbiGetTypeFn)term->vtbl->slot[BFABI_SLOT_GetType])(term);
        int const sparse = ttype == BFABI_TYPE_MAT_COO_COMPLEX || ttype == BFABI_TYPE_MAT_DIAG_REAL;
        if (sparse != pass) continue;
        uint64_t cid;
        rc = sparse ? walkSparseTerm(w, term, ttype, plain, numPlain, &cid) : walkMat(w, term, &cid, level + 1);
        if (rc) { free(list); free(plain); return rc; }
        list[i].node = cid; list[i].r0 = 0; list[i].c0 = 0;
        if (!sparse) plain[numPlain++] = cid;
      }
    free(plain);
    for (size_t i = 1; i < nt; ++i)
      if (w->ir->rows[list[i].node] != w->ir->rows[list[0].node] || w->ir->cols[list[i].node] != w->ir->cols[list[0].node]) {
        free(list);
        return bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "terms of a sum differ in shape");
      }
    w->ir->rows[id] = w->ir->rows[list[0].node];
    w->ir->cols[id] = w->ir->cols[list[0].node];
    w->lists[id] = list;
    w->counts[id] = nt;
    *outId = id;
    return 0;
  }
  case BFABI_TYPE_MAT_COO_COMPLEX:
  case BFABI_TYPE_MAT_DIAG_REAL:
    return walkSparseTerm(w, mat, type, NULL, 0, outId);
  default:
    return bfhipFail(BFABI_ERROR_TYPE_ERROR, "unsupported BfType %d in factorization graph", type);
  }
}

/* sparse corrections (Kapur-Rokhlin entries, 1/2 I): an entry is folded into a covering dense leaf of one of
 * the `foldInto` nodes when there is one; what is left becomes 1 x 1 leaves, entries of one row add up.
 * (The reference's own bfMatCooComplexMul *assigns* z * x_j to the result row, src/mat_coo_complex.c:248-251,
 * so its last entry of a row wins; that is not reproduced.) */
static int walkSparseTerm(Walk *w, BfAbiMat const *mat, int type, uint64_t const *foldInto, size_t numFoldInto, uint64_t *outId) {
  size_t ne;
  size_t const *ri = NULL, *ci = NULL;
  double const *val;
  int const isDiag = type == BFABI_TYPE_MAT_DIAG_REAL;
  int rc;
  if (isDiag) { BfAbiMatDiagReal const *d = (BfAbiMatDiagReal const *)mat; ne = d->numElts; val = d->data; }
  else { BfAbiMatCooComplex const *c = (BfAbiMatCooComplex const *)mat; ne = c->numElts; ri = c->rowInd; ci = c->colInd; val = c->value; w->sawComplex = 1; }
  if (ne && (!val || (!isDiag && (!ri || !ci)))) return bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "sparse matrix arrays are NULL");
  if (mat->props & (BFABI_MAT_PROPS_TRANS | BFABI_MAT_PROPS_CONJ)) return bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "transposed sparse term");
  uint64_t id;
  if ((rc = walkNewNode(w, BFHIP_NODE_BLOCK, mat->numRows, mat->numCols, &id))) return rc;
  WalkChild *list = malloc((ne ? ne : 1) * sizeof(WalkChild));
  if (!list) return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
  uint64_t kept = 0;
  for (size_t k = 0; k < ne; ++k) {
    size_t i = isDiag ? k : ri[k], j = isDiag ? k : ci[k];
    if (i >= mat->numRows || j >= mat->numCols) { free(list); return bfhipFail(BFABI_ERROR_OUT_OF_RANGE, "sparse entry (%zu,%zu) out of range", i, j); }
    int folded = 0;
    for (size_t t = 0; t < numFoldInto && !folded; ++t) {
      folded = tryFold(w, foldInto[t], i, j, isDiag ? val[k] : val[2 * k], isDiag ? 0.0 : val[2 * k + 1], 0);
      if (folded < 0) { free(list); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph"); }
    }
    if (folded) continue;
    uint64_t cid;
    if ((rc = walkNewNode(w, BFHIP_NODE_DENSE, 1, 1, &cid))) { free(list); return rc; }
    w->ir->leafData[cid] = isDiag ? (void const *)(val + k) : (void const *)(val + 2 * k);
    w->ir->leafReal[cid] = (uint8_t)isDiag;
    list[kept].node = cid; list[kept].r0 = i; list[kept].c0 = j;
    ++kept;
  }
  w->lists[id] = list;
  w->counts[id] = kept;
  *outId = id;
  return 0;
}

int bfIrFromBfMat(void const *bfMat, BfIr *ir) {
  memset(ir, 0, sizeof *ir);
  Walk w;
  memset(&w, 0, sizeof w);
  w.ir = ir;
  uint64_t root = 0;
  int rc = walkMat(&w, (BfAbiMat const *)bfMat, &root, 0);
  if (!rc && w.sawComplex && w.sawReal) rc = bfhipFail(BFABI_ERROR_TYPE_ERROR, "graph mixes complex and real dense leaves");
  if (!rc && w.sawComplex && w.spanMismatch)
    rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "a block is smaller than its offset span; the reference's Mul path rejects this (mat_dense_complex.c:1571-1588)");
  if (!rc) {
    ir->dtype = w.sawReal ? BFHIP_F64 : BFHIP_C128;
    ir->root = root;
    uint64_t total = 0;
    for (uint64_t i = 0; i < ir->numNodes; ++i) total += w.counts[i];
    if (irReserveChildren(ir, total + 1)) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM while walking BfMat graph");
    if (!rc) {
      uint64_t p = 0;
      for (uint64_t i = 0; i < ir->numNodes; ++i) {
        ir->childBegin[i] = p;
        for (uint64_t c = 0; c < w.counts[i]; ++c, ++p) {
          ir->childNode[p] = w.lists[i][c].node;
          ir->childRow0[p] = w.lists[i][c].r0;
          ir->childCol0[p] = w.lists[i][c].c0;
        }
      }
      ir->childBegin[ir->numNodes] = p;
      ir->numChildren = p;
      /* block-row ids of the root's children, for row sharding */
      BfAbiMat const *rm = (BfAbiMat const *)bfMat;
      int rtype = ((BfAbiGetTypeFn)rm->vtbl->slot[BFABI_SLOT_GetType])(rm);
      if (rtype == BFABI_TYPE_MAT_BLOCK_DENSE || rtype == BFABI_TYPE_MAT_BLOCK_COO || rtype == BFABI_TYPE_MAT_BLOCK_DIAG) {
        BfAbiMatBlock const *blk = (BfAbiMatBlock const *)rm;
        uint64_t rcnt = w.counts[root];
        ir->topRowBlock = malloc((rcnt ? rcnt : 1) * 8);
        if (!ir->topRowBlock) rc = bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM");
        for (uint64_t c = 0; !rc && c < rcnt; ++c) {
          /* block row = index of the child's row offset in rowOffset[] */
          uint64_t r0 = w.lists[root][c].r0;
          size_t lo = 0, hi = rm->numRows;
          while (lo < hi) { size_t mid = (lo + hi) / 2; if (blk->rowOffset[mid] < r0) lo = mid + 1; else hi = mid; }
          /* skip degenerate (empty) block rows sharing the same offset */
          while (lo + 1 < rm->numRows && blk->rowOffset[lo + 1] == r0) ++lo;
          ir->topRowBlock[c] = lo;
        }
      }
    }
  }
  for (uint64_t i = 0; i < ir->numNodes && w.lists; ++i) free(w.lists[i]);
  free(w.lists);
  free(w.counts);
  if (rc) { bfIrFree(ir); return rc; }
  return bfIrFinalize(ir);
}

static int cmpPatch(void const *pa, void const *pb) {
  BfIrPatch const *a = pa, *b = pb;
  if (a->leaf != b->leaf) return a->leaf < b->leaf ? -1 : 1;
  if (a->row != b->row) return a->row < b->row ? -1 : 1;
  return a->col < b->col ? -1 : (a->col > b->col);
}

/* validation + derived fields */
int bfIrFinalize(BfIr *ir) {
  uint64_t n = ir->numNodes;
  if (ir->numPatches) qsort(ir->patches, ir->numPatches, sizeof(BfIrPatch), cmpPatch);
  ir->depth = calloc(n, sizeof(uint32_t));
  ir->synthBase = malloc(n * 8);
  if (!ir->depth || !ir->synthBase) { bfIrFree(ir); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  uint64_t acc = 0;
  for (uint64_t i = 0; i < n; ++i) {
    ir->synthBase[i] = acc;
    if (ir->kind[i] == BFHIP_NODE_DENSE) acc += ir->rows[i] * ir->cols[i];
    if (ir->kind[i] > BFHIP_NODE_PRODUCT) { bfIrFree(ir); return bfhipFail(BFABI_ERROR_TYPE_ERROR, "node %llu has unknown kind", (unsigned long long)i); }
  }
  /* depth by iterative post-order (children may have any ids) */
  uint8_t *state = calloc(n, 1);
  /* a node is pushed once per parent entry that lists it (a block may list one child many
   * times, a descriptor may share sub-expressions), so the stack holds up to one entry per child
   * edge plus the root */
  uint64_t *stack = malloc((ir->numChildren + 2) * 8);
  if (!state || !stack) { free(state); free(stack); bfIrFree(ir); return bfhipFail(BFABI_ERROR_MEMORY_ERROR, "host OOM"); }
  uint64_t sp = 0;
  stack[sp++] = ir->root;
  int rc = 0;
  while (sp && !rc) {
    uint64_t v = stack[sp - 1];
    uint64_t b = ir->childBegin[v], e = ir->childBegin[v + 1];
    if (state[v] == 0) {
      state[v] = 1;
      if (ir->kind[v] == BFHIP_NODE_DENSE || ir->kind[v] == BFHIP_NODE_IDENTITY) {
        if (e != b) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "leaf node %llu has children", (unsigned long long)v);
        if (ir->kind[v] == BFHIP_NODE_IDENTITY && ir->rows[v] != ir->cols[v]) rc = bfhipFail(BFABI_ERROR_NOT_IMPLEMENTED, "non-square identity");
        if (!ir->rows[v] || !ir->cols[v]) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "empty leaf %llu (zgemm asserts m,n,k > 0, mat_dense_complex.c:1737)", (unsigned long long)v);
        ir->depth[v] = 1;
        state[v] = 2;
        --sp;
        continue;
      }
      for (uint64_t c = b; c < e && !rc; ++c) {
        uint64_t ch = ir->childNode[c];
        if (ch >= n) { rc = bfhipFail(BFABI_ERROR_OUT_OF_RANGE, "child id out of range"); break; }
        if (state[ch] == 1) { rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "descriptor has a cycle"); break; }
        if (state[ch] == 0) stack[sp++] = ch;
      }
    } else {
      if (state[v] == 1) {
        uint32_t dep = 0;
        if (ir->kind[v] == BFHIP_NODE_PRODUCT) {
          if (e == b) rc = bfhipFail(BFABI_ERROR_INVALID_ARGUMENTS, "product without factors");
          for (uint64_t c = b; c < e && !rc; ++c) {
            uint64_t ch = ir->childNode[c];
            dep += ir->depth[ch];
            if (c + 1 < e && ir->cols[ch] != ir->rows[ir->childNode[c + 1]]) rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "product factors do not chain");
          }
          if (!rc) {
            ir->rows[v] = ir->rows[ir->childNode[b]];
            ir->cols[v] = ir->cols[ir->childNode[e - 1]];
          }
        } else { /* BLOCK */
          for (uint64_t c = b; c < e && !rc; ++c) {
            uint64_t ch = ir->childNode[c];
            if (ir->depth[ch] > dep) dep = ir->depth[ch];
            if (ir->childRow0[c] + ir->rows[ch] > ir->rows[v] || ir->childCol0[c] + ir->cols[ch] > ir->cols[v])
              rc = bfhipFail(BFABI_ERROR_INCOMPATIBLE_SHAPES, "child %llu of block %llu sticks out", (unsigned long long)ch, (unsigned long long)v);
          }
          if (dep == 0) dep = 1;   /* an empty block is a zero operator: one (zero-fill) stage */
        }
        ir->depth[v] = dep;
        state[v] = 2;
      }
      --sp;
    }
  }
  free(state);
  free(stack);
  if (rc) { bfIrFree(ir); return rc; }
  return 0;
}
