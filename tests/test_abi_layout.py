"""include/bfhip_abi.h must reproduce the reference's struct layouts.  When the
reference tree is present (this container, never the GPU box) the real headers
are compiled and every sizeof / offsetof / vtable slot index / enum value the
engine relies on is compared; the static asserts in bfhip_abi.h are always
compiled."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"

PROBE_REF = r'''
#include <stddef.h>
#include <stdio.h>
#include <bf/mat.h>
#include <bf/mat_block.h>
#include <bf/mat_block_coo.h>
#include <bf/mat_block_dense.h>
#include <bf/mat_block_diag.h>
#include <bf/mat_dense_complex.h>
#include <bf/mat_dense_real.h>
#include <bf/mat_identity.h>
#include <bf/mat_product.h>
#include <bf/mat_sum.h>
#include <bf/mat_coo_complex.h>
#include <bf/mat_diag_real.h>
#include <bf/vec_real.h>
#include <bf/vec_complex.h>
#include <bf/error.h>
#define S(T) printf("sizeof_" #T " %zu\n", sizeof(T))
#define O(T, f) printf("offsetof_" #T "_" #f " %zu\n", offsetof(T, f))
#define V(f) printf("slot_" #f " %zu\n", offsetof(BfMatVtable, f) / sizeof(void *))
#define W(f) printf("vslot_" #f " %zu\n", offsetof(BfVecVtable, f) / sizeof(void *))
#define E(e) printf("enum_" #e " %d\n", (int)(e))
int main(void) {
  S(BfMat); O(BfMat, vtbl); O(BfMat, props); O(BfMat, numRows); O(BfMat, numCols);
  S(BfMatVtable);
  V(GetView); V(Copy); V(Steal); V(Delete); V(EmptyLike); V(ZerosLike); V(GetType); V(NumBytes);
  V(GetNumRows); V(GetNumCols); V(GetRowRange); V(SetRowRange); V(AddInplace); V(Mul); V(MulVec);
  V(MulInplace); V(Rmul); V(RmulVec); V(Transpose);
  S(BfVecVtable); W(Copy); W(Delete); W(GetType); W(GetSubvecCopy); W(GetSubvecView); W(GetSubvecViewConst); W(SetRange); W(AddInplace);
  S(BfPtrArray); O(BfPtrArray, data); O(BfPtrArray, capacity); O(BfPtrArray, num_elts); O(BfPtrArray, isView);
  S(BfMatProduct); O(BfMatProduct, factorArr);
  S(BfMatBlock); O(BfMatBlock, vtbl); O(BfMatBlock, block); O(BfMatBlock, rowOffset); O(BfMatBlock, colOffset);
  S(BfMatBlockCoo); O(BfMatBlockCoo, numBlocks); O(BfMatBlockCoo, rowInd); O(BfMatBlockCoo, colInd);
  S(BfMatBlockDiag); S(BfMatBlockDense);
  S(BfMatDenseComplex); O(BfMatDenseComplex, rowStride); O(BfMatDenseComplex, colStride); O(BfMatDenseComplex, data); O(BfMatDenseComplex, pyArray);
  S(BfMatDense); O(BfMatDense, vtable); O(BfMatDense, rowStride); O(BfMatDense, colStride);
  S(BfMatDenseReal); O(BfMatDenseReal, data);
  S(BfMatIdentity);
  S(BfMatSum); O(BfMatSum, termArr);
  S(BfMatCooComplex); O(BfMatCooComplex, numElts); O(BfMatCooComplex, rowInd); O(BfMatCooComplex, colInd); O(BfMatCooComplex, value);
  S(BfMatDiagReal); O(BfMatDiagReal, numElts); O(BfMatDiagReal, data);
  E(BF_TYPE_MAT_COO_COMPLEX); E(BF_TYPE_MAT_DIAG_REAL);
  S(BfVec); O(BfVec, props); O(BfVec, size);
  S(BfVecReal); O(BfVecReal, stride); O(BfVecReal, data);
  S(BfVecComplex); O(BfVecComplex, stride); O(BfVecComplex, data);
  E(BF_TYPE_MAT_FUNC); E(BF_TYPE_MAT_IDENTITY); E(BF_TYPE_MAT_PRODUCT); E(BF_TYPE_MAT_SUM); E(BF_TYPE_MAT_BLOCK_COO);
  E(BF_TYPE_MAT_BLOCK_DENSE); E(BF_TYPE_MAT_BLOCK_DIAG); E(BF_TYPE_MAT_DENSE_COMPLEX); E(BF_TYPE_MAT_DENSE_REAL);
  E(BF_TYPE_VEC_COMPLEX); E(BF_TYPE_VEC_REAL);
  E(BF_MAT_PROPS_VIEW); E(BF_MAT_PROPS_TRANS); E(BF_MAT_PROPS_CONJ);
  E(BF_ERROR_INVALID_ARGUMENTS); E(BF_ERROR_RUNTIME_ERROR); E(BF_ERROR_NOT_IMPLEMENTED); E(BF_ERROR_MEMORY_ERROR);
  E(BF_ERROR_OUT_OF_RANGE); E(BF_ERROR_TYPE_ERROR); E(BF_ERROR_INCOMPATIBLE_SHAPES);
  return 0;
}
'''

PROBE_OURS = PROBE_REF
for a, b in [("#include <bf/mat.h>", '#include "bfhip_abi.h"')]:
    PROBE_OURS = PROBE_OURS.replace(a, b)
PROBE_OURS = "\n".join(l for l in PROBE_OURS.splitlines() if not l.startswith("#include <bf/"))
PROBE_OURS = PROBE_OURS.replace("int main(void) {", r'''
typedef BfAbiMat BfMat; typedef BfAbiMatVtable BfMatVtable; typedef BfAbiVecVtable BfVecVtable;
typedef BfAbiPtrArray BfPtrArray; typedef BfAbiMatProduct BfMatProduct; typedef BfAbiMatBlock BfMatBlock;
typedef BfAbiMatBlockCoo BfMatBlockCoo; typedef BfAbiMatBlockDiag BfMatBlockDiag; typedef BfAbiMatBlockDense BfMatBlockDense;
typedef BfAbiMatDenseComplex BfMatDenseComplex; typedef BfAbiMatDense BfMatDense; typedef BfAbiMatDenseReal BfMatDenseReal;
typedef BfAbiMatIdentity BfMatIdentity; typedef BfAbiMatSum BfMatSum; typedef BfAbiMatCooComplex BfMatCooComplex; typedef BfAbiMatDiagReal BfMatDiagReal; typedef BfAbiVec BfVec; typedef BfAbiVecReal BfVecReal; typedef BfAbiVecComplex BfVecComplex;
#undef V
#define V(f) printf("slot_" #f " %d\n", (int)BFABI_SLOT_##f)
#undef W
#define W(f) printf("vslot_" #f " %d\n", (int)BFABI_VSLOT_##f)
#undef E
#define E(e) printf("enum_" #e " %d\n", (int)(BFABI_##e))
#define BFABI_BF_TYPE_MAT_FUNC BFABI_TYPE_MAT_FUNC
#define BFABI_BF_TYPE_MAT_IDENTITY BFABI_TYPE_MAT_IDENTITY
#define BFABI_BF_TYPE_MAT_PRODUCT BFABI_TYPE_MAT_PRODUCT
#define BFABI_BF_TYPE_MAT_SUM BFABI_TYPE_MAT_SUM
#define BFABI_BF_TYPE_MAT_COO_COMPLEX BFABI_TYPE_MAT_COO_COMPLEX
#define BFABI_BF_TYPE_MAT_DIAG_REAL BFABI_TYPE_MAT_DIAG_REAL
#define BFABI_BF_TYPE_MAT_BLOCK_COO BFABI_TYPE_MAT_BLOCK_COO
#define BFABI_BF_TYPE_MAT_BLOCK_DENSE BFABI_TYPE_MAT_BLOCK_DENSE
#define BFABI_BF_TYPE_MAT_BLOCK_DIAG BFABI_TYPE_MAT_BLOCK_DIAG
#define BFABI_BF_TYPE_MAT_DENSE_COMPLEX BFABI_TYPE_MAT_DENSE_COMPLEX
#define BFABI_BF_TYPE_MAT_DENSE_REAL BFABI_TYPE_MAT_DENSE_REAL
#define BFABI_BF_TYPE_VEC_COMPLEX BFABI_TYPE_VEC_COMPLEX
#define BFABI_BF_TYPE_VEC_REAL BFABI_TYPE_VEC_REAL
#define BFABI_BF_MAT_PROPS_VIEW BFABI_MAT_PROPS_VIEW
#define BFABI_BF_MAT_PROPS_TRANS BFABI_MAT_PROPS_TRANS
#define BFABI_BF_MAT_PROPS_CONJ BFABI_MAT_PROPS_CONJ
#define BFABI_BF_ERROR_INVALID_ARGUMENTS BFABI_ERROR_INVALID_ARGUMENTS
#define BFABI_BF_ERROR_RUNTIME_ERROR BFABI_ERROR_RUNTIME_ERROR
#define BFABI_BF_ERROR_NOT_IMPLEMENTED BFABI_ERROR_NOT_IMPLEMENTED
#define BFABI_BF_ERROR_MEMORY_ERROR BFABI_ERROR_MEMORY_ERROR
#define BFABI_BF_ERROR_OUT_OF_RANGE BFABI_ERROR_OUT_OF_RANGE
#define BFABI_BF_ERROR_TYPE_ERROR BFABI_ERROR_TYPE_ERROR
#define BFABI_BF_ERROR_INCOMPATIBLE_SHAPES BFABI_ERROR_INCOMPATIBLE_SHAPES
int main(void) {''')


def _run(src, flags, tmp_path, name):
    c = tmp_path / f"{name}.c"
    c.write_text(src)
    exe = tmp_path / name
    subprocess.check_call(["gcc", "-std=gnu11", *flags, str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True)
    return dict(line.split() for line in out.strip().splitlines())


def test_static_asserts_compile(tmp_path):
    ours = _run(PROBE_OURS, [f"-I{ROOT}/include"], tmp_path, "ours")
    assert ours["sizeof_BfMat"] == "32" and ours["sizeof_BfMatVtable"] == "528"
    assert ours["slot_Mul"] == "41" and ours["slot_MulVec"] == "42"


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference headers not present (GPU box)")
def test_layout_matches_reference_headers(tmp_path):
    ref = _run(PROBE_REF, ["-DBF_DOUBLE", "-DBF_LINUX", f"-I{REF_INC}"], tmp_path, "ref")
    ours = _run(PROBE_OURS, [f"-I{ROOT}/include"], tmp_path, "ours")
    assert set(ref) == set(ours)
    diff = {k: (ref[k], ours[k]) for k in ref if ref[k] != ours[k]}
    assert not diff, diff
