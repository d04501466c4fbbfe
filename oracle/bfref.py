"""ORACLE (test infrastructure, not product code): ctypes access to
oracle/libbfref.so, the CPU restatement of the reference's bfMatMul /
bfMatMulVec (see oracle/bfref.h for what it follows and its parity status).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module."""
from __future__ import annotations

import ctypes as C
import glob
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbfref.so")
_lib = None


class Counters(C.Structure):
    _fields_ = [("gemmCalls", C.c_uint64), ("macs", C.c_uint64), ("mallocs", C.c_uint64)]


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    # RTLD_GLOBAL: a foreign operator behind the vtable (the device shim) finds the oracle's
    # bfSetError the way it would find the reference's in a host process that links libbf
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    sz = C.c_size_t
    lib.bfMatMul.argtypes = [vp, vp]; lib.bfMatMul.restype = vp
    lib.bfMatMulVec.argtypes = [vp, vp]; lib.bfMatMulVec.restype = vp
    lib.bfMatRmulVec.argtypes = [vp, vp]; lib.bfMatRmulVec.restype = vp
    lib.bfMatRmul.argtypes = [vp, vp]; lib.bfMatRmul.restype = vp
    lib.bfMatDelete.argtypes = [C.POINTER(vp)]; lib.bfMatDelete.restype = None
    lib.bfMatTranspose.argtypes = [vp]; lib.bfMatTranspose.restype = None
    lib.bfrefMatFromDescTyped.argtypes = [vp, C.c_uint64, C.c_uint64]; lib.bfrefMatFromDescTyped.restype = vp
    lib.bfVecDelete.argtypes = [C.POINTER(vp)]; lib.bfVecDelete.restype = None
    for f in ("bfMatGetNumRows", "bfMatGetNumCols", "bfMatNumBytes"):
        getattr(lib, f).argtypes = [vp]; getattr(lib, f).restype = sz
    lib.bfMatGetType.argtypes = [vp]; lib.bfMatGetType.restype = C.c_int
    lib.bfMatDenseComplexNewFromPtr.argtypes = [sz, sz, vp, C.c_int]; lib.bfMatDenseComplexNewFromPtr.restype = vp
    lib.bfMatDenseComplexNewZeros.argtypes = [sz, sz]; lib.bfMatDenseComplexNewZeros.restype = vp
    lib.bfMatDenseRealNewFromPtr.argtypes = [sz, sz, vp, C.c_int]; lib.bfMatDenseRealNewFromPtr.restype = vp
    lib.bfMatIdentityNew.argtypes = [sz]; lib.bfMatIdentityNew.restype = vp
    lib.bfMatBlockDiagNewFromBlocks.argtypes = [sz, vp]; lib.bfMatBlockDiagNewFromBlocks.restype = vp
    lib.bfMatBlockCooNewFromArrays.argtypes = [sz, sz, sz, vp, vp, vp, vp, vp]; lib.bfMatBlockCooNewFromArrays.restype = vp
    lib.bfMatBlockDenseNewFromBlocks.argtypes = [sz, sz, vp, vp, vp]; lib.bfMatBlockDenseNewFromBlocks.restype = vp
    lib.bfMatProductNewFromFactors.argtypes = [sz, vp]; lib.bfMatProductNewFromFactors.restype = vp
    lib.bfMatSumNewFromTerms.argtypes = [sz, vp]; lib.bfMatSumNewFromTerms.restype = vp
    lib.bfMatCooComplexNewFromArrays.argtypes = [sz, sz, sz, vp, vp, vp]; lib.bfMatCooComplexNewFromArrays.restype = vp
    lib.bfMatDiagRealNewFromPtr.argtypes = [sz, sz, sz, vp]; lib.bfMatDiagRealNewFromPtr.restype = vp
    lib.bfrefCooComplexAssignQuirk.argtypes = [C.c_int]; lib.bfrefCooComplexAssignQuirk.restype = None
    lib.bfVecRealNewFromPtr.argtypes = [sz, vp, C.c_int]; lib.bfVecRealNewFromPtr.restype = vp
    lib.bfVecRealData.argtypes = [vp]; lib.bfVecRealData.restype = vp
    lib.bfMatDenseData.argtypes = [vp]; lib.bfMatDenseData.restype = vp
    lib.bfGetError.argtypes = []; lib.bfGetError.restype = C.c_int
    lib.bfClearError.argtypes = []; lib.bfClearError.restype = None
    lib.bfrefResetCounters.argtypes = []; lib.bfrefResetCounters.restype = None
    lib.bfrefGetCounters.argtypes = [C.POINTER(Counters)]; lib.bfrefGetCounters.restype = None
    lib.bfrefUseBlas.argtypes = [C.c_char_p, C.c_char_p]; lib.bfrefUseBlas.restype = C.c_int
    lib.bfrefBlasName.argtypes = []; lib.bfrefBlasName.restype = C.c_char_p
    lib.bfrefMatFromDesc.argtypes = [vp, C.c_uint64, C.c_uint64]; lib.bfrefMatFromDesc.restype = vp
    _lib = lib
    return lib


def try_use_openblas():
    """Route leaf products through the OpenBLAS bundled with the scipy/numpy
    wheels when present (what the reference links: include/bf/blas.h:3-9)."""
    lib = load()
    cands = []
    for pkg in ("scipy", "numpy"):
        try:
            mod = __import__(pkg)
            base = os.path.dirname(os.path.dirname(mod.__file__))
            cands += glob.glob(os.path.join(base, f"{pkg}.libs", "libscipy_openblas*.so"))
        except Exception:
            pass
    for path in cands:
        prefix = b"scipy_"
        if lib.bfrefUseBlas(path.encode(), prefix) == 0:
            return path
    return None


def _idx(a):
    return np.ascontiguousarray(a, dtype=np.uintp)


class Mat:
    """Owning handle on an oracle BfMat."""

    def __init__(self, ptr, keep=()):
        if not ptr:
            err = load().bfGetError()
            load().bfClearError()
            raise RuntimeError(f"oracle returned NULL (BfError {err})")
        self.ptr = C.c_void_p(ptr)
        self.keep = list(keep)
        self.owned = True

    def release(self):
        """Give up ownership (the pointer was stolen by a container)."""
        self.owned = False
        return self.ptr.value

    def __del__(self):
        if getattr(self, "owned", False) and self.ptr and _lib is not None:
            _lib.bfMatDelete(C.byref(self.ptr))

    @property
    def shape(self):
        lib = load()
        return int(lib.bfMatGetNumRows(self.ptr)), int(lib.bfMatGetNumCols(self.ptr))

    @property
    def type(self):
        return int(load().bfMatGetType(self.ptr))

    def num_bytes(self):
        return int(load().bfMatNumBytes(self.ptr))

    def to_numpy(self):
        """Copy of a dense complex / real matrix."""
        lib = load()
        m, n = self.shape
        p = lib.bfMatDenseData(self.ptr)
        if self.type == 19:
            buf = (C.c_double * (2 * m * n)).from_address(p)
            return np.frombuffer(buf, dtype=np.complex128).reshape(m, n).copy()
        buf = (C.c_double * (m * n)).from_address(p)
        return np.frombuffer(buf, dtype=np.float64).reshape(m, n).copy()


def dense_complex(a):
    a = np.ascontiguousarray(a, dtype=np.complex128)
    if a.ndim == 1:
        a = a[:, None]
    return Mat(load().bfMatDenseComplexNewFromPtr(a.shape[0], a.shape[1], a.ctypes.data, 0))


def dense_real(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return Mat(load().bfMatDenseRealNewFromPtr(a.shape[0], a.shape[1], a.ctypes.data, 0))


def identity(n):
    return Mat(load().bfMatIdentityNew(n))


def _steal(blocks):
    arr = (C.c_void_p * len(blocks))(*[b.release() for b in blocks])
    return arr


def block_diag(blocks):
    arr = _steal(blocks)
    return Mat(load().bfMatBlockDiagNewFromBlocks(len(blocks), arr))


def block_coo(row_offset, col_offset, row_ind, col_ind, blocks):
    ro, co, ri, ci = _idx(row_offset), _idx(col_offset), _idx(row_ind), _idx(col_ind)
    arr = _steal(blocks)
    return Mat(load().bfMatBlockCooNewFromArrays(len(ro) - 1, len(co) - 1, len(blocks), ro.ctypes.data,
                                                 co.ctypes.data, ri.ctypes.data, ci.ctypes.data, arr))


def block_dense(row_offset, col_offset, blocks):
    """blocks: row-major list of numBlockRows*numBlockCols Mats."""
    ro, co = _idx(row_offset), _idx(col_offset)
    arr = _steal(blocks)
    return Mat(load().bfMatBlockDenseNewFromBlocks(len(ro) - 1, len(co) - 1, ro.ctypes.data, co.ctypes.data, arr))


def product(factors):
    arr = _steal(factors)
    return Mat(load().bfMatProductNewFromFactors(len(factors), arr))


def mat_sum(terms):
    arr = _steal(terms)
    return Mat(load().bfMatSumNewFromTerms(len(terms), arr))


def coo_complex(m, n, row_ind, col_ind, values):
    ri, ci = _idx(row_ind), _idx(col_ind)
    v = np.ascontiguousarray(values, dtype=np.complex128)
    return Mat(load().bfMatCooComplexNewFromArrays(m, n, len(v), ri.ctypes.data, ci.ctypes.data, v.ctypes.data))


def diag_real(m, n, d):
    d = np.ascontiguousarray(d, dtype=np.float64)
    return Mat(load().bfMatDiagRealNewFromPtr(m, n, len(d), d.ctypes.data))


def mat_mul(a: Mat, x: np.ndarray) -> np.ndarray:
    """y = bfMatMul(A, X) with X an N x nrhs complex128 array."""
    lib = load()
    X = dense_complex(x)
    r = lib.bfMatMul(a.ptr, X.ptr)
    Y = Mat(r)
    out = Y.to_numpy()
    return out[:, 0] if np.ndim(x) == 1 else out


def mat_rmul(a: Mat, x: np.ndarray) -> np.ndarray:
    """Z = bfMatRmul(A, X) = X A with X an m x numRows(A) complex128 array (src/mat.c:195-197; Product and DenseComplex fill
    the slot, src/mat_product.c:282-310, src/mat_dense_complex.c:1075-1133; the block types do not)."""
    lib = load()
    X = dense_complex(x)
    lib.bfClearError()
    r = lib.bfMatRmul(a.ptr, X.ptr)
    if not r:
        err = lib.bfGetError()
        lib.bfClearError()
        raise RuntimeError(f"oracle bfMatRmul returned NULL (BfError {err})")
    return Mat(r).to_numpy()


def mat_mul_vec(a: Mat, x: np.ndarray) -> np.ndarray:
    """y = bfMatMulVec(A, x) with x a real vector."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = C.c_void_p(lib.bfVecRealNewFromPtr(len(x), x.ctypes.data, 0))
    r = C.c_void_p(lib.bfMatMulVec(a.ptr, v))
    lib.bfVecDelete(C.byref(v))
    if not r:
        err = lib.bfGetError()
        lib.bfClearError()
        raise RuntimeError(f"oracle bfMatMulVec returned NULL (BfError {err})")
    m = a.shape[0]
    buf = (C.c_double * m).from_address(lib.bfVecRealData(r))
    out = np.frombuffer(buf, dtype=np.float64).copy()
    lib.bfVecDelete(C.byref(r))
    return out


def mat_transpose(a: Mat) -> None:
    """bfMatTranspose(A): in place (src/mat.c:271-273).  Dense complex leaves are flagged TRANS | CONJ
    (bfMatDenseComplexTranspose = bfMatConjTrans, src/mat_dense_complex.c:1475-1478), so a transposed complex operator
    multiplies as its CONJUGATE transpose; BlockCoo and DenseReal have no Transpose slot (RuntimeError here)."""
    lib = load()
    lib.bfClearError()
    lib.bfMatTranspose(a.ptr)
    err = lib.bfGetError()
    if err:
        lib.bfClearError()
        raise RuntimeError(f"oracle bfMatTranspose failed (BfError {err})")


def mat_rmul_vec(a: Mat, x: np.ndarray) -> np.ndarray:
    """y = bfMatRmulVec(A, x) = A^T x with x a real vector of length numRows."""
    lib = load()
    x = np.ascontiguousarray(x, dtype=np.float64)
    v = C.c_void_p(lib.bfVecRealNewFromPtr(len(x), x.ctypes.data, 0))
    r = C.c_void_p(lib.bfMatRmulVec(a.ptr, v))
    lib.bfVecDelete(C.byref(v))
    if not r:
        err = lib.bfGetError()
        lib.bfClearError()
        raise RuntimeError(f"oracle bfMatRmulVec returned NULL (BfError {err})")
    n = a.shape[1]
    buf = (C.c_double * n).from_address(lib.bfVecRealData(r))
    out = np.frombuffer(buf, dtype=np.float64).copy()
    lib.bfVecDelete(C.byref(r))
    return out


def from_desc(desc, leaf_values=None, seed=0, root=None, typed=False) -> Mat:
    """Build the oracle graph for a butterfly_amd.helm2_structure.Desc (or any
    object with .arrays()); leaves without values get the engine's synthetic
    value stream for `seed`.  typed=True: BLOCK nodes the descriptor calls BlockDiag /
    BlockDense become those containers (bfrefMatFromDescTyped) instead of the general BlockCoo."""
    from butterfly_amd._capi import DescArrays
    da = DescArrays(desc, root=root, leaf_values=leaf_values)
    lib = load()
    r = (lib.bfrefMatFromDescTyped if typed else lib.bfrefMatFromDesc)(C.addressof(da.struct), seed, 0xFFFFFFFFFFFFFFFF)
    return Mat(r)


def counters():
    c = Counters()
    load().bfrefGetCounters(C.byref(c))
    return dict(gemmCalls=int(c.gemmCalls), macs=int(c.macs), mallocs=int(c.mallocs))


def reset_counters():
    load().bfrefResetCounters()
