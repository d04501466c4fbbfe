"""ctypes declarations of the C-ABI in include/bfhip.h and the loader of
libbfhip.so.  Loading fails loudly: there is no CPU fallback for the engine."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFHIP_LIB_PATH") or os.path.join(_HERE, "csrc", "libbfhip.so")      # (the override is for A/B builds of the kernels)

BFHIP_C128, BFHIP_F64, BFHIP_F32 = 0, 1, 2
FLAG_PROFILE = 1
FLAG_PLAN_ONLY = 2
FLAG_ADJOINT = 4
FLAG_FLOW = 8
FLAG_ADJOINT_PACKED = 16
FLAG_EXACT_COMPLEX = 32

ERROR_NAMES = {0: "BF_ERROR_NONE", 1: "BF_ERROR_INVALID_ARGUMENTS", 2: "BF_ERROR_RUNTIME_ERROR",
               3: "BF_ERROR_NOT_IMPLEMENTED", 4: "BF_ERROR_MEMORY_ERROR", 5: "BF_ERROR_OUT_OF_RANGE",
               6: "BF_ERROR_FILE_ERROR", 7: "BF_ERROR_TYPE_ERROR", 8: "BF_ERROR_INCOMPATIBLE_SHAPES"}


class BfhipOptions(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32),
                ("maxRhs", C.c_uint32), ("demoteToF32", C.c_uint32), ("reserved0", C.c_uint32),
                ("seed", C.c_uint64), ("rowBlockBegin", C.c_uint64), ("rowBlockEnd", C.c_uint64),
                ("rowBegin", C.c_uint64), ("rowEnd", C.c_uint64)]


class BfhipDesc(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("dtype", C.c_uint32), ("numNodes", C.c_uint64),
                ("root", C.c_uint64), ("kind", C.c_void_p), ("rows", C.c_void_p), ("cols", C.c_void_p),
                ("childBegin", C.c_void_p), ("childNode", C.c_void_p), ("childRow0", C.c_void_p),
                ("childCol0", C.c_void_p), ("leafData", C.c_void_p), ("leafRowStride", C.c_void_p),
                ("topRowBlock", C.c_void_p), ("blockKind", C.c_void_p)]


class BfhipStats(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("dtype", C.c_uint32)] + [
        (n, C.c_uint64) for n in ("numRows", "numCols", "numStages", "numLeaves", "numItems", "numPieces",
                                  "leafElems", "leafBytes", "vecElemsRead", "vecElemsWritten",
                                  "arenaBytes", "tempElems", "metaBytes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


SHARD_ROWS, SHARD_BLOCKS = 0, 1
GMRES_ORTH_DEFAULT, GMRES_ORTH_CGS2, GMRES_ORTH_MGS = 0, 1, 2


class BfhipGmresOptions(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("orthogonalization", C.c_uint32), ("tol", C.c_double),
                ("maxNumIter", C.c_size_t), ("solveM", C.c_void_p)]


class BfhipShardSpec(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("mode", C.c_uint32), ("numRowsGlobal", C.c_uint64),
                ("numSegments", C.c_uint32), ("reserved", C.c_uint32), ("segRows", C.c_void_p), ("segOwner", C.c_void_p),
                ("segGlobalOff", C.c_void_p)]


class BfhipPlanInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("structSize", "dtype", "elemSize", "epl", "xcap", "reserved")] + [
        (n, C.c_uint64) for n in ("numRows", "numCols", "numStages", "arenaElems", "tempElems", "numStagesT", "tempElemsT", "arenaElemsT")]


class BfhipStageView(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("reserved", C.c_uint32), ("numItems", C.c_uint64),
                ("numPieces", C.c_uint64), ("numReduce", C.c_uint64), ("items", C.c_void_p), ("pieces", C.c_void_p),
                ("numBundles", C.c_uint64), ("bundleBegin", C.c_void_p)]


class BfhipReduceView(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("destIsY", C.c_uint32), ("destOff", C.c_uint64),
                ("numRows", C.c_uint64), ("numIntervals", C.c_uint64), ("numSrc", C.c_uint64),
                ("rowInterval", C.c_void_p), ("ivBegin", C.c_void_p), ("srcBias", C.c_void_p)]


ITEM_DTYPE = np.dtype([("pieceBegin", "<u4"), ("numPieces", "<u4"), ("outOff", "<u4"), ("mrFlags", "<u4")])
PIECE_DTYPE = np.dtype([("dataOff", "<u8"), ("inOff", "<u4"), ("ncols", "<u4"), ("flags", "<u4"), ("ld", "<u4")])


# ---- include/bfhip_build.h -----------------------------------------------------
PTS_TREE, PTS_CIRCLE, PTS_TREE_TGT = 0, 1, 2
LEAF_KERNEL, LEAF_REEXP = 0, 1
LAYER_POTENTIALS = {"S": 1, "D": 2, "Sp": 3, "combined": 5}      # reference BfLayerPotential values (include/bf/layer_pot.h:27-42)

POINT_SET_DTYPE = np.dtype([("kind", "<u4"), ("count", "<u4"), ("first", "<u8"), ("cx", "<f8"), ("cy", "<f8"), ("r", "<f8")])
RECIPE_DTYPE = np.dtype([("node", "<u8"), ("kind", "<u4"), ("reserved", "<u4"),
                         ("src", POINT_SET_DTYPE), ("equiv", POINT_SET_DTYPE), ("tgt", POINT_SET_DTYPE)])   # = BfhipHelm2Recipe


class BfhipHelm2Problem(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("layerPot", C.c_uint32), ("wavenumber", C.c_double),
                ("points", C.c_void_p), ("numPoints", C.c_uint64), ("recipes", C.c_void_p),
                ("numRecipes", C.c_uint64), ("workspaceBytes", C.c_uint64),
                ("normals", C.c_void_p), ("colWeights", C.c_void_p), ("selfValue", C.c_double * 2),
                ("origIndex", C.c_void_p), ("krOrder", C.c_uint32), ("reserved", C.c_uint32),
                ("alpha", C.c_double * 2), ("beta", C.c_double * 2),
                ("tgtPoints", C.c_void_p), ("numTgtPoints", C.c_uint64), ("tgtNormals", C.c_void_p)]


class BfhipBuildStats(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("numBatches", C.c_uint32)] + [
        (n, C.c_uint64) for n in ("kernelLeaves", "reexpLeaves", "kernelEvals", "maxSweeps", "notConverged", "truncated", "sumSweeps")] + [
        ("seconds", C.c_double), ("qrProblems", C.c_uint64), ("qrColumns", C.c_uint64), ("qrRank", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "structSize"}


def _pts(rec, spec):
    if spec[0] in ("node", "tnode"):
        rec["kind"], rec["first"], rec["count"] = (PTS_TREE if spec[0] == "node" else PTS_TREE_TGT), spec[1], spec[2] - spec[1]
    elif spec[0] == "circle":
        rec["kind"], rec["cx"], rec["cy"], rec["r"], rec["count"] = PTS_CIRCLE, spec[1], spec[2], spec[3], spec[4]
    else:
        raise ValueError(spec)


def recipe_array(recipes: dict) -> np.ndarray:
    """helm2_structure recipes {leaf: ("kernel", src, tgt) | ("reexp", src, equiv, tgt)}
    -> BfhipHelm2Recipe[] in leaf order."""
    out = np.zeros(len(recipes), dtype=RECIPE_DTYPE)
    for i, node in enumerate(sorted(recipes)):
        rc = recipes[node]
        r = out[i]
        r["node"] = node
        if rc[0] == "kernel":
            r["kind"] = LEAF_KERNEL
            _pts(r["src"], rc[1])
            _pts(r["tgt"], rc[2])
        elif rc[0] == "reexp":
            r["kind"] = LEAF_REEXP
            _pts(r["src"], rc[1])
            _pts(r["equiv"], rc[2])
            _pts(r["tgt"], rc[3])
        else:
            raise ValueError(rc)
    return out


class Helm2Problem:
    """Keeps the arrays a BfhipHelm2Problem points to alive."""

    def __init__(self, points, wavenumber, recipes=None, workspace_bytes=0, layer_pot="S", normals=None,
                 col_weights=None, self_value=0.0, kr_order=0, orig_index=None, alpha=0.0, beta=0.0, tgt_points=None, tgt_normals=None):
        self.points = np.ascontiguousarray(points, dtype=np.float64)
        assert self.points.ndim == 2 and self.points.shape[1] == 2
        if recipes is None:
            recipes = np.zeros(0, dtype=RECIPE_DTYPE)
        self.recipes = recipes if isinstance(recipes, np.ndarray) else recipe_array(recipes)
        assert self.recipes.dtype == RECIPE_DTYPE and self.recipes.flags.c_contiguous
        self.normals = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
        self.col_weights = None if col_weights is None else np.ascontiguousarray(col_weights, dtype=np.float64)
        assert self.normals is None or self.normals.shape == self.points.shape
        assert self.col_weights is None or self.col_weights.shape == (len(self.points),)
        s = self.struct = BfhipHelm2Problem()
        s.structSize = C.sizeof(BfhipHelm2Problem)
        s.layerPot = LAYER_POTENTIALS.get(layer_pot, layer_pot)
        s.wavenumber = float(wavenumber)
        s.points = self.points.ctypes.data
        s.numPoints = len(self.points)
        s.recipes = self.recipes.ctypes.data if len(self.recipes) else None
        s.numRecipes = len(self.recipes)
        s.workspaceBytes = int(workspace_bytes)
        s.normals = None if self.normals is None else self.normals.ctypes.data
        s.colWeights = None if self.col_weights is None else self.col_weights.ctypes.data
        sv = complex(self_value)
        s.selfValue[0], s.selfValue[1] = sv.real, sv.imag
        self.orig_index = None if orig_index is None else np.ascontiguousarray(orig_index, dtype=np.uint64)
        assert self.orig_index is None or self.orig_index.shape == (len(self.points),)
        s.origIndex = None if self.orig_index is None else self.orig_index.ctypes.data
        s.krOrder = int(kr_order)
        al, be = complex(alpha), complex(beta)
        s.alpha[0], s.alpha[1], s.beta[0], s.beta[1] = al.real, al.imag, be.real, be.imag
        self.tgt_points = None if tgt_points is None else np.ascontiguousarray(tgt_points, dtype=np.float64)
        self.tgt_normals = None if tgt_normals is None else np.ascontiguousarray(tgt_normals, dtype=np.float64)
        s.tgtPoints = None if self.tgt_points is None else self.tgt_points.ctypes.data
        s.numTgtPoints = 0 if self.tgt_points is None else len(self.tgt_points)
        s.tgtNormals = None if self.tgt_normals is None else self.tgt_normals.ctypes.data

    def byref(self):
        return C.byref(self.struct)


class Helm2Layout:
    """bfhipHelm2LayoutCreate: the native (C) counterpart of helm2_structure.helm2_multilevel_structure.
    Exposes numpy copies of the descriptor arrays (`arrays()`, as Desc.arrays()), the recipes
    (RECIPE_DTYPE), the quadtree permutation and the tree-ordered points."""

    def __init__(self, points, wavenumber, tgt_points=None, single=None):
        """single = (src_path, tgt_path): the butterfly of one node pair (bfhipHelm2LayoutCreateSingle)."""
        lib = load()
        pts = np.ascontiguousarray(points, dtype=np.float64)
        tpts = None if tgt_points is None else np.ascontiguousarray(tgt_points, dtype=np.float64)
        h = C.c_void_p()
        if single is not None:
            sp, tp = (np.ascontiguousarray(p, dtype=np.uint32) for p in single)
            check(lib.bfhipHelm2LayoutCreateSingle(pts.ctypes.data, len(pts), float(wavenumber), sp.ctypes.data, len(sp), tp.ctypes.data, len(tp),
                                                   C.byref(h)))
        else:
            check(lib.bfhipHelm2LayoutCreate2(pts.ctypes.data, len(pts), None if tpts is None else tpts.ctypes.data,
                                              0 if tpts is None else len(tpts), float(wavenumber), C.byref(h)))
        try:
            d = lib.bfhipHelm2LayoutGetDesc(h).contents
            n, nch = int(d.numNodes), None

            def arr(ptr, count, dt):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (count * np.dtype(dt).itemsize,)).view(dt).copy()
            begin = arr(d.childBegin, n + 1, np.uint64)
            nch = int(begin[-1])
            self._arrays = dict(kind=arr(d.kind, n, np.uint8), rows=arr(d.rows, n, np.uint64), cols=arr(d.cols, n, np.uint64),
                                childBegin=begin, childNode=arr(d.childNode, nch, np.uint64), childRow0=arr(d.childRow0, nch, np.uint64),
                                childCol0=arr(d.childCol0, nch, np.uint64), blockKind=arr(d.blockKind, n, np.uint8))
            self.dtype, self.root, self.num_nodes = int(d.dtype), int(d.root), n
            rb, re = int(begin[self.root]), int(begin[self.root + 1])
            self.top_row_block = arr(d.topRowBlock, re - rb, np.uint64).astype(np.int64).tolist() if d.topRowBlock else None
            cnt = C.c_uint64(0)
            rp = lib.bfhipHelm2LayoutGetRecipes(h, C.byref(cnt))
            self.recipes = arr(rp, int(cnt.value), RECIPE_DTYPE) if cnt.value else np.zeros(0, dtype=RECIPE_DTYPE)
            self.perm = arr(lib.bfhipHelm2LayoutGetPerm(h), len(pts), np.uint64).astype(np.int64)
            self.tree_points = arr(lib.bfhipHelm2LayoutGetTreePoints(h), 2 * len(pts), np.float64).reshape(-1, 2)
            self.tgt_perm = self.tgt_tree_points = None
            if tpts is not None:
                self.tgt_perm = arr(lib.bfhipHelm2LayoutGetTgtPerm(h), len(tpts), np.uint64).astype(np.int64)
                self.tgt_tree_points = arr(lib.bfhipHelm2LayoutGetTgtTreePoints(h), 2 * len(tpts), np.float64).reshape(-1, 2)
        finally:
            lib.bfhipHelm2LayoutFree(C.byref(h))

    def arrays(self):
        return self._arrays


class BfhipStreamerSpec(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("colDepth", C.c_uint32), ("wmax", C.c_double), ("bandColumns", C.c_void_p),
                ("minNumRows", C.c_uint64), ("minNumCols", C.c_uint64), ("maxCols", C.c_uint64), ("alpha", C.c_double), ("delta", C.c_double)]


class BfhipStreamerStats(C.Structure):
    _fields_ = [("structSize", C.c_uint32), ("maxNest", C.c_uint32)] + [
        (n, C.c_uint64) for n in ("numRows", "numCols", "numFacs", "numW", "rowNodes", "product", "blockCoo", "blockDense", "blockDiag",
                                  "denseReal", "identity", "leafBytes", "svds", "merges", "feeds", "octreeDepth")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "structSize"}


class StreamerLayout:
    """bfhipStreamerLayoutCreate: the native (C) counterpart of streamer_structure.stream_structure + to_desc under the
    rank model.  Exposes numpy copies of the descriptor arrays, the octree permutation and the graph statistics."""

    def __init__(self, points, wmax, col_depth, band_columns, min_rows=20, min_cols=20, max_cols=None, alpha=1.75, delta=3.0):
        lib = load()
        pts = np.ascontiguousarray(points, dtype=np.float64)
        assert pts.ndim == 2 and pts.shape[1] == 3
        bands = np.ascontiguousarray(band_columns, dtype=np.uint64)
        assert len(bands) == 1 << col_depth
        spec = BfhipStreamerSpec()
        spec.structSize = C.sizeof(spec)
        spec.colDepth, spec.wmax, spec.bandColumns = int(col_depth), float(wmax), bands.ctypes.data
        spec.minNumRows, spec.minNumCols, spec.maxCols = int(min_rows), int(min_cols), int(max_cols or 0)
        spec.alpha, spec.delta = float(alpha), float(delta)
        h = C.c_void_p()
        check(lib.bfhipStreamerLayoutCreate(pts.ctypes.data, len(pts), C.byref(spec), C.byref(h)))
        try:
            d = lib.bfhipStreamerLayoutGetDesc(h).contents
            n = int(d.numNodes)

            def arr(ptr, count, dt):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (count * np.dtype(dt).itemsize,)).view(dt).copy()
            begin = arr(d.childBegin, n + 1, np.uint64)
            nch = int(begin[-1])
            self._arrays = dict(kind=arr(d.kind, n, np.uint8), rows=arr(d.rows, n, np.uint64), cols=arr(d.cols, n, np.uint64),
                                childBegin=begin, childNode=arr(d.childNode, nch, np.uint64), childRow0=arr(d.childRow0, nch, np.uint64),
                                childCol0=arr(d.childCol0, nch, np.uint64), blockKind=arr(d.blockKind, n, np.uint8))
            self.dtype, self.root, self.num_nodes = int(d.dtype), int(d.root), n
            self.perm = arr(lib.bfhipStreamerLayoutGetPerm(h), len(pts), np.uint64).astype(np.int64)
            st = BfhipStreamerStats()
            st.structSize = C.sizeof(st)
            check(lib.bfhipStreamerLayoutGetStats(h, C.byref(st)))
            self.stats = st.as_dict()
        finally:
            lib.bfhipStreamerLayoutFree(C.byref(h))

    def arrays(self):
        return self._arrays


class DescArrays:
    """Keeps the numpy arrays behind a BfhipDesc alive."""

    def __init__(self, desc, root=None, leaf_values=None):
        a = desc.arrays()
        self.arrays = a
        self.keep = []
        d = BfhipDesc()
        d.structSize = C.sizeof(BfhipDesc)
        d.dtype = desc.dtype
        d.numNodes = desc.num_nodes
        d.root = desc.root if root is None else root
        for f in ("kind", "rows", "cols", "childBegin", "childNode", "childRow0", "childCol0", "blockKind"):
            setattr(d, f, a[f].ctypes.data)
        if leaf_values is not None:
            ptrs = np.zeros(desc.num_nodes, dtype=np.uint64)
            for node, v in leaf_values.items():
                v = np.ascontiguousarray(v)
                self.keep.append(v)
                ptrs[node] = v.ctypes.data
            self.keep.append(ptrs)
            d.leafData = ptrs.ctypes.data
        if desc.top_row_block is not None and (root is None or root == desc.root):
            trb = np.asarray(desc.top_row_block, dtype=np.uint64)
            self.keep.append(trb)
            d.topRowBlock = trb.ctypes.data
        self.struct = d

    def byref(self):
        return C.byref(self.struct)


class BfhipError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")


_lib = None


def load():
    """Load libbfhip.so (built by `make -C butterfly_amd/csrc` / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the butterfly-apply engine)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.bfhipCompile.argtypes = [vp, C.POINTER(BfhipOptions), C.POINTER(vp)]
    lib.bfhipCompile.restype = C.c_int
    lib.bfhipCompileDesc.argtypes = [C.POINTER(BfhipDesc), C.POINTER(BfhipOptions), C.POINTER(vp)]
    lib.bfhipCompileDesc.restype = C.c_int
    lib.bfhipDescSubtreeLeafElems.argtypes = [C.POINTER(BfhipDesc), vp]
    lib.bfhipDescSubtreeLeafElems.restype = C.c_int
    lib.bfhipRowPartition.argtypes = [C.POINTER(BfhipDesc), C.c_uint32, vp, vp]
    lib.bfhipRowPartition.restype = C.c_int
    lib.bfhipRowPartitionMat.argtypes = [vp, C.c_uint32, vp, vp]
    lib.bfhipRowPartitionMat.restype = C.c_int
    lib.bfhipApply.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    lib.bfhipApply.restype = C.c_int
    lib.bfhipHostRegister.argtypes = [vp, C.c_size_t]
    lib.bfhipHostRegister.restype = C.c_int
    lib.bfhipHostUnregister.argtypes = [vp]
    lib.bfhipHostUnregister.restype = C.c_int
    lib.bfhipApplyDevice.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.bfhipApplyDevice.restype = C.c_int
    lib.bfhipApplyTranspose.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t]
    lib.bfhipApplyTranspose.restype = C.c_int
    lib.bfhipApplyTransposeDevice.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.bfhipApplyTransposeDevice.restype = C.c_int
    lib.bfhipSolveGMRES.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, C.c_double, C.c_size_t,
                                    C.POINTER(C.c_size_t), C.POINTER(C.c_double), vp, C.c_size_t]
    lib.bfhipSolveGMRES.restype = C.c_int
    lib.bfhipSolveGMRESDevice.argtypes = [vp, vp, C.c_size_t, vp, C.c_double, C.c_size_t, C.POINTER(C.c_size_t),
                                          C.POINTER(C.c_double), vp, vp]
    lib.bfhipSolveGMRESDevice.restype = C.c_int
    lib.bfhipSolveGMRESPrecondDevice.argtypes = [vp, vp, vp, C.c_size_t, vp, C.c_double, C.c_size_t, C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_double), vp, vp]
    lib.bfhipSolveGMRESPrecondDevice.restype = C.c_int
    lib.bfhipSolveGMRESOptsDevice.argtypes = [vp, C.POINTER(BfhipGmresOptions), vp, C.c_size_t, vp, C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_double), vp, vp]
    lib.bfhipSolveGMRESOptsDevice.restype = C.c_int
    lib.bfhipFlowStatus.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.bfhipFlowStatus.restype = C.c_int
    lib.bfhipGetStats.argtypes = [vp, C.POINTER(BfhipStats)]
    lib.bfhipGetStats.restype = C.c_int
    lib.bfhipGetNumRows.argtypes = [vp]
    lib.bfhipGetNumRows.restype = C.c_size_t
    lib.bfhipGetNumCols.argtypes = [vp]
    lib.bfhipGetNumCols.restype = C.c_size_t
    lib.bfhipNumBytes.argtypes = [vp]
    lib.bfhipNumBytes.restype = C.c_size_t
    lib.bfhipGetStageProfile.argtypes = [vp, vp, vp, vp, C.c_int]
    lib.bfhipGetStageProfile.restype = C.c_int
    lib.bfhipCovSampleDevice.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.bfhipCovSampleDevice.restype = C.c_int
    lib.bfhipCovMatvecDevice.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.bfhipCovMatvecDevice.restype = C.c_int
    lib.bfhipSetProfileSampling.argtypes = [vp, C.c_uint32]
    lib.bfhipSetProfileSampling.restype = C.c_int
    lib.bfhipPlanGetInfo.argtypes = [vp, C.POINTER(BfhipPlanInfo)]
    lib.bfhipPlanGetInfo.restype = C.c_int
    lib.bfhipPlanGetStage.argtypes = [vp, C.c_uint64, C.POINTER(BfhipStageView)]
    lib.bfhipPlanGetStage.restype = C.c_int
    lib.bfhipPlanGetReduce.argtypes = [vp, C.c_uint64, C.c_uint64, C.POINTER(BfhipReduceView)]
    lib.bfhipPlanGetReduce.restype = C.c_int
    lib.bfhipPlanPackArena.argtypes = [vp, vp]
    lib.bfhipPlanPackArena.restype = C.c_int
    lib.bfhipPlanPackArenaT.argtypes = [vp, vp]
    lib.bfhipPlanPackArenaT.restype = C.c_int
    lib.bfhipSave.argtypes = [vp, C.c_char_p]
    lib.bfhipSave.restype = C.c_int
    lib.bfhipLoad.argtypes = [C.c_char_p, C.POINTER(BfhipOptions), C.POINTER(vp)]
    lib.bfhipLoad.restype = C.c_int
    lib.bfhipFree.argtypes = [C.POINTER(vp)]
    lib.bfhipFree.restype = None
    lib.bfhipMatNew.argtypes = [vp, C.c_int]
    lib.bfhipMatNew.restype = vp
    lib.bfhipMatMulFunc.argtypes = [vp, vp]
    lib.bfhipMatMulFunc.restype = vp
    lib.bfhipSetErrorForwarding.argtypes = [C.c_int]
    lib.bfhipSetErrorForwarding.restype = None
    lib.bfhipCommGetUniqueId.argtypes = [vp]
    lib.bfhipCommGetUniqueId.restype = C.c_int
    lib.bfhipCommInitRank.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.bfhipCommInitRank.restype = C.c_int
    lib.bfhipCommDestroy.argtypes = [C.POINTER(vp)]
    lib.bfhipCommDestroy.restype = None
    lib.bfhipShardedCreate.argtypes = [vp, vp, C.POINTER(BfhipShardSpec), C.c_uint32, C.POINTER(vp)]
    lib.bfhipShardedCreate.restype = C.c_int
    lib.bfhipShardedApplyDevice.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.bfhipShardedApplyDevice.restype = C.c_int
    lib.bfhipShardedApplyTransposeDevice.argtypes = [vp, vp, C.c_size_t, vp, vp]
    lib.bfhipShardedApplyTransposeDevice.restype = C.c_int
    lib.bfhipShardedCovMatvecDevice.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.bfhipShardedCovMatvecDevice.restype = C.c_int
    lib.bfhipShardedSolveGMRESDevice.argtypes = [vp, C.POINTER(BfhipGmresOptions), vp, C.c_size_t, vp, C.POINTER(C.c_size_t),
                                                 C.POINTER(C.c_double), vp, vp]
    lib.bfhipShardedSolveGMRESDevice.restype = C.c_int
    lib.bfhipShardedGetNumRows.argtypes = [vp]
    lib.bfhipShardedGetNumRows.restype = C.c_size_t
    lib.bfhipShardedGetNumCols.argtypes = [vp]
    lib.bfhipShardedGetNumCols.restype = C.c_size_t
    lib.bfhipShardedMatNew.argtypes = [vp, C.c_int]
    lib.bfhipShardedMatNew.restype = vp
    lib.bfhipShardedLastTimes.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.bfhipShardedLastTimes.restype = C.c_int
    lib.bfhipShardedSetTiming.argtypes = [vp, C.c_int]
    lib.bfhipShardedSetTiming.restype = C.c_int
    lib.bfhipShardedFree.argtypes = [C.POINTER(vp)]
    lib.bfhipShardedFree.restype = None
    lib.bfhipErrorString.argtypes = [C.c_int]
    lib.bfhipErrorString.restype = C.c_char_p
    lib.bfhipLastErrorMessage.argtypes = []
    lib.bfhipLastErrorMessage.restype = C.c_char_p
    lib.bfhipSyntheticValue.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
    lib.bfhipSyntheticValue.restype = C.c_double
    lib.bfhipSyntheticLeafBases.argtypes = [C.POINTER(BfhipDesc), vp]
    lib.bfhipSyntheticLeafBases.restype = C.c_int
    lib.bfhipBuildHelm2.argtypes = [C.POINTER(BfhipDesc), C.POINTER(BfhipHelm2Problem), C.POINTER(BfhipOptions), C.POINTER(vp),
                                    C.POINTER(BfhipBuildStats)]
    lib.bfhipBuildHelm2.restype = C.c_int
    lib.bfhipHelm2BuildLeaf.argtypes = [C.POINTER(BfhipHelm2Problem), C.c_uint64, C.c_int, vp]
    lib.bfhipHelm2BuildLeaf.restype = C.c_int
    lib.bfhipHelm2DenseApplyDevice.argtypes = [C.POINTER(BfhipHelm2Problem), C.c_int, vp, vp, vp]
    lib.bfhipHelm2DenseApplyDevice.restype = C.c_int
    lib.bfhipHelm2DenseApply.argtypes = [C.POINTER(BfhipHelm2Problem), C.c_int, vp, vp]
    lib.bfhipHelm2DenseApply.restype = C.c_int
    lib.bfhipHelm2LayoutCreate.argtypes = [vp, C.c_uint64, C.c_double, C.POINTER(vp)]
    lib.bfhipHelm2LayoutCreate.restype = C.c_int
    lib.bfhipHelm2LayoutCreate2.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_double, C.POINTER(vp)]
    lib.bfhipHelm2LayoutCreate2.restype = C.c_int
    lib.bfhipHelm2LayoutCreateSingle.argtypes = [vp, C.c_uint64, C.c_double, vp, C.c_uint32, vp, C.c_uint32, C.POINTER(vp)]
    lib.bfhipHelm2LayoutCreateSingle.restype = C.c_int
    lib.bfhipHelm2LayoutGetTgtPerm.argtypes = [vp]
    lib.bfhipHelm2LayoutGetTgtPerm.restype = vp
    lib.bfhipHelm2LayoutGetTgtTreePoints.argtypes = [vp]
    lib.bfhipHelm2LayoutGetTgtTreePoints.restype = vp
    lib.bfhipHelm2LayoutGetDesc.argtypes = [vp]
    lib.bfhipHelm2LayoutGetDesc.restype = C.POINTER(BfhipDesc)
    lib.bfhipHelm2LayoutGetRecipes.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.bfhipHelm2LayoutGetRecipes.restype = vp
    lib.bfhipHelm2LayoutGetPerm.argtypes = [vp]
    lib.bfhipHelm2LayoutGetPerm.restype = vp
    lib.bfhipHelm2LayoutGetTreePoints.argtypes = [vp]
    lib.bfhipHelm2LayoutGetTreePoints.restype = vp
    lib.bfhipHelm2LayoutFree.argtypes = [C.POINTER(vp)]
    lib.bfhipHelm2LayoutFree.restype = None
    lib.bfhipStreamerLayoutCreate.argtypes = [vp, C.c_uint64, C.POINTER(BfhipStreamerSpec), C.POINTER(vp)]
    lib.bfhipStreamerLayoutCreate.restype = C.c_int
    lib.bfhipStreamerLayoutGetDesc.argtypes = [vp]
    lib.bfhipStreamerLayoutGetDesc.restype = C.POINTER(BfhipDesc)
    lib.bfhipStreamerLayoutGetPerm.argtypes = [vp]
    lib.bfhipStreamerLayoutGetPerm.restype = vp
    lib.bfhipStreamerLayoutGetStats.argtypes = [vp, C.POINTER(BfhipStreamerStats)]
    lib.bfhipStreamerLayoutGetStats.restype = C.c_int
    lib.bfhipStreamerLayoutFree.argtypes = [C.POINTER(vp)]
    lib.bfhipStreamerLayoutFree.restype = None
    lib.bfhipStreamerOctreeDepth.argtypes = [vp, C.c_uint64, C.POINTER(C.c_uint32)]
    lib.bfhipStreamerOctreeDepth.restype = C.c_int
    lib.bfhipFacHelm2MakeMultilevel.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(BfhipHelm2Problem), C.POINTER(BfhipOptions), C.POINTER(vp),
                                                vp, C.POINTER(BfhipBuildStats)]
    lib.bfhipFacHelm2MakeMultilevel.restype = C.c_int
    lib.bfhipFacHelm2MakeMultilevel2.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(BfhipHelm2Problem), C.POINTER(BfhipOptions),
                                                 C.POINTER(vp), vp, vp, C.POINTER(BfhipBuildStats)]
    lib.bfhipFacHelm2MakeMultilevel2.restype = C.c_int
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().bfhipLastErrorMessage().decode()
        raise BfhipError(code, msg)
