"""Host-side mirror of the engine's C-ABI (include/bfhip.h) for Python callers.

`HipOperator` owns a `BfhipOperator*`.  It is created either from a reference
`BfMat*` (the drop-in path: `bfhipCompile`) or from a flat structure
descriptor (`bfhipCompileDesc`, used for structure-exact synthetic operands
whose values are generated directly in HBM).  `apply` mirrors `bfMatMul`
(reference src/mat.c:183): Y = A X with X an N x nrhs row-major array.

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import BFHIP_C128, BFHIP_F32, BFHIP_F64, BfhipOptions, BfhipStats, DescArrays, check


def _options(device=-1, flags=0, max_rhs=1, demote_to_f32=False, seed=0, row_blocks=None, row_range=None):
    o = BfhipOptions()
    o.structSize = C.sizeof(BfhipOptions)
    o.device = device
    o.flags = flags
    o.maxRhs = max_rhs
    o.demoteToF32 = 1 if demote_to_f32 else 0
    o.seed = seed
    if row_blocks is not None:
        o.rowBlockBegin, o.rowBlockEnd = row_blocks
    if row_range is not None:
        o.rowBegin, o.rowEnd = row_range
    return o


class HipOperator:
    def __init__(self, handle, keep=()):
        self._h = C.c_void_p(handle)
        self._keep = list(keep)
        self._lib = _capi.load()

    # ---- construction ------------------------------------------------------
    @classmethod
    def from_bfmat(cls, bfmat_ptr, **opts):
        """Compile a reference BfMat object graph (bfhipCompile)."""
        lib = _capi.load()
        h = C.c_void_p()
        o = _options(**opts)
        check(lib.bfhipCompile(C.c_void_p(bfmat_ptr), C.byref(o), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_desc(cls, desc, leaf_values=None, root=None, **opts):
        """Compile a flat descriptor (bfhipCompileDesc); leaves without values are
        synthesized on the device from `seed`."""
        lib = _capi.load()
        da = DescArrays(desc, root=root, leaf_values=leaf_values)
        h = C.c_void_p()
        o = _options(**opts)
        check(lib.bfhipCompileDesc(da.byref(), C.byref(o), C.byref(h)))
        keep = [da] if (o.flags & _capi.FLAG_PLAN_ONLY) else []
        return cls(h.value, keep=keep)

    @classmethod
    def build_helm2(cls, desc, points, wavenumber, root=None, workspace_bytes=0, layer_pot="S", normals=None,
                    col_weights=None, self_value=0.0, kr_order=0, orig_index=None, alpha=0.0, beta=0.0, tgt_points=None,
                    tgt_normals=None, **opts):
        """bfhipBuildHelm2: lay out `desc` (helm2_structure with recipes=True) and
        compute every leaf on the device from its recipe.  `points` (and
        `normals` for layer_pot="Sp"): [N, 2] in quadtree order.  The operator
        built is  self_value * I + (K o KR) diag(col_weights), KR = Kapur-Rokhlin
        factors of order `kr_order` (needs orig_index = the quadtree permutation).
        Returns (operator, build statistics)."""
        lib = _capi.load()
        da = DescArrays(desc, root=root)
        recipes = getattr(desc, "recipe_array", None)
        prob = _capi.Helm2Problem(points, wavenumber, recipes if recipes is not None else desc.recipe, workspace_bytes, layer_pot, normals,
                                  col_weights, self_value,
                                  kr_order, orig_index, alpha, beta, tgt_points, tgt_normals)
        st = _capi.BfhipBuildStats()
        st.structSize = C.sizeof(st)
        h = C.c_void_p()
        o = _options(**opts)
        check(lib.bfhipBuildHelm2(da.byref(), prob.byref(), C.byref(o), C.byref(h), C.byref(st)))
        return cls(h.value), st.as_dict()

    @classmethod
    def fac_helm2_make_multilevel(cls, points, wavenumber, normals=None, col_weights=None, layer_pot="S", self_value=0.0,
                                  kr_order=0, alpha=0.0, beta=0.0, workspace_bytes=0, tgt_points=None, tgt_normals=None, **opts):
        """bfhipFacHelm2MakeMultilevel[2]: points (original order) -> device operator in one native
        call (C layout + device build).  Returns (operator, perm, build statistics), or with
        `tgt_points` (operator, (perm, tgt_perm), statistics); the operator maps x[perm] to
        y[tgt_perm] (quadtree orders)."""
        lib = _capi.load()
        f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        ptr = lambda a: None if a is None else a.ctypes.data
        pts, nrm, w, tpts, tnrm = f64(points), f64(normals), f64(col_weights), f64(tgt_points), f64(tgt_normals)
        params = _capi.Helm2Problem(pts, wavenumber, None, workspace_bytes, layer_pot, None, None, self_value, kr_order, None, alpha, beta)
        st = _capi.BfhipBuildStats()
        st.structSize = C.sizeof(st)
        perm = np.empty(len(pts), dtype=np.uint64)
        tperm = np.empty(0 if tpts is None else len(tpts), dtype=np.uint64)
        h = C.c_void_p()
        o = _options(**opts)
        check(lib.bfhipFacHelm2MakeMultilevel2(ptr(pts), ptr(nrm), ptr(w), len(pts), ptr(tpts), ptr(tnrm), len(tperm), params.byref(), C.byref(o),
                                               C.byref(h), perm.ctypes.data, tperm.ctypes.data if len(tperm) else None, C.byref(st)))
        perm = perm.astype(np.int64)
        return cls(h.value), (perm if tpts is None else (perm, tperm.astype(np.int64))), st.as_dict()

    @classmethod
    def load(cls, path, **opts):
        """bfhipLoad: a previously saved operator, straight into HBM."""
        lib = _capi.load()
        h = C.c_void_p()
        o = _options(**opts)
        check(lib.bfhipLoad(str(path).encode(), C.byref(o), C.byref(h)))
        return cls(h.value)

    def save(self, path):
        """bfhipSave: packed leaf arena + index metadata of the compiled operator."""
        check(self._lib.bfhipSave(self._h, str(path).encode()))

    def close(self):
        if self._h:
            self._lib.bfhipFree(C.byref(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- introspection -----------------------------------------------------
    @property
    def handle(self):
        return self._h

    @property
    def shape(self):
        return int(self._lib.bfhipGetNumRows(self._h)), int(self._lib.bfhipGetNumCols(self._h))

    def num_bytes(self):
        return int(self._lib.bfhipNumBytes(self._h))

    def stats(self):
        st = BfhipStats()
        st.structSize = C.sizeof(BfhipStats)
        check(self._lib.bfhipGetStats(self._h, C.byref(st)))
        return st.as_dict()

    def flow_status(self):
        """(applies of 1 - 2 RHS run as one dependency-driven launch, one of its waits ever gave up) -- bfhipFlowStatus."""
        en, bad = C.c_uint32(0), C.c_uint32(0)
        check(self._lib.bfhipFlowStatus(self._h, C.byref(en), C.byref(bad)))
        return bool(en.value), bool(bad.value)

    @property
    def dtype(self):
        return self.stats()["dtype"]

    def np_dtype(self):
        return {BFHIP_C128: np.complex128, BFHIP_F64: np.float64, BFHIP_F32: np.float32}[self.dtype]

    # ---- apply -------------------------------------------------------------
    def apply_host(self, x: np.ndarray) -> np.ndarray:
        """bfhipApply on host arrays (H2D, all stages, D2H)."""
        m, n = self.shape
        src_dtype = np.complex128 if self.dtype == BFHIP_C128 else np.float64
        x2 = np.ascontiguousarray(x, dtype=src_dtype)
        one_d = x2.ndim == 1
        if one_d:
            x2 = x2[:, None]
        if x2.shape[0] != n:
            raise ValueError(f"operator has {n} columns, x has {x2.shape[0]} rows")
        nrhs = x2.shape[1]
        y = np.empty((m, nrhs), dtype=src_dtype)
        check(self._lib.bfhipApply(self._h, x2.ctypes.data, nrhs, nrhs, y.ctypes.data, nrhs))
        return y[:, 0] if one_d else y

    def apply_host_into(self, x: np.ndarray, y: np.ndarray, nrhs=1):
        """bfhipApply on caller-owned, densely packed host arrays (nothing is allocated: what an unmodified caller's loop costs)."""
        check(self._lib.bfhipApply(self._h, x.ctypes.data, nrhs, nrhs, y.ctypes.data, nrhs))
        return y

    def apply_pointers(self, xptr, yptr, nrhs=1):
        """bfhipApply on raw pointers of any kind (device, pinned / registered host, pageable): include/bfhip.h says what each costs."""
        check(self._lib.bfhipApply(self._h, C.c_void_p(xptr), nrhs, nrhs, C.c_void_p(yptr), nrhs))

    @staticmethod
    def host_register(a: np.ndarray):
        check(_capi.load().bfhipHostRegister(C.c_void_p(a.ctypes.data), a.nbytes))

    @staticmethod
    def host_unregister(a: np.ndarray):
        check(_capi.load().bfhipHostUnregister(C.c_void_p(a.ctypes.data)))

    def apply_device(self, x, y=None, stream=None):
        """bfhipApplyDevice on torch tensors resident on the operator's GPU.
        x: [numCols] or [numCols, nrhs], contiguous; returns y (async on the
        current torch stream unless `stream` is given)."""
        import torch
        m, n = self.shape
        tdt = {BFHIP_C128: torch.complex128, BFHIP_F64: torch.float64, BFHIP_F32: torch.float32}[self.dtype]
        if x.dtype != tdt or not x.is_cuda or not x.is_contiguous():
            raise ValueError(f"x must be a contiguous CUDA tensor of dtype {tdt}")
        nrhs = 1 if x.dim() == 1 else x.shape[1]
        if x.shape[0] != n:
            raise ValueError(f"operator has {n} columns, x has {x.shape[0]} rows")
        if y is None:
            y = torch.empty((m,) if x.dim() == 1 else (m, nrhs), dtype=tdt, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        check(self._lib.bfhipApplyDevice(self._h, C.c_void_p(x.data_ptr()), nrhs, C.c_void_p(y.data_ptr()),
                                         C.c_void_p(s.cuda_stream)))
        return y

    def apply_transpose_host(self, x: np.ndarray) -> np.ndarray:
        """bfhipApplyTranspose: y = A^T x (plain transpose) on host arrays; the
        operator must have been compiled with FLAG_ADJOINT."""
        m, n = self.shape
        src_dtype = np.complex128 if self.dtype == BFHIP_C128 else np.float64
        x2 = np.ascontiguousarray(x, dtype=src_dtype)
        one_d = x2.ndim == 1
        if one_d:
            x2 = x2[:, None]
        if x2.shape[0] != m:
            raise ValueError(f"operator has {m} rows, x has {x2.shape[0]} rows")
        nrhs = x2.shape[1]
        y = np.empty((n, nrhs), dtype=src_dtype)
        check(self._lib.bfhipApplyTranspose(self._h, x2.ctypes.data, nrhs, nrhs, y.ctypes.data, nrhs))
        return y[:, 0] if one_d else y

    def apply_transpose_device(self, x, y=None, stream=None):
        import torch
        m, n = self.shape
        nrhs = 1 if x.dim() == 1 else x.shape[1]
        if y is None:
            y = torch.empty((n,) if x.dim() == 1 else (n, nrhs), dtype=x.dtype, device=x.device)
        s = stream if stream is not None else torch.cuda.current_stream(x.device)
        check(self._lib.bfhipApplyTransposeDevice(self._h, C.c_void_p(x.data_ptr()), nrhs, C.c_void_p(y.data_ptr()),
                                                  C.c_void_p(s.cuda_stream)))
        return y

    # ---- GMRES ---------------------------------------------------------------
    def solve_gmres(self, b: np.ndarray, x0=None, tol=1e-12, max_num_iter=100):
        """bfhipSolveGMRES on host arrays; returns (x, num_iter, residual), as the
        reference's bfSolveGMRES(A, B, X0, tol, maxNumIter, &numIter, NULL)."""
        n = self.shape[0]
        b2 = np.ascontiguousarray(b, dtype=np.complex128)
        one_d = b2.ndim == 1
        if one_d:
            b2 = b2[:, None]
        nrhs = b2.shape[1]
        x = np.empty((n, nrhs), dtype=np.complex128)
        x0p = None
        if x0 is not None:
            x02 = np.ascontiguousarray(x0, dtype=np.complex128).reshape(n, nrhs)
            x0p = x02.ctypes.data
        it = C.c_size_t(0)
        res = C.c_double(0)
        check(self._lib.bfhipSolveGMRES(self._h, b2.ctypes.data, nrhs, nrhs, x0p, nrhs, tol, max_num_iter,
                                        C.byref(it), C.byref(res), x.ctypes.data, nrhs))
        return (x[:, 0] if one_d else x), int(it.value), float(res.value)

    def solve_gmres_device(self, b, x0=None, tol=1e-12, max_num_iter=100, precond=None, orth="default"):
        """Device-resident form on torch tensors; returns (x, num_iter, residual).  `precond`: a HipOperator
        applying the action of M^{-1} (the reference's left preconditioner M through bfMatSolve).  `orth`:
        "cgs2" (batched, the default), "mgs" (the reference's order, src/linalg.c:174-184) or "default"
        (cgs2 unless BFHIP_GMRES_MGS=1 is set)."""
        import torch
        nrhs = 1 if b.dim() == 1 else b.shape[1]
        x = torch.empty_like(b)
        it = C.c_size_t(0)
        res = C.c_double(0)
        s = torch.cuda.current_stream(b.device)
        o = _capi.BfhipGmresOptions()
        o.structSize = C.sizeof(o)
        o.orthogonalization = {"default": _capi.GMRES_ORTH_DEFAULT, "cgs2": _capi.GMRES_ORTH_CGS2, "mgs": _capi.GMRES_ORTH_MGS}[orth]
        o.tol, o.maxNumIter = tol, max_num_iter
        o.solveM = precond.handle if precond is not None else None
        check(self._lib.bfhipSolveGMRESOptsDevice(self._h, C.byref(o), C.c_void_p(b.data_ptr()), nrhs,
                                                  C.c_void_p(x0.data_ptr()) if x0 is not None else None,
                                                  C.byref(it), C.byref(res), C.c_void_p(x.data_ptr()), C.c_void_p(s.cuda_stream)))
        return x, int(it.value), float(res.value)

    def stage_profile(self, reset=False):
        """(ms, launches, bytes) per stage, from hipEvents (needs FLAG_PROFILE)."""
        S = self.stats()["numStages"]
        ms = np.zeros(S, dtype=np.float64)
        launches = np.zeros(S, dtype=np.uint64)
        nbytes = np.zeros(S, dtype=np.uint64)
        check(self._lib.bfhipGetStageProfile(self._h, ms.ctypes.data, launches.ctypes.data, nbytes.ctypes.data,
                                             1 if reset else 0))
        return ms, launches, nbytes

    def cov_sample_device(self, gamma_lam, row_perm, w):
        """z = P A diag(gamma_lam) w on the device (sample_z of examples/covariance/lbo_cov.c:36-45); torch tensors on
        this operator's device, gamma_lam / row_perm (int64, scatter order) may be None."""
        import torch
        z = torch.empty(self.stats()["numRows"], dtype=w.dtype, device=w.device)
        s = torch.cuda.current_stream(w.device)
        check(self._lib.bfhipCovSampleDevice(self._h, C.c_void_p(gamma_lam.data_ptr()) if gamma_lam is not None else None,
                                             C.c_void_p(row_perm.data_ptr()) if row_perm is not None else None,
                                             C.c_void_p(w.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(s.cuda_stream)))
        return z

    def cov_matvec_device(self, gamma_lam, row_perm, rev_row_perm, v):
        """z = P A diag(gamma_lam)^2 A^T P' v on the device (cov_matvec of examples/covariance/lbo_cov.c:48-60)."""
        import torch
        z = torch.empty_like(v)
        s = torch.cuda.current_stream(v.device)
        check(self._lib.bfhipCovMatvecDevice(self._h, C.c_void_p(gamma_lam.data_ptr()) if gamma_lam is not None else None,
                                             C.c_void_p(row_perm.data_ptr()) if row_perm is not None else None,
                                             C.c_void_p(rev_row_perm.data_ptr()) if rev_row_perm is not None else None,
                                             C.c_void_p(v.data_ptr()), C.c_void_p(z.data_ptr()), C.c_void_p(s.cuda_stream)))
        return z

    def set_profile_sampling(self, every):
        """Bracket one apply in `every` with events (bfhipSetProfileSampling)."""
        check(self._lib.bfhipSetProfileSampling(self._h, int(every)))

    # ---- reference-vtable shim ---------------------------------------------
    def as_bfmat(self, owns=False):
        """A BfMat* whose Mul / MulVec / RmulVec run on the device (bfhipMatNew).  With
        owns=False the shim does not own the operator (keep this object alive while it is
        used); with owns=True the operator is handed over to the shim -- its Delete slot frees
        it -- and this object is left closed."""
        p = self._lib.bfhipMatNew(self._h, 1 if owns else 0)
        if not p:
            raise _capi.BfhipError(1, self._lib.bfhipLastErrorMessage().decode())
        if owns:
            self._h = C.c_void_p(None)
        return p


def helm2_build_leaf(points, wavenumber, recipe, device=-1, **problem) -> np.ndarray:
    """One leaf of a Helmholtz butterfly computed on the device (bfhipHelm2BuildLeaf);
    `recipe` as in helm2_structure ("kernel", src, tgt) / ("reexp", src, equiv, tgt);
    `problem`: layer_pot, normals, col_weights, self_value."""
    prob = _capi.Helm2Problem(points, wavenumber, {0: recipe}, **problem)
    r = prob.recipes[0]
    rows = int(r["tgt"]["count"] if r["kind"] == _capi.LEAF_KERNEL else r["equiv"]["count"])
    out = np.empty((rows, int(r["src"]["count"])), dtype=np.complex128)
    check(_capi.load().bfhipHelm2BuildLeaf(prob.byref(), 0, device, out.ctypes.data))
    return out


def helm2_dense_apply(points, wavenumber, x, device=-1, **problem):
    """y = (self_value I + K diag(col_weights)) x with the dense layer-potential
    matrix K evaluated on the fly on the device (bfhipHelm2DenseApply[Device]);
    x: numpy [N] or a CUDA tensor; `problem`: layer_pot, normals, col_weights, self_value."""
    prob = _capi.Helm2Problem(points, wavenumber, None, **problem)
    lib = _capi.load()
    m = len(prob.tgt_points) if prob.tgt_points is not None else len(prob.points)
    if isinstance(x, np.ndarray):
        xs = np.ascontiguousarray(x, dtype=np.complex128)
        y = np.empty(m, dtype=np.complex128)
        check(lib.bfhipHelm2DenseApply(prob.byref(), device, xs.ctypes.data, y.ctypes.data))
        return y
    import torch
    y = torch.empty(m, dtype=x.dtype, device=x.device)
    s = torch.cuda.current_stream(x.device)
    check(lib.bfhipHelm2DenseApplyDevice(prob.byref(), x.device.index, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                         C.c_void_p(s.cuda_stream)))
    return y
