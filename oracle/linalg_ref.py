"""ORACLE (test infrastructure, not product code).

Numpy restatement of the reference's GMRES, the production caller of the
apply path:

  bfSolveGMRES                       reference src/linalg.c:47-317
  column dots (zdotc: conj(V_i)^T W) src/mat_dense_complex.c:88-131
  Givens rotation (SciPy/templates)  src/vec_complex.c:230-292
  applying a rotation                src/vec_complex.c:155-168
  back substitution (ztrsv, upper)   src/mat_dense_complex.c:1306-1325

Restated quirks, kept on purpose so the device solver can be compared
iteration for iteration:
  * no restarts; modified Gram-Schmidt in the order i = 0..j;
  * beta = max over right-hand sides of ||r_p||, residual = max_p |s_{j+1,p}| / beta;
  * on convergence at iteration j the loop breaks *before* j is incremented, so the
    solution is built from j (not j+1) Krylov vectors (linalg.c:228-243 use `j`);
  * numIter reports that same j.

PARITY STATUS: see oracle/bfref.h.
"""
from __future__ import annotations

import numpy as np


def givens(a, b):
    """bfVecComplexGetGivensRotation, src/vec_complex.c:275-286."""
    if abs(b) == 0:
        return 1.0 + 0j, 0.0 + 0j
    if abs(b) > abs(a):
        tmp = -a / b
        s = 1 / np.sqrt(1 + abs(tmp) ** 2)
        return tmp * s, s + 0j
    tmp = -b / a
    c = 1 / np.sqrt(1 + abs(tmp) ** 2)
    return c + 0j, tmp * c


def apply_givens(vec, i0, i1, c, s):
    """mulInplace_givensComplex, src/vec_complex.c:155-168."""
    z0, z1 = vec[i0], vec[i1]
    vec[i0] = np.conj(c) * z0 - s * z1
    vec[i1] = s * z0 + c * z1


def solve_gmres(matmul, B, X0=None, tol=1e-12, max_num_iter=100, msolve=None):
    """Returns (X, num_iter, residual_history).  `matmul(X)` is bfMatMul(A, X)
    for an n x nrhs complex array; `msolve(X)` is bfMatSolve(M, X) for the left
    preconditioner M (src/linalg.c:90-97,131-135,157-163), None without one."""
    if msolve is not None:
        plain = matmul
        matmul = lambda X: msolve(plain(X))                 # W = M^{-1} (A V[j])  :157-163
    B = np.asarray(B, dtype=np.complex128)
    one_d = B.ndim == 1
    if one_d:
        B = B[:, None]
    n, nrhs = B.shape
    X0 = np.zeros_like(B) if X0 is None else np.asarray(X0, dtype=np.complex128).reshape(n, nrhs)
    R = B - matmul(X0) if msolve is None else msolve(B - plain(X0))   # :127-135
    rnorm = np.linalg.norm(R, axis=0)                    # :139
    beta = rnorm.max()                                   # :142
    V = [R / rnorm]                                      # :145-146
    S = np.zeros((max_num_iter + 1, nrhs), dtype=np.complex128)
    S[0] = rnorm                                         # :150-151
    H = []
    J = {}
    history = []
    converged = False
    j = 0
    for j in range(max_num_iter):
        W = matmul(V[j])                                 # :157
        Hj = np.zeros((j + 2, nrhs), dtype=np.complex128)
        for i in range(j + 1):                           # modified Gram-Schmidt :174-184
            hij = np.einsum("ij,ij->j", V[i].conj(), W)  # zdotc
            Hj[i] = hij
            W = W - V[i] * hij
        wnorm = np.linalg.norm(W, axis=0)                # :186
        Hj[j + 1] = wnorm
        V.append(W / wnorm)                              # :197-198
        for i in range(j):                               # earlier rotations :206-212
            for p in range(nrhs):
                c, s = J[(i, p)]
                col = Hj[:, p]
                apply_givens(col, i, i + 1, c, s)
        for p in range(nrhs):                            # new rotation :214-219
            col = Hj[:, p]
            J[(j, p)] = givens(col[j], col[j + 1])
            apply_givens(col, j, j + 1, *J[(j, p)])
            scol = S[:, p]
            apply_givens(scol, j, j + 1, *J[(j, p)])     # :222-228
        H.append(Hj)
        residual = np.abs(S[j + 1]).max() / beta         # :230-231
        history.append(float(residual))
        if residual < tol:                               # :235-241
            converged = True
            break
    else:
        j = max_num_iter
    if not converged:
        j = max_num_iter
    X = np.empty_like(B)
    for p in range(nrhs):                                # :245-285
        Hp = np.zeros((j, j), dtype=np.complex128)
        for i in range(j):
            Hp[: i + 1, i] = H[i][: i + 1, p]
        y = np.zeros(j, dtype=np.complex128)
        for r in range(j - 1, -1, -1):                   # ztrsv, upper, non-unit
            y[r] = (S[r, p] - Hp[r, r + 1:] @ y[r + 1:]) / Hp[r, r]
        x = X0[:, p].copy()
        for i in range(j):
            x = x + V[i][:, p] * y[i]
        X[:, p] = x
    return (X[:, 0] if one_d else X), j, history
