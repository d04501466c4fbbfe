"""Generator of tests/golden/sphere_phi_500x32.npz: the one data set the reference's own tests hold next to the
streamed-butterfly path -- /root/reference/tests/sphere_Phi.txt (500 x 32 Laplace-Beltrami eigenvectors of a sphere
mesh, FEM discretization), sphere_Lam.txt (their 32 eigenvalues in [50, 100]) and the 500 mesh vertices of sphere.obj
(a Fibonacci sphere; tests/generate_data_for_test_linalg.py of the reference wrote all three, test_linalg.c:21-22 reads
the first two).  This is the matrix family examples/covariance/lbo_cov.c:139-189 streams through bfFacStreamerFeed.
DATA only is copied (numbers), no source text.

    python tests/golden/make_sphere_phi_fixture.py      # needs /root/reference (the build container)
"""
import os

import numpy as np

REF = "/root/reference/tests"
HERE = os.path.dirname(os.path.abspath(__file__))

phi = np.loadtxt(os.path.join(REF, "sphere_Phi.txt"))
lam = np.loadtxt(os.path.join(REF, "sphere_Lam.txt"))
verts = np.array([[float(t) for t in ln.split()[1:4]] for ln in open(os.path.join(REF, "sphere.obj")) if ln.startswith("v ")])
assert phi.shape == (500, 32) and lam.shape == (32,) and verts.shape == (500, 3)
assert np.all(np.diff(lam) >= 0) and 50 <= lam[0] and lam[-1] <= 100
assert np.allclose(np.linalg.norm(verts, axis=1), 1.0, atol=1e-12)
np.savez_compressed(os.path.join(HERE, "sphere_phi_500x32.npz"), phi=phi, lam=lam, points=verts)
print("wrote sphere_phi_500x32.npz", phi.shape, lam[[0, -1]])
