/* helm2_bie_device.c -- the reference's BIE driver (examples/simple/helm2_bie.c) against libbfhip.so
 * alone, from plain C: no reference library, no Python.
 *
 *   exterior Neumann problem on the unit circle, field of an interior point source:
 *   (I/2 + S' KR6 W) sigma = dn G(. - x0);   u = S W sigma  must equal  G(. - x0)  outside.
 *
 * Every heavy step runs on the MI355X: layout + values of the S' butterfly with the Kapur-Rokhlin
 * correction, the trapezoid weights and the I/2 term folded in (bfhipFacHelm2MakeMultilevel),
 * unrestarted GMRES (bfhipSolveGMRES), the evaluation butterfly to the exterior targets
 * (bfhipFacHelm2MakeMultilevel2) and its apply (bfhipApply).
 *
 *   gcc -O2 -std=gnu11 -Iinclude examples/helm2_bie_device.c -Lbutterfly_amd/csrc -lbfhip -lm \
 *       -Wl,-rpath,$PWD/butterfly_amd/csrc -o /tmp/helm2_bie_device && /tmp/helm2_bie_device 16384 64
 */
#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <bfhip_build.h>

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s: %s (%s)\n", #call, bfhipErrorString(rc_), bfhipLastErrorMessage()); return 1; } } while (0)

static double complex G(double k, double dx, double dy) { double r = hypot(dx, dy); return 0.25 * I * (j0(k * r) + I * y0(k * r)); }
static double complex Sp(double k, double dx, double dy, double nx, double ny) {      /* src/helm2.c:44-51 */
  double r = hypot(dx, dy);
  return 0.25 * I * k * (j1(k * r) + I * y1(k * r)) / r * (nx * dx + ny * dy);
}

int main(int argc, char **argv) {
  size_t const n = argc > 1 ? (size_t)atol(argv[1]) : 16384, m = n / 4;
  double const k = argc > 2 ? atof(argv[2]) : 64.0, x0 = 0.1, y0_ = 0.2;
  double *pts = malloc(n * 16), *w = malloc(n * 8), *tgt = malloc(m * 16);
  uint64_t *perm = malloc(n * 8), *sperm = malloc(n * 8), *tperm = malloc(m * 8);
  double complex *b = malloc(n * 16), *sigma = malloc(n * 16), *sig2 = malloc(n * 16), *phi = malloc(m * 16);
  if (!pts || !w || !tgt || !perm || !sperm || !tperm || !b || !sigma || !sig2 || !phi) return 2;
  for (size_t i = 0; i < n; ++i) { double t = 2 * M_PI * i / n; pts[2 * i] = cos(t); pts[2 * i + 1] = sin(t); w[i] = 2 * M_PI / n; }
  for (size_t i = 0; i < m; ++i) { double t = 2 * M_PI * i / m; tgt[2 * i] = 2 * cos(t); tgt[2 * i + 1] = 2 * sin(t); }

  /* system matrix: normals of the unit circle are the points themselves */
  BfhipHelm2Problem prm = {.structSize = sizeof prm, .layerPot = BFHIP_LAYER_POTENTIAL_PV_NORMAL_DERIV_SINGLE, .wavenumber = k,
                           .selfValue = {0.5, 0.0}, .krOrder = 6};
  BfhipOperator *A = NULL, *E = NULL;
  BfhipBuildStats st = {.structSize = sizeof st};
  CHECK(bfhipFacHelm2MakeMultilevel(pts, pts, w, n, &prm, NULL, &A, perm, &st));
  printf("system matrix: %zu x %zu, %.1f MB of leaves, built in %.2f s (%llu least-squares problems)\n", bfhipGetNumRows(A), bfhipGetNumCols(A),
         bfhipNumBytes(A) / 1e6, st.seconds, (unsigned long long)st.reexpLeaves);

  /* right-hand side in quadtree order, solve */
  for (size_t t = 0; t < n; ++t) { double const *p = &pts[2 * perm[t]]; b[t] = Sp(k, p[0] - x0, p[1] - y0_, p[0], p[1]); }
  size_t iters = 0;
  double res = 0;
  CHECK(bfhipSolveGMRES(A, b, 1, 1, NULL, 0, 1e-10, 256, &iters, &res, sigma, 1));
  printf("GMRES: %zu iterations, residual %.2e\n", iters, res);

  /* evaluation butterfly: boundary (weights folded) -> exterior targets; its column order is its own
   * source permutation, so go through the original order */
  BfhipHelm2Problem evp = {.structSize = sizeof evp, .layerPot = BFHIP_LAYER_POTENTIAL_SINGLE, .wavenumber = k};
  CHECK(bfhipFacHelm2MakeMultilevel2(pts, NULL, w, n, tgt, NULL, m, &evp, NULL, &E, sperm, tperm, NULL));
  for (size_t t = 0; t < n; ++t) sig2[perm[t]] = sigma[t];                        /* tree -> original */
  for (size_t t = 0; t < n; ++t) b[t] = sig2[sperm[t]];                           /* original -> E's column order */
  CHECK(bfhipApply(E, b, 1, 1, phi, 1));
  double num = 0, den = 0;
  for (size_t t = 0; t < m; ++t) {
    double const *p = &tgt[2 * tperm[t]];
    double complex const ex = G(k, p[0] - x0, p[1] - y0_);
    num += pow(cabs(phi[t] - ex), 2); den += pow(cabs(ex), 2);
  }
  double const err = sqrt(num / den);
  printf("exterior field at %zu targets: rel l2 error %.2e\n", m, err);
  bfhipFree(&A); bfhipFree(&E);
  return err < 1e-6 ? 0 : 3;
}
