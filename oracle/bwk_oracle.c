/*
 * oracle/bwk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the second mini-app of the reference, the HOMME
 * spectral-element kernel `biharmonic_wk_scalar`:
 *   atmosphere/biharmonic_wk_kernel.F90
 *     :109-134  gradient_sphere          :138-160  divergence_sphere_wk
 *     :164-182  laplace_sphere_wk        :186-200  biharmonic_wk_scalar (CPU)
 *     :48-58, :77-91  initialize_data / myrandom (the reference's own portable LCG inputs)
 * It is the parity checker of libbwk_hip.so and the "port" CPU baseline; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  tests/test_bwk_oracle.py checks it bit-for-bit against outputs
 * of the reference program itself (oracle/build_ref.py --bwk, amdflang -O3
 * -ffp-contract=off; fixtures tests/golden/bwk_*).  Build with -ffp-contract=off.
 *
 * Layouts = the reference's (Fortran column-major):
 *   qtens(np,np,nlev,qsize,nelemd)   inout
 *   dvv(np,np)                        deriv%Dvv
 *   elem(144,nelemd): per element Dinv(np,np,2,2) | spheremp(np,np) | tensorVisc(np,np,2,2)
 *                     in declaration order (:23-27), 64 + 16 + 64 doubles
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NP 4
#define ELEM_DOUBLES 144
/* :14 -- the literal has no kind suffix: it is a default-real (fp32) constant converted to
 * real(8), like eps of the MPDATA routine */
static const double rrearth = (double)0.00000016666666666666f;

/* 0-based accessors into Fortran-ordered blocks */
#define A2(p, i, j) (p)[(i) + NP * (j)]
#define A3(p, i, j, c) (p)[(i) + NP * ((j) + NP * (c))]
#define A4(p, i, j, a, b) (p)[(i) + NP * ((j) + NP * ((a) + 2 * (b)))]

/* :109-134 */
static void gradient_sphere(const double *s, const double *dvv, const double *Dinv, double *ds) {
  double v1[NP * NP], v2[NP * NP];
  int i, j, l;
  for (j = 0; j < NP; j++)
    for (l = 0; l < NP; l++) {
      double dsdx00 = 0.0, dsdy00 = 0.0;
      for (i = 0; i < NP; i++) {
        dsdx00 = dsdx00 + A2(dvv, i, l) * A2(s, i, j);
        dsdy00 = dsdy00 + A2(dvv, i, l) * A2(s, j, i);
      }
      A2(v1, l, j) = dsdx00 * rrearth;
      A2(v2, j, l) = dsdy00 * rrearth;
    }
  for (j = 0; j < NP; j++)
    for (i = 0; i < NP; i++) {
      A3(ds, i, j, 0) = A4(Dinv, i, j, 0, 0) * A2(v1, i, j) + A4(Dinv, i, j, 1, 0) * A2(v2, i, j);
      A3(ds, i, j, 1) = A4(Dinv, i, j, 0, 1) * A2(v1, i, j) + A4(Dinv, i, j, 1, 1) * A2(v2, i, j);
    }
}

/* :138-160 */
static void divergence_sphere_wk(const double *v, const double *dvv, const double *Dinv,
                                 const double *spheremp, double *div) {
  double vtemp[NP * NP * 2];
  int i, j, m, n;
  for (j = 0; j < NP; j++)
    for (i = 0; i < NP; i++) {
      A3(vtemp, i, j, 0) = (A4(Dinv, i, j, 0, 0) * A3(v, i, j, 0) + A4(Dinv, i, j, 0, 1) * A3(v, i, j, 1));
      A3(vtemp, i, j, 1) = (A4(Dinv, i, j, 1, 0) * A3(v, i, j, 0) + A4(Dinv, i, j, 1, 1) * A3(v, i, j, 1));
    }
  for (n = 0; n < NP; n++)
    for (m = 0; m < NP; m++) {
      double acc = 0.0;
      for (j = 0; j < NP; j++)
        acc = acc - (A2(spheremp, j, n) * A3(vtemp, j, n, 0) * A2(dvv, m, j) +
                     A2(spheremp, m, j) * A3(vtemp, m, j, 1) * A2(dvv, n, j)) * rrearth;
      A2(div, m, n) = acc;
    }
}

/* :164-182 */
static void laplace_sphere_wk(const double *s, const double *dvv, const double *el, double *out) {
  const double *Dinv = el, *spheremp = el + 64, *tv = el + 80;
  double grads[NP * NP * 2], old[NP * NP * 2];
  int i, j, c;
  gradient_sphere(s, dvv, Dinv, grads);
  for (c = 0; c < NP * NP * 2; c++) old[c] = grads[c];
  for (j = 0; j < NP; j++)
    for (i = 0; i < NP; i++) {
      A3(grads, i, j, 0) = A3(old, i, j, 0) * A4(tv, i, j, 0, 0) + A3(old, i, j, 1) * A4(tv, i, j, 0, 1);
      A3(grads, i, j, 1) = A3(old, i, j, 0) * A4(tv, i, j, 1, 0) + A3(old, i, j, 1) * A4(tv, i, j, 1, 1);
    }
  divergence_sphere_wk(grads, dvv, Dinv, spheremp, out);
}

/* :186-200.  nthreads > 1: OpenMP over elements (this build's addition; the reference is
 * serial); same arithmetic per slab. */
int bwk_oracle_biharmonic(int64_t nelemd, int nlev, int qsize, const double *dvv, const double *elem,
                          double *qtens, int nthreads) {
  int64_t ie;
  if (nelemd < 1 || nlev < 1 || qsize < 1) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 1 ? nthreads : 1)
#endif
  for (ie = 0; ie < nelemd; ie++) {
    int q, k, c;
    for (q = 0; q < qsize; q++)
      for (k = 0; k < nlev; k++) {
        double *s = qtens + 16 * ((int64_t)k + (int64_t)nlev * (q + (int64_t)qsize * ie));
        double out[16];
        laplace_sphere_wk(s, dvv, elem + ELEM_DOUBLES * ie, out);
        for (c = 0; c < 16; c++) s[c] = out[c];
      }
  }
  return 0;
}

/* :77-91 myrandom + :48-58 initialize_data: the reference's own inputs.  (1301*old+97 < 2^31
 * for old < 2^17, so the reference's default-integer arithmetic does not overflow.) */
static int lcg_state = 11;
static void myrandom(int64_t n, double *a, int reset) {
  int64_t i;
  if (reset) lcg_state = 11;
  for (i = 0; i < n; i++) {
    lcg_state = (1301 * lcg_state + 97) % (1024 * 128);
    a[i] = lcg_state / (double)(1024 * 128);
  }
}
void bwk_oracle_init(int64_t nelemd, int nlev, int qsize, double *dvv, double *elem, double *qtens) {
  int64_t ie;
  myrandom(16, dvv, 1);
  for (ie = 0; ie < nelemd; ie++) {
    myrandom(64, elem + ELEM_DOUBLES * ie, 0);
    myrandom(16, elem + ELEM_DOUBLES * ie + 64, 0);
    myrandom(64, elem + ELEM_DOUBLES * ie + 80, 0);
  }
  myrandom((int64_t)16 * nlev * qsize * nelemd, qtens, 0);
}
