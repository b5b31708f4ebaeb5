/*
 * oracle/nlk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64) of the third mini-app of the reference, the MPAS-Ocean
 * high-order tracer flux gather loop ("after that"):
 *   nested_loops/nested.F90:123-157  the program's own CPU reference loop (-> refFlx)
 *                          :495-559  run_original_cpu_directive (the same arithmetic)
 *   nested_loops/nested_vars.F90:34-36  coef3rdOrder
 * It is the parity checker of libnlk_hip.so; only tests/, smoke() and bench.py's cpu_baseline
 * leg may load it.
 *
 * Parity status: PINNED.  tests/test_nlk.py checks it bit-for-bit against refFlx of the
 * reference program itself (oracle/build_ref.py --nlk, amdflang -O3 -ffp-contract=off,
 * -DNO_MPI), fixture tests/golden/nlk_ref_small.npz.  Build with -ffp-contract=off.
 *
 * Arrays = the reference's (Fortran column-major, 1-based cell indices, level index fastest,
 * leading dimension nvldim >= nVertLevels, nested_vars.F90:111-127):
 *   tracerCur(nvldim,nCells)  normalThicknessFlux, advMaskHighOrder, highOrderFlx(nvldim,nEdges)
 *   advCellsForEdge, advCoefs, advCoefs3rd(nAdv,nEdges)  nAdvCellsForEdge(nEdges)
 *   minLevelCell, maxLevelCell(nCells)
 */
#include <stdint.h>
#include <math.h>

/* nested_vars.F90:35 -- `coef3rdOrder = 2.14` has no kind suffix: an fp32 constant widened */
double nlk_oracle_coef3rd(void) { return (double)2.14f; }

int nlk_oracle_high_order_flux(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv,
                               const int *nAdvCellsForEdge, const int *advCellsForEdge,
                               const int *minLevelCell, const int *maxLevelCell,
                               const double *tracerCur, const double *normalThicknessFlux,
                               const double *advMaskHighOrder, const double *advCoefs,
                               const double *advCoefs3rd, double coef3rdOrder, double *highOrderFlx) {
  int iEdge, i, k;
  if (nEdges < 1 || nCells < 1 || nVertLevels < 1 || nvldim < nVertLevels || nAdv < 1) return -1;
  for (iEdge = 0; iEdge < nEdges; iEdge++) {
    const double *ntf = normalThicknessFlux + (int64_t)nvldim * iEdge;
    const double *msk = advMaskHighOrder + (int64_t)nvldim * iEdge;
    double *out = highOrderFlx + (int64_t)nvldim * iEdge;
    /* :125-131 common factors; :136-148 gather.  Per level the sum runs over i ascending,
     * so the loop nest can be k-outer here without changing any rounding. */
    for (k = 1; k <= nVertLevels; k++) {
      const double wgt = ntf[k - 1] * msk[k - 1];
      const double sgn = copysign(1.0, ntf[k - 1]);   /* sign(1.0_RKIND, x) */
      double acc = 0.0;
      for (i = 0; i < nAdvCellsForEdge[iEdge]; i++) {
        const int iCell = advCellsForEdge[i + (int64_t)nAdv * iEdge];
        if (iCell < 1 || iCell > nCells) return -2;
        if (k >= minLevelCell[iCell - 1] && k <= maxLevelCell[iCell - 1]) {
          const double coef1 = advCoefs[i + (int64_t)nAdv * iEdge];
          const double coef3 = advCoefs3rd[i + (int64_t)nAdv * iEdge] * coef3rdOrder;
          acc = acc + tracerCur[(k - 1) + (int64_t)nvldim * (iCell - 1)] * wgt * (coef1 + coef3 * sgn);
        }
      }
      out[k - 1] = acc;
    }
  }
  return 0;
}
