/*
 * nlk_hip.h -- C-ABI of libnlk_hip.so: the MI355X (gfx950) replacement for the compute body
 * of the reference's third mini-app, the MPAS-Ocean high-order tracer flux loop nest
 * (SURVEY.md section 8f-4).
 *
 * What it replaces in the reference (E3SM-Project/codesign-kernels):
 *   nested_loops/nested.F90
 *     :495-559  run_original_cpu_directive (OpenACC / OpenMP-offload "original" form)
 *     :123-157  the same loop nest as the program's CPU reference (refFlx)
 *   (the other forms in the program -- "GPU optimized", k-tiled, YAKL, CKE/Kokkos -- compute
 *   the same highOrderFlx; the program checks them against refFlx to errTol = 1e-10)
 *
 * Array contract = the reference's (nested_vars.F90:111-127): Fortran column-major, level index
 * fastest, leading dimension nvldim >= nVertLevels, 1-based cell indices, 32-bit integers:
 *   tracerCur(nvldim,nCells)                                          in
 *   normalThicknessFlux, advMaskHighOrder(nvldim,nEdges)              in
 *   advCellsForEdge, advCoefs, advCoefs3rd(nAdv,nEdges)               in
 *   nAdvCellsForEdge(nEdges)   minLevelCell, maxLevelCell(nCells)     in
 *   highOrderFlx(nvldim,nEdges)   out: levels 1..nVertLevels written, the padding rows left
 * For every edge and level:  highOrderFlx = sum over i = 1..nAdvCellsForEdge, for the cells whose
 * [minLevelCell, maxLevelCell] contains the level, of
 *   tracerCur(k,cell) * (normalThicknessFlux*advMaskHighOrder) * (advCoefs + advCoefs3rd*coef3rdOrder*sign)
 * in the reference's order.  A cell index outside 1..nCells contributes nothing (the reference
 * would read out of bounds).
 *
 * Functions return 0, a negative NLK_E* code, or a positive hipError_t; nlk_last_error() has
 * text.  No CPU fallback.
 */
#ifndef NLK_HIP_H
#define NLK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NLK_EINVAL (-1)
#define NLK_VARIANT_EXACT 0 /* no FMA contraction: bit-identical to the reference loop */
#define NLK_VARIANT_FAST 1  /* FMA contraction allowed */

/* device pointers, asynchronous on `stream` (hipStream_t as void*; NULL = default stream) */
int nlk_high_order_flux_device(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv,
                               const int* nAdvCellsForEdge, const int* advCellsForEdge,
                               const int* minLevelCell, const int* maxLevelCell,
                               const double* tracerCur, const double* normalThicknessFlux,
                               const double* advMaskHighOrder, const double* advCoefs,
                               const double* advCoefs3rd, double coef3rdOrder, double* highOrderFlx,
                               void* stream);
/* host arrays, synchronous, transfers included */
int nlk_high_order_flux(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv,
                        const int* nAdvCellsForEdge, const int* advCellsForEdge,
                        const int* minLevelCell, const int* maxLevelCell, const double* tracerCur,
                        const double* normalThicknessFlux, const double* advMaskHighOrder,
                        const double* advCoefs, const double* advCoefs3rd, double coef3rdOrder,
                        double* highOrderFlx);
int nlk_set_variant(int variant); /* returns the previous one; default: NLK_VARIANT env or exact */
int nlk_get_variant(void);
/* kernel form: -1 automatic (default; NLK_KERNEL=0 / 1 in the environment presets it), 0 one level per
 * lane (8-byte accesses), 1 two levels per lane (16-byte accesses: half the vector-memory instructions;
 * needs an even nvldim and 16-byte aligned arrays, else form 0 is taken).  Same results.  Returns the
 * previous setting. */
int nlk_set_kernel(int mode);
/* minimal HBM traffic of one call: every input array read once, highOrderFlx written once
 * (the gather re-reads of tracerCur hit in cache: 10 x 100 x 8 B per edge from 2.2 MB) */
int64_t nlk_algorithmic_bytes(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv);
const char* nlk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* NLK_HIP_H */
