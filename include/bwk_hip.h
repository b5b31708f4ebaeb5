/*
 * bwk_hip.h -- C-ABI of libbwk_hip.so: the MI355X (gfx950) replacement for the compute body
 * of the reference's second mini-app, the HOMME spectral-element kernel
 * `biharmonic_wk_scalar`.
 *
 * What it replaces in the reference (E3SM-Project/codesign-kernels):
 *   atmosphere/biharmonic_wk_kernel.F90
 *     :186-200  biharmonic_wk_scalar (CPU: laplace_sphere_wk per (level, tracer, element))
 *     :317-360  ... OpenACC "compiler inline" variant      :525-534  ... "push loop" variant
 *   and atmosphere/Makefile's `pgiacc` target.
 *
 * Array contract = the reference's (Fortran column-major, fp64; np = 4):
 *   qtens(np,np,nlev,qsize,nelemd)  inout: every 4x4 slab s is replaced by
 *                                   laplace_sphere_wk(s) = div_wk( tensorVisc . grad(s) ) (:164-182)
 *   dvv(np,np)                      deriv%Dvv (:19-21)
 *   elem(144,nelemd)                per element, in the declaration order of type element_t
 *                                   (:23-27): Dinv(np,np,2,2) | spheremp(np,np) | tensorVisc(np,np,2,2)
 * (a Fortran caller passes `elem` itself when element_t is a SEQUENCE / bind(C) type, else a
 * packed copy).
 *
 * Functions return 0, a negative BWK_E* code, or a positive hipError_t; bwk_last_error() has
 * text.  No CPU fallback: without a usable HIP device the calls fail.
 */
#ifndef BWK_HIP_H
#define BWK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BWK_EINVAL (-1)
#define BWK_EUNSUPPORTED (-2)
#define BWK_VARIANT_EXACT 0 /* no FMA contraction, reference summation order: bit-identical */
#define BWK_VARIANT_FAST 1  /* FMA contraction allowed: rounding differences only */

/* host arrays, synchronous, transfers included (the OpenACC variants' update device/host,
 * reference :572-576) */
int bwk_biharmonic_wk_scalar(int64_t nelemd, int nlev, int qsize, double* qtens, const double* dvv,
                             const double* elem);
/* device pointers, asynchronous on `stream` (hipStream_t as void*; NULL = default stream) */
int bwk_biharmonic_wk_scalar_device(int64_t nelemd, int nlev, int qsize, double* qtens,
                                    const double* dvv, const double* elem, void* stream);
int bwk_set_variant(int variant); /* returns the previous one; default: BWK_VARIANT env or exact */
int bwk_get_variant(void);
/* minimal HBM traffic of one call: qtens read + written once, dvv and elem read once */
int64_t bwk_algorithmic_bytes(int64_t nelemd, int nlev, int qsize);
const char* bwk_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* BWK_HIP_H */
