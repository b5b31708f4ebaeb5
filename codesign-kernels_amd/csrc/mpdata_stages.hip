// mpdata_stages.hip -- stage-by-stage DEBUG mode of the routine.
//
// The fused kernels keep every intermediate of the reference routine
//   mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642
// in registers.  When a parity test fails that is no help in finding WHERE.  This file runs
// the same arithmetic as eight small, unfused kernels -- one per stage of the reference
// (the split its OpenACC version makes, :112-235) -- that materialise the reference's
// temporaries uuu, www, mx, mn (:485-491) in device memory, and can stop after any stage.
// tests/test_stages.py compares every array after every stage with the oracle's
// (bit-for-bit: this file is compiled with -ffp-contract=off and keeps the reference's
// expression order), and the final state with the fused kernels.  One thread per
// (instance, column, level); nothing here is tuned -- it is not a fast path.
#include <hip/hip_runtime.h>

#include "mpdata_hip.h"

namespace {

struct Dims {
  long long n;  // ncrms
  int nx, nz, nzm;
};
struct Arr {
  double* f; const double* u; const double* w; const double* rho; const double* rhow; const double* adz;
  double* flux; double* uuu; double* www; double* mx; double* mn;
};

// Fortran-style element access (1-based k, signed i), instance index s 0-based
#define F_(s, i, k) a.f[(s) + d.n * ((long long)((i) + 2) + (long long)(d.nx + 6) * ((k) - 1))]
#define U_(s, i, k) a.u[(s) + d.n * ((long long)((i) + 1) + (long long)(d.nx + 5) * ((k) - 1))]
#define W_(s, i, k) a.w[(s) + d.n * ((long long)((i) + 1) + (long long)(d.nx + 4) * ((k) - 1))]
#define UUU_(s, i, k) a.uuu[(s) + d.n * ((long long)((i) + 1) + (long long)(d.nx + 5) * ((k) - 1))]
#define WWW_(s, i, k) a.www[(s) + d.n * ((long long)((i) + 1) + (long long)(d.nx + 4) * ((k) - 1))]
#define MX_(s, i, k) a.mx[(s) + d.n * ((long long)(i) + (long long)(d.nx + 2) * ((k) - 1))]
#define MN_(s, i, k) a.mn[(s) + d.n * ((long long)(i) + (long long)(d.nx + 2) * ((k) - 1))]
#define K2_(p, s, k) p[(s) + d.n * (long long)((k) - 1)]

__device__ inline double vmax(double x, double y) { return x > y ? x : y; }
__device__ inline double vmin(double x, double y) { return x < y ? x : y; }
__device__ inline double andiff(double x1, double x2, double c, double b) {
  return (fabs(c) - c * c * b) * 0.5 * (x2 - x1);  // :502
}
__device__ inline double across(double x1, double c1, double c2) { return 0.03125 * c1 * c2 * x1; }  // :503
__device__ inline double pos(double y) { return vmax(0.0, y); }
__device__ inline double neg(double y) { return -vmin(0.0, y); }

// thread -> (s, i, k): blockIdx.y walks the columns i0..i1, blockIdx.z the levels k0..k1
#define CELL(i0, k0)                                                              \
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;          \
  const int i = (i0) + (int)blockIdx.y;                                           \
  const int k = (k0) + (int)blockIdx.z;                                           \
  if (s >= d.n) return;

// ---- stage 1 (:511, :513-526): www(.,.,nz) = 0; extrema of the incoming field ---------
__global__ void st1_extrema0(Dims d, Arr a) {
  CELL(0, 1)
  const int kc = min(k + 1, d.nzm), kb = max(k - 1, 1), ib = i - 1, ic = i + 1;
  MX_(s, i, k) = vmax(vmax(vmax(vmax(F_(s, ib, k), F_(s, ic, k)), F_(s, i, kb)), F_(s, i, kc)), F_(s, i, k));
  MN_(s, i, k) = vmin(vmin(vmin(vmin(F_(s, ib, k), F_(s, ic, k)), F_(s, i, kb)), F_(s, i, kc)), F_(s, i, k));
}
__global__ void st1_wtop(Dims d, Arr a) {
  CELL(-1, d.nz)
  WWW_(s, i, k) = 0.0;
}
// ---- stage 2 (:528-548): upwind fluxes and their horizontal sum ---------------------------
__global__ void st2_uflux(Dims d, Arr a) {
  CELL(-1, 1)
  UUU_(s, i, k) = vmax(0.0, U_(s, i, k)) * F_(s, i - 1, k) + vmin(0.0, U_(s, i, k)) * F_(s, i, k);
}
__global__ void st2_wflux(Dims d, Arr a) {
  CELL(-1, 1)
  const int kb = max(k - 1, 1);
  WWW_(s, i, k) = vmax(0.0, W_(s, i, k)) * F_(s, i, kb) + vmin(0.0, W_(s, i, k)) * F_(s, i, k);
}
__global__ void st2_fluxsum(Dims d, Arr a) {  // one thread per (s, k): the reference's i order
  CELL(0, 1)
  double acc = 0.0;
  for (int ii = 1; ii <= d.nx; ++ii) acc = acc + WWW_(s, ii, k);
  K2_(a.flux, s, k) = acc;
}
// ---- stage 3 (:550-560): first-pass update, halo columns -1..nx+2 included ----------------
__global__ void st3_update1(Dims d, Arr a) {
  CELL(-1, 1)
  const double irho = 1.0 / K2_(a.rho, s, k), iadz = 1.0 / K2_(a.adz, s, k);
  F_(s, i, k) = F_(s, i, k) - (UUU_(s, i + 1, k) - UUU_(s, i, k) + (WWW_(s, i, k + 1) - WWW_(s, i, k)) * iadz) * irho;
}
// ---- stage 4 (:561-586): antidiffusive fluxes (overwrite uuu, www); www(.,.,1) = 0 ---------
__global__ void st4_uanti(Dims d, Arr a) {
  CELL(0, 1)
  const int kc = min(k + 1, d.nzm), kb = max(k - 1, 1), ib = i - 1;
  const double irho = 1.0 / K2_(a.rho, s, k);
  const double dd = (double)(2.0f / (float)(kc - kb)) / K2_(a.adz, s, k);  // :569
  UUU_(s, i, k) = andiff(F_(s, ib, k), F_(s, i, k), U_(s, i, k), irho) -
                  across(dd * (F_(s, ib, kc) + F_(s, i, kc) - F_(s, ib, kb) - F_(s, i, kb)), U_(s, i, k),
                         W_(s, ib, k) + W_(s, ib, kc) + W_(s, i, k) + W_(s, i, kc)) * irho;
}
__global__ void st4_wanti(Dims d, Arr a) {
  CELL(0, 1)
  const int kb = max(k - 1, 1), ib = i - 1, ic = i + 1;
  const double irho = 1.0 / K2_(a.rho, s, k);
  const double irhow = 1.0 / (K2_(a.rhow, s, k) * K2_(a.adz, s, k));
  const double v = andiff(F_(s, i, kb), F_(s, i, k), W_(s, i, k), irhow) -
                   across(F_(s, ic, kb) + F_(s, ic, k) - F_(s, ib, kb) - F_(s, ib, k), W_(s, i, k),
                          U_(s, i, kb) + U_(s, i, k) + U_(s, ic, k) + U_(s, ic, kb)) * irho;
  WWW_(s, i, k) = v;
}
__global__ void st4_wbottom(Dims d, Arr a) {
  CELL(-1, 1)
  WWW_(s, i, k) = 0.0;
}
// ---- stage 5 (:588-600): extrema of the first-pass field, merged into mx, mn ----------------
__global__ void st5_extrema1(Dims d, Arr a) {
  CELL(0, 1)
  const int kc = min(k + 1, d.nzm), kb = max(k - 1, 1), ib = i - 1, ic = i + 1;
  MX_(s, i, k) = vmax(vmax(vmax(vmax(vmax(F_(s, ib, k), F_(s, ic, k)), F_(s, i, kb)), F_(s, i, kc)), F_(s, i, k)), MX_(s, i, k));
  MN_(s, i, k) = vmin(vmin(vmin(vmin(vmin(F_(s, ib, k), F_(s, ic, k)), F_(s, i, kb)), F_(s, i, kc)), F_(s, i, k)), MN_(s, i, k));
}
// ---- stage 6 (:601-612): limiter ratios --------------------------------------------------
__global__ void st6_ratios(Dims d, Arr a) {
  CELL(0, 1)
  const int kc = min(k + 1, d.nzm), ic = i + 1;
  const double eps = (double)1.e-10f;  // :509
  const double rho = K2_(a.rho, s, k), iadz = 1.0 / K2_(a.adz, s, k);
  MX_(s, i, k) = rho * (MX_(s, i, k) - F_(s, i, k)) /
                 (neg(UUU_(s, ic, k)) + pos(UUU_(s, i, k)) + iadz * (neg(WWW_(s, i, kc)) + pos(WWW_(s, i, k))) + eps);
  MN_(s, i, k) = rho * (F_(s, i, k) - MN_(s, i, k)) /
                 (pos(UUU_(s, ic, k)) + neg(UUU_(s, i, k)) + iadz * (pos(WWW_(s, i, kc)) + neg(WWW_(s, i, k))) + eps);
}
// ---- stage 7 (:613-627): limited fluxes; flux gets the limited vertical flux added ----------
__global__ void st7_ulim(Dims d, Arr a) {
  CELL(1, 1)
  const int ib = i - 1;
  UUU_(s, i, k) = pos(UUU_(s, i, k)) * vmin(vmin(1.0, MX_(s, i, k)), MN_(s, ib, k)) -
                  neg(UUU_(s, i, k)) * vmin(vmin(1.0, MX_(s, ib, k)), MN_(s, i, k));
}
__global__ void st7_wlim(Dims d, Arr a) {
  CELL(1, 1)
  const int kb = max(k - 1, 1);
  WWW_(s, i, k) = pos(WWW_(s, i, k)) * vmin(vmin(1.0, MX_(s, i, k)), MN_(s, i, kb)) -
                  neg(WWW_(s, i, k)) * vmin(vmin(1.0, MX_(s, i, kb)), MN_(s, i, k));
}
__global__ void st7_fluxadd(Dims d, Arr a) {
  CELL(0, 1)
  double acc = K2_(a.flux, s, k);
  for (int ii = 1; ii <= d.nx; ++ii) acc = acc + WWW_(s, ii, k);
  K2_(a.flux, s, k) = acc;
}
// ---- stage 8 (:630-637): final positive-definite update of the interior ---------------------
__global__ void st8_update2(Dims d, Arr a) {
  CELL(1, 1)
  const double irho = 1.0 / K2_(a.rho, s, k), iadz = 1.0 / K2_(a.adz, s, k);
  F_(s, i, k) = vmax(0.0, F_(s, i, k) - (UUU_(s, i + 1, k) - UUU_(s, i, k) + (WWW_(s, i, k + 1) - WWW_(s, i, k)) * iadz) * irho);
}

}  // namespace

extern "C" int mpdata_debug_stages_device(int64_t ncrms, int nx, int nz, int last_stage, double* f,
                                          const double* u, const double* w, const double* rho,
                                          const double* rhow, const double* adz, double* flux,
                                          double* uuu, double* www, double* mx, double* mn, void* stream) {
  if (ncrms < 1 || nx < 1 || nz < 3 || last_stage < 1 || last_stage > 8) return MPDATA_EINVAL;
  if (!f || !u || !w || !rho || !rhow || !adz || !flux || !uuu || !www || !mx || !mn) return MPDATA_EINVAL;
  if (nx + 6 > 65535 || nz > 65535) return MPDATA_EUNSUPPORTED;
  const Dims d{(long long)ncrms, nx, nz, nz - 1};
  const Arr a{f, u, w, rho, rhow, adz, flux, uuu, www, mx, mn};
  hipStream_t st = (hipStream_t)stream;
  const unsigned bx = 64, gx = (unsigned)((ncrms + bx - 1) / bx);
  const int nzm = nz - 1;
  // L(kernel, number of columns, number of levels): the column / level origin is in the kernel
#define L(kern, ncols, nlev) hipLaunchKernelGGL(kern, dim3(gx, (unsigned)(ncols), (unsigned)(nlev)), dim3(bx), 0, st, d, a)
  // :511 zeroes www(.,.,nz) before anything else; stage 2 writes levels 1..nzm only
  L(st1_wtop, nx + 4, 1);
  L(st1_extrema0, nx + 2, nzm);                       // i = 0..nx+1
  if (last_stage >= 2) {
    L(st2_uflux, nx + 5, nzm);                        // i = -1..nx+3
    L(st2_wflux, nx + 4, nzm);                        // i = -1..nx+2
    L(st2_fluxsum, 1, nzm);
  }
  if (last_stage >= 3) L(st3_update1, nx + 4, nzm);   // i = -1..nx+2
  if (last_stage >= 4) {
    L(st4_uanti, nx + 3, nzm);                        // i = 0..nx+2
    // www(i,k) of the antidiffusive pass reads f and u,w only, never www: in place is safe.
    // Levels 2..nzm get the value, level 1 is zeroed afterwards (:586) as in the reference.
    L(st4_wanti, nx + 2, nzm);                        // i = 0..nx+1
    L(st4_wbottom, nx + 4, 1);                        // i = -1..nx+2, k = 1
  }
  if (last_stage >= 5) L(st5_extrema1, nx + 2, nzm);
  if (last_stage >= 6) L(st6_ratios, nx + 2, nzm);
  if (last_stage >= 7) {
    L(st7_ulim, nx + 1, nzm);                         // i = 1..nx+1 (reads mx/mn and its own uuu only)
    L(st7_wlim, nx, nzm);                             // i = 1..nx
    L(st7_fluxadd, 1, nzm);
  }
  if (last_stage >= 8) L(st8_update2, nx, nzm);
#undef L
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
