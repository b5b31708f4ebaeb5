// mpdata_kernel_body.h -- the fused MPDATA advection kernel for gfx950 (CDNA4).
//
// Included twice (mpdata_kernels_exact.hip with -ffp-contract=off,
// mpdata_kernels_fast.hip with -ffp-contract=fast); MPDATA_NS names the
// namespace of the instance.
//
// What it computes: one call of the reference routine
//   mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642
//   (advect_scalar2D_cpu; the OpenACC variants :72-244 / :247-474 compute the
//   same thing in 8 / 17 kernels with HBM-resident temporaries mx, mn, uuu,
//   www, irho, iadz, irhow -- :485-491)
// as ONE kernel that reads f,u,w,rho,rhow,adz once and writes f,flux once.
// None of the reference's temporaries is ever materialised in HBM.
//
// Decomposition:
//   * the CRM-instance axis `sl` (contiguous in memory) is the lane axis:
//     SLW = 64/SPW consecutive sl per wave, so every row access of a wave is
//     SPW contiguous segments of SLW*8 bytes;
//   * the x axis is cut into NS = SPW*NWV strips of W columns; a thread owns
//     one (sl, strip) and keeps the whole vertical rolling window of its W
//     columns in REGISTERS;
//   * the kernel marches k = 1..nzm as a 4-stage software pipeline
//         A: upwind fluxes of level p, first-pass field f1 of level p-1
//         B: antidiffusive vertical flux  W2 of level p-1
//         C: antidiffusive horizontal flux U2 + limiter ratios of level p-2
//         D: limited fluxes of level p-2, final field of level p-3
//     (dependency radius 3 in k);
//   * strips exchange only their edge columns of f1 and of the limiter
//     ratios through LDS, twice per level; with NWV==1 the exchange is
//     wave-local and needs no s_barrier;
//   * `flux` is summed in the reference's order (i ascending, upwind terms
//     then limited antidiffusive terms, :541-547 and :624) by one strip's
//     lanes reading the per-column values from LDS.
//
// Arithmetic follows the reference expression by expression (same operand
// order); with -ffp-contract=off the result is bit-identical to the
// reference built without FMA contraction.  The k clamps kb = max(1,k-1),
// kc = min(nzm,k+1) (:515-516 etc.) are realised as "ghost levels": level 0
// is a copy of level 1 and level nzm+1 a copy of level nzm in every rolling
// register set, so the loop body is branch-free per lane; all remaining
// conditions depend on the step counter only (wave-uniform scalar branches).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mpdata_args.h"

namespace MPDATA_NS {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define MPD_UNROLL _Pragma("unroll")

__device__ __forceinline__ double ld_b64(__amdgpu_buffer_rsrc_t r, int voff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
}
__device__ __forceinline__ void st_b64(__amdgpu_buffer_rsrc_t r, int voff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, 0, 0);
}
// Buffer descriptor over [p, p+bytes): loads outside it return 0 and stores
// are dropped by the hardware range check.  num_records is kept below 2^31
// so that a "negative" 32-bit offset is always out of range.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  const long long lim = 0x7FFFFFFFll;
  const int n = (int)(bytes < 0 ? 0 : (bytes > lim ? lim : bytes));
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, n, 0x00020000);
}

__device__ __forceinline__ double dmax(double x, double y) { return __builtin_fmax(x, y); }
__device__ __forceinline__ double dmin(double x, double y) { return __builtin_fmin(x, y); }
// Statement functions of the reference (:500-503), left-to-right.
__device__ __forceinline__ double andiff(double x1, double x2, double a, double b) {
  return (__builtin_fabs(a) - a * a * b) * 0.5 * (x2 - x1);
}
__device__ __forceinline__ double across(double x1, double a1, double a2) {
  return 0.03125 * a1 * a2 * x1;
}
__device__ __forceinline__ double pp(double y) { return dmax(0.0, y); }
__device__ __forceinline__ double pn(double y) { return -dmin(0.0, y); }

template <int NWV>
__device__ __forceinline__ void strip_sync() {
  if constexpr (NWV == 1) {
    // all strips live in this wave: LDS operations of one wave execute in
    // order, only the compiler must not reorder across the exchange point
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <int W, int SPW, int NWV>
struct Tile {
  static constexpr int SLW = 64 / SPW;       // CRM instances per workgroup
  static constexpr int NS = SPW * NWV;       // strips per workgroup
  static constexpr int NCOL = NS * W;        // columns covered: i = -1 .. NCOL-2
  static constexpr int THREADS = 64 * NWV;
  // LDS doubles: e1[2][NS][SLW], e2[4][NS][SLW], fl1[NCOL][SLW], fl3[NCOL][SLW], ps[4][SLW]
  static constexpr int LDS_DOUBLES = (6 * NS + 2 * NCOL + 4) * SLW;
};

template <int W, int SPW, int NWV>
__global__ void __launch_bounds__(64 * NWV)
mpdata_advect_kernel(const MpdataArgs a) {
  using T = Tile<W, SPW, NWV>;
  constexpr int SLW = T::SLW, NS = T::NS, NCOL = T::NCOL;
  constexpr int H = W + 2;  // arrays with one halo column each side: index h = c+1, c = -1..W

  __shared__ double lds[T::LDS_DOUBLES];
  double* const e1 = lds;                    // f1 edge columns            [2][NS][SLW]
  double* const e2 = e1 + 2 * NS * SLW;      // limiter-ratio edge columns [4][NS][SLW]
  double* const fl1 = e2 + 4 * NS * SLW;     // upwind vertical flux per column (level p)
  double* const fl3 = fl1 + NCOL * SLW;      // limited vertical flux per column
  double* const ps = fl3 + NCOL * SLW;       // partial flux sums, ring of 4 levels

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sll = lane % SLW;
  const int strip = wave * SPW + lane / SLW;
  const int nx = a.nx, nzm = a.nz - 1;
  const long long ncrms = a.ncrms;

  long long sl = (long long)blockIdx.x * SLW + sll;
  const bool sl_ok = sl < ncrms;
  if (!sl_ok) sl = ncrms - 1;  // ragged last tile: compute a duplicate, store nothing
  const int tr = blockIdx.y;   // tracer

  const int i0 = -1 + strip * W;    // first owned column
  const int cs = (int)(ncrms * 8);  // byte stride between columns
  // byte offsets inside one k-plane of the thread's column c = -1 (f), c = 0 (u), c = -1 (w)
  const int vf0 = (int)((sl + ncrms * (long long)(i0 + 1)) * 8);
  const int vu0 = vf0;
  const int vw0 = (int)((sl + ncrms * (long long)i0) * 8);
  const int vk = (int)(sl * 8);  // offset inside one (ncrms) row of rho/rhow/adz/flux

  const long long f_plane = ncrms * (long long)(nx + 6) * 8;
  const long long u_plane = ncrms * (long long)(nx + 5) * 8;
  const long long w_plane = ncrms * (long long)(nx + 4) * 8;
  const long long k_plane = ncrms * 8;
  const char* const fb = (const char*)(a.f + (long long)tr * a.f_tstride);
  const char* const ub = (const char*)a.u;
  const char* const wb = (const char*)a.w;
  const char* const rb = (const char*)a.rho;
  const char* const rwb = (const char*)a.rhow;
  const char* const ab = (const char*)a.adz;
  const char* const xb = (const char*)(a.flux + (long long)tr * a.flux_tstride);
  const long long f_bytes = f_plane * nzm, u_bytes = u_plane * nzm, w_bytes = w_plane * (nzm + 1);

  const double eps = (double)1.e-10f;  // reference :509, fp32 literal

  // LDS slots
  const int sL = strip > 0 ? strip - 1 : 0;
  const int sR = strip < NS - 1 ? strip + 1 : NS - 1;
  const int e_own = strip * SLW + sll;
  const int e_left = sL * SLW + sll;
  const int e_right = sR * SLW + sll;
  const int g0 = strip * W;  // global column slot of c = 0  (column i = g - 1)

  // ---- rolling state (level names are relative to the step counter p) -----
  double Fc[H], Uc[W + 1], Wc[H];          // plane p:   f0, u, w
  double F0b[W];                            // f0(p-1)
  double dU1b[W], W1b[W];                   // upwind U1(i+1)-U1(i) and W1 at level p-1
  double MX0p[W], MN0p[W];                  // pass-0 extrema of level p-1, f0(p) not yet folded in
  double MX0f[W], MN0f[W];                  // pass-0 extrema of level p-2, complete
  double F1b[H], F1a[H];                    // f1(p-2), f1(p-3)
  double Ub[W + 1], Ua[W + 1];              // u(p-1), u(p-2)
  double Wb[H], Wa[H];                      // w(p-1), w(p-2)
  double W2b[W];                            // antidiffusive W2(p-2)
  double MXa[W], MNa[W];                    // limiter ratios of level p-3
  double dU3a[W], W3a[W];                   // limited U3(i+1)-U3(i) and W3 at level p-3
  double rho_c, adz_c, rhow_c;              // level p
  double irho1 = 0, irho2 = 0, irho3 = 0, iadz1 = 0, iadz2 = 0, iadz3 = 0, irhow1 = 0, rho1 = 0,
         rho2 = 0;

  MPD_UNROLL for (int c = 0; c < W; ++c) {
    dU1b[c] = W1b[c] = MX0p[c] = MN0p[c] = MX0f[c] = MN0f[c] = 0.0;
    W2b[c] = MXa[c] = MNa[c] = dU3a[c] = W3a[c] = 0.0;
  }
  MPD_UNROLL for (int h = 0; h < H; ++h) F1b[h] = F1a[h] = Wb[h] = Wa[h] = 0.0;
  MPD_UNROLL for (int c = 0; c <= W; ++c) Ub[c] = Ua[c] = 0.0;

  // ---- plane 1 ------------------------------------------------------------
  {
    const __amdgpu_buffer_rsrc_t rf = make_rsrc(fb, f_bytes);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(ub, u_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(wb, w_bytes);
    MPD_UNROLL for (int h = 0; h < H; ++h) Fc[h] = ld_b64(rf, vf0 + h * cs);
    MPD_UNROLL for (int c = 0; c <= W; ++c) Uc[c] = ld_b64(ru, vu0 + c * cs);
    MPD_UNROLL for (int h = 0; h < H; ++h) Wc[h] = ld_b64(rw, vw0 + h * cs);
    rho_c = ld_b64(make_rsrc(rb, k_plane * nzm), vk);
    adz_c = ld_b64(make_rsrc(ab, k_plane * nzm), vk);
    rhow_c = ld_b64(make_rsrc(rwb, k_plane * (nzm + 1)), vk);
  }
  // ghost level 0 == level 1 (kb = max(1,k-1), reference :516, :529)
  MPD_UNROLL for (int c = 0; c < W; ++c) F0b[c] = Fc[c + 1];

  const int nsteps = nzm + 3;
  for (int p = 1; p <= nsteps; ++p) {
    const bool hasA = p <= nzm;                // plane p exists
    const bool hasB = p >= 2 && p - 1 <= nzm;  // level kB = p-1 exists
    const int kB = p - 1, kC = p - 2, kD = p - 3;
    const bool hasC = kC >= 1 && kC <= nzm;
    const bool hasD = kD >= 1 && kD <= nzm;

    // ================= stage A: level p upwind, level p-1 first pass =========
    double W1c[W], F1c[H], MX0n[W], MN0n[W];
    double irho0 = 0, iadz0 = 0, irhow0 = 0;
    if (hasA) {
      irho0 = 1.0 / rho_c;              // :552
      iadz0 = 1.0 / adz_c;              // :553
      irhow0 = 1.0 / (rhow_c * adz_c);  // :565
      double U1c[W + 1];
      MPD_UNROLL for (int c = 0; c <= W; ++c)  // :532
        U1c[c] = dmax(0.0, Uc[c]) * Fc[c] + dmin(0.0, Uc[c]) * Fc[c + 1];
      MPD_UNROLL for (int c = 0; c < W; ++c)  // :537
        W1c[c] = dmax(0.0, Wc[c + 1]) * F0b[c] + dmin(0.0, Wc[c + 1]) * Fc[c + 1];
      MPD_UNROLL for (int c = 0; c < W; ++c) {
        // first pass of level p-1 (:557) and its completed pass-0 extrema (:521-522)
        F1c[c + 1] = F0b[c] - (dU1b[c] + (W1c[c] - W1b[c]) * iadz1) * irho1;
        MX0n[c] = dmax(MX0p[c], Fc[c + 1]);
        MN0n[c] = dmin(MN0p[c], Fc[c + 1]);
        // pass-0 extrema of level p without the f(i,kc) term (:521-522)
        MX0p[c] = dmax(dmax(dmax(Fc[c], Fc[c + 2]), F0b[c]), Fc[c + 1]);
        MN0p[c] = dmin(dmin(dmin(Fc[c], Fc[c + 2]), F0b[c]), Fc[c + 1]);
        dU1b[c] = U1c[c + 1] - U1c[c];
        W1b[c] = W1c[c];
        F0b[c] = Fc[c + 1];
        fl1[(g0 + c) * SLW + sll] = W1c[c];
      }
    } else {
      // p == nz: www(:,:,:,nz) = 0 (:511), kc clamps to nzm (:515)
      MPD_UNROLL for (int c = 0; c < W; ++c) {
        W1c[c] = 0.0;
        F1c[c + 1] = F0b[c] - (dU1b[c] + (W1c[c] - W1b[c]) * iadz1) * irho1;
        MX0n[c] = MX0p[c];
        MN0n[c] = MN0p[c];
      }
    }
    if (!hasB) {
      // ghost level: f1(nzm+1) := f1(nzm) (kc clamp, :562); harmless at p == 1
      MPD_UNROLL for (int c = 0; c < W; ++c) F1c[c + 1] = F1b[c + 1];
    }
    e1[e_own] = F1c[1];
    e1[NS * SLW + e_own] = F1c[W];

    // ---- next plane into flight (or the clamped ghost copy of the last one)
    double Fn[H], Un[W + 1], Wn[H], rho_n, adz_n, rhow_n;
    if (p + 1 <= nzm) {
      const __amdgpu_buffer_rsrc_t rf = make_rsrc(fb + f_plane * p, f_bytes - f_plane * p);
      const __amdgpu_buffer_rsrc_t ru = make_rsrc(ub + u_plane * p, u_bytes - u_plane * p);
      const __amdgpu_buffer_rsrc_t rw = make_rsrc(wb + w_plane * p, w_bytes - w_plane * p);
      MPD_UNROLL for (int h = 0; h < H; ++h) Fn[h] = ld_b64(rf, vf0 + h * cs);
      MPD_UNROLL for (int c = 0; c <= W; ++c) Un[c] = ld_b64(ru, vu0 + c * cs);
      MPD_UNROLL for (int h = 0; h < H; ++h) Wn[h] = ld_b64(rw, vw0 + h * cs);
      rho_n = ld_b64(make_rsrc(rb + k_plane * p, k_plane * (nzm - p)), vk);
      adz_n = ld_b64(make_rsrc(ab + k_plane * p, k_plane * (nzm - p)), vk);
      rhow_n = ld_b64(make_rsrc(rwb + k_plane * p, k_plane * (nzm + 1 - p)), vk);
    } else {
      MPD_UNROLL for (int h = 0; h < H; ++h) { Fn[h] = Fc[h]; Wn[h] = Wc[h]; }
      MPD_UNROLL for (int c = 0; c <= W; ++c) Un[c] = Uc[c];
      rho_n = rho_c; adz_n = adz_c; rhow_n = rhow_c;
    }

    strip_sync<NWV>();  // ---- S1: f1 edges, fl1 (this step) and fl3 (last step) visible
    F1c[0] = e1[NS * SLW + e_left];
    F1c[W + 1] = e1[e_right];
    if (!hasB) { F1c[0] = F1b[0]; F1c[W + 1] = F1b[W + 1]; }

    // flux: strip 0 sums the upwind terms of level p (:541-547), strip 1
    // continues the sum of level p-3 with the limited terms (:624)
    if (NWV == 1 || wave * SPW <= 1) {
      const bool role1 = strip == 1;
      const double* src = role1 ? fl3 : fl1;
      double s = role1 ? ps[(kD & 3) * SLW + sll] : 0.0;
      for (int g = 2; g <= nx + 1; ++g) s = s + src[g * SLW + sll];
      if (strip == 0) ps[(p & 3) * SLW + sll] = s;
      if (role1 && hasD && sl_ok) st_b64(make_rsrc(xb + k_plane * (kD - 1), k_plane), vk, s);
    }

    // ================= stage B: W2 at level kB = p-1 (:580-582, :586) =======
    double W2c[W];
    if (kB >= 2 && kB <= nzm) {
      MPD_UNROLL for (int c = 0; c < W; ++c) {
        const double ad = andiff(F1b[c + 1], F1c[c + 1], Wb[c + 1], irhow1);
        const double x = F1b[c + 2] + F1c[c + 2] - F1b[c] - F1c[c];
        const double su = Ua[c] + Ub[c] + Ub[c + 1] + Ua[c + 1];
        W2c[c] = ad - across(x, Wb[c + 1], su) * irho1;
      }
    } else if (kB > nzm) {
      MPD_UNROLL for (int c = 0; c < W; ++c) W2c[c] = W2b[c];  // kc clamp in :602
    } else {
      MPD_UNROLL for (int c = 0; c < W; ++c) W2c[c] = 0.0;  // www(:,:,:,1) = 0 (:586)
    }

    // ================= stage C: level kC = p-2 ================================
    double U2[W + 1], MXc[H], MNc[H];
    if (hasC) {
      if (kC == 1) {  // ghost level 0 := level 1
        MPD_UNROLL for (int h = 0; h < H; ++h) F1a[h] = F1b[h];
      }
      // :569  dd = 2./(kc-kb)/adz  ==  (2 or 1) * (1/adz), exactly
      const double dd = ((kC == 1 || kC == nzm) ? 2.0 : 1.0) * iadz2;
      MPD_UNROLL for (int c = 0; c <= W; ++c) {  // :571-573, column i0+c (h=c+1), ib -> h=c
        const double ad = andiff(F1b[c], F1b[c + 1], Ua[c], irho2);
        const double x = dd * (F1c[c] + F1c[c + 1] - F1a[c] - F1a[c + 1]);
        const double sw = Wa[c] + Wb[c] + Wa[c + 1] + Wb[c + 1];
        U2[c] = ad - across(x, Ua[c], sw) * irho2;
      }
      MPD_UNROLL for (int c = 0; c < W; ++c) {
        // :596-597
        const double mx1 = dmax(dmax(dmax(dmax(dmax(F1b[c], F1b[c + 2]), F1a[c + 1]), F1c[c + 1]),
                                     F1b[c + 1]), MX0f[c]);
        const double mn1 = dmin(dmin(dmin(dmin(dmin(F1b[c], F1b[c + 2]), F1a[c + 1]), F1c[c + 1]),
                                     F1b[c + 1]), MN0f[c]);
        // :606-609
        MXc[c + 1] = rho2 * (mx1 - F1b[c + 1]) /
                     (pn(U2[c + 1]) + pp(U2[c]) + iadz2 * (pn(W2c[c]) + pp(W2b[c])) + eps);
        MNc[c + 1] = rho2 * (F1b[c + 1] - mn1) /
                     (pp(U2[c + 1]) + pn(U2[c]) + iadz2 * (pp(W2c[c]) + pn(W2b[c])) + eps);
      }
    } else {
      MPD_UNROLL for (int c = 0; c <= W; ++c) U2[c] = 0.0;
      MPD_UNROLL for (int c = 0; c < W; ++c) MXc[c + 1] = MNc[c + 1] = 0.0;
    }
    e2[0 * NS * SLW + e_own] = MXc[1];
    e2[1 * NS * SLW + e_own] = MNc[1];
    e2[2 * NS * SLW + e_own] = MXc[W];
    e2[3 * NS * SLW + e_own] = MNc[W];

    strip_sync<NWV>();  // ---- S2: limiter-ratio edges visible
    MXc[0] = e2[2 * NS * SLW + e_left];
    MNc[0] = e2[3 * NS * SLW + e_left];
    MXc[W + 1] = e2[0 * NS * SLW + e_right];
    MNc[W + 1] = e2[1 * NS * SLW + e_right];

    // ================= stage D: limited fluxes of kC, final field of kD =======
    double U3[W + 1], W3c[W];
    if (hasC) {
      if (kC == 1) {  // ghost level 0 := level 1 (kb clamp in :623)
        MPD_UNROLL for (int c = 0; c < W; ++c) { MXa[c] = MXc[c + 1]; MNa[c] = MNc[c + 1]; }
      }
      MPD_UNROLL for (int c = 0; c <= W; ++c)  // :618
        U3[c] = pp(U2[c]) * dmin(dmin(1.0, MXc[c + 1]), MNc[c]) -
                pn(U2[c]) * dmin(dmin(1.0, MXc[c]), MNc[c + 1]);
      MPD_UNROLL for (int c = 0; c < W; ++c) {  // :623
        W3c[c] = pp(W2b[c]) * dmin(dmin(1.0, MXc[c + 1]), MNa[c]) -
                 pn(W2b[c]) * dmin(dmin(1.0, MXa[c]), MNc[c + 1]);
        fl3[(g0 + c) * SLW + sll] = W3c[c];
      }
    } else {
      // kC == nz: www(:,:,:,nz) = 0 (:511)
      MPD_UNROLL for (int c = 0; c <= W; ++c) U3[c] = 0.0;
      MPD_UNROLL for (int c = 0; c < W; ++c) W3c[c] = 0.0;
    }
    if (hasD) {
      const __amdgpu_buffer_rsrc_t rfo = make_rsrc(fb + f_plane * (kD - 1), f_plane);
      MPD_UNROLL for (int c = 0; c < W; ++c) {
        const int i = i0 + c;
        const double fin = dmax(0.0, F1a[c + 1] - (dU3a[c] + (W3c[c] - W3a[c]) * iadz3) * irho3);  // :634
        const bool interior = i >= 1 && i <= nx;
        const bool halo = i == -1 || i == 0 || i == nx + 1 || i == nx + 2;  // keep f1 (:557)
        const double v = interior ? fin : F1a[c + 1];
        if ((interior || halo) && sl_ok) st_b64(rfo, vf0 + (c + 1) * cs, v);
      }
    }

    // ================= rotate the rolling window ==============================
    MPD_UNROLL for (int c = 0; c < W; ++c) {
      dU3a[c] = U3[c + 1] - U3[c];
      W3a[c] = W3c[c];
      MXa[c] = MXc[c + 1];
      MNa[c] = MNc[c + 1];
      W2b[c] = W2c[c];
      MX0f[c] = MX0n[c];
      MN0f[c] = MN0n[c];
    }
    MPD_UNROLL for (int h = 0; h < H; ++h) {
      F1a[h] = F1b[h]; F1b[h] = F1c[h];
      Wa[h] = Wb[h]; Wb[h] = Wc[h]; Wc[h] = Wn[h];
      Fc[h] = Fn[h];
    }
    MPD_UNROLL for (int c = 0; c <= W; ++c) { Ua[c] = Ub[c]; Ub[c] = Uc[c]; Uc[c] = Un[c]; }
    irho3 = irho2; irho2 = irho1; irho1 = irho0;
    iadz3 = iadz2; iadz2 = iadz1; iadz1 = iadz0;
    irhow1 = irhow0;
    rho2 = rho1; rho1 = rho_c;
    rho_c = rho_n; adz_c = adz_n; rhow_c = rhow_n;
  }
}

}  // namespace MPDATA_NS
