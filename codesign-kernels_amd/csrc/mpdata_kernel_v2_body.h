// mpdata_kernel_v2_body.h -- "x-marching" fused MPDATA kernel for gfx950.
//
// Same arithmetic as mpdata_kernel_body.h (the k-marching kernel), i.e. one call
// of the reference routine
//   mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642,
// but with the roles of the two stencil axes exchanged so that the rolling
// window fits in ~100 VGPRs (4 waves per SIMD) instead of spilling:
//
//   * a LANE owns one (CRM instance sl, vertical level k) pair; a wave holds
//     64/LPS instances x LPS lanes (LPS >= nz, a power of two); the vertical
//     neighbours kb = max(1,k-1), kc = min(nzm,k+1) (reference :515-516) are the
//     adjacent lanes: derived quantities come by DPP (wave_shr / wave_shl, the
//     clamps fused into the move as v_cndmask_b32_dpp), raw inputs by extra LDS
//     reads with the clamp in the address; lane k = nz is a "ghost" level holding
//     w = 0, which makes www(:,:,:,nz) = 0 (:511) fall out of the arithmetic;
//   * the kernel marches over the x columns q = -2 .. nx+3 (+ one epilogue that
//     writes back the last three columns); the horizontal
//     dependencies (radius 3) become a 3-column
//     software pipeline in registers:
//         step q:  upwind fluxes of column q, first-pass f1 of column q-1,
//                  antidiffusive U2 of q-1 and W2 / limiter ratios of q-2,
//                  limited fluxes of q-2, final field of column q-3;
//     every condition on q is wave-uniform; the steady-state columns
//     (4 <= q <= nx) run a condition-free body, and the rolling window is a set
//     of 3-slot rings indexed by (column mod 3) with the loop unrolled by 3, so
//     no register-to-register rotation is executed;
//   * HBM is read and written in rows that are contiguous along sl.  The 16
//     instances of a workgroup give 128-byte rows.  The 3*nzm rows (f,u,w) of
//     column q+3 are fetched while column q is computed, by LDS-DMA
//     (buffer_load_dword ... lds: no VGPR staging, no ds_write), two rows per
//     wave instruction, into a 4-column LDS ring; rows are XOR-swizzled on the
//     SOURCE address so that the transposed read (lanes along k) is free of
//     bank conflicts.  The finished column goes back through a padded LDS
//     tile (row stride 17 doubles) and full-row stores.  Every step issues the
//     same number of vector-memory operations per wave (1 store + 6 DMA), so
//     one counted s_waitcnt vmcnt(14) + a raw s_barrier per step is the only
//     synchronisation and three columns stay in flight across it; the step's
//     vector-memory instructions are issued at its end;
//   * irho, iadz, irhow, dd (:552-553,:565,:569) depend on (sl,k) only and are
//     computed ONCE per lane, not once per column.
//
// The kernel is a template over the type R a lane computes in -- double, float, or
// float2 = two adjacent instances per lane (packed fp32 arithmetic, the fp64
// kernel's data movement) -- and over BIG (arrays of 4 GiB or more: per-wave
// descriptor bases).  DESIGN.md sections 4.1, 4.4, 4.5 have the measurements.
//
// f is bit-identical to the reference in the EXACT build.  flux(k): the
// reference adds the limited terms one by one onto the finished upwind sum
// (:545,:624).  EXACT parks the nx limited fluxes of every lane and a finishing
// kernel adds them in that order (bit-identical; xmarch_flux_finish_kernel);
// FAST -- and EXACT on arrays of 4 GiB and more, or with MPDATA_EXACT_FLUX=sum --
// accumulates (sum_i upwind) + (sum_i limited), each sum in the reference's i
// order: equal to 1e-13 relative.
//
// Built with -fno-honor-nans: fmax/fmin then need no operand canonicalisation
// (v_max_f64 x,x).  No value-changing transformation is enabled by it; inputs
// containing NaN are outside the contract (the reference's own max/min
// intrinsics are processor-dependent on NaN as well).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "mpdata_args.h"

namespace MPDATA_NS {
namespace v2 {

// Register park of the EXACT kernels (bit-identical flux without a park array): PK[STRIDE * trip + OFF] = v with a
// compile-time index in every branch of a wave-uniform chain on the trip number, so that the array stays in registers.
// (The empty asm statement keeps the optimiser from merging the branches' stores into ONE store through a selected
//  pointer, which would turn the array into scratch memory.)
template <int IDX, typename R, int N> __device__ __forceinline__ void pk_set(R (&PK)[N], const R v) {
  if constexpr (IDX >= 0 && IDX < N) {
    PK[IDX] = v;
    asm volatile("; park %0" ::"i"(IDX));
  }
}
template <int STRIDE, int OFF, typename R, int N, int... TT>
__device__ __forceinline__ void pk_put(R (&PK)[N], const int trip, const R v, std::integer_sequence<int, TT...>) {
  ((trip == TT ? pk_set<STRIDE * TT + OFF>(PK, v) : (void)0), ...);
}


// ---- real-type helpers.  R is the type a LANE computes in:
//        double  the reference as shipped (:13);
//        float   its fp32 build (:12), one instance per lane;
//        f32x2   fp32, TWO adjacent instances per lane: the same 8 bytes per lane and element
//                as the fp64 kernel (identical tiling, LDS image, DMA and stores), add / mul /
//                fma as packed instructions (v_pk_*_f32); needs an even number of instances.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dmax(double x, double y) { return __builtin_fmax(x, y); }
__device__ __forceinline__ double dmin(double x, double y) { return __builtin_fmin(x, y); }
__device__ __forceinline__ float dmax(float x, float y) { return __builtin_fmaxf(x, y); }
__device__ __forceinline__ float dmin(float x, float y) { return __builtin_fminf(x, y); }
__device__ __forceinline__ f32x2 dmax(f32x2 x, f32x2 y) { return __builtin_elementwise_max(x, y); }
__device__ __forceinline__ f32x2 dmin(f32x2 x, f32x2 y) { return __builtin_elementwise_min(x, y); }
__device__ __forceinline__ double rabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float rabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ f32x2 rabs(f32x2 x) { return __builtin_elementwise_abs(x); }
__device__ __forceinline__ double rldexp(double x, int e) { return __builtin_ldexp(x, e); }
__device__ __forceinline__ float rldexp(float x, int e) { return __builtin_ldexpf(x, e); }
__device__ __forceinline__ f32x2 rldexp(f32x2 x, int e) { return f32x2{__builtin_ldexpf(x.x, e), __builtin_ldexpf(x.y, e)}; }
// a >= 0 ? x : y per element
__device__ __forceinline__ double sel_ge0(double a, double x, double y) { return a >= 0.0 ? x : y; }
__device__ __forceinline__ float sel_ge0(float a, float x, float y) { return a >= 0.0f ? x : y; }
__device__ __forceinline__ f32x2 sel_ge0(f32x2 a, f32x2 x, f32x2 y) {
  return f32x2{a.x >= 0.0f ? x.x : y.x, a.y >= 0.0f ? x.y : y.y};
}
// Statement functions of the reference (:500-503), left-to-right.
template <typename R>
__device__ __forceinline__ R andiff(R x1, R x2, R a, R b) {
  return (rabs(a) - a * a * b) * R(0.5) * (x2 - x1);
}
template <typename R>
__device__ __forceinline__ R across(R x1, R a1, R a2) {
  return R(0.03125) * a1 * a2 * x1;
}
// max(0,a)*x + min(0,a)*y of the reference's upwind / limited fluxes (:532, :537, :618, :623).
// One of the two products is an exact zero, so the value equals a * (a >= 0 ? x : y)
// (the sign of a zero result aside); one select + one multiply instead of max, min, two
// multiplies and an add.
template <typename R>
__device__ __forceinline__ R upwind(R a, R x, R y) { return a * sel_ge0(a, x, y); }
template <typename R>
__device__ __forceinline__ R pp(R y) { return dmax(R(0), y); }
template <typename R>
__device__ __forceinline__ R pn(R y) { return -dmin(R(0), y); }

// 1 / d for the FAST limiter ratios: hardware reciprocal + ONE Newton step, without the scale /
// fixup handling of subnormal and huge operands the limiter never sees (d >= eps^2 > 0, finite).
// v_rcp_f64 delivers about 2^-27 relative error, one step squares it; a second step changes
// nothing measurable (tools/fast_err.py: max |df| 8.9e-16 on conditioned inputs and relative L1
// 3.2e-16 against 2.7e-16 on the reference-raw law, tolerances 1e-12 / 1e-14) and costs two
// FMAs per cell.
__device__ __forceinline__ double recip_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
// 1 / d of the per-lane constants (rho, adz, rhow*adz: positive, normal range).  FAST, fp64: hardware
// reciprocal + two Newton steps (6 instructions against the ~30 of the IEEE division sequence,
// three times per wave; within 1 ulp of the quotient).  EXACT and fp32: the division.
__device__ __forceinline__ double recip_const(double d) {
#ifdef MPDATA_FAST_DIV
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
  return __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
#else
  return 1.0 / d;
#endif
}
__device__ __forceinline__ float recip_const(float d) { return 1.0f / d; }
__device__ __forceinline__ f32x2 recip_const(f32x2 d) { return f32x2{1.0f, 1.0f} / d; }
__device__ __forceinline__ float recip_nr(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  return __builtin_fmaf(r, e, r);
}
__device__ __forceinline__ f32x2 recip_nr(f32x2 d) {
  f32x2 r = f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f32x2 e = __builtin_elementwise_fma(-d, r, f32x2{1.0f, 1.0f});
  return __builtin_elementwise_fma(r, e, r);
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// Buffer addressing for the row transfers: wave-uniform descriptor + per-lane
// 32-bit byte offset (VGPR) + wave-uniform byte offset of the column (SGPR).
// The range (num_records) is clipped BELOW the out-of-range marker 0xFFFFFFF8 the kernel uses
// for lanes that own nothing, so those lanes stay out of range for any array size.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  const long long lim = 0xFFFFFFF0ll;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0,
                                           (int)(unsigned)(bytes > lim ? lim : bytes), 0x00020000);
}
template <int AUX = 0>
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, 0);
}
template <int AUX = 0>
__device__ __forceinline__ void st_row(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, 0);
}

// ---- vertical neighbours by DPP (register crossbar in the VALU) ---------------
// LDS-crossbar permutes (ds_bpermute) cost no VALU slot but ~5 LDS cycles per
// CU each and, measured with in-kernel clock stamps, enough power that the chip
// drops its clock by ~15 %; DPP moves are one 32-bit VALU pass per register (two per double).
// A lane without a source lane reads 0 (bound_ctrl), so the move needs no prior copy of
// its destination (with bound_ctrl off the destination is an input -- "keep the old
// value" -- and hipcc spends a v_mov per half to set it up).
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double src) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(src), CTRL, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(src), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ f32x2 dpp_mov(f32x2 src) {
  const u32x2 b = __builtin_bit_cast(u32x2, src);
  u32x2 o;
  o.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b.x, CTRL, 0xF, 0xF, true);
  o.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)b.y, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(f32x2, o);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, src), CTRL, 0xF, 0xF, true));
}
#define MPD_DPP_WAVE_SHL1 0x130
#define MPD_DPP_WAVE_SHR1 0x138

// value of lane-1 (level k-1); lane 0 reads 0
template <typename R>
__device__ __forceinline__ R shift_dn(R x) { return dpp_mov<MPD_DPP_WAVE_SHR1>(x); }
// value of lane+1 (level k+1); lane 63 reads 0
template <typename R>
__device__ __forceinline__ R shift_up(R x) { return dpp_mov<MPD_DPP_WAVE_SHL1>(x); }
// ... with the clamps kb = max(1,k-1) / kc = min(nzm,k+1): `own` is the wave mask of the
// lanes that keep their own value.  Select and move are ONE instruction per 32-bit register
// (v_cndmask_b32 with a DPP source operand; hipcc emits v_mov_b32_dpp + v_cndmask_b32 for
// the same thing written in C).  A DPP operand written by the preceding VALU instruction needs
// two wait states, which the compiler cannot see inside an asm block: the mask move and an
// s_nop 0 are those two.
// Executed by ALL lanes (under a divergent EXEC mask a switched-off source lane counts as
// missing).
#define MPD_DPP_NOP "s_nop 0"
#define MPD_CNDMASK_DPP64(CTRL)                                                                \
  int lo, hi;                                                                                  \
  const int xl = __double2loint(x), xh = __double2hiint(x);                                    \
  asm("s_mov_b64 vcc, %4\n\t" MPD_DPP_NOP "\n\t"                                                      \
      "v_cndmask_b32_dpp %0, %2, %2, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_cndmask_b32_dpp %1, %3, %3, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0"      \
      : "=&v"(lo), "=&v"(hi)                                                                   \
      : "v"(xl), "v"(xh), "s"(own)                                                             \
      : "vcc");                                                                                \
  return __hiloint2double(hi, lo);
#define MPD_CNDMASK_DPP32(CTRL)                                                                \
  int r;                                                                                       \
  const int xi = __builtin_bit_cast(int, x);                                                   \
  asm("s_mov_b64 vcc, %2\n\t" MPD_DPP_NOP "\n\t"                                                      \
      "v_cndmask_b32_dpp %0, %1, %1, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0"      \
      : "=&v"(r)                                                                               \
      : "v"(xi), "s"(own)                                                                      \
      : "vcc");                                                                                \
  return __builtin_bit_cast(float, r);
__device__ __forceinline__ double shift_dn_clamped(double x, unsigned long long own) {
  MPD_CNDMASK_DPP64("wave_shr:1")
}
__device__ __forceinline__ double shift_up_clamped(double x, unsigned long long own) {
  MPD_CNDMASK_DPP64("wave_shl:1")
}
__device__ __forceinline__ f32x2 shift_dn_clamped(f32x2 x, unsigned long long own) {
  return __builtin_bit_cast(f32x2, shift_dn_clamped(__builtin_bit_cast(double, x), own));
}
__device__ __forceinline__ f32x2 shift_up_clamped(f32x2 x, unsigned long long own) {
  return __builtin_bit_cast(f32x2, shift_up_clamped(__builtin_bit_cast(double, x), own));
}
__device__ __forceinline__ float shift_dn_clamped(float x, unsigned long long own) {
  MPD_CNDMASK_DPP32("wave_shr:1")
}
__device__ __forceinline__ float shift_up_clamped(float x, unsigned long long own) {
  MPD_CNDMASK_DPP32("wave_shl:1")
}
#undef MPD_CNDMASK_DPP64
#undef MPD_CNDMASK_DPP32

template <typename R_, int LPS, int G_>
struct TileV2 {
  using R = R_;
  static constexpr int G = G_;                 // CRM instances per workgroup: rows of G*sizeof(R) bytes
  static constexpr int ROWB = G_ * (int)sizeof(R_);   // bytes of a row segment (128 or 256)
  static constexpr int RPI = 256 / ROWB;       // rows moved by one DMA wave instruction (256 B)
  static constexpr int EPI = 256 / (int)sizeof(R_);   // elements moved by one DMA wave instruction
  static constexpr int SLP = 64 / LPS;         // instances per wave
  static constexpr int NWV = G_ / SLP;         // waves per workgroup
  static constexpr int THREADS = 64 * NWV;     // = G * LPS
  static constexpr int RS = G_ + 1;            // out tile: LDS row stride in elements
  static constexpr int NZM_MAX = LPS - 1;
  static constexpr int NSLOT = 4;              // input ring: columns q .. q+3 (a 5-slot ring,
                                               // one more column in flight, measured 1.5 % slower)
  static constexpr int ARR = LPS * G_;         // elements of one array block (LPS rows x G)
  static constexpr int IN_SLOT = 3 * ARR + G_; // f,u,w rows of one column + one row of zeros
  static constexpr int OUT_SLOT = NZM_MAX * RS;
  static constexpr int LDS_ELEMS = NSLOT * IN_SLOT + 2 * OUT_SLOT;
  // DMA instructions per array, column and wave: the NZM_MAX rows of an array are
  // ceil(NZM_MAX / RPI) instructions, dealt out over the NWV waves
  static constexpr int NI_MAX = (NZM_MAX + RPI - 1) / RPI;
  static constexpr int NIT = (NI_MAX + NWV - 1) / NWV;
  static constexpr int VM_PER_STEP = 1 + 3 * NIT;  // 1 store + 3*NIT DMA per wave and step
  // waves per SIMD the register budget must allow: 128 VGPRs (8-byte elements: 2 workgroups of
  // 8 waves per CU at LPS = 32; one-instance-per-lane fp32: 1 workgroup of 16 waves)
  static constexpr int MIN_WAVES = 4;
  static_assert(ROWB == 128 || ROWB == 256, "row segments of 128 or 256 bytes");
  static_assert(THREADS <= 1024 && NIT >= 1 && NIT <= 2, "workgroup shape");
};

// Rolling window: rings of 3 slots indexed by (column mod 3).
template <typename R>
struct Window {
  R F0[3], PMX[3], PMN[3], U1[3], DW1[3];      // written for column q
  R F1[3], F1D[3], F1U[3], MX0[3], MN0[3];     // written for column q-1
  R UR[3], UD[3], PW[3], SW[3], WR[3];         // u, u(kb), w+w(kc), w-sum, w of column q
  R SU[3], U2P[3], U2N[3];                     // written for column q-1 (pp / pn of U2)
  R MXN[3], MNN[3], U3[3], DW3[3];             // written for column q-2
  R SF[3];                                     // wave-major two-tracer FAST form only: f1 + f1(kb) of column q-1
};

// workgroup (dispatch index) -> (tracer, instance group): tracer is the FASTEST index, so the workgroups that share the
// same rows of u, w, rho, rhow, adz are dispatched together; tracer batches are additionally re-dealt so that all
// tracers of group g go to XCD g % 8, back to back (workgroups are dealt to the 8 XCDs round-robin in dispatch order:
// 24 of 25 reads of that group's u, w rows then hit in that XCD's L2)
__device__ __forceinline__ void xm_block_map(const unsigned L, const unsigned nblocks, const unsigned ntr, int& tr, unsigned& grp) {
  tr = (int)(L % ntr);
  grp = L / ntr;
  if (ntr > 1) {
    const unsigned nxcd = 8;
    const unsigned full = ((nblocks / ntr) / nxcd) * nxcd;   // groups the re-deal covers
    if (L < ntr * full) {
      const unsigned xcd = L % nxcd, j = L / nxcd;
      tr = (int)(j % ntr);
      grp = (j / ntr) * nxcd + xcd;
    }
  }
}

// BIG: some array is 4 GiB or larger.  Then every buffer descriptor starts at the first row
// (level) its wave touches, so the per-lane 32-bit offsets only span the wave's few rows (the
// column offset, a 32-bit scalar, needs ncrms * (nx+6) * sizeof(R) < 4 GiB).  Otherwise one
// descriptor per array serves all waves (7 descriptors instead of 3 cost 2 % through scalar
// register pressure, hence the two instantiations).
// NT (fp64, one tracer per launch): row fetches and row stores with the streaming cache policy.
// Every 128-byte row segment is touched by exactly one fetch and one store instruction, so
// nothing is lost by not keeping it, and the pair of hints buys 1-3 % (either alone: nothing).
// NPK > 0 (EXACT, nx <= NPK; round 5): the lane's limited vertical fluxes stay in NPK registers and are added onto the
// finished upwind sum behind the march, in the reference's order -- no park array, no finishing kernel; 2 waves per SIMD.
template <typename R, int LPS, int G, bool BIG, bool NT = false, int NPK = 0>
__global__ void __launch_bounds__(G * LPS, (NPK > 0 ? 2 : TileV2<R, LPS, G>::MIN_WAVES))
mpdata_advect_xmarch_kernel(const MpdataArgsT<R> a) {
  constexpr int LD_AUX = NT ? 2 : 0;
  constexpr int ST_AUX = (NT && std::is_same<R, double>::value) ? 2 : 0;
  using T = TileV2<R, LPS, G>;
  constexpr int RS = T::RS, SLP = T::SLP, RPI = T::RPI;
  constexpr int RB = (int)sizeof(R);  // bytes per element
  __shared__ R lds[T::LDS_ELEMS];
  R* const in_slot0 = lds;
  R* const out_slot0 = lds + T::NSLOT * T::IN_SLOT;

  const int nx = a.nx, nz = a.nz, nzm = nz - 1;
  const long long ncrms = a.ncrms;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform for the compiler too
  int tr;
  unsigned grp;
  xm_block_map(blockIdx.x, gridDim.x, (unsigned)a.ntracers, tr, grp);
  const long long sl_base = (long long)grp * G;

  R* const f = a.f + (long long)tr * a.f_tstride;
  R* const flux = a.flux + (long long)tr * a.flux_tstride;

  // ---- compute-side mapping: lane -> (instance, level) ----------------------
  const int kk = lane % LPS;               // k - 1
  const int k = kk + 1;
  const int sl_l = wave * SLP + lane / LPS;  // instance inside the workgroup
  const bool lvl_ok = k <= nzm;            // real level (else ghost / dead lane)
  const int kl = lvl_ok ? k : nzm;         // level whose data the lane loads
  long long sl_c = sl_base + sl_l;
  const bool slc_ok = sl_c < ncrms;
  if (!slc_ok) sl_c = ncrms - 1;

  // per-lane constants (:552, :553, :565, :569)
  const R eps = (R)1.e-10f;  // :509, fp32 literal
  const long long kidx = sl_c + ncrms * (long long)(kl - 1);
  const R RHO = a.rho[kidx];
  const R adz_l = a.adz[kidx];
  const R rhow_l = a.rhow[kidx];
  const R IRHO = recip_const(RHO);
  const R IADZ = recip_const(adz_l);
  const R IRHOW = recip_const(rhow_l * adz_l);
  // :569  dd = 2./(kc-kb)/adz = (2 or 1)*(1/adz) exactly; the factor 2 is applied as an
  // exponent step on dd*(...) (exact scaling, bit-identical)
  const int dd_exp = (k == 1 || k == nzm) ? 1 : 0;
#ifdef MPDATA_FAST_DIV
  // FAST carries the second-pass fluxes DOUBLED (andiff's factor 0.5 is not applied: one
  // multiplication fewer per flux) and the limiter ratios HALVED (doubled denominators, clamp
  // at 0.5), so the limited fluxes come out unscaled; every scaling is by a power of two, i.e.
  // the results are bit-identical to the unscaled evaluation
  const R EPS_D = eps + eps, LIM = R(0.5);
  const R KU = rldexp(R(0.03125) * IRHO * IADZ, dd_exp + 1);
  // www(:,:,:,1) = 0 (:586) lives in the constant: at k = 1 the advective part of W2 is an
  // exact zero already (kb = k, f - f(kb) = 0), a zero KW removes the cross part
  const R KW = (k == 1) ? R(0) : R(0.0625) * IRHO;
#else
  const R EPS_D = eps, LIM = R(1);
#endif
  const bool k_is_1 = k == 1;
  const bool k_ge_nzm = k >= nzm;
  const unsigned long long own_dn = __builtin_amdgcn_ballot_w64(k_is_1);     // kb = k
  const unsigned long long own_up = __builtin_amdgcn_ballot_w64(k_ge_nzm);   // kc = k

  // ---- write-back mapping: thread -> (row = level, instance) -----------------
  const int t_row = tid / G;  // level index k-1 of the row this thread stores
  const int t_sl = tid % G;
  const bool t_act = t_row < nzm;
  long long sl_t = sl_base + t_sl;
  const bool slt_ok = sl_t < ncrms;
  if (!slt_ok) sl_t = ncrms - 1;
  const int t_rowc = t_act ? t_row : 0;
  const int t_lds = t_rowc * RS + t_sl;  // position inside an out tile
  // byte offset of (column index 0, this level, this instance); 32-bit: the host
  // picks this kernel only when every array is smaller than 2^32 bytes.  Threads
  // that own nothing get an out-of-range offset: the buffer range check drops
  // their store, and every wave still issues the same number of stores.
  const unsigned OOB = 0xFFFFFFF8u;
  const long long lvl_f = ncrms * (long long)(nx + 6), lvl_u = ncrms * (long long)(nx + 5),
                  lvl_w = ncrms * (long long)(nx + 4);           // elements between levels
  const long long end_f = lvl_f * nzm, end_u = lvl_u * nzm, end_w = lvl_w * nz;  // elements per array
  const int st_row0 = BIG ? min(wave * (64 / G), nzm - 1) : 0;   // descriptor base row of the wave's stores
  const unsigned tf = (t_act && slt_ok) ? (unsigned)((sl_t + lvl_f * (t_rowc - st_row0)) * RB) : OOB;
  const unsigned colb = (unsigned)(ncrms * RB);  // bytes between columns
  const __amdgpu_buffer_rsrc_t rsf = make_rsrc(f + lvl_f * st_row0, (end_f - lvl_f * st_row0) * RB);

  // ---- DMA mapping: one wave instruction moves 256 bytes, 4 per lane = RPI rows of G
  //      elements (128-B rows: two rows, lanes 0-31 row 2j, lanes 32-63 row 2j+1; 256-B rows:
  //      one).  LDS image of a column: [array][row][G elements], element (row, sl) stored at
  //      position sl ^ ((row / rows-per-instruction) & (G-1)): the swizzle is applied to the
  //      SOURCE address.
  constexpr int LPR = T::ROWB / 4;       // lanes (dwords) per row
  constexpr int DPE = RB / 4;            // dwords per element
  static_assert(T::NIT * T::NWV * RPI == LPS, "the wave's DMA instructions tile the LPS rows of an array block");
  unsigned vdf[T::NIT], vdu[T::NIT], vdw[T::NIT];  // per-lane source byte offsets of the wave's instructions
  __amdgpu_buffer_rsrc_t rdf[T::NIT], rdu[T::NIT], rdw[T::NIT];  // ... and their descriptors
  int jd[T::NIT];
#pragma unroll
  for (int it = 0; it < T::NIT; ++it) {
    const int j = wave + it * T::NWV;
    jd[it] = j;
    // rows beyond nzm (the block has LPS of them) have nothing to fetch: an out-of-range
    // source offset makes the DMA write zeros to that (never read) part of the block
    // without a memory access, and the wave's instruction count stays fixed
    const int row0 = BIG ? min(j * RPI, nzm - 1) : 0;   // descriptor base row of the instruction
    const int row = j * RPI + lane / LPR;
    const bool row_ok = row < nzm;
    const int p = (lane % LPR) / DPE;
    long long sl_d = sl_base + (p ^ (j & (G - 1)));
    if (sl_d >= ncrms) sl_d = ncrms - 1;
    const unsigned part = (lane % DPE) * 4;
    const int dr = row_ok ? row - row0 : 0;
    vdf[it] = row_ok ? (unsigned)((sl_d + lvl_f * dr) * RB) + part : OOB;
    vdu[it] = row_ok ? (unsigned)((sl_d + lvl_u * dr) * RB) + part : OOB;
    vdw[it] = row_ok ? (unsigned)((sl_d + lvl_w * dr) * RB) + part : OOB;
    rdf[it] = make_rsrc(f + lvl_f * row0, (end_f - lvl_f * row0) * RB);
    rdu[it] = make_rsrc(a.u + lvl_u * row0, (end_u - lvl_u * row0) * RB);
    rdw[it] = make_rsrc(a.w + lvl_w * row0, (end_w - lvl_w * row0) * RB);
  }
  // compute-side read position inside one array block of a slot
  auto lds_pos = [&](int level) __attribute__((always_inline)) {  // position of (level, this instance)
    return (level - 1) * G + (sl_l ^ (((level - 1) / RPI) & (G - 1)));
  };
  const int c_lds = lds_pos(kl);
  // the raw inputs of the vertical neighbours come straight from the staged column (an LDS
  // read each, the kb / kc clamps live in the address) instead of a cross-lane move
  const int c_dn = lds_pos(max(kl - 1, 1));
  const int c_up = lds_pos(min(kl + 1, nzm));
  // w of the lanes above nzm (ghost level nz and dead lanes) is 0: they read the zero row of the slot
  const int c_lds_w = lvl_ok ? 2 * T::ARR + c_lds : 3 * T::ARR + sl_l;


  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // all DMA of column `col` into its ring slot (6 instructions per wave)
  auto dma_col = [&](const int col) __attribute__((always_inline)) {
#ifdef MPD2_ABL_NODMA  // timing ablation only: the arithmetic on whatever the LDS ring holds
    return;
#endif
    const unsigned cf = colb * (unsigned)(min(max(col, -2), nx + 3) + 2);
    const unsigned cu = colb * (unsigned)(min(max(col, -1), nx + 3) + 1);
    const unsigned cw = colb * (unsigned)(min(max(col, -1), nx + 2) + 1);
    R* slot = in_slot0 + (col & (T::NSLOT - 1)) * T::IN_SLOT;
#pragma unroll
    for (int it = 0; it < T::NIT; ++it) {
      R* d = slot + jd[it] * T::EPI;  // 256 bytes per instruction
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdf[it], (lds_ptr_t)(d), 4, (int)vdf[it], (int)cf, 0, LD_AUX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdu[it], (lds_ptr_t)(d + T::ARR), 4, (int)vdu[it], (int)cu, 0, LD_AUX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rdw[it], (lds_ptr_t)(d + 2 * T::ARR), 4, (int)vdw[it], (int)cw, 0, LD_AUX);
    }
  };

  Window<R> S;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S.F0[j] = S.PMX[j] = S.PMN[j] = S.U1[j] = S.DW1[j] = R(0);
    S.F1[j] = S.F1D[j] = S.F1U[j] = S.MX0[j] = S.MN0[j] = R(0);
    S.UR[j] = S.UD[j] = S.PW[j] = S.SW[j] = S.WR[j] = S.SU[j] = S.U2P[j] = S.U2N[j] = R(0);
    S.MXN[j] = S.MNN[j] = S.U3[j] = S.DW3[j] = R(0);
  }
  R S1 = R(0), S3 = R(0);
  // EXACT with a park array: the reference adds the limited vertical fluxes ONE BY ONE onto the finished upwind sum
  // (:545, :624), and the first of them exists 30 columns before that sum is complete -- every lane parks its nx limited
  // fluxes ([workgroup][column][thread]), flux gets the upwind sum alone, xmarch_flux_finish_kernel adds the rest in order
  // (not in the instantiation for arrays of 4 GiB and more: its per-wave descriptors leave no scalar registers for it)
#ifdef MPDATA_FAST_DIV
  constexpr bool CAN_PARK = false, REG_PARK = false;
#else
  constexpr bool CAN_PARK = !BIG && NPK == 0, REG_PARK = NPK > 0;
#endif
  static_assert(NPK % 3 == 0, "register park: whole trips of three columns");
  [[maybe_unused]] R PK[NPK > 0 ? NPK : 1];   // REG_PARK: limited vertical flux of column i = PK[i - 1] (constant indices only)
  [[maybe_unused]] const bool park = CAN_PARK && a.wpark != nullptr;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsp =
      make_rsrc(a.wpark + (long long)blockIdx.x * nx * T::THREADS, park ? (long long)nx * T::THREADS * RB : 0);

  // One step of the march.  PH = (q+2) mod 3 selects the ring slots at compile
  // time; FULL = steady state (4 <= q <= nx): every stage is active.
  auto step = [&](auto ph_tag, auto full_tag, const int q) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int C0 = PH, C1 = (PH + 2) % 3, C2 = (PH + 1) % 3, C3 = PH;  // slots of q, q-1, q-2, q-3

    // column q landed (this wave's DMA of it is 2 steps = 14 vector-memory ops
    // old), out tile of column q-4 written: then everyone's are, after the barrier
    static_assert(T::NSLOT == 4 && (T::VM_PER_STEP == 7 || T::VM_PER_STEP == 4), "the counted waits below assume them");
    // (only the DMA instructions of the two newer columns may still be outstanding: loads return in
    //  issue order, but a row store may be acknowledged before an older load has landed, so the
    //  stores issued in between must not be counted -- vmcnt(14) / vmcnt(8) would be a race)
    if constexpr (T::VM_PER_STEP == 7) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");


    // ---- this column, transposed: lanes along k -------------------------------
    const R* s = in_slot0 + (q & (T::NSLOT - 1)) * T::IN_SLOT;
    const R f0q = s[c_lds];
    const R uq = s[T::ARR + c_lds];
    const R wq = s[c_lds_w];  // ghost level: w = 0 (zero row)

#ifdef MPD2_ABL_NOCOMPUTE  // timing ablation only: data movement without the arithmetic
    if (q - 3 >= -1 && q - 3 <= nx + 2 && lvl_ok) out_slot0[(q & 1) * T::OUT_SLOT + (k - 1) * RS + sl_l] = f0q + uq + wq;
    asm volatile("" ::: "memory");
    st_row<ST_AUX>(rsf, (q - 4 >= -1 && q - 4 <= nx + 2) ? tf : OOB, colb * (unsigned)max(q - 4 + 2, 0),
           out_slot0[((q - 1) & 1) * T::OUT_SLOT + t_lds]);
    dma_col(q + T::NSLOT - 1);
    return;
#endif
#define DN_C(x) shift_dn_clamped((x), own_dn)
#define DN_P(x) shift_dn(x)
#define UP_C(x) shift_up_clamped((x), own_up)
#define UP_G(x) shift_up(x)
    const R f0d = s[c_dn];
    const R f0u = s[c_up];
    const R F0p = S.F0[C1];

    // ================= stage A =================================================
    // (every ring slot is written unconditionally so that the slot of column
    //  q-3 is dead afterwards: inactive stages store zeros)
    R U1q = R(0), DW1q = R(0), f1_1 = R(0), F1D_1 = R(0), F1U_1 = R(0), MX0_1 = R(0), MN0_1 = R(0);
    if (FULL || (q >= -1 && q <= nx + 3)) {
      U1q = upwind(uq, F0p, f0q);  // :532
      if (FULL || q <= nx + 2) {
        const R W1q = upwind(wq, f0d, f0q);  // :537
        DW1q = UP_G(W1q) - W1q;
        if (FULL || (q >= 1 && q <= nx)) S1 = S1 + W1q;  // :545
      }
      if (FULL || q >= 0) {
        f1_1 = F0p - ((U1q - S.U1[C1]) + S.DW1[C1] * IADZ) * IRHO;  // :557, column q-1
        F1D_1 = DN_C(f1_1);
        F1U_1 = UP_C(f1_1);
        MX0_1 = dmax(S.PMX[C1], f0q);  // :521-522 complete for column q-1
        MN0_1 = dmin(S.PMN[C1], f0q);
      }
    }
    S.U1[C0] = U1q;
    S.DW1[C0] = DW1q;
    S.F1[C1] = f1_1;
    S.F1D[C1] = F1D_1;
    S.F1U[C1] = F1U_1;
    S.MX0[C1] = MX0_1;
    S.MN0[C1] = MN0_1;
    // :521-522 for column q without its f(ic) term
    S.PMX[C0] = dmax(dmax(dmax(F0p, f0d), f0u), f0q);
    S.PMN[C0] = dmin(dmin(dmin(F0p, f0d), f0u), f0q);
    S.F0[C0] = f0q;

    // u / w sums for the antidiffusive cross terms (:573, :582), reference order
    const R ud = s[T::ARR + c_dn];
    const R wu = s[2 * T::ARR + c_up];
#ifdef MPDATA_FAST_DIV
    S.UD[C0] = uq + ud;                        // (the ring holds the pair sum here)
    S.SU[C1] = S.UD[C1] + S.UD[C0];
    S.PW[C0] = wq + wu;
    S.SW[C0] = S.PW[C1] + S.PW[C0];
#else
    S.SU[C1] = S.UD[C1] + S.UR[C1] + uq + ud;  // u(i,kb)+u(i,k)+u(ic,k)+u(ic,kb), i = q-1
    S.UD[C0] = ud;
    S.SW[C0] = S.PW[C1] + wq + wu;             // w(ib,k)+w(ib,kc)+w(i,k)+w(i,kc), i = q
    S.PW[C0] = wq + wu;
#endif
    S.UR[C0] = uq;
    S.WR[C0] = wq;

    // ================= stage B/C ===============================================
    R U2_1 = R(0), U2p_1 = R(0), U2n_1 = R(0), W2_2 = R(0), W2p = R(0), W2n = R(0), MXN_2 = R(0), MNN_2 = R(0);
    if (FULL || (q >= 1 && q <= nx + 3)) {
      {  // :571-573, column q-1
#ifdef MPDATA_FAST_DIV
        // FAST: the constant factors 0.03125 * irho * dd of the cross term are one per-lane
        // constant (same real-arithmetic value, 3 operations fewer)
        const R u1 = S.UR[C1];
        const R t1 = rabs(u1) - (u1 * u1) * IRHO;
        const R x4 = S.F1U[C2] + F1U_1 - S.F1D[C2] - F1D_1;
        U2_1 = t1 * (f1_1 - S.F1[C2]) - (KU * (u1 * S.SW[C1])) * x4;   // = 2 x (:571-573)
#else
        const R ad = andiff(S.F1[C2], f1_1, S.UR[C1], IRHO);
        const R x = rldexp(IADZ * (S.F1U[C2] + F1U_1 - S.F1D[C2] - F1D_1), dd_exp);
        U2_1 = ad - across(x, S.UR[C1], S.SW[C1]) * IRHO;
#endif
        U2p_1 = pp(U2_1);
        U2n_1 = pn(U2_1);
      }
      if (FULL || q >= 2) {  // column q-2
        {  // :580-582, :586
#ifdef MPDATA_FAST_DIV
          const R w2 = S.WR[C2];
          const R t1 = rabs(w2) - (w2 * w2) * IRHOW;
          const R x4 = F1D_1 + f1_1 - S.F1D[C3] - S.F1[C3];
          W2_2 = t1 * (S.F1[C2] - S.F1D[C2]) - (KW * (w2 * S.SU[C2])) * x4;  // = 2 x (:580-582); k = 1: 0
#else
          const R ad = andiff(S.F1D[C2], S.F1[C2], S.WR[C2], IRHOW);
          const R x = F1D_1 + f1_1 - S.F1D[C3] - S.F1[C3];
          const R v = ad - across(x, S.WR[C2], S.SU[C2]) * IRHO;
          W2_2 = k_is_1 ? R(0) : v;  // www(:,:,:,1) = 0 (:586)
#endif
        }
        const R W2u = UP_C(W2_2);
        // :596-597
        const R mx1 = dmax(dmax(dmax(dmax(dmax(S.F1[C3], f1_1), S.F1D[C2]), S.F1U[C2]), S.F1[C2]), S.MX0[C2]);
        const R mn1 = dmin(dmin(dmin(dmin(dmin(S.F1[C3], f1_1), S.F1D[C2]), S.F1U[C2]), S.F1[C2]), S.MN0[C2]);
        // :606-609
        W2p = pp(W2_2);
        W2n = pn(W2_2);
        const R den_mx = U2n_1 + S.U2P[C2] + IADZ * (pn(W2u) + W2p) + EPS_D;
        const R den_mn = U2p_1 + S.U2N[C2] + IADZ * (pp(W2u) + W2n) + EPS_D;
#ifdef MPDATA_FAST_DIV
        // FAST: one reciprocal for both ratios, r = 1/(den_mx*den_mn) (both >= eps = 1e-10,
        // finite), two Newton steps, then a/b = a * (other denominator) * r.
        // (Forming the denominators as (S -+ D)/2 from |.|-sums would save 5 operations but
        //  cancels: on the reference-raw input law it costs 4 digits of the rel-L1 agreement.)
        {
          const R dd2 = den_mx * den_mn;
          const R r = RHO * recip_nr(dd2);   // (rho once for both ratios)
          MXN_2 = (mx1 - S.F1[C2]) * (den_mn * r);
          MNN_2 = (S.F1[C2] - mn1) * (den_mx * r);
        }
#else
        MXN_2 = RHO * (mx1 - S.F1[C2]) / den_mx;
        MNN_2 = RHO * (S.F1[C2] - mn1) / den_mn;
#endif
        // the ratios are only ever used as min(1, ratio, ...) (:618, :623): keep them clamped
        MXN_2 = dmin(LIM, MXN_2);
        MNN_2 = dmin(LIM, MNN_2);
      }
    }
    S.U2P[C1] = U2p_1;
    S.U2N[C1] = U2n_1;
    S.MXN[C2] = MXN_2;
    S.MNN[C2] = MNN_2;

    // ================= stage D =================================================
    R U3_2 = R(0), DW3_2 = R(0);
    if (FULL || (q >= 3 && q <= nx + 3)) {
      U3_2 = S.U2P[C2] * dmin(MXN_2, S.MNN[C3]) - S.U2N[C2] * dmin(S.MXN[C3], MNN_2);  // :618
      if (FULL || q <= nx + 2) {
        const R mxd = DN_C(MXN_2);
        const R mnd = DN_C(MNN_2);
        const R W3 = W2p * dmin(MXN_2, mnd) - W2n * dmin(mxd, MNN_2);  // :623
        if constexpr (REG_PARK) {
          // column i = q - 2 = 3 * trip + PH - 4 (PH = (q + 2) mod 3): a wave-uniform branch on the trip number
          pk_put<3, PH - 5>(PK, (q + 2 - PH) / 3, W3, std::make_integer_sequence<int, NPK / 3 + 2>{});
        } else if (CAN_PARK && park) {   // EXACT, bit-identical flux: parked, added behind the launch in the reference's order
          unsigned z;   // (the lane's byte offset from the execution mask: no register carries it through the march)
          asm volatile("s_mov_b32 %0, 0" : "=s"(z));
          const unsigned lo = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) * (unsigned)RB;
          st_row<0>(rsp, lo, (unsigned)(((q - 3) * T::THREADS + wave * 64) * RB), W3);
        } else {
          S3 = S3 + W3;  // :624
        }
        DW3_2 = UP_G(W3) - W3;
      }
    }
    {
      const int n = q - 3;  // column finished in this step
      if (FULL || (n >= -1 && n <= nx + 2)) {
        R v = S.F1[C3];  // halo columns keep the first-pass value (:557)
        if (FULL || (n >= 1 && n <= nx))
          v = dmax(R(0), S.F1[C3] - ((U3_2 - S.U3[C3]) + S.DW3[C3] * IADZ) * IRHO);  // :634
        if (lvl_ok) out_slot0[(q & 1) * T::OUT_SLOT + (k - 1) * RS + sl_l] = v;
      }
    }
    S.U3[C2] = U3_2;
    S.DW3[C2] = DW3_2;

    // ---- the step's vector-memory instructions, at its END: a wave that the saturated
    //      memory pipeline holds up at issue would be waiting for the barrier here anyway
    //      (issued at the start of the step they delayed its arithmetic: 2 % slower).
    //      Write back the column finished in the previous step (n = q-4; an inactive step
    //      stores out of range -- dropped -- so the op count stays fixed), then column q+3
    //      into flight (its ring slot held column q-1, last read before this step's barrier).
    asm volatile("" ::: "memory");
    {
      const bool act = q - 4 >= -1 && q - 4 <= nx + 2;
      st_row<ST_AUX>(rsf, (FULL || act) ? tf : OOB, colb * (unsigned)max(q - 4 + 2, 0),
             out_slot0[((q - 1) & 1) * T::OUT_SLOT + t_lds]);
    }
    dma_col(q + T::NSLOT - 1);
  };

#undef DN_C
#undef DN_P
#undef UP_C
#undef UP_G

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using Full = std::true_type;
  using Part = std::false_type;

  const int q_first = -2;
  // the zero row of every ring slot (never touched by the DMA)
  if (tid < T::NSLOT * G) in_slot0[(tid / G) * T::IN_SLOT + 3 * T::ARR + (tid % G)] = R(0);

  // columns -2, -1, 0 into flight, each behind a dropped store so that the
  // counted wait of the first steps sees the steady-state op pattern
#pragma unroll
  for (int c = q_first; c < q_first + T::NSLOT - 1; ++c) {
    st_row<ST_AUX>(rsf, OOB, 0, R(0));
    dma_col(c);
  }

  // q advances by 3 per trip so that the ring phase is a compile-time constant.
  // Steps always run in whole triples (no step is skipped, so no ring slot stays
  // live across the loop edge); the march needs q up to nx+6 (write-back of the
  // last halo column), any further step of the last triple does nothing.
  {
    int q = q_first;
    for (; q < 4; q += 3) {  // columns -2 .. 3: pipeline fill
      step(P0{}, Part{}, q);
      step(P1{}, Part{}, q + 1);
      step(P2{}, Part{}, q + 2);
    }
    for (; q + 2 <= nx; q += 3) {  // steady state: every stage active, no conditions
      step(P0{}, Full{}, q);
      step(P1{}, Full{}, q + 1);
      step(P2{}, Full{}, q + 2);
    }
    // Remaining columns up to q = nx+3, the last step that computes anything (final field
    // of column nx, first-pass values of nx+1 and nx+2).  What is left after it -- three
    // columns to write back -- is done at once by the epilogue instead of by three more
    // (almost empty) steps with their barriers.
    for (; q + 2 <= nx + 3; q += 3) {
      step(P0{}, Part{}, q);
      step(P1{}, Part{}, q + 1);
      step(P2{}, Part{}, q + 2);
    }
    auto epilogue = [&](auto ph_tag) __attribute__((always_inline)) {
      constexpr int PH = decltype(ph_tag)::value;            // ring phase of the last step qe = nx+3
      constexpr int C1 = (PH + 2) % 3, C2 = (PH + 1) % 3;    // slots of columns nx+2, nx+1
      const int qe = nx + 3;
      // every wave has finished step qe (its out tile holds column nx), nothing is in
      // flight into the input ring any more: its first slot serves as a third tile
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      R* const tA = out_slot0 + (qe & 1) * T::OUT_SLOT;
      R* const tB = out_slot0 + ((qe & 1) ^ 1) * T::OUT_SLOT;
      R* const tC = in_slot0;
      if (lvl_ok) {
        tB[(k - 1) * RS + sl_l] = S.F1[C2];  // halo columns keep the first-pass value (:557)
        tC[(k - 1) * RS + sl_l] = S.F1[C1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      st_row<ST_AUX>(rsf, tf, colb * (unsigned)(nx + 2), tA[t_lds]);
      st_row<ST_AUX>(rsf, tf, colb * (unsigned)(nx + 3), tB[t_lds]);
      st_row<ST_AUX>(rsf, tf, colb * (unsigned)(nx + 4), tC[t_lds]);
    };
    const int rem = nx + 4 - q;  // 0, 1 or 2 single steps left
    if (rem == 0) {
      epilogue(P2{});
    } else if (rem == 1) {
      step(P0{}, Part{}, q);
      epilogue(P0{});
    } else {
      step(P0{}, Part{}, q);
      step(P1{}, Part{}, q + 1);
      epilogue(P1{});
    }
  }

  R fl = ((CAN_PARK && park) || REG_PARK) ? S1 : S1 + S3;  // :541-547, :624
  if constexpr (REG_PARK) {   // ... + www(1) + www(2) + ... + www(nx), one by one (:624)
#pragma unroll
    for (int i = 0; i < NPK; ++i)
      if (i < nx) fl = fl + PK[i];
  }
  if (lvl_ok && slc_ok) flux[sl_c + ncrms * (long long)(k - 1)] = fl;
}

// EXACT, bit-identical flux: flux (the upwind sum, as the kernel above left it) += the nx parked limited vertical
// fluxes of the lane, one by one in the reference's order i = 1 .. nx (:624).  Same grid, same thread -> (tracer,
// instance, level) mapping as the kernel above.  (Built without contraction in both variants' translation units?  No:
// only the EXACT one launches it.)
template <typename R, int LPS, int G>
__global__ void __launch_bounds__(G * LPS) xmarch_flux_finish_kernel(const MpdataArgsT<R> a) {
  using T = TileV2<R, LPS, G>;
  const int nx = a.nx, nzm = a.nz - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int tr;
  unsigned grp;
  xm_block_map(blockIdx.x, gridDim.x, (unsigned)a.ntracers, tr, grp);
  const int kk = lane % LPS;
  const long long sl_c = (long long)grp * G + wave * T::SLP + lane / LPS;
  if (kk >= nzm || sl_c >= a.ncrms) return;
  R* fp = a.flux + (long long)tr * a.flux_tstride + sl_c + a.ncrms * (long long)kk;
  const R* pp = a.wpark + (long long)blockIdx.x * nx * T::THREADS + tid;
  R acc = *fp;
  for (int i = 0; i < nx; ++i) acc = acc + pp[(long long)i * T::THREADS];
  *fp = acc;
}

}  // namespace v2
}  // namespace MPDATA_NS
