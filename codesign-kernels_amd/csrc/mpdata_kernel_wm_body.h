// mpdata_kernel_wm_body.h -- "wave-major" fused MPDATA kernel for gfx950: the kernel behind the
// plan API (include/mpdata_hip.h section 3).
//
// Same routine as mpdata_kernel_v2_body.h -- one call of the reference's
//   mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:477-642
// with the same lane mapping (a LANE owns one (CRM instance, level k) pair, a wave holds
// SLP = 64/LPS adjacent instances, vertical neighbours by DPP), the same march over the x
// columns and the same 3-column register pipeline -- but on a device layout that is PRIVATE to a
// plan.  The reference layout (sl fastest, :33-38) makes every (column,
// level) row of every array its own stream, of which a wave can use 16 bytes; the x-march kernel
// therefore needs a workgroup of 8 waves, a transposing LDS ring and a barrier per column to
// read 128-byte row segments.  Here the arrays are stored
//
//     f, u, w :  [tracer][tile][column c = i+2][instance-in-tile s][level k]     (k fastest)
//     rho, adz, rhow(1:nzm) :  [tile][3][s][k]         flux :  [tracer][tile][s][k]
//
// with tile = SLP adjacent instances = what ONE WAVE works on.  A column of a tile ("chunk",
// SLP*nzm elements) is contiguous, consecutive columns are adjacent: every wave reads three
// linear streams and writes one.  Consequences:
//   * waves are independent: no workgroup barrier, no transpose (the LDS image of a chunk IS
//     the lane order), no LDS tile for the write-back (a finished column is stored from
//     registers, SLP*nzm*8 contiguous bytes per wave instruction);
//   * columns arrive by LDS-DMA, one 16-byte-per-lane instruction per array and column PAIR
//     (lanes 0-31 the even column, 32-63 the odd one), into a per-wave ring of 3 pairs; one
//     counted s_waitcnt vmcnt per pair is the only synchronisation, two pairs stay in flight;
//   * the LDS ring only serves as a prefetch buffer that costs no VGPRs and as the source of
//     the raw inputs of the vertical neighbours (kb / kc clamps in the read address).
// mpdata_layout.hip converts between this layout and the reference's (upload / download /
// device import / export of a plan); the C-ABI contract stays the reference layout.
//
// u, w use the column index of f (c = i+2): u has no column c = 0, w none at c = 0 and
// c = nx+5; those chunks exist in memory but are never fetched.
//
// f is bit-identical to the reference in the EXACT build; flux as in the x-march kernel
// (sum of upwind terms + sum of limited terms, each in the reference's i order).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "mpdata_args.h"

namespace MPDATA_NS {
namespace wm {

// arithmetic helpers, DPP shifts and row stores of the x-march kernel
using v2::dmax; using v2::dmin; using v2::rabs; using v2::rldexp; using v2::andiff; using v2::across;
using v2::upwind; using v2::pp; using v2::pn; using v2::recip_nr; using v2::st_row; using v2::shift_dn;
using v2::shift_up; using v2::shift_dn_clamped; using v2::shift_up_clamped; using v2::Window;

#ifndef MPDWM_ST_AUX
#define MPDWM_ST_AUX 0
#endif
#ifndef MPDWM_LD_AUX
#define MPDWM_LD_AUX 0
#endif

template <typename R_, int LPS, int WPB_>
struct TileWm {
  using R = R_;
  static constexpr int SLP = 64 / LPS;          // instances per wave = per tile
  static constexpr int WPB = WPB_;              // waves (= tiles) per workgroup; they never synchronise
  static constexpr int THREADS = 64 * WPB_;
  static constexpr int NZM_MAX = LPS - 1;       // lane k = nz is the ghost level (w = 0)
  static constexpr int NS = 3;                  // ring: column pairs P .. P+2
  static constexpr int HALF = 64;               // elements between the two columns of a pair in LDS
  static constexpr int ARR = 128;               // elements of one array block of a slot (1 KiB)
  static constexpr int SLOT = 3 * ARR;          // f, u, w
  static constexpr int LDS_ELEMS = WPB_ * NS * SLOT;
  static constexpr int MIN_WAVES = 4;           // 128 VGPRs
  static_assert(sizeof(R_) == 8, "8-byte elements (double, or two fp32 instances per lane)");
};

// COLDMA: column-granular fetch (five columns in flight; best for one tracer, whose inputs come
// from HBM) instead of pair-granular fetch (fewer instructions; best for tracer batches, which
// are VALU-bound and read u, w from L2)
template <typename R, int LPS, int WPB, bool COLDMA>
__global__ void __launch_bounds__(64 * WPB, (TileWm<R, LPS, WPB>::MIN_WAVES))
mpdata_advect_wm_kernel(const MpdataWmArgsT<R> a) {
  using T = TileWm<R, LPS, WPB>;
  constexpr int SLP = T::SLP, RB = (int)sizeof(R);
  __shared__ R lds[T::LDS_ELEMS];

  const int nx = a.nx, nz = a.nz, nzm = nz - 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  R* const my = lds + wave * (T::NS * T::SLOT);

  // ---- wave -> (tile, tracer).  Workgroups are dealt to the 8 XCDs round-robin in dispatch
  //      order.  Tracer batches: the waves that an XCD receives walk through the tracers of one
  //      tile after the other, so the ntracers waves that share a tile's u, w, rho, ... run
  //      back to back on ONE XCD and all but the first find them in that XCD's L2.
  const unsigned ntr = (unsigned)a.ntracers;
  unsigned tile, tr;
  if (ntr == 1) {
    tile = blockIdx.x * WPB + wave;
    tr = 0;
  } else {
    const unsigned nxcd = 8;
    const unsigned v = (blockIdx.x / nxcd) * WPB + wave;  // position in the XCD's wave sequence
    tr = v % ntr;
    tile = (v / ntr) * nxcd + blockIdx.x % nxcd;
  }
  if (tile >= (unsigned)a.ntiles) return;  // (no barrier anywhere below)
  // serpentine: every other run of a plan walks the tiles from the other end, so that it starts
  // on what the previous run touched last (u, w and the like are still in the Infinity Cache)
  if (a.reverse) tile = (unsigned)a.ntiles - 1u - tile;

  const int chunk = SLP * nzm;                    // elements of one column of the tile
  const unsigned chunkB = (unsigned)(chunk * RB);
  const long long toff = (long long)tile * a.tile_elems;
  R* const f = a.f + (long long)tr * a.f_tstride + toff;
  const R* const u = a.u + toff;
  const R* const w = a.w + toff;
  const R* const kc = a.kc + (long long)tile * (3 * chunk);
  R* const flux = a.flux + (long long)tr * a.flux_tstride + (long long)tile * chunk;

  // ---- lane -> (instance s, level k) -----------------------------------------
  const int s_l = lane / LPS;
  const int kk = lane % LPS;               // k - 1
  const int k = kk + 1;
  const bool lvl_ok = k <= nzm;            // real level (else ghost / dead lane)
  const int kl = lvl_ok ? kk : nzm - 1;    // level index whose data the lane reads
  const int pos = s_l * nzm + kl;          // element of the chunk

  // per-lane constants (:552, :553, :565, :569): the loads go out here, the arithmetic on them
  // follows the DMA prologue below (a wave's first column fetch must not queue behind a
  // dependent load round trip)
  const R eps = (R)1.e-10f;  // :509, fp32 literal
  const R RHO = kc[pos];
  const R adz_l = kc[chunk + pos];
  const R rhow_l = kc[2 * chunk + pos];
  const int dd_exp = (k == 1 || k == nzm) ? 1 : 0;  // :569, the factor 2./(kc-kb) as an exponent step
  const bool k_is_1 = k == 1;
  const bool k_ge_nzm = k >= nzm;
  const unsigned long long own_dn = __builtin_amdgcn_ballot_w64(k_is_1);     // kb = k
  const unsigned long long own_up = __builtin_amdgcn_ballot_w64(k_ge_nzm);   // kc = k

  // ---- LDS read positions (elements, relative to `my`): slot, array and column-of-pair are
  //      compile-time offsets on top of these.  The raw inputs of the vertical neighbours are
  //      extra reads with the kb / kc clamps in the address.  Lanes above nzm (ghost level nz
  //      and dead lanes) read w = 0: element 63 of a column block is never written with data
  //      (chunk <= 62; LPS = 64 selects instead) and the ring is zeroed at kernel start.
  const R* const p_own = my + pos;
  const R* const p_dn = my + (s_l * nzm + (kl > 0 ? kl - 1 : 0));
  const R* const p_up = my + (s_l * nzm + (kl + 1 < nzm ? kl + 1 : nzm - 1));
  const R* const p_w = (LPS == 64 || lvl_ok) ? p_own : my + 63;

  // ---- global addressing: one descriptor per array, based at the wave's tile (32-bit offsets
  //      inside a tile, arrays of any size).
  const unsigned OOB = 0xFFFFFFF8u;
  const int ncol = nx + 6;
  const long long tileB = (long long)ncol * chunkB;
  const __amdgpu_buffer_rsrc_t rsf = v2::make_rsrc(f, tileB);
  const __amdgpu_buffer_rsrc_t rsu = v2::make_rsrc(u, tileB);
  const __amdgpu_buffer_rsrc_t rsw = v2::make_rsrc(w, tileB);
  const unsigned st_off = lvl_ok ? (unsigned)(pos * RB) : OOB;  // the lane's element of a chunk
  // DMA source offsets of a pair instruction: lane L < 32 fetches bytes 16L.. of the even column,
  // lane L >= 32 bytes 16(L-32).. of the odd one (the LDS image keeps the two columns 512 B apart)
  const unsigned in_col = (unsigned)((lane & 31) * 16);
  const bool in_ok = in_col < chunkB;
  const unsigned v_all = in_ok ? in_col + (lane >= 32 ? chunkB : 0u) : OOB;
  const unsigned v_odd = lane >= 32 ? v_all : OOB;   // only the odd column of the pair
  const unsigned v_even = lane < 32 ? v_all : OOB;   // only the even one

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  auto dma_issue = [&](const int P, const unsigned vf, const unsigned vu, const unsigned vw) __attribute__((always_inline)) {
    R* d = my + (P % T::NS) * T::SLOT;
    const unsigned so = (unsigned)(2 * P) * chunkB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)vf, (int)so, 0, MPDWM_LD_AUX);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::ARR), 16, (int)vu, (int)so, 0, MPDWM_LD_AUX);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + 2 * T::ARR), 16, (int)vw, (int)so, 0, MPDWM_LD_AUX);
  };
  // pair P of all three arrays into its ring slot (3 instructions).  General form: u has no
  // column c = 0, w none at c = 0 and c = nx+5, nothing exists past the last pair.
  auto dma_pair = [&](const int P) __attribute__((always_inline)) {
#ifdef MPDWM_ABL_NODMA  // timing ablation only
    return;
#endif
    const int c0 = 2 * P, c1 = 2 * P + 1;  // the pair's columns
    auto sel = [&](const bool e, const bool o) __attribute__((always_inline)) {
      return e ? (o ? v_all : v_even) : (o ? v_odd : OOB);
    };
    const unsigned vf = sel(c0 < ncol, c1 < ncol);
    const unsigned vu = sel(c0 >= 1 && c0 < ncol, c1 < ncol);
    const unsigned vw = sel(c0 >= 1 && c0 <= nx + 4, c1 <= nx + 4);
    dma_issue(P, vf, vu, vw);
  };
  // interior pairs (1 <= P, 2P+1 < nx+5): no conditions
  auto dma_pair_full = [&](const int P) __attribute__((always_inline)) {
#ifdef MPDWM_ABL_NODMA
    return;
#endif
    dma_issue(P, v_all, v_all, v_all);
  };

  // Column-granular fetch (default): column c of the three arrays by three instructions with
  // lanes 32-63 switched off (27 lanes x 16 bytes each at nz = 28), into the half of its pair's
  // slot.  Same LDS image as the pair fetch, but a column's slot is free again right after its
  // own step, so FIVE columns are in flight instead of four (more bytes in flight per CU for
  // the same LDS), at the price of one counted wait per step.
  const unsigned v_col = in_ok ? in_col : OOB;
  auto dma_col_issue = [&](const int c, const unsigned vf, const unsigned vu, const unsigned vw) __attribute__((always_inline)) {
    R* d = my + ((c >> 1) % T::NS) * T::SLOT + (c & 1) * T::HALF;
    const unsigned so = (unsigned)c * chunkB;
    if (lane < 32) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)vf, (int)so, 0, MPDWM_LD_AUX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::ARR), 16, (int)vu, (int)so, 0, MPDWM_LD_AUX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + 2 * T::ARR), 16, (int)vw, (int)so, 0, MPDWM_LD_AUX);
    }
  };
  auto dma_col = [&](const int c) __attribute__((always_inline)) {
#ifdef MPDWM_ABL_NODMA
    return;
#endif
    dma_col_issue(c, c < ncol ? v_col : OOB, (c >= 1 && c < ncol) ? v_col : OOB, (c >= 1 && c <= nx + 4) ? v_col : OOB);
  };
  auto dma_col_full = [&](const int c) __attribute__((always_inline)) {  // 1 <= c <= nx+4
#ifdef MPDWM_ABL_NODMA
    return;
#endif
    dma_col_issue(c, v_col, v_col, v_col);
  };

  // ---- prologue: zero the ring (the never-fetched tails of the column blocks supply w = 0 of
  //      the ghost level), then pairs 0 and 1 into flight, each behind two dropped stores so
  //      that the counted wait of the first pairs sees the steady-state op pattern
#pragma unroll
  for (int j = 0; j < T::NS * T::SLOT / 64; ++j) my[j * 64 + lane] = R(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!COLDMA) {
#pragma unroll
    for (int P = 0; P < T::NS - 1; ++P) {
      st_row(rsf, OOB, 0, R(0));
      st_row(rsf, OOB, 0, R(0));
      dma_pair(P);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 2 * T::NS; ++c) {  // columns 0 .. 5, each behind a dropped store
      st_row(rsf, OOB, 0, R(0));
      dma_col(c);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const R IRHO = R(1) / RHO;
  const R IADZ = R(1) / adz_l;
  const R IRHOW = R(1) / (rhow_l * adz_l);
#ifdef MPDATA_FAST_DIV
  const R KU = rldexp(R(0.03125) * IRHO * IADZ, dd_exp);
  const R KW = (k == 1) ? R(0) : R(0.03125) * IRHO;  // www(:,:,:,1) = 0 (:586)
#endif
  Window<R> S;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S.F0[j] = S.PMX[j] = S.PMN[j] = S.U1[j] = S.DW1[j] = R(0);
    S.F1[j] = S.F1D[j] = S.F1U[j] = S.MX0[j] = S.MN0[j] = R(0);
    S.UR[j] = S.UD[j] = S.PW[j] = S.SW[j] = S.WR[j] = S.SU[j] = S.U2P[j] = S.U2N[j] = R(0);
    S.MXN[j] = S.MNN[j] = S.U3[j] = S.DW3[j] = R(0);
  }
  R S1 = R(0), S3 = R(0);


  // One column step.  PH = c mod 3 selects the register ring slots, SL = (c/2) mod 3 the LDS
  // ring slot, H = c mod 2 the column of the pair (c = q+2); FULL = steady state (4 <= q <= nx).
  auto step = [&](auto ph_tag, auto sl_tag, auto h_tag, auto full_tag, const int q) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr int LO = decltype(sl_tag)::value * T::SLOT + decltype(h_tag)::value * T::HALF;  // LDS offset
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int C0 = PH, C1 = (PH + 2) % 3, C2 = (PH + 1) % 3, C3 = PH;  // slots of q, q-1, q-2, q-3

    const R f0q = p_own[LO];
    const R uq = p_own[LO + T::ARR];
    R wq = p_w[LO + 2 * T::ARR];  // ghost level: w = 0
    if constexpr (LPS == 64) wq = lvl_ok ? wq : R(0);

#ifdef MPDWM_ABL_NOCOMPUTE  // timing ablation only: data movement without the arithmetic
    st_row(rsf, (q - 3 >= -1 && q - 3 <= nx + 2) ? st_off : OOB, (unsigned)max(q - 1, 0) * chunkB, f0q + uq + wq);
    return;
#endif
#define DN_C(x) shift_dn_clamped((x), own_dn)
#define UP_C(x) shift_up_clamped((x), own_up)
#define UP_G(x) shift_up(x)
    const R f0d = p_dn[LO];
    const R f0u = p_up[LO];
    const R F0p = S.F0[C1];

    // ================= stage A =================================================
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
        // the last two halo columns (nx+1, nx+2) keep this first-pass value (:557) and no later
        // step finishes them: store them now (their input values are already in the LDS ring)
        if (!FULL && q - 1 >= nx + 1) st_row(rsf, q - 1 <= nx + 2 ? st_off : OOB, (unsigned)(q + 1) * chunkB, f1_1);
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
    const R ud = p_dn[LO + T::ARR];
    const R wu = p_up[LO + 2 * T::ARR];
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
        const R u1 = S.UR[C1];
        const R t1 = rabs(u1) - (u1 * u1) * IRHO;
        const R x4 = S.F1U[C2] + F1U_1 - S.F1D[C2] - F1D_1;
        U2_1 = R(0.5) * (t1 * (f1_1 - S.F1[C2])) - KU * ((u1 * S.SW[C1]) * x4);
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
          W2_2 = R(0.5) * (t1 * (S.F1[C2] - S.F1D[C2])) - KW * ((w2 * S.SU[C2]) * x4);  // k = 1: 0
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
        const R num_mx = RHO * (mx1 - S.F1[C2]);
        const R den_mx = U2n_1 + S.U2P[C2] + IADZ * (pn(W2u) + W2p) + eps;
        const R num_mn = RHO * (S.F1[C2] - mn1);
        const R den_mn = U2p_1 + S.U2N[C2] + IADZ * (pp(W2u) + W2n) + eps;
#ifdef MPDATA_FAST_DIV
        {  // one reciprocal for both ratios (both denominators >= eps > 0, finite)
          const R dd2 = den_mx * den_mn;
          const R r = recip_nr(dd2);
          MXN_2 = num_mx * (den_mn * r);
          MNN_2 = num_mn * (den_mx * r);
        }
#else
        MXN_2 = num_mx / den_mx;
        MNN_2 = num_mn / den_mn;
#endif
        // the ratios are only ever used as min(1, ratio, ...) (:618, :623): keep them clamped
        MXN_2 = dmin(R(1), MXN_2);
        MNN_2 = dmin(R(1), MNN_2);
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
        S3 = S3 + W3;  // :624
        DW3_2 = UP_G(W3) - W3;
      }
    }
    {
      const int n = q - 3;  // column finished in this step: stored straight from the register
      const bool act = n >= -1 && n <= nx + 2;
      R v = S.F1[C3];  // halo columns keep the first-pass value (:557)
      if (FULL || (n >= 1 && n <= nx))
        v = dmax(R(0), S.F1[C3] - ((U3_2 - S.U3[C3]) + S.DW3[C3] * IADZ) * IRHO);  // :634
      st_row(rsf, (FULL || act) ? st_off : OOB, (unsigned)max(n + 2, 0) * chunkB, v);
    }
    S.U3[C2] = U3_2;
    S.DW3[C2] = DW3_2;
  };
#undef DN_C
#undef UP_C
#undef UP_G

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using Full = std::true_type;
  using Part = std::false_type;

  // Six columns = three pairs per trip, so that the register-ring phase (c mod 3), the LDS slot
  // ((c/2) mod 3) and the column of the pair (c mod 2) are compile-time constants.
  //   fill (q = -2 .. 3) and drain trips: wave-uniform conditions; steps beyond q = nx+3 do nothing;
  //   steady-state trips (4 <= q, q+5 <= nx): every stage active, no conditions.
  const int q_last = nx + 3;
  // pair-granular: wait for the pair (2 stores + 3 DMA were issued since its own DMA), two
  // column steps, fetch pair P+2 into the slot of pair P-1.
  // column-granular: wait for the column (5 x (1 store + 3 DMA) since its own DMA), the step,
  // fetch column c+6 into the place of column c.
  auto dma_p = [&](const int q) __attribute__((always_inline)) {   // q: the column just done
    if constexpr (COLDMA) dma_col(q + 8);
    else dma_pair(((q + 1) >> 1) + 2);
  };
  auto dma_f = [&](const int q) __attribute__((always_inline)) {
    if constexpr (COLDMA) {
      if (q + 8 <= nx + 4) dma_col_full(q + 8); else dma_col(q + 8);
    } else {
      const int P = ((q + 1) >> 1) + 2;
      if (2 * P + 1 < nx + 5) dma_pair_full(P); else dma_pair(P);
    }
  };
#define MPDWM_PAIR(PHA, PHB, SL, TAG, q, DMA)                          \
  if constexpr (COLDMA) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); \
  else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                 \
  step(PHA{}, SL{}, I0{}, TAG{}, (q));                                 \
  asm volatile("" ::: "memory");                                       \
  if constexpr (COLDMA) {                                              \
    DMA(q);                                                            \
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");                  \
  }                                                                    \
  step(PHB{}, SL{}, I1{}, TAG{}, (q) + 1);                             \
  asm volatile("" ::: "memory");                                       \
  DMA((q) + 1);
  int q0 = -2;
  {  // fill: columns -2 .. 3 (nx >= 1: all of them exist)
    MPDWM_PAIR(I0, I1, I0, Part, q0, dma_p)
    MPDWM_PAIR(I2, I0, I1, Part, q0 + 2, dma_p)
    MPDWM_PAIR(I1, I2, I2, Part, q0 + 4, dma_p)
    q0 += 6;
  }
  for (; q0 + 5 <= nx; q0 += 6) {  // steady state
    MPDWM_PAIR(I0, I1, I0, Full, q0, dma_f)
    MPDWM_PAIR(I2, I0, I1, Full, q0 + 2, dma_f)
    MPDWM_PAIR(I1, I2, I2, Full, q0 + 4, dma_f)
  }
  for (; q0 <= q_last; q0 += 6) {  // drain
    MPDWM_PAIR(I0, I1, I0, Part, q0, dma_p)
    MPDWM_PAIR(I2, I0, I1, Part, q0 + 2, dma_p)
    MPDWM_PAIR(I1, I2, I2, Part, q0 + 4, dma_p)
  }
#undef MPDWM_PAIR

  if (lvl_ok) flux[pos] = S1 + S3;  // :541-547, :624
}

}  // namespace wm
}  // namespace MPDATA_NS
