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
// SLP*nzm elements, 432 bytes at nz = 28) is split into its whole 128-byte lines (the "main" part,
// 384 bytes) and the rest (48 bytes); a tile stores the main parts of all its columns first, then
// the remainders:   [tile][ main: column][..]  [ remainder: column][..] , tiles at an odd number of
// 128-byte lines.  Every line of a main part is then touched by exactly ONE fetch and ONE store
// instruction, which is what makes non-temporal (streaming) accesses pay: +20 % in the bare data
// movement (tools/wave_stream.hip: 6.4 TB/s against 5.3 TB/s for contiguous 432-byte chunks;
// with a line shared by two instructions the streaming hint evicts it in between).  The
// remainders -- one line serves 2.7 columns -- are fetched and stored with the default policy.
// Every wave reads three linear streams (+ three small ones) and writes one.  Consequences:
//   * waves are independent: no workgroup barrier, no transpose (the LDS image of a chunk IS
//     the lane order), no LDS tile for the write-back (a finished column is stored from
//     registers, SLP*nzm*8 contiguous bytes per wave instruction);
//   * columns arrive by LDS-DMA, two 16-byte-per-lane instructions per array and column PAIR
//     (main parts, remainders; lanes 0-31 the even column, 32-63 the odd one), into a per-wave
//     ring of 3 pairs; one counted s_waitcnt vmcnt per pair is the only synchronisation; a pair's
//     slot is refilled as soon as its two columns have been read (nobody else reads it: the ring
//     is private to the wave), so three pairs are in flight while one is being worked on;
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
using v2::upwind; using v2::pp; using v2::pn; using v2::recip_nr; using v2::shift_dn;
using v2::shift_up; using v2::shift_dn_clamped; using v2::shift_up_clamped; using v2::Window;

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
  // 128 VGPRs; one instance per wave (LPS = 64) carries the ghost-level select of w and gets 168
  static constexpr int MIN_WAVES = LPS == 64 ? 3 : 4;
  static_assert(sizeof(R_) == 8, "8-byte elements (double, or two fp32 instances per lane)");
};

// STREAM (one tracer per launch: every byte is used once and the kernel is bound by its data
// movement): main parts fetched and stored with the streaming hint, remainders with the default
// policy -- two instructions per array and pair, two per column store.  !STREAM (tracer batches:
// VALU-bound, u and w are re-read from L2 by the other tracers of the tile): ONE instruction per
// array and pair / per column store, default policy, every lane addressing its own part.
template <typename R, int LPS, int WPB, bool STREAM>
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
  //      and dead lanes) read zeros for f, u and w -- element 63 of a column block is never
  //      written with data (chunk <= 62; LPS = 64 selects w instead) and the ring is zeroed at
  //      kernel start: with w = 0 the ghost level's fluxes are exact zeros, which is all the
  //      level below ever takes from it (www(:,:,:,nz) = 0, :511).
  const R* const p_own = my + ((LPS == 64 || lvl_ok) ? pos : 63);
  const R* const p_dn = my + (s_l * nzm + (kl > 0 ? kl - 1 : 0));
  const R* const p_up = my + (s_l * nzm + (kl + 1 < nzm ? kl + 1 : nzm - 1));

  // ---- global addressing: one descriptor per array, based at the wave's tile (32-bit offsets
  //      inside a tile, arrays of any size).  Element e of column c of the tile lives at
  //      c*mainB + 8e (e in the main part) or remBase + c*remB + (8e - mainB).
  const unsigned OOB = 0xFFFFFFF8u;
  const int ncol = nx + 6;
  const unsigned mainB = chunkB / 128u * 128u, remB = chunkB - mainB;
  const unsigned remBase = (unsigned)ncol * mainB;
  const long long tileB = (long long)ncol * chunkB;
  const __amdgpu_buffer_rsrc_t rsf = v2::make_rsrc(f, tileB);
  const __amdgpu_buffer_rsrc_t rsu = v2::make_rsrc(u, tileB);
  const __amdgpu_buffer_rsrc_t rsw = v2::make_rsrc(w, tileB);
  const unsigned posB = (unsigned)(pos * RB);
  const unsigned st_main = (lvl_ok && posB < mainB) ? posB : OOB;               // the lane's element of a chunk:
  const unsigned st_rem = (lvl_ok && posB >= mainB) ? remBase + (posB - mainB) : OOB;  // in one of the two parts
  // DMA source offsets of a pair instruction: lane L < 32 fetches bytes 16L.. of the even column,
  // lane L >= 32 bytes 16(L-32).. of the odd one (the LDS image keeps the two columns 512 B apart);
  // the lanes of a column's main part in the one instruction, those of its remainder in the other
  const unsigned in_col = (unsigned)((lane & 31) * 16);
  const unsigned hi = lane >= 32 ? 1u : 0u;
  const bool lane_main = in_col < mainB;
  const unsigned vA = lane_main ? in_col + hi * mainB : OOB;
  const unsigned vB = (in_col >= mainB && in_col < chunkB) ? remBase + (in_col - mainB) + hi * remB : OOB;
  // only the even / only the odd column of a pair: the other half of the lanes out of range
  auto halves = [&](const unsigned v, const bool e, const bool o) __attribute__((always_inline)) {
    if (e && o) return v;
    if (!e && !o) return OOB;
    unsigned r;
    const unsigned long long m = e ? 0x00000000FFFFFFFFull : 0xFFFFFFFF00000000ull;
    asm("v_cndmask_b32 %0, -8, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
  };
  constexpr int AUX_NT = 2;                      // streaming (non-temporal) cache policy
  // !STREAM: one instruction serves both parts; a lane's offset advances by its part's pair stride
  // (running per-lane offsets: the DMA offset advances by the lane's part's pair stride with every
  //  pair issued -- pairs are issued in order --, the store offset by its column stride with every
  //  column step; lanes that own nothing keep the out-of-range marker: stride 0)
  unsigned dcur = lane_main ? vA : vB;
  // the two per-lane strides share one register: pair stride of the lane's DMA part in the low
  // half, column stride of its store part in the high half (both <= 1024); a stride is added with
  // the half-word select of the add itself (SDWA), so unpacking costs no instruction
  const unsigned pstride = dcur == OOB ? 0u : (lane_main ? 2u * mainB : 2u * remB);
  const unsigned cstride = lvl_ok ? (posB < mainB ? mainB : remB) : 0u;
  const unsigned strides = pstride | (cstride << 16);
  auto add_lo = [](const unsigned a, const unsigned packed) __attribute__((always_inline)) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(packed));
    return r;
  };
  auto add_hi = [](const unsigned a, const unsigned packed) __attribute__((always_inline)) {
    unsigned r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(packed));
    return r;
  };
  // store offset of column c = q - 1 of the step about to run; the march starts at q = -2, behind one
  // empty flush of the deferred-store slot, which advances the offset as well
  unsigned scur = lvl_ok ? (posB < mainB ? posB : remBase + (posB - mainB)) - 4u * cstride : OOB;

  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  // pair P of all three arrays into its ring slot.  e*/o*: which columns of the pair exist for
  // the array (u has no column c = 0, w none at c = 0 and c = nx+5, nothing exists past the last
  // column).
  auto dma_issue = [&](const int P, const bool ef, const bool of, const bool eu, const bool ou, const bool ew,
                       const bool ow) __attribute__((always_inline)) {
#ifdef MPDWM_ABL_NODMA  // timing ablation only
    return;
#endif
    R* d = my + (P % T::NS) * T::SLOT;
    if constexpr (STREAM) {
      const unsigned soA = (unsigned)(2 * P) * mainB, soB = (unsigned)(2 * P) * remB;
      // A lane that is out of range in an LDS-DMA instruction still writes (zeros) to its LDS
      // position, so the two instructions of an array must not both be executed by a lane: the
      // main-part fetch runs with the main lanes only, the remainder fetch with the others.
      if (lane_main) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)halves(vA, ef, of), (int)soA, 0, AUX_NT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::ARR), 16, (int)halves(vA, eu, ou), (int)soA, 0, AUX_NT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + 2 * T::ARR), 16, (int)halves(vA, ew, ow), (int)soA, 0, AUX_NT);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)halves(vB, ef, of), (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::ARR), 16, (int)halves(vB, eu, ou), (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + 2 * T::ARR), 16, (int)halves(vB, ew, ow), (int)soB, 0, 0);
      }
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)halves(dcur, ef, of), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::ARR), 16, (int)halves(dcur, eu, ou), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + 2 * T::ARR), 16, (int)halves(dcur, ew, ow), 0, 0, 0);
      dcur = add_lo(dcur, strides);   // (pairs are issued in order, P = 0, 1, 2, ...)
    }
  };
  auto dma_pair = [&](const int P) __attribute__((always_inline)) {
    const int c0 = 2 * P, c1 = 2 * P + 1;  // the pair's columns
    dma_issue(P, c0 < ncol, c1 < ncol, c0 >= 1 && c0 < ncol, c1 < ncol, c0 >= 1 && c0 <= nx + 4, c1 <= nx + 4);
  };
  // interior pairs (1 <= P, 2P+1 <= nx+4): no conditions
  auto dma_pair_full = [&](const int P) __attribute__((always_inline)) {
    dma_issue(P, true, true, true, true, true, true);
  };
  // column c of f back to memory.  STREAM: main part with the streaming hint, remainder without
  // (two instructions); else one instruction, every lane to its part.
  // AHEAD = 0: the step's regular store (column c = q - 1; advances the running offset);
  // AHEAD = 2: the early store of a halo column c = q + 1.
  auto st_col = [&](const bool act, const int c, const R v, auto ahead_tag) __attribute__((always_inline)) {
    constexpr int AHEAD = decltype(ahead_tag)::value;
#ifdef MPD2_ABL_NOMEM  // timing ablation only (wrong results)
    if (__builtin_bit_cast(double, v) != 1.2345e300) return;
#endif
    const v2::u32x2 b = __builtin_bit_cast(v2::u32x2, v);
    if constexpr (STREAM) {
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)(act ? st_main : OOB), (int)((unsigned)c * mainB), AUX_NT);
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)(act ? st_rem : OOB), (int)((unsigned)c * remB), 0);
    } else {
      // (c is implied by the running offset)
      unsigned o = scur;
      if (AHEAD == 2) o = add_hi(add_hi(scur, strides), strides);
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)(act ? o : OOB), 0, 0);
      if (AHEAD == 0) scur = add_hi(scur, strides);
    }
  };

  // ---- prologue: zero the ring (the never-fetched tails of the column blocks supply w = 0 of
  //      the ghost level), then pairs 0, 1, 2 into flight, each behind two dropped column stores so
  //      that the counted wait of the first pairs sees the steady-state op pattern
#pragma unroll
#ifdef MPDWM_ABL_NODMA  // timing ablation: arithmetic on (non-zero, finite) stand-in data
  for (int j = 0; j < T::NS * T::SLOT / 64; ++j) my[j * 64 + lane] = R(0.25) + R(0.001) * R((j * 64 + lane) % 97) - R(0.3) * R(lane & 1);
#else
  for (int j = 0; j < T::NS * T::SLOT / 64; ++j) my[j * 64 + lane] = R(0);
#endif
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int P = 0; P < T::NS; ++P) {
    st_col(false, 0, R(0), std::integral_constant<int, 1>{});
    st_col(false, 0, R(0), std::integral_constant<int, 1>{});
    dma_pair(P);
  }
  __builtin_amdgcn_sched_barrier(0);
  const R IRHO = R(1) / RHO;
  const R IADZ = R(1) / adz_l;
  const R IRHOW = R(1) / (rhow_l * adz_l);
#ifdef MPDATA_FAST_DIV
  // FAST carries the second-pass fluxes DOUBLED (andiff's factor 0.5 is not applied: one
  // multiplication fewer per flux) and the limiter ratios HALVED (doubled denominators, clamp
  // at 0.5), so the limited fluxes come out unscaled; every scaling is by a power of two, i.e.
  // the results are bit-identical to the unscaled evaluation
  const R EPS_D = eps + eps, LIM = R(0.5);
  const R KU = rldexp(R(0.03125) * IRHO * IADZ, dd_exp + 1);
  const R KW = (k == 1) ? R(0) : R(0.0625) * IRHO;  // www(:,:,:,1) = 0 (:586)
#else
  const R EPS_D = eps, LIM = R(1);
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
  R v_def = R(0);   // the deferred store of a pair's odd column
  bool act_def = false;
  int c_def = 0;


  // One column step.  PH = c mod 3 selects the register ring slots, SL = (c/2) mod 3 the LDS
  // ring slot, H = c mod 2 the column of the pair (c = q+2); FULL = steady state (4 <= q <= nx).
  auto step = [&](auto ph_tag, auto sl_tag, auto h_tag, auto full_tag, const int q) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr int LO = decltype(sl_tag)::value * T::SLOT + decltype(h_tag)::value * T::HALF;  // LDS offset
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int C0 = PH, C1 = (PH + 2) % 3, C2 = (PH + 1) % 3, C3 = PH;  // slots of q, q-1, q-2, q-3

    const R f0q = p_own[LO];
    const R uq = p_own[LO + T::ARR];
    R wq = p_own[LO + 2 * T::ARR];  // ghost level: w = 0
    if constexpr (LPS == 64) wq = lvl_ok ? wq : R(0);

#ifdef MPDWM_ABL_NOCOMPUTE  // timing ablation only: data movement without the arithmetic
    st_col(q - 3 >= -1 && q - 3 <= nx + 2, max(q - 1, 0), f0q + uq + wq, std::integral_constant<int, 0>{});
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
        if (!FULL && q - 1 >= nx + 1) st_col(q - 1 <= nx + 2, q + 1, f1_1, std::integral_constant<int, 2>{});
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
        U2_1 = t1 * (f1_1 - S.F1[C2]) - KU * ((u1 * S.SW[C1]) * x4);   // = 2 x (:571-573)
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
          W2_2 = t1 * (S.F1[C2] - S.F1D[C2]) - KW * ((w2 * S.SU[C2]) * x4);  // = 2 x (:580-582); k = 1: 0
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
        {  // one reciprocal for both ratios (both denominators >= eps > 0, finite)
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
      // the odd column's store waits until the next pair has been waited for: the counted wait
      // must see every store issued after the pair's DMA complete, and a store issued right
      // before it would stall it for a store round trip (+1 % one tracer, +3.5 % tracer batches)
      if constexpr (decltype(h_tag)::value == 1) {
        v_def = v; act_def = FULL || act; c_def = max(n + 2, 0);
      } else {
        st_col(FULL || act, max(n + 2, 0), v, std::integral_constant<int, 0>{});
      }
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
  // per pair: wait for it (two newer pairs: STREAM 2 x 6 DMA instructions, else 2 x 3), two column steps,
  // fetch pair P+3 into the slot just read
  auto dma_p = [&](const int q) __attribute__((always_inline)) { dma_pair(((q + 1) >> 1) + 3); };  // q: odd column of the pair
  auto dma_f = [&](const int q) __attribute__((always_inline)) {
    const int P = ((q + 1) >> 1) + 3;
    if (2 * P + 1 <= nx + 4) dma_pair_full(P); else dma_pair(P);
  };
// The counted waits only rely on LOADS returning in issue order: "at most as many operations
// outstanding as DMA instructions were issued after the pair's own".  (Counting the column stores
// issued in between as well -- vmcnt(10) / vmcnt(5) -- is a race: a store may be acknowledged
// before an older load has landed, and the count then drops below the mark with a DMA of the
// pair still in flight; seen as sporadic wrong columns with default-policy stores.)
#define MPDWM_FLUSH_DEFERRED st_col(act_def, c_def, v_def, std::integral_constant<int, 0>{});
#define MPDWM_PAIR(PHA, PHB, SL, TAG, q, DMA)           \
  if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); \
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  \
  MPDWM_FLUSH_DEFERRED                                  \
  step(PHA{}, SL{}, I0{}, TAG{}, (q));                  \
  asm volatile("" ::: "memory");                        \
  step(PHB{}, SL{}, I1{}, TAG{}, (q) + 1);              \
  asm volatile("" ::: "memory");                        \
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
#undef MPDWM_FLUSH_DEFERRED

  st_col(act_def, c_def, v_def, std::integral_constant<int, 0>{});
  if (lvl_ok) flux[pos] = S1 + S3;  // :541-547, :624
}

}  // namespace wm
}  // namespace MPDATA_NS
