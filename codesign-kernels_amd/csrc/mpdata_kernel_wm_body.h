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
// f is bit-identical to the reference in the EXACT build; flux as in the x-march kernel: EXACT
// parks the limited fluxes ([tracer][tile][column][lane], flux_finish_kernel adds them in the
// reference's order: bit-identical), FAST accumulates sum of upwind terms + sum of limited terms.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "mpdata_args.h"

namespace MPDATA_NS {
namespace wm {

// arithmetic helpers, DPP shifts and row stores of the x-march kernel
using v2::dmax; using v2::dmin; using v2::rabs; using v2::rldexp; using v2::andiff; using v2::across;
using v2::upwind; using v2::pp; using v2::pn; using v2::recip_nr; using v2::recip_const; using v2::shift_dn;
using v2::shift_up; using v2::shift_dn_clamped; using v2::shift_up_clamped; using v2::Window;

// Two tracers per wave (tracer batches): the tracer-dependent quantities of the column step are
// PAIRS, one member per tracer, and every statement on them expands member by member -- the two
// tracers' (independent) instruction chains come out interleaved, which is what keeps the fp64
// pipe busy with only two waves per SIMD.  u, w and everything formed from them alone stay
// scalars of type R: fetched / computed once for both tracers.
template <typename R>
struct Pair {
  R a, b;
  __device__ __forceinline__ Pair() {}
  __device__ __forceinline__ explicit Pair(R x) : a(x), b(x) {}
  __device__ __forceinline__ Pair(R x, R y) : a(x), b(y) {}
};
#define MPDWM_PAIR_OP(OP)                                                                                                  \
  template <typename R> __device__ __forceinline__ Pair<R> operator OP(Pair<R> x, Pair<R> y) { return {x.a OP y.a, x.b OP y.b}; } \
  template <typename R> __device__ __forceinline__ Pair<R> operator OP(Pair<R> x, R y) { return {x.a OP y, x.b OP y}; }           \
  template <typename R> __device__ __forceinline__ Pair<R> operator OP(R x, Pair<R> y) { return {x OP y.a, x OP y.b}; }
MPDWM_PAIR_OP(+) MPDWM_PAIR_OP(-) MPDWM_PAIR_OP(*) MPDWM_PAIR_OP(/)
#undef MPDWM_PAIR_OP
template <typename R> __device__ __forceinline__ Pair<R> operator-(Pair<R> x) { return {-x.a, -x.b}; }
template <typename R> __device__ __forceinline__ Pair<R> dmax(Pair<R> x, Pair<R> y) { return {dmax(x.a, y.a), dmax(x.b, y.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> dmin(Pair<R> x, Pair<R> y) { return {dmin(x.a, y.a), dmin(x.b, y.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> dmax(R x, Pair<R> y) { return {dmax(x, y.a), dmax(x, y.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> dmin(R x, Pair<R> y) { return {dmin(x, y.a), dmin(x, y.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> rldexp(Pair<R> x, int e) { return {rldexp(x.a, e), rldexp(x.b, e)}; }
template <typename R> __device__ __forceinline__ Pair<R> recip_nr(Pair<R> x) { return {recip_nr(x.a), recip_nr(x.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> shift_dn(Pair<R> x) { return {shift_dn(x.a), shift_dn(x.b)}; }
template <typename R> __device__ __forceinline__ Pair<R> shift_up(Pair<R> x) { return {shift_up(x.a), shift_up(x.b)}; }
// clamped shifts of a pair: ONE mask move and ONE hazard wait (see v2::shift_dn_clamped)
// for the four 32-bit select-moves
#define MPDWM_CNDMASK_DPP_PAIR(CTRL)                                                             \
  const v2::u32x2 xa = __builtin_bit_cast(v2::u32x2, x.a), xb = __builtin_bit_cast(v2::u32x2, x.b); \
  v2::u32x2 ra, rb;                                                                              \
  asm("s_mov_b64 vcc, %8\n\t" MPD_DPP_NOP "\n\t"                                                        \
      "v_cndmask_b32_dpp %0, %4, %4, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"   \
      "v_cndmask_b32_dpp %1, %5, %5, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"   \
      "v_cndmask_b32_dpp %2, %6, %6, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"   \
      "v_cndmask_b32_dpp %3, %7, %7, vcc " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:0"        \
      : "=&v"(ra.x), "=&v"(ra.y), "=&v"(rb.x), "=&v"(rb.y)                                       \
      : "v"(xa.x), "v"(xa.y), "v"(xb.x), "v"(xb.y), "s"(own)                                     \
      : "vcc");                                                                                  \
  return {__builtin_bit_cast(R, ra), __builtin_bit_cast(R, rb)};
template <typename R> __device__ __forceinline__ Pair<R> shift_dn_clamped(Pair<R> x, unsigned long long own) {
  static_assert(sizeof(R) == 8, "8-byte elements");
  MPDWM_CNDMASK_DPP_PAIR("wave_shr:1")
}
template <typename R> __device__ __forceinline__ Pair<R> shift_up_clamped(Pair<R> x, unsigned long long own) {
  static_assert(sizeof(R) == 8, "8-byte elements");
  MPDWM_CNDMASK_DPP_PAIR("wave_shl:1")
}
#undef MPDWM_CNDMASK_DPP_PAIR
// upwind flux with a shared velocity (:532, :537): a * (a >= 0 ? x : y) per tracer
template <typename R> __device__ __forceinline__ Pair<R> upwind(R a, Pair<R> x, Pair<R> y) {
  return {a * v2::sel_ge0(a, x.a, y.a), a * v2::sel_ge0(a, x.b, y.b)};
}
// statement functions :500-501 with tracer-independent velocity arguments (same operation order
// as v2::andiff / v2::across; the velocity factors are common to both tracers)
template <typename X, typename R> __device__ __forceinline__ X andiff_s(X x1, X x2, R a, R b) {
  return ((rabs(a) - a * a * b) * R(0.5)) * (x2 - x1);
}
template <typename X, typename R> __device__ __forceinline__ X across_s(X x1, R a1, R a2) {
  return ((R(0.03125) * a1) * a2) * x1;
}
template <typename R> __device__ __forceinline__ R sel(bool c, R x, R y) { return c ? x : y; }
template <typename R> __device__ __forceinline__ Pair<R> sel(bool c, Pair<R> x, Pair<R> y) { return {c ? x.a : y.a, c ? x.b : y.b}; }
template <typename R> __device__ __forceinline__ R first(R x) { return x; }
template <typename R> __device__ __forceinline__ R second(R x) { return x; }
template <typename R> __device__ __forceinline__ R first(Pair<R> x) { return x.a; }
template <typename R> __device__ __forceinline__ R second(Pair<R> x) { return x.b; }

// UWREF_ (mpdata_plan_run_uw): u and w are NOT in the plan layout but in the caller's reference layout
// (sl fastest).  A workgroup is then the 16/SLP waves whose tiles are 16 ADJACENT instances: the
// 128-byte row segments (16 instances of one (column, level)) of u and w arrive by 16-byte LDS-DMA in
// a ring of column pairs that the whole workgroup shares -- one barrier per column PAIR --, f still
// streams per wave in the plan layout.
template <typename R_, int LPS, int WPB_, int TPW_ = 1, bool UWREF_ = false>
struct TileWm {
  using R = R_;
  static constexpr bool XU = UWREF_;
  static constexpr int GX = 16;                 // UWREF: instances per workgroup (128-byte rows)
  static constexpr int XRG = LPS / 8;           // UWREF: groups of 8 rows (one DMA instruction) per array block
  static constexpr int XARR = LPS * GX;         // UWREF: elements of one (column, array) block: LPS rows x 16
  static constexpr int XSLOT = 4 * XARR;        // UWREF: a pair slot of the shared ring: [column of pair][u, w]
  // LPS = 128 = "KS", nz > 64 (round 5): an instance is wider than a wave, so a.nkw = 1 + ceil((nz - 64) / 58)
  // waves share it.  Wave h works on the 64 levels koff + 1 .. koff + 64 (koff = 58 h; the last wave: the even
  // number >= nz - 64), lanes along the levels as ever, and nothing crosses between the waves: the routine's
  // dependency cone has radius 3 in k, so a wave's results are right from its 4th level to
  // its 61st (its 1st if that is level 1, its last real level if the wave holds level nzm), and those ranges
  // tile the column: 58 h + 4 .. 58 h + 61.  The other six levels are computed twice -- 5 % of the lanes at
  // nz = 72 ... 122 instead of a barrier and an LDS exchange at each of the four places of a column step where a
  // level takes something from its neighbour.  The layout is the one-instance-per-tile layout (chunk = nzm elements).
  // LPS = 128 + 16 / 128 + 32 = "KT", the LAST window of an instance when it needs no more than 16 / 32 levels (nz <= 74 /
  // 90 with two windows): a wave then holds the last windows of 4 / 2 ADJACENT instances (= tiles), 16 / 32 lanes each,
  // window start 58 (nkw - 1) -- 80 / 96 lanes per instance instead of 128 (mpdata_advect_wm_ks2_kernel).
  static constexpr bool KS = LPS >= 128;
  static constexpr bool KT = LPS > 128;
  static constexpr int LW = KT ? LPS - 128 : (KS ? 64 : LPS);      // lanes of a wave per instance
  static constexpr int SLP = KS ? 1 : 64 / LW;  // instances per tile (the layout; = per wave except KT)
  static constexpr int TPV = KT ? 64 / LW : 1;  // tiles per wave
  static constexpr int WPB = WPB_;              // waves (= tiles) per workgroup; they never synchronise
  static constexpr int THREADS = 64 * WPB_;
  static constexpr int NZM_MAX = LPS - 1;       // lane k = nz is the ghost level (w = 0)
  static constexpr int NS = 3;                  // ring: column pairs P .. P+2
  static constexpr int HALF = 64;               // elements between the two columns of a pair in LDS
  static constexpr int ARR = 128;               // elements of one array block of a slot (1 KiB)
  static constexpr int TPW = TPW_;              // tracers per wave (tracer batches: 1 or 2)
  static constexpr int UO = TPW_ * ARR, WO = (TPW_ + 1) * ARR;   // a slot: f of every tracer of the wave, u, w
  static constexpr int SLOT = UWREF_ ? ARR : (2 + TPW_) * ARR;   // (UWREF: the wave's own ring holds f only)
  static constexpr int LDS_ELEMS = WPB_ * NS * SLOT + (UWREF_ ? NS * XSLOT : 0);
  static_assert(!UWREF_ || (!KS && WPB_ * (64 / LW) == GX && TPW_ == 1), "UWREF: 16 instances per workgroup, one tracer per wave");
  // 128 VGPRs; one instance per wave (LPS = 64) carries the ghost-level select of w and gets 168
  // (two tracers per wave: twice the tracer state, 256 VGPRs, 2 waves per SIMD)
  // UWREF: 16 / SLP waves per workgroup.  FAST fits 128 VGPRs (two workgroups of 8 waves per CU at
  // LPS = 32); EXACT (the IEEE divisions, no folded constants) does not: it gets 256 and runs one
  // workgroup per CU -- it is the parity variant of this kernel, not the fast one.
#ifdef MPDATA_FAST_DIV
  static constexpr int MIN_WAVES_X = 4;
#else
  static constexpr int MIN_WAVES_X = 2;
#endif
  static constexpr int MIN_WAVES = UWREF_ ? MIN_WAVES_X : (TPW_ == 2 ? 2 : ((LW == 64 || KT) ? 3 : 4));
  static_assert(sizeof(R_) == 8, "8-byte elements (double, or two fp32 instances per lane)");
};

// STREAM (one tracer per launch: every byte is used once and the kernel is bound by its data
// movement): main parts fetched and stored with the streaming hint, remainders with the default
// policy -- two instructions per array and pair, two per column store.  !STREAM (tracer batches:
// VALU-bound, u and w are re-read from L2 by the other tracers of the tile): ONE instruction per
// array and pair / per column store, default policy, every lane addressing its own part.
// TPW = 2 (tracer batches only): a wave advects TWO tracers of its tile at once.  u, w, the
// upwind selects, the u / w sums and the tracer-independent factors of the antidiffusive fluxes
// (:571-573, :580-582) are fetched / formed once for both, the tracer state (register pipeline,
// flux sums, store) is a Pair.
// UWCONV (with UWREF; tracer batches on fresh velocities, mpdata_plan_run_uw with ntracers > 1): the wave also
// WRITES the u, w values it reads from the workgroup's ring into the plan's own u, w arrays (plan layout: its
// lane's element of the column, the very offsets f's store uses) -- the first tracer of the batch is advected by
// this kernel, and the batch kernel behind it finds the converted velocities without a conversion pass.
// wm_body: everything one wave does for its (tile, first tracer `tr`) -- the body of the kernels below.  `lds` =
// the workgroup's LDS, `my` = the wave's own ring inside it (the caller lays the rings out: a launch that mixes
// the two-tracer and the one-tracer form gives every wave a two-tracer-sized ring).
// NPK > 0 (EXACT, one tracer per wave, nx <= NPK; round 5): the lane's nx limited vertical fluxes are parked in NPK
// REGISTERS instead of in HBM and added onto the finished upwind sum behind the march, in the reference's order
// (:545, :624: bit-identical flux) -- no park array, no finishing kernel, no extra HBM bytes; 2 waves per SIMD.
template <typename R, int LPS, int WPB, bool STREAM, int TPW = 1, bool UWREF = false, bool UWCONV = false, int NPK = 0>
__device__ __forceinline__ void wm_body(const MpdataWmArgsT<R>& a, R* const lds, R* const my, const int wave,
                                        const unsigned tile, const unsigned tr, const unsigned ntr, const bool tile_ok,
                                        const int kwave = 0, const int ntl_in = 1) {
  using T = TileWm<R, LPS, WPB, TPW, UWREF>;
  static_assert(!UWCONV || UWREF, "UWCONV: the kernel that reads u, w from the reference layout");
  static_assert(TPW == 1 || (TPW == 2 && !STREAM), "two tracers per wave: batch form only");
  static_assert(!UWREF || (STREAM && TPW == 1), "u, w from the reference layout: one tracer per launch");
  static_assert(NPK == 0 || (TPW == 1 && !UWREF && NPK % 6 == 0), "register park: one tracer per wave, whole trips of six columns");
  constexpr bool KS = T::KS, KT = T::KT;
  constexpr int LW = T::LW;
  constexpr bool WSEL = LW == 64 || KT;   // the ghost level's w = 0 by a select (no zero row of the LDS image to read it from)
  static_assert(!KS || (!STREAM && !UWREF), "nz > 64: the batch form of the data movement (one fetch instruction per array and pair)");
  using V = std::conditional_t<TPW == 1, R, Pair<R>>;   // a tracer-dependent quantity
  // PRE: the LDS inputs of a column step are read one step AHEAD of their use (software pipelining; the pair loop
  // holds them in registers) -- two tracers per wave: with two waves per SIMD an exposed LDS round trip at the top of
  // every step is not hidden by other waves.  (The register-park form of EXACT, also two waves per SIMD, was measured
  // with it in round 5: 0.531 against 0.533 ms, 25 tracers 11.44 against 11.37 -- it is bound by its IEEE divisions,
  // not by LDS latency; profiles/r05_ab_exact_prefetch.txt.  Left as it was.)
  constexpr bool PRE = TPW == 2;
  // DMA instructions one pair fetch issues (what the counted waits count)
  constexpr int DMA_PER_PAIR = STREAM ? 6 : 2 + TPW;
  // T1X (FAST, one tracer per wave; fp64 and the two-instances-per-lane fp32 form): the 7-operation extrema and the ring sums of the two-tracer form
  // (XNEW, XSUM below: 6 operations per column fewer) in 128 VGPRs.  The registers come from: rho folded into
  // the limiter's reciprocal, one flux accumulator, ONE ring value dW/adz - U per column instead of U and dW
  // (ESUM), the flux position formed again behind the march.  +1 % on the headline (and on fp32); in the u, w-ring form it
  // spills (-7 %), and in the two-tracer form ESUM alone costs 1 % (scheduling): both left as they were.
#ifdef MPDATA_FAST_DIV
  constexpr bool FASTV = true;
#else
  constexpr bool FASTV = false;
#endif
  constexpr bool T1X = FASTV && TPW == 1 && !UWREF;
  // The u, w-ring form (FAST) takes what fits of it: the merged ring value, the folded rho and the flux position make
  // room for the 7-operation extrema, not for the ring sums (they spill inside the march: -8 %).  +0.5 ... 0.9 %.
  constexpr bool UWX2 = FASTV && UWREF;
  // the 7-operation extrema (stage A below): where the registers allow it (the u, w-ring form with one instance per
  // wave, LPS = 64, and its converting form UWCONV are two registers short: they keep the merged ring value and take
  // the extrema as written)
  // (the register-park form of EXACT runs in a 256-register budget like the two-tracer form: it takes them too -- the
  //  extremum of a set does not depend on the order, the results are the same bits; -1.1 % one tracer and 25, interleaved:
  //  profiles/r05_ab_exact_extrema.txt)
  constexpr bool XNEW = TPW == 2 || T1X || (UWX2 && LW < 64 && !UWCONV) || NPK > 0;
  // FAST, two tracers per wave (the VALU- / power-bound form): three operations per tracer and column fewer
  // by carrying sums in the rings and sharing the velocity parts of the upwind fluxes (below: XSUM)
  constexpr bool XSUM = FASTV && XNEW && !UWX2;
  constexpr bool ESUM = T1X || UWX2;
  // the upwind fluxes (:532, :537): two tracers per wave share the velocity parts max(0,u), min(0,u) (a multiply and
  // an FMA per tracer); one tracer per wave takes the select form (compare, two 32-bit selects, multiply: the same
  // count, measured 1 % faster -- the chip runs at the clock its power draw leaves it, and these are cheaper operations)
  // (the select form in the two-tracer kernel: 7 operations per pair instead of 6, no difference measured)
  constexpr bool SELUP = TPW == 1;
  // fp64: same operations in an order with shorter dependency chains (the extrema as a tree with the shifted
  // values last; FAST: the limiter denominators' horizontal part formed ahead of W2u, one multiplication behind
  // the reciprocal instead of two): +0.5 % on the headline.  (The fp32 forms spill with it.)
  constexpr bool CHAIN = std::is_same<R, double>::value;
  constexpr int SLP = T::SLP, RB = (int)sizeof(R);

  const int nx = a.nx, nz = a.nz, nzm = nz - 1;
  const int lane = threadIdx.x & 63;

#ifdef MPDWM_STAMPS
  // diagnostic build (tools/wave_timeline.py): every wave records its start / end on the 100-MHz
  // real-time counter and the shader clock, the shader cycles it spent in the counted DMA waits and
  // where it ran (HW_ID, XCC_ID): 8 words per wave at dbg[8 * (tracer-launch wave index)]
  unsigned long long st_r0, st_c0, st_wait = 0;
  asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_r0), "=s"(st_c0)::"memory");
#define MPDWM_WAIT(STR)                                                                             \
  {                                                                                                 \
    unsigned long long ta_, tb_;                                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ta_)::"memory");                     \
    asm volatile(STR ::: "memory");                                                                 \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb_)::"memory");                     \
    st_wait += tb_ - ta_;                                                                           \
  }
#else
#define MPDWM_WAIT(STR) asm volatile(STR ::: "memory");
#endif
  const int chunk = SLP * nzm;                    // elements of one column of the tile
  const unsigned chunkB = (unsigned)(chunk * RB);
  const long long toff = (long long)tile * a.tile_elems;
  R* const f = a.f + (long long)tr * a.f_tstride + toff;
  const R* const u = a.u + toff;
  const R* const w = a.w + toff;
  // ---- lane -> (instance s, level k) -----------------------------------------
  const int s_l = lane / LW;
  // KT: the wave's tiles tile .. tile + ntl - 1 (ntl_in <= TPV of them, as the caller says); lane's tile = tile + s_l
  [[maybe_unused]] const int ntl = tile_ok ? ntl_in : 0;
  [[maybe_unused]] const bool tl_ok = !KT || s_l < ntl;
  [[maybe_unused]] const int s_t = KT ? (tl_ok ? s_l : 0) : 0;   // tile of the lane's constants and flux (a lane without a tile: the first)
  const R* const kc = a.kc + ((long long)tile + s_t) * (3 * chunk);
  R* const flux = a.flux + (long long)tr * a.flux_tstride + ((long long)tile + s_t) * chunk;
  // second tracer of the wave (TPW = 2); if it does not exist (odd ntracers) it is addressed through
  // an EMPTY buffer range: its fetches deliver zeros, its stores are dropped
  const bool has1 = TPW == 2 && tr + 1 < ntr;
  R* const f1 = has1 ? f + a.f_tstride : f;
  R* const flux1 = has1 ? flux + a.flux_tstride : flux;

  // KS: the wave's first level - 1.  58 kwave, but never past the (even) start from which the wave still reaches nz
  // (KT: 58 kwave: the host chooses this form only where 58 (nkw - 1) + LW >= nz)
  int koff = 0;
  if constexpr (KT) koff = 58 * kwave;
  else if constexpr (KS) koff = min(58 * kwave, max(0, (nz - 64 + 1) & ~1));
  const int kk = lane % LW + koff;         // k - 1
  const int k = kk + 1;
  const bool lvl_ok = k <= nzm;            // real level (else ghost / dead lane)
  const int kl = lvl_ok ? kk : nzm - 1;    // level index whose data the lane reads
  const int pos = (KS ? 0 : s_l * nzm) + kl;          // element of the chunk
  // the levels whose results this wave stores (KS: the part of its 64 that is outside the cone of its artificial ends)
  bool out_ok = lvl_ok;
  if constexpr (KS) out_ok = lvl_ok && tl_ok && (kwave == 0 || k >= 58 * kwave + 4) && (kwave + 1 == a.nkw || k <= 58 * kwave + 61);

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
  //      written with data (chunk <= 62: the lane that fetches it is out of range and delivers
  //      zeros with every pair; LPS = 64 selects w instead): with w = 0 the ghost level's fluxes are exact zeros, which is all the
  //      level below ever takes from it (www(:,:,:,nz) = 0, :511).
  // (KS: the LDS image of a column holds the wave's 64 levels; the first and the last lane of a wave that does not hold
  //  level 1 / level nzm read themselves as their neighbour: they are inside the cone of the wave's artificial end)
  // (KT: the image of a column = the windows of the wave's tiles, LW rows each)
  constexpr int IST = KT ? LW : 0;   // rows between the images of two tiles of a wave
  const R* const p_own = my + ((WSEL || lvl_ok) ? (KT ? s_l * IST + kl - koff : pos - koff) : 63);
  const R* const p_dn = my + ((KT ? s_l * IST : (KS ? 0 : s_l * nzm)) + max((kl > 0 ? kl - 1 : 0) - koff, 0));
  const R* const p_up = my + ((KT ? s_l * IST : (KS ? 0 : s_l * nzm)) + min((kl + 1 < nzm ? kl + 1 : nzm - 1) - koff, KS ? LW - 1 : 1 << 30));

  // ---- global addressing: one descriptor per array, based at the wave's tile (32-bit offsets
  //      inside a tile, arrays of any size).  Element e of column c of the tile lives at
  //      c*mainB + 8e (e in the main part) or remBase + c*remB + (8e - mainB).
  const unsigned OOB = 0xFFFFFFF8u;
  const int ncol = nx + 6;
  const unsigned mainB = chunkB / 128u * 128u, remB = chunkB - mainB;
  const unsigned remBase = (unsigned)ncol * mainB;
  const long long tileB = (long long)ncol * chunkB;
  // KT: the wave's descriptors span its ntl tiles (a.tile_elems apart); a lane adds its tile's offset
  [[maybe_unused]] const unsigned tstrB = (unsigned)(a.tile_elems * RB);
  const long long spanB = KT ? (ntl > 0 ? (long long)(ntl - 1) * tstrB + tileB : 0) : tileB;
  // (!tile_ok -- the u, w-ring form and the nz > 64 form, whose workgroups synchronise: a wave beyond the last tile stays
  //  and works on EMPTY ranges: its fetches deliver zeros, its stores are dropped)
  const __amdgpu_buffer_rsrc_t rsf = v2::make_rsrc(f, !tile_ok ? 0 : spanB);
  const __amdgpu_buffer_rsrc_t rsf1 = v2::make_rsrc(f1, (has1 && tile_ok) ? spanB : 0);
  // (UWREF: u, w of the plan are not read; UWCONV writes them)
  const __amdgpu_buffer_rsrc_t rsu = v2::make_rsrc(u, UWREF ? ((UWCONV && tile_ok) ? tileB : 0) : (tile_ok ? spanB : 0));
  const __amdgpu_buffer_rsrc_t rsw = v2::make_rsrc(w, UWREF ? ((UWCONV && tile_ok) ? tileB : 0) : (tile_ok ? spanB : 0));
  // EXACT with a park array (a.wpark != null): bit-identical flux.  The reference adds the limited vertical fluxes
  // ONE BY ONE onto the finished upwind sum (:545, :624), and the first of them exists 30 columns before that sum is
  // complete: every lane parks its nx limited fluxes in [tracer][tile][column 1..nx][lane] (one 512-byte row per wave
  // and column step), stores the upwind sum alone as flux, and a finishing kernel (mpdata_plan.hip: flux_finish_kernel)
  // adds the parked terms in the reference's order.  FAST never parks (its flux is a sum in another order anyway).
  constexpr bool CAN_PARK = !FASTV && NPK == 0;
  constexpr bool REG_PARK = !FASTV && NPK > 0;
  [[maybe_unused]] R PK[NPK > 0 ? NPK : 1];   // REG_PARK: limited vertical flux of column i = PK[i - 1] (constant indices only: registers)
  [[maybe_unused]] const bool park = CAN_PARK && a.wpark != nullptr;
  [[maybe_unused]] const long long parkB = (long long)nx * 64 * RB;   // bytes of one (tracer, tile) block
  // (KS: a block per wave of the instance, [tracer][tile][wave][column][lane]; a.nkw = 1 otherwise)
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsp = v2::make_rsrc(
      a.wpark + (((long long)tr * a.ntiles + tile) * a.nkw + kwave) * ((long long)nx * 64), (park && tile_ok) ? parkB : 0);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsp1 = v2::make_rsrc(
      a.wpark + (((long long)(tr + 1) * a.ntiles + tile) * a.nkw + kwave) * ((long long)nx * 64), (park && has1) ? parkB : 0);
  [[maybe_unused]] auto park_st = [&](const int col, const V w3) __attribute__((always_inline)) {   // col = 1 .. nx
    unsigned z;   // (the lane's byte offset formed from the execution mask here: no register carries it through the march)
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    const unsigned lo = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) * (unsigned)RB;
    const unsigned so = (unsigned)(col - 1) * (64u * RB);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2::u32x2, first(w3)), rsp, (int)lo, (int)so, 0);
    if constexpr (TPW == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2::u32x2, second(w3)), rsp1, (int)lo, (int)so, 0);
  };
  // ---- UWREF: u, w in the reference layout (sl fastest, :33-38): u(ncrms, nx+5, nzm), w(ncrms, nx+4, nz).
  //      Row (column c, level kk) of the workgroup = 16 instances = 128 bytes at
  //        ((c-1) + ncols*kk) * ncrms*8 + sl_base*8          (u, w have no column c = 0).
  //      One 16-byte-per-lane LDS-DMA instruction fetches 8 rows (8 lanes each); wave wv fetches row
  //      group wv % XRG of array wv / XRG (u: 0, w: 1), the two columns of a pair with one instruction
  //      each -- same per-lane offset, the column is the scalar offset.  LDS image of a block:
  //      [row][16 instances], the 16-byte chunk (2 instances) at position p of row r holds source
  //      chunk p ^ ((r >> 1) & 7) (swizzle on the SOURCE address): a wave's transposed read (lanes
  //      along the rows, 8 bytes of one chunk each) is then 2-way bank-conflicted at worst.
  R* const xring = lds + WPB * (T::NS * T::SLOT);
  [[maybe_unused]] unsigned xv = 0xFFFFFFF8u, x_colB = 0;
  [[maybe_unused]] int x_ncols = 0, x_dst = 0;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t rsx = rsu;
  [[maybe_unused]] const R *x_own = lds, *x_dn = lds, *x_up = lds;
  if constexpr (UWREF) {
    const int xa = wave / T::XRG, xrg = wave % T::XRG;          // this wave's array and row group
    x_ncols = xa == 0 ? nx + 5 : nx + 4;                         // columns of the array
    const long long lvl = a.ncrms * (long long)x_ncols;          // elements between levels
    const long long rows_a = xa == 0 ? nzm : nz;                 // levels the array holds
    rsx = v2::make_rsrc(xa == 0 ? a.u_ref : a.w_ref, lvl * rows_a * RB);
    x_colB = (unsigned)(a.ncrms * RB);
    const int rl = lane >> 3, row = xrg * 8 + rl;                // the lane's row of the block
    const int pch = lane & 7;                                     // its 16-byte chunk position in the LDS row
    const long long sl_src = (long long)blockIdx.x * T::GX + 2 * (pch ^ ((row >> 1) & 7));
    // rows >= nzm (w: level nz is never read, :511) and instances beyond the array's end fetch nothing:
    // out of range = zeros into that part of the block
    const bool ok = row < nzm && sl_src + 1 < a.ncrms;
    xv = ok ? (unsigned)((sl_src + lvl * row) * RB) : 0xFFFFFFF8u;
    x_dst = xa * T::XARR + xrg * 8 * T::GX;                      // element offset inside a [column] half slot
    // read positions: row = level index (ghost and dead lanes: the zero rows nzm .. LPS-1)
    const int s_g = wave * SLP + lane / LW;                      // instance inside the workgroup
    auto xpos = [&](const int r) __attribute__((always_inline)) {
      return r * T::GX + ((((s_g >> 1) ^ ((r >> 1) & 7)) << 1) | (s_g & 1));
    };
    const int kk_ = lane % LW;
    const bool real = kk_ < nzm;
    x_own = xring + xpos(kk_);
    x_dn = xring + xpos(kk_ > 0 ? kk_ - 1 : 0);
    x_up = xring + xpos(real ? (kk_ + 1 < nzm ? kk_ + 1 : nzm - 1) : (kk_ + 1 < LW ? kk_ + 1 : LW - 1));
  }
  const unsigned posB = (unsigned)(pos * RB);
  const unsigned st_main = (out_ok && posB < mainB) ? posB : OOB;               // the lane's element of a chunk:
  const unsigned st_rem = (out_ok && posB >= mainB) ? remBase + (posB - mainB) : OOB;  // in one of the two parts
  [[maybe_unused]] const bool st_in_main = posB < mainB;
  [[maybe_unused]] const unsigned st_ab = out_ok ? (st_in_main ? posB : remBase + (posB - mainB)) : OOB;
  // DMA source offsets of a pair instruction: lane L < 32 fetches bytes 16L.. of the even column,
  // lane L >= 32 bytes 16(L-32).. of the odd one (the LDS image keeps the two columns 512 B apart);
  // the lanes of a column's main part in the one instruction, those of its remainder in the other
  // (KT: the 32 lanes of a column = LW / 2 pieces of each of the wave's tiles: the image keeps the windows LW rows apart)
  constexpr int PPT = KT ? LW / 2 : 32;   // 16-byte pieces of a column per tile
  const unsigned in_col = (unsigned)(((lane & 31) % PPT) * 16 + koff * RB);   // (KS: the wave's 512 bytes of the column start at level koff + 1)
  [[maybe_unused]] const unsigned dtB = KT ? (unsigned)((lane & 31) / PPT) * tstrB : 0u;   // the tile the lane FETCHES for
  const unsigned hi = lane >= 32 ? 1u : 0u;
  const bool lane_main = in_col < mainB;
  const unsigned vA = lane_main ? in_col + hi * mainB + dtB : OOB;
  const unsigned vB = (in_col >= mainB && in_col < chunkB) ? remBase + (in_col - mainB) + hi * remB + dtB : OOB;
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
  // (the register-starved forms of the u, w-ring kernel: ONE register for both branches of the f fetch -- seen as a
  //  select of vA / vB the compiler keeps both)
  if constexpr (UWREF && (LW == 64 || UWCONV)) asm volatile("" : "+v"(dcur));
  // the two per-lane strides share one register: pair stride of the lane's DMA part in the low
  // half, column stride of its store part in the high half (both <= 1024); a stride is added with
  // the half-word select of the add itself (SDWA), so unpacking costs no instruction
  const unsigned pstride = dcur == OOB ? 0u : (lane_main ? 2u * mainB : 2u * remB);
  const unsigned cstride = out_ok ? (posB < mainB ? mainB : remB) : 0u;
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
  unsigned scur = out_ok ? (posB < mainB ? posB : remBase + (posB - mainB)) + (KT ? (unsigned)s_l * tstrB : 0u) - (PRE ? 3u : 4u) * cstride : OOB;

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
    if constexpr (UWREF) {   // f only (u, w: dma_uw); ONE offset register for both parts (dcur: never advanced here)
      const unsigned soA = (unsigned)(2 * P) * mainB, soB = (unsigned)(2 * P) * remB;
      const unsigned of_ = halves(dcur, ef, of);
      if (lane_main) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)of_, (int)soA, 0, AUX_NT);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)of_, (int)soB, 0, 0);
    } else if constexpr (STREAM) {
      const unsigned soA = (unsigned)(2 * P) * mainB, soB = (unsigned)(2 * P) * remB;
      // A lane that is out of range in an LDS-DMA instruction still writes (zeros) to its LDS
      // position, so the two instructions of an array must not both be executed by a lane: the
      // main-part fetch runs with the main lanes only, the remainder fetch with the others.
      if (lane_main) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)halves(vA, ef, of), (int)soA, 0, AUX_NT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::UO), 16, (int)halves(vA, eu, ou), (int)soA, 0, AUX_NT);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + T::WO), 16, (int)halves(vA, ew, ow), (int)soA, 0, AUX_NT);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)halves(vB, ef, of), (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::UO), 16, (int)halves(vB, eu, ou), (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + T::WO), 16, (int)halves(vB, ew, ow), (int)soB, 0, 0);
      }
    } else {
      const unsigned of_ = halves(dcur, ef, of);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf, (lds_ptr_t)(d), 16, (int)of_, 0, 0, 0);
      if constexpr (TPW == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsf1, (lds_ptr_t)(d + T::ARR), 16, (int)of_, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsu, (lds_ptr_t)(d + T::UO), 16, (int)halves(dcur, eu, ou), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr_t)(d + T::WO), 16, (int)halves(dcur, ew, ow), 0, 0, 0);
      dcur = add_lo(dcur, strides);   // (pairs are issued in order, P = 0, 1, 2, ...)
    }
  };
  auto dma_pair = [&](const int P) __attribute__((always_inline)) {
    const int c0 = 2 * P, c1 = 2 * P + 1;  // the pair's columns
    dma_issue(P, c0 < ncol, c1 < ncol, c0 >= 1 && c0 < ncol, c1 < ncol, c0 >= 1 && c0 <= nx + 4, c1 <= nx + 4);
  };
  // UWREF: this wave's share of u / w of pair P into the workgroup's ring (two instructions: the
  // pair's two columns).  A column that does not exist for the array fetches nothing (zeros).
  [[maybe_unused]] auto dma_uw = [&](const int P) __attribute__((always_inline)) {
    if constexpr (UWREF) {
      R* d = xring + (P % T::NS) * T::XSLOT + x_dst;
      const int c0 = 2 * P, c1 = 2 * P + 1;
      const bool e0 = c0 >= 1 && c0 <= x_ncols, e1 = c1 <= x_ncols;   // u: c = 1 .. nx+5, w: c = 1 .. nx+4
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(d), 16, (int)(e0 ? xv : OOB), (int)((unsigned)max(c0 - 1, 0) * x_colB), 0, AUX_NT);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr_t)(d + 2 * T::XARR), 16, (int)(e1 ? xv : OOB), (int)((unsigned)max(c1 - 1, 0) * x_colB), 0, AUX_NT);
    }
  };
  // interior pairs (1 <= P, 2P+1 <= nx+4): no conditions
  auto dma_pair_full = [&](const int P) __attribute__((always_inline)) {
    dma_issue(P, true, true, true, true, true, true);
  };
  // column c of f back to memory.  STREAM: main part with the streaming hint, remainder without
  // (two instructions); else one instruction, every lane to its part.
  // AHEAD = 0: the step's regular store (column c = q - 1; advances the running offset);
  // AHEAD = 2: the early store of a halo column c = q + 1.
  auto st_col = [&](const bool act, const int c, const V v, auto ahead_tag) __attribute__((always_inline)) {
    constexpr int AHEAD = decltype(ahead_tag)::value;
    const v2::u32x2 b = __builtin_bit_cast(v2::u32x2, first(v));
    if constexpr (UWREF) {   // as STREAM, with ONE offset register: a lane's element lies in one of the two parts
      const unsigned o = act ? st_ab : OOB;
      if (st_in_main) __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)o, (int)((unsigned)c * mainB), AUX_NT);
      else __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)o, (int)((unsigned)c * remB), 0);
    } else if constexpr (STREAM) {
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)(act ? st_main : OOB), (int)((unsigned)c * mainB), AUX_NT);
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)(act ? st_rem : OOB), (int)((unsigned)c * remB), 0);
    } else {
      // (c is implied by the running offset)
      unsigned o = scur;
      if (AHEAD == 2) o = add_hi(add_hi(scur, strides), strides);
      const unsigned oa = act ? o : OOB;
      __builtin_amdgcn_raw_buffer_store_b64(b, rsf, (int)oa, 0, 0);
      if constexpr (TPW == 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2::u32x2, second(v)), rsf1, (int)oa, 0, 0);
      if (AHEAD == 0) scur = add_hi(scur, strides);
    }
  };

  // ---- prologue: pairs 0, 1, 2 into flight, each behind two dropped column stores so that the
  //      counted wait of the first pairs sees the steady-state op pattern.  A fetch instruction
  //      group writes the 16 bytes of every ACTIVE lane of its array block -- a lane that is out of range
  //      (a column that does not exist for u or w) writes zeros --; the tails of the blocks behind the
  //      chunk belong to lanes that are switched off below and are zeroed once (u, w-ring form: all
  //      lanes stay on and rewrite them with every pair).
#ifdef MPDWM_ABL_NODMA  // timing ablation: arithmetic on (non-zero, finite) stand-in data
#pragma unroll
  for (int j = 0; j < T::NS * T::SLOT / 64; ++j) my[j * 64 + lane] = R(0.25) + R(0.001) * R((j * 64 + lane) % 97) - R(0.3) * R(lane & 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  // Lanes that own neither a real level (k <= nzm) nor a 16-byte piece of a column fetch are switched OFF for the whole
  // march (nz = 28: 10 of 64 lanes): the kernel runs at the socket's power cap, and a lane that computes on zeros still
  // draws (round 5: -0.8 % one tracer and 25 tracers, interleaved; profiles/r05_ab_exec_mask.txt).  That includes the
  // ghost level k = nz: a DPP read of a disabled lane delivers +0 (bound_ctrl), which IS www(:,:,:,nz) = 0 (:511) for
  // the upward shifts of the level below (W1, W3; its other two upward reads are clamped to itself, :602).  The zero tails
  // of the LDS blocks, which those lanes' fetches would rewrite with every pair, are written once here instead.
  // (Not the u, w-ring form: all 64 lanes of its waves fetch rows for the whole workgroup.  Switching them back on for
  //  those two instructions was built: -0.6 % on mpdata_plan_run_uw -- and wrong at the edges, because every per-lane
  //  value the re-enabled lanes need must have been formed BEFORE the mask and must never be copied under it; a mask
  //  that is never lifted has no such lane.)
#ifndef MPDWM_NO_EXECMASK
  if constexpr (!UWREF) {
#ifndef MPDWM_ABL_NODMA   // (that ablation computes on the stand-in data it has just written)
#pragma unroll
    for (int j = 0; j < T::NS * T::SLOT / T::ARR; ++j) {
      my[j * T::ARR + 2 * lane] = R(0);
      my[j * T::ARR + 2 * lane + 1] = R(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    const unsigned long long km = __builtin_amdgcn_ballot_w64(lvl_ok || in_col < chunkB);
    asm volatile("s_mov_b64 exec, %0" ::"s"(km) : "memory");
  }
#endif
#pragma unroll
  for (int P = 0; P < T::NS; ++P) {
    st_col(false, 0, V(R(0)), std::integral_constant<int, 1>{});
    st_col(false, 0, V(R(0)), std::integral_constant<int, 1>{});
    dma_pair(P);
    if (UWREF && P < 2) dma_uw(P);   // issue order f(0) uw(0) f(1) uw(1) f(2): what the counted wait assumes
  }
  __builtin_amdgcn_sched_barrier(0);
  const R IRHO = recip_const(RHO);
  const R IADZ = recip_const(adz_l);
  const R IRHOW = recip_const(rhow_l * adz_l);
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
  const V ZV = V(R(0));
  Window<V> S;   // the tracer state: register pipeline of the column march
  Window<R> G;   // the rings of u, w and their sums (UR .. SU): common to the tracers of the wave
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    S.F0[j] = S.PMX[j] = S.PMN[j] = S.U1[j] = S.DW1[j] = ZV;
    S.F1[j] = S.F1D[j] = S.F1U[j] = S.MX0[j] = S.MN0[j] = ZV;
    G.UR[j] = G.UD[j] = G.PW[j] = G.SW[j] = G.WR[j] = G.SU[j] = R(0);
    S.U2P[j] = S.U2N[j] = ZV;
    S.MXN[j] = S.MNN[j] = S.U3[j] = S.DW3[j] = S.SF[j] = ZV;
  }
  V S1 = ZV, S3 = ZV;
  V v_def = ZV;   // the deferred store of a pair's odd column
  bool act_def = false;
  int c_def = 0;


  // a tracer-dependent input from the LDS ring (f of the wave's tracers: array blocks 0 .. TPW-1)
  auto ldv = [&](const R* p) __attribute__((always_inline)) {
    if constexpr (TPW == 1) return p[0];
    else return V(p[0], p[T::ARR]);
  };
  // The LDS inputs of one column step.  One tracer per wave: the step reads them itself (`in` is
  // not used: with 128 VGPRs they cannot be held ahead of time).  Two tracers per wave: the pair
  // loop fetches them a whole step ahead of their use (software pipelining; with two waves per
  // SIMD an exposed LDS round trip at the top of every step is not hidden by other waves).
  struct LdsIn { V f0q, f0d, f0u; R uq, wq, ud, wu; };
  auto lds_in = [&](auto sl_tag, auto h_tag) __attribute__((always_inline)) {
    constexpr int LO = decltype(sl_tag)::value * T::SLOT + decltype(h_tag)::value * T::HALF;
    LdsIn r;
    r.f0q = ldv(p_own + LO);
    r.uq = p_own[LO + T::UO];
    r.wq = p_own[LO + T::WO];
    if constexpr (WSEL) r.wq = lvl_ok ? r.wq : R(0);
    r.f0d = ldv(p_dn + LO);
    r.f0u = ldv(p_up + LO);
    r.ud = p_dn[LO + T::UO];
    r.wu = p_up[LO + T::WO];
    return r;
  };
  // One column step.  PH = c mod 3 selects the register ring slots, SL = (c/2) mod 3 the LDS
  // ring slot, H = c mod 2 the column of the pair (c = q+2); FULL = steady state (4 <= q <= nx).
  auto step = [&](auto ph_tag, auto sl_tag, auto h_tag, auto full_tag, const int q, const LdsIn& in) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_tag)::value;
    constexpr int LO = decltype(sl_tag)::value * T::SLOT + decltype(h_tag)::value * T::HALF;  // LDS offset
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int C0 = PH, C1 = (PH + 2) % 3, C2 = (PH + 1) % 3, C3 = PH;  // slots of q, q-1, q-2, q-3

    V f0q;
    R uq, wq;   // (ghost level: w = 0)
    // UWREF: u, w of the column from the workgroup's ring ([pair slot][column of pair][u, w] blocks)
    constexpr int XO = decltype(sl_tag)::value * T::XSLOT + decltype(h_tag)::value * 2 * T::XARR;
    if constexpr (UWREF) {
      f0q = ldv(p_own + LO);
      uq = x_own[XO];
      wq = x_own[XO + T::XARR];   // (ghost level: a zero row)
      if constexpr (UWCONV) {   // column c = q + 2 of u (c = 1 .. nx+5) and w (c = 1 .. nx+4) into the plan's arrays
        const int c = q + 2;
        const unsigned ou = (FULL || (c >= 1 && c <= nx + 5)) ? st_ab : OOB, ow = (FULL || (c >= 1 && c <= nx + 4)) ? st_ab : OOB;
        const v2::u32x2 bu = __builtin_bit_cast(v2::u32x2, uq), bw = __builtin_bit_cast(v2::u32x2, wq);
        // (as f's store: the line-aligned main part with the streaming policy, the remainder without)
        if (st_in_main) {
          __builtin_amdgcn_raw_buffer_store_b64(bu, rsu, (int)ou, (int)((unsigned)max(c, 0) * mainB), AUX_NT);
          __builtin_amdgcn_raw_buffer_store_b64(bw, rsw, (int)ow, (int)((unsigned)max(c, 0) * mainB), AUX_NT);
        } else {
          __builtin_amdgcn_raw_buffer_store_b64(bu, rsu, (int)ou, (int)((unsigned)max(c, 0) * remB), 0);
          __builtin_amdgcn_raw_buffer_store_b64(bw, rsw, (int)ow, (int)((unsigned)max(c, 0) * remB), 0);
        }
      }
    } else if constexpr (!PRE) {
      f0q = ldv(p_own + LO);
      uq = p_own[LO + T::UO];
      wq = p_own[LO + T::WO];
      if constexpr (WSEL) wq = lvl_ok ? wq : R(0);
    } else {
      f0q = in.f0q; uq = in.uq; wq = in.wq;
    }

#ifdef MPDWM_ABL_NOCOMPUTE  // timing ablation only: data movement without the arithmetic
    st_col(q - 3 >= -1 && q - 3 <= nx + 2, max(q - 1, 0), f0q + uq + wq, std::integral_constant<int, 0>{});
    return;
#endif
#define DN_C(x) shift_dn_clamped((x), own_dn)
#define UP_C(x) shift_up_clamped((x), own_up)
#define UP_G(x) shift_up(x)
    V f0d, f0u;
    if constexpr (!PRE) {
      // (by DPP from f0q instead -- two 32-bit select-moves per value in place of an LDS read, no address registers:
      //  0.3885 against 0.3868 ms interleaved, round 4: the LDS reads are the cheaper form, also under the power cap)
      f0d = ldv(p_dn + LO);
      f0u = ldv(p_up + LO);
    } else {
      f0d = in.f0d; f0u = in.f0u;
    }
    const V F0p = S.F0[C1];

    // ================= stage A =================================================
    V U1q = ZV, DW1q = ZV, f1_1 = ZV, F1D_1 = ZV, F1U_1 = ZV, MX0_1 = ZV, MN0_1 = ZV;
    [[maybe_unused]] V G1mx = F0p, G1mn = F0p;   // G of column q-1 (an inactive step: never used by a valid star)
    if (FULL || (q >= -1 && q <= nx + 3)) {
      // XSUM: max(0,u) f(ib) + min(0,u) f(i) as written (:532) -- one product is an exact zero, so the value is
      // the select form's; the two velocity parts are formed once for the wave's tracers: a multiply and an
      // FMA per tracer instead of a 64-bit select and a multiply
      if constexpr (XSUM && !SELUP) U1q = dmax(uq, R(0)) * F0p + dmin(uq, R(0)) * f0q;
      else U1q = upwind(uq, F0p, f0q);  // :532
      if (FULL || q <= nx + 2) {
        V W1q;
        if constexpr (XSUM && !SELUP) W1q = dmax(wq, R(0)) * f0d + dmin(wq, R(0)) * f0q;
        else W1q = upwind(wq, f0d, f0q);  // :537
        DW1q = UP_G(W1q) - W1q;
        if (FULL || (q >= 1 && q <= nx)) S1 = S1 + W1q;  // :545 (UWREF: S1 also takes the limited terms)
      }
      if (FULL || q >= 0) {
        // ESUM: the ring carries the column's own share  dW1 / adz - U1  of the next column's update (one value
        // instead of two; same operation count)
        if constexpr (ESUM) f1_1 = F0p - (U1q + S.DW1[C1]) * IRHO;
        else f1_1 = F0p - ((U1q - S.U1[C1]) + S.DW1[C1] * IADZ) * IRHO;  // :557, column q-1
        F1D_1 = DN_C(f1_1);
        F1U_1 = UP_C(f1_1);
        if constexpr (!XNEW) {
        MX0_1 = dmax(S.PMX[C1], f0q);  // :521-522 complete for column q-1
        MN0_1 = dmin(S.PMN[C1], f0q);
        } else {
        // Two tracers per wave (256 VGPRs; the one-tracer kernels have no room for G across the stages):
        // extrema of BOTH passes at once (:521-522 and :596-597 take the max / min of the same five
        // points of f0 and of f1 -- the extremum of a set does not depend on the order): with
        //   G(i) = max(f0(i,k), f1(i,k)),  A(i) = max(f0(i,kb), f0(i,kc), G(i-1)),
        //   P(i) = max(A(i), f1(i,kb), f1(i,kc), G(i))    the star of column i is  max(P(i), G(i+1)):
        // 7 operations per column instead of 9, each for max and min.  PMX / PMN carry A, MX0 / MN0 carry P.
        G1mx = dmax(F0p, f1_1);
        G1mn = dmin(F0p, f1_1);
        if constexpr (CHAIN) {   // as a tree: the shifted values arrive last
          MX0_1 = dmax(dmax(S.PMX[C1], G1mx), dmax(F1D_1, F1U_1));   // P of column q-1
          MN0_1 = dmin(dmin(S.PMN[C1], G1mn), dmin(F1D_1, F1U_1));
        } else {
          MX0_1 = dmax(dmax(dmax(S.PMX[C1], F1D_1), F1U_1), G1mx);
          MN0_1 = dmin(dmin(dmin(S.PMN[C1], F1D_1), F1U_1), G1mn);
        }
        }
        // the last two halo columns (nx+1, nx+2) keep this first-pass value (:557) and no later
        // step finishes them: store them now (their input values are already in the LDS ring)
        if (!FULL && q - 1 >= nx + 1) st_col(q - 1 <= nx + 2, q + 1, f1_1, std::integral_constant<int, 2>{});
      }
    }
    if constexpr (ESUM) {
      S.DW1[C0] = DW1q * IADZ - U1q;
    } else {
      S.U1[C0] = U1q;
      S.DW1[C0] = DW1q;
    }
    S.F1[C1] = f1_1;
    S.F1D[C1] = F1D_1;
    // XSUM: the ring carries f1(kc) - f1(kb) and f1 + f1(kb) of the column (the cross terms of :573 / :582 take
    // differences of such sums over neighbouring columns: 2 operations each instead of 3)
    if constexpr (XSUM) {
      S.F1U[C1] = F1U_1 - F1D_1;
      S.SF[C1] = F1D_1 + f1_1;
    } else {
      S.F1U[C1] = F1U_1;
    }
    S.MX0[C1] = MX0_1;
    S.MN0[C1] = MN0_1;
    // :521-522 for column q without its f(ic) term
    if constexpr (!XNEW) {
      S.PMX[C0] = dmax(dmax(dmax(F0p, f0d), f0u), f0q);
      S.PMN[C0] = dmin(dmin(dmin(F0p, f0d), f0u), f0q);
    } else {
      S.PMX[C0] = dmax(dmax(f0d, f0u), G1mx);   // A of column q
      S.PMN[C0] = dmin(dmin(f0d, f0u), G1mn);
    }
    S.F0[C0] = f0q;

    // u / w sums for the antidiffusive cross terms (:573, :582), reference order
    R ud, wu;
    if constexpr (UWREF) {
      ud = x_dn[XO];
      wu = x_up[XO + T::XARR];
    } else if constexpr (!PRE) {
      ud = p_dn[LO + T::UO];
      wu = p_up[LO + T::WO];
    } else {
      ud = in.ud; wu = in.wu;
    }
#ifdef MPDATA_FAST_DIV
    G.UD[C0] = uq + ud;                        // (the ring holds the pair sum here)
    G.SU[C1] = G.UD[C1] + G.UD[C0];
    G.PW[C0] = wq + wu;
    G.SW[C0] = G.PW[C1] + G.PW[C0];
#else
    G.SU[C1] = G.UD[C1] + G.UR[C1] + uq + ud;  // u(i,kb)+u(i,k)+u(ic,k)+u(ic,kb), i = q-1
    G.UD[C0] = ud;
    G.SW[C0] = G.PW[C1] + wq + wu;             // w(ib,k)+w(ib,kc)+w(i,k)+w(i,kc), i = q
    G.PW[C0] = wq + wu;
#endif
    G.UR[C0] = uq;
#ifdef MPDATA_FAST_DIV
    // UWREF: the ring holds w = 0 at level 1 instead of a zero per-lane constant KW (www(:,:,:,1) = 0, :586:
    // both parts of W2 are then exact zeros there)
    if constexpr (UWREF) G.WR[C0] = k_is_1 ? R(0) : wq;
    else G.WR[C0] = wq;
#else
    G.WR[C0] = wq;
#endif

#ifdef MPDWM_ABL_FIRSTPASS  // timing ablation only (wrong results): stop after the first (upwind) pass
    {
      const bool act = q - 3 >= -1 && q - 3 <= nx + 2;
      const V vs = f1_1 + S.F1[C2] + S.MX0[C2] + S.MN0[C2] + S1 + (G.SU[C1] + G.SW[C0]);
      if constexpr (decltype(h_tag)::value == (PRE ? 0 : 1)) {
        v_def = vs; act_def = FULL || act; c_def = max(q - 1, 0);
      } else {
        st_col(FULL || act, max(q - 1, 0), vs, std::integral_constant<int, 0>{});
      }
      return;
    }
#endif
    // ================= stage B/C ===============================================
    V U2_1 = ZV, U2p_1 = ZV, U2n_1 = ZV, W2_2 = ZV, W2p = ZV, W2n = ZV, MXN_2 = ZV, MNN_2 = ZV;
    if (FULL || (q >= 1 && q <= nx + 3)) {
      {  // :571-573, column q-1
#ifdef MPDATA_FAST_DIV
        const R u1 = G.UR[C1];
        const R t1 = rabs(u1) - (u1 * u1) * IRHO;
        V x4;
        if constexpr (XSUM) x4 = S.F1U[C2] + S.F1U[C1];   // (the ring holds f1(kc) - f1(kb))
        else x4 = S.F1U[C2] + F1U_1 - S.F1D[C2] - F1D_1;
        U2_1 = t1 * (f1_1 - S.F1[C2]) - (KU * (u1 * G.SW[C1])) * x4;   // = 2 x (:571-573)
#else
        const V ad = andiff_s(S.F1[C2], f1_1, G.UR[C1], IRHO);
        const V x = rldexp(IADZ * (S.F1U[C2] + F1U_1 - S.F1D[C2] - F1D_1), dd_exp);
        U2_1 = ad - across_s(x, G.UR[C1], G.SW[C1]) * IRHO;
#endif
        U2p_1 = pp(U2_1);
        U2n_1 = pn(U2_1);
      }
      if (FULL || q >= 2) {  // column q-2
        {  // :580-582, :586
#ifdef MPDATA_FAST_DIV
          const R w2 = G.WR[C2];
          const R t1 = rabs(w2) - (w2 * w2) * IRHOW;
          V x4;
          if constexpr (XSUM) x4 = S.SF[C1] - S.SF[C3];   // (the ring holds f1 + f1(kb))
          else x4 = F1D_1 + f1_1 - S.F1D[C3] - S.F1[C3];
          // (0.0625 on the non-constant factor: exact scaling, and no per-lane constant for the compiler to keep)
          if constexpr (UWREF) W2_2 = t1 * (S.F1[C2] - S.F1D[C2]) - (IRHO * (w2 * G.SU[C2])) * (R(0.0625) * x4);
          else W2_2 = t1 * (S.F1[C2] - S.F1D[C2]) - (KW * (w2 * G.SU[C2])) * x4;  // = 2 x (:580-582); k = 1: 0
#else
          const V ad = andiff_s(S.F1D[C2], S.F1[C2], G.WR[C2], IRHOW);
          const V x = F1D_1 + f1_1 - S.F1D[C3] - S.F1[C3];
          const V v = ad - across_s(x, G.WR[C2], G.SU[C2]) * IRHO;
          W2_2 = sel(k_is_1, ZV, v);  // www(:,:,:,1) = 0 (:586)
#endif
        }
        const V W2u = UP_C(W2_2);
        // :596-597
        V mx1, mn1;
        if constexpr (!XNEW) {
          mx1 = dmax(dmax(dmax(dmax(dmax(S.F1[C3], f1_1), S.F1D[C2]), S.F1U[C2]), S.F1[C2]), S.MX0[C2]);
          mn1 = dmin(dmin(dmin(dmin(dmin(S.F1[C3], f1_1), S.F1D[C2]), S.F1U[C2]), S.F1[C2]), S.MN0[C2]);
        } else {
          mx1 = dmax(S.MX0[C2], G1mx);   // star of column q-2: P(q-2) and G(q-1)
          mn1 = dmin(S.MN0[C2], G1mn);
        }
        // :606-609
        W2p = pp(W2_2);
        W2n = pn(W2_2);
        V den_mx, den_mn;
#ifdef MPDATA_FAST_DIV
        if constexpr (CHAIN) {   // (the horizontal part does not wait for W2u)
          den_mx = IADZ * (pn(W2u) + W2p) + ((U2n_1 + S.U2P[C2]) + EPS_D);
          den_mn = IADZ * (pp(W2u) + W2n) + ((U2p_1 + S.U2N[C2]) + EPS_D);
        } else
#endif
        {
          den_mx = U2n_1 + S.U2P[C2] + IADZ * (pn(W2u) + W2p) + EPS_D;
          den_mn = U2p_1 + S.U2N[C2] + IADZ * (pp(W2u) + W2n) + EPS_D;
        }
#ifdef MPDATA_FAST_DIV
        {  // one reciprocal for both ratios (both denominators >= eps > 0, finite)
          const V dd2 = den_mx * den_mn;
          V r;
          if constexpr (T1X || ESUM) r = recip_nr(dd2 * IRHO);   // (no rho register)
          else r = RHO * recip_nr(dd2);   // (rho once for both ratios)
          if constexpr (CHAIN) {   // (one multiplication behind the reciprocal instead of two)
            MXN_2 = ((mx1 - S.F1[C2]) * den_mn) * r;
            MNN_2 = ((S.F1[C2] - mn1) * den_mx) * r;
          } else {
            MXN_2 = (mx1 - S.F1[C2]) * (den_mn * r);
            MNN_2 = (S.F1[C2] - mn1) * (den_mx * r);
          }
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
    V U3_2 = ZV, DW3_2 = ZV;
    if (FULL || (q >= 3 && q <= nx + 3)) {
      U3_2 = S.U2P[C2] * dmin(MXN_2, S.MNN[C3]) - S.U2N[C2] * dmin(S.MXN[C3], MNN_2);  // :618
      if (FULL || q <= nx + 2) {
        const V mxd = DN_C(MXN_2);
        const V mnd = DN_C(MNN_2);
        const V W3 = W2p * dmin(MXN_2, mnd) - W2n * dmin(mxd, MNN_2);  // :623
        if constexpr (REG_PARK) {
          // column i = q - 2 = 6 * trip + M - 4 with M = (q + 2) mod 6 known at compile time (from the register-ring
          // phase and the column of the pair): a wave-uniform branch on the trip number picks the register
          constexpr int M = (PH * 4 + decltype(h_tag)::value * 3) % 6;
          const int trip = (q + 2 - M) / 6;
          v2::pk_put<6, M - 5>(PK, trip, W3, std::make_integer_sequence<int, NPK / 6 + 1>{});
        } else if (CAN_PARK && park) park_st(q - 2, W3);   // (EXACT, bit-identical flux: added behind the march, in order)
        else if constexpr (UWREF || T1X) S1 = S1 + W3;  // one accumulator (two registers that kernel does not have)
        else S3 = S3 + W3;  // :624
        DW3_2 = UP_G(W3) - W3;
      }
    }
    {
      const int n = q - 3;  // column finished in this step: stored straight from the register
      const bool act = n >= -1 && n <= nx + 2;
      V v = S.F1[C3];  // halo columns keep the first-pass value (:557)
      if (FULL || (n >= 1 && n <= nx))
        v = ESUM ? dmax(ZV, S.F1[C3] - (U3_2 + S.DW3[C3]) * IRHO)
                 : dmax(ZV, S.F1[C3] - ((U3_2 - S.U3[C3]) + S.DW3[C3] * IADZ) * IRHO);  // :634
      // the odd column's store waits until the next pair has been waited for: the counted wait
      // must see every store issued after the pair's DMA complete, and a store issued right
      // before it would stall it for a store round trip (+1 % one tracer, +3.5 % tracer batches)
      // (two tracers per wave: the wait sits between the two steps of a pair, and it is the even
      //  column's store that is held back past it)
      if constexpr (decltype(h_tag)::value == (PRE ? 0 : 1)) {
        v_def = v; act_def = FULL || act; c_def = max(n + 2, 0);
      } else {
        st_col(FULL || act, max(n + 2, 0), v, std::integral_constant<int, 0>{});
      }
    }
    if constexpr (ESUM) {
      S.DW3[C2] = DW3_2 * IADZ - U3_2;
    } else {
      S.U3[C2] = U3_2;
      S.DW3[C2] = DW3_2;
    }
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
  //   fill (q = -2 .. 3) and drain: wave-uniform conditions; the drain ends with the pair that holds
  //   column q = nx+3 (a step beyond it, the second of that pair for odd nx, does nothing);
  //   steady-state trips (4 <= q, q+5 <= nx): every stage active, no conditions.
  const int q_last = nx + 3;
  // per pair: wait for it (two newer pairs: STREAM 2 x 6 DMA instructions, else 2 x 3), two column steps,
  // fetch pair P+3 into the slot just read
  auto dma_p = [&](const int q) __attribute__((always_inline)) { dma_pair(((q + 1) >> 1) + 3); };  // q: odd column of the pair
  auto dma_f = [&](const int q) __attribute__((always_inline)) {
    const int P = ((q + 1) >> 1) + 3;
    if (2 * P + 1 <= nx + 4) dma_pair_full(P); else dma_pair(P);
  };
// The counts are INSTRUCTION counts: a pair fetch of an array issues one DMA instruction for the lanes of
// the main part and one for the other lanes (hipcc keeps an s_cbranch_execz around each, so an empty lane
// set would issue none) -- both sets are non-empty because 128 <= mainB <= 384, which plan_create asserts
// (mpdata_plan.hip).
// The counted waits only rely on LOADS returning in issue order: "at most as many operations
// outstanding as DMA instructions were issued after the pair's own".  (Counting the column stores
// issued in between as well -- vmcnt(10) / vmcnt(5) -- is a race: a store may be acknowledged
// before an older load has landed, and the count then drops below the mark with a DMA of the
// pair still in flight; seen as sporadic wrong columns with default-policy stores.)
#define MPDWM_FLUSH_DEFERRED st_col(act_def, c_def, v_def, std::integral_constant<int, 0>{});
// nz > 64: the waves of an instance read each other's OUTPUT levels as the halo levels of their own windows, and f is
// updated in place -- a wave may store a column only when its siblings have FETCHED it.  The waves of an instance
// are one workgroup; the barrier behind the counted wait of a pair says "everybody's fetch of this pair has landed",
// and every store issued up to the next barrier is of a column at least one pair older.  (Found in round 5 when
// 4096 x 32 x 72 failed under load: with 130 instances the siblings had stayed within a pair of each other by luck.)
#define MPDWM_KS_BARRIER                  \
  if constexpr (KS) {                     \
    __builtin_amdgcn_s_barrier();         \
    asm volatile("" ::: "memory");        \
  }
#define MPDWM_PAIR(PHA, PHB, SL, SLN, TAG, q, DMA)      \
  if constexpr (UWREF) {                                \
    /* own f and own share of u, w of this pair have landed (newer loads: f, uw of the next  \
       pair and f of the one after: 3 x 2 instructions), every LDS read of the previous pair \
       has returned; after the barrier the same holds for the whole workgroup: the pair's u, \
       w are complete and the slot of the previous pair is free for the pair after next */   \
    MPDWM_WAIT("s_waitcnt vmcnt(6) lgkmcnt(0)")         \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
    MPDWM_FLUSH_DEFERRED                                \
    step(PHA{}, SL{}, I0{}, TAG{}, (q), in_e);          \
    asm volatile("" ::: "memory");                      \
    dma_uw((((q) + 2) >> 1) + 2);   /* (between the two steps; behind the barrier or behind the pair: no better) */ \
    step(PHB{}, SL{}, I1{}, TAG{}, (q) + 1, in_e);      \
    asm volatile("" ::: "memory");                      \
    DMA((q) + 1);                                       \
  } else if constexpr (!PRE) {                          \
    if constexpr (STREAM) MPDWM_WAIT("s_waitcnt vmcnt(12)") \
    else MPDWM_WAIT("s_waitcnt vmcnt(6)")               \
    MPDWM_KS_BARRIER                                    \
    MPDWM_FLUSH_DEFERRED                                \
    step(PHA{}, SL{}, I0{}, TAG{}, (q), in_e);          \
    asm volatile("" ::: "memory");                      \
    step(PHB{}, SL{}, I1{}, TAG{}, (q) + 1, in_e);      \
    asm volatile("" ::: "memory");                      \
    DMA((q) + 1);                                       \
  } else {                                              \
    const LdsIn in_o = lds_in(SL{}, I1{});              \
    step(PHA{}, SL{}, I0{}, TAG{}, (q), in_e);          \
    if constexpr (DMA_PER_PAIR == 4) MPDWM_WAIT("s_waitcnt vmcnt(4)")      \
    else if constexpr (DMA_PER_PAIR == 3) MPDWM_WAIT("s_waitcnt vmcnt(3)") \
    else MPDWM_WAIT("s_waitcnt vmcnt(6)")               \
    MPDWM_KS_BARRIER                                    \
    MPDWM_FLUSH_DEFERRED                                \
    in_e = lds_in(SLN{}, I0{});                         \
    step(PHB{}, SL{}, I1{}, TAG{}, (q) + 1, in_o);      \
    asm volatile("" ::: "memory");                      \
    DMA((q) + 1);                                       \
  }
  // Two tracers per wave: the LDS inputs of a step are read one step ahead (in_o at the top of the
  // pair, the next pair's in_e in its middle, behind the wait for that pair: the only DMA group
  // issued after it is the one of the pair after next -- 4 instructions).
  LdsIn in_e;
  if constexpr (PRE) {   // pair 0 (pairs 1, 2 were issued after it: two groups of DMA_PER_PAIR instructions)
    if constexpr (DMA_PER_PAIR == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (DMA_PER_PAIR == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    in_e = lds_in(I0{}, I0{});
  }
  int q0 = -2;
  {  // fill: columns -2 .. 3 (nx >= 1: all of them exist)
    MPDWM_PAIR(I0, I1, I0, I1, Part, q0, dma_p)
    MPDWM_PAIR(I2, I0, I1, I2, Part, q0 + 2, dma_p)
    MPDWM_PAIR(I1, I2, I2, I0, Part, q0 + 4, dma_p)
    q0 += 6;
  }
  for (; q0 + 5 <= nx; q0 += 6) {  // steady state
    MPDWM_PAIR(I0, I1, I0, I1, Full, q0, dma_f)
    MPDWM_PAIR(I2, I0, I1, I2, Full, q0 + 2, dma_f)
    MPDWM_PAIR(I1, I2, I2, I0, Full, q0 + 4, dma_f)
  }
  // drain: pairs beyond the last column are not run (+2 % at 25 tracers).  The two forms of the
  // same thing are what the register allocator accepts without spilling in either case.
  if constexpr (PRE) {
    for (; q0 <= q_last; q0 += 6) {
      MPDWM_PAIR(I0, I1, I0, I1, Part, q0, dma_p)
      if (q0 + 2 > q_last) break;
      MPDWM_PAIR(I2, I0, I1, I2, Part, q0 + 2, dma_p)
      if (q0 + 4 > q_last) break;
      MPDWM_PAIR(I1, I2, I2, I0, Part, q0 + 4, dma_p)
    }
  } else {
    for (; q0 + 5 <= q_last; q0 += 6) {   // whole trips, then the pairs that are left (up to three:
      MPDWM_PAIR(I0, I1, I0, I1, Part, q0, dma_p)   // five columns = two pairs and a half)
      MPDWM_PAIR(I2, I0, I1, I2, Part, q0 + 2, dma_p)
      MPDWM_PAIR(I1, I2, I2, I0, Part, q0 + 4, dma_p)
    }
    if (q0 <= q_last) {
      MPDWM_PAIR(I0, I1, I0, I1, Part, q0, dma_p)
      if (q0 + 2 <= q_last) {
        MPDWM_PAIR(I2, I0, I1, I2, Part, q0 + 2, dma_p)
        if (q0 + 4 <= q_last) {
          MPDWM_PAIR(I1, I2, I2, I0, Part, q0 + 4, dma_p)
        }
      }
    }
  }
#undef MPDWM_PAIR
#undef MPDWM_FLUSH_DEFERRED
#undef MPDWM_KS_BARRIER

  if constexpr (!PRE) st_col(act_def, c_def, v_def, std::integral_constant<int, 0>{});
  {  // flux (:541-547, :624)
    V fl = ((UWREF || T1X) || (CAN_PARK && park) || REG_PARK) ? S1 : S1 + S3;   // (parked: the upwind sum alone)
    if constexpr (REG_PARK) {   // ... + www(1) + www(2) + ... + www(nx), one by one (:624)
#pragma unroll
      for (int i = 0; i < NPK; ++i)
        if (i < nx) fl = fl + PK[i];
    }
    int posf = pos;
    bool okf = out_ok;
    if constexpr ((T1X || UWX2) && !KS) {   // the lane's element, formed again behind the march: no register carries it through
      unsigned z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      const int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
      const int kk2 = l2 % LW;
      okf = kk2 < nzm;
      posf = (l2 / LW) * nzm + (okf ? kk2 : nzm - 1);
    }
    if constexpr (KT) {   // tail waves: tile and element of the lane formed again behind the march (three registers the
                          // 66-column register park does not have)
      unsigned z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      const int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
      const int s2 = l2 / LW, k2 = l2 % LW + koff + 1;
      const bool ok2 = k2 <= nzm && s2 < ntl && k2 >= 58 * kwave + 4;   // (= out_ok: the tail is an instance's LAST window)
      R* const fl2 = a.flux + (long long)tr * a.flux_tstride + ((long long)tile + (s2 < ntl ? s2 : 0)) * chunk;
      if (ok2) fl2[k2 - 1] = first(fl);
      if constexpr (TPW == 2)
        if (ok2 && has1) fl2[a.flux_tstride + (k2 - 1)] = second(fl);
    } else {
      if (okf && tile_ok) flux[posf] = first(fl);
      if constexpr (TPW == 2)
        if (out_ok && has1 && tile_ok) flux1[pos] = second(fl);
    }
  }
#ifdef MPDWM_STAMPS
  if (a.dbg) {
    unsigned long long r1, c1;
    unsigned hw, xcc;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_memtime %1\n\ts_getreg_b32 %2, hwreg(HW_REG_HW_ID)\n\t"
                 "s_getreg_b32 %3, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(r1), "=s"(c1), "=s"(hw), "=s"(xcc)::"memory");
    unsigned long long* o = a.dbg + 8ull * ((unsigned long long)blockIdx.x * WPB + (unsigned)wave);
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
      o[0] = st_r0; o[1] = r1; o[2] = st_c0; o[3] = c1; o[4] = st_wait; o[5] = hw; o[6] = xcc; o[7] = tile;
    }
  }
#endif
}
#undef MPDWM_WAIT

// ---- wave -> (tile, first tracer).  Workgroups are dealt to the 8 XCDs round-robin in dispatch
//      order.  Tracer batches: the waves that an XCD receives walk through the tracers of one
//      tile after the other, so the waves that share a tile's u, w, rho, ... run
//      back to back on ONE XCD and all but the first find them in that XCD's L2.
// nkw (nz > 64 only, else 1): waves per instance and tracer; kwave = which of them this wave is
template <int WPB, int TPW>
__device__ __forceinline__ void wm_wave_to_tile(const unsigned ntr, const int wave, unsigned& tile, unsigned& tr,
                                                const unsigned nkw = 1, int* kwave = nullptr) {
  if (ntr == 1 && TPW == 1) {
    const unsigned gw = blockIdx.x * WPB + wave;
    tile = gw / nkw;
    if (kwave) *kwave = (int)(gw % nkw);
    tr = 0;
  } else {
    const unsigned nxcd = 8, ntw = ((ntr + TPW - 1) / TPW) * nkw;   // waves per tile
    const unsigned v = (blockIdx.x / nxcd) * WPB + wave;  // position in the XCD's wave sequence
    const unsigned r = v % ntw;
    tr = (r / nkw) * TPW;                                 // first tracer of the wave
    if (kwave) *kwave = (int)(r % nkw);
    tile = (v / ntw) * nxcd + blockIdx.x % nxcd;
  }
}

template <typename R, int LPS, int WPB, bool STREAM, int TPW = 1, bool UWREF = false, bool UWCONV = false, int NPK = 0>
__global__ void __launch_bounds__(64 * WPB, (NPK > 0 ? 2 : TileWm<R, LPS, WPB, TPW, UWREF>::MIN_WAVES))
mpdata_advect_wm_kernel(const MpdataWmArgsT<R> a) {
  using T = TileWm<R, LPS, WPB, TPW, UWREF>;
  __shared__ R lds[T::LDS_ELEMS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntr = (unsigned)a.ntracers;
  unsigned tile, tr;
  int kwave = 0;
  if constexpr (T::KS) {
    // a workgroup = the nkw waves of each of its ipw = max(1, WPB / nkw) instances (they synchronise: MPDWM_KS_BARRIER);
    // launched with 64 * ipw * nkw threads.  Batches: the workgroups an XCD receives walk through the tracer slots of
    // one group of ipw instances after the other (u, w of the group stay in that XCD's L2)
    const unsigned nkw = (unsigned)a.nkw, ipw = WPB / nkw > 0 ? WPB / nkw : 1u;
    const unsigned ngrp = ((unsigned)a.ntiles + ipw - 1) / ipw;
    kwave = (int)((unsigned)wave % nkw);
    unsigned grp;
    if (ntr == 1 && TPW == 1) {
      grp = blockIdx.x;
      tr = 0;
    } else {
      const unsigned nxcd = 8, ntw = (ntr + TPW - 1) / TPW;
      const unsigned v = blockIdx.x / nxcd;
      tr = (v % ntw) * TPW;
      grp = (v / ntw) * nxcd + blockIdx.x % nxcd;
    }
    if (grp >= ngrp) return;   // (the whole workgroup: before any barrier)
    tile = grp * ipw + (unsigned)wave / nkw;
  } else {
    wm_wave_to_tile<WPB, TPW>(ntr, wave, tile, tr);
  }
  // UWREF: the waves of a workgroup share the u, w ring (barriers, a share of the row fetches each):
  // a wave beyond the last tile stays, works on an EMPTY f range (fetches deliver zeros, stores are
  // dropped) and reads the constants of the last tile
  bool tile_ok = true;
  if constexpr (UWREF || T::KS) {
    tile_ok = tile < (unsigned)a.ntiles;
    if (!tile_ok) tile = (unsigned)a.ntiles - 1u;
    if constexpr (T::KS)
      if (a.reverse && tile_ok) tile = (unsigned)a.ntiles - 1u - tile;
  } else {
    if (tile >= (unsigned)a.ntiles) return;  // (no barrier anywhere below)
    // serpentine: every other run of a plan walks the tiles from the other end, so that it starts
    // on what the previous run touched last (u, w and the like are still in the Infinity Cache)
    if (a.reverse) tile = (unsigned)a.ntiles - 1u - tile;
  }
  wm_body<R, LPS, WPB, STREAM, TPW, UWREF, UWCONV, NPK>(a, lds, lds + wave * (T::NS * T::SLOT), wave, tile, tr, ntr, tile_ok, kwave);
}

// nz > 64 with a SHORT last window (round 5): the last of an instance's two windows needs nz - 58 levels; where that is at most
// LWT = 16 / 32 (nz <= 74 / 90), "tail" waves hold the last windows of 64 / LWT adjacent instances each (body form
// LPS = 128 + LWT) beside the instances' full-width waves.  A workgroup = a.ksg instances = ksg full-width waves +
// ceil(ksg LWT / 64) tail waves, which meet in the barrier of the column pairs like the waves of the plain form.  ksg is
// 3 at LWT = 16 (3 + 1 = FOUR waves: 85 lanes per instance instead of 128) and 2 at LWT = 32 (2 + 1 waves: 96 lanes).
// With 4 + 1 waves at LWT = 16 only ONE workgroup was ever resident on a CU (4.8 waves per CU on average) and the form
// was no faster than the plain one at 38 % fewer instructions: a workgroup takes ceil(waves / 4) wave slots on EVERY
// SIMD of its CU (tools/wg_residency.hip, profiles/r05_wg_residency.txt), so five to eight waves cost two slots per
// SIMD -- of the three a SIMD has at more than 128 registers.  Hence 3 + 1; 5 + 3 = eight waves at LWT = 32 (one workgroup
// per CU) and 4 + 2 lost against 2 + 1.
// Three waves per SIMD (130 registers, FAST): at four (128 registers, 20 bytes of scratch) equal to 1 % slower.
// LDS: a ring per wave, sized by the launch.
template <typename R, int LWT, int TPW = 1, int NPK = 0>
__global__ void __launch_bounds__(64 * 4, ((TPW == 2 || NPK > 0) ? 2 : 3))
mpdata_advect_wm_ks2_kernel(const MpdataWmArgsT<R> a) {
  using TM = TileWm<R, 128, 4, TPW, false>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  R* const lds = reinterpret_cast<R*>(lds_raw);
  constexpr unsigned TP = 64 / LWT;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntr = (unsigned)a.ntracers, G = (unsigned)a.ksg;
  const unsigned ngrp = ((unsigned)a.ntiles + G - 1) / G;
  unsigned grp, tr;
  if (ntr == 1 && TPW == 1) {
    grp = blockIdx.x;
    tr = 0;
  } else {   // (as the plain form: the workgroups an XCD receives walk through the tracer slots of one group after the other)
    const unsigned nxcd = 8, ntw = (ntr + TPW - 1) / TPW;
    const unsigned v = blockIdx.x / nxcd;
    tr = (v % ntw) * TPW;
    grp = (v / ntw) * nxcd + blockIdx.x % nxcd;
  }
  if (grp >= ngrp) return;   // (the whole workgroup: before any barrier)
  if (a.reverse) grp = ngrp - 1u - grp;
  R* const my = lds + wave * (TM::NS * TM::SLOT);
  if ((unsigned)wave < G) {
    unsigned tile = grp * G + (unsigned)wave;
    const bool tile_ok = tile < (unsigned)a.ntiles;
    if (!tile_ok) tile = (unsigned)a.ntiles - 1u;
    wm_body<R, 128, 4, false, TPW, false, false, NPK>(a, lds, my, wave, tile, tr, ntr, tile_ok, 0);
  } else {
    const unsigned first = ((unsigned)wave - G) * TP;   // first instance of the group in this tail wave
    unsigned tile = grp * G + first;
    const int ntl = max(0, min(min((int)TP, (int)G - (int)first), a.ntiles - (int)tile));
    if (ntl == 0) tile = (unsigned)a.ntiles - 1u;
    wm_body<R, 128 + LWT, 4, false, TPW, false, false, NPK>(a, lds, my, wave, tile, tr, ntr, ntl > 0, 1, ntl);
  }
}

// Tracer batches with an ODD number of tracers, one launch (round 5): the waves of a tile are its tracer pairs
// (the two-tracer form) and, last, ONE wave that takes the odd tracer through the one-tracer batch form -- a
// wave-uniform branch at kernel entry, 128-register code inside the 256-register allocation of the launch.  The odd
// tracer runs beside its tile's pairs and finds u, w in its XCD's L2; as a launch of its own (rounds 2-4) it was 4 %
// of the work for 5.4 % of the time and fetched u, w from HBM once more (traffic 1.056 x).  A two-tracer wave with an
// empty second half in its place (MPDATA_WM_NOSPLIT) costs as much as a full one.
template <typename R, int LPS, int WPB>
__global__ void __launch_bounds__(64 * WPB, 2)
mpdata_advect_wm_odd_kernel(const MpdataWmArgsT<R> a) {
  using T2 = TileWm<R, LPS, WPB, 2, false>;
  __shared__ R lds[T2::LDS_ELEMS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned ntr = (unsigned)a.ntracers;
  unsigned tile, tr;
  wm_wave_to_tile<WPB, 2>(ntr, wave, tile, tr);
  if (tile >= (unsigned)a.ntiles) return;
  if (a.reverse) tile = (unsigned)a.ntiles - 1u - tile;
  R* const my = lds + wave * (T2::NS * T2::SLOT);
  if (tr + 1 < ntr) wm_body<R, LPS, WPB, false, 2>(a, lds, my, wave, tile, tr, ntr, true);
  else wm_body<R, LPS, WPB, false, 1>(a, lds, my, wave, tile, tr, ntr, true);
}

}  // namespace wm
}  // namespace MPDATA_NS
