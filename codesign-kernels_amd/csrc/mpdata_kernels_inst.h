// mpdata_kernels_inst.h -- instantiates the tilings of one arithmetic variant.
// Included by mpdata_kernels_exact.hip / mpdata_kernels_fast.hip after
// defining MPDATA_NS; exports <MPDATA_NS>::launch(tile, ...).
#include <cstdlib>
#include "mpdata_kernel_body.h"
#include "mpdata_kernel_v2_body.h"
#include "mpdata_kernel_wm_body.h"

// One variant is compiled as FOUR translation units (make -j: the kernels are the build time): -DMPDATA_PART=
//   0  the kernels of the calls on reference-layout arrays (x-march, k-march), tile table, build flags
//   1  the plan kernels for nz <= 64 (launch_wm, launch_wm_f32)
//   2  the plan kernels for nz > 64 (window form, tail form)
//   3  the plan kernels that read u, w from the reference layout (launch_wm_uw)
// (no MPDATA_PART: everything in one unit)
#ifndef MPDATA_PART
#define MPDATA_PART -1
#endif
#define MPD_PART(n) (MPDATA_PART < 0 || MPDATA_PART == (n))

namespace MPDATA_NS {

// nz > 64: defined in part 2, called from launch_wm / launch_wm_f32 (part 1)
void launch_wm_ks_f64(const MpdataWmArgsT<double>& a, void* stream);
void launch_wm_ks_f32(const MpdataWmArgsT<v2::f32x2>& a, void* stream);

#if MPD_PART(0)
template <int W, int SPW, int NWV>
static void launch_tile(const MpdataArgs& a, int ntracers, void* stream) {
  using T = Tile<W, SPW, NWV>;
  const unsigned gx = (unsigned)((a.ncrms + T::SLW - 1) / T::SLW);
  dim3 grid(gx, (unsigned)ntracers, 1), block(T::THREADS, 1, 1);
  hipLaunchKernelGGL((mpdata_advect_kernel<W, SPW, NWV>), grid, block, 0, (hipStream_t)stream, a);
}

// id, W, SPW, NWV
#define MPDATA_TILES(X) \
  X(0, 9, 4, 1)         \
  X(1, 9, 1, 4)         \
  X(2, 9, 2, 2)         \
  X(3, 9, 4, 2)         \
  X(4, 9, 4, 4)         \
  X(5, 3, 4, 3)         \
  X(6, 1, 4, 9)         \
  X(7, 2, 2, 9)         \
  X(8, 6, 2, 3)         \
  X(9, 3, 2, 6)         \
  X(10, 3, 1, 12)

// x-marching kernels (mpdata_kernel_v2_body.h): id, LPS (lanes per instance >= nz),
// G (instances per workgroup: 128- or 256-byte row segments)
#define MPDATA_TILES_V2(X) \
  X(20, 8, 16)             \
  X(21, 16, 16)            \
  X(22, 32, 16)            \
  X(23, 64, 16)            \
  X(24, 32, 32)

// fp32 x-marching kernels, one instance per lane: 32 instances per workgroup = 128-byte rows
// (any ncrms; 16 waves per workgroup at LPS = 32)
#define MPDATA_TILES_V2_F32(X) \
  X(30, 8, 32)                 \
  X(31, 16, 32)                \
  X(32, 32, 32)
// fp32 x-marching kernels, two adjacent instances per lane (packed fp32 arithmetic): id, LPS,
// G in PAIRS (16 pairs = 32 instances = 128-byte rows).  Even ncrms only.
#define MPDATA_TILES_V2_F32X2(X) \
  X(40, 8, 16)                   \
  X(41, 16, 16)                  \
  X(42, 32, 16)                  \
  X(43, 64, 16)

template <typename R, int LPS, int G>
static void launch_tile_v2(const MpdataArgsT<R>& a, int ntracers, void* stream) {
  using T = v2::TileV2<R, LPS, G>;
  const unsigned groups = (unsigned)((a.ncrms + G - 1) / G);
  MpdataArgsT<R> b = a;
  b.ntracers = ntracers;
  dim3 grid((unsigned)ntracers * groups, 1, 1), block(T::THREADS, 1, 1);  // tracer fastest
  // arrays of 4 GiB or more: the instantiation with per-wave descriptor bases
  const bool big = (double)a.ncrms * (a.nx + 6) * a.nz * (double)sizeof(R) >= 4294967000.0;
#ifndef MPDATA_FAST_DIV
  // EXACT, bit-identical flux from registers (workgroups of up to 8 waves: a 16-wave workgroup -- nz 33 .. 64, the
  // 256-byte-row tiling -- caps a wave at 128 registers; those keep the park array)
  // (arrays of 4 GiB and more: the instantiation with per-wave descriptor bases has no scalar registers for a park
  //  DESCRIPTOR -- the register park needs none: bit-identical flux there as well since round 5)
  if constexpr (G * LPS <= 512) {
    if (b.park_regs && b.nx <= MPDATA_WM_NPK && !b.wpark) {
      if (big)
        hipLaunchKernelGGL((v2::mpdata_advect_xmarch_kernel<R, LPS, G, true, false, MPDATA_WM_NPK>), grid, block, 0,
                           (hipStream_t)stream, b);
      else
        hipLaunchKernelGGL((v2::mpdata_advect_xmarch_kernel<R, LPS, G, false, false, MPDATA_WM_NPK>), grid, block, 0,
                           (hipStream_t)stream, b);
      return;
    }
  }
#endif
  if (big)
    hipLaunchKernelGGL((v2::mpdata_advect_xmarch_kernel<R, LPS, G, true>), grid, block, 0, (hipStream_t)stream, b);
  else if (std::is_same<R, double>::value && G == 16 && ntracers == 1)   // one tracer: streaming rows (+1..3 %)
    hipLaunchKernelGGL((v2::mpdata_advect_xmarch_kernel<R, LPS, G, false, true>), grid, block, 0, (hipStream_t)stream, b);
  else
    hipLaunchKernelGGL((v2::mpdata_advect_xmarch_kernel<R, LPS, G, false>), grid, block, 0, (hipStream_t)stream, b);
#ifndef MPDATA_FAST_DIV
  // EXACT with a park array: the limited vertical fluxes onto the upwind sum, in the reference's order (bit-identical flux)
  if (b.wpark && !big) hipLaunchKernelGGL((v2::xmarch_flux_finish_kernel<R, LPS, G>), grid, block, 0, (hipStream_t)stream, b);
#endif
}

#endif   // part 0

// wave-major kernels (mpdata_kernel_wm_body.h, the plan API): LPS = lanes per instance (>= nz),
// WPB = waves per workgroup.  Returns false if (lps, wpb) is not instantiated.
#define MPDATA_WM_LPS(X) X(8) X(16) X(32) X(64)
// waves (= tiles) per workgroup of the plan kernels; the waves of a workgroup never synchronise, the workgroup is
// only the unit in which the dispatcher hands out wave slots and LDS (1, 2, 4, 8 measured alike, profiles/r04_ablation.json)
#define MPDWM_WPB 4
#if MPD_PART(1)
template <typename R, int LPS, int WPB>
static void launch_wm_t(const MpdataWmArgsT<R>& a, void* stream, int flags) {
  // ntracers == 1: wave = tile in dispatch order, u and w streamed; else the per-XCD tracer walk with
  // u, w kept in L2 (see the kernel)
  // MPDATA_WM_NOSTREAM (tests): run the batch form of the kernel on a single tracer as well
  // flags: MPDATA_WMF_* test switches (mpdata_args.h; mpdata_set_wm_flags / the MPDATA_WM_*
  // environment variables, read once by the C-ABI layer -- not in this timed launch path)
  const bool no_stream = flags & MPDATA_WMF_NOSTREAM, tpw1 = flags & MPDATA_WMF_TPW1, no_split = flags & MPDATA_WMF_NOSPLIT;
#ifndef MPDATA_FAST_DIV
  // EXACT, bit-identical flux without a park array (round 5): the nx limited vertical fluxes of a lane stay in
  // REGISTERS (one tracer per wave, 2 waves per SIMD) and are added onto the finished upwind sum behind the march
  if (a.park_regs && a.nx <= MPDATA_WM_NPK2 && !a.wpark) {
    const bool small = a.nx <= MPDATA_WM_NPK;   // (two instantiations: 36 columns at 187 registers, 66 at 247)
    if (a.ntracers == 1 && !no_stream) {
      const unsigned blocks = (unsigned)((a.ntiles + WPB - 1) / WPB);
      if (small)
        hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, true, 1, false, false, MPDATA_WM_NPK>), dim3(blocks),
                           dim3(64 * WPB), 0, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, true, 1, false, false, MPDATA_WM_NPK2>), dim3(blocks),
                           dim3(64 * WPB), 0, (hipStream_t)stream, a);
    } else {   // tracer batches: one tracer per wave, the per-XCD tracer walk
      const long long per_xcd = ((long long)(a.ntiles + 7) / 8) * a.ntracers;
      const unsigned blocks = (unsigned)(8 * ((per_xcd + WPB - 1) / WPB));
      if (small)
        hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, false, 1, false, false, MPDATA_WM_NPK>), dim3(blocks),
                           dim3(64 * WPB), 0, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, false, 1, false, false, MPDATA_WM_NPK2>), dim3(blocks),
                           dim3(64 * WPB), 0, (hipStream_t)stream, a);
    }
    return;
  }
#endif
  if (a.ntracers == 1 && no_stream) {
    const unsigned blocks = (unsigned)((a.ntiles + WPB - 1) / WPB);
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, false>), dim3(blocks), dim3(64 * WPB), 0,
                       (hipStream_t)stream, a);
  } else if (a.ntracers == 1) {
    const unsigned blocks = (unsigned)((a.ntiles + WPB - 1) / WPB);
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, true>), dim3(blocks), dim3(64 * WPB), 0,
                       (hipStream_t)stream, a);
  } else if (tpw1) {   // (tests / A-B: one tracer per wave)
    const long long per_xcd = ((long long)(a.ntiles + 7) / 8) * a.ntracers;  // waves of one XCD
    const unsigned blocks = (unsigned)(8 * ((per_xcd + WPB - 1) / WPB));
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, false>), dim3(blocks), dim3(64 * WPB), 0,
                       (hipStream_t)stream, a);
  } else {   // tracer batches: two tracers per wave
    // An odd count (round 5): ONE launch in which the last wave of every tile takes the odd tracer through the
    // one-tracer batch form (mpdata_advect_wm_odd_kernel).  MPDATA_WM_SPLIT (A/B): the odd tracer through the
    // one-tracer kernel BEHIND the batch, rounds 2-4 (4 % of the work, 5.4 % of the time, u and w fetched from HBM
    // once more).  MPDATA_WM_NOSPLIT (tests): the odd tracer in a two-tracer wave with an empty second half (costs
    // as much as a full one: 4 % at 25 tracers).
    MpdataWmArgsT<R> b = a;
    const bool odd = a.ntracers & 1;
    const bool split = odd && (flags & MPDATA_WMF_SPLIT) && !no_split;
    if (split) b.ntracers = a.ntracers - 1;
    const long long per_xcd = ((long long)(b.ntiles + 7) / 8) * ((b.ntracers + 1) / 2);
    const unsigned blocks = (unsigned)(8 * ((per_xcd + WPB - 1) / WPB));
    // (The one-tracer kernel of the odd tracer BESIDE the batch kernel on a second stream was measured:
    //  7.394 ms either way at 25 tracers -- two batch waves fill a SIMD's registers, so the second
    //  kernel's workgroups only get slots in the batch kernel's tail; removed again.  Round 4: a wave per TILE that
    //  walks through the tile's tracer pairs itself -- no slot-refill gap between pairs, zero spills, parity green --
    //  measured 7.58 ms against 7.32, profiles/r04_ablation.json: removed again.)
    if (odd && !split && !no_split)
      hipLaunchKernelGGL((wm::mpdata_advect_wm_odd_kernel<R, LPS, WPB>), dim3(blocks), dim3(64 * WPB), 0,
                         (hipStream_t)stream, b);
    else
      hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, false, 2>), dim3(blocks), dim3(64 * WPB), 0,
                         (hipStream_t)stream, b);
    if (split) {
      MpdataWmArgsT<R> c = a;
      c.f = a.f + (long long)(a.ntracers - 1) * a.f_tstride;
      c.flux = a.flux + (long long)(a.ntracers - 1) * a.flux_tstride;
      if (a.wpark) c.wpark = a.wpark + (long long)(a.ntracers - 1) * a.ntiles * ((long long)a.nx * 64);
      c.ntracers = 1;
      const unsigned blocks1 = (unsigned)((c.ntiles + WPB - 1) / WPB);
      hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, LPS, WPB, true>), dim3(blocks1), dim3(64 * WPB), 0,
                         (hipStream_t)stream, c);
    }
  }
}
#endif   // part 1
#if MPD_PART(2)
// nz > 64 (kernel form LPS = 128): a.nkw waves per instance and tracer, one tracer per wave, the batch form of the
// data movement; EXACT with the register park (the caller makes sure that nx <= MPDATA_WM_NPK or that no flux order
// is asked for)
template <typename R>
static void launch_wm_ks2(const MpdataWmArgsT<R>& a, void* stream);
template <typename R>
static void launch_wm_ks(const MpdataWmArgsT<R>& a, void* stream) {
  constexpr int WPB = MPDWM_WPB;
  // a short last window: the tail form (mpdata_advect_wm_ks2_kernel)
  if (a.lwt != 0) {
    launch_wm_ks2<R>(a, stream);
    return;
  }
  // a workgroup = the nkw waves of each of its ipw instances (they synchronise once per column pair, see the kernel)
  const int ipw = WPB / a.nkw > 0 ? WPB / a.nkw : 1;
  const unsigned threads = 64u * (unsigned)(ipw * a.nkw);
  const long long ngrp = ((long long)a.ntiles + ipw - 1) / ipw;
  auto blocks_for = [&](const int slots) -> unsigned {   // slots: tracer slots (waves per instance window) of the launch
    if (slots == 1) return (unsigned)ngrp;
    return (unsigned)(8 * ((ngrp + 7) / 8) * slots);
  };
#ifndef MPDATA_FAST_DIV
  if (a.park_regs && a.nx <= MPDATA_WM_NPK) {
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, 128, WPB, false, 1, false, false, MPDATA_WM_NPK>),
                       dim3(blocks_for(a.ntracers)), dim3(threads), 0, (hipStream_t)stream, a);
    return;
  }
  if (a.park_regs && a.nx <= MPDATA_WM_NPK2) {
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, 128, WPB, false, 1, false, false, MPDATA_WM_NPK2>),
                       dim3(blocks_for(a.ntracers)), dim3(threads), 0, (hipStream_t)stream, a);
    return;
  }
#endif
#ifdef MPDATA_FAST_DIV
  // FAST tracer batches: two tracers per wave here as well (an odd last tracer: a wave with an empty second half)
  if (a.ntracers >= 2) {
    hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, 128, WPB, false, 2>), dim3(blocks_for((a.ntracers + 1) / 2)),
                       dim3(threads), 0, (hipStream_t)stream, a);
    return;
  }
#endif
  hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<R, 128, WPB, false, 1>), dim3(blocks_for(a.ntracers)), dim3(threads), 0,
                     (hipStream_t)stream, a);
}
// nz > 64 with a short last window (a.lwt = 16 / 32): workgroups of G instances = G full-width waves and
// ceil(G lwt / 64) tail waves; a ring per wave in dynamic LDS
template <typename R>
static void launch_wm_ks2(const MpdataWmArgsT<R>& a_in, void* stream) {
  MpdataWmArgsT<R> a = a_in;
  // instances per workgroup (see the kernel): 3 + 1 waves at lwt = 16, 2 + 1 at lwt = 32
  const int G = a.lwt == 16 ? 3 : 2, TP = 64 / a.lwt, waves = G + (G + TP - 1) / TP;
  a.ksg = G;
  const unsigned threads = 64u * (unsigned)waves;
  const long long ngrp = ((long long)a.ntiles + G - 1) / G;
  auto blocks_for = [&](const int slots) -> unsigned {
    if (slots == 1) return (unsigned)ngrp;
    return (unsigned)(8 * ((ngrp + 7) / 8) * slots);
  };
  auto lds_for = [&](const int tpw) -> unsigned { return (unsigned)(waves * 3 * (2 + tpw) * 128 * 8); };
#define KS2_LAUNCH(TPW_, NPK_, SLOTS_)                                                                                   \
  {                                                                                                                      \
    if (a.lwt == 16)                                                                                                     \
      hipLaunchKernelGGL((wm::mpdata_advect_wm_ks2_kernel<R, 16, TPW_, NPK_>), dim3(blocks_for(SLOTS_)), dim3(threads),  \
                         lds_for(TPW_), (hipStream_t)stream, a);                                                         \
    else                                                                                                                 \
      hipLaunchKernelGGL((wm::mpdata_advect_wm_ks2_kernel<R, 32, TPW_, NPK_>), dim3(blocks_for(SLOTS_)), dim3(threads),  \
                         lds_for(TPW_), (hipStream_t)stream, a);                                                         \
    return;                                                                                                              \
  }
#ifndef MPDATA_FAST_DIV
  if (a.park_regs && a.nx <= MPDATA_WM_NPK) KS2_LAUNCH(1, MPDATA_WM_NPK, a.ntracers)
  if (a.park_regs && a.nx <= MPDATA_WM_NPK2) KS2_LAUNCH(1, MPDATA_WM_NPK2, a.ntracers)
#else
  if (a.ntracers >= 2) KS2_LAUNCH(2, 0, (a.ntracers + 1) / 2)
#endif
  KS2_LAUNCH(1, 0, a.ntracers)
#undef KS2_LAUNCH
}
void launch_wm_ks_f64(const MpdataWmArgsT<double>& a, void* stream) { launch_wm_ks<double>(a, stream); }
void launch_wm_ks_f32(const MpdataWmArgsT<v2::f32x2>& a, void* stream) { launch_wm_ks<v2::f32x2>(a, stream); }
#endif   // part 2
#if MPD_PART(1)
bool launch_wm(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags) {
  if (wpb != MPDWM_WPB) return false;
  if (lps == 128) {
    launch_wm_ks_f64(a, stream);
    return true;
  }
#define X(LPS_)                             \
  if (lps == LPS_) {                        \
    launch_wm_t<double, LPS_, MPDWM_WPB>(a, stream, flags); \
    return true;                            \
  }
  MPDATA_WM_LPS(X)
#undef X
  return false;
}
#endif   // part 1
#if MPD_PART(3)
// mpdata_plan_run_uw, one fp64 tracer: u, w read from the REFERENCE layout (a.u_ref, a.w_ref), f in
// the plan layout; workgroups of 16 adjacent instances (16 / SLP waves)
bool launch_wm_uw(int lps, const MpdataWmArgs& a, void* stream, bool conv) {
  // conv: the kernel also writes the velocities it reads into the plan's u, w (plan layout)
#define X(LPS_, CONV_OK)                                                                               \
  if (lps == LPS_) {                                                                                   \
    constexpr int WPB = LPS_ / 4;                                                                      \
    const unsigned blocks = (unsigned)((a.ntiles + WPB - 1) / WPB);                                    \
    if (conv) {                                                                                        \
      if constexpr (!CONV_OK) return false;                                                            \
      else hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<double, LPS_, WPB, true, 1, true, true>), dim3(blocks), \
                              dim3(64 * WPB), 0, (hipStream_t)stream, a);                              \
    } else {                                                                                           \
      hipLaunchKernelGGL((wm::mpdata_advect_wm_kernel<double, LPS_, WPB, true, 1, true>), dim3(blocks),  \
                         dim3(64 * WPB), 0, (hipStream_t)stream, a);                                   \
    }                                                                                                  \
    return true;                                                                                       \
  }
  // (LPS = 64, nz 33 .. 64: one 16-wave workgroup per CU; its converting form does not fit 128 registers: the
  //  caller converts u, w in a pass of its own for tracer batches at that size)
  X(8, true) X(16, true) X(32, true) X(64, false)
#undef X
  return false;
}
#endif   // part 3
#if MPD_PART(1)
// fp32 plans: two adjacent instances per lane (8-byte elements = pairs of fp32 values, packed
// arithmetic); `a` describes the arrays in PAIRS (ncrms / 2 of them)
bool launch_wm_f32(int lps, int wpb, const MpdataWmArgsT<double>& a8, void* stream, int flags) {
  if (wpb != MPDWM_WPB) return false;
  MpdataWmArgsT<v2::f32x2> a;
  a.f = reinterpret_cast<v2::f32x2*>(a8.f);
  a.u = reinterpret_cast<const v2::f32x2*>(a8.u);
  a.w = reinterpret_cast<const v2::f32x2*>(a8.w);
  a.kc = reinterpret_cast<const v2::f32x2*>(a8.kc);
  a.flux = reinterpret_cast<v2::f32x2*>(a8.flux);
  a.ntiles = a8.ntiles; a.nx = a8.nx; a.nz = a8.nz; a.ntracers = a8.ntracers;
  a.tile_elems = a8.tile_elems; a.f_tstride = a8.f_tstride; a.flux_tstride = a8.flux_tstride; a.reverse = a8.reverse;
  a.u_ref = nullptr; a.w_ref = nullptr; a.ncrms = 0; a.dbg = a8.dbg;
  a.wpark = reinterpret_cast<v2::f32x2*>(a8.wpark);
  a.park_regs = a8.park_regs;
  a.nkw = a8.nkw;
  a.lwt = a8.lwt;
  if (lps == 128) {
    launch_wm_ks_f32(a, stream);
    return true;
  }
#define X(LPS_)                                \
  if (lps == LPS_) {                           \
    launch_wm_t<v2::f32x2, LPS_, MPDWM_WPB>(a, stream, flags); \
    return true;                               \
  }
  MPDATA_WM_LPS(X)
#undef X
  return false;
}

#endif   // part 1
#if MPD_PART(0)
int max_tile_id() { return 43; }

// Experiment / timing-ablation macros this translation unit was compiled with (some of them
// produce wrong results by design).  Empty for a production build; part of mpdata_version().
const char* build_flags() {
  return ""
#ifdef MPDWM_ABL_NODMA
         " MPDWM_ABL_NODMA"
#endif
#ifdef MPDWM_ABL_NOCOMPUTE
         " MPDWM_ABL_NOCOMPUTE"
#endif
#ifdef MPDWM_ABL_FIRSTPASS
         " MPDWM_ABL_FIRSTPASS"
#endif
#ifdef MPD2_ABL_NODMA
         " MPD2_ABL_NODMA"
#endif
#ifdef MPD2_ABL_NOCOMPUTE
         " MPD2_ABL_NOCOMPUTE"
#endif
#ifdef MPDWM_STAMPS
         " MPDWM_STAMPS"
#endif
#ifdef MPDWM_NO_EXECMASK
         " MPDWM_NO_EXECMASK"
#endif
      ;
}

bool tile_info(int id, MpdataTileInfo* info) {
#define X(ID, W_, SPW_, NWV_)                                                     \
  if (id == ID) {                                                                 \
    using T = Tile<W_, SPW_, NWV_>;                                               \
    *info = MpdataTileInfo{ID, W_, SPW_, NWV_, T::SLW, T::NCOL, 1 << 30, 8, T::THREADS, \
                           "kmarch_W" #W_ "_SPW" #SPW_ "_NWV" #NWV_};             \
    return true;                                                                  \
  }
  MPDATA_TILES(X)
#undef X
#define X(ID, LPS_, G_)                                                                  \
  if (id == ID) {                                                                        \
    using T = v2::TileV2<double, LPS_, G_>;                                              \
    *info = MpdataTileInfo{ID, 0, 0, T::NWV, G_, 1 << 30, LPS_, 8, T::THREADS,           \
                           "xmarch_LPS" #LPS_ "_G" #G_};                                 \
    return true;                                                                         \
  }
  MPDATA_TILES_V2(X)
#undef X
#define X(ID, LPS_, G_)                                                                  \
  if (id == ID) {                                                                        \
    using T = v2::TileV2<float, LPS_, G_>;                                               \
    *info = MpdataTileInfo{ID, 0, 0, T::NWV, G_, 1 << 30, LPS_, 4, T::THREADS,           \
                           "xmarch_f32_LPS" #LPS_ "_G" #G_};                             \
    return true;                                                                         \
  }
  MPDATA_TILES_V2_F32(X)
#undef X
#define X(ID, LPS_, G_)                                                                  \
  if (id == ID) {                                                                        \
    using T = v2::TileV2<v2::f32x2, LPS_, G_>;                                           \
    *info = MpdataTileInfo{ID, 0, 0, T::NWV, 2 * G_, 1 << 30, LPS_, 4, T::THREADS,       \
                           "xmarch_f32x2_LPS" #LPS_ "_G" #G_};                           \
    return true;                                                                         \
  }
  MPDATA_TILES_V2_F32X2(X)
#undef X
  return false;
}

bool launch(int id, const MpdataArgs& a, int ntracers, void* stream) {
#define X(ID, W_, SPW_, NWV_)                          \
  if (id == ID) {                                      \
    launch_tile<W_, SPW_, NWV_>(a, ntracers, stream);  \
    return true;                                       \
  }
  MPDATA_TILES(X)
#undef X
#define X(ID, LPS_, G_)                                    \
  if (id == ID) {                                          \
    launch_tile_v2<double, LPS_, G_>(a, ntracers, stream); \
    return true;                                           \
  }
  MPDATA_TILES_V2(X)
#undef X
  return false;
}

bool launch_f32(int id, const MpdataArgsF32& a, int ntracers, void* stream) {
#define X(ID, LPS_, G_)                                   \
  if (id == ID) {                                         \
    launch_tile_v2<float, LPS_, G_>(a, ntracers, stream); \
    return true;                                          \
  }
  MPDATA_TILES_V2_F32(X)
#undef X
  // two instances per lane: the same arrays seen as ncrms/2 pairs of adjacent instances
  if (id >= 40 && (a.ncrms & 1) == 0) {
    MpdataArgsT<v2::f32x2> p;
    p.f = reinterpret_cast<v2::f32x2*>(a.f);
    p.u = reinterpret_cast<const v2::f32x2*>(a.u);
    p.w = reinterpret_cast<const v2::f32x2*>(a.w);
    p.rho = reinterpret_cast<const v2::f32x2*>(a.rho);
    p.rhow = reinterpret_cast<const v2::f32x2*>(a.rhow);
    p.adz = reinterpret_cast<const v2::f32x2*>(a.adz);
    p.flux = reinterpret_cast<v2::f32x2*>(a.flux);
    p.ncrms = a.ncrms / 2; p.nx = a.nx; p.nz = a.nz; p.ntracers = ntracers;
    p.f_tstride = a.f_tstride / 2; p.flux_tstride = a.flux_tstride / 2;
    p.dbg = a.dbg;
    p.wpark = reinterpret_cast<v2::f32x2*>(a.wpark);
    p.park_regs = a.park_regs;
#define X(ID, LPS_, G_)                                        \
  if (id == ID) {                                              \
    launch_tile_v2<v2::f32x2, LPS_, G_>(p, ntracers, stream);  \
    return true;                                               \
  }
    MPDATA_TILES_V2_F32X2(X)
#undef X
  }
  return false;
}
#endif   // part 0

}  // namespace MPDATA_NS
