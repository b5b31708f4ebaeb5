// mpdata_layout.hip -- conversion between the reference array layout (the C-ABI contract:
// Fortran order, CRM instance `sl` fastest, reference
// mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:33-38) and the plan-private
// "wave-major" layout of mpdata_kernel_wm_body.h:
//     [tracer][tile][column][instance-in-tile][level]      (level fastest)
// with, for f, u and w, every column chunk split into its whole 128-byte lines (stored first,
// column after column) and the rest (stored behind them).
// A workgroup moves all levels of 64 consecutive instances of one column through an LDS tile:
// the reference side is read/written in 512-byte row segments, the private side in the
// contiguous chunks of 64/slp tiles.  These kernels run in upload / download / device
// import / export of a plan -- outside the timed region, like the reference's
// `!$acc update device / host` (:107, :241).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "mpdata_layout.h"

namespace {

constexpr int TI = 64;        // instances per workgroup
constexpr int TPAD = TI + 1;  // LDS row stride

// to_private = true :  ref -> private;  false: private -> ref
template <typename R, bool TO_PRIVATE>
__global__ void __launch_bounds__(256) wm_convert_kernel(const MpdataLayoutJob j) {
  extern __shared__ double lds_raw[];
  R* tile = reinterpret_cast<R*>(lds_raw);   // [nlev][TPAD]
  const int tid = threadIdx.x;
  const long long sl0 = (long long)blockIdx.x * TI;
  const int cs = blockIdx.y;                 // column of the reference-side array
  const int tr = blockIdx.z;
  const int nlev = j.nlev, slp = j.slp;
  const int n = nlev * TI;
  R* ref = static_cast<R*>(j.ref) + (long long)tr * j.ref_tstride;
  R* prv = static_cast<R*>(j.prv) + (long long)tr * j.prv_tstride;
  const long long ninst_p = (long long)j.ntiles * slp;   // instances the private side holds (padded)

  auto ref_at = [&](long long sl, int kk) -> long long {
    return sl + j.ncrms * ((long long)cs * j.ref_colmul + (long long)kk * j.ref_levmul);
  };
  auto prv_at = [&](long long inst, int kk) -> long long {
    const long long t = inst / slp;
    const int s = (int)(inst - t * slp);
    const long long e = (long long)s * nlev + kk, c = cs + j.prv_col0;
    if (j.main_e == 0) return t * j.prv_tile_stride + c * j.chunk + e;
    const long long rem_e = j.chunk - j.main_e;
    return t * j.prv_tile_stride + (e < j.main_e ? c * j.main_e + e : j.ncol_p * j.main_e + c * rem_e + (e - j.main_e));
  };
  if (TO_PRIVATE) {
    for (int i = tid; i < n; i += 256) {
      const int kk = i / TI, t = i - kk * TI;
      long long sl = sl0 + t;
      if (sl >= j.ncrms) sl = j.ncrms - 1;   // the padding instances of the last tile: copies
      tile[kk * TPAD + t] = ref[ref_at(sl, kk)];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int t = i / nlev, kk = i - t * nlev;
      const long long inst = sl0 + t;
      if (inst < ninst_p) prv[prv_at(inst, kk)] = tile[kk * TPAD + t];
    }
  } else {
    for (int i = tid; i < n; i += 256) {
      const int t = i / nlev, kk = i - t * nlev;
      const long long inst = sl0 + t;
      if (inst < ninst_p) tile[kk * TPAD + t] = prv[prv_at(inst, kk)];
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      const int kk = i / TI, t = i - kk * TI;
      const long long sl = sl0 + t;
      if (sl < j.ncrms) ref[ref_at(sl, kk)] = tile[kk * TPAD + t];
    }
  }
}

// The same conversion for arrays with many columns (f, u, w): a workgroup owns TI consecutive
// instances of one array (tracer) and walks through ALL its columns.  The private side of those
// instances is then one contiguous region (TI/slp tiles) that this workgroup alone writes (reads),
// column after column: the 48-byte remainders of neighbouring columns, which share 128-byte lines,
// meet in L2 before the line is written back, and the DRAM pages stay open.  Column c+1 is fetched
// into registers while column c goes out of the LDS tile.  Up to two arrays per launch (u and w of
// an import: grid.z = array * ntr + tracer).
struct MpdataLayoutJobs {
  MpdataLayoutJob j[2];
  int ntr_max;
};
template <typename R, int TI2, bool TO_PRIVATE>
__global__ void __launch_bounds__(256) wm_convert_cols_kernel(const MpdataLayoutJobs js) {
  extern __shared__ double lds_raw[];
  constexpr int TP = TI2 + 1;
  constexpr int NPT = 8;   // elements per thread and column: nlev * TI2 <= 8 * 256
  const MpdataLayoutJob& j = js.j[blockIdx.z / js.ntr_max];
  const int tr = blockIdx.z % js.ntr_max;
  if (tr >= j.ntr) return;
  R* tile = reinterpret_cast<R*>(lds_raw);   // [2][nlev][TP]
  const int tid = threadIdx.x;
  const long long sl0 = (long long)blockIdx.x * TI2;
  const int nlev = j.nlev, slp = j.slp;
  const int n = nlev * TI2;
  const int tsz = nlev * TP;
  R* ref = static_cast<R*>(j.ref) + (long long)tr * j.ref_tstride;
  R* prv = static_cast<R*>(j.prv) + (long long)tr * j.prv_tstride;
  const long long ninst_p = (long long)j.ntiles * slp;
  const long long rem_e = j.chunk - j.main_e;

  // per-thread element lists of the two sides (the same for every column)
  long long ro[NPT], po[NPT];   // reference-side offset (column 0), private-side offset (column 0, + column stride below)
  int rl[NPT], pl[NPT];         // LDS positions; -1: nothing
  bool pmain[NPT];
#pragma unroll
  for (int e = 0; e < NPT; ++e) {
    const int i = tid + e * 256;
    rl[e] = pl[e] = -1; ro[e] = po[e] = 0; pmain[e] = true;
    if (i < n) {
      {  // reference side: i -> (level, instance): 8 * TI2 contiguous bytes per level
        const int kk = i / TI2, t = i - kk * TI2;
        long long sl = sl0 + t;
        const bool in = sl < j.ncrms;
        if (!in) sl = j.ncrms - 1;     // (import: the padding instances of the last tile are copies)
        if (TO_PRIVATE || in) { rl[e] = kk * TP + t; ro[e] = sl + j.ncrms * (long long)kk * j.ref_levmul; }
      }
      {  // private side: i -> (instance, level): the chunk of a tile is contiguous
        const int t = i / nlev, kk = i - t * nlev;
        const long long inst = sl0 + t;
        if (inst < ninst_p) {
          const long long tl = inst / slp;
          const long long el = (long long)(inst - tl * slp) * nlev + kk;
          pl[e] = kk * TP + t;
          pmain[e] = el < j.main_e;
          po[e] = tl * j.prv_tile_stride + (pmain[e] ? el : (long long)j.ncol_p * j.main_e + (el - j.main_e));
        }
      }
    }
  }
  // running pointers: every column advances them by a constant (no 64-bit multiply per element and column)
  const long long rcol = j.ncrms * j.ref_colmul;   // reference-side elements between columns
  R* rp[NPT];
  R* pp[NPT];
  long long pstep[NPT];
#pragma unroll
  for (int e = 0; e < NPT; ++e) {
    rp[e] = ref + ro[e];
    pstep[e] = pmain[e] ? j.main_e : rem_e;
    pp[e] = prv + po[e] + (long long)j.prv_col0 * pstep[e];
  }
  // MPD_CONV_DEPTH columns in flight: the fetch of column c + DEPTH goes out before column c + 1 is waited for.
#ifndef MPD_CONV_DEPTH
#define MPD_CONV_DEPTH 2
#endif
  constexpr int DEPTH = MPD_CONV_DEPTH;
  R vs[DEPTH][NPT];
  auto fetch = [&](R (&v)[NPT]) {   // the column the source pointers stand on; then on to the next
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      if (TO_PRIVATE) { if (rl[e] >= 0) v[e] = *rp[e]; rp[e] += rcol; }
      else { if (pl[e] >= 0) v[e] = *pp[e]; pp[e] += pstep[e]; }
    }
  };
  // destination pointers of their own (the source pointers run two columns ahead)
  R* dp[NPT];
#pragma unroll
  for (int e = 0; e < NPT; ++e) dp[e] = TO_PRIVATE ? pp[e] : rp[e];
  auto column = [&](R (&v)[NPT], const int cs) {
    R* tb = tile + (cs & 1) * tsz;
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      const int l = TO_PRIVATE ? rl[e] : pl[e];
      if (l >= 0) tb[l] = v[e];
    }
    __syncthreads();   // (two buffers: the previous column's readers are past the barrier of this one's predecessor)
    if (cs + DEPTH < j.ncols) fetch(v);
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      // (streaming hint on the reference-side row stores of an export: +9 %; on the loads, or on the 8-byte stores
      //  of an import, it costs 15-35 %)
      if (TO_PRIVATE) { if (pl[e] >= 0) *dp[e] = tb[pl[e]]; dp[e] += pstep[e]; }
      else { if (rl[e] >= 0) __builtin_nontemporal_store(tb[rl[e]], dp[e]); dp[e] += rcol; }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (d < j.ncols) fetch(vs[d]);
  for (int cs = 0; cs < j.ncols; cs += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (cs + d < j.ncols) column(vs[d], cs + d);
  }
}

}  // namespace

// f, u, w (split arrays, many columns) in one launch; nj = 1 or 2 jobs of equal nlev / slp
hipError_t mpdata_layout_convert_cols(const MpdataLayoutJob* jobs, int nj, bool to_private, hipStream_t stream) {
  if (nj < 1 || nj > 2) return hipErrorInvalidValue;
  MpdataLayoutJobs js;
  js.ntr_max = 1;
  long long nc = 0;
  for (int i = 0; i < nj; ++i) {
    const MpdataLayoutJob& j = jobs[i];
    if (j.ncrms < 1 || j.ncols < 1 || j.ntr < 1 || j.nlev < 1 || j.main_e <= 0 || j.nlev != jobs[0].nlev) return hipErrorInvalidValue;
    js.j[i] = j;
    js.ntr_max = j.ntr > js.ntr_max ? j.ntr : js.ntr_max;
    nc = j.ncrms > nc ? j.ncrms : nc;
  }
  if (nj == 1) js.j[1] = js.j[0];
  const int nlev = jobs[0].nlev;
  // instances per workgroup: 8 elements per thread and column
  // (an export is 7 % faster with twice as many, half as wide workgroups; an import is not)
  static const int ti_env = [] { const char* v = getenv("MPDATA_CONV_TI"); return v ? atoi(v) : 0; }();   // (experiments: 32 / 64)
  const int ti = ti_env == 32 || (ti_env == 0 && !to_private) ? 32 : (nlev * 64 <= 2048 ? 64 : 32);
  const dim3 grid((unsigned)((nc + ti - 1) / ti), 1, (unsigned)(nj * js.ntr_max)), block(256);
  const size_t lds = (size_t)2 * nlev * (ti + 1) * 8;
  if (ti == 64) {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 64, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 64, false>), grid, block, lds, stream, js);
  } else {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 32, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 32, false>), grid, block, lds, stream, js);
  }
  return hipGetLastError();
}

hipError_t mpdata_layout_convert(const MpdataLayoutJob& j, int elem_bytes, bool to_private, hipStream_t stream) {
  if (j.ncrms < 1 || j.ncols < 1 || j.ntr < 1 || j.nlev < 1) return hipErrorInvalidValue;
  const dim3 grid((unsigned)((j.ncrms + TI - 1) / TI), (unsigned)j.ncols, (unsigned)j.ntr), block(256);
  const size_t lds = (size_t)j.nlev * TPAD * elem_bytes;
  if (elem_bytes == 8) {
    if (to_private) hipLaunchKernelGGL((wm_convert_kernel<double, true>), grid, block, lds, stream, j);
    else hipLaunchKernelGGL((wm_convert_kernel<double, false>), grid, block, lds, stream, j);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
