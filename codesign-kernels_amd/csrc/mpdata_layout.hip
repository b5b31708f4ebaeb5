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
#include <cstdint>
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
  // DEPTH columns in flight: the fetch of column c + DEPTH goes out before column c + 1 is waited for.
  constexpr int DEPTH = 2;
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


// ---- Import (reference -> plan) of f, u, w by ROW SEGMENTS through LDS-DMA (round 4).  The data path of the
// kernel that reads u, w from the reference layout (mpdata_kernel_wm_body.h, UWREF) without the arithmetic: a
// workgroup owns 16 adjacent instances, so a (column, level) row of the array is a 128-byte segment that belongs to
// it alone; the rows of a column PAIR arrive by 16-byte-per-lane LDS-DMA (one instruction = 8 rows; the workgroup's
// LPS/4 waves fetch the 2 x LPS rows of a pair with one instruction each) into a ring of three pairs, one barrier
// per pair; every wave then takes the two columns of its own tile (64/LPS instances, lanes along the levels) out of
// LDS -- the transposed read is 2-way bank-conflicted at worst thanks to the XOR swizzle on the SOURCE address --
// and stores them as the plan layout wants them: 8 bytes per lane, a contiguous chunk per wave and column, the
// line-aligned main part with the streaming policy.  Nothing passes through vector registers on the way in, so
// three column pairs are in flight per workgroup at 48 VGPRs.
// Conditions (else the column-walking kernel above): even ncrms, 16-byte aligned base, array below 4 GiB.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
struct MpdataRowJobs {
  MpdataLayoutJob j[2];
  int ntr_max;
};
template <int LPS>
__global__ void __launch_bounds__(16 * LPS) wm_import_rows_kernel(const MpdataRowJobs js) {
  constexpr int SLP = 64 / LPS;          // instances per wave = per tile
  constexpr int WPB = LPS / 4;           // waves per workgroup: 16 instances
  constexpr int GX = 16;
  constexpr int XRG = LPS / 8;           // groups of 8 rows (one DMA instruction) per column
  constexpr int XARR = LPS * GX;         // elements of one column block: LPS rows x 16
  constexpr int NS = 3;
  static_assert(WPB == 2 * XRG, "one DMA instruction per wave and column pair");
  __shared__ double ring[NS * 2 * XARR];
  const MpdataLayoutJob& j = js.j[blockIdx.z / js.ntr_max];
  const int tr = blockIdx.z % js.ntr_max;
  if (tr >= j.ntr) return;   // (uniform for the workgroup: before any barrier)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nlev = j.nlev, ncols = j.ncols;
  const double* ref = static_cast<const double*>(j.ref) + (long long)tr * j.ref_tstride;
  double* prv = static_cast<double*>(j.prv) + (long long)tr * j.prv_tstride;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned OOB = 0xFFFFFFF8u;

  // ---- source side: this wave's rows.  Row (column cs, level kk) of the workgroup = 16 instances at
  //      (cs + ncols*kk) * ncrms*8 + sl_base*8; the column is the scalar offset, everything else per lane.
  const long long lvl = j.ncrms * (long long)j.ref_levmul;        // elements between levels
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(ref), (short)0,
                                                                        (int)(unsigned)(lvl * nlev * 8), 0x00020000);
  const unsigned colB = (unsigned)(j.ncrms * 8);
  const int xcol = wave / XRG, xrg = wave % XRG;                   // column of the pair and row group this wave fetches
  const int row = xrg * 8 + (lane >> 3), pch = lane & 7;
  const long long sl_src = (long long)blockIdx.x * GX + 2 * (pch ^ ((row >> 1) & 7));
  const unsigned xv = (row < nlev && sl_src + 1 < j.ncrms) ? (unsigned)((sl_src + lvl * row) * 8) : OOB;   // else: zeros
  auto dma = [&](const int P) __attribute__((always_inline)) {
    const int cs = 2 * P + xcol;
    double* d = ring + (P % NS) * (2 * XARR) + xcol * XARR + xrg * 8 * GX;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)d, 16, (int)(cs < ncols ? xv : OOB), (int)((unsigned)min(cs, ncols - 1) * colB), 0, 2);
  };

  // ---- destination side: lane -> (instance s of the wave's tile, level kk); element e = s*nlev + kk of the chunk
  const int s_l = lane / LPS, kk = lane % LPS;
  const long long tile = (long long)blockIdx.x * WPB + wave;
  const bool own = kk < nlev && tile < j.ntiles;
  const int e = s_l * nlev + (kk < nlev ? kk : 0);
  const long long main_e = j.main_e, rem_e = j.chunk - j.main_e;
  const bool in_main = e < main_e;
  const unsigned eoff = own ? (unsigned)((in_main ? e : (long long)j.ncol_p * main_e + (e - main_e)) * 8) : OOB;
  const unsigned mainB = (unsigned)(main_e * 8), remB = (unsigned)(rem_e * 8);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
      prv + (tile < j.ntiles ? tile : 0) * j.prv_tile_stride, (short)0, (int)(unsigned)(tile < j.ntiles ? j.prv_tile_stride * 8 : 0), 0x00020000);
  // LDS read position: row kk, instance s_g of the workgroup, un-swizzled
  const int s_g = wave * SLP + s_l;
  const int kr = kk < nlev ? kk : 0;
  const double* rp = ring + kr * GX + ((((s_g >> 1) ^ ((kr >> 1) & 7)) << 1) | (s_g & 1));

  const int npairs = (ncols + 1) / 2;
  dma(0);
  if (npairs > 1) dma(1);
  for (int P = 0; P < npairs; ++P) {
    // this wave's share of pair P has landed (at most the fetch of pair P+1 is newer; loads return in order -- the
    // column stores of the previous pair may still be out: the count then waits for them too, which is the safe
    // side); after the barrier the whole pair is there and everybody is done reading pair P-1, whose slot the
    // fetch of pair P+2 takes
    if (P + 1 < npairs) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (P + 2 < npairs) dma(P + 2);
    const double* q = rp + (P % NS) * (2 * XARR);
    const double v0 = q[0], v1 = q[XARR];
    const int c0 = 2 * P + j.prv_col0, c1 = c0 + 1;
    const unsigned o1 = (2 * P + 1 < ncols) ? eoff : OOB;
    if (in_main) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v0), rd, (int)eoff, (int)((unsigned)c0 * mainB), 2);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v1), rd, (int)o1, (int)((unsigned)c1 * mainB), 2);
    } else {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v0), rd, (int)eoff, (int)((unsigned)c0 * remB), 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, v1), rd, (int)o1, (int)((unsigned)c1 * remB), 0);
    }
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
  // (nz > 64, one instance per tile: 16 instances per workgroup, 8 above 128 levels)
  if (nlev > 256) return hipErrorInvalidValue;
  const int ti = nlev > 128 ? 8 : nlev > 64 ? 16 : (!to_private ? 32 : (nlev * 64 <= 2048 ? 64 : 32));
  const dim3 grid((unsigned)((nc + ti - 1) / ti), 1, (unsigned)(nj * js.ntr_max)), block(256);
  const size_t lds = (size_t)2 * nlev * (ti + 1) * 8;
  if (ti == 64) {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 64, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 64, false>), grid, block, lds, stream, js);
  } else if (ti == 16) {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 16, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 16, false>), grid, block, lds, stream, js);
  } else if (ti == 8) {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 8, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 8, false>), grid, block, lds, stream, js);
  } else {
    if (to_private) hipLaunchKernelGGL((wm_convert_cols_kernel<double, 32, true>), grid, block, lds, stream, js);
    else hipLaunchKernelGGL((wm_convert_cols_kernel<double, 32, false>), grid, block, lds, stream, js);
  }
  return hipGetLastError();
}

// f, u, w by row segments (wm_import_rows_kernel): nj = 1 or 2 jobs of equal nlev / slp / ncrms.  Returns
// hipErrorNotSupported when the conditions of that kernel are not met (the caller then takes the column-walking one).
hipError_t mpdata_layout_import_rows(const MpdataLayoutJob* jobs, int nj, hipStream_t stream) {
  if (nj < 1 || nj > 2) return hipErrorInvalidValue;
  static const bool off = getenv("MPDATA_LAYOUT_NOROWS") != nullptr;   // (A/B: the column-walking kernel)
  if (off) return hipErrorNotSupported;
  MpdataRowJobs js;
  js.ntr_max = 1;
  for (int i = 0; i < nj; ++i) {
    const MpdataLayoutJob& j = jobs[i];
    if (j.ncrms < 1 || j.ncols < 1 || j.ntr < 1 || j.nlev < 1 || j.main_e <= 0 || j.nlev != jobs[0].nlev || j.slp != jobs[0].slp ||
        j.ncrms != jobs[0].ncrms)
      return hipErrorInvalidValue;
    if ((j.ncrms & 1) || ((uintptr_t)j.ref & 15) || (j.ref_tstride & 1) || j.ref_colmul != 1 ||
        (double)j.ncrms * (double)j.ref_levmul * j.nlev * 8.0 >= 4294967000.0 || (double)j.prv_tile_stride * 8.0 >= 4294967000.0)
      return hipErrorNotSupported;
    js.j[i] = j;
    js.ntr_max = j.ntr > js.ntr_max ? j.ntr : js.ntr_max;
  }
  if (nj == 1) js.j[1] = js.j[0];
  const int lps = 64 / jobs[0].slp;
  if (jobs[0].nlev >= lps) return hipErrorNotSupported;
  const dim3 grid((unsigned)((jobs[0].ncrms + 15) / 16), 1, (unsigned)(nj * js.ntr_max));
  switch (lps) {
    case 8: hipLaunchKernelGGL((wm_import_rows_kernel<8>), grid, dim3(128), 0, stream, js); break;
    case 16: hipLaunchKernelGGL((wm_import_rows_kernel<16>), grid, dim3(256), 0, stream, js); break;
    case 32: hipLaunchKernelGGL((wm_import_rows_kernel<32>), grid, dim3(512), 0, stream, js); break;
    case 64: hipLaunchKernelGGL((wm_import_rows_kernel<64>), grid, dim3(1024), 0, stream, js); break;
    default: return hipErrorNotSupported;
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
