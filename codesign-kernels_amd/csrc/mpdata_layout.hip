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

}  // namespace

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
