// nlk_kernel_body.h -- HIP kernel of the MPAS-Ocean high-order flux loop nest for gfx950
// (reference nested_loops/nested.F90:123-157 / :495-559; included once per arithmetic variant
// with NLK_NS defined).
//
// A WAVE owns one edge, its lanes the vertical levels (level index is the contiguous one, so
// every row access is a coalesced 512-byte segment; 100 levels = two trips).  The loop over
// the contributing cells stays sequential per level -- the reference's summation order --
// and everything that is per (edge, i) is wave-uniform: cell index, coefficients and the
// cell's level range come through scalar loads, the level-range test is a lane mask.  The
// gathered tracer columns (2.2 MB at the reference's size) live in L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nlk_args.h"

namespace NLK_NS {

constexpr int NLK_WAVES = 4;  // edges per workgroup

__global__ void __launch_bounds__(64 * NLK_WAVES) nlk_kernel(const NlkArgs g) {
  const int lane = threadIdx.x & 63;
  const int iEdge = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NLK_WAVES + (threadIdx.x >> 6)));
  if (iEdge >= g.nEdges) return;
  const long long erow = (long long)g.nvldim * iEdge;
  int nadv = g.nAdvCellsForEdge[iEdge];
  nadv = nadv < g.nAdv ? nadv : g.nAdv;
  const int* cells = g.advCellsForEdge + (long long)g.nAdv * iEdge;
  const double* c1p = g.advCoefs + (long long)g.nAdv * iEdge;
  const double* c3p = g.advCoefs3rd + (long long)g.nAdv * iEdge;
  // two 64-level trips are in flight together (100 levels = lanes 0..63 of trip 0 + lanes
  // 0..35 of trip 1): twice the independent gathers per wave
  for (int k0 = 0; k0 < g.nVertLevels; k0 += 128) {
    int k[2];
    bool lvl[2];
    double wgt[2], sgn[2], acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      k[t] = k0 + 64 * t + lane + 1;  // 1-based level
      lvl[t] = k[t] <= g.nVertLevels;
      const double ntf = lvl[t] ? g.normalThicknessFlux[erow + k[t] - 1] : 0.0;
      wgt[t] = ntf * (lvl[t] ? g.advMaskHighOrder[erow + k[t] - 1] : 0.0);   // :126-127
      sgn[t] = __builtin_copysign(1.0, ntf);                                  // :128-129
      acc[t] = 0.0;
    }
    // the gathers of CH cells are issued together (they do not depend on the running sum);
    // the additions then follow in the reference's order i ascending
    constexpr int CH = 5;
    for (int i0 = 0; i0 < nadv; i0 += CH) {                                   // :136-148
      double tv[CH][2], c1[CH], c3[CH];
      bool on[CH][2];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = i0 + j;
        const bool have = i < nadv;
        const int iCell = have ? cells[i] : 0;
        const bool cell_ok = iCell >= 1 && iCell <= g.nCells;   // (the reference would read out of bounds)
        const int ic = cell_ok ? iCell - 1 : 0;
        const int kmin = g.minLevelCell[ic], kmax = g.maxLevelCell[ic];
        c1[j] = have ? c1p[i] : 0.0;
        c3[j] = (have ? c3p[i] : 0.0) * g.coef3rdOrder;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          on[j][t] = cell_ok && lvl[t] && k[t] >= kmin && k[t] <= kmax;
          tv[j][t] = on[j][t] ? g.tracerCur[(long long)g.nvldim * ic + k[t] - 1] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          if (on[j][t]) acc[t] = acc[t] + tv[j][t] * wgt[t] * (c1[j] + c3[j] * sgn[t]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (lvl[t]) g.highOrderFlx[erow + k[t] - 1] = acc[t];
  }
}

void launch(const NlkArgs& g, void* stream) {
  const unsigned gx = (unsigned)((g.nEdges + NLK_WAVES - 1) / NLK_WAVES);
  hipLaunchKernelGGL(nlk_kernel, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g);
}

}  // namespace NLK_NS
