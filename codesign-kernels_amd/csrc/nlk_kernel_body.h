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
#include <stdlib.h>

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

// ---- large meshes: PERSISTENT waves with a software pipeline over their edges -------------------
// One edge per wave as above is a chain of four dependent memory round trips (cell count -> cell list
// -> the cells' level ranges -> the gathered tracer columns) for 2.6 KB of compulsory traffic: at full
// occupancy a wave still spends ~12 us per edge (1.23 ms for 819200 edges, the same with gathers that
// hit in L2 and with gathers that miss: latency, not bandwidth).  Here a wave owns a contiguous RANGE
// of edges and the chain runs ahead of the arithmetic:
//   * the per-(edge, cell) metadata is LANE-distributed -- lane i < nAdv holds cell i, its two
//     coefficients and its level range (5 small vector loads per edge, 7 VGPRs per edge in flight)
//     -- and turned into wave-uniform scalars by v_readlane where the gather needs them;
//   * while edge e is gathered and summed, the cell list of edge e+2 and the level ranges of edge e+1
//     are already in flight.
// The sum over the cells keeps the reference's order (i ascending, :136-148): same results.
template <int UNUSED = 0>
__global__ void __launch_bounds__(64 * NLK_WAVES) nlk_kernel_pipe(const NlkArgs g, const int epw) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NLK_WAVES + (threadIdx.x >> 6)));
  const long long e_lo = (long long)wid * epw;
  if (e_lo >= g.nEdges) return;
  const int e_begin = (int)e_lo;
  const int e_end = (int)((e_lo + epw < g.nEdges) ? e_lo + epw : g.nEdges);
  const bool mine = lane < g.nAdv;   // lanes that hold a cell of the list (nAdv <= 64)

  struct Cells { int cell; double c1, c3; };
  struct Levels { int kmin, kmax; };
  auto load_cells = [&](const int e) __attribute__((always_inline)) {
    Cells c{0, 0.0, 0.0};
    if (mine && e < e_end) {
      const long long o = (long long)g.nAdv * e + lane;
      c.cell = g.advCellsForEdge[o];
      c.c1 = g.advCoefs[o];
      c.c3 = g.advCoefs3rd[o];
    }
    return c;
  };
  auto load_levels = [&](const Cells& c) __attribute__((always_inline)) {
    Levels l{1, 0};   // an empty range: contributes nothing
    const bool ok = c.cell >= 1 && c.cell <= g.nCells;   // (the reference would read out of bounds)
    if (mine && ok) {
      l.kmin = g.minLevelCell[c.cell - 1];
      l.kmax = g.maxLevelCell[c.cell - 1];
    }
    return l;
  };
  auto rl_i = [](const int v, const int j) __attribute__((always_inline)) { return __builtin_amdgcn_readlane(v, j); };
  auto rl_d = [&](const double v, const int j) __attribute__((always_inline)) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
  };

  Cells C0 = load_cells(e_begin), C1 = load_cells(e_begin + 1);
  Levels L0 = load_levels(C0);
  for (int e = e_begin; e < e_end; ++e) {
    const Cells C2 = load_cells(e + 2);      // two edges ahead
    const Levels L1 = load_levels(C1);       // one edge ahead (its cell list arrived during the previous edge)
    const long long erow = (long long)g.nvldim * e;
    int nadv = g.nAdvCellsForEdge[e];
    nadv = nadv < g.nAdv ? nadv : g.nAdv;
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
      constexpr int CH = 5;
      for (int i0 = 0; i0 < nadv; i0 += CH) {                                   // :136-148
        double tv[CH][2], c1[CH], c3[CH];
        bool on[CH][2];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int i = i0 + j;
          const bool have = i < nadv;
          const int ii = have ? i : 0;
          const int iCell = rl_i(C0.cell, ii);
          const bool cell_ok = have && iCell >= 1 && iCell <= g.nCells;
          const int ic = cell_ok ? iCell - 1 : 0;
          const int kmin = rl_i(L0.kmin, ii), kmax = rl_i(L0.kmax, ii);
          c1[j] = have ? rl_d(C0.c1, ii) : 0.0;
          c3[j] = (have ? rl_d(C0.c3, ii) : 0.0) * g.coef3rdOrder;
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
    C0 = C1; C1 = C2; L0 = L1;
  }
}

// force: -1 automatic, 0 one edge per wave, 1 the pipelined persistent kernel (nlk_set_kernel / NLK_PIPE)
void launch(const NlkArgs& g, void* stream, int force) {
  // wave slots of the chip (CUs x 32 waves); meshes with several edges per slot take the pipelined
  // persistent kernel
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    slots = cus * 32;
  }
  const int epw = (g.nEdges + slots - 1) / slots;
  const bool pipe = g.nAdv <= 64 && (force == 1 || (force < 0 && epw >= 2));
  if (pipe) {
    const int epw1 = epw < 1 ? 1 : epw;
    const long long waves = ((long long)g.nEdges + epw1 - 1) / epw1;
    const unsigned gx = (unsigned)((waves + NLK_WAVES - 1) / NLK_WAVES);
    hipLaunchKernelGGL(nlk_kernel_pipe<0>, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g, epw1);
  } else {
    const unsigned gx = (unsigned)((g.nEdges + NLK_WAVES - 1) / NLK_WAVES);
    hipLaunchKernelGGL(nlk_kernel, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g);
  }
}

}  // namespace NLK_NS
