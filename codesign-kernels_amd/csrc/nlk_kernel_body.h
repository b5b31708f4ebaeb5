// nlk_kernel_body.h -- HIP kernel of the MPAS-Ocean high-order flux loop nest for gfx950
// (reference nested_loops/nested.F90:123-157 / :495-559; included once per arithmetic variant
// with NLK_NS defined).
//
// A WAVE owns one edge, its lanes the vertical levels (level index is the contiguous one, so
// every row access is a coalesced 512-byte segment; 100 levels = two trips).  The loop over
// the contributing cells stays sequential per level -- the reference's summation order --
// and everything that is per (edge, i) is wave-uniform: cell index, coefficients and the
// cell's level range come through scalar loads.
//
// The nest is bound by instruction ISSUE, not by memory (round 3, tools/nlk_probe.py on the 32 x
// mesh: the three edge streams alone 0.31 ms = 6.3 TB/s; with ten gathers of ONE cell, all L1 hits,
// 1.0 ms; with the real gathers 1.2 ms, local or random connectivity alike).  So the per-(cell,
// level) work is kept to four instructions:
//   * the level-range test `minLevelCell <= k <= maxLevelCell` (:139) is the RANGE CHECK of a buffer
//     descriptor built per cell -- base = the cell's column, num_records = 8 * maxLevelCell (clipped
//     to nVertLevels; 0 for a cell index out of range): a lane above the range reads 0.0 without a
//     compare, a select or an EXEC mask, and the per-lane byte offset 8(k-1) is the same for every
//     cell (no per-gather address arithmetic); minLevelCell > 1 (ice shelves, :60) costs a compare
//     and a select only for such cells (a wave-uniform branch);
//   * a masked level then ADDS an exact zero instead of being skipped: 0.0 * wgt * coef = +-0, and
//     x + (+-0) = x (0 + (+-0) = +0 for the empty sum): bit-identical to the reference's skip;
//   * coef1 + coef3 * sgn, tracer * wgt, the accumulation: three fp64 operations.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "nlk_args.h"

namespace NLK_NS {

constexpr int NLK_WAVES = 4;  // edges per workgroup

// A wave-uniform load that really IS a scalar load.  hipcc selects s_load only when it can prove the
// memory invariant; for plain global pointers in a kernel that also stores (highOrderFlx may alias)
// it falls back to a 64-lane vector load of one address + v_readfirstlane: round 3 counted 62
// vector-memory reads and 7 scalar ones per edge (rocprofv3 SQ_INSTS_VMEM_RD / SQ_INSTS_SMEM) where
// the design has 12 and 50 -- the nest was waiting on its own metadata in the vector-memory pipe.
// Reading through the constant address space states the invariance (the inputs are never written
// by the kernel).
template <typename T>
__device__ __forceinline__ T sload(const T* p) {
  typedef const T __attribute__((address_space(4))) * cptr_t;
  return *(cptr_t)(unsigned long long)p;
}
// ... at base + a 32-bit unsigned BYTE offset (s_load with a scalar offset operand: no 64-bit add per load)
template <typename T>
__device__ __forceinline__ T sload_off(const T* base, const unsigned byte_off) {
  typedef const char __attribute__((address_space(4))) * cbase_t;
  typedef const T __attribute__((address_space(4))) * cptr_t;
  return *(cptr_t)((cbase_t)(unsigned long long)base + byte_off);
}

typedef unsigned int nlk_u32x2 __attribute__((ext_vector_type(2)));

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
    unsigned koff[2];   // byte offset of the lane's level inside a cell's column
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      k[t] = k0 + 64 * t + lane + 1;  // 1-based level
      lvl[t] = k[t] <= g.nVertLevels;
      koff[t] = (unsigned)(k[t] - 1) * 8u;
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
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int i = i0 + j;
        const bool have = i < nadv;
        const int iCell = have ? cells[i] : 0;
        const bool cell_ok = iCell >= 1 && iCell <= g.nCells;   // (the reference would read out of bounds)
        const int ic = cell_ok ? iCell - 1 : 0;
        const int kmin = g.minLevelCell[ic];
        int kmax = g.maxLevelCell[ic];
        kmax = kmax < g.nVertLevels ? kmax : g.nVertLevels;
        c1[j] = have ? c1p[i] : 0.0;
        c3[j] = (have ? c3p[i] : 0.0) * g.coef3rdOrder;
        // levels 1 .. kmax of the cell's column: the descriptor's range check masks the rest (zeros)
        const int nrec = (cell_ok && kmax > 0) ? kmax * 8 : 0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double*>(g.tracerCur + (long long)g.nvldim * ic), (short)0, nrec, 0x00020000);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          tv[j][t] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)koff[t], 0, 0));
        if (kmin > 1) {   // (wave-uniform; the usual case is 1, :60)
#pragma unroll
          for (int t = 0; t < 2; ++t) tv[j][t] = k[t] >= kmin ? tv[j][t] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = acc[t] + tv[j][t] * wgt[t] * (c1[j] + c3[j] * sgn[t]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (lvl[t]) g.highOrderFlx[erow + k[t] - 1] = acc[t];
  }
}

// ---- two levels per lane: 16-byte accesses.  The nest is bound by the ISSUE of vector-memory
// instructions (round 3, tools/nlk_probe.py on the 32 x mesh: the edge streams alone run at 6.3 TB/s;
// ten gathers per edge cost the same 0.95 ms whether they hit one L1 line, miss L2 or are masked down to
// 3 lanes, and 20 zero-range buffer loads per edge that touch no memory at all still cost 0.58 ms: the
// texture addresser takes a 64-lane 8-byte instruction at ~16 cycles per CU).  With a lane owning the
// levels 2L+1, 2L+2 every row access is one 16-byte-per-lane instruction: 100 levels = one trip of 50
// lanes, 13 vector-memory instructions per edge instead of 26.  Needs even nvldim and 16-byte aligned
// arrays (else the kernel above).  The range check of a raw buffer descriptor is per dword, so a level
// range that ends between a lane's two levels still masks exactly.
#define NLK_CH2 10  // cells per batch of the two-levels-per-lane kernel (32 x mesh, local connectivity: 5 -> 0.83 ms, 10 -> 0.75 ms)
typedef unsigned int nlk_u32x4 __attribute__((ext_vector_type(4)));
typedef double nlk_f64x2 __attribute__((ext_vector_type(2)));

// The scalar side is the other half of the story: a CU has ONE scalar unit for its four SIMDs, and the
// first form of this kernel spent 457 scalar instructions per edge (rocprofv3 SQ_INSTS_SALU / SQ_WAVES:
// 64-bit pointer arithmetic per cell and array, per-cell buffer descriptors, lane-mask logic for the
// `i < nadv` / `cell in range` guards) -- 510 scalar issue cycles per edge and CU against the ~590 cycles
// per edge and CU the launch took: the nest was SCALAR-ISSUE bound.  So here
//   * the cell loop runs in whole batches of CH cells without guards (a wave-uniform branch), single
//     cells only for a ragged remainder;
//   * every per-cell quantity is a true scalar load (sload) with an unsigned 32-bit index, the tracer
//     columns go through ONE descriptor base whose num_records word is the
//     end of the cell's level range and whose column is the scalar offset operand of the load (the range
//     check sees the sum of scalar and per-lane offset) -- no 64-bit address arithmetic per cell (tables
//     below 4 GiB).
__global__ void __launch_bounds__(64 * NLK_WAVES) nlk_kernel_x2(const NlkArgs g) {
  const int lane = threadIdx.x & 63;
  const int iEdge = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NLK_WAVES + (threadIdx.x >> 6)));
  if (iEdge >= g.nEdges) return;
  const long long erow = (long long)g.nvldim * iEdge;
  int nadv = sload(g.nAdvCellsForEdge + iEdge);
  nadv = nadv < g.nAdv ? nadv : g.nAdv;
  const int* cells = g.advCellsForEdge + (long long)g.nAdv * iEdge;
  const double* c1p = g.advCoefs + (long long)g.nAdv * iEdge;
  const double* c3p = g.advCoefs3rd + (long long)g.nAdv * iEdge;
  // descriptors of the per-cell tables (scalar buffer loads: 32-bit offsets) and of the tracer table
  const unsigned colB = (unsigned)g.nvldim * 8u;   // bytes of a cell's column
  for (int k0 = 0; k0 < g.nVertLevels; k0 += 128) {
    const int kA = k0 + 2 * lane + 1, kB = kA + 1;   // the lane's two (1-based) levels
    const bool lA = kA <= g.nVertLevels, lB = kB <= g.nVertLevels;
    const unsigned koff = (unsigned)(kA - 1) * 8u;
    nlk_f64x2 ntf = {0.0, 0.0}, msk = {0.0, 0.0};
    if (lA) {   // (kB = nVertLevels + 1 reads the row's padding: nvldim is even, so it exists; that half is never stored)
      ntf = *reinterpret_cast<const nlk_f64x2*>(g.normalThicknessFlux + erow + kA - 1);
      msk = *reinterpret_cast<const nlk_f64x2*>(g.advMaskHighOrder + erow + kA - 1);
    }
    double wgt[2], sgn[2], acc[2] = {0.0, 0.0};
    wgt[0] = ntf.x * msk.x; wgt[1] = ntf.y * msk.y;                                        // :126-127
    sgn[0] = __builtin_copysign(1.0, ntf.x); sgn[1] = __builtin_copysign(1.0, ntf.y);      // :128-129
    // a batch of N cells i0 .. i0+N-1 (all of them exist): gathers together, then the sum in order
    auto batch = [&](const int i0, auto n_tag) __attribute__((always_inline)) {
      constexpr int N = decltype(n_tag)::value;
      double tv[N][2], c1[N], c3[N];
      int kmin[N], kmax[N];
      unsigned ic[N];
      bool ok[N];
      // phase 1: the cell indices; phase 2: ALL level-range loads of the batch in flight before any is
      // used (scalar loads return out of order: the first use waits for all of them -- once per batch,
      // not once per cell); phase 3: range ends and gathers
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int iCell = sload(cells + i0 + j);
        ok[j] = (unsigned)(iCell - 1) < (unsigned)g.nCells;   // (the reference would read out of bounds)
        ic[j] = ok[j] ? (unsigned)(iCell - 1) : 0u;
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        kmin[j] = sload_off(g.minLevelCell, ic[j] * 4u);
        kmax[j] = sload_off(g.maxLevelCell, ic[j] * 4u);
        c1[j] = sload(c1p + i0 + j);
        c3[j] = sload(c3p + i0 + j);
      }
      // (the compiler would sink every load next to its use and wait per cell: the empty asm takes all
      //  level ranges as operands, so they are all issued above it and waited for once)
      if constexpr (N == 10)
        asm volatile("" : "+s"(kmin[0]), "+s"(kmin[1]), "+s"(kmin[2]), "+s"(kmin[3]), "+s"(kmin[4]), "+s"(kmin[5]), "+s"(kmin[6]),
                          "+s"(kmin[7]), "+s"(kmin[8]), "+s"(kmin[9]), "+s"(kmax[0]), "+s"(kmax[1]), "+s"(kmax[2]), "+s"(kmax[3]),
                          "+s"(kmax[4]), "+s"(kmax[5]), "+s"(kmax[6]), "+s"(kmax[7]), "+s"(kmax[8]), "+s"(kmax[9]));
      else if constexpr (N == 5)
        asm volatile("" : "+s"(kmin[0]), "+s"(kmin[1]), "+s"(kmin[2]), "+s"(kmin[3]), "+s"(kmin[4]), "+s"(kmax[0]), "+s"(kmax[1]),
                          "+s"(kmax[2]), "+s"(kmax[3]), "+s"(kmax[4]));
#pragma unroll
      for (int j = 0; j < N; ++j) {
        // levels 1 .. kmax of the cell's column; the range check sees scalar + per-lane offset, so the range
        // ends at column start + 8 kmax (kmax <= 0 or a cell out of range: ends at the column start = empty)
        int km = kmax[j] < g.nVertLevels ? kmax[j] : g.nVertLevels;
        km = ok[j] ? (km > 0 ? km : 0) : 0;
        const unsigned cb = ic[j] * colB;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.tracerCur), (short)0, (int)(cb + (unsigned)km * 8u), 0x00020000);
        const nlk_f64x2 v = __builtin_bit_cast(nlk_f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)koff, (int)cb, 0));
        tv[j][0] = v.x; tv[j][1] = v.y;
        c3[j] = c3[j] * g.coef3rdOrder;
      }
      int kmin_max = kmin[0];
#pragma unroll
      for (int j = 1; j < N; ++j) kmin_max = kmin_max > kmin[j] ? kmin_max : kmin[j];
      if (kmin_max > 1) {   // some cell of the batch starts above level 1 (wave-uniform; the usual case is 1, :60)
#pragma unroll
        for (int j = 0; j < N; ++j) {
          tv[j][0] = kA >= kmin[j] ? tv[j][0] : 0.0;
          tv[j][1] = kB >= kmin[j] ? tv[j][1] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < N; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = acc[t] + tv[j][t] * wgt[t] * (c1[j] + c3[j] * sgn[t]);   // :136-148
    };
    constexpr int CH = NLK_CH2;
    int i0 = 0;
    for (; i0 + CH <= nadv; i0 += CH) batch(i0, std::integral_constant<int, CH>{});
    for (; i0 < nadv; ++i0) batch(i0, std::integral_constant<int, 1>{});
    if (lB) *reinterpret_cast<nlk_f64x2*>(g.highOrderFlx + erow + kA - 1) = nlk_f64x2{acc[0], acc[1]};
    else if (lA) g.highOrderFlx[erow + kA - 1] = acc[0];
  }
}

// ---- the same, with the per-cell bookkeeping in the VECTOR unit.  Even batched, the scalar form above
// spends ~25 scalar instructions per cell on the CU's one scalar unit (index checks, two level-range
// loads, clamps, range end, column offset).  Here lane i < nadv owns cell i of the edge and does all of
// that once, for all cells at the same time, in a dozen vector instructions (the cell list, the two
// coefficients and the level range arrive as vector loads of 10 lanes); what the gather of cell i needs
// wave-uniform -- column offset and range end -- comes out of the lanes by v_readlane (executed by the
// vector pipes, of which a CU has four).  Needs nAdv <= 64.
__global__ void __launch_bounds__(64 * NLK_WAVES) nlk_kernel_x2v(const NlkArgs g) {
  const int lane = threadIdx.x & 63;
  const int iEdge = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NLK_WAVES + (threadIdx.x >> 6)));
  if (iEdge >= g.nEdges) return;
  const long long erow = (long long)g.nvldim * iEdge;
  int nadv = sload(g.nAdvCellsForEdge + iEdge);
  nadv = nadv < g.nAdv ? nadv : g.nAdv;
  const unsigned colB = (unsigned)g.nvldim * 8u;   // bytes of a cell's column
  // ---- lane i: cell i of the edge
  const bool mine = lane < nadv;
  const long long cpos = (long long)g.nAdv * iEdge + lane;
  const int cellv = mine ? g.advCellsForEdge[cpos] : 0;
  const double c1v = mine ? g.advCoefs[cpos] : 0.0;
  const double c3v = (mine ? g.advCoefs3rd[cpos] : 0.0) * g.coef3rdOrder;
  const bool okv = (unsigned)(cellv - 1) < (unsigned)g.nCells;   // (a cell out of range: the reference would read out of bounds)
  const unsigned icv = okv ? (unsigned)(cellv - 1) : 0u;
  const int kminv = okv ? g.minLevelCell[icv] : 1;
  int kmv = okv ? g.maxLevelCell[icv] : 0;
  kmv = kmv < g.nVertLevels ? kmv : g.nVertLevels;
  kmv = kmv > 0 ? kmv : 0;
  // levels 1 .. kmax of the cell's column: the range check sees scalar + per-lane offset, so the range ends
  // at column start + 8 kmax (nothing to read: ends at the column start)
  const unsigned cbv = icv * colB;
  const unsigned nrecv = cbv + (unsigned)kmv * 8u;
  const bool any_kmin = __builtin_amdgcn_ballot_w64(kminv > 1) != 0ull;   // some cell starts above level 1 (usual case: none, :60)
  auto rl_d = [](const double v, const int j) __attribute__((always_inline)) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), j), __builtin_amdgcn_readlane(__double2loint(v), j));
  };
  for (int k0 = 0; k0 < g.nVertLevels; k0 += 128) {
    const int kA = k0 + 2 * lane + 1, kB = kA + 1;   // the lane's two (1-based) levels
    const bool lA = kA <= g.nVertLevels, lB = kB <= g.nVertLevels;
    const unsigned koff = (unsigned)(kA - 1) * 8u;
    nlk_f64x2 ntf = {0.0, 0.0}, msk = {0.0, 0.0};
    if (lA) {   // (kB = nVertLevels + 1 reads the row's padding: nvldim is even, so it exists; that half is never stored)
      ntf = *reinterpret_cast<const nlk_f64x2*>(g.normalThicknessFlux + erow + kA - 1);
      msk = *reinterpret_cast<const nlk_f64x2*>(g.advMaskHighOrder + erow + kA - 1);
    }
    double wgt[2], sgn[2], acc[2] = {0.0, 0.0};
    wgt[0] = ntf.x * msk.x; wgt[1] = ntf.y * msk.y;                                        // :126-127
    sgn[0] = __builtin_copysign(1.0, ntf.x); sgn[1] = __builtin_copysign(1.0, ntf.y);      // :128-129
    constexpr int CH = NLK_CH2;
    // a batch of CH cells i0 .. i0+CH-1.  FULL: all of them exist (the usual case: every edge of the reference's
    // mesh has nAdv cells) -- no guards: the `i < nadv` compares, selects and branches of the guarded form were 8 of
    // the ~10 scalar instructions per gather on the CU's one scalar unit (round 4: the nest is scalar-issue bound)
    auto batch = [&](const int i0, auto full_tag) __attribute__((always_inline)) {
      constexpr bool FULL = decltype(full_tag)::value;
      double tv[CH][2];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const bool have = FULL || i0 + j < nadv;       // (wave-uniform; a missing cell: empty range, reads zeros)
        const int jj = have ? i0 + j : 0;
        const unsigned cb = (unsigned)__builtin_amdgcn_readlane((int)cbv, jj);
        const unsigned nrec = have ? (unsigned)__builtin_amdgcn_readlane((int)nrecv, jj) : 0u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(g.tracerCur), (short)0, (int)nrec, 0x00020000);
        const nlk_f64x2 v = __builtin_bit_cast(nlk_f64x2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)koff, (int)cb, 0));
        tv[j][0] = v.x; tv[j][1] = v.y;
      }
      if (any_kmin) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int kmin = __builtin_amdgcn_readlane(kminv, (FULL || i0 + j < nadv) ? i0 + j : 0);
          tv[j][0] = kA >= kmin ? tv[j][0] : 0.0;
          tv[j][1] = kB >= kmin ? tv[j][1] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int jj = (FULL || i0 + j < nadv) ? i0 + j : 0;
        const double c1 = rl_d(c1v, jj), c3 = rl_d(c3v, jj);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = acc[t] + tv[j][t] * wgt[t] * (c1 + c3 * sgn[t]);   // :136-148
      }
    };
    int i0 = 0;
    if (nadv == CH) {   // (the reference's own shape: one batch, lane indices known at compile time)
      batch(0, std::true_type{});
      i0 = CH;
    }
    for (; i0 + CH <= nadv; i0 += CH) batch(i0, std::true_type{});
    if (i0 < nadv) batch(i0, std::false_type{});
    if (lB) *reinterpret_cast<nlk_f64x2*>(g.highOrderFlx + erow + kA - 1) = nlk_f64x2{acc[0], acc[1]};
    else if (lA) g.highOrderFlx[erow + kA - 1] = acc[0];
  }
}

// mode: -1 automatic, 0 one level per lane, 1 two levels per lane with scalar bookkeeping, 2 ... with vector bookkeeping
void launch(const NlkArgs& g, void* stream, int mode) {
  const unsigned gx = (unsigned)((g.nEdges + NLK_WAVES - 1) / NLK_WAVES);
  const uintptr_t al = (uintptr_t)g.tracerCur | (uintptr_t)g.normalThicknessFlux | (uintptr_t)g.advMaskHighOrder |
                       (uintptr_t)g.highOrderFlx;
  const bool can = (g.nvldim & 1) == 0 && (al & 15) == 0 && (double)g.nCells * g.nvldim * 8.0 < 4294967000.0;
  if (can && g.nAdv <= 64 && (mode == 2 || mode < 0)) hipLaunchKernelGGL(nlk_kernel_x2v, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g);
  else if (can && mode != 0) hipLaunchKernelGGL(nlk_kernel_x2, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(nlk_kernel, dim3(gx), dim3(64 * NLK_WAVES), 0, (hipStream_t)stream, g);
}

}  // namespace NLK_NS
