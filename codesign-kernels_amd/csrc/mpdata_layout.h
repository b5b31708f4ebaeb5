// mpdata_layout.h -- host interface of the layout-conversion kernels (mpdata_layout.hip).
#ifndef MPDATA_LAYOUT_H
#define MPDATA_LAYOUT_H
#include <hip/hip_runtime.h>

// One array (all tracers of it) between the two layouts.
//   reference side: element (sl, column cs, level kk) of tracer tr at
//       ref + tr*ref_tstride + sl + ncrms*(cs*ref_colmul + kk*ref_levmul)
//     f, u, w: ref_colmul = 1, ref_levmul = number of columns; rho, rhow, adz, flux: one
//     column, ref_levmul = 1
//   private side: element e = s*nlev + kk (instance-in-tile s, level kk) of column
//   c = cs + prv_col0 of tile t at prv + tr*prv_tstride + t*prv_tile_stride +
//       c*chunk + e                                              (main_e == 0: unsplit arrays)
//       c*main_e + e                    if e <  main_e           (f, u, w: the whole 128-byte
//       ncol_p*main_e + c*rem_e + e - main_e   otherwise          lines of a column first)
struct MpdataLayoutJob {
  void* ref;
  void* prv;
  long long ncrms;
  int ncols;                 // columns of the reference-side array that are converted
  int nlev;                  // levels converted (nzm)
  int ntr;
  long long ref_colmul, ref_levmul, ref_tstride;
  int slp;                   // instances per tile
  int ntiles;
  int prv_col0;              // column shift: u, w start at c = 1 of the private column index
  long long chunk;           // slp * nlev
  long long main_e;          // elements of the line-aligned part of a column chunk (0: array not split)
  int ncol_p;                // column slots of a tile on the private side (split arrays)
  long long prv_tile_stride, prv_tstride;
};

hipError_t mpdata_layout_convert(const MpdataLayoutJob& j, int elem_bytes, bool to_private, hipStream_t stream);

// the split arrays (f, u, w: main_e > 0) column-walking: one workgroup per 64 (32) instances and array,
// all columns; nj = 1, or 2 arrays of equal nlev in one launch (u and w of an import)
hipError_t mpdata_layout_convert_cols(const MpdataLayoutJob* jobs, int nj, bool to_private, hipStream_t stream);

// import (reference -> plan) of f, u, w by 128-byte row segments through LDS-DMA; hipErrorNotSupported: conditions not
// met (odd ncrms, unaligned base, array of 4 GiB or more), take mpdata_layout_convert_cols
hipError_t mpdata_layout_import_rows(const MpdataLayoutJob* jobs, int nj, hipStream_t stream);

#endif
