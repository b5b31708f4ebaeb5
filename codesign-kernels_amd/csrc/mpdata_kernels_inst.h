// mpdata_kernels_inst.h -- instantiates the tilings of one arithmetic variant.
// Included by mpdata_kernels_exact.hip / mpdata_kernels_fast.hip after
// defining MPDATA_NS; exports <MPDATA_NS>::launch(tile, ...).
#include "mpdata_kernel_body.h"

namespace MPDATA_NS {

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
  X(4, 9, 4, 4)

int num_tiles() {
  int n = 0;
#define X(id, W, SPW, NWV) ++n;
  MPDATA_TILES(X)
#undef X
  return n;
}

bool tile_info(int id, MpdataTileInfo* info) {
#define X(ID, W_, SPW_, NWV_)                                                     \
  if (id == ID) {                                                                 \
    using T = Tile<W_, SPW_, NWV_>;                                               \
    *info = MpdataTileInfo{ID, W_, SPW_, NWV_, T::SLW, T::NCOL, T::THREADS,       \
                           "W" #W_ "_SPW" #SPW_ "_NWV" #NWV_};                    \
    return true;                                                                  \
  }
  MPDATA_TILES(X)
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
  return false;
}

}  // namespace MPDATA_NS
