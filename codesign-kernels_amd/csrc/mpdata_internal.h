// mpdata_internal.h -- what the translation units of libmpdata_hip.so share behind the C-ABI of
// include/mpdata_hip.h: the error text, the variant / tile / layout settings, argument validation and tile
// choice, the staggered device arena, the reference-layout launch (advect_device), device guards.
//   mpdata_core.hip      errors, settings, tile choice, reference-layout device calls, device utilities
//   mpdata_plan.hip      plans (device state in the library's own layout), run / run_uw dispatch, multi-GPU handles
//   mpdata_hostcall.hip  the host-array calls (chunked, pipelined H2D / kernel / D2H)
//   mpdata_multi.hip     multi-GPU orchestration on top of single-device plans;   mpdata_diag.hip  stream ceilings
#ifndef MPDATA_INTERNAL_H
#define MPDATA_INTERNAL_H
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "mpdata_args.h"
#include "mpdata_hip.h"
#include "mpdata_layout.h"
#include "mpdata_multi.h"

// the two arithmetic variants of the kernels (mpdata_kernels_exact.hip / mpdata_kernels_fast.hip via mpdata_kernels_inst.h)
#define MPDATA_VARIANT_NS(NS)                                                                   \
  namespace NS {                                                                                \
  const char* build_flags();                                                                    \
  int max_tile_id();                                                                            \
  bool tile_info(int id, MpdataTileInfo* info);                                                 \
  bool launch(int id, const MpdataArgs& a, int ntracers, void* stream);                         \
  bool launch_f32(int id, const MpdataArgsF32& a, int ntracers, void* stream);                  \
  bool launch_wm(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);             \
  bool launch_wm_f32(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);         \
  bool launch_wm_uw(int lps, const MpdataWmArgs& a, void* stream, bool conv);                   \
  }
MPDATA_VARIANT_NS(mpdata_exact)
MPDATA_VARIANT_NS(mpdata_fast)
#undef MPDATA_VARIANT_NS

#pragma GCC visibility push(hidden)
namespace mpd {

// thread-local error text behind mpdata_last_error(); returns `code`
int set_err(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_err(hipError_t e, const char* what);

// settings (environment on first use, or the mpdata_set_* calls)
int variant();                 // MPDATA_VARIANT_*
int tile_override();           // kernel tiling id forced by MPDATA_TILE / mpdata_set_tile; -1 = automatic
int plan_layout_default();     // MPDATA_LAYOUT_* new plans get
int serpentine();              // serpentine tile order of wave-major plans (off by default)
int wm_flags();                // MPDATA_WMF_* test switches of the wave-major launch
int wm_wpb();                  // waves (tiles) per workgroup of the wave-major kernels
unsigned long long* debug_buffer();   // diagnostic builds: per-wave stamp buffer (mpdata_set_debug_buffer), else null

int validate(int64_t ncrms, int nx, int nz, int ntracers);
int choose_tile(int var, int64_t ncrms, int nx, int nz, MpdataTileInfo* out, int elem_bytes = 8);

struct Sizes {
  size_t f, u, w, k, kz;  // elements
};
Sizes sizes_of(int64_t ncrms, int nx, int nz, int ntracers);
unsigned grid_for(long long total, int block);

// Device buffers the library owns (plans, the host-array calls) come out of one allocation
// per set, with f, u and w placed at DIFFERENT offsets modulo 1 KiB.  A workgroup reads the
// same instance range of all three arrays at about the same time; with the three bases
// equally aligned those requests land on the same HBM channel, and the kernel runs 8 %
// slower (ncrms = 65536: 0.494 vs 0.458 ms, tools/placement3.py).
struct Arena {
  void* base = nullptr;
  void* p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // f,u,w,rho,rhow,adz,flux
};
hipError_t arena_alloc(Arena& a, const size_t bytes[7]);
void arena_free(Arena& a);
// the same placement inside an allocation that is kept between calls (cap = its size; grown when needed)
hipError_t arena_place(Arena& a, size_t& cap, const size_t bytes[7]);

// EXACT: flux in the reference's summation order (bit-identical) unless MPDATA_EXACT_FLUX=sum; parked in registers
// where the kernel has a form for it unless MPDATA_EXACT_FLUX=hbm
bool exact_flux_in_order();
bool exact_flux_in_regs();

// one call on reference-layout device arrays (x-march / k-march kernels); var: MPDATA_VARIANT_* (< 0: the global one);
// staged: 65 <= nz <= 238 may go through the calling thread's wave-major plan (staged_device_call below)
template <typename R>
int advect_device(int64_t ncrms, int nx, int nz, int ntracers, R* f, const R* u, const R* w, const R* rho, const R* rhow,
                  const R* adz, R* flux, void* stream, int var = -1, bool staged = true);
extern template int advect_device<double>(int64_t, int, int, int, double*, const double*, const double*, const double*,
                                          const double*, const double*, double*, void*, int, bool);
extern template int advect_device<float>(int64_t, int, int, int, float*, const float*, const float*, const float*,
                                         const float*, const float*, float*, void*, int, bool);

// frees the calling thread's park buffers of the EXACT device calls (mpdata_core.hip: park_buffer)
void park_buffers_release();
// calls on reference-layout device arrays at 65 <= nz <= 238 through a wave-major plan kept per host thread
// (mpdata_plan.hip); eb = bytes per real
bool staged_call_applies(int64_t ncrms, int nz, int eb);
int staged_device_call(int eb, int64_t ncrms, int nx, int nz, int ntracers, void* f, const void* u, const void* w, const void* rho,
                       const void* rhow, const void* adz, void* flux, void* stream, int var);
void staged_plan_release();

// the device a pointer lives on (the current one if HIP does not know the pointer)
int device_of(const void* p);
// switch to a device for a scope
struct DevGuard {
  int prev = -1, dev;
  explicit DevGuard(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DevGuard() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

}  // namespace mpd
#pragma GCC visibility pop

#define HIP_TRY(expr)                                     \
  do {                                                    \
    hipError_t e_ = (expr);                               \
    if (e_ != hipSuccess) return mpd::hip_err(e_, #expr); \
  } while (0)

#endif
