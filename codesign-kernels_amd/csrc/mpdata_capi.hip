// mpdata_capi.hip -- the C-ABI of libmpdata_hip.so (include/mpdata_hip.h):
// argument validation, tiling choice, plan/buffer management, the synthetic
// input generator and the shard pack/unpack kernels.  No CPU compute path
// exists in this library: every entry point either runs HIP kernels or
// returns an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>

#include "mpdata_args.h"
#include "mpdata_hip.h"
#include "mpdata_layout.h"
#include "mpdata_multi.h"

namespace mpdata_exact {
const char* build_flags();
int max_tile_id();
bool tile_info(int id, MpdataTileInfo* info);
bool launch(int id, const MpdataArgs& a, int ntracers, void* stream);
bool launch_f32(int id, const MpdataArgsF32& a, int ntracers, void* stream);
}  // namespace mpdata_exact
namespace mpdata_fast {
const char* build_flags();
int max_tile_id();
bool tile_info(int id, MpdataTileInfo* info);
bool launch(int id, const MpdataArgs& a, int ntracers, void* stream);
bool launch_f32(int id, const MpdataArgsF32& a, int ntracers, void* stream);
}  // namespace mpdata_fast

namespace {

thread_local std::string g_err;
int g_variant = -1;  // -1: read MPDATA_VARIANT on first use
int g_tile = -2;     // -2: read MPDATA_TILE on first use; -1: automatic
unsigned long long* g_dbg = nullptr;  // diagnostic stamp buffer (device), see mpdata_set_debug_buffer

int set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace
int mpdata_internal_set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
namespace {
int hip_err(hipError_t e, const char* what) {
  return set_err((int)e, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr)                                  \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) return hip_err(e_, #expr);   \
  } while (0)

int variant() {
  if (g_variant < 0) {
    const char* v = getenv("MPDATA_VARIANT");
    g_variant = (v && (!strcmp(v, "fast") || !strcmp(v, "1"))) ? MPDATA_VARIANT_FAST
                                                               : MPDATA_VARIANT_EXACT;
  }
  return g_variant;
}
int tile_override() {
  if (g_tile == -2) {
    const char* v = getenv("MPDATA_TILE");
    g_tile = v ? atoi(v) : -1;
  }
  return g_tile;
}

bool get_tile(int var, int id, MpdataTileInfo* t) {
  return var == MPDATA_VARIANT_FAST ? mpdata_fast::tile_info(id, t) : mpdata_exact::tile_info(id, t);
}

// Automatic choice: the x-marching kernel with the fewest lanes per instance
// that holds nz (lanes along k, any nx); if nz is too large for one wave, the
// k-marching kernel with the smallest column coverage that fits nx.
// fp32 (elem_bytes = 4): the two-instances-per-lane kernels (tile ids >= 40) when ncrms is
// even, else the one-instance-per-lane ones (nz <= 32).
int choose_tile(int var, int64_t ncrms, int nx, int nz, MpdataTileInfo* out, int elem_bytes = 8) {
  // the x-marching kernels use 32-bit byte offsets relative to the first row a wave touches
  // (at most 4 rows = levels of one column) plus a 32-bit column offset: arrays may exceed
  // 4 GiB as long as four levels of one array stay below it
  const bool small32 = (double)ncrms * (nx + 6) * 4.0 * (double)elem_bytes < 4294967000.0;
  // the k-marching kernels (fp64 only) use 32-bit byte offsets inside one k-plane
  const bool plane31 = (double)ncrms * (nx + 8) * 8.0 < 2147483648.0;
  int forced = tile_override();
  MpdataTileInfo t;
  if (forced >= 0 && get_tile(var, forced, &t) && t.elem_bytes != elem_bytes) forced = -1;  // other precision
  if (forced >= 0) {
    if (!get_tile(var, forced, &t)) return set_err(MPDATA_EINVAL, "unknown tile id %d", forced);
    if (t.id >= 40 && (ncrms & 1))
      return set_err(MPDATA_EUNSUPPORTED, "tile %s needs an even ncrms", t.name);
    if (t.ncol < nx + 4 || t.nz_max < nz || (t.nz_max < (1 << 30) ? !small32 : !plane31))
      return set_err(MPDATA_EUNSUPPORTED, "tile %s covers %d columns / nz<=%d; nx=%d nz=%d", t.name,
                     t.ncol, t.nz_max, nx, nz);
    *out = t;
    return 0;
  }
  int best = -1, best_cost = 1 << 30;
  const int n = (var == MPDATA_VARIANT_FAST ? mpdata_fast::max_tile_id() : mpdata_exact::max_tile_id()) + 1;
  for (int id = 0; id < n; ++id) {
    if (!get_tile(var, id, &t)) continue;
    if (t.elem_bytes != elem_bytes) continue;
    if (t.id >= 40 && (ncrms & 1)) continue;
    if (t.ncol < nx + 4 || t.nz_max < nz) continue;
    if (t.nz_max < (1 << 30) ? !small32 : !plane31) continue;
    // x-marching tiles (finite nz_max) first, by lanes per instance; then k-marching by columns
    // (among x-marching tiles of equal lanes-per-instance the smaller workgroup is the default)
    const int cost = t.nz_max < (1 << 30) ? t.nz_max * 100 + t.slw - (t.id >= 40 ? 50 : 0) : 100000 + t.ncol;
    if (cost < best_cost) { best = id; best_cost = cost; }
  }
  if (best < 0)
    return set_err(MPDATA_EUNSUPPORTED,
                   elem_bytes == 8 ? "no kernel tiling covers nx=%d nz=%d at this ncrms (need nz<=64 with 4 levels of one "
                                     "array < 4 GiB, or nx<=140 with one k-plane < 2 GiB)"
                                   : "no fp32 kernel tiling covers nx=%d nz=%d (need nz<=64, and nz<=32 for odd ncrms)",
                   nx, nz);
  get_tile(var, best, out);
  return 0;
}

int validate(int64_t ncrms, int nx, int nz, int ntracers) {
  if (ncrms < 1 || nx < 1 || nz < 3 || ntracers < 1)
    return set_err(MPDATA_EINVAL, "bad sizes ncrms=%lld nx=%d nz=%d ntracers=%d (need >=1,>=1,>=3,>=1)",
                   (long long)ncrms, nx, nz, ntracers);
  if (ntracers > 65535) return set_err(MPDATA_EUNSUPPORTED, "ntracers > 65535");
  // one workgroup per (tracer, group of >= 16 instances) in a 1-D grid
  if ((double)ntracers * (double)((ncrms + 15) / 16) > 2147483647.0)
    return set_err(MPDATA_EUNSUPPORTED, "ntracers * ncrms/16 must be < 2^31 workgroups per call");
  return 0;
}

struct Sizes {
  size_t f, u, w, k, kz;  // elements
};
Sizes sizes_of(int64_t ncrms, int nx, int nz, int ntracers) {
  Sizes s;
  const size_t n = (size_t)ncrms, nzm = (size_t)nz - 1;
  s.f = n * (nx + 6) * nzm * (size_t)ntracers;
  s.u = n * (nx + 5) * nzm;
  s.w = n * (nx + 4) * (size_t)nz;
  s.k = n * nzm;
  s.kz = n * (size_t)nz;
  return s;
}

// ---- synthetic generator (same law as oracle/mpdata_oracle.c:mpdata_oracle_fill)
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <typename R>
__global__ void fill_kernel(R* a, unsigned long long base, double shift, long long rows,
                            long long ng, long long sl0, long long nloc) {
  const long long total = rows * nloc;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / nloc, s = t - r * nloc;
    const unsigned long long j = (unsigned long long)(r * ng + sl0 + s);
    const unsigned long long z = mix64(base + (j + 1) * 0x9E3779B97F4A7C15ull);
    a[t] = (R)__dadd_rn((double)(z >> 11) * 0x1.0p-53, shift);  // fp32: the fp64 value, rounded
  }
}

__global__ void pack_kernel(const double* full, double* shard, long long rows, long long ncrms,
                            long long sl0, long long nloc, int unpack) {
  const long long total = rows * nloc;
  double* fullw = const_cast<double*>(full);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / nloc, s = t - r * nloc;
    if (unpack) fullw[r * ncrms + sl0 + s] = shard[t];
    else shard[t] = full[r * ncrms + sl0 + s];
  }
}

// EXACT, bit-identical flux: the plan kernels store the upwind sum as flux and PARK the nx limited vertical fluxes of
// every lane ([tracer][tile][column][lane], mpdata_kernel_wm_body.h); this adds them, one by one in the reference's
// order i = 1 .. nx (:624), onto that sum.  A wave per (tracer, tile), lane -> (instance, level) as in the kernels.
// R2 = double, or float2 for fp32 plans (two adjacent instances per lane; no contraction: this file is built with
// -ffp-contract=off).
template <typename R2>
__global__ void __launch_bounds__(256) flux_finish_kernel(R2* flux, const R2* park, int ntiles, int nx, int nzm, int lps,
                                                          long long flux_tstride, int ntr) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave = tracer * ntiles + tile
  if (w >= (long long)ntr * ntiles) return;
  const int tr = (int)(w / ntiles), tile = (int)(w % ntiles);
  const int s_l = lane / lps, kk = lane % lps;
  if (kk >= nzm) return;
  const int chunk = (64 / lps) * nzm;
  R2* fp = flux + (long long)tr * flux_tstride + (long long)tile * chunk + s_l * nzm + kk;
  const R2* pp = park + w * ((long long)nx * 64) + lane;
  R2 acc = *fp;
  for (int i = 0; i < nx; ++i) {
    const R2 t = pp[(long long)i * 64];
    if constexpr (sizeof(R2) == 8 && alignof(R2) == 8 && !std::is_same<R2, double>::value) { acc.x = acc.x + t.x; acc.y = acc.y + t.y; }
    else acc = acc + t;
  }
  *fp = acc;
}

// the kernel launch(es) of one run of a single-device plan, on the plan's stream
// (u_ref, w_ref != null: the kernel that reads u, w from these reference-layout arrays)
unsigned grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

// Device buffers the library owns (plans, the host-array calls) come out of one allocation
// per set, with f, u and w placed at DIFFERENT offsets modulo 1 KiB.  A workgroup reads the
// same instance range of all three arrays at about the same time; with the three bases
// equally aligned those requests land on the same HBM channel, and the kernel runs 8 %
// slower (ncrms = 65536: 0.494 vs 0.458 ms, tools/placement3.py).
struct Arena {
  void* base = nullptr;
  void* p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // f,u,w,rho,rhow,adz,flux
};
hipError_t arena_alloc(Arena& a, const size_t bytes[7]) {
  static const size_t residue[7] = {0, 256, 512, 768, 0, 0, 0};
  size_t off = 0, offs[7];
  for (int i = 0; i < 7; ++i) {
    off = (off + 1023) / 1024 * 1024 + residue[i];
    offs[i] = off;
    off += bytes[i];
  }
  hipError_t e = hipMalloc(&a.base, off + 1024);
  if (e != hipSuccess) { a.base = nullptr; return e; }
  const uintptr_t b0 = ((uintptr_t)a.base + 1023) / 1024 * 1024;
  for (int i = 0; i < 7; ++i) a.p[i] = (void*)(b0 + offs[i]);
  return hipSuccess;
}
void arena_free(Arena& a) {
  if (a.base) (void)hipFree(a.base);
  a = Arena();
}
// the same placement inside an allocation that is kept between calls (cap = its size; grown when needed)
hipError_t arena_place(Arena& a, size_t& cap, const size_t bytes[7]) {
  static const size_t residue[7] = {0, 256, 512, 768, 0, 0, 0};
  size_t off = 0, offs[7];
  for (int i = 0; i < 7; ++i) {
    off = (off + 1023) / 1024 * 1024 + residue[i];
    offs[i] = off;
    off += bytes[i];
  }
  if (!a.base || off + 1024 > cap) {
    if (a.base) (void)hipFree(a.base);
    a = Arena();
    cap = 0;
    const hipError_t e = hipMalloc(&a.base, off + 1024);
    if (e != hipSuccess) { a.base = nullptr; return e; }
    cap = off + 1024;
  }
  const uintptr_t b0 = ((uintptr_t)a.base + 1023) / 1024 * 1024;
  for (int i = 0; i < 7; ++i) a.p[i] = (void*)(b0 + offs[i]);
  return hipSuccess;
}

// EXACT plans: flux bit-identical to the reference (the limited vertical fluxes parked and added in the reference's
// order) unless MPDATA_EXACT_FLUX=sum
bool exact_flux_in_order() {
  static const bool sum = getenv("MPDATA_EXACT_FLUX") && !strcmp(getenv("MPDATA_EXACT_FLUX"), "sum");
  return !sum;
}
// ... and where they are parked: in REGISTERS where the kernel has a form for it (wave-major plans with nx <= MPDATA_WM_NPK:
// no park array, no finishing kernel, round 5) unless MPDATA_EXACT_FLUX=hbm (round 4's park array everywhere; A/B)
bool exact_flux_in_regs() {
  static const bool hbm = getenv("MPDATA_EXACT_FLUX") && !strcmp(getenv("MPDATA_EXACT_FLUX"), "hbm");
  return exact_flux_in_order() && !hbm;
}
// var: MPDATA_VARIANT_* (a plan passes the variant it was created with; < 0: the global one)
template <typename R>
int advect_device(int64_t ncrms, int nx, int nz, int ntracers, R* f, const R* u, const R* w,
                  const R* rho, const R* rhow, const R* adz, R* flux, void* stream, int var = -1) {
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  if (!f || !u || !w || !rho || !rhow || !adz || !flux)
    return set_err(MPDATA_EINVAL, "null array pointer");
  if (var < 0) var = variant();
  MpdataTileInfo t;
  rc = choose_tile(var, ncrms, nx, nz, &t, (int)sizeof(R));
  if (rc) return rc;
  MpdataArgsT<R> a;
  a.f = f; a.u = u; a.w = w; a.rho = rho; a.rhow = rhow; a.adz = adz; a.flux = flux;
  a.ncrms = ncrms; a.nx = nx; a.nz = nz; a.ntracers = ntracers;
  a.f_tstride = (long long)ncrms * (nx + 6) * (nz - 1);
  a.flux_tstride = (long long)ncrms * nz;
  a.dbg = g_dbg;
  // EXACT, x-marching kernels: the park array of the bit-identical flux (see xmarch_flux_finish_kernel), allocated and
  // freed in stream order around the launch: [workgroup][nx][thread].  (The k-marching kernels add in the reference's
  // order by construction; FAST never parks; MPDATA_EXACT_FLUX=sum does without.)
  a.wpark = nullptr;
  void* park_mem = nullptr;
  const bool big = (double)ncrms * (nx + 6) * nz * (double)(t.id >= 40 ? 8 : t.elem_bytes) >= 4294967000.0 * (t.id >= 40 ? 2 : 1);
  if (var == MPDATA_VARIANT_EXACT && t.nz_max < (1 << 30) && !big && exact_flux_in_order()) {
    const size_t groups = (size_t)((ncrms + t.slw - 1) / t.slw);
    const size_t esz = t.id >= 40 ? 8 : (size_t)t.elem_bytes;   // (tile ids 40..: two fp32 instances per lane)
    const size_t bytes = (size_t)ntracers * groups * (size_t)nx * (size_t)t.threads * esz;
    HIP_TRY(hipMallocAsync(&park_mem, bytes, (hipStream_t)stream));
    a.wpark = (R*)park_mem;
  }
  bool ok;
  if constexpr (sizeof(R) == 8)
    ok = var == MPDATA_VARIANT_FAST ? mpdata_fast::launch(t.id, a, ntracers, stream)
                                    : mpdata_exact::launch(t.id, a, ntracers, stream);
  else
    ok = var == MPDATA_VARIANT_FAST ? mpdata_fast::launch_f32(t.id, a, ntracers, stream)
                                    : mpdata_exact::launch_f32(t.id, a, ntracers, stream);
  if (park_mem) (void)hipFreeAsync(park_mem, (hipStream_t)stream);
  if (!ok) return set_err(MPDATA_EINVAL, "tile %d not instantiated", t.id);
  HIP_TRY(hipGetLastError());
  return 0;
}

// the device a pointer lives on (the current one if HIP does not know the pointer)
int device_of(const void* p) {
  int cur = 0;
  (void)hipGetDevice(&cur);
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return cur;
  }
  return at.type == hipMemoryTypeDevice ? at.device : cur;
}
struct DevGuardP {   // (DevGuard is declared further down, beside the plans)
  int prev = -1, dev;
  explicit DevGuardP(int d) : dev(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DevGuardP() {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
};

template <typename R>
int fill_device(R* a, int sid, int64_t rows, int64_t ncrms_global, int64_t sl0, int64_t nloc,
                uint64_t seed, int dist, void* stream) {
  if (!a || sid < 0 || sid > 6 || rows < 1 || nloc < 1 || sl0 < 0 || sl0 + nloc > ncrms_global ||
      dist < 1 || dist > 3)
    return set_err(MPDATA_EINVAL, "bad argument to mpdata_fill_synthetic_device");
  double shift = 0.0;
  if (dist == 1) {
    if (sid == 2 || sid == 3) shift = -0.5;
    else if (sid == 0 || sid == 4 || sid == 5) shift = 0.5;
  } else if (dist == 3) {
    if (sid == 2 || sid == 3) shift = -0.5;
  }
  const unsigned long long base = seed + (unsigned long long)sid * 0xD1B54A32D192ED03ull;
  // (a null stream: the array's own device -- the Fortran driver fills arrays on a plan's root GPU)
  int cur_dev = 0;
  (void)hipGetDevice(&cur_dev);
  DevGuardP g(stream ? cur_dev : device_of(a));
  hipLaunchKernelGGL(fill_kernel<R>, dim3(grid_for(rows * nloc, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, base, shift, (long long)rows, (long long)ncrms_global,
                     (long long)sl0, (long long)nloc);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int mpdata_advect_scalar2d_device(int64_t ncrms, int nx, int nz, int ntracers, double* f,
                                  const double* u, const double* w, const double* rho,
                                  const double* rhow, const double* adz, double* flux,
                                  void* stream) {
  return advect_device<double>(ncrms, nx, nz, ntracers, f, u, w, rho, rhow, adz, flux, stream);
}

int mpdata_advect_scalar2d_f32_device(int64_t ncrms, int nx, int nz, int ntracers, float* f,
                                      const float* u, const float* w, const float* rho,
                                      const float* rhow, const float* adz, float* flux,
                                      void* stream) {
  return advect_device<float>(ncrms, nx, nz, ntracers, f, u, w, rho, rhow, adz, flux, stream);
}

}  // extern "C"

// ---- plans ---------------------------------------------------------------------------------
// A plan owns the device state of one problem (what the OpenACC `enter data pcreate` of the
// reference does, :105, :280, :662) on the device that was current when it was created; every
// plan call switches to that device and back.  Variant and layout are fixed at creation.
// fp64 plans with nz <= 64 keep the arrays in the wave-major layout of
// mpdata_kernel_wm_body.h and convert in upload / download / import / export; other plans
// (fp32; nz > 64) keep the reference layout and run the x-/k-marching kernels.
namespace mpdata_exact {
bool launch_wm(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);
bool launch_wm_f32(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);
bool launch_wm_uw(int lps, const MpdataWmArgs& a, void* stream, bool conv);
}
namespace mpdata_fast {
bool launch_wm_uw(int lps, const MpdataWmArgs& a, void* stream, bool conv);
bool launch_wm(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);
bool launch_wm_f32(int lps, int wpb, const MpdataWmArgs& a, void* stream, int flags);
}

namespace {
int g_layout = -1;  // -1: read MPDATA_PLAN_LAYOUT on first use
int plan_layout_default() {
  if (g_layout < 0) {
    const char* v = getenv("MPDATA_PLAN_LAYOUT");
    g_layout = (v && (!strcmp(v, "reference") || !strcmp(v, "0"))) ? MPDATA_LAYOUT_REFERENCE : MPDATA_LAYOUT_WAVEMAJOR;
  }
  return g_layout;
}
// Serpentine tile order (every other run of a plan walks the tiles from the other end, so that it
// starts on what the previous run left in the Infinity Cache): OFF by default -- it only pays when
// consecutive runs of a plan share u, w (consecutive tracers of one CRM step), and a timing that
// inherits cache state from the previous call is not the timing of a call.  MPDATA_SERPENTINE=1 or
// mpdata_set_serpentine(1) turns it on.
int g_serpentine = -1;
int serpentine() {
  if (g_serpentine < 0) {
    const char* e = getenv("MPDATA_SERPENTINE");
    g_serpentine = (e && atoi(e) != 0) ? 1 : 0;
  }
  return g_serpentine;
}
// test switches of the wave-major launch (MPDATA_WMF_*): from the environment once, or set
int g_wm_flags = -1;
int wm_flags() {
  if (g_wm_flags < 0)
    g_wm_flags = (getenv("MPDATA_WM_NOSTREAM") ? MPDATA_WMF_NOSTREAM : 0) | (getenv("MPDATA_WM_TPW1") ? MPDATA_WMF_TPW1 : 0) |
                 (getenv("MPDATA_WM_NOSPLIT") ? MPDATA_WMF_NOSPLIT : 0) | (getenv("MPDATA_WM_SPLIT") ? MPDATA_WMF_SPLIT : 0);
  return g_wm_flags;
}
int wm_wpb() { return 4; }
  // waves (tiles) per workgroup of the wave-major kernels
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
}  // namespace

struct mpdata_plan {
  int64_t ncrms;
  int nx, nz, ntracers;
  int eb;        // bytes per real: 8 (fp64 plan) or 4 (fp32 plan)
  int device;    // the plan's device
  int variant;   // MPDATA_VARIANT_* at creation
  int layout;    // MPDATA_LAYOUT_*
  Sizes sz;      // element counts of the reference-layout arrays
  // reference-layout plans
  Arena arena;
  void *f, *u, *w, *rho, *rhow, *adz, *flux;  // = arena.p[0..6]
  // wave-major plans
  int lps, slp, wpb, ntiles;
  int64_t wm_ncrms;  // instances as the wave-major side sees them: ncrms (fp64) or ncrms / 2 pairs (fp32)
  long long chunk, tile_elems, main_e;   // main_e: elements of the line-aligned part of a column chunk
  void *pf, *pu, *pw, *pkc, *pflux;  // private arrays
  void* stage;                       // reference-layout staging: one tracer of f (or u, w)
  size_t stage_elems;
  void* flux_ref;                    // flux in the reference layout (level nz is carried through)
  void* wpark;                       // EXACT: park array of the limited vertical fluxes (bit-identical flux); with park_regs
  size_t wpark_bytes;                // only mpdata_plan_run_uw needs it: allocated by its first call
  bool park_regs;                    // EXACT, nx <= MPDATA_WM_NPK: mpdata_plan_run parks in registers (no park array)
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  bool uploaded, ran;
  bool have_u, have_w;   // the plan holds velocities (imported since the last mpdata_plan_run_uw)
  bool timing;     // record the event pair around every run (mpdata_plan_last_kernel_ms); mpdata_plan_set_timing
  unsigned runs;   // launches so far (serpentine tile order)
  mpdata_multi* multi;  // != null: a multi-GPU plan (mpdata_multi.hip); nothing else above is used
};

namespace {

// conversion jobs of a wave-major plan: which = 0 f, 1 u, 2 w, 3 rho, 4 rhow, 5 adz, 6 flux
MpdataLayoutJob wm_job(const mpdata_plan* p, int which, void* ref, int first_tracer, int ntr) {
  MpdataLayoutJob j;
  const int nzm = p->nz - 1, nx = p->nx;
  // (fp32 plans: every array seen as wm_ncrms = ncrms / 2 pairs of adjacent instances, 8 bytes each)
  j.ref = ref; j.ncrms = p->wm_ncrms; j.nlev = nzm; j.ntr = 1; j.slp = p->slp; j.ntiles = p->ntiles;
  j.chunk = p->chunk; j.ref_tstride = 0; j.prv_tstride = 0; j.prv_col0 = 0;
  j.main_e = which <= 2 ? p->main_e : 0;   // f, u, w are split into line-aligned part + rest
  j.ncol_p = nx + 6;
  j.ref_colmul = 1;
  switch (which) {
    case 0:
      j.prv = (double*)p->pf + (long long)first_tracer * p->ntiles * p->tile_elems;
      j.ncols = nx + 6; j.ref_levmul = nx + 6; j.prv_tile_stride = p->tile_elems;
      j.ntr = ntr; j.ref_tstride = (long long)p->wm_ncrms * (nx + 6) * nzm; j.prv_tstride = (long long)p->ntiles * p->tile_elems;
      break;
    case 1: j.prv = p->pu; j.ncols = nx + 5; j.ref_levmul = nx + 5; j.prv_col0 = 1; j.prv_tile_stride = p->tile_elems; break;
    case 2: j.prv = p->pw; j.ncols = nx + 4; j.ref_levmul = nx + 4; j.prv_col0 = 1; j.prv_tile_stride = p->tile_elems; break;
    case 3: case 4: case 5:   // kc = [tile][rho, adz, rhow][chunk]
      j.prv = p->pkc; j.ncols = 1; j.ref_colmul = 0; j.ref_levmul = 1; j.prv_tile_stride = 3 * p->chunk;
      j.prv_col0 = which == 3 ? 0 : (which == 5 ? 1 : 2);
      break;
    default:
      j.prv = (double*)p->pflux + (long long)first_tracer * p->ntiles * p->chunk;
      j.ncols = 1; j.ref_colmul = 0; j.ref_levmul = 1; j.prv_tile_stride = p->chunk;
      j.ntr = ntr; j.ref_tstride = (long long)p->wm_ncrms * p->nz; j.prv_tstride = (long long)p->ntiles * p->chunk;
      break;
  }
  return j;
}

// lanes per instance of the wave-major kernels; 128 = an instance wider than a wave (65 <= nz <= 127: several waves
// per instance, mpdata_kernel_wm_body.h "KS"; the layout kernels hold a column of one instance group in LDS: nzm <= 126)
#define MPDATA_WM_NZ_MAX 127
int wm_lps_for(int nz) { return nz <= 8 ? 8 : nz <= 16 ? 16 : nz <= 32 ? 32 : nz <= 64 ? 64 : nz <= MPDATA_WM_NZ_MAX ? 128 : 0; }
int wm_nkw_for(int nz) { return nz <= 64 ? 1 : 1 + (nz - 64 + 57) / 58; }

int plan_check(const mpdata_plan* p, int eb) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (p->eb != eb) return set_err(MPDATA_ESTATE, "plan precision (%d-byte reals) does not match the call", p->eb);
  return 0;
}
int tracer_range(const mpdata_plan* p, int first, int count) {
  if (first < 0 || count < 1 || first + count > p->ntracers)
    return set_err(MPDATA_EINVAL, "tracer range [%d, %d) outside the plan's %d tracers", first, first + count, p->ntracers);
  return 0;
}

// The reference-layout staging buffer of a wave-major plan (one tracer of f, or u / w): only host
// transfers need it, so it is allocated by the first of them (a plan that is only ever fed from
// device arrays -- bench.py keeps one per field set -- never pays its 538 MB).
int plan_stage(mpdata_plan* p) {
  if (p->stage) return 0;
  HIP_TRY(hipMalloc(&p->stage, p->stage_elems * p->eb));
  return 0;
}

// MPDATA_LAYOUT_LEGACY=1: the round-2 conversion kernel (one workgroup per column) for f, u, w as well (A/B)
bool legacy_convert() {
  static const bool v = getenv("MPDATA_LAYOUT_LEGACY") != nullptr;
  return v;
}

// Arrays in the reference layout -> the plan.  `dev` says where the pointers live.  Null
// pointers are skipped (the plan keeps what it has).  f / flux cover `count` tracers.
int plan_import(mpdata_plan* p, const void* f, const void* u, const void* w, const void* rho,
                const void* rhow, const void* adz, const void* flux, int first, int count, bool dev) {
  const int eb = p->eb;
  const size_t f1 = p->sz.f / p->ntracers;  // elements of one tracer of f
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (p->layout == MPDATA_LAYOUT_REFERENCE) {
    if (f) HIP_TRY(hipMemcpyAsync((char*)p->f + first * f1 * eb, f, f1 * count * eb, kind, p->stream));
    if (u) { HIP_TRY(hipMemcpyAsync(p->u, u, p->sz.u * eb, kind, p->stream)); p->have_u = true; }
    if (w) { HIP_TRY(hipMemcpyAsync(p->w, w, p->sz.w * eb, kind, p->stream)); p->have_w = true; }
    if (rho) HIP_TRY(hipMemcpyAsync(p->rho, rho, p->sz.k * eb, kind, p->stream));
    if (rhow) HIP_TRY(hipMemcpyAsync(p->rhow, rhow, p->sz.kz * eb, kind, p->stream));
    if (adz) HIP_TRY(hipMemcpyAsync(p->adz, adz, p->sz.k * eb, kind, p->stream));
    if (flux) HIP_TRY(hipMemcpyAsync((char*)p->flux + first * p->sz.kz * eb, flux, p->sz.kz * count * eb, kind, p->stream));
    return 0;
  }
  // wave-major: device sources are converted in place, host sources go through the staging
  // buffer one array (one tracer of f) at a time
  if (!dev) {
    const int rs = plan_stage(p);
    if (rs) return rs;
  }
  // f, u, w (many columns, split chunks): the column-walking kernel; the small arrays: the per-column one
  auto conv = [&](int which, void* ref, int tr, int ntr) -> int {
    const MpdataLayoutJob j = wm_job(p, which, ref, tr, ntr);
    if (which <= 2 && !legacy_convert()) {
      const hipError_t e = mpdata_layout_import_rows(&j, 1, p->stream);   // (row segments through LDS-DMA where possible)
      if (e == hipErrorNotSupported) HIP_TRY(mpdata_layout_convert_cols(&j, 1, true, p->stream));
      else HIP_TRY(e);
    } else {
      HIP_TRY(mpdata_layout_convert(j, 8, true, p->stream));
    }
    return 0;
  };
  auto one = [&](int which, const void* src, size_t elems, int tr) -> int {
    void* ref = const_cast<void*>(src);
    if (!dev) {
      HIP_TRY(hipMemcpyAsync(p->stage, src, elems * eb, hipMemcpyHostToDevice, p->stream));
      ref = p->stage;
    }
    return conv(which, ref, tr, 1);
  };
  int rc = 0;
  if (f) {
    if (dev) {
      rc = conv(0, const_cast<void*>(f), first, count);
    } else {
      for (int t = 0; t < count && !rc; ++t) rc = one(0, (const char*)f + (size_t)t * f1 * eb, f1, first + t);
    }
  }
  if (!rc && u && w && dev && !legacy_convert()) {   // u and w of a device import: ONE launch
    const MpdataLayoutJob j2[2] = {wm_job(p, 1, const_cast<void*>(u), 0, 1), wm_job(p, 2, const_cast<void*>(w), 0, 1)};
    const hipError_t e = mpdata_layout_import_rows(j2, 2, p->stream);
    if (e == hipErrorNotSupported) HIP_TRY(mpdata_layout_convert_cols(j2, 2, true, p->stream));
    else HIP_TRY(e);
    p->have_u = p->have_w = true;
  } else {
    if (!rc && u) { rc = one(1, u, p->sz.u, 0); if (!rc) p->have_u = true; }
    if (!rc && w) { rc = one(2, w, p->sz.w, 0); if (!rc) p->have_w = true; }
  }
  if (!rc && rho) rc = one(3, rho, p->sz.k, 0);
  if (!rc && rhow) rc = one(4, rhow, p->sz.kz, 0);
  if (!rc && adz) rc = one(5, adz, p->sz.k, 0);
  if (!rc && flux) {
    // kept twice: in the reference layout (level nz is carried through to the export) and in the
    // private array (a tracer that is never run exports what was imported)
    void* fr = (char*)p->flux_ref + (size_t)first * p->sz.kz * eb;
    HIP_TRY(hipMemcpyAsync(fr, flux, p->sz.kz * count * eb, kind, p->stream));
    HIP_TRY(mpdata_layout_convert(wm_job(p, 6, fr, first, count), 8, true, p->stream));
  }
  return rc;
}

int plan_export(mpdata_plan* p, void* f, void* flux, int first, int count, bool dev) {
  const int eb = p->eb;
  const size_t f1 = p->sz.f / p->ntracers;
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (p->layout == MPDATA_LAYOUT_REFERENCE) {
    if (f) HIP_TRY(hipMemcpyAsync(f, (char*)p->f + first * f1 * eb, f1 * count * eb, kind, p->stream));
    if (flux) HIP_TRY(hipMemcpyAsync(flux, (char*)p->flux + first * p->sz.kz * eb, p->sz.kz * count * eb, kind, p->stream));
    return 0;
  }
  auto conv_out = [&](void* ref, int tr, int ntr) -> int {
    const MpdataLayoutJob j = wm_job(p, 0, ref, tr, ntr);
    if (!legacy_convert()) HIP_TRY(mpdata_layout_convert_cols(&j, 1, false, p->stream));
    else HIP_TRY(mpdata_layout_convert(j, 8, false, p->stream));
    return 0;
  };
  if (f) {
    if (dev) {
      const int rc = conv_out(f, first, count);
      if (rc) return rc;
    } else {
      const int rs = plan_stage(p);
      if (rs) return rs;
      for (int t = 0; t < count; ++t) {
        const int rc = conv_out(p->stage, first + t, 1);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync((char*)f + (size_t)t * f1 * eb, p->stage, f1 * eb, hipMemcpyDeviceToHost, p->stream));
      }
    }
  }
  if (flux) {
    // levels 1..nzm from the kernel's result; level nz is whatever was uploaded (the reference
    // never writes it, :541, :624)
    void* fr = (char*)p->flux_ref + (size_t)first * p->sz.kz * eb;
    HIP_TRY(mpdata_layout_convert(wm_job(p, 6, fr, first, count), 8, false, p->stream));
    HIP_TRY(hipMemcpyAsync(flux, fr, p->sz.kz * count * eb, kind, p->stream));
  }
  return 0;
}

}  // namespace

extern "C" {

static int plan_create(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan, int eb) {
  if (!plan) return set_err(MPDATA_EINVAL, "null plan pointer");
  *plan = nullptr;
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  const int var = variant();
  // wave-major: fp64, and fp32 with an even ncrms (two adjacent instances per lane = 8-byte elements)
  // (nz > 64: fp64; EXACT with the flux in the reference's order needs the register park there: nx <= MPDATA_WM_NPK)
  const bool ks_ok = nz <= 64 || (eb == 8 && (var != MPDATA_VARIANT_EXACT || !exact_flux_in_order() ||
                                              (exact_flux_in_regs() && nx <= MPDATA_WM_NPK)));
  const bool wmaj = (eb == 8 || (ncrms & 1) == 0) && wm_lps_for(nz) != 0 && ks_ok &&
                    plan_layout_default() == MPDATA_LAYOUT_WAVEMAJOR && tile_override() < 0;
  MpdataTileInfo t;
  if (!wmaj) {
    rc = choose_tile(var, ncrms, nx, nz, &t, eb);
    if (rc) return rc;
  }
  mpdata_plan* p = (mpdata_plan*)calloc(1, sizeof(mpdata_plan));
  if (!p) return set_err(MPDATA_EINVAL, "out of host memory");
  p->ncrms = ncrms; p->nx = nx; p->nz = nz; p->ntracers = ntracers; p->eb = eb;
  p->variant = var;
  p->layout = wmaj ? MPDATA_LAYOUT_WAVEMAJOR : MPDATA_LAYOUT_REFERENCE;
  p->sz = sizes_of(ncrms, nx, nz, ntracers);
  hipError_t e = hipGetDevice(&p->device);
  if (e == hipSuccess && !wmaj) {
    const size_t nb[7] = {p->sz.f * eb, p->sz.u * eb, p->sz.w * eb, p->sz.k * eb, p->sz.kz * eb, p->sz.k * eb,
                          p->sz.kz * ntracers * eb};
    e = arena_alloc(p->arena, nb);
    if (e == hipSuccess) e = hipMemset(p->arena.p[6], 0, nb[6]);
    if (e == hipSuccess) {
      p->f = p->arena.p[0]; p->u = p->arena.p[1]; p->w = p->arena.p[2]; p->rho = p->arena.p[3];
      p->rhow = p->arena.p[4]; p->adz = p->arena.p[5]; p->flux = p->arena.p[6];
    }
  }
  if (e == hipSuccess && wmaj) {
    const int nzm = nz - 1;
    const int web = 8;   // bytes of an element on the wave-major side (fp32: a pair of instances)
    p->wm_ncrms = eb == 8 ? ncrms : ncrms / 2;
    p->lps = wm_lps_for(nz); p->slp = p->lps >= 64 ? 1 : 64 / p->lps; p->wpb = wm_wpb();
    p->ntiles = (int)((p->wm_ncrms + p->slp - 1) / p->slp);
    p->chunk = (long long)p->slp * nzm;
    p->main_e = p->chunk * web / 128 * (128 / web);
    // The counted waits of the wave-major kernels (s_waitcnt vmcnt(N), mpdata_kernel_wm_body.h) count
    // INSTRUCTIONS: every column-pair fetch of an array must issue exactly two -- the lanes of the
    // line-aligned main part in one, the other lanes in the second --, so both lane sets must be
    // non-empty: 128 <= main part <= 384 bytes.  True for every (LPS, nz) the kernels are built for
    // (chunk = (64/LPS) * nzm * 8 bytes with LPS/2 <= nzm < LPS, nz >= 3: 128 .. 504 bytes); checked
    // here so that a future tiling cannot break the wait silently.
    if (p->lps <= 64 && (p->main_e * web < 128 || p->main_e * web > 384)) {   // (nz > 64: one fetch instruction per array and pair)
      free(p);
      return set_err(MPDATA_EUNSUPPORTED, "internal: column chunk of %lld bytes breaks the two-instructions-per-fetch "
                                          "invariant of the wave-major kernels", (long long)(p->chunk * web));
    }
    {  // tiles start on 128-byte lines, an ODD number of lines apart (a power-of-two-ish stride
       // would put the same column of every tile on the same HBM channels: measured -5 %)
      long long lines = ((long long)(nx + 6) * p->chunk * web + 127) / 128;
      if ((lines & 1) == 0) ++lines;
      p->tile_elems = lines * (128 / web);
    }
    const size_t tile_arr = (size_t)p->ntiles * p->tile_elems * web;
    const size_t f1 = p->sz.f / ntracers;
    p->stage_elems = f1 > p->sz.w ? f1 : p->sz.w;
    if (e == hipSuccess) e = hipMalloc(&p->pf, tile_arr * ntracers);
    if (e == hipSuccess) e = hipMalloc(&p->pu, tile_arr);
    if (e == hipSuccess) e = hipMalloc(&p->pw, tile_arr);
    if (e == hipSuccess) e = hipMalloc(&p->pkc, (size_t)p->ntiles * 3 * p->chunk * web);
    if (e == hipSuccess) e = hipMalloc(&p->pflux, (size_t)p->ntiles * p->chunk * ntracers * web);
    if (e == hipSuccess) e = hipMalloc(&p->flux_ref, p->sz.kz * ntracers * eb);
    // EXACT: the park array of the limited vertical fluxes (bit-identical flux, see plan_flux_finish): [tracer][tile][nx][64]
    // 8-byte elements, the size of f's interior.  MPDATA_EXACT_FLUX=sum does without it (flux = upwind sum + limited sum,
    // <= 1e-13 relative, the behaviour up to round 3; a quarter faster in the EXACT variant).
    p->park_regs = var == MPDATA_VARIANT_EXACT && exact_flux_in_regs() && nx <= MPDATA_WM_NPK;
    if (e == hipSuccess && var == MPDATA_VARIANT_EXACT && exact_flux_in_order() && !p->park_regs) {
      p->wpark_bytes = (size_t)ntracers * p->ntiles * (size_t)nx * 64 * 8;
      e = hipMalloc(&p->wpark, p->wpark_bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        mpdata_plan_destroy(p);
        return set_err((int)e, "mpdata_plan_create (EXACT): no memory for the %.1f-GB park array of the bit-identical flux; "
                               "MPDATA_EXACT_FLUX=sum does without it (flux then equal to 1e-13 relative)", p->wpark_bytes / 1e9);
      }
    }
    // flux: level nz and tracers that are never run export what was imported -- or zeros
    if (e == hipSuccess) e = hipMemset(p->flux_ref, 0, p->sz.kz * ntracers * eb);
    if (e == hipSuccess) e = hipMemset(p->pflux, 0, (size_t)p->ntiles * p->chunk * ntracers * web);
    // u, w have column slots that nothing ever fills or fetches (c = 0; c = nx+5 of w)
    if (e == hipSuccess) e = hipMemset(p->pu, 0, tile_arr);
    if (e == hipSuccess) e = hipMemset(p->pw, 0, tile_arr);
  }
  if (e == hipSuccess) e = hipStreamCreate(&p->stream);
  if (e == hipSuccess) p->own_stream = true;
  if (e == hipSuccess) e = hipEventCreate(&p->ev0);
  if (e == hipSuccess) e = hipEventCreate(&p->ev1);
  p->timing = true;
  if (e != hipSuccess) {
    mpdata_plan_destroy(p);
    return hip_err(e, "mpdata_plan_create");
  }
  *plan = p;
  return 0;
}
int mpdata_plan_create(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan) {
  return plan_create(ncrms, nx, nz, ntracers, plan, 8);
}
int mpdata_plan_create_f32(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan) {
  return plan_create(ncrms, nx, nz, ntracers, plan, 4);
}

// flux of every tracer := 0 on the plan's stream (an upload without a flux array: level nz and tracers
// that are never run then download as zeros); also used by the multi-GPU upload on its per-GPU plans
extern "C++" int mpdata_plan_zero_flux_internal(mpdata_plan* p) {
  DevGuard g(p->device);
  void* fl = p->layout == MPDATA_LAYOUT_REFERENCE ? p->flux : p->flux_ref;
  HIP_TRY(hipMemsetAsync(fl, 0, p->sz.kz * p->ntracers * p->eb, p->stream));
  if (p->layout == MPDATA_LAYOUT_WAVEMAJOR)
    HIP_TRY(hipMemsetAsync(p->pflux, 0, (size_t)p->ntiles * p->chunk * p->ntracers * 8, p->stream));
  return 0;
}

static int plan_upload(mpdata_plan* p, const void* f, const void* u, const void* w, const void* rho,
                       const void* rhow, const void* adz, const void* flux, int eb) {
  int rc = plan_check(p, eb);
  if (rc) return rc;
  if (!f || !u || !w || !rho || !rhow || !adz) return set_err(MPDATA_EINVAL, "null array pointer");
  if (p->multi) {
    rc = mpdata_multi_upload(p->multi, f, u, w, rho, rhow, adz, flux);
    if (!rc) p->uploaded = true;
    return rc;
  }
  DevGuard g(p->device);
  // flux is intent(out) in the reference but its level nz is never written
  // (reference :541, :624 touch 1..nzm only): carry the caller's values over
  if (!flux) {
    rc = mpdata_plan_zero_flux_internal(p);
    if (rc) return rc;
  }
  rc = plan_import(p, f, u, w, rho, rhow, adz, flux, 0, p->ntracers, false);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->uploaded = true;
  return 0;
}
int mpdata_plan_upload(mpdata_plan* p, const double* f, const double* u, const double* w,
                       const double* rho, const double* rhow, const double* adz,
                       const double* flux) {
  return plan_upload(p, f, u, w, rho, rhow, adz, flux, 8);
}
int mpdata_plan_upload_f32(mpdata_plan* p, const float* f, const float* u, const float* w,
                           const float* rho, const float* rhow, const float* adz,
                           const float* flux) {
  return plan_upload(p, f, u, w, rho, rhow, adz, flux, 4);
}

int mpdata_plan_import_device(mpdata_plan* p, const void* f, const void* u, const void* w, const void* rho,
                              const void* rhow, const void* adz, const void* flux, int first_tracer,
                              int ntracers) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  int rc = tracer_range(p, first_tracer, ntracers);
  if (rc) return rc;
  if (p->multi) {   // the arrays live on the ROOT GPU (shard 0's device), full width: scatter them (RCCL over xGMI)
    rc = mpdata_multi_scatter_device(p->multi, f, u, w, rho, rhow, adz, flux, first_tracer, ntracers);
    if (!rc) {
      p->uploaded = true;
      for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g) mpdata_multi_sub(p->multi, g)->uploaded = true;
    }
    return rc;
  }
  DevGuard g(p->device);
  rc = plan_import(p, f, u, w, rho, rhow, adz, flux, first_tracer, ntracers, true);
  if (rc) return rc;
  p->uploaded = true;  // (the caller is responsible for having provided every array once)
  return 0;
}

int mpdata_plan_export_device(mpdata_plan* p, void* f, void* flux, int first_tracer, int ntracers) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  int rc = tracer_range(p, first_tracer, ntracers);
  if (rc) return rc;
  if (p->multi) {   // gather to arrays on the root GPU
    for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g)
      if (!mpdata_multi_sub(p->multi, g)->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_export_device before upload / import (shard %d)", g);
    return mpdata_multi_gather_device(p->multi, f, flux, first_tracer, ntracers);
  }
  if (!p->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_export_device before upload / import");
  DevGuard g(p->device);
  return plan_export(p, f, flux, first_tracer, ntracers, true);
}

// (EXACT wave-major runs: the finishing kernel of the bit-identical flux, behind the plan kernels on the same stream)
static int plan_flux_finish(mpdata_plan* p, const MpdataWmArgs& a, int count) {
  if (!a.wpark) return 0;
  const long long waves = (long long)count * p->ntiles;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  if (p->eb == 8)
    hipLaunchKernelGGL(flux_finish_kernel<double>, dim3(blocks), dim3(256), 0, p->stream, a.flux, (const double*)a.wpark, p->ntiles,
                       p->nx, p->nz - 1, p->lps, a.flux_tstride, count);
  else
    hipLaunchKernelGGL(flux_finish_kernel<float2>, dim3(blocks), dim3(256), 0, p->stream, (float2*)a.flux, (const float2*)a.wpark,
                       p->ntiles, p->nx, p->nz - 1, p->lps, a.flux_tstride, count);
  HIP_TRY(hipGetLastError());
  return 0;
}

// (uw_conv: that kernel also writes the velocities into the plan's own u, w)
static int plan_launch(mpdata_plan* p, int first, int count, const void* u_ref = nullptr, const void* w_ref = nullptr,
                       bool uw_conv = false) {
  int rc = 0;
  if (p->layout == MPDATA_LAYOUT_WAVEMAJOR) {
    MpdataWmArgs a;
    a.f = (double*)p->pf + (long long)first * p->ntiles * p->tile_elems;
    a.u = (const double*)p->pu; a.w = (const double*)p->pw; a.kc = (const double*)p->pkc;
    a.flux = (double*)p->pflux + (long long)first * p->ntiles * p->chunk;
    a.ntiles = p->ntiles; a.nx = p->nx; a.nz = p->nz; a.ntracers = count;
    a.tile_elems = p->tile_elems;
    a.f_tstride = (long long)p->ntiles * p->tile_elems;
    a.flux_tstride = (long long)p->ntiles * p->chunk;
    a.reverse = serpentine() ? (int)(p->runs++ & 1u) : 0;
    a.u_ref = (const double*)u_ref; a.w_ref = (const double*)w_ref; a.ncrms = p->ncrms;
    a.dbg = g_dbg;   // (null unless a diagnostic build was handed a stamp buffer)
    // EXACT plans: the park array of the limited vertical fluxes (bit-identical flux; allocated with the plan)
    a.park_regs = (p->park_regs && !u_ref) ? 1 : 0;
    a.nkw = wm_nkw_for(p->nz);
    if (u_ref && p->park_regs && !p->wpark) {
      // the kernel that reads u, w from the reference layout has no register-park form (its EXACT build takes every
      // register it can get): the park array after all, allocated by the first such call
      p->wpark_bytes = (size_t)p->ntracers * p->ntiles * (size_t)p->nx * 64 * 8;
      const hipError_t e = hipMalloc(&p->wpark, p->wpark_bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        p->wpark = nullptr;
        return set_err((int)e, "mpdata_plan_run_uw (EXACT): no memory for the %.1f-GB park array of the bit-identical flux; "
                               "MPDATA_EXACT_FLUX=sum does without it", p->wpark_bytes / 1e9);
      }
    }
    a.wpark = (p->wpark && !a.park_regs) ? (double*)p->wpark + (long long)first * p->ntiles * ((long long)p->nx * 64) : nullptr;
    const bool fast = p->variant == MPDATA_VARIANT_FAST;
    if (u_ref) {
      a.reverse = 0;
      const bool okx = fast ? mpdata_fast::launch_wm_uw(p->lps, a, (void*)p->stream, uw_conv)
                            : mpdata_exact::launch_wm_uw(p->lps, a, (void*)p->stream, uw_conv);
      if (!okx) return set_err(MPDATA_EINVAL, "wave-major u,w-reference kernel LPS=%d not instantiated", p->lps);
      HIP_TRY(hipGetLastError());
      return plan_flux_finish(p, a, count);
    }
    const int fl = wm_flags();
    const bool ok = p->eb == 8 ? (fast ? mpdata_fast::launch_wm(p->lps, p->wpb, a, (void*)p->stream, fl)
                                       : mpdata_exact::launch_wm(p->lps, p->wpb, a, (void*)p->stream, fl))
                               : (fast ? mpdata_fast::launch_wm_f32(p->lps, p->wpb, a, (void*)p->stream, fl)
                                       : mpdata_exact::launch_wm_f32(p->lps, p->wpb, a, (void*)p->stream, fl));
    if (!ok) return set_err(MPDATA_EINVAL, "wave-major kernel LPS=%d WPB=%d not instantiated", p->lps, p->wpb);
    HIP_TRY(hipGetLastError());
    rc = plan_flux_finish(p, a, count);
    if (rc) return rc;
  } else {
    const size_t f1 = p->sz.f / p->ntracers;
    // (the plan's own variant, passed explicitly: the global one may be changed by other threads)
    if (p->eb == 8)
      rc = advect_device<double>(p->ncrms, p->nx, p->nz, count, (double*)p->f + first * f1, (const double*)p->u,
                                 (const double*)p->w, (const double*)p->rho, (const double*)p->rhow,
                                 (const double*)p->adz, (double*)p->flux + first * p->sz.kz, (void*)p->stream, p->variant);
    else
      rc = advect_device<float>(p->ncrms, p->nx, p->nz, count, (float*)p->f + first * f1, (const float*)p->u,
                                (const float*)p->w, (const float*)p->rho, (const float*)p->rhow,
                                (const float*)p->adz, (float*)p->flux + first * p->sz.kz, (void*)p->stream, p->variant);
    if (rc) return rc;
  }
  return 0;
}
int mpdata_plan_run_tracers(mpdata_plan* p, int first, int count) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  int rc = tracer_range(p, first, count);
  if (rc) return rc;
  if (p->multi) {   // (the per-GPU plans check their own state: they may have been filled directly)
    rc = mpdata_multi_run(p->multi, first, count);
    if (!rc) p->ran = true;
    return rc;
  }
  if (!p->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_run before mpdata_plan_upload");
  if (!p->have_u || !p->have_w)
    return set_err(MPDATA_ESTATE, "mpdata_plan_run: the plan holds no velocities (none imported yet, or mpdata_plan_run_uw "
                                  "ran since -- it leaves none behind): import u and w first");
  DevGuard g(p->device);
  if (p->timing) HIP_TRY(hipEventRecord(p->ev0, p->stream));
  rc = plan_launch(p, first, count);
  if (rc) return rc;
  if (p->timing) HIP_TRY(hipEventRecord(p->ev1, p->stream));
  p->ran = p->timing;
  return 0;
}
int mpdata_plan_run(mpdata_plan* p) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  return mpdata_plan_run_tracers(p, 0, p->ntracers);
}

// One step on FRESH velocities: u, w are reference-layout device arrays (what a CRM whose state
// lives on the device hands over every step, reference :107 `update device` then kernels), f stays
// in the plan.  The layout entry of u, w is part of the call (and of its event time).
int mpdata_plan_run_uw(mpdata_plan* p, int first, int count, const void* u, const void* w) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (!u || !w) return set_err(MPDATA_EINVAL, "null array pointer");
  int rc = tracer_range(p, first, count);
  if (rc) return rc;
  if (p->multi) {   // u, w: full-width arrays on the root GPU -- scatter them (RCCL over xGMI), then every GPU runs
    for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g)
      if (!mpdata_multi_sub(p->multi, g)->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_run_uw before upload / import (shard %d)", g);
    rc = mpdata_multi_scatter_device(p->multi, nullptr, u, w, nullptr, nullptr, nullptr, nullptr, 0, 0);
    if (!rc) rc = mpdata_multi_run(p->multi, first, count);
    // the same post-condition as on one GPU: no velocities are left behind
    for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g) {
      mpdata_plan* q = mpdata_multi_sub(p->multi, g);
      q->have_u = q->have_w = false;
    }
    if (!rc) p->ran = true;
    return rc;
  }
  if (!p->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_run_uw before upload / import");
  DevGuard g(p->device);
  if (p->timing) HIP_TRY(hipEventRecord(p->ev0, p->stream));
  // one fp64 tracer of a wave-major plan: the kernel fetches u, w from the caller's arrays itself
  // (16-byte row pieces: even ncrms, 16-byte aligned bases; 32-bit offsets: arrays below 4 GiB);
  // MPDATA_RUN_UW=import forces the conversion path (tests, A/B)
  static const bool force_import = getenv("MPDATA_RUN_UW") && !strcmp(getenv("MPDATA_RUN_UW"), "import");
  const bool ring = p->layout == MPDATA_LAYOUT_WAVEMAJOR && p->eb == 8 && (p->ncrms & 1) == 0 && p->lps <= 64 &&
                    (((uintptr_t)u | (uintptr_t)w) & 15) == 0 &&
                    (double)p->ncrms * (p->nx + 5) * p->nz * 8.0 < 4294967000.0 && !force_import;
  // POST-CONDITION, the same on every path (which one runs depends on alignment, parity of ncrms, nz,
  // the tracer count ...): the plan holds NO velocities afterwards.  The one-tracer kernel never writes the
  // plan's u, w (they would be stale), the other paths overwrite them (they would be the new ones):
  // neither is promised, mpdata_plan_run returns MPDATA_ESTATE until u, w are imported again.
  p->have_u = p->have_w = false;
  if (ring && count == 1) {
    rc = plan_launch(p, first, count, u, w);
  } else if (ring && p->lps <= 32) {
    // a tracer batch: its FIRST tracer goes through the kernel that reads the caller's u, w -- in the form
    // that also writes them into the plan's arrays as it goes --, the others through the batch kernel behind
    // it: no conversion pass (0.54 ms at ncrms = 65536) in front of the batch
    rc = plan_launch(p, first, 1, u, w, true);
    if (!rc) rc = plan_launch(p, first + 1, count - 1);
  } else {
    rc = plan_import(p, nullptr, u, w, nullptr, nullptr, nullptr, nullptr, 0, 1, true);
    if (!rc) rc = plan_launch(p, first, count);
    p->have_u = p->have_w = false;
  }
  if (rc) return rc;
  if (p->timing) HIP_TRY(hipEventRecord(p->ev1, p->stream));
  p->ran = p->timing;
  return 0;
}

int mpdata_plan_sync(mpdata_plan* p) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (p->multi) return mpdata_multi_sync(p->multi);
  DevGuard g(p->device);
  HIP_TRY(hipStreamSynchronize(p->stream));
  return 0;
}

static int plan_download(mpdata_plan* p, void* f, void* flux, int eb) {
  int rc = plan_check(p, eb);
  if (rc) return rc;
  if (p->multi) {   // (the shards may have been filled directly: mpdata_plan_shard_plan + import)
    for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g)
      if (!mpdata_multi_sub(p->multi, g)->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_download before upload (shard %d)", g);
    return mpdata_multi_download(p->multi, f, flux);
  }
  if (!p->uploaded) return set_err(MPDATA_ESTATE, "mpdata_plan_download before upload");
  DevGuard g(p->device);
  rc = plan_export(p, f, flux, 0, p->ntracers, false);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(p->stream));
  return 0;
}
int mpdata_plan_download(mpdata_plan* p, double* f, double* flux) { return plan_download(p, f, flux, 8); }
int mpdata_plan_download_f32(mpdata_plan* p, float* f, float* flux) { return plan_download(p, f, flux, 4); }

int mpdata_plan_last_kernel_ms(mpdata_plan* p, double* ms) {
  if (!p || !ms) return set_err(MPDATA_EINVAL, "null argument");
  if (!p->ran) return set_err(MPDATA_ESTATE, "no run recorded");
  if (p->multi) return mpdata_multi_last_kernel_ms(p->multi, ms);
  DevGuard g(p->device);
  HIP_TRY(hipEventSynchronize(p->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, p->ev0, p->ev1));
  *ms = t;
  return 0;
}

int mpdata_plan_set_stream(mpdata_plan* p, void* stream) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (p->multi) return set_err(MPDATA_EUNSUPPORTED, "a multi-GPU plan runs on its own streams, one per device");
  DevGuard g(p->device);
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (p->own_stream) (void)hipStreamDestroy(p->stream);
  p->stream = (hipStream_t)stream;
  p->own_stream = false;
  return 0;
}
// The event pair a plan records around every run costs two marker packets between consecutive
// launches of a stream (about 1.5 % of a 0.4-ms kernel): a caller that times a whole loop itself
// switches it off (mpdata_plan_last_kernel_ms then reports MPDATA_ESTATE).
int mpdata_plan_set_timing(mpdata_plan* p, int on) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (p->multi) {
    for (int g = 0; g < mpdata_multi_ngpus(p->multi); ++g) mpdata_multi_sub(p->multi, g)->timing = on != 0;
    return 0;
  }
  p->timing = on != 0;
  return 0;
}
int mpdata_plan_layout(const mpdata_plan* p) { return p ? p->layout : MPDATA_EINVAL; }
int mpdata_plan_device(const mpdata_plan* p) { return p ? p->device : MPDATA_EINVAL; }
int mpdata_set_plan_layout(int layout) {
  const int prev = plan_layout_default();
  if (layout == MPDATA_LAYOUT_REFERENCE || layout == MPDATA_LAYOUT_WAVEMAJOR) g_layout = layout;
  return prev;
}

// ---- multi-GPU plans (mpdata_multi.hip): the same handle type; upload / run / run_tracers /
// sync / download / last_kernel_ms / destroy dispatch to the per-device plans.
static int plan_create_multi(int64_t ncrms, int nx, int nz, int ntracers, int ngpus, const int* devices,
                             mpdata_plan** plan, int eb) {
  if (!plan) return set_err(MPDATA_EINVAL, "null plan pointer");
  *plan = nullptr;
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  mpdata_plan* p = (mpdata_plan*)calloc(1, sizeof(mpdata_plan));
  if (!p) return set_err(MPDATA_EINVAL, "out of host memory");
  p->ncrms = ncrms; p->nx = nx; p->nz = nz; p->ntracers = ntracers; p->eb = eb;
  p->variant = variant(); p->device = -1; p->layout = -1;
  p->sz = sizes_of(ncrms, nx, nz, ntracers);
  rc = mpdata_multi_create(ncrms, nx, nz, ntracers, ngpus, devices, eb, &p->multi);
  if (rc) { free(p); return rc; }
  *plan = p;
  return 0;
}
int mpdata_plan_create_multi(int64_t ncrms, int nx, int nz, int ntracers, int ngpus, mpdata_plan** plan) {
  // MPDATA_MULTI_DEVICES="0,0,1": explicit device list (tests on a one-GPU box repeat a device)
  const char* e = getenv("MPDATA_MULTI_DEVICES");
  if (e && *e) {
    int devs[64], n = 0;
    for (const char* q = e; *q && n < 64;) {
      devs[n++] = atoi(q);
      while (*q && *q != ',') ++q;
      if (*q == ',') ++q;
    }
    if (n >= ngpus) return plan_create_multi(ncrms, nx, nz, ntracers, ngpus, devs, plan, 8);
  }
  return plan_create_multi(ncrms, nx, nz, ntracers, ngpus, nullptr, plan, 8);
}
int mpdata_plan_create_multi_devices(int64_t ncrms, int nx, int nz, int ntracers, int ngpus, const int* devices,
                                     mpdata_plan** plan) {
  return plan_create_multi(ncrms, nx, nz, ntracers, ngpus, devices, plan, 8);
}
int mpdata_plan_ranks_seen(const mpdata_plan* p) {
  if (!p) return MPDATA_EINVAL;
  return p->multi ? mpdata_multi_ranks_seen(p->multi) : 0;
}
int mpdata_plan_ngpus(const mpdata_plan* p) { return !p ? MPDATA_EINVAL : (p->multi ? mpdata_multi_ngpus(p->multi) : 1); }
int mpdata_plan_shard(const mpdata_plan* p, int g, int* device, int64_t* sl0, int64_t* nloc) {
  if (!p) return set_err(MPDATA_EINVAL, "null plan");
  if (p->multi) return mpdata_multi_info(p->multi, g, device, sl0, nloc);
  if (g != 0) return set_err(MPDATA_EINVAL, "shard %d of a single-GPU plan", g);
  if (device) *device = p->device;
  if (sl0) *sl0 = 0;
  if (nloc) *nloc = p->ncrms;
  return 0;
}
mpdata_plan* mpdata_plan_shard_plan(mpdata_plan* p, int g) {
  if (!p) return nullptr;
  if (p->multi) return mpdata_multi_sub(p->multi, g);
  return g == 0 ? p : nullptr;
}
int mpdata_plan_transfer_stats(const mpdata_plan* p, double* scatter_s, double* gather_s, int64_t* scatter_bytes_per_peer,
                               int64_t* gather_bytes_per_peer, int* transport) {
  if (!p || !p->multi) return set_err(MPDATA_EINVAL, "not a multi-GPU plan");
  mpdata_multi_stats(p->multi, scatter_s, gather_s, scatter_bytes_per_peer, gather_bytes_per_peer, transport);
  return 0;
}

int mpdata_plan_destroy(mpdata_plan* p) {
  if (!p) return 0;
  if (p->multi) {
    const int rc = mpdata_multi_destroy(p->multi);
    free(p);
    return rc;
  }
  DevGuard g(p->device);
  arena_free(p->arena);
  void* bufs[8] = {p->pf, p->pu, p->pw, p->pkc, p->pflux, p->stage, p->flux_ref, p->wpark};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (p->ev0) (void)hipEventDestroy(p->ev0);
  if (p->ev1) (void)hipEventDestroy(p->ev1);
  if (p->stream && p->own_stream) (void)hipStreamDestroy(p->stream);
  free(p);
  return 0;
}

// Host-array call = the drop-in for `call advect_scalar2D_openacc_N(f,u,w,rho,rhow,flux)` with
// its `!$acc update device / host` traffic inside (reference :107, :241).  The reference spends
// 72 % + 10 % of its GPU time in exactly these copies (results/advect.pgiacc.17.7-nvprof:18-19).
// The ncrms axis is cut into chunks and pipelined over three streams and three device buffer
// sets: chunk c+1 goes host -> device (2-D copies: a chunk is cw*8 bytes of every ncrms*8-byte
// row) while chunk c is advected (a problem of its own, leading dimension cw) and chunk c-1
// comes back.  A copy from / to PAGEABLE host memory blocks its calling thread while the runtime
// stages it through pinned buffers (at full PCIe rate once the pages are warm, measured 56 GB/s),
// so the device -> host leg is driven by a second host thread: both PCIe directions are then
// busy at once without page-locking anything.  Page-locking the caller's arrays for one call
// does not pay (hipHostRegister: 20 ms per 538 MB, tools/h2d_rate.hip, against 10 ms to copy
// them) and is not done: the library never registers or unregisters caller memory (with this
// runtime a later copy from a recycled address of a once-registered range ended in a GPU memory
// access fault).  Arrays the CALLER has registered are simply used as they are (the copies are
// then true asynchronous DMA).
// MPDATA_HOST_CHUNK=<instances per chunk> (default ncrms/8, at least 1024, a multiple of 64).
namespace {
struct ChunkBufs {
  Arena arena;
  double *f = nullptr, *u = nullptr, *w = nullptr, *rho = nullptr, *rhow = nullptr, *adz = nullptr, *flux = nullptr;
  hipEvent_t run = nullptr;
  bool busy = false;   // handed to the device -> host thread, not yet copied back
};
void free_chunk(ChunkBufs& b) {
  arena_free(b.arena);
  if (b.run) (void)hipEventDestroy(b.run);
  b = ChunkBufs();
}
int64_t host_chunk(int64_t ncrms) {
  const char* v = getenv("MPDATA_HOST_CHUNK");
  int64_t c = v ? atoll(v) : (ncrms + 7) / 8;
  if (!v && c < 1024) c = 1024;
  if (c < 16) c = 16;
  c = (c + 63) / 64 * 64;
  return c < ncrms ? c : ncrms;
}
struct OutJob {
  int set;
  int64_t c0, cw;
};
// What a host-array call needs besides the caller's arrays -- two streams, up to three sets of chunk buffers, an event
// per set -- is kept per HOST THREAD between calls: creating and destroying them costs 6.7 ms per call with this
// runtime (two hipStreamCreate 4.9 ms, two hipStreamDestroy 1.8 ms; MPDATA_HOST_TRACE=1 prints the phases), which is
// a sixth of the call at ncrms = 65536 and 95 % of it at the reference's shipped size (48 instances).
// mpdata_release_host_buffers() frees the calling thread's set; MPDATA_HOST_CACHE=0 keeps nothing (round 1-3's behaviour).
struct HostCtx {
  int dev = -1;
  hipStream_t s_in = nullptr, s_out = nullptr;
  ChunkBufs set[3];
  size_t cap[3] = {0, 0, 0};
  void release() {
    for (hipStream_t* st : {&s_in, &s_out})
      if (*st) { (void)hipStreamSynchronize(*st); (void)hipStreamDestroy(*st); *st = nullptr; }
    for (int i = 0; i < 3; ++i) { free_chunk(set[i]); cap[i] = 0; }
    dev = -1;
  }
  ~HostCtx() {   // (a thread that ends gives its buffers back; the process' last thread: before the runtime shuts down)
    if (dev < 0) return;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (hipSetDevice(dev) == hipSuccess) { release(); (void)hipSetDevice(cur); }
  }
};
thread_local HostCtx t_host;
bool host_cache_on() {
  static const bool off = getenv("MPDATA_HOST_CACHE") && !strcmp(getenv("MPDATA_HOST_CACHE"), "0");
  return !off;
}
}  // namespace

int mpdata_release_host_buffers(void) {
  if (t_host.dev < 0) return 0;
  int cur = 0;
  hipError_t e = hipGetDevice(&cur);
  if (e == hipSuccess) e = hipSetDevice(t_host.dev);
  if (e != hipSuccess) return hip_err(e, "mpdata_release_host_buffers");
  t_host.release();
  (void)hipSetDevice(cur);
  return 0;
}

int mpdata_advect_scalar2d(int64_t ncrms, int nx, int nz, int ntracers, double* f, const double* u,
                           const double* w, const double* rho, const double* rhow,
                           const double* adz, double* flux) {
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  if (!f || !u || !w || !rho || !rhow || !adz || !flux) return set_err(MPDATA_EINVAL, "null array pointer");
  // MPDATA_HOST_TRACE=1: microseconds since the call began at every phase boundary, on stderr
  static const bool trace = getenv("MPDATA_HOST_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto mark = [&](const char* what) {
    if (trace) fprintf(stderr, "[mpdata host call] %9.1f us  %s\n",
                       std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(), what);
  };
  const int64_t C = host_chunk(ncrms);
  const int64_t nchunks = (ncrms + C - 1) / C;
  const int nsets = nchunks >= 3 ? 3 : (int)nchunks;
  const size_t nzm = (size_t)nz - 1;
  const size_t rows_f = (size_t)(nx + 6) * nzm * ntracers, rows_u = (size_t)(nx + 5) * nzm,
               rows_w = (size_t)(nx + 4) * nz, rows_k = nzm, rows_kz = (size_t)nz, rows_x = (size_t)nz * ntracers;
  const size_t hp = (size_t)ncrms * 8;  // host pitch: one row of all instances
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  HostCtx once;                                         // (MPDATA_HOST_CACHE=0: released when the call returns)
  HostCtx& cx = host_cache_on() ? t_host : once;
  if (e == hipSuccess && cx.dev != dev) {               // the thread moved to another device: start afresh there
    if (cx.dev >= 0 && hipSetDevice(cx.dev) == hipSuccess) { cx.release(); (void)hipSetDevice(dev); }
    cx.dev = dev;
  }
  ChunkBufs* const set = cx.set;
  if (e == hipSuccess && !cx.s_in) e = hipStreamCreateWithFlags(&cx.s_in, hipStreamNonBlocking);
  if (e == hipSuccess && !cx.s_out) e = hipStreamCreateWithFlags(&cx.s_out, hipStreamNonBlocking);
  const hipStream_t s_in = cx.s_in, s_out = cx.s_out;
  for (int i = 0; i < nsets && e == hipSuccess; ++i) {
    ChunkBufs& b = set[i];
    const size_t nb[7] = {rows_f * C * 8, rows_u * C * 8, rows_w * C * 8, rows_k * C * 8, rows_kz * C * 8,
                          rows_k * C * 8, rows_x * C * 8};
    b.busy = false;
    e = arena_place(b.arena, cx.cap[i], nb);
    if (e == hipSuccess) {
      b.f = (double*)b.arena.p[0]; b.u = (double*)b.arena.p[1]; b.w = (double*)b.arena.p[2];
      b.rho = (double*)b.arena.p[3]; b.rhow = (double*)b.arena.p[4]; b.adz = (double*)b.arena.p[5];
      b.flux = (double*)b.arena.p[6];
    }
    if (e == hipSuccess && !b.run) e = hipEventCreateWithFlags(&b.run, hipEventDisableTiming);
  }
  mark("streams, buffers, events");
  // ---- the device -> host thread: takes finished chunks in order
  std::mutex mu;
  std::condition_variable cv;
  std::deque<OutJob> q;
  bool done = false;
  hipError_t e_out = hipSuccess;
  // (one chunk -- small problems: nothing to overlap, everything goes through s_in on this thread)
  const bool piped = nchunks > 1;
  std::thread out_thread;
  if (piped) out_thread = std::thread([&]() {
    (void)hipSetDevice(dev);
    for (;;) {
      OutJob j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done || !q.empty(); });
        if (q.empty()) return;
        j = q.front();
        q.pop_front();
      }
      ChunkBufs& b = set[j.set];
      const size_t dp = (size_t)j.cw * 8;
      hipError_t x = hipStreamWaitEvent(s_out, b.run, 0);
      if (x == hipSuccess) x = hipMemcpy2DAsync(f + j.c0, hp, b.f, dp, dp, rows_f, hipMemcpyDeviceToHost, s_out);
      if (x == hipSuccess) x = hipMemcpy2DAsync(flux + j.c0, hp, b.flux, dp, dp, rows_x, hipMemcpyDeviceToHost, s_out);
      if (x == hipSuccess) x = hipStreamSynchronize(s_out);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (x != hipSuccess && e_out == hipSuccess) e_out = x;
        b.busy = false;   // the set is free again
      }
      cv.notify_all();
    }
  });
  // ---- this thread: host -> device and the kernel of every chunk
  for (int64_t c = 0; c < nchunks && e == hipSuccess && rc == 0; ++c) {
    ChunkBufs& b = set[c % nsets];
    const int64_t c0 = c * C, cw = (c0 + C <= ncrms) ? C : ncrms - c0;
    const size_t dp = (size_t)cw * 8;  // device pitch: the chunk is its own problem, ld = cw
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !b.busy; });
      if (e_out != hipSuccess) break;
    }
    auto h2d = [&](double* d, const double* h, size_t rows) {
      if (e == hipSuccess) e = hipMemcpy2DAsync(d, dp, h + c0, hp, dp, rows, hipMemcpyHostToDevice, s_in);
    };
    h2d(b.f, f, rows_f);
    h2d(b.u, u, rows_u);
    h2d(b.w, w, rows_w);
    h2d(b.rho, rho, rows_k);
    h2d(b.rhow, rhow, rows_kz);
    h2d(b.adz, adz, rows_k);
    h2d(b.flux, flux, rows_x);  // level nz is never written (reference :541,:624): carry it through
    if (e != hipSuccess) break;
    if (c == 0) mark("first chunk: host -> device queued");
    rc = mpdata_advect_scalar2d_device(cw, nx, nz, ntracers, b.f, b.u, b.w, b.rho, b.rhow, b.adz, b.flux, (void*)s_in);
    if (rc) break;
    if (c == 0) mark("first chunk: kernel queued");
    if (!piped) {
      e = hipMemcpy2DAsync(f + c0, hp, b.f, dp, dp, rows_f, hipMemcpyDeviceToHost, s_in);
      if (e == hipSuccess) e = hipMemcpy2DAsync(flux + c0, hp, b.flux, dp, dp, rows_x, hipMemcpyDeviceToHost, s_in);
      break;   // (the stream is drained below)
    }
    e = hipEventRecord(b.run, s_in);
    if (e != hipSuccess) break;
    {
      std::lock_guard<std::mutex> lk(mu);
      b.busy = true;
      q.push_back(OutJob{(int)(c % nsets), c0, cw});
    }
    cv.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    done = true;
  }
  cv.notify_all();
  if (piped) out_thread.join();
  mark("all chunks back on the host");
  if (e == hipSuccess) e = e_out;
  for (hipStream_t st : {s_in, s_out})
    if (st) {
      const hipError_t e2 = hipStreamSynchronize(st);
      if (e == hipSuccess) e = e2;
    }
  mark("streams drained");
  if (rc || e != hipSuccess) cx.release();              // after an error nothing is kept
  if (rc) return rc;
  if (e != hipSuccess) return hip_err(e, "mpdata_advect_scalar2d (streamed host call)");
  mark("done");
  return 0;
}

int mpdata_fill_synthetic_device(double* a, int sid, int64_t rows, int64_t ncrms_global,
                                 int64_t sl0, int64_t nloc, uint64_t seed, int dist,
                                 void* stream) {
  return fill_device<double>(a, sid, rows, ncrms_global, sl0, nloc, seed, dist, stream);
}
int mpdata_fill_synthetic_f32_device(float* a, int sid, int64_t rows, int64_t ncrms_global,
                                     int64_t sl0, int64_t nloc, uint64_t seed, int dist,
                                     void* stream) {
  return fill_device<float>(a, sid, rows, ncrms_global, sl0, nloc, seed, dist, stream);
}

// fp32 host-array call: one piece (allocate, copy in, run, copy f and flux back).
int mpdata_advect_scalar2d_f32(int64_t ncrms, int nx, int nz, int ntracers, float* f, const float* u,
                               const float* w, const float* rho, const float* rhow,
                               const float* adz, float* flux) {
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  if (!f || !u || !w || !rho || !rhow || !adz || !flux) return set_err(MPDATA_EINVAL, "null array pointer");
  const Sizes sz = sizes_of(ncrms, nx, nz, ntracers);
  const size_t n[7] = {sz.f, sz.u, sz.w, sz.k, sz.kz, sz.k, sz.kz * (size_t)ntracers};
  const float* h[7] = {f, u, w, rho, rhow, adz, flux};
  float* d[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  Arena arena;
  const size_t nb[7] = {n[0] * 4, n[1] * 4, n[2] * 4, n[3] * 4, n[4] * 4, n[5] * 4, n[6] * 4};
  hipError_t e = arena_alloc(arena, nb);
  for (int i = 0; i < 7 && e == hipSuccess; ++i) d[i] = (float*)arena.p[i];
  for (int i = 0; i < 7 && e == hipSuccess; ++i) e = hipMemcpy(d[i], h[i], n[i] * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = mpdata_advect_scalar2d_f32_device(ncrms, nx, nz, ntracers, d[0], d[1], d[2], d[3], d[4], d[5], d[6], nullptr);
    if (rc == 0) e = hipMemcpy(f, d[0], n[0] * 4, hipMemcpyDeviceToHost);
    if (rc == 0 && e == hipSuccess) e = hipMemcpy(flux, d[6], n[6] * 4, hipMemcpyDeviceToHost);
  }
  arena_free(arena);
  if (rc) return rc;
  if (e != hipSuccess) return hip_err(e, "mpdata_advect_scalar2d_f32");
  return 0;
}

static int pack_common(const double* full, double* shard, int64_t rows, int64_t ncrms, int64_t sl0,
                       int64_t nloc, void* stream, int unpack) {
  if (!full || !shard || rows < 1 || nloc < 1 || sl0 < 0 || sl0 + nloc > ncrms)
    return set_err(MPDATA_EINVAL, "bad argument to mpdata_(un)pack_shard_device");
  hipLaunchKernelGGL(pack_kernel, dim3(grid_for(rows * nloc, 256)), dim3(256), 0,
                     (hipStream_t)stream, full, shard, (long long)rows, (long long)ncrms,
                     (long long)sl0, (long long)nloc, unpack);
  HIP_TRY(hipGetLastError());
  return 0;
}
int mpdata_pack_shard_device(const double* full, double* shard, int64_t rows, int64_t ncrms,
                             int64_t sl0, int64_t nloc, void* stream) {
  return pack_common(full, shard, rows, ncrms, sl0, nloc, stream, 0);
}
int mpdata_unpack_shard_device(double* full, const double* shard, int64_t rows, int64_t ncrms,
                               int64_t sl0, int64_t nloc, void* stream) {
  return pack_common(full, const_cast<double*>(shard), rows, ncrms, sl0, nloc, stream, 1);
}

// ---- small device workspace API for hosts without HIP access of their own (the Fortran driver's
//      device-resident mode: global arrays generated on the root GPU, scattered from there)
int mpdata_device_alloc(void** p, int64_t bytes) {
  if (!p || bytes < 1) return set_err(MPDATA_EINVAL, "bad argument to mpdata_device_alloc");
  *p = nullptr;
  HIP_TRY(hipMalloc(p, (size_t)bytes));
  return 0;
}
// ... on the device a plan's full-width arrays must live on: the plan's own device, the ROOT GPU
// (shard 0's device) of a multi-GPU plan
int mpdata_plan_device_alloc(mpdata_plan* plan, void** p, int64_t bytes) {
  if (!plan) return set_err(MPDATA_EINVAL, "null plan");
  int dev = plan->device;
  if (plan->multi) {
    const int rc = mpdata_multi_info(plan->multi, 0, &dev, nullptr, nullptr);
    if (rc) return rc;
  }
  DevGuard g(dev);
  return mpdata_device_alloc(p, bytes);
}
int mpdata_device_free(void* p) {
  if (p) HIP_TRY(hipFree(p));
  return 0;
}
namespace {
constexpr int SUM_BLOCKS = 1024, SUM_THREADS = 256;
// partial sums of the elements j < n with (j % stride) < block; fixed grid and tree: the result
// does not depend on the run
__global__ void sum_kernel(const double* a, long long n, long long block, long long stride, double* part) {
  __shared__ double sh[SUM_THREADS];
  double s = 0.0;
  for (long long j = (long long)blockIdx.x * SUM_THREADS + threadIdx.x; j < n; j += (long long)SUM_BLOCKS * SUM_THREADS)
    if (j % stride < block) s += a[j];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int w = SUM_THREADS / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
}  // namespace
int mpdata_device_sum(const double* a, int64_t n, int64_t block, int64_t stride, double* sum) {
  if (!a || !sum || n < 1 || block < 1 || stride < block) return set_err(MPDATA_EINVAL, "bad argument to mpdata_device_sum");
  DevGuard g(device_of(a));
  double* part = nullptr;
  HIP_TRY(hipMalloc(&part, SUM_BLOCKS * sizeof(double)));
  hipLaunchKernelGGL(sum_kernel, dim3(SUM_BLOCKS), dim3(SUM_THREADS), 0, nullptr, a, (long long)n, (long long)block,
                     (long long)stride, part);
  double h[SUM_BLOCKS];
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpy(h, part, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(part);
  if (e != hipSuccess) return hip_err(e, "mpdata_device_sum");
  double s = 0.0;
  for (int i = 0; i < SUM_BLOCKS; ++i) s += h[i];
  *sum = s;
  return 0;
}

int mpdata_set_variant(int v) {
  const int prev = variant();
  if (v == MPDATA_VARIANT_EXACT || v == MPDATA_VARIANT_FAST) g_variant = v;
  return prev;
}
int mpdata_get_variant(void) { return variant(); }
int mpdata_set_wm_flags(int flags) {
  const int prev = wm_flags();
  if (flags >= 0) g_wm_flags = flags & (MPDATA_WMF_NOSTREAM | MPDATA_WMF_TPW1 | MPDATA_WMF_NOSPLIT | MPDATA_WMF_SPLIT);
  return prev;
}
int mpdata_set_serpentine(int on) {
  const int prev = serpentine();
  if (on == 0 || on == 1) g_serpentine = on;
  return prev;
}
int mpdata_set_tile(int tile) {
  const int prev = tile_override();
  g_tile = tile < 0 ? -1 : tile;
  return prev;
}
// Diagnostic builds (-DMPDWM_STAMPS, tools/wave_timeline.py) write 8 words per wave of the plan kernels into
// this device buffer (start / end on the 100-MHz counter and the shader clock, cycles in the counted DMA
// waits, HW_ID, XCC_ID, tile); production builds ignore it.
int mpdata_set_debug_buffer(void* dev_ptr) {
  g_dbg = (unsigned long long*)dev_ptr;
  return 0;
}
int mpdata_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
// minimal HBM traffic of one call, in bytes.
int64_t mpdata_algorithmic_bytes(int64_t ncrms, int nx, int nz, int ntracers) {
  const int64_t nzm = nz - 1;
  return ncrms * 8 * nzm * ((int64_t)ntracers * (2 * nx + 11) + 2 * nx + 12);
}
int64_t mpdata_algorithmic_bytes_f32(int64_t ncrms, int nx, int nz, int ntracers) {
  return mpdata_algorithmic_bytes(ncrms, nx, nz, ntracers) / 2;
}
const char* mpdata_last_error(void) { return g_err.c_str(); }
// The version string names every timing-ablation / experiment macro the kernels were compiled with
// (mpdata_kernels_inst.h: build_flags); a shipped library has none, tests/test_capi_abi.py checks.
const char* mpdata_version(void) {
  static std::string v;
  if (v.empty()) {
    const char* fe = mpdata_exact::build_flags();
    const char* ff = mpdata_fast::build_flags();
    v = std::string("mpdata-hip 0.3 (gfx950) exact[") + fe + "] fast[" + ff + "]";
  }
  return v.c_str();
}

}  // extern "C"
