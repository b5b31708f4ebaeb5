// mpdata_core.hip -- the part of the C-ABI of libmpdata_hip.so (include/mpdata_hip.h) that is not a plan and not
// a host-array call: error text, variant / tile / layout settings, argument validation and tile choice, the calls on
// reference-layout DEVICE arrays, the synthetic input generator, shard pack / unpack, the small device workspace API.
// No CPU compute path exists in this library: every entry point either runs HIP kernels or returns an error.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "mpdata_internal.h"

namespace {
thread_local std::string g_err;
int g_variant = -1;  // -1: read MPDATA_VARIANT on first use
int g_tile = -2;     // -2: read MPDATA_TILE on first use; -1: automatic
unsigned long long* g_dbg = nullptr;  // diagnostic stamp buffer (device), see mpdata_set_debug_buffer
}  // namespace

namespace mpd {
int set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
unsigned long long* debug_buffer() { return g_dbg; }
}  // namespace mpd
int mpdata_internal_set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

namespace mpd {
int hip_err(hipError_t e, const char* what) {
  return set_err((int)e, "%s: %s", what, hipGetErrorString(e));
}


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
int choose_tile(int var, int64_t ncrms, int nx, int nz, MpdataTileInfo* out, int elem_bytes) {
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

// the kernel launch(es) of one run of a single-device plan, on the plan's stream
// (u_ref, w_ref != null: the kernel that reads u, w from these reference-layout arrays)
unsigned grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride the rest
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

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
// The park array of an EXACT device call is kept per HOST THREAD and STREAM between calls (calls on one stream are ordered,
// so they can share it; another stream gets its own).  Rounds 4-5 allocated it in stream order around every call
// (hipMallocAsync / hipFreeAsync): on the legacy default stream that produced, once in ~50 runs of the fp32 Fortran
// driver, a launch whose finishing kernel read something else than the main kernel had parked for ONE workgroup
// (flux of 32 instances wrong, f right; never with a plain hipMalloc: round 5, 0 of 150 against 3 of 150).
// mpdata_release_host_buffers() frees the calling thread's buffers; a thread that ends frees its own.
struct ParkBuf {
  int dev;
  hipStream_t stream;
  void* p;
  size_t cap;
};
struct ParkCache {
  ParkBuf b[8];
  int n = 0;
  void drop(int i) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;   // (the runtime is gone: process exit)
    if (hipSetDevice(b[i].dev) == hipSuccess) {
      (void)hipDeviceSynchronize();   // (the stream itself may have been destroyed by its owner)
      (void)hipFree(b[i].p);
      (void)hipSetDevice(cur);
    }
    b[i] = b[--n];
  }
  void release() {
    while (n > 0) drop(n - 1);
  }
  ~ParkCache() { release(); }
};
thread_local ParkCache t_park;
int park_buffer(hipStream_t stream, size_t bytes, void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  for (int i = 0; i < t_park.n; ++i)
    if (t_park.b[i].dev == dev && t_park.b[i].stream == stream) {
      if (t_park.b[i].cap >= bytes) { *out = t_park.b[i].p; return 0; }
      t_park.drop(i);
      break;
    }
  if (t_park.n == 8) t_park.drop(0);
  void* p = nullptr;
  const size_t cap = bytes + bytes / 4;
  HIP_TRY(hipMalloc(&p, cap));
  t_park.b[t_park.n++] = ParkBuf{dev, stream, p, cap};
  *out = p;
  return 0;
}
void park_buffers_release() { t_park.release(); }

// var: MPDATA_VARIANT_* (a plan passes the variant it was created with; < 0: the global one)
template <typename R>
int advect_device(int64_t ncrms, int nx, int nz, int ntracers, R* f, const R* u, const R* w,
                  const R* rho, const R* rhow, const R* adz, R* flux, void* stream, int var, bool staged) {
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  if (!f || !u || !w || !rho || !rhow || !adz || !flux)
    return set_err(MPDATA_EINVAL, "null array pointer");
  if (var < 0) var = variant();
  if (staged && staged_call_applies(ncrms, nz, (int)sizeof(R)))   // 65 <= nz <= 127: through a wave-major plan (mpdata_plan.hip)
    return staged_device_call((int)sizeof(R), ncrms, nx, nz, ntracers, f, u, w, rho, rhow, adz, flux, stream, var);
  MpdataTileInfo t;
  rc = choose_tile(var, ncrms, nx, nz, &t, (int)sizeof(R));
  if (rc) return rc;
  MpdataArgsT<R> a;
  a.f = f; a.u = u; a.w = w; a.rho = rho; a.rhow = rhow; a.adz = adz; a.flux = flux;
  a.ncrms = ncrms; a.nx = nx; a.nz = nz; a.ntracers = ntracers;
  a.f_tstride = (long long)ncrms * (nx + 6) * (nz - 1);
  a.flux_tstride = (long long)ncrms * nz;
  a.dbg = g_dbg;
  // EXACT, x-marching kernels: the park array of the bit-identical flux (see xmarch_flux_finish_kernel), [workgroup][nx]
  // [thread], where the register park does not apply (nx > 36, 16-wave workgroups): a buffer kept per host thread and
  // stream (park_buffer).  (The k-marching kernels add in the reference's
  // order by construction; FAST never parks; MPDATA_EXACT_FLUX=sum does without.)
  a.wpark = nullptr;
  a.park_regs = 0;
  void* park_mem = nullptr;
  const bool big = (double)ncrms * (nx + 6) * nz * (double)(t.id >= 40 ? 8 : t.elem_bytes) >= 4294967000.0 * (t.id >= 40 ? 2 : 1);
  if (var == MPDATA_VARIANT_EXACT && t.nz_max < (1 << 30) && exact_flux_in_regs() && nx <= MPDATA_WM_NPK && t.threads <= 512) {
    a.park_regs = 1;   // (round 5) in registers: no park array, no finishing kernel
  } else if (var == MPDATA_VARIANT_EXACT && t.nz_max < (1 << 30) && !big && exact_flux_in_order()) {
    const size_t groups = (size_t)((ncrms + t.slw - 1) / t.slw);
    const size_t esz = t.id >= 40 ? 8 : (size_t)t.elem_bytes;   // (tile ids 40..: two fp32 instances per lane)
    const size_t bytes = (size_t)ntracers * groups * (size_t)nx * (size_t)t.threads * esz;
    const int prc = park_buffer((hipStream_t)stream, bytes, &park_mem);
    if (prc) return prc;
    a.wpark = (R*)park_mem;
  }
  bool ok;
  if constexpr (sizeof(R) == 8)
    ok = var == MPDATA_VARIANT_FAST ? mpdata_fast::launch(t.id, a, ntracers, stream)
                                    : mpdata_exact::launch(t.id, a, ntracers, stream);
  else
    ok = var == MPDATA_VARIANT_FAST ? mpdata_fast::launch_f32(t.id, a, ntracers, stream)
                                    : mpdata_exact::launch_f32(t.id, a, ntracers, stream);
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
  DevGuard g(stream ? cur_dev : device_of(a));
  hipLaunchKernelGGL(fill_kernel<R>, dim3(grid_for(rows * nloc, 256)), dim3(256), 0,
                     (hipStream_t)stream, a, base, shift, (long long)rows, (long long)ncrms_global,
                     (long long)sl0, (long long)nloc);
  HIP_TRY(hipGetLastError());
  return 0;
}

template int advect_device<double>(int64_t, int, int, int, double*, const double*, const double*, const double*,
                                   const double*, const double*, double*, void*, int, bool);
template int advect_device<float>(int64_t, int, int, int, float*, const float*, const float*, const float*,
                                  const float*, const float*, float*, void*, int, bool);

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

}  // namespace mpd
using namespace mpd;

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

int mpdata_set_plan_layout(int layout) {
  const int prev = plan_layout_default();
  if (layout == MPDATA_LAYOUT_REFERENCE || layout == MPDATA_LAYOUT_WAVEMAJOR) g_layout = layout;
  return prev;
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
