// mpdata_plan.hip -- plans: the device-resident state of one problem behind the C-ABI (include/mpdata_hip.h 3).
// A plan owns the device state of one problem (what the OpenACC `enter data pcreate` of the
// reference does, :105, :280, :662) on the device that was current when it was created; every
// plan call switches to that device and back.  Variant and layout are fixed at creation.
// fp64 plans with nz <= 64 keep the arrays in the wave-major layout of
// mpdata_kernel_wm_body.h and convert in upload / download / import / export; other plans
// (fp32; nz > 64) keep the reference layout and run the x-/k-marching kernels.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "mpdata_internal.h"

using namespace mpd;

namespace {
// EXACT, bit-identical flux: the plan kernels store the upwind sum as flux and PARK the nx limited vertical fluxes of
// every lane ([tracer][tile][column][lane], mpdata_kernel_wm_body.h); this adds them, one by one in the reference's
// order i = 1 .. nx (:624), onto that sum.  A wave per (tracer, tile), lane -> (instance, level) as in the kernels.
// R2 = double, or float2 for fp32 plans (two adjacent instances per lane; no contraction: this file is built with
// -ffp-contract=off).
template <typename R2>
__global__ void __launch_bounds__(256) flux_finish_kernel(R2* flux, const R2* park, int ntiles, int nx, int nzm, int lps,
                                                          long long flux_tstride, int ntr, int nkw) {
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // wave = (tracer * ntiles + tile) * nkw + wave of the instance
  if (w >= (long long)ntr * ntiles * nkw) return;
  const int h = (int)(w % nkw);
  const long long tt = w / nkw;
  const int tr = (int)(tt / ntiles), tile = (int)(tt % ntiles);
  int s_l = lane / lps, kk = lane % lps, chunk = (64 / lps) * nzm;
  if (lps > 64) {   // nz > 64: the wave's 64-level window and the part of it the plan kernel stores (mpdata_kernel_wm_body.h: koff, out_ok)
    const int nz = nzm + 1;
    const int koff = min(58 * h, max(0, (nz - 64 + 1) & ~1));
    s_l = 0; kk = lane + koff; chunk = nzm;
    const int k = kk + 1;
    if (!((h == 0 || k >= 58 * h + 4) && (h + 1 == nkw || k <= 58 * h + 61))) return;
  }
  if (kk >= nzm) return;
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

// lanes per instance of the wave-major kernels; 128 = an instance wider than a wave (65 <= nz <= 238: several waves
// per instance, mpdata_kernel_wm_body.h "KS"; the layout kernels hold a column of one instance group in LDS: nzm <= 126)
#define MPDATA_WM_NZ_MAX 238   // 64 + 3 * 58: four windows = the four waves of a workgroup
int wm_lps_for(int nz) { return nz <= 8 ? 8 : nz <= 16 ? 16 : nz <= 32 ? 32 : nz <= 64 ? 64 : nz <= MPDATA_WM_NZ_MAX ? 128 : 0; }
int wm_nkw_for(int nz) { return nz <= 64 ? 1 : 1 + (nz - 64 + 57) / 58; }
// nz > 64: lanes of the last window where it is a share of a wave (16: the window needs <= 16 levels, 32: <= 32; else 0 = a
// wave of its own); MPDATA_KS_TAIL=0: never (A/B)
int wm_lwt_for(int nz) {
  static const bool off = getenv("MPDATA_KS_TAIL") && !strcmp(getenv("MPDATA_KS_TAIL"), "0");
  if (nz <= 64 || off || wm_nkw_for(nz) != 2) return 0;   // (three windows: a 9-wave workgroup would cap the two-waves-per-SIMD forms' registers)
  const int need = nz - 58 * (wm_nkw_for(nz) - 1);
  return need <= 16 ? 16 : need <= 32 ? 32 : 0;
}

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

static int plan_create(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan, int eb, int var_in = -1) {
  if (!plan) return set_err(MPDATA_EINVAL, "null plan pointer");
  *plan = nullptr;
  int rc = validate(ncrms, nx, nz, ntracers);
  if (rc) return rc;
  const int var = var_in >= 0 ? var_in : variant();
  // wave-major: fp64, and fp32 with an even ncrms (two adjacent instances per lane = 8-byte elements)
  const bool wmaj = (eb == 8 || (ncrms & 1) == 0) && wm_lps_for(nz) != 0 &&
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
    p->park_regs = var == MPDATA_VARIANT_EXACT && exact_flux_in_regs() && nx <= MPDATA_WM_NPK2;
    if (e == hipSuccess && var == MPDATA_VARIANT_EXACT && exact_flux_in_order() && !p->park_regs) {
      p->wpark_bytes = (size_t)ntracers * p->ntiles * wm_nkw_for(nz) * (size_t)nx * 64 * 8;
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
  const long long waves = (long long)count * p->ntiles * a.nkw;
  const unsigned blocks = (unsigned)((waves + 3) / 4);
  if (p->eb == 8)
    hipLaunchKernelGGL(flux_finish_kernel<double>, dim3(blocks), dim3(256), 0, p->stream, a.flux, (const double*)a.wpark, p->ntiles,
                       p->nx, p->nz - 1, p->lps, a.flux_tstride, count, a.nkw);
  else
    hipLaunchKernelGGL(flux_finish_kernel<float2>, dim3(blocks), dim3(256), 0, p->stream, (float2*)a.flux, (const float2*)a.wpark,
                       p->ntiles, p->nx, p->nz - 1, p->lps, a.flux_tstride, count, a.nkw);
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
    a.dbg = debug_buffer();   // (null unless a diagnostic build was handed a stamp buffer)
    // EXACT plans: the park array of the limited vertical fluxes (bit-identical flux; allocated with the plan)
    a.park_regs = (p->park_regs && !u_ref) ? 1 : 0;
    a.nkw = wm_nkw_for(p->nz);
    if (u_ref && p->park_regs && !p->wpark) {
      // the kernel that reads u, w from the reference layout has no register-park form (its EXACT build takes every
      // register it can get): the park array after all, allocated by the first such call
      p->wpark_bytes = (size_t)p->ntracers * p->ntiles * wm_nkw_for(p->nz) * (size_t)p->nx * 64 * 8;
      const hipError_t e = hipMalloc(&p->wpark, p->wpark_bytes);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        p->wpark = nullptr;
        return set_err((int)e, "mpdata_plan_run_uw (EXACT): no memory for the %.1f-GB park array of the bit-identical flux; "
                               "MPDATA_EXACT_FLUX=sum does without it", p->wpark_bytes / 1e9);
      }
    }
    a.wpark = (p->wpark && !a.park_regs) ? (double*)p->wpark + (long long)first * p->ntiles * a.nkw * ((long long)p->nx * 64) : nullptr;
    a.lwt = (a.wpark || u_ref) ? 0 : wm_lwt_for(p->nz);   // (the park array is laid out for whole-wave windows)
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

// ---- calls on reference-layout device arrays with 65 <= nz <= 238 (round 5).  No x-marching kernel holds such an
//      instance in a wave and the k-marching fall-back runs at 13-16 Gcu/s (fp64, nx <= 140 only; fp32: nothing).  Such
//      a call goes through a wave-major plan kept per host thread instead: import f, u, w, rho, rhow, adz, flux
//      (the layout kernels), the plan kernel (several waves per instance), export f and flux -- everything on the
//      caller's stream, asynchronous as the direct call is, same results (EXACT bit-identical incl. flux; flux(:,nz)
//      carried through).  Costs the plan's memory (as much again as the call's arrays) until
//      mpdata_release_host_buffers() or the end of the thread; MPDATA_DEVICE_CALL=direct keeps the k-marching kernel.
namespace {
struct StagedPlan {
  mpdata_plan* p = nullptr;
  int var = -1;
  void release() {
    if (!p) return;
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess) mpdata_plan_destroy(p);   // (else the runtime is gone: process exit)
    p = nullptr;
  }
  ~StagedPlan() { release(); }
};
thread_local StagedPlan t_staged;
}  // namespace
extern "C++" void mpd::staged_plan_release() { t_staged.release(); }
extern "C++" bool mpd::staged_call_applies(int64_t ncrms, int nz, int eb) {
  static const bool direct = getenv("MPDATA_DEVICE_CALL") && !strcmp(getenv("MPDATA_DEVICE_CALL"), "direct");
  return !direct && nz > 64 && wm_lps_for(nz) != 0 && (eb == 8 || (ncrms & 1) == 0) &&
         plan_layout_default() == MPDATA_LAYOUT_WAVEMAJOR && tile_override() < 0;
}
extern "C++" int mpd::staged_device_call(int eb, int64_t ncrms, int nx, int nz, int ntracers, void* f, const void* u, const void* w,
                                         const void* rho, const void* rhow, const void* adz, void* flux, void* stream, int var) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  mpdata_plan* p = t_staged.p;
  if (p && !(p->ncrms == ncrms && p->nx == nx && p->nz == nz && p->ntracers == ntracers && p->eb == eb && p->variant == var &&
             p->device == dev)) {
    t_staged.release();
    p = nullptr;
  }
  if (!p) {
    const int rc = plan_create(ncrms, nx, nz, ntracers, &p, eb, var);
    if (rc) return rc;
    if (p->layout != MPDATA_LAYOUT_WAVEMAJOR) {   // (staged_call_applies and plan_create disagree: a bug, not a fall-back)
      mpdata_plan_destroy(p);
      return set_err(MPDATA_EINVAL, "internal: staged device call without a wave-major plan");
    }
    (void)hipStreamDestroy(p->stream);
    p->stream = (hipStream_t)stream;
    p->own_stream = false;
    p->timing = false;
    t_staged.p = p;
  }
  if (p->stream != (hipStream_t)stream) {   // the previous call's work may still use the plan's arrays
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->stream = (hipStream_t)stream;
  }
  int rc = plan_import(p, f, u, w, rho, rhow, adz, flux, 0, ntracers, true);
  if (rc) return rc;
  p->uploaded = true;
  rc = plan_launch(p, 0, ntracers);
  if (rc) return rc;
  return plan_export(p, f, flux, 0, ntracers, true);
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
}  // extern "C"
