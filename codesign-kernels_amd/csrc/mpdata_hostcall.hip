// mpdata_hostcall.hip -- the host-array calls of the C-ABI: mpdata_advect_scalar2d (the literal drop-in: host arrays in,
// host arrays out, transfers inside; chunked and pipelined) and its fp32 form.
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "mpdata_internal.h"

using namespace mpd;

extern "C" {

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
  park_buffers_release();   // (the park arrays of this thread's EXACT device calls)
  staged_plan_release();    // (the plan behind this thread's device calls at nz > 64)
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
    // (nz > 64: a device call may go through a plan kept per thread and SHAPE -- chunks of two widths would rebuild it
    //  twice per call, and the transfers bound this path anyway: only where nothing else runs the shape)
    rc = advect_device<double>(cw, nx, nz, ntracers, b.f, b.u, b.w, b.rho, b.rhow, b.adz, b.flux, (void*)s_in, -1,
                               !piped || nx > 140);
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

}  // extern "C"
