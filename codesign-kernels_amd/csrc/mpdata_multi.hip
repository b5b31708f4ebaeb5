// mpdata_multi.hip -- one problem sharded over the GPUs of a node, behind the plan API
// (mpdata_plan_create_multi, include/mpdata_hip.h section 3b).
//
// The routine couples no two CRM instances (reference
// mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:505-637 index every array with the
// same `sl`), so the ncrms axis is cut into one contiguous block per GPU and every GPU runs an
// ordinary single-device plan on its block: NO data-path collective.  What replaces the
// reference's `!$acc update device / update host` traffic (:107, :241-242, :662-663) is a
// scatter of the inputs from, and a gather of f and flux to, the root GPU:
//
//   upload:   host array --H2D--> root staging (full width) --pack kernel--> one contiguous
//             block per peer --ncclSend / ncclRecv in ONE group (RCCL over xGMI: every peer has
//             its own link to the root, no ring)--> peer --layout conversion--> its plan
//   download: the reverse (export, ncclSend to the root, unpack kernel, D2H).
//
// Because `sl` is the FASTEST array axis, a block is a strided slab of every array (rows of
// nloc*8 bytes at a pitch of ncrms*8): hence the pack / unpack kernels on the root.
// One host thread drives all devices (hipSetDevice per device, one stream each; the single
// process owns all communicators, ncclCommInitAll).  MPDATA_MULTI_XFER selects the transport:
//   rccl   (default) as above;
//   p2p    hipMemcpyPeerAsync root <-> peer instead of ncclSend/ncclRecv (also xGMI);
//   direct no root at all: every GPU copies its slab from / to the host arrays itself
//          (hipMemcpy2DAsync), eight PCIe links in parallel.
// Results are bitwise those of a single-GPU plan (tests/test_multi_plan.py).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mpdata_hip.h"
#include "mpdata_multi.h"

namespace {

enum Xfer { XFER_RCCL = 0, XFER_P2P = 1, XFER_DIRECT = 2 };

__global__ void slab_kernel(const char* full, char* shard, long long rows, long long ncrms, long long sl0,
                            long long nloc, int eb, int unpack) {
  // 8-byte (or 4-byte) elements; one thread per element, grid-stride
  const long long total = rows * nloc;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / nloc, s = t - r * nloc;
    const long long fi = r * ncrms + sl0 + s;
    if (eb == 8) {
      if (unpack) ((double*)full)[fi] = ((const double*)shard)[t];
      else ((double*)shard)[t] = ((const double*)full)[fi];
    } else {
      if (unpack) ((float*)full)[fi] = ((const float*)shard)[t];
      else ((float*)shard)[t] = ((const float*)full)[fi];
    }
  }
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct mpdata_multi {
  int64_t ncrms;
  int nx, nz, ntracers, eb, ngpus;
  Xfer xfer;
  std::vector<int> dev;
  std::vector<int64_t> sl0, nloc;
  std::vector<mpdata_plan*> sub;
  std::vector<hipStream_t> stream;
  std::vector<void*> rb;          // per device: one block of one array (tracer), contiguous
  std::vector<void*> pk;          // root: packed block per peer (index g; pk[0] unused)
  void* stage_full = nullptr;     // root: one array (tracer) at full width
  std::vector<ncclComm_t> comm;
  bool comm_ok = false;
  size_t max_rows = 0;
  double scatter_s = 0, gather_s = 0;
  int64_t scatter_bytes_peer = 0, gather_bytes_peer = 0;
};

namespace {

#define M_HIP(expr)                                                                         \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess) return mpdata_internal_set_err((int)e_, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)
#define M_NCCL(expr)                                                                        \
  do {                                                                                      \
    ncclResult_t r_ = (expr);                                                               \
    if (r_ != ncclSuccess) return mpdata_internal_set_err(MPDATA_ECOMM, "%s: %s", #expr, ncclGetErrorString(r_)); \
  } while (0)
#define M_TRY(expr)            \
  do {                         \
    int rc_ = (expr);          \
    if (rc_) return rc_;       \
  } while (0)

// MPDATA_MULTI_TRACE=1: one stderr line per step of scatter / gather (diagnosis of stalls)
bool trace_on() {
  static const bool on = getenv("MPDATA_MULTI_TRACE") != nullptr;
  return on;
}
#define M_TRACE(...)                                   \
  do {                                                 \
    if (trace_on()) {                                  \
      fprintf(stderr, "[mpdata_multi %.3f] ", now_s()); \
      fprintf(stderr, __VA_ARGS__);                    \
      fputc('\n', stderr);                             \
      fflush(stderr);                                  \
    }                                                  \
  } while (0)

unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int sync_all(mpdata_multi* m) {
  for (int g = 0; g < m->ngpus; ++g) {
    M_HIP(hipSetDevice(m->dev[g]));
    M_HIP(hipStreamSynchronize(m->stream[g]));
  }
  return 0;
}

// which: 0 f, 1 u, 2 w, 3 rho, 4 rhow, 5 adz, 6 flux
size_t rows_of(const mpdata_multi* m, int which) {
  const size_t nzm = (size_t)m->nz - 1;
  switch (which) {
    case 0: return (size_t)(m->nx + 6) * nzm;
    case 1: return (size_t)(m->nx + 5) * nzm;
    case 2: return (size_t)(m->nx + 4) * m->nz;
    case 3: case 5: return nzm;
    default: return (size_t)m->nz;
  }
}

int import_block(mpdata_multi* m, int g, int which, int tracer) {
  const void* a[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  a[which] = m->rb[g];
  return mpdata_plan_import_device(m->sub[g], a[0], a[1], a[2], a[3], a[4], a[5], a[6], tracer, 1);
}

// One array (one tracer of f / flux) from the host to every GPU's plan.
int scatter_array(mpdata_multi* m, int which, const void* host, int tracer) {
  const size_t rows = rows_of(m, which), eb = (size_t)m->eb;
  const int G = m->ngpus;
  M_TRACE("scatter array %d tracer %d transport %d", which, tracer, (int)m->xfer);
  if (m->xfer == XFER_DIRECT) {
    // every GPU pulls its own slab (the library does not page-lock caller memory, see
    // mpdata_capi.hip: the runtime stages the strided copies)
    int rc = 0;
    for (int g = 0; g < G && !rc; ++g) {
      hipError_t e = hipSetDevice(m->dev[g]);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync(m->rb[g], (size_t)m->nloc[g] * eb, (const char*)host + (size_t)m->sl0[g] * eb,
                             (size_t)m->ncrms * eb, (size_t)m->nloc[g] * eb, rows, hipMemcpyHostToDevice, m->stream[g]);
      if (e != hipSuccess) rc = mpdata_internal_set_err((int)e, "direct scatter to device %d: %s", m->dev[g], hipGetErrorString(e));
      if (!rc) rc = import_block(m, g, which, tracer);
    }
    M_TRACE("direct scatter: copies + imports queued, rc=%d", rc);
    const int rs = sync_all(m);   // (also on the error path: nothing may be left in flight)
    M_TRACE("direct scatter: synchronised");
    return rc ? rc : rs;
  }
  // root: full-width copy in, one pack per GPU (the root's own block straight into its buffer)
  M_HIP(hipSetDevice(m->dev[0]));
  M_HIP(hipMemcpyAsync(m->stage_full, host, rows * (size_t)m->ncrms * eb, hipMemcpyHostToDevice, m->stream[0]));
  for (int g = 0; g < G; ++g) {
    void* dst = g == 0 ? m->rb[0] : m->pk[g];
    hipLaunchKernelGGL(slab_kernel, dim3(grid_for((long long)rows * m->nloc[g])), dim3(256), 0, m->stream[0],
                       (const char*)m->stage_full, (char*)dst, (long long)rows, (long long)m->ncrms,
                       (long long)m->sl0[g], (long long)m->nloc[g], m->eb, 0);
  }
  M_HIP(hipGetLastError());
  M_TRACE("scatter: staged + packed");
  if (m->xfer == XFER_RCCL) {
    const ncclDataType_t dt = m->eb == 8 ? ncclDouble : ncclFloat;
    M_NCCL(ncclGroupStart());
    for (int g = 1; g < G; ++g) {
      const size_t n = rows * (size_t)m->nloc[g];
      M_NCCL(ncclSend(m->pk[g], n, dt, g, m->comm[0], m->stream[0]));
      M_NCCL(ncclRecv(m->rb[g], n, dt, 0, m->comm[g], m->stream[g]));
    }
    M_NCCL(ncclGroupEnd());
  } else {
    M_HIP(hipStreamSynchronize(m->stream[0]));
    for (int g = 1; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipMemcpyPeerAsync(m->rb[g], m->dev[g], m->pk[g], m->dev[0], rows * (size_t)m->nloc[g] * eb, m->stream[g]));
    }
  }
  M_TRACE("scatter: transfers queued");
  for (int g = 0; g < G; ++g) {
    M_HIP(hipSetDevice(m->dev[g]));
    M_TRY(import_block(m, g, which, tracer));
  }
  M_TRACE("scatter: imports queued");
  const int rs = sync_all(m);
  M_TRACE("scatter: synchronised");
  return rs;
}

// f (which = 0) or flux (6) of one tracer from every GPU's plan to the host array.
int gather_array(mpdata_multi* m, int which, void* host, int tracer) {
  const size_t rows = rows_of(m, which), eb = (size_t)m->eb;
  const int G = m->ngpus;
  M_TRACE("gather array %d tracer %d transport %d", which, tracer, (int)m->xfer);
  for (int g = 0; g < G; ++g) {
    M_HIP(hipSetDevice(m->dev[g]));
    M_TRY(mpdata_plan_export_device(m->sub[g], which == 0 ? m->rb[g] : nullptr, which == 6 ? m->rb[g] : nullptr, tracer, 1));
  }
  if (m->xfer == XFER_DIRECT) {
    int rc = 0;
    for (int g = 0; g < G && !rc; ++g) {
      hipError_t e = hipSetDevice(m->dev[g]);
      if (e == hipSuccess)
        e = hipMemcpy2DAsync((char*)host + (size_t)m->sl0[g] * eb, (size_t)m->ncrms * eb, m->rb[g], (size_t)m->nloc[g] * eb,
                             (size_t)m->nloc[g] * eb, rows, hipMemcpyDeviceToHost, m->stream[g]);
      if (e != hipSuccess) rc = mpdata_internal_set_err((int)e, "direct gather from device %d: %s", m->dev[g], hipGetErrorString(e));
    }
    const int rs = sync_all(m);
    return rc ? rc : rs;
  }
  if (m->xfer == XFER_RCCL) {
    const ncclDataType_t dt = m->eb == 8 ? ncclDouble : ncclFloat;
    M_NCCL(ncclGroupStart());
    for (int g = 1; g < G; ++g) {
      const size_t n = rows * (size_t)m->nloc[g];
      M_NCCL(ncclSend(m->rb[g], n, dt, 0, m->comm[g], m->stream[g]));
      M_NCCL(ncclRecv(m->pk[g], n, dt, g, m->comm[0], m->stream[0]));
    }
    M_NCCL(ncclGroupEnd());
  } else {
    for (int g = 1; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipStreamSynchronize(m->stream[g]));
      M_HIP(hipMemcpyPeerAsync(m->pk[g], m->dev[0], m->rb[g], m->dev[g], rows * (size_t)m->nloc[g] * eb, m->stream[g]));
      M_HIP(hipStreamSynchronize(m->stream[g]));
    }
  }
  M_TRACE("gather: transfers done / queued");
  M_HIP(hipSetDevice(m->dev[0]));
  for (int g = 0; g < G; ++g) {
    const void* src = g == 0 ? m->rb[0] : m->pk[g];
    hipLaunchKernelGGL(slab_kernel, dim3(grid_for((long long)rows * m->nloc[g])), dim3(256), 0, m->stream[0],
                       (const char*)m->stage_full, (char*)src, (long long)rows, (long long)m->ncrms,
                       (long long)m->sl0[g], (long long)m->nloc[g], m->eb, 1);
  }
  M_HIP(hipGetLastError());
  M_HIP(hipMemcpyAsync(host, m->stage_full, rows * (size_t)m->ncrms * eb, hipMemcpyDeviceToHost, m->stream[0]));
  M_TRACE("gather: unpack + copy out queued");
  const int rs = sync_all(m);
  M_TRACE("gather: synchronised");
  return rs;
}

}  // namespace

void mpdata_shard_range(int64_t ncrms, int ngpus, int g, int64_t* sl0, int64_t* nloc) {
  // contiguous blocks, the remainder spread over the low ranks (= shard.partition of the Python face)
  const int64_t base = ncrms / ngpus, rem = ncrms % ngpus;
  if (nloc) *nloc = base + (g < rem ? 1 : 0);
  if (sl0) *sl0 = (int64_t)g * base + (g < rem ? g : rem);
}

int mpdata_multi_create(int64_t ncrms, int nx, int nz, int ntracers, int ngpus, const int* devices, int eb,
                        mpdata_multi** out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
  if (ngpus < 1 || ngpus > 64) return mpdata_internal_set_err(MPDATA_EINVAL, "ngpus=%d (need 1..64)", ngpus);
  if (ncrms < ngpus) return mpdata_internal_set_err(MPDATA_EINVAL, "ncrms=%lld < ngpus=%d", (long long)ncrms, ngpus);
  mpdata_multi* m = new mpdata_multi();
  m->ncrms = ncrms; m->nx = nx; m->nz = nz; m->ntracers = ntracers; m->eb = eb; m->ngpus = ngpus;
  const char* x = getenv("MPDATA_MULTI_XFER");
  m->xfer = (x && !strcmp(x, "p2p")) ? XFER_P2P : (x && !strcmp(x, "direct")) ? XFER_DIRECT : XFER_RCCL;
  bool distinct = true;
  for (int g = 0; g < ngpus; ++g) {
    const int d = devices ? devices[g] : g;
    if (d < 0 || d >= ndev) {
      delete m;
      return mpdata_internal_set_err(MPDATA_EINVAL, "device %d of the multi-GPU plan does not exist (%d visible)", d, ndev);
    }
    for (int h = 0; h < g; ++h) distinct = distinct && m->dev[h] != d;
    m->dev.push_back(d);
  }
  // RCCL cannot put two ranks on one device: a repeated device (tests on a one-GPU box) falls
  // back to peer copies
  if (!distinct && m->xfer == XFER_RCCL) m->xfer = XFER_P2P;
  m->sl0.resize(ngpus); m->nloc.resize(ngpus); m->sub.assign(ngpus, nullptr); m->stream.assign(ngpus, nullptr);
  m->rb.assign(ngpus, nullptr); m->pk.assign(ngpus, nullptr);
  for (int which = 0; which < 7; ++which) m->max_rows = rows_of(m, which) > m->max_rows ? rows_of(m, which) : m->max_rows;
  int rc = 0;
  int prev = 0;
  (void)hipGetDevice(&prev);
  auto fail = [&](int code) {
    (void)hipSetDevice(prev);
    mpdata_multi_destroy(m);
    return code;
  };
  for (int g = 0; g < ngpus && !rc; ++g) {
    mpdata_shard_range(ncrms, ngpus, g, &m->sl0[g], &m->nloc[g]);
    hipError_t e = hipSetDevice(m->dev[g]);
    if (e == hipSuccess) e = hipStreamCreate(&m->stream[g]);
    if (e == hipSuccess) e = hipMalloc(&m->rb[g], m->max_rows * (size_t)m->nloc[g] * eb);
    if (e == hipSuccess && g > 0 && m->xfer != XFER_DIRECT) {
      e = hipSetDevice(m->dev[0]);
      if (e == hipSuccess) e = hipMalloc(&m->pk[g], m->max_rows * (size_t)m->nloc[g] * eb);
      if (e == hipSuccess) e = hipSetDevice(m->dev[g]);
    }
    if (e != hipSuccess) return fail(mpdata_internal_set_err((int)e, "multi-GPU plan, device %d: %s", m->dev[g], hipGetErrorString(e)));
    rc = eb == 8 ? mpdata_plan_create(m->nloc[g], nx, nz, ntracers, &m->sub[g])
                 : mpdata_plan_create_f32(m->nloc[g], nx, nz, ntracers, &m->sub[g]);
    if (!rc) rc = mpdata_plan_set_stream(m->sub[g], (void*)m->stream[g]);
  }
  if (rc) return fail(rc);
  if (m->xfer != XFER_DIRECT) {
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = hipMalloc(&m->stage_full, m->max_rows * (size_t)ncrms * eb);
    if (e != hipSuccess) return fail(mpdata_internal_set_err((int)e, "multi-GPU plan, root staging: %s", hipGetErrorString(e)));
  }
  if (m->xfer == XFER_RCCL) {
    m->comm.assign(ngpus, nullptr);
    ncclResult_t r = ncclCommInitAll(m->comm.data(), ngpus, m->dev.data());
    if (r != ncclSuccess) return fail(mpdata_internal_set_err(MPDATA_ECOMM, "ncclCommInitAll(%d): %s", ngpus, ncclGetErrorString(r)));
    m->comm_ok = true;
  } else if (m->xfer == XFER_P2P) {
    for (int g = 1; g < ngpus; ++g) {  // best effort: without peer access the copies are staged
      if (m->dev[g] == m->dev[0]) continue;
      (void)hipSetDevice(m->dev[g]); (void)hipDeviceEnablePeerAccess(m->dev[0], 0);
      (void)hipSetDevice(m->dev[0]); (void)hipDeviceEnablePeerAccess(m->dev[g], 0);
      (void)hipGetLastError();
    }
  }
  (void)hipSetDevice(prev);
  *out = m;
  return 0;
}

int mpdata_multi_upload(mpdata_multi* m, const void* f, const void* u, const void* w, const void* rho,
                        const void* rhow, const void* adz, const void* flux) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  const double t0 = now_s();
  const size_t eb = (size_t)m->eb;
  const size_t f1 = (size_t)m->ncrms * rows_of(m, 0) * eb, x1 = (size_t)m->ncrms * rows_of(m, 6) * eb;
  int rc = 0;
  if (!rc) rc = scatter_array(m, 1, u, 0);
  if (!rc) rc = scatter_array(m, 2, w, 0);
  if (!rc) rc = scatter_array(m, 3, rho, 0);
  if (!rc) rc = scatter_array(m, 4, rhow, 0);
  if (!rc) rc = scatter_array(m, 5, adz, 0);
  for (int t = 0; t < m->ntracers && !rc; ++t) {
    rc = scatter_array(m, 0, (const char*)f + (size_t)t * f1, t);
    if (!rc && flux) rc = scatter_array(m, 6, (const char*)flux + (size_t)t * x1, t);
  }
  if (!rc && !flux) {  // flux(:,nz) is never written by the routine: define it
    std::vector<char> z(x1, 0);
    for (int t = 0; t < m->ntracers && !rc; ++t) rc = scatter_array(m, 6, z.data(), t);
  }
  m->scatter_s = now_s() - t0;
  size_t rows = rows_of(m, 1) + rows_of(m, 2) + rows_of(m, 3) + rows_of(m, 4) + rows_of(m, 5) +
                (size_t)m->ntracers * (rows_of(m, 0) + rows_of(m, 6));
  m->scatter_bytes_peer = (int64_t)(rows * (size_t)m->nloc[m->ngpus - 1] * eb);
  (void)hipSetDevice(prev);
  return rc;
}

int mpdata_multi_run(mpdata_multi* m, int first, int count) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  int rc = 0;
  for (int g = 0; g < m->ngpus && !rc; ++g) rc = mpdata_plan_run_tracers(m->sub[g], first, count);
  (void)hipSetDevice(prev);
  return rc;
}

int mpdata_multi_sync(mpdata_multi* m) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  const int rc = sync_all(m);
  (void)hipSetDevice(prev);
  return rc;
}

int mpdata_multi_download(mpdata_multi* m, void* f, void* flux) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  const double t0 = now_s();
  const size_t eb = (size_t)m->eb;
  const size_t f1 = (size_t)m->ncrms * rows_of(m, 0) * eb, x1 = (size_t)m->ncrms * rows_of(m, 6) * eb;
  int rc = 0;
  for (int t = 0; t < m->ntracers && !rc; ++t) {
    if (f) rc = gather_array(m, 0, (char*)f + (size_t)t * f1, t);
    if (!rc && flux) rc = gather_array(m, 6, (char*)flux + (size_t)t * x1, t);
  }
  m->gather_s = now_s() - t0;
  m->gather_bytes_peer = (int64_t)((size_t)m->ntracers * ((f ? rows_of(m, 0) : 0) + (flux ? rows_of(m, 6) : 0)) *
                                   (size_t)m->nloc[m->ngpus - 1] * eb);
  (void)hipSetDevice(prev);
  return rc;
}

int mpdata_multi_last_kernel_ms(mpdata_multi* m, double* ms) {
  double mx = 0;
  for (int g = 0; g < m->ngpus; ++g) {
    double t = 0;
    const int rc = mpdata_plan_last_kernel_ms(m->sub[g], &t);
    if (rc) return rc;
    mx = t > mx ? t : mx;
  }
  *ms = mx;
  return 0;
}

int mpdata_multi_info(const mpdata_multi* m, int g, int* device, int64_t* sl0, int64_t* nloc) {
  if (g < 0 || g >= m->ngpus) return mpdata_internal_set_err(MPDATA_EINVAL, "shard %d of %d", g, m->ngpus);
  if (device) *device = m->dev[g];
  if (sl0) *sl0 = m->sl0[g];
  if (nloc) *nloc = m->nloc[g];
  return 0;
}

void mpdata_multi_stats(const mpdata_multi* m, double* scatter_s, double* gather_s, int64_t* scatter_bytes_peer,
                        int64_t* gather_bytes_peer, int* xfer) {
  if (scatter_s) *scatter_s = m->scatter_s;
  if (gather_s) *gather_s = m->gather_s;
  if (scatter_bytes_peer) *scatter_bytes_peer = m->scatter_bytes_peer;
  if (gather_bytes_peer) *gather_bytes_peer = m->gather_bytes_peer;
  if (xfer) *xfer = (int)m->xfer;
}

int mpdata_multi_ngpus(const mpdata_multi* m) { return m->ngpus; }
mpdata_plan* mpdata_multi_sub(const mpdata_multi* m, int g) { return (g >= 0 && g < m->ngpus) ? m->sub[g] : nullptr; }

int mpdata_multi_destroy(mpdata_multi* m) {
  if (!m) return 0;
  M_TRACE("destroy (%d GPUs)", m->ngpus);
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (m->comm_ok)
    for (ncclComm_t c : m->comm)
      if (c) (void)ncclCommDestroy(c);
  for (int g = 0; g < (int)m->sub.size(); ++g) {
    (void)hipSetDevice(m->dev[g]);
    if (m->sub[g]) (void)mpdata_plan_destroy(m->sub[g]);
    if (m->rb[g]) (void)hipFree(m->rb[g]);
    if (m->stream[g]) (void)hipStreamDestroy(m->stream[g]);
  }
  if (!m->dev.empty()) (void)hipSetDevice(m->dev[0]);
  for (void* p : m->pk)
    if (p) (void)hipFree(p);
  if (m->stage_full) (void)hipFree(m->stage_full);
  (void)hipSetDevice(prev);
  delete m;
  return 0;
}
