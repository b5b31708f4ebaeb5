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
//   arrays on the ROOT GPU (mpdata_plan_import_device / _export_device on the multi-GPU plan):
//     scatter: pack kernel (one contiguous block per peer) --ncclSend / ncclRecv in ONE group (RCCL
//              over xGMI: every peer has its own link to the root, no ring)--> peer --layout
//              conversion--> its plan;  gather: the reverse (export, ncclSend to the root, unpack).
//   HOST arrays (upload / download): every GPU copies its own slab from / to the host array
//     ("direct": 8 PCIe links in parallel, nothing funnels through the root's single link).
// Nothing synchronises the host between the arrays of one upload / download: the buffers every
// array reuses are protected by stream order (and events for the peer-copy transport), the host
// waits once at the end.
//
// Because `sl` is the FASTEST array axis, a block is a strided slab of every array (rows of
// nloc*8 bytes at a pitch of ncrms*8): hence the pack / unpack kernels on the root.
// One host thread drives all devices (hipSetDevice per device, one stream each; the single
// process owns all communicators, ncclCommInitAll).  MPDATA_MULTI_XFER selects the transport:
//   rccl   host arrays go through the root as well (H2D of array n+1 on a copy stream into the other
//          half of a double staging buffer while array n is packed and sent);
//   p2p    hipMemcpyPeerAsync root <-> peer instead of ncclSend/ncclRecv (also xGMI), both origins;
//   direct the default for host arrays, spelled out.
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
  Xfer xfer_host;                 // transport for HOST-origin arrays (upload / download): default direct
  Xfer xfer_dev;                  // transport for arrays that live on the root GPU (scatter / gather _device): default rccl
  bool xfer_forced = false;       // MPDATA_MULTI_XFER was given
  Xfer last_xfer = XFER_DIRECT;   // what the last upload / scatter used (mpdata_plan_transfer_stats)
  std::vector<int> dev;
  std::vector<int64_t> sl0, nloc;
  std::vector<mpdata_plan*> sub;
  std::vector<hipStream_t> stream;
  std::vector<void*> rb;          // per device: one block of one array (tracer), contiguous
  std::vector<void*> pk;          // root: packed block per peer (index g; pk[0] unused)
  void* stage_full[2] = {nullptr, nullptr};   // root: one host array (tracer) at full width, double-buffered
  hipStream_t copy_stream = nullptr;          // root: H2D of the next array while the previous one is packed / sent
  hipEvent_t stage_filled[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};
  int stage_turn = 0;
  // p2p transport: who may touch pk[g] next.  scatter: sc_packed (root's stream: block g packed) ->
  // peer copies -> sc_taken (peer's stream) -> root packs the next array.  gather: ga_sent (peer's
  // stream: block copied into pk[g]) -> root unpacks -> ga_unpacked (root's stream) -> next copy.
  // An event is created on the device whose stream records it.
  std::vector<hipEvent_t> sc_packed, sc_taken, ga_sent, ga_unpacked;
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
#define M_TRY(expr)            \
  do {                         \
    int rc_ = (expr);          \
    if (rc_) return rc_;       \
  } while (0)

// MPDATA_MULTI_SYNC=1: synchronise every stream after EVERY array of a scatter / gather (the round-2
// behaviour) instead of once per transfer.  A fallback for bring-up on real multi-GPU nodes: the default
// ordering -- per-array stream order and events, no host synchronisation in between -- has run against the
// recording RCCL stand-in, with peer copies on one device and with one real RCCL rank, never yet on xGMI.
bool sync_each_array() {
  static const bool on = getenv("MPDATA_MULTI_SYNC") != nullptr;
  return on;
}
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
  int rc = 0;
  for (int g = 0; g < m->ngpus; ++g) {   // (every stream, also after an error: nothing may stay in flight)
    hipError_t e = hipSetDevice(m->dev[g]);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream[g]);
    if (e != hipSuccess && !rc) rc = mpdata_internal_set_err((int)e, "synchronising device %d: %s", m->dev[g], hipGetErrorString(e));
  }
  if (m->copy_stream) {
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(m->copy_stream);
    if (e != hipSuccess && !rc) rc = mpdata_internal_set_err((int)e, "synchronising the root copy stream: %s", hipGetErrorString(e));
  }
  return rc;
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

void pack_launch(mpdata_multi* m, const void* full, void* shard, size_t rows, int g, int unpack) {
  hipLaunchKernelGGL(slab_kernel, dim3(grid_for((long long)rows * m->nloc[g])), dim3(256), 0, m->stream[0],
                     (const char*)full, (char*)shard, (long long)rows, (long long)m->ncrms, (long long)m->sl0[g],
                     (long long)m->nloc[g], m->eb, unpack);
}

// The grouped exchange between the root and every peer: root -> peers (to_peers) or peers -> root.
// ONE ncclGroupStart / ncclGroupEnd around the G-1 send / recv pairs -- one host thread drives all
// ranks, so an ungrouped blocking send would never meet its recv.  On an error inside the group the
// group is still closed (an open group would swallow every later RCCL call of this thread).
int rccl_exchange(mpdata_multi* m, size_t rows, bool to_peers) {
  const ncclDataType_t dt = m->eb == 8 ? ncclDouble : ncclFloat;
  ncclResult_t bad = ncclSuccess;
  const char* where = "";
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return mpdata_internal_set_err(MPDATA_ECOMM, "ncclGroupStart: %s", ncclGetErrorString(r));
  for (int g = 1; g < m->ngpus && bad == ncclSuccess; ++g) {
    const size_t n = rows * (size_t)m->nloc[g];
    if (to_peers) {
      r = ncclSend(m->pk[g], n, dt, g, m->comm[0], m->stream[0]);
      if (r != ncclSuccess) { bad = r; where = "ncclSend (root)"; break; }
      r = ncclRecv(m->rb[g], n, dt, 0, m->comm[g], m->stream[g]);
      if (r != ncclSuccess) { bad = r; where = "ncclRecv (peer)"; }
    } else {
      r = ncclSend(m->rb[g], n, dt, 0, m->comm[g], m->stream[g]);
      if (r != ncclSuccess) { bad = r; where = "ncclSend (peer)"; break; }
      r = ncclRecv(m->pk[g], n, dt, g, m->comm[0], m->stream[0]);
      if (r != ncclSuccess) { bad = r; where = "ncclRecv (root)"; }
    }
  }
  r = ncclGroupEnd();
  if (bad != ncclSuccess) {
    (void)sync_all(m);
    return mpdata_internal_set_err(MPDATA_ECOMM, "%s: %s", where, ncclGetErrorString(bad));
  }
  if (r != ncclSuccess) {
    (void)sync_all(m);
    return mpdata_internal_set_err(MPDATA_ECOMM, "ncclGroupEnd: %s", ncclGetErrorString(r));
  }
  return 0;
}

Xfer transport_for(const mpdata_multi* m, bool host_origin) { return host_origin ? m->xfer_host : m->xfer_dev; }

// One array (one tracer of f / flux) to every GPU's plan.  `src` is a HOST array (host_origin) or an
// array on the ROOT GPU, full width either way.  Everything is only QUEUED here -- the buffers every
// array reuses (rb, pk, stage_full) are protected by stream order and events, not by host
// synchronisation --; the caller synchronises once per upload.
int scatter_array(mpdata_multi* m, int which, const void* src, bool host_origin, int tracer) {
  const size_t rows = rows_of(m, which), eb = (size_t)m->eb;
  const int G = m->ngpus;
  const Xfer xf = transport_for(m, host_origin);
  M_TRACE("scatter array %d tracer %d transport %d host=%d", which, tracer, (int)xf, (int)host_origin);
  if (host_origin && xf == XFER_DIRECT) {
    // every GPU pulls its own slab over its own PCIe link (the library does not page-lock caller
    // memory, see mpdata_hostcall.hip: the runtime stages the strided copies)
    for (int g = 0; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipMemcpy2DAsync(m->rb[g], (size_t)m->nloc[g] * eb, (const char*)src + (size_t)m->sl0[g] * eb,
                             (size_t)m->ncrms * eb, (size_t)m->nloc[g] * eb, rows, hipMemcpyHostToDevice, m->stream[g]));
      M_TRY(import_block(m, g, which, tracer));
    }
    if (sync_each_array()) M_TRY(sync_all(m));
    return 0;
  }
  // through the root: (host arrays: full-width copy in, on the copy stream, into the staging buffer
  // whose previous contents have been packed), then one pack per GPU on the root's stream
  M_HIP(hipSetDevice(m->dev[0]));
  const void* full = src;
  if (host_origin) {
    const int t = m->stage_turn;
    m->stage_turn ^= 1;
    M_HIP(hipStreamWaitEvent(m->copy_stream, m->stage_free[t], 0));
    M_HIP(hipMemcpyAsync(m->stage_full[t], src, rows * (size_t)m->ncrms * eb, hipMemcpyHostToDevice, m->copy_stream));
    M_HIP(hipEventRecord(m->stage_filled[t], m->copy_stream));
    M_HIP(hipStreamWaitEvent(m->stream[0], m->stage_filled[t], 0));
    full = m->stage_full[t];
    for (int g = 0; g < G; ++g) {
      if (g > 0 && xf == XFER_P2P) M_HIP(hipStreamWaitEvent(m->stream[0], m->sc_taken[g], 0));
      pack_launch(m, full, g == 0 ? m->rb[0] : m->pk[g], rows, g, 0);
    }
    M_HIP(hipGetLastError());
    M_HIP(hipEventRecord(m->stage_free[t], m->stream[0]));
  } else {
    for (int g = 0; g < G; ++g) {
      if (g > 0 && xf == XFER_P2P) M_HIP(hipStreamWaitEvent(m->stream[0], m->sc_taken[g], 0));
      pack_launch(m, full, g == 0 ? m->rb[0] : m->pk[g], rows, g, 0);
    }
    M_HIP(hipGetLastError());
  }
  if (xf == XFER_RCCL) {
    M_TRY(rccl_exchange(m, rows, true));
  } else {   // peer copies: each peer's stream waits for the pack, the root's next pack for the copy
    for (int g = 1; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[0]));
      M_HIP(hipEventRecord(m->sc_packed[g], m->stream[0]));
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipStreamWaitEvent(m->stream[g], m->sc_packed[g], 0));
      M_HIP(hipMemcpyPeerAsync(m->rb[g], m->dev[g], m->pk[g], m->dev[0], rows * (size_t)m->nloc[g] * eb, m->stream[g]));
      M_HIP(hipEventRecord(m->sc_taken[g], m->stream[g]));
    }
  }
  for (int g = 0; g < G; ++g) {
    M_HIP(hipSetDevice(m->dev[g]));
    M_TRY(import_block(m, g, which, tracer));
  }
  if (sync_each_array()) M_TRY(sync_all(m));
  return 0;
}

// f (which = 0) or flux (6) of one tracer from every GPU's plan to a host array or to an array on the
// root GPU.  Queued only, like scatter_array.
int gather_array(mpdata_multi* m, int which, void* dst, bool host_origin, int tracer) {
  const size_t rows = rows_of(m, which), eb = (size_t)m->eb;
  const int G = m->ngpus;
  const Xfer xf = transport_for(m, host_origin);
  M_TRACE("gather array %d tracer %d transport %d host=%d", which, tracer, (int)xf, (int)host_origin);
  for (int g = 0; g < G; ++g) {
    M_HIP(hipSetDevice(m->dev[g]));
    M_TRY(mpdata_plan_export_device(m->sub[g], which == 0 ? m->rb[g] : nullptr, which == 6 ? m->rb[g] : nullptr, tracer, 1));
  }
  if (host_origin && xf == XFER_DIRECT) {
    for (int g = 0; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipMemcpy2DAsync((char*)dst + (size_t)m->sl0[g] * eb, (size_t)m->ncrms * eb, m->rb[g], (size_t)m->nloc[g] * eb,
                             (size_t)m->nloc[g] * eb, rows, hipMemcpyDeviceToHost, m->stream[g]));
    }
    if (sync_each_array()) M_TRY(sync_all(m));
    return 0;
  }
  if (xf == XFER_RCCL) {
    M_TRY(rccl_exchange(m, rows, false));
  } else {
    for (int g = 1; g < G; ++g) {
      M_HIP(hipSetDevice(m->dev[g]));
      M_HIP(hipStreamWaitEvent(m->stream[g], m->ga_unpacked[g], 0));   // the root has unpacked pk[g]'s previous contents
      M_HIP(hipMemcpyPeerAsync(m->pk[g], m->dev[0], m->rb[g], m->dev[g], rows * (size_t)m->nloc[g] * eb, m->stream[g]));
      M_HIP(hipEventRecord(m->ga_sent[g], m->stream[g]));
      M_HIP(hipSetDevice(m->dev[0]));
      M_HIP(hipStreamWaitEvent(m->stream[0], m->ga_sent[g], 0));
    }
  }
  M_HIP(hipSetDevice(m->dev[0]));
  void* full = host_origin ? m->stage_full[0] : dst;   // (stage_full[0]: unpack and copy-out are ordered on the root's stream)
  for (int g = 0; g < G; ++g) pack_launch(m, full, g == 0 ? m->rb[0] : m->pk[g], rows, g, 1);
  M_HIP(hipGetLastError());
  if (xf == XFER_P2P)
    for (int g = 1; g < G; ++g) M_HIP(hipEventRecord(m->ga_unpacked[g], m->stream[0]));
  if (host_origin) M_HIP(hipMemcpyAsync(dst, full, rows * (size_t)m->ncrms * eb, hipMemcpyDeviceToHost, m->stream[0]));
  if (sync_each_array()) M_TRY(sync_all(m));
  return 0;
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
  // Transport by data origin.  HOST arrays (upload / download): every GPU copies its slab itself
  // ("direct": eight PCIe links in parallel, no root in the way).  Arrays on the ROOT GPU
  // (scatter / gather of device-resident state): pack + RCCL send / recv, one xGMI link per peer.
  // MPDATA_MULTI_XFER = rccl | p2p | direct overrides both (direct: host arrays only).
  const char* x = getenv("MPDATA_MULTI_XFER");
  m->xfer_host = XFER_DIRECT;
  m->xfer_dev = XFER_RCCL;
  if (x && !strcmp(x, "rccl")) { m->xfer_host = XFER_RCCL; m->xfer_forced = true; }
  else if (x && !strcmp(x, "p2p")) { m->xfer_host = m->xfer_dev = XFER_P2P; m->xfer_forced = true; }
  else if (x && !strcmp(x, "direct")) { m->xfer_forced = true; }
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
  // back to peer copies.  (MPDATA_MULTI_FORCE_RCCL: the test build that links a recording stand-in
  // for RCCL keeps the RCCL branch, tests/stubs/fake_rccl.cpp.)
  if (!distinct && getenv("MPDATA_MULTI_FORCE_RCCL") == nullptr) {
    if (m->xfer_host == XFER_RCCL) m->xfer_host = XFER_P2P;
    if (m->xfer_dev == XFER_RCCL) m->xfer_dev = XFER_P2P;
  }
  m->last_xfer = m->xfer_host;
  m->sl0.resize(ngpus); m->nloc.resize(ngpus); m->sub.assign(ngpus, nullptr); m->stream.assign(ngpus, nullptr);
  m->rb.assign(ngpus, nullptr); m->pk.assign(ngpus, nullptr);
  m->sc_packed.assign(ngpus, nullptr); m->sc_taken.assign(ngpus, nullptr);
  m->ga_sent.assign(ngpus, nullptr); m->ga_unpacked.assign(ngpus, nullptr);
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
    if (e == hipSuccess && g > 0) e = hipEventCreateWithFlags(&m->sc_taken[g], hipEventDisableTiming);
    if (e == hipSuccess && g > 0) e = hipEventCreateWithFlags(&m->ga_sent[g], hipEventDisableTiming);
    if (e == hipSuccess && g > 0) {
      e = hipSetDevice(m->dev[0]);
      if (e == hipSuccess) e = hipMalloc(&m->pk[g], m->max_rows * (size_t)m->nloc[g] * eb);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->sc_packed[g], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ga_unpacked[g], hipEventDisableTiming);
      if (e == hipSuccess) e = hipSetDevice(m->dev[g]);
    }
    if (e != hipSuccess) return fail(mpdata_internal_set_err((int)e, "multi-GPU plan, device %d: %s", m->dev[g], hipGetErrorString(e)));
    rc = eb == 8 ? mpdata_plan_create(m->nloc[g], nx, nz, ntracers, &m->sub[g])
                 : mpdata_plan_create_f32(m->nloc[g], nx, nz, ntracers, &m->sub[g]);
    if (!rc) rc = mpdata_plan_set_stream(m->sub[g], (void*)m->stream[g]);
  }
  if (rc) return fail(rc);
  if (m->xfer_host != XFER_DIRECT) {   // host arrays travel through the root: full-width staging, double-buffered
    hipError_t e = hipSetDevice(m->dev[0]);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking);
    for (int t = 0; t < 2 && e == hipSuccess; ++t) {
      e = hipMalloc(&m->stage_full[t], m->max_rows * (size_t)ncrms * eb);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->stage_filled[t], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&m->stage_free[t], hipEventDisableTiming);
    }
    if (e != hipSuccess) return fail(mpdata_internal_set_err((int)e, "multi-GPU plan, root staging: %s", hipGetErrorString(e)));
  }
  if (m->xfer_host == XFER_RCCL || m->xfer_dev == XFER_RCCL) {
    m->comm.assign(ngpus, nullptr);
    ncclResult_t r = ncclCommInitAll(m->comm.data(), ngpus, m->dev.data());
    if (r != ncclSuccess) return fail(mpdata_internal_set_err(MPDATA_ECOMM, "ncclCommInitAll(%d): %s", ngpus, ncclGetErrorString(r)));
    m->comm_ok = true;
    for (int g = 0; g < ngpus; ++g) {   // the communicator agrees with the plan about who is who
      int cnt = 0, rank = -1;
      ncclResult_t r1 = ncclCommCount(m->comm[g], &cnt), r2 = ncclCommUserRank(m->comm[g], &rank);
      if (r1 != ncclSuccess || r2 != ncclSuccess || cnt != ngpus || rank != g)
        return fail(mpdata_internal_set_err(MPDATA_ECOMM, "communicator %d reports rank %d of %d (expected %d of %d)", g, rank, cnt, g, ngpus));
    }
  }
  if (m->xfer_host == XFER_P2P || m->xfer_dev == XFER_P2P) {
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

// All arrays of an upload / a device scatter: queue everything, synchronise ONCE.
static int scatter_all(mpdata_multi* m, const void* f, const void* u, const void* w, const void* rho, const void* rhow,
                       const void* adz, const void* flux, int first, int count, bool host_origin) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  const double t0 = now_s();
  const size_t eb = (size_t)m->eb;
  const size_t f1 = (size_t)m->ncrms * rows_of(m, 0) * eb, x1 = (size_t)m->ncrms * rows_of(m, 6) * eb;
  int rc = 0;
  size_t rows = 0;
  if (!rc && u) { rc = scatter_array(m, 1, u, host_origin, 0); rows += rows_of(m, 1); }
  if (!rc && w) { rc = scatter_array(m, 2, w, host_origin, 0); rows += rows_of(m, 2); }
  if (!rc && rho) { rc = scatter_array(m, 3, rho, host_origin, 0); rows += rows_of(m, 3); }
  if (!rc && rhow) { rc = scatter_array(m, 4, rhow, host_origin, 0); rows += rows_of(m, 4); }
  if (!rc && adz) { rc = scatter_array(m, 5, adz, host_origin, 0); rows += rows_of(m, 5); }
  for (int t = 0; t < count && !rc; ++t) {
    if (f) { rc = scatter_array(m, 0, (const char*)f + (size_t)t * f1, host_origin, first + t); rows += rows_of(m, 0); }
    if (!rc && flux) { rc = scatter_array(m, 6, (const char*)flux + (size_t)t * x1, host_origin, first + t); rows += rows_of(m, 6); }
  }
  const int rs = sync_all(m);   // (also on the error path: nothing may be left in flight)
  if (!rc) rc = rs;
  m->scatter_s = now_s() - t0;
  m->scatter_bytes_peer = (int64_t)(rows * (size_t)m->nloc[m->ngpus - 1] * eb);
  m->last_xfer = transport_for(m, host_origin);
  (void)hipSetDevice(prev);
  return rc;
}
static int gather_all(mpdata_multi* m, void* f, void* flux, int first, int count, bool host_origin) {
  int prev = 0;
  (void)hipGetDevice(&prev);
  const double t0 = now_s();
  const size_t eb = (size_t)m->eb;
  const size_t f1 = (size_t)m->ncrms * rows_of(m, 0) * eb, x1 = (size_t)m->ncrms * rows_of(m, 6) * eb;
  int rc = 0;
  for (int t = 0; t < count && !rc; ++t) {
    if (f) rc = gather_array(m, 0, (char*)f + (size_t)t * f1, host_origin, first + t);
    if (!rc && flux) rc = gather_array(m, 6, (char*)flux + (size_t)t * x1, host_origin, first + t);
  }
  const int rs = sync_all(m);
  if (!rc) rc = rs;
  m->gather_s = now_s() - t0;
  m->gather_bytes_peer = (int64_t)((size_t)count * ((f ? rows_of(m, 0) : 0) + (flux ? rows_of(m, 6) : 0)) *
                                   (size_t)m->nloc[m->ngpus - 1] * eb);
  m->last_xfer = transport_for(m, host_origin);
  (void)hipSetDevice(prev);
  return rc;
}

int mpdata_multi_upload(mpdata_multi* m, const void* f, const void* u, const void* w, const void* rho,
                        const void* rhow, const void* adz, const void* flux) {
  // flux == NULL: zeros for every tracer, on every upload, exactly as a single-GPU plan does it (a
  // memset per GPU instead of a transfer; flux(:,nz) is never written by the routine, reference
  // :541, :624, so whatever an earlier upload left there would otherwise come back)
  if (!flux) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = 0;
    for (int g = 0; g < m->ngpus && !rc; ++g) rc = mpdata_plan_zero_flux_internal(m->sub[g]);
    (void)hipSetDevice(prev);
    if (rc) return rc;
  }
  return scatter_all(m, f, u, w, rho, rhow, adz, flux, 0, m->ntracers, true);
}
// reference-layout arrays of the GLOBAL problem that live on the root GPU (device 0 of the plan):
// NULL pointers are skipped; f / flux cover tracers [first, first + count)
// The full-width arrays of a device scatter / gather must live on the ROOT GPU (shard 0's device): the
// pack kernel runs there and dereferences them.  A pointer that HIP knows to belong to another
// device (or to the host) is refused instead of being handed to that kernel.
static int on_root(const mpdata_multi* m, const void* p, const char* name) {
  if (!p) return 0;
  hipPointerAttribute_t a;
  const hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return mpdata_internal_set_err(MPDATA_EINVAL, "%s: not a device pointer HIP knows (%s); the arrays of a device scatter / gather "
                                   "live on the plan's root GPU, device %d", name, hipGetErrorString(e), m->dev[0]);
  }
  if (a.type == hipMemoryTypeManaged) return 0;
  if (a.type != hipMemoryTypeDevice || a.device != m->dev[0])
    return mpdata_internal_set_err(MPDATA_EINVAL, "%s lives on %s %d, but the full-width arrays of a multi-GPU plan must be on its "
                                   "root GPU, device %d (mpdata_plan_shard(plan, 0, &device, ...))", name,
                                   a.type == hipMemoryTypeDevice ? "device" : "the host / memory type", a.type == hipMemoryTypeDevice ? a.device : (int)a.type,
                                   m->dev[0]);
  return 0;
}
int mpdata_multi_scatter_device(mpdata_multi* m, const void* f, const void* u, const void* w, const void* rho,
                                const void* rhow, const void* adz, const void* flux, int first, int count) {
  const void* ps[7] = {f, u, w, rho, rhow, adz, flux};
  static const char* const nm[7] = {"f", "u", "w", "rho", "rhow", "adz", "flux"};
  for (int i = 0; i < 7; ++i) M_TRY(on_root(m, ps[i], nm[i]));
  return scatter_all(m, f, u, w, rho, rhow, adz, flux, first, count, false);
}
int mpdata_multi_gather_device(mpdata_multi* m, void* f, void* flux, int first, int count) {
  M_TRY(on_root(m, f, "f"));
  M_TRY(on_root(m, flux, "flux"));
  return gather_all(m, f, flux, first, count, false);
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
  return gather_all(m, f, flux, 0, m->ntracers, true);
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
  if (xfer) *xfer = (int)m->last_xfer;
}

int mpdata_multi_ngpus(const mpdata_multi* m) { return m->ngpus; }
// ranks the RCCL communicator reports (ncclCommCount of the root's communicator); 0: the plan has none
int mpdata_multi_ranks_seen(const mpdata_multi* m) {
  if (!m->comm_ok) return 0;
  int n = 0;
  if (ncclCommCount(m->comm[0], &n) != ncclSuccess) return MPDATA_ECOMM;
  return n;
}
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
  for (int t = 0; t < 2; ++t) {
    if (m->stage_full[t]) (void)hipFree(m->stage_full[t]);
    if (m->stage_filled[t]) (void)hipEventDestroy(m->stage_filled[t]);
    if (m->stage_free[t]) (void)hipEventDestroy(m->stage_free[t]);
  }
  if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
  for (auto* v : {&m->sc_packed, &m->sc_taken, &m->ga_sent, &m->ga_unpacked})
    for (hipEvent_t e : *v)
      if (e) (void)hipEventDestroy(e);
  (void)hipSetDevice(prev);
  delete m;
  return 0;
}
