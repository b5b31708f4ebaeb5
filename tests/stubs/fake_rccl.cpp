// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product libmpdata_hip.so).
//
// A recording stand-in for the nine RCCL entry points csrc/mpdata_multi.hip calls, so that the
// library's RCCL branch -- ncclCommInitAll, then per array ONE ncclGroupStart / ncclGroupEnd around
// the ncclSend / ncclRecv pairs between the root and every peer, all driven by one host thread -- can
// be executed and checked where RCCL itself cannot put it: several ranks on ONE device of a one-GPU
// test box.  tests/test_multi_fake_rccl.py links a copy of the product objects against this file
// (csrc/Makefile target ../libmpdata_hip_fakerccl.so) and checks
//   * the results (bitwise those of a single-GPU plan), and
//   * the recorded call sequence: every group pairs each send with a recv of the same count and type
//     on the right communicator, nothing is posted outside a group (one host thread would block for
//     ever in real RCCL), no group is left open.
// Semantics implemented: a send / recv pair of a group becomes, at ncclGroupEnd, a device-to-device
// copy on the RECEIVER's stream ordered after everything queued before it on the SENDER's stream, and
// the sender's stream waits for the copy (its buffer must not be reused earlier) -- the ordering
// guarantees of the real calls.  An unmatched operation at ncclGroupEnd, or an operation outside a
// group, is an error (in real RCCL: a hang).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct ncclComm {
  int rank, nranks, dev;
  bool alive;
};

namespace {
struct Op {
  bool send;
  const void* sbuf;
  void* rbuf;
  size_t count;
  ncclDataType_t dt;
  int peer, rank;
  hipStream_t stream;
  int dev;
};
int g_depth = 0;
std::vector<Op> g_ops;
std::string g_log;      // one line per call
int g_errors = 0;

void logf(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_log += buf;
  g_log += '\n';
}
size_t dt_size(ncclDataType_t dt) { return dt == ncclDouble ? 8 : dt == ncclFloat ? 4 : 0; }
}  // namespace

extern "C" {

// test hooks
const char* fake_rccl_log(void) { return g_log.c_str(); }
void fake_rccl_reset(void) { g_log.clear(); g_errors = 0; }
int fake_rccl_errors(void) { return g_errors; }
int fake_rccl_open_groups(void) { return g_depth; }

ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist) {
  logf("init_all ndev=%d", ndev);
  for (int i = 0; i < ndev; ++i) comm[i] = new ncclComm{i, ndev, devlist ? devlist[i] : i, true};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c || !c->alive) { ++g_errors; logf("ERROR destroy of a dead communicator"); return ncclInvalidArgument; }
  logf("destroy rank=%d", c->rank);
  c->alive = false;
  delete c;
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) {
  return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (fake RCCL)" : "error (fake RCCL)";
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart(void) {
  ++g_depth;
  logf("group_start depth=%d", g_depth);
  return ncclSuccess;
}

static ncclResult_t post(bool send, const void* sbuf, void* rbuf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c,
                         hipStream_t st) {
  logf("%s rank=%d peer=%d count=%zu dtype=%d", send ? "send" : "recv", c->rank, peer, count, (int)dt);
  if (g_depth == 0) {   // one host thread: the matching call would never be reached
    ++g_errors;
    logf("ERROR %s outside a group", send ? "send" : "recv");
    return ncclInvalidUsage;
  }
  if (peer < 0 || peer >= c->nranks || peer == c->rank || dt_size(dt) == 0) {
    ++g_errors;
    logf("ERROR bad peer / type");
    return ncclInvalidArgument;
  }
  g_ops.push_back(Op{send, sbuf, rbuf, count, dt, peer, c->rank, st, c->dev});
  return ncclSuccess;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
  return post(true, buf, nullptr, count, dt, peer, c, st);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t c, hipStream_t st) {
  return post(false, nullptr, buf, count, dt, peer, c, st);
}

ncclResult_t ncclGroupEnd(void) {
  if (g_depth <= 0) { ++g_errors; logf("ERROR group_end without start"); return ncclInvalidUsage; }
  --g_depth;
  logf("group_end depth=%d ops=%zu", g_depth, g_ops.size());
  if (g_depth > 0) return ncclSuccess;
  ncclResult_t res = ncclSuccess;
  std::vector<bool> used(g_ops.size(), false);
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (size_t i = 0; i < g_ops.size(); ++i) {
    if (!g_ops[i].send) continue;
    const Op& s = g_ops[i];
    size_t j = 0;
    for (; j < g_ops.size(); ++j)   // first unmatched recv on the peer that names this rank (FIFO per pair)
      if (!used[j] && !g_ops[j].send && g_ops[j].rank == s.peer && g_ops[j].peer == s.rank) break;
    if (j == g_ops.size()) { ++g_errors; logf("ERROR send %d->%d has no recv (real RCCL: hang)", s.rank, s.peer); res = ncclInvalidUsage; continue; }
    const Op& r = g_ops[j];
    used[i] = used[j] = true;
    if (r.count != s.count || r.dt != s.dt) {
      ++g_errors;
      logf("ERROR %d->%d count/type mismatch: send %zu recv %zu", s.rank, s.peer, s.count, r.count);
      res = ncclInvalidUsage;
      continue;
    }
    hipEvent_t ready = nullptr, done = nullptr;
    hipError_t e = hipSetDevice(s.dev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ready, s.stream);
    if (e == hipSuccess) e = hipSetDevice(r.dev);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamWaitEvent(r.stream, ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(r.rbuf, s.sbuf, s.count * dt_size(s.dt), hipMemcpyDeviceToDevice, r.stream);
    if (e == hipSuccess) e = hipEventRecord(done, r.stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s.stream, done, 0);
    if (ready) (void)hipEventDestroy(ready);   // (destruction is deferred until the event has completed)
    if (done) (void)hipEventDestroy(done);
    if (e != hipSuccess) { ++g_errors; logf("ERROR hip: %s", hipGetErrorString(e)); res = ncclUnhandledCudaError; }
    else logf("matched %d->%d count=%zu", s.rank, s.peer, s.count);
  }
  for (size_t j = 0; j < g_ops.size(); ++j)
    if (!used[j] && !g_ops[j].send) { ++g_errors; logf("ERROR recv on %d from %d has no send (real RCCL: hang)", g_ops[j].rank, g_ops[j].peer); res = ncclInvalidUsage; }
  g_ops.clear();
  (void)hipSetDevice(prev);
  return res;
}

}  // extern "C"
