// mpdata_diag.hip -- diagnostic: what this GPU sustains for the routine's own traffic mix
// (three arrays read, one written, in place) with the friendliest possible access pattern: every
// workgroup moves one contiguous, aligned 16-KiB piece of each array with 16-byte loads / stores,
// all loads issued before the first store.  The arrays hold pseudo-random values (on all-zero
// data the chip clocks higher and the figure flatters: DESIGN.md section 4.1).  bench.py prints it beside the 8 TB/s specification
// ("also record a measured ... ceiling on the box").  Not on the hot path.
#include <hip/hip_runtime.h>

#include "mpdata_hip.h"
#include "mpdata_multi.h"

namespace {
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NT>
__global__ void __launch_bounds__(256) stream_3r1w(const d2* a, const d2* b, const d2* c, d2* o) {
  const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
  d2 x[4], y[4], z[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = base + j * 256;
    if (NT) { x[j] = __builtin_nontemporal_load(a + i); y[j] = __builtin_nontemporal_load(b + i); z[j] = __builtin_nontemporal_load(c + i); }
    else { x[j] = a[i]; y[j] = b[i]; z[j] = c[i]; }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = base + j * 256;
    const d2 r = (x[j] + y[j] + z[j]) * 0.3125;   // (in place over many launches: stays O(1))
    if (NT) __builtin_nontemporal_store(r, o + i);
    else o[i] = r;
  }
}
// pseudo-random doubles in [-1, 1): splitmix64 of the element index
__global__ void __launch_bounds__(256) fill_random(double* a, size_t n, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    a[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
  }
}
}  // namespace

extern "C" int mpdata_diag_stream_3r1w(int64_t bytes_per_array, int nontemporal, int iters, double* gbs) {
  if (bytes_per_array < (1 << 20) || iters < 1 || !gbs) return mpdata_internal_set_err(MPDATA_EINVAL, "bad argument to mpdata_diag_stream_3r1w");
  const size_t n16k = (size_t)bytes_per_array / 16384;   // whole 16-KiB pieces
  const size_t bytes = n16k * 16384;
  char* buf[3] = {nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipMalloc(&buf[i], bytes);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  for (int i = 0; i < 3 && e == hipSuccess; ++i) {
    hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (double*)buf[i], bytes / 8, 1234567ull + (unsigned long long)i);
    e = hipGetLastError();
  }
  float ms = 0.f;
  if (e == hipSuccess) {
    auto launch = [&]() {   // in place: the output is the first input, as f is in the routine
      if (nontemporal) hipLaunchKernelGGL(stream_3r1w<1>, dim3((unsigned)n16k), dim3(256), 0, 0, (const d2*)buf[0], (const d2*)buf[1], (const d2*)buf[2], (d2*)buf[0]);
      else hipLaunchKernelGGL(stream_3r1w<0>, dim3((unsigned)n16k), dim3(256), 0, 0, (const d2*)buf[0], (const d2*)buf[1], (const d2*)buf[2], (d2*)buf[0]);
    };
    for (int r = 0; r < 20; ++r) launch();
    e = hipEventRecord(e0, 0);
    for (int r = 0; r < iters; ++r) launch();
    if (e == hipSuccess) e = hipEventRecord(e1, 0);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  }
  for (int i = 0; i < 3; ++i)
    if (buf[i]) (void)hipFree(buf[i]);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess) return mpdata_internal_set_err((int)e, "mpdata_diag_stream_3r1w: %s", hipGetErrorString(e));
  *gbs = 4.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;
  return 0;
}
