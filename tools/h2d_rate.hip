// Diagnostic: host<->device transfer options for the host-array drop-in call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t ncrms = 65536, rows = 1026, C = 8192;   // f: 1026 rows of ncrms doubles (538 MB)
  const size_t bytes = ncrms * rows * 8;
  double* h = (double*)malloc(bytes); memset(h, 1, bytes);
  double* d; hipMalloc(&d, bytes);
  hipStream_t s; hipStreamCreate(&s);
  double t0 = now(); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); double t1 = now();
  printf("pageable 1D H2D          : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now(); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); t1 = now();
  printf("pageable 1D H2D (again)  : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now();
  for (size_t c = 0; c < ncrms; c += C) hipMemcpy2DAsync(d + c * rows, C * 8, h + c, ncrms * 8, C * 8, rows, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s); t1 = now();
  printf("pageable 2D chunks H2D   : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now(); hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault); t1 = now();
  printf("hipHostRegister 538 MB   : %.1f ms (%s)\n", (t1 - t0) * 1e3, hipGetErrorString(e));
  t0 = now(); hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t1 = now();
  printf("registered 1D H2D        : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now();
  for (size_t c = 0; c < ncrms; c += C) hipMemcpy2DAsync(d + c * rows, C * 8, h + c, ncrms * 8, C * 8, rows, hipMemcpyHostToDevice, s);
  hipStreamSynchronize(s); t1 = now();
  printf("registered 2D chunks H2D : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now(); hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); t1 = now();
  printf("registered 1D D2H        : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now(); hipHostUnregister(h); t1 = now();
  printf("hipHostUnregister        : %.1f ms\n", (t1 - t0) * 1e3);
  double* p; t0 = now(); hipHostMalloc(&p, bytes, hipHostMallocDefault); t1 = now();
  printf("hipHostMalloc 538 MB     : %.1f ms\n", (t1 - t0) * 1e3);
  t0 = now(); memcpy(p, h, bytes); t1 = now();
  printf("CPU memcpy 1 thread      : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  t0 = now(); hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); t1 = now();
  printf("pinned 1D H2D            : %.1f ms  %.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
  return 0;
}
