// Diagnostic: practical HBM streaming rates of this GPU for linear (perfectly coalesced,
// 16 B per lane) access: read-only, copy (1R:1W) and the kernel's mix (3R:1W).
// Arrays of 538 MB each (the size of f at ncrms=65536, nx=32, nz=28).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_read(const d2* a, const d2* b, const d2* c, d2* o, size_t n) {
  d2 acc = {0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i] + b[i] + c[i];
  if (acc.x == 1.2345e300) o[0] = acc;
}
__global__ void __launch_bounds__(256) k_copy(const d2* a, d2* o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i];
}
// in place: o[i] = 1.0000001 * o[i] (what biharmonic_wk_scalar does to qtens); 32 B per lane
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_inplace(d4* o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) o[i] = o[i] * 1.0000001;
}
__global__ void __launch_bounds__(256) k_mix(const d2* a, const d2* b, const d2* c, d2* o, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = a[i] + b[i] + c[i];
}
// one workgroup per contiguous 16-KB piece (4 x 16 B per thread, all loads first); o may alias a
template <int NT>
__global__ void __launch_bounds__(256) k_mix4(const d2* a, const d2* b, const d2* c, d2* o, size_t n) {
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
    if (NT) __builtin_nontemporal_store(x[j] + y[j] + z[j], o + i); else o[i] = x[j] + y[j] + z[j];
  }
}
int main(int argc, char** argv) {
  // argv[1] = 1: stagger the four arrays by 256 B each (different offsets modulo 1 KiB)
  const size_t stag = argc > 1 && argv[1][0] == '1' ? 256 : 0;
  const size_t bytes = 65536ull * 38 * 27 * 8, n = bytes / 16;
  d2 *a, *b, *c, *o;
  char *ra, *rb, *rc, *ro;
  (void)hipMalloc(&ra, bytes + 4096); (void)hipMalloc(&rb, bytes + 4096); (void)hipMalloc(&rc, bytes + 4096); (void)hipMalloc(&ro, bytes + 4096);
  a = (d2*)ra; b = (d2*)(rb + stag); c = (d2*)(rc + 2 * stag); o = (d2*)(ro + 3 * stag);
  printf("stagger %zu B\n", stag);
  (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes); (void)hipMemset(c, 0, bytes); (void)hipMemset(o, 0, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {2048, 8192, 65536}) {
    for (int which = 0; which < 4; ++which) {
      float best = 1e9f, sum = 0; int cnt = 0;
      for (int r = 0; r < 60; ++r) {
        (void)hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, c, o, n);
        if (which == 1) hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, o, n);
        if (which == 2) hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, c, o, n);
        if (which == 3) hipLaunchKernelGGL(k_inplace, dim3(grid), dim3(256), 0, 0, (d4*)o, n / 2);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= 20) { sum += ms; ++cnt; }
        if (r > 0 && ms < best) best = ms;
      }
      const double moved = (double)bytes * (which == 0 ? 3 : which == 1 ? 2 : which == 2 ? 4 : 2);
      printf("grid %6d %-10s best %.3f ms %.2f TB/s   mean(steady) %.3f ms %.2f TB/s\n", grid, which == 0 ? "read 3R" : which == 1 ? "copy 1R:1W" : which == 2 ? "mix 3R:1W" : "in place 1R:1W", best, moved / (best * 1e-3) / 1e12, sum / cnt, moved / (sum / cnt * 1e-3) / 1e12);
    }
  }
  for (int which = 0; which < 4; ++which) {
    float sum = 0; int cnt = 0;
    d2* dst = (which & 1) ? a : o;
    for (int r = 0; r < 60; ++r) {
      (void)hipEventRecord(e0);
      if (which < 2) hipLaunchKernelGGL(k_mix4<0>, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, a, b, c, dst, n);
      else hipLaunchKernelGGL(k_mix4<1>, dim3((unsigned)(n / 1024)), dim3(256), 0, 0, a, b, c, dst, n);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (r >= 20) { sum += ms; ++cnt; }
    }
    printf("mix4 3R:1W one 16-KB piece per workgroup, %s, %s: mean(steady) %.3f ms %.2f TB/s\n", which < 2 ? "plain" : "nontemporal",
           (which & 1) ? "in place (o = a)" : "separate output", sum / cnt, (double)bytes * 4 / (sum / cnt * 1e-3) / 1e12);
  }
  return 0;
}
