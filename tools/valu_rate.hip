// Diagnostic microbenchmark: cycles per wave64 fp64 VALU instruction on gfx950,
// for 1/2/4/8 waves per SIMD.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_IT 2000
template <int OP>
__global__ void k(double* out, unsigned long long* cyc, double a0, double b0) {
  double x0 = a0 + threadIdx.x, x1 = a0 * 2, x2 = a0 * 3, x3 = a0 * 4, x4 = a0 * 5, x5 = a0 * 6, x6 = a0 * 7, x7 = a0 * 8;
  const double b = b0, c = b0 * 0.5; const int ln4 = (threadIdx.x & 63) * 4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < N_IT; ++i) {
#define DO(V)                                                     \
    if (OP == 0) V = V + b;                                       \
    else if (OP == 1) V = V * b;                                  \
    else if (OP == 2) V = __builtin_fma(V, b, c);                 \
    else if (OP == 3) V = __builtin_fmax(V, b);                   \
    else if (OP == 4) V = __builtin_amdgcn_rcp(V);                \
    else if (OP == 5) { int lo = __double2loint(V); lo = __builtin_amdgcn_ds_bpermute(ln4, lo); V = __hiloint2double(__double2hiint(V), lo); } \
    else if (OP == 6) { float f = (float)i; asm volatile("v_mov_b32 %0, %0" : "+v"(f)); }  \
    else if (OP == 7) { asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(x0)); }
    DO(x0) DO(x1) DO(x2) DO(x3) DO(x4) DO(x5) DO(x6) DO(x7)
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name) {
  for (int wpb : {1, 2, 4, 8, 16}) {  // waves per block; one block per CU (grid = 256)
    int threads = 64 * wpb;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * threads * 8); hipMalloc(&cyc, 256 * 8);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001, 0.99991);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, 1.0001, 0.99991);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    // waves per SIMD = wpb/4 (min 1); instructions per wave = N_IT*8
    double per = avg / (N_IT * 8.0);
    printf("%-10s waves/CU=%2d  cycles per instr per wave = %6.2f   => per SIMD-instr = %5.2f\n", name, wpb, per,
           per / (wpb >= 4 ? wpb / 4.0 : 1.0));
    hipFree(out); hipFree(cyc);
  }
}
template <int OP>
void wall(const char* name) {
  // 2048 blocks x 256 threads (8 blocks per CU = 8 waves per SIMD); wall-clock throughput
  const int blocks = 2048, threads = 256;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001, 0.99991);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0001, 0.99991);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double instr = (double)blocks * (threads / 64) * N_IT * 8.0;   // wave-instructions
  double per_simd = instr / 1024.0;                               // per SIMD
  printf("%-10s wall %.3f ms: %.3e wave-instr/s per SIMD => at 2.0 GHz %.2f cycles per wave-instr; lane-ops/s = %.2f T\n", name, ms,
         per_simd / (ms * 1e-3), 2.0e9 / (per_simd / (ms * 1e-3)), instr * 64 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  wall<0>("add_f64"); wall<1>("mul_f64"); wall<2>("fma_f64"); wall<4>("rcp_f64"); wall<5>("bpermute");
  run<0>("add_f64"); run<1>("mul_f64"); run<2>("fma_f64"); run<3>("max_f64"); run<4>("rcp_f64"); run<5>("bpermute");
  return 0;
}
