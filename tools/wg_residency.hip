// How many workgroups of W waves share a CU?  Every workgroup spins for SPIN_US on the real-time counter; with G workgroups on
// 256 CUs the launch takes ceil(G / (256 r)) * SPIN_US, so r = workgroups resident per CU falls out of the duration.
// Register use is pinned by touching a high VGPR (v130: more than 128 -> at most 3 waves per SIMD; v60: 8 allowed), LDS by
// the dynamic allocation (never touched: only its size matters here).  Written to find out why workgroups of 4 + 1 waves of the nz > 64 tail form were resident once
// per CU (docs/EXPERIMENTS.md D).   hipcc --offload-arch=gfx950 -O2 -o wg_residency wg_residency.hip && ./wg_residency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int HI>
__global__ void spin_kernel(unsigned long long ticks, int* sink) {
  if (HI) asm volatile("v_mov_b32 v130, 0" ::: "v130");
  else asm volatile("v_mov_b32 v60, 0" ::: "v60");
  unsigned long long t0, t1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  do {
    asm volatile("s_sleep 8\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  } while (t1 - t0 < ticks);
  if (t1 == 0) *sink = 1;   // (never: keeps the loop)
}

int main() {
  const double spin_us = 200.0;
  const unsigned long long ticks = (unsigned long long)(spin_us * 100.0);   // 100 MHz counter
  int* sink;
  hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int waves[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16};
  const int ldsk[] = {0, 9, 28, 37, 46};
  for (int hi = 0; hi < 2; ++hi)
    for (int lk : ldsk) {
      printf("%s LDS %2d KB per workgroup:", hi ? "v130 (<=3 waves/SIMD)" : "v60  (<=8 waves/SIMD)", lk);
      for (int w : waves) {
        if (hi && w > 12) continue;   // (more than 3 waves per SIMD at > 128 registers: not launchable)
        const int G = 256 * 48;
        const size_t lds = (size_t)lk * 1024;
        auto launch = [&]() {
          if (hi) hipLaunchKernelGGL(spin_kernel<1>, dim3(G), dim3(64 * w), lds, 0, ticks, sink);
          else hipLaunchKernelGGL(spin_kernel<0>, dim3(G), dim3(64 * w), lds, 0, ticks, sink);
        };
        launch();
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { printf("  W=%d: not launchable", w); continue; }
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double r = (double)G * spin_us / (ms * 1000.0) / 256.0;
        printf("  W=%d: %.1f WG = %.0f waves", w, r, r * w);
      }
      printf("\n");
    }
  return 0;
}
