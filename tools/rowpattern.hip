// Diagnostic microbenchmark: achievable HBM read bandwidth for the MPDATA row pattern
// (arrays with sl fastest; a workgroup owns SEG bytes of every row and visits, for each
// column q, the 3*nzm rows of that column).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void rows(const double* __restrict__ f, const double* __restrict__ u, const double* __restrict__ w,
                     double* __restrict__ out, long long ncrms, int nx, int nzm, int seg_elems, int do_store,
                     double* __restrict__ fo) {
  const long long sl0 = (long long)blockIdx.x * seg_elems;
  const int per_col = nzm * seg_elems;  // elements of one array per column
  double acc = 0;
  for (int q = 0; q < nx + 4; ++q) {
    for (int e = threadIdx.x; e < per_col; e += blockDim.x) {
      const int row = e / seg_elems, s = e - row * seg_elems;
      const long long sl = sl0 + s;
      const double a = f[sl + ncrms * ((long long)(q + 1) + (long long)(nx + 6) * row)];
      const double b = u[sl + ncrms * ((long long)q + (long long)(nx + 5) * row)];
      const double c = w[sl + ncrms * ((long long)q + (long long)(nx + 4) * row)];
      acc += a + b + c;
      if (do_store) fo[sl + ncrms * ((long long)(q + 1) + (long long)(nx + 6) * row)] = a + b;
    }
  }
  out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const long long ncrms = 65536; const int nx = 32, nzm = 27;
  size_t nf = ncrms * (nx + 6) * nzm, nu = ncrms * (nx + 5) * nzm, nw = ncrms * (nx + 4) * (nzm + 1);
  double *f, *u, *w, *out, *fo;
  hipMalloc(&f, nf * 8); hipMalloc(&u, nu * 8); hipMalloc(&w, nw * 8); hipMalloc(&fo, nf * 8); hipMalloc(&out, 64ull << 20);  // >= max(blocks*threads)*8 = 4096*512*8 = 16 MiB
  hipMemset(f, 0, nf * 8); hipMemset(u, 0, nu * 8); hipMemset(w, 0, nw * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int st = 0; st < 2; ++st)
    for (int seg : {128, 256, 512, 1024, 2048, 8192}) {
      for (int threads : {256, 512}) {
        const int seg_elems = seg / 8; const int blocks = (int)(ncrms / seg_elems);
        hipLaunchKernelGGL(rows, dim3(blocks), dim3(threads), 0, 0, f, u, w, out, ncrms, nx, nzm, seg_elems, st, fo);
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(rows, dim3(blocks), dim3(threads), 0, 0, f, u, w, out, ncrms, nx, nzm, seg_elems, st, fo);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        double bytes = (double)ncrms * nzm * (nx + 4) * 8.0 * (3 + st);
        printf("store=%d seg=%5d B threads=%4d blocks=%6d : %.3f ms  %.2f TB/s\n", st, seg, threads, blocks, ms, bytes / (ms * 1e-3) / 1e12);
      }
    }
  return 0;
}
