// Diagnostic microbenchmark: the data-movement skeleton of the x-march kernel (LDS-DMA of the
// 3*nzm rows of one column per step into a 4-slot LDS ring, one barrier per step, one row
// store per thread and step) with 4-byte vs 16-byte LDS-DMA and 128/256-byte row segments.
// No arithmetic.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)(unsigned)bytes, 0x00020000);
}
// G = instances per workgroup (row segment = 8*G bytes), DW = bytes per lane of one DMA
template <int G, int DW, int THREADS = 512, bool BLK = false>
__global__ void __launch_bounds__(THREADS) skel(const double* f, const double* u, const double* w, double* fo,
                                            long long ncrms, int nx, int nzm) {
  // gridDim.x < number of groups: persistent workgroups, each sweeps several groups one
  // after the other (all resident workgroups then stay on about the same column)
  for (long long grp = blockIdx.x; grp < ncrms / G; grp += gridDim.x) {
  constexpr int ROWB = G * 8;                 // bytes of a row segment
  constexpr int RPI = 64 * DW / ROWB;         // rows per DMA wave instruction
  constexpr int ARR = 32 * G;                 // doubles per array block (32 rows)
  constexpr int SLOT = 3 * ARR;
  constexpr int NI = 3 * 32 / RPI;            // instructions per column (3 arrays x 32 rows)
  constexpr int NW = THREADS / 64;
  constexpr int PER_WAVE = (NI + NW - 1) / NW;
  __shared__ double lds[4 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long sl0 = grp * G;
  const unsigned colb = BLK ? (unsigned)(nzm * G * 8) : (unsigned)(ncrms * 8);
  const __amdgpu_buffer_rsrc_t rsf = make_rsrc(f, ncrms * 8ll * (nx + 6) * nzm);
  const __amdgpu_buffer_rsrc_t rsu = make_rsrc(u, ncrms * 8ll * (nx + 6) * nzm);
  const __amdgpu_buffer_rsrc_t rsw = make_rsrc(w, ncrms * 8ll * (nx + 6) * nzm);
  const __amdgpu_buffer_rsrc_t rso = make_rsrc(fo, ncrms * 8ll * (nx + 6) * nzm);
  unsigned voff[PER_WAVE]; int arr[PER_WAVE], blk[PER_WAVE];
#pragma unroll
  for (int it = 0; it < PER_WAVE; ++it) {
    int j = wave + NW * it; if (j >= NI) j = NI - 1;
    arr[it] = j / (32 / RPI); blk[it] = j % (32 / RPI);
    const int lanes_per_row = ROWB / DW;
    int row = blk[it] * RPI + lane / lanes_per_row; if (row > nzm - 1) row = nzm - 1;
    // BLK: the arrays are stored [group][column][row][G] (a workgroup's bytes are contiguous)
    voff[it] = BLK ? (unsigned)(((grp * (nx + 6)) * nzm + row) * (long long)G * 8 % 4000000000ll) + (lane % lanes_per_row) * DW
                   : (unsigned)((sl0 + ncrms * (long long)(nx + 6) * row) * 8) + (lane % lanes_per_row) * DW;
  }
  // store mapping: thread -> (row, instance)
  constexpr int NST = (32 * G + THREADS - 1) / THREADS;  // row stores per thread and step
  constexpr int VM = NST + PER_WAVE;          // vector-memory ops per wave and step
  const int t_sl = tid % G;
  int t_row[NST]; unsigned tf[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    t_row[i] = tid / G + i * (THREADS / G);
    const bool act = t_row[i] < nzm;
    tf[i] = !act ? 0xFFFFFFF8u
            : BLK ? (unsigned)((((grp * (nx + 6)) * nzm + t_row[i]) * (long long)G + t_sl) * 8 % 4000000000ll)
                  : (unsigned)((sl0 + t_sl + ncrms * (long long)(nx + 6) * t_row[i]) * 8);
    if (!act) t_row[i] = 0;
  }
  auto dma = [&](int col) __attribute__((always_inline)) {
    const unsigned c = colb * (unsigned)(col < nx + 6 ? col : nx + 5);
    double* slot = lds + (col & 3) * SLOT;
#pragma unroll
    for (int it = 0; it < PER_WAVE; ++it) {
      double* d = slot + arr[it] * ARR + blk[it] * (64 * DW / 8);
      const __amdgpu_buffer_rsrc_t r = arr[it] == 0 ? rsf : (arr[it] == 1 ? rsu : rsw);
      if constexpr (DW == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)d, 16, (int)voff[it], (int)c, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)d, 4, (int)voff[it], (int)c, 0, 0);
    }
  };
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, rso, (int)0xFFFFFFF8u, 0, 0);
    dma(c);
  }
  for (int q = 0; q < nx + 6; ++q) {
    static_assert(VM == 7 || VM == 3 || VM == 14 || VM == 5 || VM == 2, "add the wait for this op count");
    if (VM == 7) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
    else if (VM == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (VM == 14) asm volatile("s_waitcnt vmcnt(28) lgkmcnt(0)" ::: "memory");
    else if (VM == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const double* s = lds + (q & 3) * SLOT;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const double v = s[t_row[i] * G + t_sl];
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rso, (int)tf[i], (int)(colb * (unsigned)q), 0);
    }
    dma(q + 3);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  }
}
template <int G, int DW, int THREADS = 512, bool BLK = false>
void run(const double* f, const double* u, const double* w, double* fo, long long ncrms, int nx, int nzm, int persist = 0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = persist ? persist : (int)(ncrms / G);
  for (int r = 0; r < 60; ++r)  // warm-up: the first ~30 ms after idle run 5-10 % slow
    hipLaunchKernelGGL((skel<G, DW, THREADS, BLK>), dim3(blocks), dim3(THREADS), 0, 0, f, u, w, fo, ncrms, nx, nzm);
  hipEventRecord(e0);
  for (int r = 0; r < 60; ++r) hipLaunchKernelGGL((skel<G, DW, THREADS, BLK>), dim3(blocks), dim3(THREADS), 0, 0, f, u, w, fo, ncrms, nx, nzm);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 60;
  const double bytes = (double)ncrms * nzm * (nx + 6) * 8.0 * 4;  // 3 reads + 1 write
  printf("%s threads %4d blocks %5d G=%2d (%3d-B rows) DMA %2d B/lane: %.3f ms  %.2f TB/s (err %d)\n", BLK ? "BLOCKED" : "       ", THREADS, blocks, G, G * 8, DW, ms, bytes / (ms * 1e-3) / 1e12, (int)hipGetLastError());
}
int main(int argc, char** argv) {
  // argv[1] = 1: stagger the arrays by 256 B each (different offsets modulo 1 KiB)
  const size_t stag = argc > 1 && argv[1][0] == '1' ? 256 : 0;
  const long long ncrms = 65536; const int nx = 32, nzm = 27;
  size_t n = ncrms * (nx + 6) * nzm;
  double *f, *u, *w, *fo;
  char *rf, *ru, *rw, *ro;
  hipMalloc(&rf, n * 8 + 4096); hipMalloc(&ru, n * 8 + 4096); hipMalloc(&rw, n * 8 + 4096); hipMalloc(&ro, n * 8 + 4096);
  f = (double*)rf; u = (double*)(ru + stag); w = (double*)(rw + 2 * stag); fo = (double*)(ro + 3 * stag);
  printf("stagger %zu B\n", stag);
  hipMemset(f, 0, n * 8); hipMemset(u, 0, n * 8); hipMemset(w, 0, n * 8);
  for (int rep = 0; rep < 1; ++rep) {
    run<16, 4>(f, u, w, fo, ncrms, nx, nzm);
    run<16, 16>(f, u, w, fo, ncrms, nx, nzm);
    run<32, 4>(f, u, w, fo, ncrms, nx, nzm);
    run<32, 16>(f, u, w, fo, ncrms, nx, nzm);
    printf("in place (stores go back into f):\n");
    run<16, 4>(f, u, w, f, ncrms, nx, nzm);
    run<32, 4>(f, u, w, f, ncrms, nx, nzm);
    run<32, 16>(f, u, w, f, ncrms, nx, nzm);
    printf("blocked layout [group][column][row][G], separate output and in place:\n");
    run<16, 4, 512, true>(f, u, w, fo, ncrms, nx, nzm);
    run<16, 16, 512, true>(f, u, w, fo, ncrms, nx, nzm);
    run<16, 4, 512, true>(f, u, w, f, ncrms, nx, nzm);
    run<32, 16, 512, true>(f, u, w, f, ncrms, nx, nzm);
    printf("1024-thread workgroups:\n");
    run<32, 4, 1024>(f, u, w, fo, ncrms, nx, nzm);
    run<32, 16, 1024>(f, u, w, fo, ncrms, nx, nzm);
    run<32, 4, 1024>(f, u, w, f, ncrms, nx, nzm);
    printf("persistent:\n");
    run<16, 4>(f, u, w, fo, ncrms, nx, nzm, 512);
    run<16, 4>(f, u, w, fo, ncrms, nx, nzm, 1024);
    run<32, 16>(f, u, w, fo, ncrms, nx, nzm, 256);
    run<32, 16>(f, u, w, fo, ncrms, nx, nzm, 512);
  }
  return 0;
}
