// Diagnostic microbenchmark: data-movement skeleton of the "wave-major" private layout
// (DESIGN.md section 4.1): every wave owns SLP adjacent instances and streams its own linear
// image  [tile][column][instance-in-tile][level]  of f, u, w -- no workgroup barrier, no LDS
// transpose -- through a per-wave LDS-DMA ring, and stores the column back from registers.
// Variants: 16-B DMA of column PAIRS vs 4-B DMA of single columns, ring depth, waves per
// workgroup, in-place stores, and register-staged loads (no LDS at all).  No arithmetic.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long long bytes) {
  const long long lim = 0xFFFFFFF0ll;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)(unsigned)(bytes > lim ? lim : bytes), 0x00020000);
}
#define OOB 0xFFFFFFF8u

__global__ void __launch_bounds__(256) fill_random(double* a, size_t n, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    a[i] = (double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
  }
}

// MODE 0: 16-B DMA, one instruction per array and column PAIR (864 B), ring of NS pairs
// MODE 1: 4-B DMA, two instructions per array and column (432 B), ring of NS columns
// MODE 2: register loads (buffer_load_dwordx2 per lane), software pipeline depth NS columns
template <int MODE, int NS, int WPB, int LA = 0, int SA = 0, int AH = NS - 1, int WN = 10, int SB = 0>
__global__ void __launch_bounds__(64 * WPB) skel3(const char* f, const char* u, const char* w, char* fo,
                                                  int ntiles, int ncol, int chunkB, long long tileB, int tmap) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int tile = blockIdx.x * WPB + wave;
  if (tile >= ntiles) return;
  if (tmap == 1) {   // XCD-contiguous: the blocks an XCD receives (b % 8) walk one eighth of the tiles
    const int nb = gridDim.x, per = nb / 8;
    const int b = (blockIdx.x % 8) * per + blockIdx.x / 8;
    tile = b * WPB + wave;
  } else if (tmap == 2) {  // CU-contiguous-ish: 32 consecutive blocks of an XCD get adjacent tiles
    const int b = blockIdx.x;
    const int x = b % 8, j = b / 8;
    tile = ((j / 32) * 8 * 32 + x * 32 + j % 32) * WPB + wave;
  }
  // descriptors relative to this wave's tile (arrays may exceed 4 GiB)
  const __amdgpu_buffer_rsrc_t rf = make_rsrc(f + tile * tileB, tileB);
  const __amdgpu_buffer_rsrc_t ru = make_rsrc(u + tile * tileB, tileB);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(w + tile * tileB, tileB);
  const __amdgpu_buffer_rsrc_t ro = make_rsrc(fo + tile * tileB, tileB);
  const unsigned st_off = lane * 8 < chunkB ? lane * 8 : OOB;
  if constexpr (MODE == 0) {
    __shared__ double lds[WPB * NS * 3 * 128];
    double* my = lds + wave * NS * 3 * 128;
    const unsigned v = lane * 16 < 2 * chunkB ? lane * 16 : OOB;
    const int npair = ncol / 2;
    auto dma = [&](int p) __attribute__((always_inline)) {
      if (p >= npair) p = npair - 1;
      double* d = my + (p % NS) * 3 * 128;
      const unsigned so = (unsigned)(p * 2 * chunkB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)v, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)v, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)v, (int)so, 0, LA);
    };
    for (int p = 0; p < NS - 1; ++p) {
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      dma(p);
    }
    for (int p = 0; p < npair; ++p) {
      static_assert(NS >= 2 && NS <= 4, "");
      if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (NS == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      if (NS == 4) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      const double* s = my + (p % NS) * 3 * 128;
      const int kl = lane * 8 < chunkB ? lane : 0;
      const double a0 = s[kl] + s[128 + kl] + s[256 + kl];
      const double a1 = s[chunkB / 8 + kl] + s[128 + chunkB / 8 + kl] + s[256 + chunkB / 8 + kl];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)st_off, (int)(p * 2 * chunkB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)st_off, (int)((p * 2 + 1) * chunkB), SA);
      dma(p + NS - 1);
    }
  } else if constexpr (MODE == 1) {
    __shared__ double lds[WPB * NS * 3 * 64];
    double* my = lds + wave * NS * 3 * 64;
    const unsigned v0 = lane * 4, v1 = 256 + lane * 4 < chunkB ? 256 + lane * 4 : OOB;
    auto dma = [&](int c) __attribute__((always_inline)) {
      if (c >= ncol) c = ncol - 1;
      double* d = my + (c % NS) * 3 * 64;
      const unsigned so = (unsigned)(c * chunkB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 4, (int)v0, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)(d + 32), 4, (int)v1, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 64), 4, (int)v0, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 96), 4, (int)v1, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 128), 4, (int)v0, (int)so, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 160), 4, (int)v1, (int)so, 0, LA);
    };
    for (int c = 0; c < NS - 1; ++c) {
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      dma(c);
    }
    for (int c = 0; c < ncol; ++c) {
      static_assert(NS >= 3 && NS <= 5, "");
      if (NS == 3) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      if (NS == 4) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      if (NS == 5) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
      const double* s = my + (c % NS) * 3 * 64;
      const int kl = lane * 8 < chunkB ? lane : 0;
      const double a0 = s[kl] + s[64 + kl] + s[128 + kl];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)st_off, (int)(c * chunkB), SA);
      dma(c + NS - 1);
    }
  } else if constexpr (MODE == 3) {
    // split layout: per tile [col][mainB] (line-aligned part of every column chunk) followed by
    // [col][remB]; main part fetched / stored non-temporally (policy LA / SA), remainder cached
    __shared__ double lds[WPB * NS * 3 * 128];
    double* my = lds + wave * NS * 3 * 128;
    const int mainB = chunkB / 128 * 128, remB = chunkB - mainB;
    const unsigned remBase = (unsigned)(ncol * mainB);
    const int l32 = lane & 31;
    const unsigned hi = lane >= 32 ? 1u : 0u;
    const unsigned vA = l32 * 16 < mainB ? (unsigned)(l32 * 16) + hi * mainB : OOB;
    const unsigned vB = (l32 * 16 >= mainB && l32 * 16 < chunkB) ? remBase + (unsigned)(l32 * 16 - mainB) + hi * remB : OOB;
    const unsigned sA = lane * 8 < mainB ? lane * 8 : OOB;
    const unsigned sB = (lane * 8 >= mainB && lane * 8 < chunkB) ? remBase + (unsigned)(lane * 8 - mainB) : OOB;
    const int npair = ncol / 2;
    auto dma = [&](int p) __attribute__((always_inline)) {
      if (p >= npair) p = npair - 1;
      double* d = my + (p % NS) * 3 * 128;
      const unsigned soA = (unsigned)(p * 2 * mainB), soB = (unsigned)(p * 2 * remB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)vA, (int)soA, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)vA, (int)soA, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)vA, (int)soA, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)vB, (int)soB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)vB, (int)soB, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)vB, (int)soB, 0, 0);
    };
    for (int p = 0; p < NS - 1; ++p) {
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      dma(p);
    }
    for (int p = 0; p < npair; ++p) {
      static_assert(NS == 3, "");
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      const double* s = my + (p % NS) * 3 * 128;
      const int kl = lane * 8 < chunkB ? lane : 0;
      const double a0 = s[kl] + s[128 + kl] + s[256 + kl];
      const double a1 = s[64 + kl] + s[128 + 64 + kl] + s[256 + 64 + kl];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sA, (int)(p * 2 * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sB, (int)(p * 2 * remB), 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)sA, (int)((p * 2 + 1) * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)sB, (int)((p * 2 + 1) * remB), 0);
      dma(p + NS - 1);
    }
  } else if constexpr (MODE == 4) {
    // main parts [col][mainB] + one 128-byte line per column PAIR holding both remainders: one
    // fetch instruction per array and pair, every line touched exactly once; stores: main part
    // (policy SA) + remainder (default policy)
    __shared__ double lds[WPB * NS * 3 * 128];
    double* my = lds + wave * NS * 3 * 128;
    const int mainB = chunkB / 128 * 128, remB = chunkB - mainB;
    const int npair = ncol / 2;
    const unsigned remBase = (unsigned)(ncol * mainB);
    const int l32 = lane & 31;
    const unsigned hi = lane >= 32 ? 1u : 0u;
    const unsigned vM = l32 * 16 < mainB ? (unsigned)(l32 * 16) + hi * mainB : OOB;
    const unsigned vR = (l32 * 16 >= mainB && l32 * 16 < chunkB) ? remBase + (unsigned)(l32 * 16 - mainB) + hi * remB : OOB;
    const bool isM = l32 * 16 < mainB;
    const unsigned sA = lane * 8 < mainB ? lane * 8 : OOB;
    const unsigned sB = (lane * 8 >= mainB && lane * 8 < chunkB) ? remBase + (unsigned)(lane * 8 - mainB) : OOB;
    auto dma = [&](int p) __attribute__((always_inline)) {
      if (p >= npair) p = npair - 1;
      double* d = my + (p % NS) * 3 * 128;
      // per-lane byte offset: main lanes advance 2*mainB per pair, remainder lanes 128 per pair
      const unsigned v = isM ? vM + (unsigned)(p * 2 * mainB) : (vR == OOB ? OOB : vR + (unsigned)(p * 128));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)v, 0, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)v, 0, 0, LA);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)v, 0, 0, LA);
    };
    for (int p = 0; p < NS - 1; ++p) {
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      dma(p);
    }
    for (int p = 0; p < npair; ++p) {
      static_assert(NS == 3, "");
      asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      const double* s = my + (p % NS) * 3 * 128;
      const int kl = lane * 8 < chunkB ? lane : 0;
      const double a0 = s[kl] + s[128 + kl] + s[256 + kl];
      const double a1 = s[64 + kl] + s[128 + 64 + kl] + s[256 + 64 + kl];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sA, (int)(p * 2 * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sB, (int)(p * 128), 0);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)sA, (int)((p * 2 + 1) * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)(sB == OOB ? OOB : sB + remB), (int)(p * 128), 0);
      dma(p + NS - 1);
    }
  } else if constexpr (MODE == 5) {
    // MODE 3 with the two fetch instructions of an array executed by disjoint lane sets (EXEC)
    __shared__ double lds[WPB * NS * 3 * 128];
    double* my = lds + wave * NS * 3 * 128;
    const int mainB = chunkB / 128 * 128, remB = chunkB - mainB;
    const unsigned remBase = (unsigned)(ncol * mainB);
    const int l32 = lane & 31;
    const unsigned hi = lane >= 32 ? 1u : 0u;
    const bool isM = l32 * 16 < mainB;
    const unsigned vA = isM ? (unsigned)(l32 * 16) + hi * mainB : OOB;
    const unsigned vB = (l32 * 16 >= mainB && l32 * 16 < chunkB) ? remBase + (unsigned)(l32 * 16 - mainB) + hi * remB : OOB;
    const unsigned sA = lane * 8 < mainB ? lane * 8 : OOB;
    const unsigned sB = (lane * 8 >= mainB && lane * 8 < chunkB) ? remBase + (unsigned)(lane * 8 - mainB) : OOB;
    const int npair = ncol / 2;
    auto dma = [&](int p) __attribute__((always_inline)) {
      if (p >= npair) p = npair - 1;
      double* d = my + (p % NS) * 3 * 128;
      const unsigned soA = (unsigned)(p * 2 * mainB), soB = (unsigned)(p * 2 * remB);
      if (isM) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)vA, (int)soA, 0, LA);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)vA, (int)soA, 0, LA);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)vA, (int)soA, 0, LA);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rf, (lds_ptr_t)d, 16, (int)vB, (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (lds_ptr_t)(d + 128), 16, (int)vB, (int)soB, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d + 256), 16, (int)vB, (int)soB, 0, 0);
      }
    };
    // AH pairs in flight while one is worked on (AH = NS: the slot is refilled right after it was
    // read); WN = operations allowed outstanding at the wait: 10 counts the 4 stores of a pair as
    // well (AH = 2), 6 * (AH - 1) counts only the newer fetches (the safe wait of the kernel)
    for (int p = 0; p < AH; ++p) {
      for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b64(u32x2{0, 0}, ro, (int)OOB, 0, 0);
      dma(p);
    }
    for (int p = 0; p < npair; ++p) {
      if constexpr (WN == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if constexpr (WN == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if constexpr (WN == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if constexpr (WN == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      const double* s = my + (p % NS) * 3 * 128;
      const int kl = lane * 8 < chunkB ? lane : 0;
      const double a0 = s[kl] + s[128 + kl] + s[256 + kl];
      const double a1 = s[64 + kl] + s[128 + 64 + kl] + s[256 + 64 + kl];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sA, (int)(p * 2 * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a0), ro, (int)sB, (int)(p * 2 * remB), SB);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)sA, (int)((p * 2 + 1) * mainB), SA);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a1), ro, (int)sB, (int)((p * 2 + 1) * remB), SB);
      dma(p + AH);
    }
  } else {
    // register-staged: NS columns of (f,u,w) in flight per lane
    double rf_[NS], ru_[NS], rw_[NS];
    auto ld = [&](int c, int j) __attribute__((always_inline)) {
      if (c >= ncol) c = ncol - 1;
      const unsigned so = (unsigned)(c * chunkB);
      rf_[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rf, (int)st_off, (int)so, LA));
      ru_[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(ru, (int)st_off, (int)so, LA));
      rw_[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rw, (int)st_off, (int)so, LA));
    };
#pragma unroll
    for (int j = 0; j < NS; ++j) ld(j, j);
    for (int c0 = 0; c0 < ncol; c0 += NS) {
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const double a = rf_[j] + ru_[j] + rw_[j];
        if (c0 + j < ncol)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), ro, (int)st_off, (int)((c0 + j) * chunkB), SA);
        ld(c0 + j + NS, j);
      }
    }
  }
}

template <int MODE, int NS, int WPB, int LA = 0, int SA = 0, int AH = NS - 1, int WN = 10, int SB = 0>
void run(const char* f, const char* u, const char* w, char* fo, int ntiles, int ncol, int chunkB, long long tileB, const char* tag, int tmap = 0, int ldsper = 10240) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = (ntiles + WPB - 1) / WPB;
  // occupancy as in the real kernel (128 VGPRs: 16 waves per CU): pad every wave to 10 KB of LDS
  const int stat = (MODE == 0 || MODE >= 3) ? NS * 3 * 1024 : MODE == 1 ? NS * 3 * 512 : 0;
  const int dyn = stat < ldsper ? WPB * (ldsper - stat) : 0;
  for (int r = 0; r < 60; ++r)
    hipLaunchKernelGGL((skel3<MODE, NS, WPB, LA, SA, AH, WN, SB>), dim3(blocks), dim3(64 * WPB), dyn, 0, f, u, w, fo, ntiles, ncol, chunkB, tileB, tmap);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 60; ++r)
    hipLaunchKernelGGL((skel3<MODE, NS, WPB, LA, SA, AH, WN, SB>), dim3(blocks), dim3(64 * WPB), dyn, 0, f, u, w, fo, ntiles, ncol, chunkB, tileB, tmap);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 60;
  const double bytes = (double)ntiles * ncol * chunkB * 4;
  printf("%-10s SB %d tmap %d lds/wave %d LA %d SA %d mode %d ring %d ahead %d wait %d waves/wg %d tileB %lld: %.3f ms  %.2f TB/s (err %d)\n", tag, SB, tmap, ldsper, LA, SA, MODE, NS, AH, WN, WPB, tileB, ms,
         bytes / (ms * 1e-3) / 1e12, (int)hipGetLastError());
}

int main(int argc, char** argv) {
  const int ntiles = 32768, ncol = 38, chunkB = 432;
  {
    const long long tileB = 38ll * 432 + 96;   // 16512 = 129 lines
    const size_t n = (size_t)ntiles * 17024;
    char *f, *u, *w, *fo;
    (void)hipMalloc(&f, n + 8192); (void)hipMalloc(&u, n + 8192); (void)hipMalloc(&w, n + 8192); (void)hipMalloc(&fo, n + 8192);
    // pseudo-random contents by default (`./wave_stream zero`: all-zero arrays -- the chip then clocks
    // higher and the rates flatter: DESIGN.md section 4.1)
    if (argc > 1 && argv[1][0] == 'z') {
      (void)hipMemset(f, 0, n); (void)hipMemset(u, 0, n); (void)hipMemset(w, 0, n); (void)hipMemset(fo, 0, n);
    } else {
      char* arr[4] = {f, u, w, fo};
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (double*)arr[i], n / 8, 777ull + i);
      (void)hipDeviceSynchronize();
    }
    for (int rep = 0; rep < 2; ++rep) {
      run<0, 3, 4>(f, u, w, f, ntiles, ncol, chunkB, tileB, "plain");
      run<3, 3, 4, 2, 2>(f, u, w, f, ntiles, ncol, chunkB, tileB, "split OOB");
      run<5, 3, 4, 2, 2>(f, u, w, f, ntiles, ncol, chunkB, tileB, "splitEXEC");
      run<5, 3, 4, 2, 2, 2, 6>(f, u, w, f, ntiles, ncol, chunkB, tileB, "safe wait");      // 2 in flight, stores not counted
      run<5, 3, 4, 2, 2, 3, 12>(f, u, w, f, ntiles, ncol, chunkB, tileB, "safe wait");     // 3 in flight (the kernel)
      run<5, 3, 4, 2, 2, 3, 12>(f, u, w, fo, ntiles, ncol, chunkB, tileB, "safe, f -> fo");   // out of place
      run<5, 3, 4, 2, 2, 3, 16>(f, u, w, f, ntiles, ncol, chunkB, tileB, "3 ahead+st");    // 3 in flight, stores counted
      run<5, 3, 4, 2, 2, 3, 12, 2>(f, u, w, f, ntiles, ncol, chunkB, tileB, "rem st nt");       // remainder stores streaming as well
      run<5, 3, 4, 2, 2, 3, 12, 1>(f, u, w, f, ntiles, ncol, chunkB, tileB, "rem st sc0");
      run<5, 3, 4, 2, 2, 3, 12, 16>(f, u, w, f, ntiles, ncol, chunkB, tileB, "rem st sc1");
      run<5, 3, 4, 2, 0, 3, 12>(f, u, w, f, ntiles, ncol, chunkB, tileB, "safe,cached st");
      run<5, 3, 4, 2, 0, 2, 6>(f, u, w, f, ntiles, ncol, chunkB, tileB, "safe,cached st");
      run<4, 3, 4, 2, 2>(f, u, w, f, ntiles, ncol, chunkB, 17024, "pair-line");   // 38*384 + 19*128 = 133 lines
      run<4, 3, 4, 0, 2>(f, u, w, f, ntiles, ncol, chunkB, 17024, "pair-line");
      run<4, 3, 4, 0, 0>(f, u, w, f, ntiles, ncol, chunkB, 17024, "pair-line");
    }
  }
  return 0;
}
