// bwk_kernel_body.h -- fused HIP kernel of biharmonic_wk_scalar for gfx950
// (reference atmosphere/biharmonic_wk_kernel.F90:109-200; included once per arithmetic
// variant with BWK_NS defined).
//
// A 4x4 slab of qtens is 128 contiguous bytes; the whole array is one linear stream that is
// read once and written once in place, so the kernel is a streaming kernel with ~3 VALU
// lane-operations per byte in between.
//   * A LANE owns one COLUMN of a slab: the 4 points (a, n), a = 0..3, of column n = lane & 3;
//     the 4 lanes of a DPP quad own one slab, a wave 16 slabs = 2 KB of contiguous memory,
//     loaded and stored as 32 contiguous bytes per lane.
//   * Sums over the FIRST index (reference `s(i,j)`, `vtemp(j,n,1)`) run over the lane's own
//     registers; sums over the SECOND index (`s(j,i)`, `vtemp(m,j,2)`) take the values of the
//     other three columns by DPP quad broadcasts (quad_perm, constant control, no LDS).
//   * Dvv is wave-uniform (scalar registers); the element's Dinv / spheremp / tensorVisc at
//     the lane's 4 points are loaded once per workgroup (a workgroup stays inside one element).
//   * Every sum keeps the reference's order (i, j = 1..np ascending, starting from 0), so the
//     EXACT build (-ffp-contract=off) is bit-identical to the reference CPU routine.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace BWK_NS {

constexpr int BWK_THREADS = 256;  // 4 waves = 64 slabs per workgroup pass (128 / 512 threads: -4 ... -8 %)
constexpr int BWK_SLABS_PER_PASS = BWK_THREADS / 4;

struct BwkArgs {
  double* qtens;
  const double* dvv;
  const double* elem;
  long long nelemd;
  int nslab;      // nlev * qsize: slabs per element
  int passes;     // workgroup passes per workgroup (each BWK_SLABS_PER_PASS slabs)
};

typedef double d4 __attribute__((ext_vector_type(4)));
// qtens is read once and written once in place, default cache policy (non-temporal loads: -6 %, stores: +-0,
// both: -8 %, measured in round 3)
#define BWK_LOAD(p) (*reinterpret_cast<const d4*>(p))
#define BWK_STORE(p, v) (*reinterpret_cast<d4*>(p) = (v))

// value of lane `SRC` of the lane's DPP quad
template <int SRC>
__device__ __forceinline__ double quad_bcast(double x) {
  constexpr int ctrl = SRC * 0x55;  // quad_perm:[SRC,SRC,SRC,SRC]
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), ctrl, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), ctrl, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(BWK_THREADS, 2) bwk_kernel(const BwkArgs g) {
  // reference :14 -- default-real (fp32) literal widened to real(8)
  const double rr = (double)0.00000016666666666666f;
  const int tid = threadIdx.x;
  const int n = tid & 3;               // the lane's column (second index of the reference)
  const long long ie = blockIdx.y;
  const double* el = g.elem + 144 * ie;

  // wave-uniform derivative matrix (reference :19-21), Fortran order D[i + 4*j] = Dvv(i+1,j+1)
  double D[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) D[c] = g.dvv[c];
  // per-lane views of it: Dvv(i, n) and Dvv(n, j)
  double dvb[4], dvn[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dvb[c] = g.dvv[c + 4 * n];
    dvn[c] = g.dvv[n + 4 * c];
  }
  // the element at the lane's points (a, n): Dinv(a,n,c,d) = el[a + 4*(n + 4*(c + 2*d))],
  // spheremp(a,n) = el[64 + a + 4n], tensorVisc(a,n,c,d) = el[80 + a + 4*(n + 4*(c + 2*d))]
  // (the 4 points of the lane's column are 4 consecutive doubles of each block: 32-byte loads)
  double Di[4][2][2], Tv[4][2][2], Sp[4];
  {
    const d4 sp = *reinterpret_cast<const d4*>(el + 64 + 4 * n);
    Sp[0] = sp.x; Sp[1] = sp.y; Sp[2] = sp.z; Sp[3] = sp.w;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const d4 di = *reinterpret_cast<const d4*>(el + 4 * (n + 4 * (c + 2 * d)));
        const d4 tv = *reinterpret_cast<const d4*>(el + 80 + 4 * (n + 4 * (c + 2 * d)));
        Di[0][c][d] = di.x; Di[1][c][d] = di.y; Di[2][c][d] = di.z; Di[3][c][d] = di.w;
        Tv[0][c][d] = tv.x; Tv[1][c][d] = tv.y; Tv[2][c][d] = tv.z; Tv[3][c][d] = tv.w;
      }
  }

  const long long slab0 = (long long)blockIdx.x * g.passes * BWK_SLABS_PER_PASS + (tid >> 2);
  double* const base = g.qtens + (ie * g.nslab) * 16 + n * 4;
  // software pipeline: the loads of pass p+1 are in flight while pass p is computed (a wave
  // moves only 2 KB per pass; without this the chip has too few bytes in flight to fill HBM)
  d4 nxt = d4{0.0, 0.0, 0.0, 0.0};
  if (slab0 < g.nslab) nxt = BWK_LOAD(base + slab0 * 16);
  for (int p = 0; p < g.passes; ++p) {
    const long long slab = slab0 + (long long)p * BWK_SLABS_PER_PASS;
    // a quad is whole or not at all inside the element: the DPP broadcasts below stay valid
    if (slab >= g.nslab) break;
    double* const ptr = base + slab * 16;
    const d4 sv = nxt;
    if (p + 1 < g.passes && slab + BWK_SLABS_PER_PASS < g.nslab)
      nxt = BWK_LOAD(ptr + BWK_SLABS_PER_PASS * 16);
    const double sc[4] = {sv.x, sv.y, sv.z, sv.w};

    // ---- gradient_sphere (:109-134) -----------------------------------------------------
    double v1[4], v2[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {  // v1(l, n) = rr * sum_i Dvv(i,l) * s(i,n): own column
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = acc + D[i + 4 * l] * sc[i];
      v1[l] = acc * rr;
    }
    {  // v2(a, n) = rr * sum_i Dvv(i,n) * s(a,i): s(a,i) lives in lane i of the quad
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = acc[a] + dvb[0] * quad_bcast<0>(sc[a]);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = acc[a] + dvb[1] * quad_bcast<1>(sc[a]);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = acc[a] + dvb[2] * quad_bcast<2>(sc[a]);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = acc[a] + dvb[3] * quad_bcast<3>(sc[a]);
#pragma unroll
      for (int a = 0; a < 4; ++a) v2[a] = acc[a] * rr;
    }
    double p1[4], p2[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double ds1 = Di[a][0][0] * v1[a] + Di[a][1][0] * v2[a];
      const double ds2 = Di[a][0][1] * v1[a] + Di[a][1][1] * v2[a];
      // ---- laplace_sphere_wk (:164-182): tensorVisc . grad ---------------------------------
      const double g1 = ds1 * Tv[a][0][0] + ds2 * Tv[a][0][1];
      const double g2 = ds1 * Tv[a][1][0] + ds2 * Tv[a][1][1];
      // ---- divergence_sphere_wk (:138-160) -------------------------------------------------
      const double vt1 = (Di[a][0][0] * g1 + Di[a][0][1] * g2);
      const double vt2 = (Di[a][1][0] * g1 + Di[a][1][1] * g2);
      p1[a] = Sp[a] * vt1;   // spheremp(j,n)*vtemp(j,n,1)
      p2[a] = Sp[a] * vt2;   // spheremp(m,j)*vtemp(m,j,2), read by the other columns
    }
    double out[4] = {0.0, 0.0, 0.0, 0.0};
    // div(m,n) = div(m,n) - ( p1(j,n)*Dvv(m,j) + p2(m,j)*Dvv(n,j) ) * rr,  j ascending
#define BWK_DIV_STEP(J)                                                                        \
    _Pragma("unroll") for (int m = 0; m < 4; ++m)                                              \
        out[m] = out[m] - (p1[J] * D[m + 4 * J] + quad_bcast<J>(p2[m]) * dvn[J]) * rr;
    BWK_DIV_STEP(0)
    BWK_DIV_STEP(1)
    BWK_DIV_STEP(2)
    BWK_DIV_STEP(3)
#undef BWK_DIV_STEP
    BWK_STORE(ptr, (d4{out[0], out[1], out[2], out[3]}));
  }
}

void launch(double* qtens, const double* dvv, const double* elem, long long nelemd, int nlev, int qsize,
            void* stream) {
  BwkArgs g;
  g.qtens = qtens; g.dvv = dvv; g.elem = elem; g.nelemd = nelemd;
  g.nslab = nlev * qsize;
  // passes per workgroup: more amortise the per-workgroup constant loads, fewer give the
  // dispatcher more, shorter workgroups; measured at nelemd=5400: 8 -> 5.05, 16 -> 5.3,
  // 24 -> 4.8 TB/s (BWK_PASSES overrides the cap of 16, for experiments)
  const int total_passes = (g.nslab + BWK_SLABS_PER_PASS - 1) / BWK_SLABS_PER_PASS;
  static const int want = [] { const char* v = getenv("BWK_PASSES"); return v ? atoi(v) : 16; }();
  const int cap = want < 1 ? 1 : want;
  const int nwg = (total_passes + cap - 1) / cap;          // workgroups per element ...
  g.passes = (total_passes + nwg - 1) / nwg;               // ... of equal length (45 passes -> 3 x 15)
  const unsigned gx = (unsigned)((total_passes + g.passes - 1) / g.passes);
  hipLaunchKernelGGL(bwk_kernel, dim3(gx, (unsigned)nelemd, 1), dim3(BWK_THREADS), 0, (hipStream_t)stream, g);
}

}  // namespace BWK_NS
