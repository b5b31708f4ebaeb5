// mpdata_args.h -- kernel argument block shared by host and device code.
#ifndef MPDATA_ARGS_H
#define MPDATA_ARGS_H

// Arrays are the reference's dummy arguments (reference
// mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90:479-484) plus the
// host-associated adz (:30); see include/mpdata_hip.h for the layout.
// R = double (the reference as shipped, :13) or float (its fp32 build, :12).
template <typename R>
struct MpdataArgsT {
  R* f;
  const R* u;
  const R* w;
  const R* rho;
  const R* rhow;
  const R* adz;
  R* flux;
  long long ncrms;         // CRM instances = leading dimension of every array
  int nx, nz;
  int ntracers;            // x-marching kernels: the 1-D grid is ntracers * groups, tracer fastest
  long long f_tstride;     // elements between consecutive tracers of f
  long long flux_tstride;  // ... of flux
  unsigned long long* dbg;  // unused by these kernels (the wave-major kernels' diagnostic build has its own: MpdataWmArgsT::dbg)
  R* wpark;                 // x-marching kernels, EXACT only, may be null: park array of the limited vertical fluxes,
                            // [workgroup][column 1..nx][thread] (bit-identical flux: xmarch_flux_finish_kernel adds them in order)
  int park_regs;            // x-marching kernels, EXACT only: 1 = keep the limited vertical fluxes in registers instead
                            // (nx <= MPDATA_WM_NPK, arrays below 4 GiB; wpark is then null)
};
typedef MpdataArgsT<double> MpdataArgs;
typedef MpdataArgsT<float> MpdataArgsF32;

// Wave-major kernels (mpdata_kernel_wm_body.h): arrays in the plan-private layout
//   f,u,w [tracer][tile][column 0..nx+5][instance-in-tile][level 0..nzm-1],
//   kc    [tile][3 = rho,adz,rhow][instance-in-tile][level],   flux [tracer][tile][inst][level]
template <typename R>
struct MpdataWmArgsT {
  R* f;
  const R* u;
  const R* w;
  const R* kc;
  R* flux;
  int ntiles;              // ceil(ncrms / instances per tile)
  int nx, nz;
  int ntracers;            // tracers this launch works on (f, flux point at the first of them)
  long long tile_elems;    // elements between consecutive tiles of f, u, w: (nx+6) * SLP * nzm
  long long f_tstride;     // elements between consecutive tracers of f: ntiles * tile_elems
  long long flux_tstride;  // ... of flux: ntiles * SLP * nzm
  int reverse;             // walk the tiles from the last to the first
  // mpdata_plan_run_uw (kernel instantiation UWREF): u, w in the REFERENCE layout instead of a.u, a.w
  const R* u_ref;          // u(ncrms, nx+5, nzm)
  const R* w_ref;          // w(ncrms, nx+4, nz)
  long long ncrms;         // leading dimension of u_ref, w_ref
  R* wpark;                // EXACT only, may be null: park array of the limited vertical fluxes [tracer][tile][nx][64]
                           // (bit-identical flux: the finishing kernel adds them in the reference's order)
  unsigned long long* dbg; // diagnostic builds only (-DMPDWM_STAMPS): 8 words per wave (tools/wave_timeline.py); else null
  int nkw;                 // nz > 64 (kernel form LPS = 128): waves per instance and tracer, 1 + ceil((nz - 64) / 58); else 1
  int lwt;                 // nz > 64: 16 / 32 = the last window of an instance is a 16- / 32-lane share of a wave (else 0)
  int ksg;                 // ... and the instances per workgroup of that form (set by the launcher)
  int park_regs;           // EXACT only: 1 = park the limited vertical fluxes in REGISTERS (nx <= MPDATA_WM_NPK, one tracer per
                           // wave; wpark is then null: no park array, no finishing kernel)
};
typedef MpdataWmArgsT<double> MpdataWmArgs;
#define MPDATA_WM_NPK 36   // columns the register park of the EXACT kernels holds (six trips of six columns)
#define MPDATA_WM_NPK2 66  // ... of the second instantiation of the wave-major kernels (nx 37 .. 66: 247 VGPRs, no spill; the
                           // x-march kernels spill at that size and keep the park array)
// test switches of the wave-major launch (mpdata_set_wm_flags; MPDATA_WM_NOSTREAM / _TPW1 / _NOSPLIT)
#define MPDATA_WMF_NOSTREAM 1   // run the batch form of the kernel on a single tracer as well
#define MPDATA_WMF_TPW1 2       // tracer batches: one tracer per wave
#define MPDATA_WMF_NOSPLIT 4    // an odd last tracer stays in the two-tracer launch, as a two-tracer wave with an empty half
#define MPDATA_WMF_SPLIT 8      // an odd last tracer goes through the one-tracer kernel behind the batch (rounds 2-4; A/B)

// One tiling of the kernel template (W columns per thread, SPW strips per
// wave, NWV waves per workgroup).
struct MpdataTileInfo {
  int id;
  int W, SPW, NWV;
  int slw;      // CRM instances per workgroup
  int ncol;     // columns covered (needs nx + 4 <= ncol); k-marching kernels only
  int nz_max;   // largest nz; x-marching kernels only (lanes along k)
  int elem_bytes;  // 8: fp64 kernels, 4: fp32 kernels
  int threads;
  const char* name;
};

typedef void (*mpdata_launch_fn)(const MpdataArgs& a, int ntracers, void* stream);

#endif
