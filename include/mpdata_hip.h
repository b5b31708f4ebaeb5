/*
 * mpdata_hip.h -- C-ABI of libmpdata_hip.so: the MI355X (gfx950) replacement
 * for the compute body of the E3SM-MMF 2D MPDATA tracer-advection routine.
 *
 * What it replaces in the reference (E3SM-Project/codesign-kernels):
 *   mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90
 *     :72-244   subroutine advect_scalar2D_openacc_2   (8 OpenACC kernels)
 *     :247-474  subroutine advect_scalar2D_openacc_1   (17 OpenACC kernels)
 *   both of which compute what :477-642 advect_scalar2D_cpu computes, and
 *   mmf-mpdata-tracer/Makefile:17-20 (the `pgiacc` accelerator target).
 *
 * Array contract (identical to the reference's dummy arguments, :479-484, and
 * the host-associated global adz, :30).  Fortran column-major, the CRM
 * instance index `sl` (reference: nslices; here: ncrms) FASTEST, fp64:
 *   f   (ncrms, -2:nx+3, 1, nzm [, ntracers])  inout
 *   u   (ncrms, -1:nx+3, 1, nzm)               in
 *   w   (ncrms, -1:nx+2, 1, nz )               in   (level nz never read)
 *   rho (ncrms, nzm)  rhow(ncrms, nz)  adz(ncrms, nzm)   in
 *   flux(ncrms, nz [, ntracers])               out  (levels 1..nzm written;
 *                                                    level nz left untouched,
 *                                                    as the reference does)
 * with nzm = nz-1.  On return f holds: interior columns 1..nx = the advected
 * field; halo columns -1,0,nx+1,nx+2 = the first-pass (upwind) value;
 * columns -2 and nx+3 unchanged -- exactly the reference's in-place result.
 * `ntracers` > 1 is this library's extension: the same u,w,rho,rhow,adz
 * applied to ntracers fields, tracer index slowest.
 *
 * All functions return 0 on success, a negative MPDATA_E* code on argument
 * errors, or a positive hipError_t value; mpdata_last_error() gives text.
 * Nothing here falls back to a CPU path: without a usable HIP device the
 * calls fail.
 */
#ifndef MPDATA_HIP_H
#define MPDATA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPDATA_EINVAL (-1)      /* bad sizes / null pointer */
#define MPDATA_EUNSUPPORTED (-2) /* shape outside what the kernels cover */
#define MPDATA_ESTATE (-3)      /* plan used before upload, etc. */
#define MPDATA_ECOMM (-4)       /* RCCL error (multi-GPU plans) */

/* kernel variants (mpdata_set_variant / MPDATA_VARIANT env): */
#define MPDATA_VARIANT_EXACT 0  /* no FMA contraction, IEEE divide, the reference's
                                   expression order: f (every element, halos
                                   included) AND flux(:,1:nzm) are BIT-IDENTICAL
                                   to the reference CPU routine built with
                                   -ffp-contract=off.  For flux the kernels keep
                                   the nx limited vertical fluxes of every lane
                                   and add them onto the FINISHED upwind sum one
                                   by one, the reference's order :545, :624:
                                   in registers (plans at nx <= 66, device calls
                                   at nx <= 36 and nz <= 32: no extra memory, no
                                   extra kernel); the other cases -- larger nx,
                                   mpdata_plan_run_uw, device calls with nz 33 ..
                                   64 -- in a park
                                   array of the size of f's interior per tracer
                                   (with the plan, or kept per host thread and
                                   stream for device calls) that a
                                   finishing kernel adds.  MPDATA_EXACT_FLUX=sum
                                   in the environment does without either
                                   (=hbm: the park array everywhere; device calls
                                   on arrays of 4 GiB and more then do not park):
                                   flux is then the sum of the reference's terms in
                                   another order (sum of upwind terms + sum of
                                   limited terms, each in the reference's i
                                   order), equal to <= 1e-13 relative. */
#define MPDATA_VARIANT_FAST 1   /* FMA contraction allowed, Newton reciprocal in
                                   the limiter: differs from the above by
                                   rounding only (< 1e-12 abs on conditioned
                                   inputs, < 1e-14 relative L1 on the reference's
                                   own input law) */

/* ---- 1. Drop-in call: host arrays, synchronous, transfers included. -------
 * Replaces `call advect_scalar2D_openacc_N(f,u,w,rho,rhow,flux)` (reference
 * :53,:57) including its `!$acc update device/host` traffic (:107,:241).
 * Sizes and adz, which the reference routine takes by host association
 * (:7-30), are explicit arguments here. */
int mpdata_advect_scalar2d(int64_t ncrms, int nx, int nz, int ntracers,
                           double* f, const double* u, const double* w,
                           const double* rho, const double* rhow,
                           const double* adz, double* flux);
/* The call keeps what it needs besides the caller's arrays -- two streams and up to three sets of chunk
 * buffers (3/8 of the arrays' size at the default chunking) -- for the next call of the same HOST
 * THREAD (creating and destroying them costs 6.7 ms per call).  This releases the calling thread's
 * set; a thread that ends releases its own; MPDATA_HOST_CACHE=0 in the environment keeps nothing.
 * The same call releases the park arrays that EXACT calls on reference-layout DEVICE arrays keep per
 * host thread and stream (only where the limited fluxes do not fit registers: nx > 36, nz 33 .. 64), and
 * the wave-major plan that calls on reference-layout device arrays with 65 <= nz <= 238 run through
 * (the size of the call's arrays; kept per host thread for the next call of the same shape). */
int mpdata_release_host_buffers(void);

/* ---- 2. Device-resident call: device pointers, asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = the default stream).  This is the
 * reference's timed region (:110-238: kernels only, data already on the
 * device).  Arrays cover `ncrms` CRM instances with leading dimension
 * `ncrms`.  In-place on f.  nz <= 64: one kernel on the caller's arrays.
 * 65 <= nz <= 238: through a wave-major plan kept per host thread (import,
 * plan kernel, export, all on `stream`; the first call of a shape allocates;
 * MPDATA_DEVICE_CALL=direct: the k-marching kernel on the caller's arrays,
 * a third of the rate).  nz > 238: the k-marching kernel (fp64, nx <= 140). */
int mpdata_advect_scalar2d_device(int64_t ncrms, int nx, int nz, int ntracers,
                                  double* f, const double* u, const double* w,
                                  const double* rho, const double* rhow,
                                  const double* adz, double* flux, void* stream);

/* ---- 3. Plan API: device state owned by the library (what the OpenACC
 * `enter data pcreate` / `update device` / `update host` directives do,
 * reference :105-107, :241, :662-663).  A plan lives on the device that is
 * current when it is created (every plan call switches to it and back), and
 * fixes the kernel variant at creation.
 *
 * Device layout.  The arrays a caller passes are ALWAYS in the reference
 * layout above.  Inside a plan with nz <= 238 (fp64; fp32 with an even ncrms) the library keeps them in
 * its own "wave-major" order -- [tile of 64/LPS adjacent instances][column]
 * [instance][level], LPS = 8/16/32/64 >= nz (nz > 64: one instance per tile, worked on by several waves) -- so that
 * every wave streams contiguous memory; upload / download / import / export
 * convert on the device.  Other plans (nz > 238; fp32 with an odd ncrms), MPDATA_PLAN_LAYOUT=
 * reference or mpdata_set_plan_layout(MPDATA_LAYOUT_REFERENCE) keep the
 * reference layout.  Results do not depend on the layout. */
#define MPDATA_LAYOUT_REFERENCE 0
#define MPDATA_LAYOUT_WAVEMAJOR 1
typedef struct mpdata_plan mpdata_plan;
int mpdata_plan_create(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan);
int mpdata_plan_upload(mpdata_plan* plan, const double* f, const double* u, const double* w,
                       const double* rho, const double* rhow, const double* adz,
                       const double* flux);              /* host arrays; flux may be NULL */
int mpdata_plan_run(mpdata_plan* plan);            /* all tracers; async on the plan's stream */
int mpdata_plan_run_tracers(mpdata_plan* plan, int first_tracer, int ntracers); /* a sub-range */
/* One step on FRESH velocities: u, w are reference-layout DEVICE arrays of the plan's precision
 * (what a CRM whose state lives on the device produces every step; the reference's timed region
 * takes whatever u, w the device arrays hold, :107-110), f, rho, rhow, adz stay in the plan.
 * Entering the plan layout is part of the call and of its event time.  One fp64 tracer of a
 * wave-major plan: a kernel that reads u, w straight from the reference layout (128-byte row
 * segments through an LDS ring shared by the waves of a workgroup) while f streams in the
 * plan layout -- no conversion pass; other calls (tracer batches, fp32, odd ncrms, unaligned
 * bases ...) convert on the way.  On a multi-GPU plan u, w are full-width arrays on the root GPU:
 * they are scattered (section 3b), then every GPU runs.
 * POST-CONDITION (the same on every path): the plan holds NO velocities afterwards -- whether its
 * own u, w were left alone or overwritten depends on the path and is not promised.
 * mpdata_plan_run / _run_tracers return MPDATA_ESTATE until u AND w have been handed over again
 * (mpdata_plan_upload, or mpdata_plan_import_device with u and w); further mpdata_plan_run_uw
 * calls need nothing. */
int mpdata_plan_run_uw(mpdata_plan* plan, int first_tracer, int ntracers, const void* u, const void* w);
int mpdata_plan_sync(mpdata_plan* plan);           /* the `!$acc wait` (:237) */
int mpdata_plan_download(mpdata_plan* plan, double* f, double* flux);  /* host arrays */
int mpdata_plan_last_kernel_ms(mpdata_plan* plan, double* ms); /* hipEvent time of the last run */
/* The event pair behind mpdata_plan_last_kernel_ms is recorded around EVERY run (two marker packets
 * between consecutive launches of a stream, about 1.5 % of a 0.4-ms kernel).  on = 0 switches it
 * off for callers that time a whole loop themselves; last_kernel_ms then returns MPDATA_ESTATE. */
int mpdata_plan_set_timing(mpdata_plan* plan, int on);
int mpdata_plan_destroy(mpdata_plan* plan);
/* Device-side exchange with a plan: reference-layout DEVICE arrays of the plan's precision on
 * the plan's device, asynchronous on the plan's stream.  Import: NULL pointers are skipped
 * (the plan keeps what it has); f and flux cover tracers [first_tracer, first_tracer+ntracers).
 * A caller whose state lives on the device imports once, runs many times, exports when it
 * needs the field back. */
int mpdata_plan_import_device(mpdata_plan* plan, const void* f, const void* u, const void* w,
                              const void* rho, const void* rhow, const void* adz,
                              const void* flux, int first_tracer, int ntracers);
int mpdata_plan_export_device(mpdata_plan* plan, void* f, void* flux, int first_tracer, int ntracers);
/* Run on the caller's stream (hipStream_t as void*; NULL = default stream) from now on. */
int mpdata_plan_set_stream(mpdata_plan* plan, void* stream);
int mpdata_plan_layout(const mpdata_plan* plan);   /* MPDATA_LAYOUT_* */
int mpdata_plan_device(const mpdata_plan* plan);   /* HIP device ordinal */
int mpdata_set_plan_layout(int layout);            /* default for new plans; returns previous */

/* ---- 3b. One problem on several GPUs of the node.  No statement of the routine couples two
 * CRM instances (reference :505-637), so the ncrms axis is cut into `ngpus` contiguous blocks
 * (mpdata_shard_range), each GPU runs a plan of its own on its block and there is NO data-path
 * collective.  What replaces the reference's `!$acc update device / update host` (:107,
 * :241-242) depends on where the global arrays live:
 *   on the ROOT GPU (shard 0's device; mpdata_plan_import_device / _export_device on the
 *   multi-GPU plan, full-width reference-layout arrays): scatter = pack kernel (a block is a
 *   strided slab: `sl` is the fastest axis) + ncclSend / ncclRecv in ONE group -- RCCL over
 *   xGMI, one direct link per peer --, gather the reverse;
 *   on the HOST (mpdata_plan_upload / _download): every GPU copies its own slab ("direct": all
 *   PCIe links in parallel, nothing funnels through the root's one link).
 * One host thread drives all devices; the arrays of one transfer are queued back to back and
 * synchronised once.  The handle is an ordinary plan: upload, import_device, run, run_tracers,
 * sync, download, export_device, last_kernel_ms (slowest GPU) and destroy work on it; results
 * are bitwise those of a single-GPU plan.  MPDATA_MULTI_XFER = rccl | p2p (hipMemcpyPeerAsync)
 * | direct forces one transport for both origins (direct: host arrays only). */
int mpdata_plan_create_multi(int64_t ncrms, int nx, int nz, int ntracers, int ngpus, mpdata_plan** plan);
int mpdata_plan_create_multi_devices(int64_t ncrms, int nx, int nz, int ntracers, int ngpus,
                                     const int* devices, mpdata_plan** plan); /* explicit HIP ordinals */
void mpdata_shard_range(int64_t ncrms, int ngpus, int g, int64_t* sl0, int64_t* nloc); /* block of GPU g */
int mpdata_plan_ngpus(const mpdata_plan* plan);
int mpdata_plan_ranks_seen(const mpdata_plan* plan); /* ncclCommCount of the plan's communicator; 0: none */
int mpdata_plan_shard(const mpdata_plan* plan, int g, int* device, int64_t* sl0, int64_t* nloc);
/* the single-device plan of GPU g (owned by the multi-GPU plan: do not destroy): for callers whose
 * shard already lives on that device -- mpdata_plan_import_device / _export_device on it */
mpdata_plan* mpdata_plan_shard_plan(mpdata_plan* plan, int g);
/* wall seconds and bytes per peer link of the last upload (scatter) / download (gather);
 * transport (of the last transfer): 0 rccl, 1 p2p, 2 direct */
int mpdata_plan_transfer_stats(const mpdata_plan* plan, double* scatter_s, double* gather_s,
                               int64_t* scatter_bytes_per_peer, int64_t* gather_bytes_per_peer,
                               int* transport);

/* ---- 4. Synthetic inputs on the device (bench/tests; the reference's init,
 * :645-660, with a portable counter-based generator instead of the
 * compiler's random_number).  Fills `rows` x `nloc` doubles of array `sid`
 * (0..6 = adz,f,u,w,rho,rhow,flux) for CRM instances [sl0, sl0+nloc) of a
 * global problem of ncrms_global instances. dist: 1 conditioned, 2 raw
 * U[0,1), 3 raw with signed u,w. */
int mpdata_fill_synthetic_device(double* a, int sid, int64_t rows, int64_t ncrms_global,
                                 int64_t sl0, int64_t nloc, uint64_t seed, int dist,
                                 void* stream);

/* ---- 4b. A minimal device workspace for hosts that cannot call HIP themselves (the Fortran
 * driver's device-resident mode: the global arrays are generated on the root GPU with
 * mpdata_fill_synthetic_device and handed to mpdata_plan_import_device).  mpdata_device_sum: the sum
 * of the elements j < n with (j mod stride) < block (block = stride = n: all of them), in a fixed
 * order (a checksum, reproducible run to run). */
int mpdata_device_alloc(void** ptr, int64_t bytes);   /* on the current device */
/* ... on the device a plan takes full-width device arrays from: its own device; the ROOT GPU (shard
 * 0's device) of a multi-GPU plan.  mpdata_plan_import_device / _export_device / _run_uw on a
 * multi-GPU plan return MPDATA_EINVAL for an array that lives anywhere else. */
int mpdata_plan_device_alloc(mpdata_plan* plan, void** ptr, int64_t bytes);
int mpdata_device_free(void* ptr);
int mpdata_device_sum(const double* a, int64_t n, int64_t block, int64_t stride, double* sum);
/* (mpdata_fill_synthetic_device with a NULL stream and mpdata_device_sum run on the device the array
 * lives on, whatever the current device is.) */

/* ---- 5. Shard pack/unpack for the multi-GPU scatter/gather (device
 * pointers).  A shard [sl0, sl0+nloc) of an array with leading dimension
 * ncrms is a strided slab; pack makes it contiguous (leading dimension
 * nloc), unpack writes it back. */
int mpdata_pack_shard_device(const double* full, double* shard, int64_t rows, int64_t ncrms,
                             int64_t sl0, int64_t nloc, void* stream);
int mpdata_unpack_shard_device(double* full, const double* shard, int64_t rows, int64_t ncrms,
                               int64_t sl0, int64_t nloc, void* stream);

/* ---- 6. fp32: the reference's precision switch (`rp`, reference :12-13; note that the
 * `selected_real_kind(7)` it asks for is fp64 on conforming compilers -- IEEE single is
 * `selected_real_kind(6)`).  Same array contract with 4-byte reals.  Kernels cover nz <= 238
 * for even ncrms (two adjacent instances per lane, packed fp32 arithmetic; above 64 levels through
 * a wave-major plan, as the fp64 device call) and nz <= 32 for odd ncrms.  EXACT variant: f bit-identical to an fp32 build of the reference. */
int mpdata_advect_scalar2d_f32(int64_t ncrms, int nx, int nz, int ntracers,
                               float* f, const float* u, const float* w,
                               const float* rho, const float* rhow,
                               const float* adz, float* flux);
int mpdata_advect_scalar2d_f32_device(int64_t ncrms, int nx, int nz, int ntracers,
                                      float* f, const float* u, const float* w,
                                      const float* rho, const float* rhow,
                                      const float* adz, float* flux, void* stream);
int mpdata_fill_synthetic_f32_device(float* a, int sid, int64_t rows, int64_t ncrms_global,
                                     int64_t sl0, int64_t nloc, uint64_t seed, int dist,
                                     void* stream);
int64_t mpdata_algorithmic_bytes_f32(int64_t ncrms, int nx, int nz, int ntracers);
/* fp32 plans: create / upload / download have _f32 forms; run, sync, last_kernel_ms and
 * destroy are the functions of section 3 (a plan remembers its precision; mixing the two
 * returns MPDATA_ESTATE). */
int mpdata_plan_create_f32(int64_t ncrms, int nx, int nz, int ntracers, mpdata_plan** plan);
int mpdata_plan_upload_f32(mpdata_plan* plan, const float* f, const float* u, const float* w,
                           const float* rho, const float* rhow, const float* adz,
                           const float* flux);
int mpdata_plan_download_f32(mpdata_plan* plan, float* f, float* flux);

/* ---- 7. Stage-by-stage debug mode (not a fast path).  The same routine as eight unfused
 * kernels, one per stage of the reference -- the split its OpenACC version makes, reference
 * :112-235 -- that materialise the reference's temporaries (:485-491) in caller-provided
 * device arrays uuu(ncrms,-1:nx+3,nzm), www(ncrms,-1:nx+2,nz), mx/mn(ncrms,0:nx+1,nzm), and
 * stop after stage `last_stage`:
 *   1 extrema of the incoming field (:513-526)   2 upwind fluxes + flux sum (:528-548)
 *   3 first-pass update (:550-560)               4 antidiffusive fluxes (:561-586)
 *   5 extrema of the first-pass field (:588-600) 6 limiter ratios (:601-612)
 *   7 limited fluxes, flux += (:613-627)         8 final update (:630-637)
 * Every array is then what the reference holds at that point, bit for bit (no FMA
 * contraction, reference expression and summation order), so a parity failure of the fused
 * kernels can be localised to a stage.  fp64, one tracer. */
int mpdata_debug_stages_device(int64_t ncrms, int nx, int nz, int last_stage, double* f,
                               const double* u, const double* w, const double* rho,
                               const double* rhow, const double* adz, double* flux,
                               double* uuu, double* www, double* mx, double* mn, void* stream);

/* ---- 8. Misc. */
int mpdata_set_variant(int variant);      /* MPDATA_VARIANT_*; returns previous */
int mpdata_get_variant(void);
/* Serpentine tile order of wave-major plans (every other run of a plan walks its tiles from the
 * other end and so starts on what the previous run left in the Infinity Cache; +2 % when
 * consecutive runs share u, w).  OFF by default (MPDATA_SERPENTINE=1 in the environment turns it
 * on); returns the previous setting. */
int mpdata_set_serpentine(int on);
/* Test switches of the wave-major launch (bit 0: the batch form of the kernel for one tracer as
 * well, bit 1: one tracer per wave in tracer batches, bit 2: an odd last tracer as a two-tracer
 * wave with an empty half, bit 3: an odd last tracer through a launch of its own behind the batch
 * -- the default takes it through one more wave per tile of the batch launch; MPDATA_WM_NOSTREAM /
 * MPDATA_WM_TPW1 / MPDATA_WM_NOSPLIT / MPDATA_WM_SPLIT in the environment set the initial value);
 * flags < 0 only queries.  Returns the previous value. */
int mpdata_set_wm_flags(int flags);
int mpdata_set_tile(int tile);            /* kernel tiling id; -1 = default */
int mpdata_set_debug_buffer(void* dev_ptr); /* diagnostic builds only (-DMPDWM_STAMPS): per-wave stamp buffer */
int mpdata_device_count(void);
int64_t mpdata_algorithmic_bytes(int64_t ncrms, int nx, int nz, int ntracers);
/* Diagnostic: GB/s this GPU sustains for the routine's traffic mix (3 arrays read, 1 written in
 * place) as a linear, aligned, 16-byte-per-lane stream; nontemporal = 1: streaming loads/stores. */
int mpdata_diag_stream_3r1w(int64_t bytes_per_array, int nontemporal, int iters, double* gbs);
const char* mpdata_last_error(void);
const char* mpdata_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MPDATA_HIP_H */
