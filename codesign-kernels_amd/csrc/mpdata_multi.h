// mpdata_multi.h -- internal interface between mpdata_plan.hip (the plan API) and
// mpdata_multi.hip (the multi-GPU orchestration on top of single-device plans).
#ifndef MPDATA_MULTI_H
#define MPDATA_MULTI_H
#include <stdint.h>

struct mpdata_multi;
struct mpdata_plan;

// error text of the library (thread-local, returned by mpdata_last_error)
__attribute__((visibility("hidden"))) int mpdata_internal_set_err(int code, const char* fmt, ...);

__attribute__((visibility("hidden"))) int mpdata_multi_create(int64_t ncrms, int nx, int nz, int ntracers, int ngpus,
                                                              const int* devices, int eb, mpdata_multi** out);
__attribute__((visibility("hidden"))) int mpdata_multi_upload(mpdata_multi* m, const void* f, const void* u, const void* w,
                                                              const void* rho, const void* rhow, const void* adz, const void* flux);
__attribute__((visibility("hidden"))) int mpdata_multi_scatter_device(mpdata_multi* m, const void* f, const void* u, const void* w,
                                                                      const void* rho, const void* rhow, const void* adz,
                                                                      const void* flux, int first, int count);
__attribute__((visibility("hidden"))) int mpdata_multi_gather_device(mpdata_multi* m, void* f, void* flux, int first, int count);
__attribute__((visibility("hidden"))) int mpdata_multi_ranks_seen(const mpdata_multi* m);
__attribute__((visibility("hidden"))) int mpdata_multi_run(mpdata_multi* m, int first, int count);
__attribute__((visibility("hidden"))) int mpdata_multi_sync(mpdata_multi* m);
__attribute__((visibility("hidden"))) int mpdata_multi_download(mpdata_multi* m, void* f, void* flux);
__attribute__((visibility("hidden"))) int mpdata_multi_last_kernel_ms(mpdata_multi* m, double* ms);
__attribute__((visibility("hidden"))) int mpdata_multi_info(const mpdata_multi* m, int g, int* device, int64_t* sl0, int64_t* nloc);
__attribute__((visibility("hidden"))) void mpdata_multi_stats(const mpdata_multi* m, double* scatter_s, double* gather_s,
                                                              int64_t* scatter_bytes_peer, int64_t* gather_bytes_peer, int* xfer);
__attribute__((visibility("hidden"))) int mpdata_multi_ngpus(const mpdata_multi* m);
__attribute__((visibility("hidden"))) struct mpdata_plan* mpdata_multi_sub(const mpdata_multi* m, int g);
__attribute__((visibility("hidden"))) int mpdata_multi_destroy(mpdata_multi* m);
// (mpdata_plan.hip) flux of every tracer of a single-device plan := 0, queued on the plan's stream
__attribute__((visibility("hidden"))) int mpdata_plan_zero_flux_internal(struct mpdata_plan* p);
#endif
