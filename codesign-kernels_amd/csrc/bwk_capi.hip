// bwk_capi.hip -- the C-ABI of libbwk_hip.so (include/bwk_hip.h).  No CPU compute path.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bwk_hip.h"

namespace bwk_exact {
void launch(double*, const double*, const double*, long long, int, int, void*);
}
namespace bwk_fast {
void launch(double*, const double*, const double*, long long, int, int, void*);
}

namespace {
thread_local std::string g_err;
int g_variant = -1;
int set_err(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
int variant() {
  if (g_variant < 0) {
    const char* v = getenv("BWK_VARIANT");
    g_variant = (v && (!strcmp(v, "fast") || !strcmp(v, "1"))) ? BWK_VARIANT_FAST : BWK_VARIANT_EXACT;
  }
  return g_variant;
}
int validate(int64_t nelemd, int nlev, int qsize) {
  if (nelemd < 1 || nlev < 1 || qsize < 1)
    return set_err(BWK_EINVAL, "bad sizes nelemd=%lld nlev=%d qsize=%d", (long long)nelemd, nlev, qsize);
  if (nelemd > 65535) return set_err(BWK_EUNSUPPORTED, "nelemd > 65535 per call");
  if ((long long)nlev * qsize > (1ll << 27)) return set_err(BWK_EUNSUPPORTED, "nlev*qsize too large");
  return 0;
}
}  // namespace

extern "C" {

int bwk_biharmonic_wk_scalar_device(int64_t nelemd, int nlev, int qsize, double* qtens, const double* dvv,
                                    const double* elem, void* stream) {
  int rc = validate(nelemd, nlev, qsize);
  if (rc) return rc;
  if (!qtens || !dvv || !elem) return set_err(BWK_EINVAL, "null array pointer");
  if (((uintptr_t)qtens & 31) != 0 || ((uintptr_t)elem & 31) != 0)
    return set_err(BWK_EINVAL, "qtens and elem must be 32-byte aligned");
  if (variant() == BWK_VARIANT_FAST) bwk_fast::launch(qtens, dvv, elem, nelemd, nlev, qsize, stream);
  else bwk_exact::launch(qtens, dvv, elem, nelemd, nlev, qsize, stream);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err((int)e, "bwk kernel launch: %s", hipGetErrorString(e));
  return 0;
}

int bwk_biharmonic_wk_scalar(int64_t nelemd, int nlev, int qsize, double* qtens, const double* dvv,
                             const double* elem) {
  int rc = validate(nelemd, nlev, qsize);
  if (rc) return rc;
  if (!qtens || !dvv || !elem) return set_err(BWK_EINVAL, "null array pointer");
  const size_t nq = (size_t)16 * nlev * qsize * nelemd * 8, ne = (size_t)144 * nelemd * 8;
  double *dq = nullptr, *dd = nullptr, *de = nullptr;
  hipError_t e = hipMalloc((void**)&dq, nq);
  if (e == hipSuccess) e = hipMalloc((void**)&dd, 16 * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&de, ne);
  if (e == hipSuccess) e = hipMemcpy(dq, qtens, nq, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dd, dvv, 16 * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(de, elem, ne, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = bwk_biharmonic_wk_scalar_device(nelemd, nlev, qsize, dq, dd, de, nullptr);
    if (rc == 0) e = hipMemcpy(qtens, dq, nq, hipMemcpyDeviceToHost);
  }
  if (dq) (void)hipFree(dq);
  if (dd) (void)hipFree(dd);
  if (de) (void)hipFree(de);
  if (rc) return rc;
  if (e != hipSuccess) return set_err((int)e, "bwk_biharmonic_wk_scalar: %s", hipGetErrorString(e));
  return 0;
}

int bwk_set_variant(int v) {
  const int prev = variant();
  if (v == BWK_VARIANT_EXACT || v == BWK_VARIANT_FAST) g_variant = v;
  return prev;
}
int bwk_get_variant(void) { return variant(); }
int64_t bwk_algorithmic_bytes(int64_t nelemd, int nlev, int qsize) {
  return (int64_t)2 * 8 * 16 * nlev * qsize * nelemd + 8 * (144 * nelemd + 16);
}
const char* bwk_last_error(void) { return g_err.c_str(); }

}  // extern "C"
