// nlk_capi.hip -- the C-ABI of libnlk_hip.so (include/nlk_hip.h).  No CPU compute path.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "nlk_hip.h"

#include "nlk_args.h"

namespace nlk_exact {
void launch(const NlkArgs& g, void* stream, int mode);
}
namespace nlk_fast {
void launch(const NlkArgs& g, void* stream, int mode);
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
int g_kernel = -2;   // -2: read NLK_KERNEL on first use; -1 automatic, 0 one level per lane, 1 two levels per lane
int kernel_mode() {
  if (g_kernel == -2) {
    const char* f = getenv("NLK_KERNEL");
    g_kernel = f ? atoi(f) : -1;
  }
  return g_kernel;
}
int variant() {
  if (g_variant < 0) {
    const char* v = getenv("NLK_VARIANT");
    g_variant = (v && (!strcmp(v, "fast") || !strcmp(v, "1"))) ? NLK_VARIANT_FAST : NLK_VARIANT_EXACT;
  }
  return g_variant;
}
int validate(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv) {
  if (nEdges < 1 || nCells < 1 || nVertLevels < 1 || nvldim < nVertLevels || nAdv < 1)
    return set_err(NLK_EINVAL, "bad sizes nEdges=%d nCells=%d nVertLevels=%d nvldim=%d nAdv=%d", nEdges, nCells,
                   nVertLevels, nvldim, nAdv);
  return 0;
}
NlkArgs fill(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv, const int* a, const int* b, const int* c,
       const int* d, const double* t, const double* n, const double* m, const double* c1, const double* c3,
       double coef, double* out) {
  NlkArgs g;
  g.nAdvCellsForEdge = a; g.advCellsForEdge = b; g.minLevelCell = c; g.maxLevelCell = d;
  g.tracerCur = t; g.normalThicknessFlux = n; g.advMaskHighOrder = m; g.advCoefs = c1; g.advCoefs3rd = c3;
  g.highOrderFlx = out; g.coef3rdOrder = coef;
  g.nEdges = nEdges; g.nCells = nCells; g.nVertLevels = nVertLevels; g.nvldim = nvldim; g.nAdv = nAdv;
  return g;
}
}  // namespace

extern "C" {

int nlk_high_order_flux_device(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv,
                               const int* nAdvCellsForEdge, const int* advCellsForEdge,
                               const int* minLevelCell, const int* maxLevelCell, const double* tracerCur,
                               const double* normalThicknessFlux, const double* advMaskHighOrder,
                               const double* advCoefs, const double* advCoefs3rd, double coef3rdOrder,
                               double* highOrderFlx, void* stream) {
  int rc = validate(nEdges, nCells, nVertLevels, nvldim, nAdv);
  if (rc) return rc;
  if (!nAdvCellsForEdge || !advCellsForEdge || !minLevelCell || !maxLevelCell || !tracerCur ||
      !normalThicknessFlux || !advMaskHighOrder || !advCoefs || !advCoefs3rd || !highOrderFlx)
    return set_err(NLK_EINVAL, "null array pointer");
  if (variant() == NLK_VARIANT_FAST)
    nlk_fast::launch(fill(nEdges, nCells, nVertLevels, nvldim, nAdv, nAdvCellsForEdge,
                                             advCellsForEdge, minLevelCell, maxLevelCell, tracerCur,
                                             normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd,
                                             coef3rdOrder, highOrderFlx), stream, kernel_mode());
  else
    nlk_exact::launch(fill(nEdges, nCells, nVertLevels, nvldim, nAdv, nAdvCellsForEdge,
                                               advCellsForEdge, minLevelCell, maxLevelCell, tracerCur,
                                               normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd,
                                               coef3rdOrder, highOrderFlx), stream, kernel_mode());
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_err((int)e, "nlk kernel launch: %s", hipGetErrorString(e));
  return 0;
}

int nlk_high_order_flux(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv,
                        const int* nAdvCellsForEdge, const int* advCellsForEdge, const int* minLevelCell,
                        const int* maxLevelCell, const double* tracerCur, const double* normalThicknessFlux,
                        const double* advMaskHighOrder, const double* advCoefs, const double* advCoefs3rd,
                        double coef3rdOrder, double* highOrderFlx) {
  int rc = validate(nEdges, nCells, nVertLevels, nvldim, nAdv);
  if (rc) return rc;
  if (!nAdvCellsForEdge || !advCellsForEdge || !minLevelCell || !maxLevelCell || !tracerCur ||
      !normalThicknessFlux || !advMaskHighOrder || !advCoefs || !advCoefs3rd || !highOrderFlx)
    return set_err(NLK_EINVAL, "null array pointer");
  const size_t nb[10] = {(size_t)nEdges * 4, (size_t)nAdv * nEdges * 4, (size_t)nCells * 4, (size_t)nCells * 4,
                         (size_t)nvldim * nCells * 8, (size_t)nvldim * nEdges * 8, (size_t)nvldim * nEdges * 8,
                         (size_t)nAdv * nEdges * 8, (size_t)nAdv * nEdges * 8, (size_t)nvldim * nEdges * 8};
  const void* h[10] = {nAdvCellsForEdge, advCellsForEdge, minLevelCell, maxLevelCell, tracerCur,
                       normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd, highOrderFlx};
  void* d[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 10 && e == hipSuccess; ++i) e = hipMalloc(&d[i], nb[i]);
  // highOrderFlx goes in too: its padding rows (nVertLevels+1..nvldim) are not written
  for (int i = 0; i < 10 && e == hipSuccess; ++i) e = hipMemcpy(d[i], h[i], nb[i], hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = nlk_high_order_flux_device(nEdges, nCells, nVertLevels, nvldim, nAdv, (const int*)d[0], (const int*)d[1],
                                    (const int*)d[2], (const int*)d[3], (const double*)d[4], (const double*)d[5],
                                    (const double*)d[6], (const double*)d[7], (const double*)d[8], coef3rdOrder,
                                    (double*)d[9], nullptr);
    if (rc == 0) e = hipMemcpy(highOrderFlx, d[9], nb[9], hipMemcpyDeviceToHost);
  }
  for (int i = 0; i < 10; ++i)
    if (d[i]) (void)hipFree(d[i]);
  if (rc) return rc;
  if (e != hipSuccess) return set_err((int)e, "nlk_high_order_flux: %s", hipGetErrorString(e));
  return 0;
}

int nlk_set_variant(int v) {
  const int prev = variant();
  if (v == NLK_VARIANT_EXACT || v == NLK_VARIANT_FAST) g_variant = v;
  return prev;
}
int nlk_get_variant(void) { return variant(); }
int nlk_set_kernel(int mode) {
  const int prev = kernel_mode();
  if (mode >= -1 && mode <= 2) g_kernel = mode;
  return prev;
}
int64_t nlk_algorithmic_bytes(int nEdges, int nCells, int nVertLevels, int nvldim, int nAdv) {
  (void)nvldim;
  return (int64_t)8 * nVertLevels * ((int64_t)nCells + 3ll * nEdges) + (int64_t)nEdges * (4 + (int64_t)nAdv * 20) +
         (int64_t)nCells * 8;
}
const char* nlk_last_error(void) { return g_err.c_str(); }

}  // extern "C"
