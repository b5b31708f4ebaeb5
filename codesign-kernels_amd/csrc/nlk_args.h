// nlk_args.h -- kernel argument block of libnlk_hip.so, shared by host and device code.
// Arrays are the reference's (nested_loops/nested_vars.F90:111-127); see include/nlk_hip.h.
#ifndef NLK_ARGS_H
#define NLK_ARGS_H
struct NlkArgs {
  const int* nAdvCellsForEdge; const int* advCellsForEdge; const int* minLevelCell; const int* maxLevelCell;
  const double* tracerCur; const double* normalThicknessFlux; const double* advMaskHighOrder;
  const double* advCoefs; const double* advCoefs3rd;
  double* highOrderFlx;
  double coef3rdOrder;
  int nEdges, nCells, nVertLevels, nvldim, nAdv;
};
#endif
