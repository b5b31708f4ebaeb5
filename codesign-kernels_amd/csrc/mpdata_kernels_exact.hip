// Arithmetic variant EXACT: compiled with -ffp-contract=off (see Makefile);
// bit-identical to the reference CPU routine built without FMA contraction.
#define MPDATA_NS mpdata_exact
#include "mpdata_kernels_inst.h"
