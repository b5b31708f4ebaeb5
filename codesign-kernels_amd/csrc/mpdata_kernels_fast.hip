// Arithmetic variant FAST: compiled with -ffp-contract=fast (see Makefile);
// same expressions, FMA contraction allowed.
#define MPDATA_NS mpdata_fast
#include "mpdata_kernels_inst.h"
