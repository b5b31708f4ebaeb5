// EXACT arithmetic variant (-ffp-contract=off): bit-identical to the reference CPU routine.
#define BWK_NS bwk_exact
#include "bwk_kernel_body.h"
