// FAST arithmetic variant (-ffp-contract=fast): FMA contraction, rounding differences only.
#define BWK_NS bwk_fast
#include "bwk_kernel_body.h"
