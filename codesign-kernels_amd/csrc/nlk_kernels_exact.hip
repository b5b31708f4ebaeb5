// EXACT arithmetic variant (-ffp-contract=off): bit-identical to the reference loop.
#define NLK_NS nlk_exact
#include "nlk_kernel_body.h"
