// FAST arithmetic variant (-ffp-contract=fast).
#define NLK_NS nlk_fast
#include "nlk_kernel_body.h"
