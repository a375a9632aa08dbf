// pt_bounce_g5.hip -- k_bounce instances of geometry path 5 (pt_bounce.h GEOM_*), a translation unit of its own so that
// the paths compile in parallel.
#include "pt_bounce.h"

namespace pt {
const void *bounce_kernel_g5(int workgroup, bool first, int compact, int feat) { return bounce_kernel_for<5>(workgroup, first, compact, feat); }
}  // namespace pt
