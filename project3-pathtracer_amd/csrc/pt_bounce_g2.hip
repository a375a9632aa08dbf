// pt_bounce_g2.hip -- k_bounce instances of geometry path 2 (pt_bounce.h GEOM_*), a translation unit of its own so that
// the paths compile in parallel.
#include "pt_bounce.h"

namespace pt {
const void *bounce_kernel_g2(int workgroup, bool first, int compact, int feat) { return bounce_kernel_for<2>(workgroup, first, compact, feat); }
}  // namespace pt
