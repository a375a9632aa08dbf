// pt_bounce_g7.hip -- k_bounce instances of geometry path 7 (pt_bounce.h GEOM_*), a translation unit of its own so that
// the paths compile in parallel.
#include "pt_bounce.h"

namespace pt {
const void *bounce_kernel_g7(int workgroup, bool first, int compact, int feat) { return bounce_kernel_for<7>(workgroup, first, compact, feat); }
}  // namespace pt
