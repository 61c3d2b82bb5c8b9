// raycast_tiled.hip -- K1 (production form): LDS-staged tile raster.  (placeholder: routes to
// the direct kernel until the tiled pipeline lands)
#include "qs_internal.h"
size_t qs_tiled_workspace_bytes(const qs_ctx *, size_t) { return 0; }
hipError_t qs_launch_raycast_tiled(qs_ctx *c, size_t n, uint64_t seq0) { return qs_launch_raycast_direct(c, n, seq0); }
