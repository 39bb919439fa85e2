// Fused path dispatch (placeholder until the MFMA kernels land): fails loudly, never falls back.
#include <hip/hip_runtime.h>

#include "internal.h"

namespace gator {
int fused_create(gator_ctx*, void*) { return GATOR_OK; }
void fused_destroy(gator_ctx*) {}
int fused_gat_forward(gator_ctx*, const float*, int, float*, float*, void*) { return fail(GATOR_EUNSUPPORTED, "fused GAT kernel not built yet"); }
int fused_mdr_forward(gator_ctx*, const float*, int, float*, void*) { return fail(GATOR_EUNSUPPORTED, "fused MDR kernels not built yet"); }
int fused_upsample(gator_ctx*, const float*, int, float*, void*) { return fail(GATOR_EUNSUPPORTED, "fused upsample kernel not built yet"); }
int fused_forward(gator_ctx*, const float*, int, float*, float*, void*) { return fail(GATOR_EUNSUPPORTED, "fused forward not built yet"); }
}  // namespace gator
