// placeholder, replaced below
#include "ctx.hpp"
namespace mpm {
struct FastState {};
int fast_init(mpmhip_ctx *c) { return fail(c, MPMHIP_ERR_INVALID, "fast mode not built yet"); }
void fast_destroy(mpmhip_ctx *) {}
int fast_step(mpmhip_ctx *c, const StepArgs &) { return fail(c, MPMHIP_ERR_INVALID, "fast mode not built yet"); }
int fast_pull(mpmhip_ctx *c) { return MPMHIP_OK; }
int fast_export_grid(mpmhip_ctx *c, float *, float *, float *) { return fail(c, MPMHIP_ERR_INVALID, "n/a"); }
int fast_stats(mpmhip_ctx *c, mpmhip_stats *) { return MPMHIP_OK; }
int fast_add_collider_storage(mpmhip_ctx *c, MeshCollider &) { return MPMHIP_OK; }
int fast_add_mover_storage(mpmhip_ctx *c, Mover &) { return MPMHIP_OK; }
}
