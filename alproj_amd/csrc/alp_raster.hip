// placeholder until the raster lands
#include "alp_internal.h"
using namespace alp;
struct alp_mesh { int dummy; };
extern "C" {
int alp_mesh_create(const float *, const float *, int64_t, const void *, int, int64_t, int64_t, int64_t, alp_mesh_t **) { return fail(ALP_ESTATE, "render not built yet"); }
int alp_mesh_destroy(alp_mesh_t *) { return ALP_OK; }
int alp_render(alp_mesh_t *, const double *, const double *, double, float *) { return fail(ALP_ESTATE, "render not built yet"); }
int alp_render_enqueue(alp_mesh_t *, const double *, const double *, double) { return fail(ALP_ESTATE, "render not built yet"); }
int alp_render_fetch(alp_mesh_t *, float *) { return fail(ALP_ESTATE, "render not built yet"); }
int alp_distort_image(const float *, int64_t, int64_t, int64_t, const double *, float *) { return fail(ALP_ESTATE, "render not built yet"); }
}
