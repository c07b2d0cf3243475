/* Headless OpenGL context on Mesa's software rasteriser (llvmpipe / softpipe), no X, no EGL.
 *
 * BUILD-CONTAINER TOOL of the golden-vector generators (tests/golden/gen_golden_gl.py): it gives the
 * reference's own persp_proj (src/alproj/project.py:145-294) a real OpenGL to run on, so that the
 * render fixtures come from a conformant GL implementation and not from this repository's reading of
 * the GL specification.  Nothing of the product or of the test suite links or loads it.
 *
 * How: dlopen the swrast DRI driver, take its DRI_Core / DRI_SWRast extensions, create a screen with
 * a do-nothing swrast loader (we never present a window-system drawable: everything is rendered into
 * framebuffer objects), create a core-profile context and bind it to a dummy drawable.  GL entry
 * points are then served by libglapi (_glapi_get_proc_address); the Python side calls them through
 * ctypes.
 *
 *   gcc -O2 -shared -fPIC dri_ctx.c -o libdri_ctx.so -ldl
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <GL/gl.h>
#include <GL/internal/dri_interface.h>

static void *g_driver, *g_glapi;
static const __DRIcoreExtension *g_core;
static const __DRIswrastExtension *g_swrast;
static __DRIscreen *g_screen;
static __DRIcontext *g_ctx;
static __DRIdrawable *g_draw;
static const __DRIconfig **g_configs;
static char g_err[256];

static void ld_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p)
{ (void)d; (void)p; *x = 0; *y = 0; *w = 16; *h = 16; }
static void ld_put(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void ld_get(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p)
{ (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }

static const __DRIswrastLoaderExtension g_loader = {
    .base = { __DRI_SWRAST_LOADER, 1 },
    .getDrawableInfo = ld_info, .putImage = ld_put, .getImage = ld_get,
};
static const __DRIextension *g_loader_exts[] = { &g_loader.base, NULL };

const char *dri_ctx_error(void) { return g_err; }

/* returns 0 on success; the context is current on the calling thread afterwards */
int dri_ctx_create(const char *driver_path, int gl_major, int gl_minor)
{
    if (g_ctx) return 0;
    g_glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!g_glapi) { snprintf(g_err, sizeof g_err, "libglapi: %s", dlerror()); return -1; }
    g_driver = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
    if (!g_driver) { snprintf(g_err, sizeof g_err, "driver: %s", dlerror()); return -2; }
    const __DRIextension **(*get_exts)(void) =
        (const __DRIextension **(*)(void))dlsym(g_driver, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) { snprintf(g_err, sizeof g_err, "no %s_swrast", __DRI_DRIVER_GET_EXTENSIONS); return -3; }
    const __DRIextension **exts = get_exts();
    for (int i = 0; exts[i]; i++) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) g_core = (const __DRIcoreExtension *)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) g_swrast = (const __DRIswrastExtension *)exts[i];
    }
    if (!g_core || !g_swrast || g_swrast->base.version < 4) {
        snprintf(g_err, sizeof g_err, "driver lacks DRI_Core / DRI_SWRast v4"); return -4;
    }
    g_screen = g_swrast->createNewScreen2(0, g_loader_exts, exts, &g_configs, NULL);
    if (!g_screen || !g_configs || !g_configs[0]) { snprintf(g_err, sizeof g_err, "createNewScreen2 failed"); return -5; }
    uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, (uint32_t)gl_major,
                           __DRI_CTX_ATTRIB_MINOR_VERSION, (uint32_t)gl_minor };
    unsigned err = 0;
    g_ctx = g_swrast->createContextAttribs(g_screen, __DRI_API_OPENGL_CORE, g_configs[0], NULL, 2, attribs, &err, NULL);
    if (!g_ctx) { snprintf(g_err, sizeof g_err, "createContextAttribs failed (error %u)", err); return -6; }
    g_draw = g_swrast->createNewDrawable(g_screen, g_configs[0], NULL);
    if (!g_draw) { snprintf(g_err, sizeof g_err, "createNewDrawable failed"); return -7; }
    if (!g_core->bindContext(g_ctx, g_draw, g_draw)) { snprintf(g_err, sizeof g_err, "bindContext failed"); return -8; }
    return 0;
}

void *dri_ctx_proc(const char *name)
{
    void *(*gpa)(const char *) = (void *(*)(const char *))dlsym(g_glapi, "_glapi_get_proc_address");
    return gpa ? gpa(name) : NULL;
}

void dri_ctx_destroy(void)
{
    if (!g_ctx) return;
    g_core->unbindContext(g_ctx);
    g_core->destroyDrawable(g_draw);
    g_core->destroyContext(g_ctx);
    g_core->destroyScreen(g_screen);
    g_ctx = NULL; g_draw = NULL; g_screen = NULL;
}
