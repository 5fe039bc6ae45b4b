/* abi_shim.c -- TEST INFRASTRUCTURE.  The entry points of include/pbrt_gpu.h answered by the CPU restatement
 * (pbrt_oracle.c) instead of the device, so that a HOST of the C ABI -- the compiled reference-side binding
 * oracle/_ref/pbrt_gpubind -- can be checked where there is no GPU: PBRT_GPU_LIB=oracle/liboracle_abi_shim.so makes the
 * binding flatten the reference's Scene and "render" it through the oracle, and the image must equal the reference's.
 * Never loaded by the product: pbrt-v3_amd/ binds libpbrt_gpu.so only, and tests name this library explicitly. */
#include <stdlib.h>
#include <string.h>
#include "../include/pbrt_gpu.h"

int oracle_render_tile_count(const PgRenderDesc *rd);
int oracle_render(const PgSceneDesc *s, const PgRenderDesc *rd, PgFilmPixel *film, PgStraySample *strays, int32_t max_strays,
                  int32_t *n_strays, PgCounters *counters);

struct PgScene { const PgSceneDesc *desc; PgCounters counters; };  /* the caller keeps its arrays alive while it renders */
static const char *g_err = "";

int pg_device_count(void) { return 1; }
int pg_set_device(int device) { (void)device; return PG_OK; }
const char *pg_last_error(void) { return g_err; }
int pg_scene_create(const PgSceneDesc *desc, PgScene **out) {
    if (!desc || !out) { g_err = "pg_scene_create: null argument"; return PG_ERR_INVALID; }
    if (desc->abi_version != PG_ABI_VERSION) { g_err = "ABI version mismatch"; return PG_ERR_INVALID; }
    for (int k = 0; k < desc->n_bssrdfs; ++k) {  /* what libpbrt_gpu.so refuses (pg_abi.hip, "BSSRDF %d: ... out of range"), refused here too: a host that passes the shim passes the device */
        const PgBSSRDF *b = &desc->bssrdfs[k];
        const int64_t need = (int64_t)b->n_rho + b->n_radius + 2 * (int64_t)b->n_rho * b->n_radius + b->n_rho;
        if (b->n_rho < 2 || b->n_radius < 2 || b->table < 0 || b->table + need > desc->n_bssrdf_floats || b->match_material < 0 ||
            b->match_material >= desc->n_materials || b->a.tex >= desc->n_textures || b->b.tex >= desc->n_textures) {
            g_err = "BSSRDF: table / material / texture out of range"; return PG_ERR_INVALID;
        }
    }
    PgScene *s = (PgScene *)calloc(1, sizeof(PgScene));
    s->desc = desc;
    *out = s;
    return PG_OK;
}
void pg_scene_destroy(PgScene *s) { free(s); }
int pg_render_tile_count(const PgRenderDesc *rd) { return oracle_render_tile_count(rd); }
int pg_render(PgScene *s, const PgRenderDesc *rd, PgFilmPixel *film, PgStraySample *strays, int32_t max_strays, int32_t *n_strays, int mem,
              void *stream) {
    (void)stream;
    if (mem != PG_MEM_HOST) { g_err = "the oracle shim takes host buffers only"; return PG_ERR_INVALID; }
    int st = oracle_render(s->desc, rd, film, strays, max_strays, n_strays, &s->counters);
    if (st != PG_OK) g_err = "oracle_render failed";
    return st;
}
int pg_counters(PgScene *s, PgCounters *out) { *out = s->counters; return PG_OK; }
int pg_counters_reset(PgScene *s) { memset(&s->counters, 0, sizeof(s->counters)); return PG_OK; }
int pg_scene_set_option(PgScene *s, int32_t option, int32_t value) { (void)s; (void)value; if (option != PG_OPT_OVERLAP_SHADOW) { g_err = "unknown option"; return PG_ERR_INVALID; } return PG_OK; }  /* nothing to overlap on the CPU */
