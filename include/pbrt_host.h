/*
 * pbrt_host.h -- C entry points of the host front end (libpbrt_host.so).
 *
 * The host library is the C++ mirror of pbrt-v3's scene-file front end and
 * API state machine (src/core/parser.cpp, src/core/api.cpp): it parses a
 * .pbrt file, builds the BVHAccel and flattens everything into the
 * PgSceneDesc / PgRenderDesc consumed by the HIP back end (pbrt_gpu.h).
 * These functions exist so that non-C++ hosts (the Python test-suite and
 * bench.py through ctypes) can drive the same code the `pbrt_amd` CLI uses.
 */
#ifndef PBRT_HOST_H
#define PBRT_HOST_H
#include "pbrt_gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct PbrtHostScene PbrtHostScene;

/* pbrtInit + pbrtParseFile + pbrtCleanup with WorldEnd stopping before Render
 * (main/pbrt.cpp:76-173).  quick = --quick; crop = NULL or {x0,x1,y0,y1}
 * (--cropwindow).  Returns NULL if the file produced no renderable scene.   */
PbrtHostScene *pbrt_host_load_file(const char *filename, int quick, const float *crop);
PbrtHostScene *pbrt_host_load_string(const char *text, int quick, const float *crop);
void pbrt_host_free(PbrtHostScene *s);

const PgSceneDesc *pbrt_host_scene_desc(PbrtHostScene *s);
void pbrt_host_render_desc(PbrtHostScene *s, PgRenderDesc *out);

/* Film: cropped image size; MergeFilmTile for one shard; final RGB image
 * (row-major, top row first, 3 floats per pixel) as Film::WriteImage computes it. */
void pbrt_host_film_size(PbrtHostScene *s, int *width, int *height);
void pbrt_host_film_clear(PbrtHostScene *s);
void pbrt_host_film_merge(PbrtHostScene *s, const PgRenderDesc *rd, const PgFilmPixel *film,
                          const PgStraySample *strays, int n_strays);
/* The n shards of ONE frame (shard r = the tiles t = r (mod n) of the frame `full` describes: tile_first 0, tile_step 1), merged in the
 * frame's own tile order -- what a one-device render merges, bit for bit, also for filters whose tile blocks overlap
 * (Film::MergeFilmTile, core/film.cpp:117-130; the reference's cross-machine analogue is imgtool assemble, tools/imgtool.cpp:190-285). */
void pbrt_host_film_merge_shards(PbrtHostScene *s, const PgRenderDesc *full, int n, const PgFilmPixel *const *film,
                                 const PgStraySample *const *strays, const int *n_strays);
void pbrt_host_film_image(PbrtHostScene *s, float *rgb);
int pbrt_host_write_pfm(const char *filename, const float *rgb, int width, int height);
/* WriteImage (core/imageio.cpp:81-122): PFM, gamma-encoded 8-bit PNG / TGA, or half-float OpenEXR (uncompressed), chosen by the
 * file name's suffix. */
int pbrt_host_write_image(const char *filename, const float *rgb, int width, int height);
/* The same for a (cropped) film: width x height pixels whose upper left corner sits at (x_offset, y_offset) of a total_x x total_y
 * frame -- WriteImage's outputBounds / totalResolution arguments, which only OpenEXR files record (data and display window). */
int pbrt_host_write_image_window(const char *filename, const float *rgb, int width, int height, int x_offset, int y_offset, int total_x, int total_y);

/* The host's own HLBVH build over bare bounds (n x {pMin, pMax}); nodes has room for 2n entries.  Same contract as
 * pg_hlbvh_build (pbrt_gpu.h), which must reproduce it bit for bit. */
void pbrt_host_hlbvh_build(int n, const float *bounds, int max_prims_in_node, PgBVHNode *nodes, int *n_nodes, int *ordered_prims);
/* AnimatedTransform(Transform(start), start_time, Transform(end), end_time).MotionBounds(bounds) (core/transform.cpp:1215-1247): the box of a
 * TransformedPrimitive whose object bound is `bounds` ({pMin, pMax}); start / end row-major 4 x 4.  Returns hasRotation (transform.cpp:411),
 * or -1 (with an Error) where the reference's CHECK_LE ends its process: a motion derivative with more than 8 zeros (transform.cpp:385). */
int pbrt_host_motion_bounds(const float *start, const float *end, float start_time, float end_time, const float *bounds, float *out);
/* Build "hlbvh" accelerators of subsequently loaded scenes on the device (the CLI's --devicebvh). */
void pbrt_host_set_device_bvh(int on);

/* Number of distinct Error() messages reported so far in this process. */
int pbrt_host_error_count(void);

#ifdef __cplusplus
}
#endif
#endif
