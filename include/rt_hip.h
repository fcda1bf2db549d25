/*
 * rt_hip.h - C ABI of the MI355X (gfx950) trace-loop library.
 *
 * The library replaces exactly one call of the reference ray tracer:
 *
 *     void render(camera* c, world* w)            RayTracer/RayTracer.c:1574
 *
 * i.e. for every pixel: ray_for_pixel (RayTracer.c:1535) -> color_at(w, r, 5)
 * (RayTracer.c:1426) -> write_pixel (RayTracer.c:257), followed by the 8-bit
 * quantisation of color_convert (RayTracer.c:1322) that write_canvas_to_file
 * (RayTracer.c:1333) applies when it emits the image.
 *
 * The caller keeps building its scene with its own constructors, flattens the
 * `world` linked list (RayTracer.c:92) into the plain arrays of `rt_scene`
 * below (rt_scene.h has a C helper that does it with the reference's
 * arithmetic), uploads it once, and calls rt_hip_render() wherever it called
 * render().  Plain C types only; the caller owns every host buffer, the
 * library owns every device buffer except the ones handed to the *_device
 * entry points.
 *
 * All entry points return RT_OK (0) or a negative rt_status; none aborts.
 * The library keeps one context per HIP device of the process (scene, stream,
 * buffers); rt_hip_init / rt_hip_set_device choose the one the other entry
 * points act on, and the rt_hip_multi_* entry points drive several at once.
 * Like the reference (a single global `canvas`/`contents`,
 * RayTracer.c:102,255) it is not re-entrant: one host thread at a time.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rt_status {
  RT_OK = 0,
  RT_ERR_INVALID = -1,     /* NULL / out-of-range argument                      */
  RT_ERR_NO_DEVICE = -2,   /* no HIP device, or ordinal out of range             */
  RT_ERR_HIP = -3,         /* a HIP runtime call failed (see rt_hip_last_error)  */
  RT_ERR_UNSUPPORTED = -4, /* scene uses something outside the path (e.g. a
                              non-affine or singular transform)                  */
  RT_ERR_NOT_READY = -5,   /* init / upload_scene has not been called            */
  RT_ERR_TOO_LARGE = -6    /* scene or image exceeds a documented limit          */
} rt_status;

/* The part of `material` (RayTracer.c:57) that lighting() (RayTracer.c:1273) and
 * reflected_color() (RayTracer.c:1443) read when has_pattern == false and
 * transparency == 0; rt_material_ext below holds the rest. */
typedef struct rt_material {
  double color[3];
  double ambient;
  double diffuse;
  double specular;
  double shininess;
  double reflective;
} rt_material;

/* The rest of `material` (RayTracer.c:57), one entry per rt_material, for worlds that use patterns or
 * transparency: refracted_color/schlick (RayTracer.c:1453-1483) read transparency and refractive_index,
 * lighting() -> stripe_at_object -> pattern_at (RayTracer.c:1276, 1259, 1166) read the pattern.  The caller
 * supplies inverse(pattern.transform), which stripe_at_object recomputes per call (RayTracer.c:1267). */
enum {
  RT_PATTERN_NONE = -1,    /* has_pattern == false */
  RT_PATTERN_TEST = 0,     /* enum pattern_type, RayTracer.c:53 */
  RT_PATTERN_STRIPE = 1,
  RT_PATTERN_GRADIENT = 2,
  RT_PATTERN_RING = 3,
  RT_PATTERN_CHECKER = 4
};
typedef struct rt_material_ext {
  double transparency;
  double refractive_index;
  double pattern_from[3];
  double pattern_to[3];
  double pattern_inv[16];
  int32_t pattern_type;
  int32_t reserved;
} rt_material_ext;

/* A plane, cube or cylinder of the world list (RayTracer.c:69-89; intersected by the PLANE / CUBE / CYLINDER
 * branches of intersect_all_but_triangle, RayTracer.c:904-918).  `type` holds the reference's enum value. */
enum { RT_SHAPE_PLANE = 0, RT_SHAPE_CUBE = 3, RT_SHAPE_CYLINDER = 4 };
typedef struct rt_shape {
  double inv[16];    /* inverse(transform), row-major                         */
  double minimum;    /* cylinder: shape.minimum / maximum / closed            */
  double maximum;
  int32_t type;
  int32_t closed;
  int32_t material;  /* index into materials                                  */
  int32_t order;     /* position in the world list                            */
} rt_shape;

/* `camera` (RayTracer.c:96) with the loop-invariant inverse(view_transform)
 * that ray_for_pixel recomputes twice per pixel (RayTracer.c:1553,1562)
 * hoisted to the host.  Row-major 4x4. */
typedef struct rt_camera {
  int32_t hsize;
  int32_t vsize;
  double pixel_size;
  double half_width;
  double half_height;
  double inv_view[16];
} rt_camera;

/* First (and only used) `point_light` of the world, RayTracer.c:59,1411,1489. */
typedef struct rt_light {
  double position[3];
  double intensity[3];
} rt_light;

/* One node of the host-built bounding volume hierarchy over the triangles.
 * Inner node (right > 0): children are nodes `left` and `right`.
 * Leaf (right < 0): holds -right triangles, bvh_tri[left .. left - right).
 * Node 0 is the root.  Boxes are padded outwards so that a box test can only
 * produce false positives, never drop a triangle the exhaustive walk of
 * intersect_world (RayTracer.c:1043) would have hit. */
typedef struct rt_bvh_node {
  float lo[3];
  int32_t left;   /* inner: left child index;  leaf: first slot in bvh_tri */
  float hi[3];
  int32_t right;  /* inner: right child index (> 0);  leaf: -(count)        */
} rt_bvh_node;

/* Flattened `world` (RayTracer.c:92).  Every primitive keeps its position in
 * the world's linked list (`*_order`): when two primitives are hit at exactly
 * the same t the reference's stable sort + hit() (RayTracer.c:139-157) keeps
 * the one that comes first in the list, and so does the library. */
typedef struct rt_scene {
  rt_camera camera;
  rt_light light;

  int32_t n_materials;
  const rt_material* materials;

  /* spheres: unit sphere under `transform`; the library needs only
   * inverse(transform) (row-major, bottom row must be exactly 0 0 0 1). */
  int32_t n_spheres;
  const double* sph_inv;        /* [n_spheres][16]                               */
  const int32_t* sph_material;  /* [n_spheres] index into materials             */
  const int32_t* sph_order;     /* [n_spheres] position in the world list       */

  /* triangles, structure of arrays: 12 rows of n_tris doubles, row r at
   * tri_soa + r * n_tris, rows = p1.x p1.y p1.z e1.x e1.y e1.z e2.x e2.y e2.z
   * n.x n.y n.z with e1 = p2 - p1, e2 = p3 - p1, n = normalize(e2 x e1)
   * exactly as create_triangle computes them (RayTracer.c:1004-1006). */
  int32_t n_tris;
  const double* tri_soa;
  const int32_t* tri_material;  /* [n_tris]                                      */
  const int32_t* tri_order;     /* [n_tris]                                      */

  /* optional BVH over the triangles (NULL / 0 = none; brute force is used). */
  int32_t n_bvh_nodes;
  const rt_bvh_node* bvh_nodes;
  const int32_t* bvh_tri;       /* [n_tris] triangle indices in leaf order      */

  /* ---- worlds beyond spheres and triangles (all optional: 0 / NULL) ---- */
  int32_t n_shapes;                     /* planes, cubes, cylinders                       */
  const rt_shape* shapes;
  const rt_material_ext* materials_ext; /* [n_materials] or NULL: no pattern, no transparency */
  /* inverse(transform) of the FIRST object of the world list: shade_hit hands w->objects, not the object
   * that was hit, to lighting() as the shape whose transform places the pattern (RayTracer.c:1489, 1277,
   * 1263).  Read only when a material has a pattern. */
  double first_object_inv[16];
} rt_scene;

enum {
  RT_ACCEL_AUTO = 0,   /* BVH when the scene has one and enough triangles      */
  RT_ACCEL_BRUTE = 1,  /* every ray tests every primitive (the reference's way) */
  RT_ACCEL_BVH = 2,
  /* the kernel that keeps the reference's data structures (100-slot intersection list with its slot re-use and
   * overflow, 4-component tuples; rt_general.hip.inc).  Worlds with planes, cubes, cylinders, patterns or
   * transparency always use it; for sphere / triangle worlds of up to 65536 objects it can be asked for, e.g. to
   * cross-check the fast kernels on the GPU at sizes the CPU reference takes minutes for. */
  RT_ACCEL_REFERENCE = 3
};

/* One render() call.  The image is camera.hsize x camera.vsize.  With
 * n_ranks > 1 the rows are dealt to ranks in bands of band_rows rows,
 * round-robin (band b belongs to rank b % n_ranks) and the call renders only
 * this rank's rows, packed in increasing y. */
typedef struct rt_render_params {
  int32_t max_depth;    /* RECURSION_DEPTH, RayTracer.c:40 (the reference: 5)   */
  int32_t spp_grid;     /* g: g x g sub-pixel samples per pixel; 1 = reference  */
  int32_t accel;        /* RT_ACCEL_*                                            */
  int32_t count_rays;   /* != 0: fill rt_stats ray counters (slower kernel)     */
  int32_t band_rows;    /* rows per band (>= 1); ignored when n_ranks == 1      */
  int32_t n_ranks;      /* >= 1                                                  */
  int32_t rank;         /* 0 .. n_ranks-1                                        */
  int32_t flags;        /* RT_RENDER_* bits, 0 = none                            */
} rt_render_params;

/* rt_render_params.flags */
enum {
  /* The caller keeps several frames in flight at once (rt_hip_render_device on
   * more than one stream).  Such frames fill the GPU between them, so the
   * launch is planned for throughput: no tile is split over several
   * workgroups to shorten a lone frame's slowest wave.  Images are the same
   * either way. */
  RT_RENDER_FRAMES_OVERLAP = 1,
  /* Render with the tile plan as it stands; never measure.  A launch that walks a BVH is as long as its slowest
   * wave, so the library refines its tile plan by measurement (rt_hip_tune below): dozens of timed launches,
   * milliseconds, which pay off over a run of frames and cost a one-frame caller - the reference's main(): one
   * render(), one canvas.ppm, RayTracer.c:7236-7251 - more than the frame itself.  Such a caller sets this bit.
   * Images are the same either way. */
  RT_RENDER_NO_TUNE = 2,
  /* With count_rays: the counters say what the production kernel of the same launch executes, not what the
   * reference's loop does.  The two differ where the library knows a result without tracing: tiles that no camera
   * ray can reach the mesh through are filled with color_at's black (RayTracer.c:1433-1435) instead of being
   * traced, and a camera ray whose whole wave looks past the mesh bounds is never set up.  rays_primary is then the
   * number of camera rays set up (not hsize x vsize x samples) and the test counters leave out the skipped work:
   * the figures a roofline of the timed kernel has to be built from.  Images are the same. */
  RT_RENDER_COUNT_AS_RUN = 4
};

typedef struct rt_stats {
  uint64_t rays_primary;    /* ray_for_pixel calls                               */
  uint64_t rays_shadow;     /* is_shadowed calls (one per shaded hit)            */
  uint64_t rays_secondary;  /* reflected rays spawned                            */
  uint64_t prim_tests;      /* ray-primitive tests executed (triangles + spheres) */
  uint64_t node_tests;      /* ray-box tests executed (BVH nodes, mesh bounds)   */
  uint64_t sphere_tests;    /* the ray-sphere part of prim_tests                 */
  double kernel_ms;         /* hipEvent time of the trace kernel, this call      */
  int32_t rows_rendered;    /* rows this rank produced                           */
  int32_t accel_used;       /* RT_ACCEL_BRUTE, RT_ACCEL_BVH or RT_ACCEL_REFERENCE */
  /* Every ray-primitive test starts in single precision; a primitive the ray cannot be shown to miss goes on
   * to the double-precision routine that follows intersect_triangle / the sphere branch statement by statement
   * (RayTracer.c:943-966, 923-940).  How many did: */
  uint64_t tri_exact;       /* of the triangle tests in prim_tests                */
  uint64_t sphere_exact;    /* of sphere_tests                                    */
  double exchange_ms;       /* rt_hip_render_multi: hipEvent time on device 0 from the end of its own kernel
                               until every device's rows are in place (RCCL receive + placement); else 0 */
  /* The sphere/triangle kernels keep a running (t, list position) minimum where the reference fills a list of
   * 100 entries.  Two properties of that list have no counterpart in a minimum, and a world that triggers them
   * renders differently from the reference unless it is rendered with RT_ACCEL_REFERENCE, whose kernel keeps
   * the list itself.  With count_rays set these say whether the frame met them (both 0: it did not):
   * zero_t_entries: intersections whose t is exactly 0.0 (a ray that starts exactly on a surface); such an
   *   entry reads as a free slot to add_intersection and the next append overwrites it, RayTracer.c:168.
   * list_cap_rays: rays that appended more than 100 entries (two per sphere with real roots, one per triangle
   *   hit, of either sign of t); the reference drops the appends beyond INTERSECTIONS_SIZE, RayTracer.c:161.
   *   Exact for camera and reflected rays under RT_ACCEL_BRUTE (camera rays that run down their tile's list of
   *   triangles included: the counting kernel evaluates every triangle the ray can hit); a BVH walk and a shadow
   *   ray that stops at its first blocker see fewer entries than the reference appends, so there the count is a
   *   lower bound.  A world with 2 * spheres + triangles <= 100 cannot reach the cap at all. */
  uint64_t zero_t_entries;
  uint64_t list_cap_rays;
} rt_stats;

/* Create the context of HIP device `device` (stream, events) if it does not
 * exist yet and make it the current one.  Replaces nothing in the reference
 * (it has no device).  Contexts of other devices stay as they are. */
int rt_hip_init(int device);

/* Make the context of an initialised device the current one (also makes the
 * device current for the calling thread's HIP calls). */
int rt_hip_set_device(int device);

/* Copy the flattened world + camera + light to the device.  Replaces the
 * pointer arguments of render(camera*, world*), RayTracer.c:1574: the scene
 * stays resident until the next upload.  A scene with 4096 or more triangles
 * and no BVH of its own gets one built on the device (see
 * rt_hip_build_bvh_device); RT_ACCEL_BRUTE still walks every triangle. */
int rt_hip_upload_scene(const rt_scene* scene);

/* What rt_hip_build_bvh_device made. */
typedef struct rt_bvh_device_info {
  int32_t n_nodes;  /* inner nodes in device memory (0: the mesh fits one leaf) */
  int32_t depth;    /* inner nodes on the longest root-to-leaf path              */
  double build_ms;  /* hipEvent time of the whole build                          */
} rt_bvh_device_info;

/* Build a bounding volume hierarchy on the device over the triangles of the
 * uploaded scene (Morton order + radix tree) and use it from the next render
 * on, in place of whatever tree the scene came with: for meshes too large or
 * too short-lived for the host builder of rt_scene_finish.  leaf_size in 1..7.
 * The reference has no counterpart (intersect_world, RayTracer.c:1043, tests
 * every object); as with the host tree the image does not depend on it.
 * RT_ERR_TOO_LARGE if the tree would be deeper than the walk's stack, in which
 * case the scene is left as it was.  info may be NULL. */
int rt_hip_build_bvh_device(int leaf_size, rt_bvh_device_info* info);

/* render(c, &w) + color_convert: fills rgb_out (host memory) with this rank's
 * rows, 3 bytes per pixel, row-major (y outer, x inner: the order in which
 * write_canvas_to_file walks the canvas, RayTracer.c:1337-1338).
 * rgb_out must hold rt_hip_rank_rows(...) * hsize * 3 bytes.  stats may be NULL. */
int rt_hip_render(const rt_render_params* params, uint8_t* rgb_out, rt_stats* stats);

/* Same, but rgb_out_device is device memory of the current device and the
 * kernel is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 * library's own stream).  Nothing is copied to the host.
 * stats == NULL: the call only enqueues.  It allocates nothing, creates no
 * event and waits for nothing, so it may be issued on a stream that is being
 * captured into a HIP graph.  It uses the tile plan that exists for these
 * parameters - the one rt_hip_upload_scene makes for a whole frame with
 * RT_ACCEL_AUTO, or one left by rt_hip_tune or by a synchronous call (rt_hip_render,
 * or this one with stats) with the same band layout, kernel choice and
 * RT_RENDER_FRAMES_OVERLAP bit; the library keeps the four most recent.  With no
 * plan at hand the tiles are launched in raster order: the same image, a
 * longer launch.
 * stats != NULL: waits for the kernel; plans, and unless RT_RENDER_NO_TUNE is
 * set tunes, as rt_hip_render does. */
int rt_hip_render_device(const rt_render_params* params, void* rgb_out_device, void* stream,
                         rt_stats* stats);

/* Make and tune the tile plan for `params` ahead of time, so that the frames
 * that follow only enqueue.  Planning sorts the image's 16 x 16 tiles by what
 * they hold (host work plus one upload).  Tuning applies to launches that walk a
 * BVH with few enough mesh tiles to be resident at once: such a launch lasts as
 * long as its slowest wave, so every wave's cycle count is measured, the tiles
 * with the slowest waves are split over several workgroups, and the shortest
 * of up to ten candidate orders is kept (7-10 ms for a 1080p frame of 30-100
 * us; lighthouse 1080p 41 -> 31 us).  Synchronous; renders into a library-owned
 * image.  The reference has no counterpart (its render(), RayTracer.c:1574, is a
 * serial loop with nothing to schedule).  RT_OK also when there is nothing to
 * tune for these parameters. */
int rt_hip_tune(const rt_render_params* params);

/* The colours before quantisation: what write_pixel() stores into `canvas`
 * (RayTracer.c:255-259), as 3 doubles per pixel, row-major (y outer).  Host
 * memory.  For parity diagnostics. */
int rt_hip_render_f64(const rt_render_params* params, double* canvas_out, rt_stats* stats);

/* ---- the image file, formatted on the device ----
 * write_canvas_to_file() (RayTracer.c:1333-1348) prints the canvas as ASCII P3:
 * "P3\n<W> <H>\n255\n", then per pixel (y outer, x inner) "%d %d %d \n".  These
 * entry points produce exactly those bytes from an 8-bit image in device
 * memory (what rt_hip_render_device leaves there); the caller writes them to
 * the file with one fwrite. */

/* Upper bound of the text length for a width x height image (header + 13 bytes per pixel). */
size_t rt_hip_ppm_capacity(int width, int height);

/* rgb_device: width * height * 3 bytes, 4-byte aligned; text_device: at least
 * rt_hip_ppm_capacity() bytes of device memory, 16-byte aligned.  Enqueued on
 * `stream` (NULL = the library's).  bytes_out != NULL: waits and stores the
 * length of the text; NULL: returns at once (the length can be had later from
 * rt_hip_ppm_last_length, which waits).  Calls share library-owned length
 * arrays: a call on another stream than the previous one is ordered after it. */
int rt_hip_format_ppm_device(const void* rgb_device, int width, int height, void* text_device, size_t capacity,
                             size_t* bytes_out, void* stream);
int rt_hip_ppm_last_length(size_t* bytes_out);

/* rt_hip_format_ppm_device for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  It has
 * no counterpart in the reference.  The call allocates nothing and frees
 * nothing: the per-chunk text lengths live in a fixed array in the library's
 * device image (256 KiB per GPU, held until the process ends), which limits
 * the image to 67,108,864 pixels (RT_ERR_TOO_LARGE beyond).  The length of the
 * text is stored to *length_device (device memory, 8-byte aligned) by the
 * kernel; the call never waits.  Same bytes as rt_hip_format_ppm_device.
 * All calls on one GPU share the array: calls on different streams are
 * ordered one after the other by the library (the stream of the previous call
 * must still exist when the next call is made); calls from different host
 * threads at once would race.  On a stream that is being captured the two
 * kernels are recorded without any such ordering: the caller orders the
 * graph's launches against every other use of this entry point on the GPU. */
int rt_hip_format_ppm_device_noalloc(const void* rgb_device, int width, int height, void* text_device,
                                     size_t capacity, unsigned long long* length_device, void* stream);

/* The same into host memory: formats into a library-owned device buffer and
 * copies the text (not the 13-bytes-per-pixel bound) to text_host.
 * RT_ERR_TOO_LARGE with *bytes_out = needed size if capacity is too small. */
int rt_hip_format_ppm(const void* rgb_device, int width, int height, char* text_host, size_t capacity,
                      size_t* bytes_out);

/* render(c, &w) followed by write_canvas_to_file()'s formatting, all on the
 * device: the whole frame (n_ranks must be 1) arrives in text_host as the
 * bytes of canvas.ppm.  stats may be NULL. */
int rt_hip_render_ppm(const rt_render_params* params, char* text_host, size_t capacity, size_t* bytes_out,
                      rt_stats* stats);

/* ---- several GPUs driven by one host process ----
 * The reference is one process and one render(camera*, world*) call (RayTracer.c:1574, 7236-7251); a C host that
 * has several MI355X renders the frame on all of them: the scene is replicated, the image rows are dealt out in
 * round-robin bands, and the bands come together on device 0 in ONE exchange step over xGMI (a grouped RCCL
 * send/recv: every device sends its rows straight to device 0).  RCCL is loaded (dlopen) by rt_hip_multi_init; a
 * one-GPU host never touches it.  The image does not depend on the number of devices. */

/* Below this many samples per frame (pixels x spp_grid^2) several GPUs do not pay: a frame of 2 M samples is one
 * launch of 13-30 us, which already sits at the latency of a single wave, while the gather over xGMI and its RCCL
 * launches cost tens of microseconds; at 50 M samples a GPU works for a millisecond or more and N devices approach
 * N x (8K x 16 samples: 530 M).  Hosts use the group only when asked to, and never below this. */
#define RT_MULTI_GPU_MIN_SAMPLES 50000000ll

/* HIP devices visible to the process (0 if there is none or the runtime fails). */
int rt_hip_device_count(void);
/* Contexts for devices 0 .. n_devices-1 (0 = every visible device) and the RCCL communicators between them. */
int rt_hip_multi_init(int n_devices);
/* Number of devices in the group (0 before rt_hip_multi_init). */
int rt_hip_multi_devices(void);
/* rt_hip_upload_scene on every device of the group. */
int rt_hip_multi_upload_scene(const rt_scene* scene);
/* rt_hip_render over the group: the whole frame arrives in rgb_out (hsize * vsize * 3 bytes, host memory).
 * params->n_ranks must be 1 and rank 0 (the library deals the bands out: band_rows > 1 sets the band height,
 * otherwise 16 rows).  stats: counters summed over the devices, kernel_ms the longest a device rendered, exchange_ms
 * from the end of device 0's own rendering until every device's rows are in place.  Unless RT_RENDER_NO_TUNE is set the
 * devices' tile plans are tuned after the first frame has been delivered. */
int rt_hip_render_multi(const rt_render_params* params, uint8_t* rgb_out, rt_stats* stats);
/* In how many parts a device's rows travel (1 .. 4; 0 = the library decides per frame: two for frames of
 * RT_MULTI_GPU_MIN_SAMPLES samples or more, else one).  With c parts a device is c ranks of the round-robin over
 * n x c bands and renders them in c launches; part k leaves for device 0 while part k + 1 is being rendered, and
 * device 0 places and copies out part after part, so the exchange and the copy to the host hide under the tail of
 * the rendering.  The image does not depend on it. */
int rt_hip_multi_set_chunks(int chunks);

/* Where rank `rank`'s packed rows start in a buffer that holds every rank's rows, rank after rank (bytes). */
size_t rt_hip_gather_offset(int hsize, int vsize, int band_rows, int n_ranks, int rank);
/* Device memory to device memory: the gathered buffer (every rank's packed rows, rank after rank) into image
 * order, on `stream` (NULL = the library's).  The step after the exchange, on the root. */
int rt_hip_place_bands_device(const void* gathered_device, void* rgb_out_device, int hsize, int vsize, int band_rows,
                              int n_ranks, void* stream);

/* The same for a run of frames, two in flight: rt_hip_render_ppm_submit(params, slot) enqueues render + formatting
 * of a frame into slot 0 or 1 and returns at once; rt_hip_render_ppm_collect(slot, ...) waits for that frame,
 * copies exactly the bytes of its text to pinned host memory owned by the library (on a copy stream of its own)
 * and hands out the pointer, valid until the slot is submitted again.  Submit frame f + 1 to the other slot before
 * collecting frame f and the copy of f overlaps the rendering of f + 1: write_canvas_to_file()'s bytes
 * (RayTracer.c:1333-1348) at the rate of the host link instead of one synchronous frame at a time.
 * RT_ERR_NOT_READY: the slot still holds an uncollected frame (submit) / holds none (collect).  stats: kernel time
 * and kernel choice of that frame; ray counters only through rt_hip_render_ppm. */
int rt_hip_render_ppm_submit(const rt_render_params* params, int slot);
int rt_hip_render_ppm_collect(int slot, const char** text_host, size_t* bytes_out, rt_stats* stats);

/* Number of image rows that belong to `rank` (see rt_render_params). */
int rt_hip_rank_rows(int vsize, int band_rows, int n_ranks, int rank);

/* Image row held at packed row `i` of rank `rank`'s buffer. */
int rt_hip_rank_row_to_y(int i, int band_rows, int n_ranks, int rank);

/* Timing helpers for callers that pipeline several renders: record a start
 * (which == 0) or stop (which == 1) hipEvent on `stream` (NULL = the library's
 * stream), then read the elapsed milliseconds (waits for the stop event).
 * The reference times render() with clock() around the call (RayTracer.c:7236-7249). */
int rt_hip_event_record(int which, void* stream);
int rt_hip_event_elapsed_ms(double* ms_out);

/* Release device memory, streams and events of every context. */
void rt_hip_shutdown(void);

const char* rt_hip_strerror(int status);
/* Text of the last failing HIP call in this process ("" if none). */
const char* rt_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
