/*
 * rt_oracle.h - TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement, in plain C, of the reference's per-pixel trace loop
 * (render(), RayTracer/RayTracer.c:1574 and everything below it: spheres,
 * triangles, planes, cubes, cylinders, patterns, reflection, refraction),
 * taking the same flattened scene the HIP library takes (include/rt_hip.h).  It follows
 * the reference statement by statement - 4-component tuples, the 100-slot
 * intersection list, libc qsort, hit(), the container list - so that it can be compared with the
 * reference bit for bit on the double-precision canvas, and the HIP kernels
 * are then compared with it.
 *
 * Pinned by: tests/test_oracle_kat.py (the reference's own unit-test values,
 * RayTracer.c:1710-7044) and tests/test_oracle_golden.py (canvases produced
 * by the reference itself through oracle/ref_harness.c).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rto_ray {
  double origin[4];
  double direction[4];
} rto_ray;

/* counters of the last rto_render_* call (probes, not part of the algorithm) */
typedef struct rto_probe {
  uint64_t max_list;     /* largest intersections.count of any ray            */
  uint64_t exact_zero_t; /* list entries with t == 0.0 exactly when the list is sorted (RayTracer.c:168 quirk) */
  uint64_t list_overflow;/* appends refused because the list held 100 entries */
} rto_probe;

/* render() + (optionally) g x g sub-pixel averaging; rows of params->rank only */
int rto_render_f64(const rt_scene* scene, const rt_render_params* params, double* canvas, rt_stats* stats);
int rto_render_rgb(const rt_scene* scene, const rt_render_params* params, uint8_t* rgb, rt_stats* stats);
void rto_get_probe(rto_probe* out);

/* the pieces, exposed for known-answer tests */
int rto_color_convert(double x);                                          /* RayTracer.c:1322 */
void rto_ray_for_pixel(const rt_camera* cam, double px, double py, rto_ray* out); /* RayTracer.c:1535 */
int rto_intersect_sphere(const double inv[16], const rto_ray* r, double t_out[2]); /* RayTracer.c:892 */
int rto_intersect_triangle(const double p1[3], const double e1[3], const double e2[3], const rto_ray* r,
                           double* t_out);                                /* RayTracer.c:943 */
void rto_normal_at_sphere(const double inv[16], const double point[3], double n_out[3]); /* RayTracer.c:721 */
void rto_reflect(const double in[3], const double n[3], double out[3]);   /* RayTracer.c:1065 */
void rto_lighting(const rt_material* m, const rt_light* light, const double point[3], const double eyev[3],
                  const double normalv[3], int in_shadow, double rgb_out[3]); /* RayTracer.c:1273 */
/* intersect_world + hit: returns world-list position of the hit or -1 */
int rto_hit_world(const rt_scene* scene, const rto_ray* r, double* t_out, int* n_intersections, double* ts_sorted,
                  int ts_cap);                                            /* RayTracer.c:1043, 143 */
int rto_is_shadowed(const rt_scene* scene, const double point[3]);        /* RayTracer.c:1410 */
void rto_color_at(const rt_scene* scene, const rto_ray* r, int remaining, double rgb_out[3]); /* RayTracer.c:1426 */
/* prepare_computations: out = point[3], eyev[3], normalv[3], over_point[3], reflectv[3], inside */
int rto_prepare(const rt_scene* scene, const rto_ray* r, int order, double t, double out[16]); /* RayTracer.c:1360 */

/* plane / cube / cylinder branch of intersect_all_but_triangle (RayTracer.c:904-918): t values in append order */
int rto_intersect_shape(const rt_shape* sp, const rto_ray* r, double* ts, int cap);
void rto_normal_at_shape(const rt_shape* sp, const double point[3], double n_out[4]); /* RayTracer.c:721-790 */
void rto_pattern_at(const rt_material_ext* x, const double point[3], double rgb_out[3]); /* RayTracer.c:1166 */
double rto_schlick(const double eyev[3], const double normalv[3], double n1, double n2); /* RayTracer.c:1470 */
int rto_refractive_indices(const rt_scene* scene, const rto_ray* r, int hit_index, double n_out[2]); /* :1378-1406 */

/* intersections list alone: append ts[0..n) in order, sort, hit(); returns the insertion index of the hit or -1
 * (RayTracer.c:159, 139, 143) */
int rto_hit_list(const double* ts, int n);
/* how many of `attempts` appends of t = 1.0 the 100-slot list accepts (RayTracer.c:161-165) */
int rto_list_capacity(int attempts);

/* P3 text exactly as write_canvas_to_file emits it (RayTracer.c:1333-1348) */
int rto_write_ppm(const char* path, const uint8_t* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
