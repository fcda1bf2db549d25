/*
 * rt_oracle.c - TEST INFRASTRUCTURE, not product code.  See rt_oracle.h.
 *
 * Statement-by-statement CPU restatement of the reference's trace loop.  All
 * line numbers refer to RayTracer/RayTracer.c of steven-schronk/RayTracer.
 * Compile without FMA contraction (-ffp-contract=off, no -mfma): the
 * reference binary is plain x86-64 SSE2 code.
 *
 * Deliberate differences from the reference, none of which changes a value:
 *   - inverse(transform), inverse(pattern.transform) and inverse(view_transform)
 *     arrive precomputed (the reference recomputes them per ray: lines 900,
 *     724, 1263, 1267, 1553, 1562);
 *   - the container list of prepare_computations (lines 1378-1406) holds
 *     world-list positions instead of shape pointers.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RTO_EPSILON 0.000001 /* line 37 */
#define RTO_LIST_SIZE 100    /* line 65 */

typedef struct { double x, y, z, w; } v4; /* tuple, line 51 */

static v4 v4_make(double x, double y, double z, double w) { v4 t = {x, y, z, w}; return t; }
static v4 v4_point(double x, double y, double z) { return v4_make(x, y, z, 1.0); }   /* line 269 */
static v4 v4_vector(double x, double y, double z) { return v4_make(x, y, z, 0.0); }  /* line 274 */
static v4 v4_add(v4 a, v4 b) { return v4_make(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); } /* line 304 */
static v4 v4_sub(v4 a, v4 b) { return v4_make(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); } /* line 309 */
static v4 v4_neg(v4 a) { return v4_sub(v4_make(0.0, 0.0, 0.0, 0.0), a); }                    /* line 314 */
static v4 v4_scale(v4 a, double s) { return v4_make(a.x * s, a.y * s, a.z * s, a.w * s); }    /* line 320 */
static v4 v4_mul(v4 a, v4 b) { return v4_make(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }  /* line 330 */
static double v4_mag(v4 a) { /* line 335: pow(.,2) is an exact square */
  double mx = a.x * a.x, my = a.y * a.y, mz = a.z * a.z, mw = a.w * a.w;
  return sqrt(mx + my + mz + mw);
}
static v4 v4_normalize(v4 a) { /* line 344: four divisions, no reciprocal */
  double m = v4_mag(a);
  return v4_make(a.x / m, a.y / m, a.z / m, a.w / m);
}
static double v4_dot(v4 a, v4 b) { /* line 350 */
  double p1 = a.x * b.x, p2 = a.y * b.y, p3 = a.z * b.z, p4 = a.w * b.w;
  return p1 + p2 + p3 + p4;
}
static v4 v4_cross(v4 a, v4 b) { /* line 359 */
  return v4_vector(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static v4 mat_mul_v4(const double m[16], v4 b) { /* line 419, row-major */
  v4 c;
  c.x = b.x * m[0] + b.y * m[1] + b.z * m[2] + b.w * m[3];
  c.y = b.x * m[4] + b.y * m[5] + b.z * m[6] + b.w * m[7];
  c.z = b.x * m[8] + b.y * m[9] + b.z * m[10] + b.w * m[11];
  c.w = b.x * m[12] + b.y * m[13] + b.z * m[14] + b.w * m[15];
  return c;
}
static v4 v4_reflect(v4 in, v4 normal) { /* line 1065 */
  v4 two_n = v4_scale(normal, 2.0);
  double d = v4_dot(in, normal);
  return v4_sub(in, v4_scale(two_n, d));
}

typedef struct { v4 origin, direction; } ray_t; /* line 61 */

/* ---- intersection list, lines 63-67, 112-178 ---- */
typedef struct { double t; long id; } isect_t;
typedef struct { isect_t e[RTO_LIST_SIZE]; int count; } isect_list;

static rto_probe g_probe;

static void list_init(isect_list* l) { /* line 112 */
  l->count = 0;
  for (int i = 0; i < RTO_LIST_SIZE; ++i) { l->e[i].t = DBL_MIN; l->e[i].id = -1; }
}
static int list_add(isect_list* l, double t, long id) { /* line 159 */
  if (l->count == RTO_LIST_SIZE) { ++g_probe.list_overflow; return 0; }
  int i = 0;
  /* a slot counts as used only while its t is neither 0.0 nor DBL_MIN (line 168) */
  while (i < RTO_LIST_SIZE - 1 && l->e[i].t != 0.0 && l->e[i].t != DBL_MIN) ++i;
  l->count++;
  l->e[i].id = id;
  l->e[i].t = t;
  return 1;
}
static int isect_cmp(const void* a, const void* b) { /* line 131 */
  const isect_t* c = a; const isect_t* d = b;
  if (c->t > d->t) return 1;
  if (c->t < d->t) return -1;
  return 0;
}
static void list_sort(isect_list* l) { /* line 139: same libc qsort as the reference */
  if ((uint64_t)l->count > g_probe.max_list) g_probe.max_list = (uint64_t)l->count;
  for (int i = 0; i < l->count; ++i) if (l->e[i].t == 0.0) ++g_probe.exact_zero_t; /* as the harness probe counts them */
  qsort(l->e, (size_t)l->count, sizeof(isect_t), isect_cmp);
}
static const isect_t* list_hit(const isect_list* l) { /* line 143 */
  if (l->count == 0) return NULL;
  const isect_t* best = NULL;
  double t = DBL_MAX;
  for (int i = 0; i < RTO_LIST_SIZE; ++i) {
    if (l->e[i].t == DBL_MIN) break;
    if (l->e[i].t < 0) continue;
    if (l->e[i].t < t) { t = l->e[i].t; best = &l->e[i]; }
  }
  return best;
}

/* ---- world in list order ---- */
typedef struct { int kind; int idx; } prim_ref; /* kind 0 sphere, 1 triangle, 2 plane / cube / cylinder */
typedef struct {
  const rt_scene* s;
  int n;
  prim_ref* order; /* order[k] = primitive at world-list position k */
  uint64_t rays_shadow, rays_secondary;
} world_t;

static int world_open(world_t* w, const rt_scene* s) {
  w->s = s;
  w->n = s->n_spheres + s->n_tris + s->n_shapes;
  w->rays_shadow = w->rays_secondary = 0;
  w->order = malloc(sizeof(prim_ref) * (size_t)(w->n > 0 ? w->n : 1));
  if (!w->order) return RT_ERR_INVALID;
  for (int k = 0; k < w->n; ++k) w->order[k].kind = -1;
  for (int i = 0; i < s->n_spheres; ++i) {
    int k = s->sph_order[i];
    if (k < 0 || k >= w->n || w->order[k].kind != -1) { free(w->order); return RT_ERR_INVALID; }
    w->order[k].kind = 0; w->order[k].idx = i;
  }
  for (int i = 0; i < s->n_tris; ++i) {
    int k = s->tri_order[i];
    if (k < 0 || k >= w->n || w->order[k].kind != -1) { free(w->order); return RT_ERR_INVALID; }
    w->order[k].kind = 1; w->order[k].idx = i;
  }
  for (int i = 0; i < s->n_shapes; ++i) {
    int k = s->shapes[i].order;
    if (k < 0 || k >= w->n || w->order[k].kind != -1) { free(w->order); return RT_ERR_INVALID; }
    w->order[k].kind = 2; w->order[k].idx = i;
  }
  return RT_OK;
}
static void world_close(world_t* w) { free(w->order); w->order = NULL; }

static v4 tri_row(const rt_scene* s, int i, int r0, double w) {
  const double* b = s->tri_soa;
  size_t n = (size_t)s->n_tris;
  return v4_make(b[(size_t)r0 * n + i], b[(size_t)(r0 + 1) * n + i], b[(size_t)(r0 + 2) * n + i], w);
}

/* ---- intersect_all_but_triangle, sphere branch: lines 898-901, 923-940 ---- */
static void isect_sphere(const double inv[16], const ray_t* r, isect_list* l, long id) {
  ray_t r2; /* transform(), line 792 */
  r2.origin = mat_mul_v4(inv, r->origin);
  r2.direction = mat_mul_v4(inv, r->direction);
  v4 sphere_to_ray = v4_sub(r2.origin, v4_point(0.0, 0.0, 0.0));
  double a = v4_dot(r2.direction, r2.direction);
  double b = 2 * v4_dot(r2.direction, sphere_to_ray);
  double c = v4_dot(sphere_to_ray, sphere_to_ray) - 1.0;
  double disc = b * b - 4 * a * c;
  if (disc < 0) return;
  double t1 = (-b - sqrt(disc)) / (2 * a);
  double t2 = (-b + sqrt(disc)) / (2 * a);
  if (t1 < t2) { list_add(l, t1, id); list_add(l, t2, id); }
  else { list_add(l, t2, id); list_add(l, t1, id); }
}

/* ---- intersect_triangle, lines 943-966 ---- */
static int isect_triangle_t(v4 p1, v4 e1, v4 e2, const ray_t* r, double* t_out) {
  v4 dir_cross_e2 = v4_cross(r->direction, e2);
  double det = v4_dot(e1, dir_cross_e2);
  if (fabs(det) < RTO_EPSILON) return 0;
  double f = 1.0 / det;
  v4 p1_to_origin = v4_sub(r->origin, p1);
  double u = f * v4_dot(p1_to_origin, dir_cross_e2);
  if (u < 0.0 || u > 1) return 0;
  v4 origin_cross_e1 = v4_cross(p1_to_origin, e1);
  double v = f * v4_dot(r->direction, origin_cross_e1);
  if (v < 0.0 || (u + v) > 1.0) return 0;
  *t_out = f * v4_dot(e2, origin_cross_e1);
  return 1;
}

/* ---- plane / cube / cylinder branches of intersect_all_but_triangle, lines 898-918 ---- */
static double max3(double a, double b, double c) { /* line 687 */
  double m = -INFINITY;
  if (a > m) m = a;
  if (b > m) m = b;
  if (c > m) m = c;
  return m;
}
static double min3(double a, double b, double c) { /* line 695 */
  double m = INFINITY;
  if (a < m) m = a;
  if (b < m) m = b;
  if (c < m) m = c;
  return m;
}
static void check_axis(double origin, double direction, double* tmin_out, double* tmax_out) { /* line 799 */
  double tmin_numerator = -1 - origin;
  double tmax_numerator = 1 - origin;
  double tmin, tmax;
  if (fabs(direction) >= DBL_EPSILON) {
    tmin = tmin_numerator / direction;
    tmax = tmax_numerator / direction;
  } else {
    tmin = tmin_numerator * INFINITY;
    tmax = tmax_numerator * INFINITY;
  }
  if (tmin > tmax) { double tt = tmin; tmin = tmax; tmax = tt; }
  *tmin_out = tmin;
  *tmax_out = tmax;
}
static void isect_cube(const ray_t* r, isect_list* l, long id) { /* line 826 */
  double xmin, xmax, ymin, ymax, zmin, zmax;
  check_axis(r->origin.x, r->direction.x, &xmin, &xmax);
  check_axis(r->origin.y, r->direction.y, &ymin, &ymax);
  check_axis(r->origin.z, r->direction.z, &zmin, &zmax);
  double tmin = max3(xmin, ymin, zmin);
  double tmax = min3(xmax, ymax, zmax);
  if (tmin >= tmax) return;
  if (tmax < 0) return;
  list_add(l, tmin, id);
  list_add(l, tmax, id);
}
static int check_cap(const ray_t* r, double t) { /* line 842 */
  double x = r->origin.x + t * r->direction.x;
  double z = r->origin.z + t * r->direction.z;
  return x * x + z * z <= 1;
}
static void isect_cylinder(const rt_shape* sp, const ray_t* r, isect_list* l, long id) { /* lines 863-890, 849-861 */
  double a = r->direction.x * r->direction.x + r->direction.z * r->direction.z;
  if ((!a) < RTO_EPSILON) { /* as written in the reference: true whenever a != 0 */
    double b = 2.0 * r->origin.x * r->direction.x + 2.0 * r->origin.z * r->direction.z;
    double c = r->origin.x * r->origin.x + r->origin.z * r->origin.z - 1.0;
    double disc = b * b - 4.0 * a * c;
    if (disc < 0) return;
    double t0 = (-b - (sqrt(disc))) / (2 * a);
    double t1 = (-b + (sqrt(disc))) / (2 * a);
    if (t0 > t1) { double tt = t0; t0 = t1; t1 = tt; }
    double y0 = r->origin.y + t0 * r->direction.y;
    if (sp->minimum < y0 && y0 < sp->maximum) list_add(l, t0, id);
    double y1 = r->origin.y + t1 * r->direction.y;
    if (sp->minimum < y1 && y1 < sp->maximum) list_add(l, t1, id);
  }
  if (!sp->closed) return; /* intersect_caps */
  double t = (sp->minimum - r->origin.y) / r->direction.y;
  if (check_cap(r, t)) list_add(l, t, id);
  t = (sp->maximum - r->origin.y) / r->direction.y;
  if (check_cap(r, t)) list_add(l, t, id);
}
static void isect_shape(const rt_shape* sp, const ray_t* r, isect_list* l, long id) {
  ray_t r2; /* transform(), line 792 */
  r2.origin = mat_mul_v4(sp->inv, r->origin);
  r2.direction = mat_mul_v4(sp->inv, r->direction);
  switch (sp->type) {
    case RT_SHAPE_PLANE: /* lines 904-912 */
      if (fabs(r2.direction.y) < RTO_EPSILON) return;
      list_add(l, (-r2.origin.y) / r2.direction.y, id);
      return;
    case RT_SHAPE_CUBE:
      isect_cube(&r2, l, id);
      return;
    case RT_SHAPE_CYLINDER:
      isect_cylinder(sp, &r2, l, id);
      return;
    default:
      return;
  }
}

/* ---- intersect_world, lines 1043-1055 ---- */
static void isect_world(const world_t* w, const ray_t* r, isect_list* l) {
  const rt_scene* s = w->s;
  for (int k = 0; k < w->n; ++k) {
    prim_ref p = w->order[k];
    if (p.kind == 0) {
      isect_sphere(s->sph_inv + 16 * (size_t)p.idx, r, l, k);
    } else if (p.kind == 2) {
      isect_shape(&s->shapes[p.idx], r, l, k);
    } else {
      double t;
      if (isect_triangle_t(tri_row(s, p.idx, 0, 1.0), tri_row(s, p.idx, 3, 0.0), tri_row(s, p.idx, 6, 0.0), r, &t))
        list_add(l, t, k);
    }
  }
  list_sort(l);
}

/* ---- normal_at, lines 721-754 ---- */
static v4 normal_at_sphere(const double inv[16], v4 point) {
  v4 local_point = mat_mul_v4(inv, point);
  v4 local_normal = v4_vector(local_point.x, local_point.y, local_point.z);
  double tr[16]; /* mat4x4_transpose, line 426 */
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tr[i * 4 + j] = inv[j * 4 + i];
  v4 world_normal = mat_mul_v4(tr, local_normal);
  world_normal.w = 0.0;
  return v4_normalize(world_normal);
}
static v4 normal_at_shape(const rt_shape* sp, v4 point) {
  if (sp->type == RT_SHAPE_PLANE) { /* normal_at_shape_sphere_plane, lines 721-749 */
    double tr[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) tr[i * 4 + j] = sp->inv[j * 4 + i];
    v4 world_normal = mat_mul_v4(tr, v4_vector(0.0, 1.0, 0.0));
    world_normal.w = 0.0;
    return v4_normalize(world_normal);
  }
  if (sp->type == RT_SHAPE_CUBE) { /* normal_at_cube, lines 756-775: world point, not normalised, w == 1 */
    double maxc = max3(fabs(point.x), fabs(point.y), fabs(point.z));
    if (fabs(maxc - fabs(point.x)) < RTO_EPSILON) return v4_point(point.x, 0, 0);
    if (fabs(maxc - fabs(point.y)) < RTO_EPSILON) return v4_point(0, point.y, 0);
    return v4_point(0, 0, point.z);
  }
  return v4_vector(point.x, 0, point.z); /* normal_at_cylinder, line 777 */
}
static v4 normal_at(const world_t* w, long id, v4 point) {
  prim_ref p = w->order[id];
  if (p.kind == 0) return normal_at_sphere(w->s->sph_inv + 16 * (size_t)p.idx, point);
  if (p.kind == 2) return normal_at_shape(&w->s->shapes[p.idx], point);
  /* normalize(cross(e2, e1)), line 752; create_triangle stores the same value (line 1006) */
  return v4_normalize(v4_cross(tri_row(w->s, p.idx, 6, 0.0), tri_row(w->s, p.idx, 3, 0.0)));
}
static const rt_material* material_of(const world_t* w, long id) {
  prim_ref p = w->order[id];
  int m = p.kind == 0 ? w->s->sph_material[p.idx] : (p.kind == 2 ? w->s->shapes[p.idx].material : w->s->tri_material[p.idx]);
  return &w->s->materials[m];
}
static const rt_material_ext* material_ext_of(const world_t* w, const rt_material* m) {
  return w->s->materials_ext ? &w->s->materials_ext[m - w->s->materials] : NULL;
}

/* ---- comps / prepare_computations, lines 94, 1360-1408 ---- */
typedef struct {
  double t; long id; v4 point, eyev, normalv, over_point, under_point, reflectv; int inside; double n1, n2;
} comps_t;

/* containers, lines 98-102, 180-253: entries are world-list positions, -1 = NULL */
typedef struct { long e[RTO_LIST_SIZE]; } containers_t;
static void cont_clear(containers_t* c) { for (int i = 0; i < RTO_LIST_SIZE; ++i) c->e[i] = -1; }
static int cont_is_empty(const containers_t* c) { return c->e[0] == -1; }
static long cont_last(const containers_t* c) { /* line 201 */
  long last = -1;
  for (int i = 0; i < RTO_LIST_SIZE && c->e[i] != -1; ++i) last = c->e[i];
  return last;
}
static int cont_includes(const containers_t* c, long id) { /* line 213 */
  for (int i = 0; i < RTO_LIST_SIZE; ++i) if (c->e[i] == id) return 1;
  return 0;
}
static void cont_append(containers_t* c, long id) { /* line 222 */
  for (int i = 0; i < RTO_LIST_SIZE; ++i) if (c->e[i] == -1) { c->e[i] = id; break; }
}
static void cont_remove(containers_t* c, long id) { /* line 233 */
  int i = 0, found = 0;
  while (i < RTO_LIST_SIZE && c->e[i] != -1) {
    if (c->e[i] == id) { found = 1; break; }
    i++;
  }
  if (found) {
    while (i < RTO_LIST_SIZE - 1 && c->e[i] != -1) {
      c->e[i] = c->e[i + 1];
      c->e[i + 1] = -1;
      i++;
    }
  }
}
static double refractive_index_of(const world_t* w, long id) {
  const rt_material_ext* x = material_ext_of(w, material_of(w, id));
  return x ? x->refractive_index : 1.0; /* create_material_default, line 713 */
}

static comps_t prepare(const world_t* w, const isect_t* hit, const ray_t* r, const isect_list* xs) {
  comps_t c;
  c.t = hit->t;
  c.id = hit->id;
  c.point = v4_add(r->origin, v4_scale(r->direction, c.t)); /* position(), line 1026 */
  c.eyev = v4_neg(r->direction);
  c.normalv = normal_at(w, c.id, c.point);
  if (v4_dot(c.normalv, c.eyev) < 0.0f) { c.inside = 1; c.normalv = v4_neg(c.normalv); }
  else c.inside = 0;
  c.over_point = v4_add(c.point, v4_scale(c.normalv, RTO_EPSILON));
  c.under_point = v4_sub(c.point, v4_scale(c.normalv, RTO_EPSILON));
  c.reflectv = v4_reflect(r->direction, c.normalv);
  c.n1 = c.n2 = 1.0; /* the reference leaves them unset when the loop below never reaches the hit */
  if (xs) { /* refraction section, lines 1378-1406 */
    containers_t cont;
    cont_clear(&cont);
    for (int k = 0; k < xs->count; k++) {
      const isect_t* i = &xs->e[k];
      if (i->t == hit->t) c.n1 = cont_is_empty(&cont) ? 1.0f : refractive_index_of(w, cont_last(&cont));
      if (cont_includes(&cont, i->id)) cont_remove(&cont, i->id);
      else cont_append(&cont, i->id);
      if (i->t == hit->t) {
        c.n2 = cont_is_empty(&cont) ? 1.0f : refractive_index_of(w, cont_last(&cont));
        break;
      }
    }
  }
  return c;
}

/* ---- pattern_at / stripe_at_object, lines 1166-1206, 1259-1271 ---- */
static v4 pattern_at(const rt_material_ext* x, v4 point) {
  v4 from = v4_point(x->pattern_from[0], x->pattern_from[1], x->pattern_from[2]);
  v4 to = v4_point(x->pattern_to[0], x->pattern_to[1], x->pattern_to[2]);
  switch (x->pattern_type) {
    case RT_PATTERN_TEST:
      return v4_point(point.x, point.y, point.z);
    case RT_PATTERN_STRIPE:
      return ((int)floor(point.x) % 2 == 0) ? v4_point(from.x, from.y, from.z) : v4_point(to.x, to.y, to.z);
    case RT_PATTERN_GRADIENT: {
      v4 distance = v4_sub(to, from);
      double fraction = point.x - floor(point.x);
      return v4_add(from, v4_scale(distance, fraction));
    }
    case RT_PATTERN_RING:
      return ((int)floor(sqrt(point.x * point.x + point.z * point.z)) % 2 == 0) ? from : to;
    case RT_PATTERN_CHECKER: {
      int position = (int)(floor(abs((int)point.x)) + floor(abs((int)point.y)) + floor(abs((int)point.z)));
      return (position % 2 == 0) ? from : to;
    }
    default:
      return v4_point(0.0, 0.0, 0.0);
  }
}

/* ---- lighting, lines 1273-1320.  `first_inv`: inverse transform of the world's first object, which is what
 * shade_hit passes as `sh` (line 1489) ---- */
static v4 lighting(const rt_material* m, const rt_material_ext* x, const double* first_inv, const rt_light* light,
                   v4 point, v4 eyev, v4 normalv, int in_shadow) {
  v4 color = v4_make(m->color[0], m->color[1], m->color[2], 1.0);
  if (x && x->pattern_type != RT_PATTERN_NONE) { /* stripe_at_object, line 1259 */
    v4 object_point = mat_mul_v4(first_inv, point);
    v4 pattern_point = mat_mul_v4(x->pattern_inv, object_point);
    color = pattern_at(x, pattern_point);
  }
  v4 intensity = v4_make(light->intensity[0], light->intensity[1], light->intensity[2], 1.0);
  v4 light_pos = v4_point(light->position[0], light->position[1], light->position[2]);
  v4 effective_color = v4_mul(color, intensity);
  v4 diffuse, specular;
  v4 black = v4_vector(0.0, 0.0, 0.0);
  v4 lightv = v4_normalize(v4_sub(light_pos, point));
  v4 ambient = v4_scale(effective_color, m->ambient);
  double light_dot_normal = v4_dot(lightv, normalv);
  if (light_dot_normal < 0) {
    diffuse = black;
    specular = black;
  } else {
    diffuse = v4_scale(v4_scale(effective_color, m->diffuse), light_dot_normal);
    v4 reflectv = v4_reflect(v4_neg(lightv), normalv);
    double reflect_dot_eye = v4_dot(reflectv, eyev);
    if (reflect_dot_eye <= 0) specular = black;
    else specular = v4_scale(v4_scale(intensity, m->specular), pow(reflect_dot_eye, m->shininess));
  }
  if (in_shadow) return ambient;
  return v4_add(v4_add(ambient, specular), diffuse);
}

/* ---- is_shadowed, lines 1410-1422 ---- */
static int is_shadowed(world_t* w, v4 point) {
  const rt_light* L = &w->s->light;
  v4 v = v4_sub(v4_point(L->position[0], L->position[1], L->position[2]), point);
  double distance = v4_mag(v);
  v4 direction = v4_normalize(v);
  ray_t r; /* create_ray, line 279 */
  r.origin = v4_point(point.x, point.y, point.z);
  r.direction = v4_vector(direction.x, direction.y, direction.z);
  isect_list l;
  list_init(&l);
  ++w->rays_shadow;
  isect_world(w, &r, &l);
  const isect_t* h = list_hit(&l);
  return h != NULL && h->t < distance;
}

/* ---- color_at / shade_hit / reflected_color, lines 1426-1451, 1485-1499 ---- */
static v4 color_at(world_t* w, const ray_t* r, int remaining);

static double schlick(const comps_t* c) { /* lines 1470-1483 */
  double cosine = v4_dot(c->eyev, c->normalv);
  if (c->n1 > c->n2) {
    double n = c->n1 / c->n2;
    double sin2_t = (n * n) * (1.0 - cosine * cosine);
    if (sin2_t > 1.0) return 1.0;
    double cos_t = sqrt(1.0 - sin2_t);
    cosine = cos_t;
  }
  double q = (c->n1 - c->n2) / (c->n1 + c->n2);
  double r0 = q * q;
  return r0 + (1.0 - r0) * pow((1.0 - cosine), 5);
}

static v4 shade_hit(world_t* w, const comps_t* c, int remaining) {
  const rt_material* m = material_of(w, c->id);
  const rt_material_ext* x = material_ext_of(w, m);
  const double transparency = x ? x->transparency : 0.0;
  int shadowed = is_shadowed(w, c->over_point);
  v4 surface = lighting(m, x, w->s->first_object_inv, &w->s->light, c->over_point, c->eyev, c->normalv, shadowed);
  v4 reflected;
  if (remaining < 1 || fabs(m->reflective - 0.0) < RTO_EPSILON) { /* equal(), line 262 */
    reflected = v4_point(0.0, 0.0, 0.0);
  } else {
    ray_t rr;
    rr.origin = v4_point(c->over_point.x, c->over_point.y, c->over_point.z);
    rr.direction = v4_vector(c->reflectv.x, c->reflectv.y, c->reflectv.z);
    ++w->rays_secondary;
    reflected = v4_scale(color_at(w, &rr, remaining - 1), m->reflective);
  }
  v4 refracted; /* refracted_color, lines 1453-1468 */
  {
    double n_ratio = c->n1 / c->n2;
    double cos_i = v4_dot(c->eyev, c->normalv);
    double sin2_t = (n_ratio * n_ratio) * (1.0 - cos_i * cos_i);
    if (remaining <= 0 || fabs(transparency - 0.0) < RTO_EPSILON || sin2_t > 1.0) {
      refracted = v4_point(0.0, 0.0, 0.0);
    } else {
      double cos_t = sqrt(1.0 - sin2_t);
      double n_ratio_cos = n_ratio * cos_i - cos_t;
      v4 eyev_n_ratio = v4_scale(c->eyev, n_ratio);
      v4 direction = v4_sub(v4_scale(c->normalv, n_ratio_cos), eyev_n_ratio);
      ray_t fr;
      fr.origin = v4_point(c->under_point.x, c->under_point.y, c->under_point.z);
      fr.direction = v4_vector(direction.x, direction.y, direction.z);
      ++w->rays_secondary;
      refracted = v4_scale(color_at(w, &fr, remaining - 1), transparency);
    }
  }
  if (m->reflective > 0.0 && transparency > 0.0) { /* lines 1494-1497 */
    double reflectance = schlick(c);
    return v4_add(surface, v4_add(v4_scale(reflected, reflectance), v4_scale(refracted, (1.0 - reflectance))));
  }
  return v4_add(v4_add(surface, reflected), refracted);
}

static v4 color_at(world_t* w, const ray_t* r, int remaining) {
  isect_list l;
  list_init(&l);
  isect_world(w, r, &l);
  const isect_t* h = list_hit(&l);
  if (h == NULL) return v4_point(0.0, 0.0, 0.0);
  comps_t c = prepare(w, h, r, &l);
  return shade_hit(w, &c, remaining);
}

/* ---- ray_for_pixel, lines 1535-1572 ---- */
static ray_t ray_for_pixel(const rt_camera* cam, double px, double py) {
  double x_offset = cam->pixel_size * (px + 0.5);
  double y_offset = cam->pixel_size * (py + 0.5);
  double world_x = cam->half_width - x_offset;
  double world_y = cam->half_height - y_offset;
  v4 pixel = mat_mul_v4(cam->inv_view, v4_point(world_x, world_y, -1));
  v4 origin = mat_mul_v4(cam->inv_view, v4_point(0.0, 0.0, 0.0));
  ray_t r;
  r.direction = v4_normalize(v4_sub(pixel, origin));
  r.origin = origin;
  return r;
}

int rto_color_convert(double x) { /* line 1322 */
  int color = (int)(x * 255);
  if (color < 0) color = 0;
  if (color > 255) color = 255;
  return color;
}

/* ------------------------------------------------------------- rendering -- */
static int rank_owns_row(const rt_render_params* p, int y) {
  if (p->n_ranks <= 1) return 1;
  return (y / p->band_rows) % p->n_ranks == p->rank;
}

static int check_params(const rt_scene* s, const rt_render_params* p) {
  if (!s || !p) return RT_ERR_INVALID;
  if (p->spp_grid < 1 || p->n_ranks < 1 || p->rank < 0 || p->rank >= p->n_ranks) return RT_ERR_INVALID;
  if (p->n_ranks > 1 && p->band_rows < 1) return RT_ERR_INVALID;
  if (s->camera.hsize < 1 || s->camera.vsize < 1) return RT_ERR_INVALID;
  return RT_OK;
}

int rto_render_f64(const rt_scene* s, const rt_render_params* p, double* canvas, rt_stats* stats) {
  int rc = check_params(s, p);
  if (rc != RT_OK || !canvas) return rc != RT_OK ? rc : RT_ERR_INVALID;
  world_t w;
  rc = world_open(&w, s);
  if (rc != RT_OK) return rc;
  memset(&g_probe, 0, sizeof g_probe);
  const int W = s->camera.hsize, H = s->camera.vsize, G = p->spp_grid;
  uint64_t primary = 0;
  int row = 0;
  for (int y = 0; y < H; ++y) { /* render(), lines 1576-1587 */
    if (!rank_owns_row(p, y)) continue;
    for (int x = 0; x < W; ++x) {
      double sx = 0.0, sy = 0.0, sz = 0.0;
      for (int j = 0; j < G; ++j)
        for (int i = 0; i < G; ++i) {
          double px = x + (i + 0.5) / G - 0.5; /* G == 1: px == x exactly */
          double py = y + (j + 0.5) / G - 0.5;
          ray_t r = ray_for_pixel(&s->camera, px, py);
          ++primary;
          v4 col = color_at(&w, &r, p->max_depth);
          sx += col.x; sy += col.y; sz += col.z;
        }
      double n = (double)(G * G);
      double* o = canvas + ((size_t)row * W + x) * 3;
      o[0] = sx / n; o[1] = sy / n; o[2] = sz / n;
    }
    ++row;
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    stats->rays_primary = primary;
    stats->rays_shadow = w.rays_shadow;
    stats->rays_secondary = w.rays_secondary;
    stats->rows_rendered = row;
    stats->accel_used = RT_ACCEL_BRUTE;
  }
  world_close(&w);
  return RT_OK;
}

int rto_render_rgb(const rt_scene* s, const rt_render_params* p, uint8_t* rgb, rt_stats* stats) {
  int rc = check_params(s, p);
  if (rc != RT_OK || !rgb) return rc != RT_OK ? rc : RT_ERR_INVALID;
  int rows = 0;
  for (int y = 0; y < s->camera.vsize; ++y) rows += rank_owns_row(p, y);
  size_t n = (size_t)rows * s->camera.hsize * 3;
  double* canvas = malloc(sizeof(double) * (n ? n : 1));
  if (!canvas) return RT_ERR_INVALID;
  rc = rto_render_f64(s, p, canvas, stats);
  if (rc == RT_OK) for (size_t k = 0; k < n; ++k) rgb[k] = (uint8_t)rto_color_convert(canvas[k]);
  free(canvas);
  return rc;
}

void rto_get_probe(rto_probe* out) { if (out) *out = g_probe; }

/* --------------------------------------------------- pieces for the KATs -- */
static ray_t ray_in(const rto_ray* r) {
  ray_t q;
  q.origin = v4_make(r->origin[0], r->origin[1], r->origin[2], r->origin[3]);
  q.direction = v4_make(r->direction[0], r->direction[1], r->direction[2], r->direction[3]);
  return q;
}

void rto_ray_for_pixel(const rt_camera* cam, double px, double py, rto_ray* out) {
  ray_t r = ray_for_pixel(cam, px, py);
  out->origin[0] = r.origin.x; out->origin[1] = r.origin.y; out->origin[2] = r.origin.z; out->origin[3] = r.origin.w;
  out->direction[0] = r.direction.x; out->direction[1] = r.direction.y; out->direction[2] = r.direction.z;
  out->direction[3] = r.direction.w;
}

int rto_intersect_sphere(const double inv[16], const rto_ray* r, double t_out[2]) {
  isect_list l;
  list_init(&l);
  ray_t q = ray_in(r);
  isect_sphere(inv, &q, &l, 0);
  for (int i = 0; i < l.count && i < 2; ++i) t_out[i] = l.e[i].t;
  return l.count;
}

int rto_intersect_triangle(const double p1[3], const double e1[3], const double e2[3], const rto_ray* r,
                           double* t_out) {
  ray_t q = ray_in(r);
  return isect_triangle_t(v4_point(p1[0], p1[1], p1[2]), v4_vector(e1[0], e1[1], e1[2]),
                          v4_vector(e2[0], e2[1], e2[2]), &q, t_out);
}

void rto_normal_at_sphere(const double inv[16], const double point[3], double n_out[3]) {
  v4 n = normal_at_sphere(inv, v4_point(point[0], point[1], point[2]));
  n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z;
}

void rto_reflect(const double in[3], const double n[3], double out[3]) {
  v4 r = v4_reflect(v4_vector(in[0], in[1], in[2]), v4_vector(n[0], n[1], n[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

void rto_lighting(const rt_material* m, const rt_light* light, const double point[3], const double eyev[3],
                  const double normalv[3], int in_shadow, double rgb_out[3]) {
  v4 c = lighting(m, NULL, NULL, light, v4_point(point[0], point[1], point[2]), v4_vector(eyev[0], eyev[1], eyev[2]),
                  v4_vector(normalv[0], normalv[1], normalv[2]), in_shadow);
  rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
}

int rto_hit_world(const rt_scene* scene, const rto_ray* r, double* t_out, int* n_intersections, double* ts_sorted,
                  int ts_cap) {
  world_t w;
  if (world_open(&w, scene) != RT_OK) return -2;
  isect_list l;
  list_init(&l);
  ray_t q = ray_in(r);
  isect_world(&w, &q, &l);
  if (n_intersections) *n_intersections = l.count;
  for (int i = 0; ts_sorted && i < l.count && i < ts_cap; ++i) ts_sorted[i] = l.e[i].t;
  const isect_t* h = list_hit(&l);
  int id = -1;
  if (h) { id = (int)h->id; if (t_out) *t_out = h->t; }
  world_close(&w);
  return id;
}

int rto_is_shadowed(const rt_scene* scene, const double point[3]) {
  world_t w;
  if (world_open(&w, scene) != RT_OK) return -2;
  int s = is_shadowed(&w, v4_point(point[0], point[1], point[2]));
  world_close(&w);
  return s;
}

void rto_color_at(const rt_scene* scene, const rto_ray* r, int remaining, double rgb_out[3]) {
  world_t w;
  rgb_out[0] = rgb_out[1] = rgb_out[2] = NAN;
  if (world_open(&w, scene) != RT_OK) return;
  ray_t q = ray_in(r);
  v4 c = color_at(&w, &q, remaining);
  rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
  world_close(&w);
}

int rto_prepare(const rt_scene* scene, const rto_ray* r, int order, double t, double out[16]) {
  world_t w;
  if (world_open(&w, scene) != RT_OK) return -2;
  if (order < 0 || order >= w.n) { world_close(&w); return -2; }
  ray_t q = ray_in(r);
  isect_t h = {t, order};
  comps_t c = prepare(&w, &h, &q, NULL);
  const v4* src[5] = {&c.point, &c.eyev, &c.normalv, &c.over_point, &c.reflectv};
  for (int k = 0; k < 5; ++k) { out[3 * k] = src[k]->x; out[3 * k + 1] = src[k]->y; out[3 * k + 2] = src[k]->z; }
  out[15] = c.inside;
  world_close(&w);
  return c.inside;
}

int rto_intersect_shape(const rt_shape* sp, const rto_ray* r, double* ts, int cap) {
  isect_list l;
  list_init(&l);
  ray_t q = ray_in(r);
  isect_shape(sp, &q, &l, 0);
  for (int i = 0; i < l.count && i < cap; ++i) ts[i] = l.e[i].t;
  return l.count;
}

void rto_normal_at_shape(const rt_shape* sp, const double point[3], double n_out[4]) {
  v4 n = normal_at_shape(sp, v4_point(point[0], point[1], point[2]));
  n_out[0] = n.x; n_out[1] = n.y; n_out[2] = n.z; n_out[3] = n.w;
}

void rto_pattern_at(const rt_material_ext* x, const double point[3], double rgb_out[3]) {
  v4 c = pattern_at(x, v4_point(point[0], point[1], point[2]));
  rgb_out[0] = c.x; rgb_out[1] = c.y; rgb_out[2] = c.z;
}

double rto_schlick(const double eyev[3], const double normalv[3], double n1, double n2) {
  comps_t c;
  memset(&c, 0, sizeof c);
  c.eyev = v4_vector(eyev[0], eyev[1], eyev[2]);
  c.normalv = v4_vector(normalv[0], normalv[1], normalv[2]);
  c.n1 = n1;
  c.n2 = n2;
  return schlick(&c);
}

/* n1 and n2 of the hit at sorted position `hit_index` of a ray through the world (prepare_computations, :1378-1406) */
int rto_refractive_indices(const rt_scene* scene, const rto_ray* r, int hit_index, double n_out[2]) {
  world_t w;
  if (world_open(&w, scene) != RT_OK) return -2;
  isect_list l;
  list_init(&l);
  ray_t q = ray_in(r);
  isect_world(&w, &q, &l);
  if (hit_index < 0 || hit_index >= l.count) { world_close(&w); return -1; }
  comps_t c = prepare(&w, &l.e[hit_index], &q, &l);
  n_out[0] = c.n1; n_out[1] = c.n2;
  world_close(&w);
  return l.count;
}

int rto_hit_list(const double* ts, int n) {
  isect_list l;
  list_init(&l);
  for (int i = 0; i < n; ++i) list_add(&l, ts[i], i);
  list_sort(&l);
  const isect_t* h = list_hit(&l);
  return h ? (int)h->id : -1;
}

int rto_list_capacity(int attempts) {
  isect_list l;
  list_init(&l);
  int ok = 0;
  for (int i = 0; i < attempts; ++i) ok += list_add(&l, 1.0, i);
  return ok;
}

int rto_write_ppm(const char* path, const uint8_t* rgb, int width, int height) { /* lines 1333-1348 */
  FILE* fp = fopen(path, "w");
  if (!fp) return RT_ERR_INVALID;
  fprintf(fp, "P3\n");
  fprintf(fp, "%d %d\n255\n", width, height);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const uint8_t* p = rgb + ((size_t)y * width + x) * 3;
      fprintf(fp, "%d ", p[0]);
      fprintf(fp, "%d ", p[1]);
      fprintf(fp, "%d \n", p[2]);
    }
  fclose(fp);
  return RT_OK;
}
