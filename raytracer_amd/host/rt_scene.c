/*
 * rt_scene.c - host side of the drop-in boundary: scene description ->
 * flat rt_scene, in the reference's arithmetic (see include/rt_scene.h for the
 * function-by-function map onto RayTracer/RayTracer.c).
 *
 * Every floating-point expression below keeps the reference's operation
 * order (C evaluates a + b + c as (a + b) + c; the reference binary is
 * x86-64 SSE2 without FMA), because the values computed here are the ones the
 * reference recomputes inside its per-ray loop.  Build with
 * -ffp-contract=off.
 */
#include "rt_scene.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_bvh.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------- matrices -- */
void rt_mat4_identity(double m[16]) {
  for (int i = 0; i < 16; ++i) m[i] = 0.0;
  m[0] = m[5] = m[10] = m[15] = 1.0;
}
void rt_mat4_translate(double x, double y, double z, double m[16]) {
  rt_mat4_identity(m);
  m[3] = x; m[7] = y; m[11] = z;
}
void rt_mat4_scale(double x, double y, double z, double m[16]) {
  rt_mat4_identity(m);
  m[0] = x; m[5] = y; m[10] = z;
}
void rt_mat4_rotate_x(double rad, double m[16]) {
  rt_mat4_identity(m);
  m[5] = cos(rad); m[6] = -sin(rad);
  m[9] = sin(rad); m[10] = cos(rad);
}
void rt_mat4_rotate_y(double rad, double m[16]) {
  rt_mat4_identity(m);
  m[0] = cos(rad); m[2] = sin(rad);
  m[8] = -sin(rad); m[10] = cos(rad);
}
void rt_mat4_rotate_z(double rad, double m[16]) {
  rt_mat4_identity(m);
  m[0] = cos(rad); m[1] = -sin(rad);
  m[4] = sin(rad); m[5] = cos(rad);
}
void rt_mat4_mul(const double a[16], const double b[16], double out[16]) {
  double t[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      t[r * 4 + c] = a[r * 4 + 0] * b[0 * 4 + c] + a[r * 4 + 1] * b[1 * 4 + c] + a[r * 4 + 2] * b[2 * 4 + c] +
                     a[r * 4 + 3] * b[3 * 4 + c];
  memcpy(out, t, sizeof t);
}

/* determinant of the 3x3 left after deleting row `dr` and column `dc`:
 * mat4x4_submat3x3 + mat3x3_det, RayTracer.c:548, 524 */
static double minor3(const double a[16], int dr, int dc) {
  double m[3][3];
  int rr = 0;
  for (int i = 0; i < 4; ++i) {
    if (i == dr) continue;
    int cc = 0;
    for (int j = 0; j < 4; ++j) {
      if (j == dc) continue;
      m[rr][cc++] = a[i * 4 + j];
    }
    ++rr;
  }
  double p = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]);
  double q = m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]);
  double r = m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  return p - q + r;
}
static double cofactor4(const double a[16], int r, int c) { /* RayTracer.c:581 */
  double v = minor3(a, r, c);
  if ((r + c) % 2 != 0) v *= -1;
  return v;
}
static double det4(const double a[16]) { /* RayTracer.c:587 */
  double d = 0.0;
  for (int c = 0; c < 4; ++c) d = d + a[c] * cofactor4(a, 0, c);
  return d;
}
int rt_mat4_inverse(const double a[16], double out[16]) {
  double d = det4(a);
  if (fabs(d - 0.0) < 0.000001) return 0; /* equal(det, 0), RayTracer.c:597 */
  double t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) t[j * 4 + i] = cofactor4(a, i, j) / d;
  memcpy(out, t, sizeof t);
  return 1;
}

/* 3-vectors with the w lane the reference carries written out where it is not
 * identically zero */
static double mag3(const double v[3]) { /* tuple_mag_vec with w == 0, RayTracer.c:335 */
  return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + 0.0 * 0.0);
}
static void normalize3(const double v[3], double o[3]) {
  double m = mag3(v);
  o[0] = v[0] / m; o[1] = v[1] / m; o[2] = v[2] / m;
}
static void cross3(const double a[3], const double b[3], double o[3]) {
  double x = a[1] * b[2] - a[2] * b[1];
  double y = a[2] * b[0] - a[0] * b[2];
  double z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

void rt_view_transform(const double from[3], const double to[3], const double up[3], double m[16]) {
  double d[3] = {to[0] - from[0], to[1] - from[1], to[2] - from[2]};
  double forward[3], upn[3], left[3], true_up[3];
  normalize3(d, forward);
  normalize3(up, upn);
  cross3(forward, upn, left);
  cross3(left, forward, true_up);
  double o[16] = {left[0],     left[1],     left[2],     0.0,
                  true_up[0],  true_up[1],  true_up[2],  0.0,
                  -forward[0], -forward[1], -forward[2], 0.0,
                  0.0,         0.0,         0.0,         1.0};
  double t[16];
  rt_mat4_translate(-from[0], -from[1], -from[2], t);
  rt_mat4_mul(o, t, m);
}

rt_material rt_material_default(void) {
  rt_material m;
  m.color[0] = m.color[1] = m.color[2] = 1.0;
  m.ambient = 0.1;
  m.diffuse = 0.9;
  m.specular = 0.9;
  m.shininess = 200.0;
  m.reflective = 0.0;
  return m;
}

/* -------------------------------------------------------------- builder -- */
typedef struct { double transform[16]; int material; int order; } sphere_rec;
typedef struct { double p1[3], p2[3], p3[3]; int material; int order; } tri_rec;
typedef struct { double transform[16]; double minimum, maximum; int type, closed, material, order; } shape_rec;
/* the pattern half of a material as the caller gives it (transform, not its inverse) */
typedef struct { rt_material_ext x; double pattern_transform[16]; } ext_rec;

struct rt_scene_builder {
  rt_camera camera;
  int have_camera, have_light;
  rt_light light;
  sphere_rec* spheres; int n_spheres, cap_spheres;
  tri_rec* tris; int n_tris, cap_tris;
  shape_rec* shapes; int n_shapes, cap_shapes;
  rt_material* materials; int n_materials, cap_materials;
  ext_rec* exts; int cap_exts;      /* parallel to materials */
  int any_ext;                      /* some material has a pattern or transparency / refractive index */
  int n_objects;                    /* length of the world list so far */
  double first_transform[16];       /* transform of the world's first object */
  /* outputs of rt_scene_finish */
  rt_shape* out_shapes; rt_material_ext* out_exts;
  double* sph_inv; int32_t* sph_material; int32_t* sph_order;
  double* tri_soa; int32_t* tri_material; int32_t* tri_order;
  rt_bvh_node* bvh_nodes; int32_t* bvh_tri; int n_bvh_nodes;
};

rt_scene_builder* rt_scene_create(void) { return calloc(1, sizeof(rt_scene_builder)); }

static void free_outputs(rt_scene_builder* b) {
  free(b->sph_inv); free(b->sph_material); free(b->sph_order);
  free(b->tri_soa); free(b->tri_material); free(b->tri_order);
  free(b->bvh_nodes); free(b->bvh_tri);
  free(b->out_shapes); free(b->out_exts);
  b->out_shapes = NULL; b->out_exts = NULL;
  b->sph_inv = b->tri_soa = NULL;
  b->sph_material = b->sph_order = b->tri_material = b->tri_order = b->bvh_tri = NULL;
  b->bvh_nodes = NULL;
  b->n_bvh_nodes = 0;
}

void rt_scene_destroy(rt_scene_builder* b) {
  if (!b) return;
  free_outputs(b);
  free(b->spheres); free(b->tris); free(b->shapes); free(b->materials); free(b->exts);
  free(b);
}

static int grow(void** p, int* cap, int need, size_t elem) {
  if (need <= *cap) return 1;
  int nc = *cap ? *cap * 2 : 64;
  while (nc < need) nc *= 2;
  void* q = realloc(*p, (size_t)nc * elem);
  if (!q) return 0;
  *p = q; *cap = nc;
  return 1;
}

rt_material_ext rt_material_ext_default(void) { /* create_material_default, RayTracer.c:712-717 */
  rt_material_ext x;
  memset(&x, 0, sizeof x);
  x.transparency = 0.0;
  x.refractive_index = 1.0;
  x.pattern_to[0] = x.pattern_to[1] = x.pattern_to[2] = 1.0;
  rt_mat4_identity(x.pattern_inv);
  x.pattern_type = RT_PATTERN_NONE;
  return x;
}

/* `x` NULL = default (opaque, no pattern); pattern_transform NULL = identity (the pattern constructors, :1209-1257) */
static int intern_material_ext(rt_scene_builder* b, const rt_material* m, const rt_material_ext* x,
                               const double* pattern_transform) {
  rt_material d = rt_material_default();
  if (!m) m = &d;
  ext_rec e;
  memset(&e, 0, sizeof e);
  e.x = x ? *x : rt_material_ext_default();
  e.x.reserved = 0;
  if (pattern_transform) memcpy(e.pattern_transform, pattern_transform, sizeof e.pattern_transform);
  else rt_mat4_identity(e.pattern_transform);
  rt_mat4_identity(e.x.pattern_inv); /* derived in rt_scene_finish */
  for (int i = b->n_materials - 1; i >= 0 && i >= b->n_materials - 8; --i)
    if (memcmp(&b->materials[i], m, sizeof *m) == 0 && memcmp(&b->exts[i], &e, sizeof e) == 0) return i;
  if (!grow((void**)&b->materials, &b->cap_materials, b->n_materials + 1, sizeof *m)) return -1;
  if (!grow((void**)&b->exts, &b->cap_exts, b->n_materials + 1, sizeof e)) return -1;
  b->materials[b->n_materials] = *m;
  b->exts[b->n_materials] = e;
  return b->n_materials++;
}

/* position in the world list of the object being appended (add_shape_to_world, RayTracer.c:1078) */
static int next_order(rt_scene_builder* b, const double* transform) {
  if (b->n_objects == 0) {
    if (transform) memcpy(b->first_transform, transform, sizeof b->first_transform);
    else rt_mat4_identity(b->first_transform); /* a triangle keeps create_shape's identity, :976 */
  }
  return b->n_objects++;
}

int rt_scene_set_camera_view(rt_scene_builder* b, int hsize, int vsize, double fov, const double view[16]) {
  if (!b || !view || hsize < 1 || vsize < 1) return RT_ERR_INVALID;
  /* create_camera, RayTracer.c:1135-1148; hsize and vsize are doubles there */
  double hs = hsize, vs = vsize;
  double half_view = tan(fov / 2.0);
  double half_height, half_width;
  double aspect = hs / vs;
  if (aspect >= 1) { half_width = half_view; half_height = half_view / aspect; }
  else { half_width = half_view * aspect; half_height = half_view; }
  b->camera.hsize = hsize;
  b->camera.vsize = vsize;
  b->camera.pixel_size = (half_width * 2.0) / hs;
  b->camera.half_width = half_width;
  b->camera.half_height = half_height;
  if (!rt_mat4_inverse(view, b->camera.inv_view)) return RT_ERR_UNSUPPORTED;
  b->have_camera = 1;
  return RT_OK;
}

int rt_scene_set_camera(rt_scene_builder* b, int hsize, int vsize, double fov, const double from[3],
                        const double to[3], const double up[3]) {
  if (!b || !from || !to || !up) return RT_ERR_INVALID;
  double view[16];
  rt_view_transform(from, to, up, view);
  return rt_scene_set_camera_view(b, hsize, vsize, fov, view);
}

int rt_scene_set_light(rt_scene_builder* b, const double position[3], const double intensity[3]) {
  if (!b || !position || !intensity) return RT_ERR_INVALID;
  memcpy(b->light.position, position, sizeof b->light.position);
  memcpy(b->light.intensity, intensity, sizeof b->light.intensity);
  b->have_light = 1;
  return RT_OK;
}

int rt_scene_add_sphere_ext(rt_scene_builder* b, const double transform[16], const rt_material* m,
                            const rt_material_ext* x, const double* pattern_transform) {
  if (!b || !transform) return RT_ERR_INVALID;
  if (!grow((void**)&b->spheres, &b->cap_spheres, b->n_spheres + 1, sizeof(sphere_rec))) return RT_ERR_INVALID;
  int mi = intern_material_ext(b, m, x, pattern_transform);
  if (mi < 0) return RT_ERR_INVALID;
  sphere_rec* s = &b->spheres[b->n_spheres++];
  memcpy(s->transform, transform, sizeof s->transform);
  s->material = mi;
  s->order = next_order(b, transform);
  return s->order;
}

int rt_scene_add_sphere(rt_scene_builder* b, const double transform[16], const rt_material* m) {
  return rt_scene_add_sphere_ext(b, transform, m, NULL, NULL);
}

int rt_scene_add_shape(rt_scene_builder* b, int type, const double transform[16], const rt_material* m,
                       const rt_material_ext* x, const double* pattern_transform, double minimum, double maximum,
                       int closed) {
  if (!b || !transform) return RT_ERR_INVALID;
  if (type != RT_SHAPE_PLANE && type != RT_SHAPE_CUBE && type != RT_SHAPE_CYLINDER) return RT_ERR_INVALID;
  if (!grow((void**)&b->shapes, &b->cap_shapes, b->n_shapes + 1, sizeof(shape_rec))) return RT_ERR_INVALID;
  int mi = intern_material_ext(b, m, x, pattern_transform);
  if (mi < 0) return RT_ERR_INVALID;
  shape_rec* s = &b->shapes[b->n_shapes++];
  memcpy(s->transform, transform, sizeof s->transform);
  s->type = type;
  s->minimum = minimum;
  s->maximum = maximum;
  s->closed = closed != 0;
  s->material = mi;
  s->order = next_order(b, transform);
  return s->order;
}

int rt_scene_add_triangle(rt_scene_builder* b, const double p1[3], const double p2[3], const double p3[3],
                          const rt_material* m) {
  return rt_scene_add_triangle_ext(b, p1, p2, p3, m, NULL, NULL);
}

int rt_scene_add_triangle_ext(rt_scene_builder* b, const double p1[3], const double p2[3], const double p3[3],
                              const rt_material* m, const rt_material_ext* x, const double* pattern_transform) {
  if (!b || !p1 || !p2 || !p3) return RT_ERR_INVALID;
  if (!grow((void**)&b->tris, &b->cap_tris, b->n_tris + 1, sizeof(tri_rec))) return RT_ERR_INVALID;
  int mi = intern_material_ext(b, m, x, pattern_transform);
  if (mi < 0) return RT_ERR_INVALID;
  tri_rec* t = &b->tris[b->n_tris++];
  memcpy(t->p1, p1, sizeof t->p1);
  memcpy(t->p2, p2, sizeof t->p2);
  memcpy(t->p3, p3, sizeof t->p3);
  t->material = mi;
  t->order = next_order(b, NULL);
  return t->order;
}

int rt_scene_add_triangles_f32(rt_scene_builder* b, const float* xyz9, int n, const rt_material* m) {
  if (!b || (!xyz9 && n > 0) || n < 0) return RT_ERR_INVALID;
  for (int i = 0; i < n; ++i) {
    const float* f = xyz9 + 9 * (size_t)i;
    double p1[3] = {f[0], f[1], f[2]}, p2[3] = {f[3], f[4], f[5]}, p3[3] = {f[6], f[7], f[8]};
    int rc = rt_scene_add_triangle(b, p1, p2, p3, m);
    if (rc < 0) return rc;
  }
  return RT_OK;
}

int rt_scene_set_reflective(rt_scene_builder* b, double tri_reflective, double sph_reflective) {
  if (!b) return RT_ERR_INVALID;
  /* materials are shared by index: give every primitive class its own copies */
  for (int i = 0; i < b->n_tris; ++i) {
    rt_material m = b->materials[b->tris[i].material];
    const ext_rec e = b->exts[b->tris[i].material];
    m.reflective = tri_reflective;
    int mi = intern_material_ext(b, &m, &e.x, e.pattern_transform);
    if (mi < 0) return RT_ERR_INVALID;
    b->tris[i].material = mi;
  }
  for (int i = 0; i < b->n_spheres; ++i) {
    rt_material m = b->materials[b->spheres[i].material];
    const ext_rec e = b->exts[b->spheres[i].material];
    m.reflective = sph_reflective;
    int mi = intern_material_ext(b, &m, &e.x, e.pattern_transform);
    if (mi < 0) return RT_ERR_INVALID;
    b->spheres[i].material = mi;
  }
  return RT_OK;
}

/* ------------------------------------------------------------------ OBJ -- */
/* Positions go through (float)atof like objpar.h does (its objpar_atof), and
 * are widened back to double by the caller, RayTracer.c:1657-1669. */
int rt_scene_load_obj(rt_scene_builder* b, const char* text, size_t len, const rt_material* m) {
  if (!b || !text) return RT_ERR_INVALID;
  float* pos = NULL;
  int npos = 0, cap = 0;
  int rc = RT_OK;
  size_t i = 0;
  char tok[64];
  while (i < len) {
    size_t e = i;
    while (e < len && text[e] != '\n') ++e;
    if (e - i > 2 && text[i] == 'v' && (text[i + 1] == ' ' || text[i + 1] == '\t')) {
      float v[3] = {0, 0, 0};
      size_t p = i + 1;
      for (int k = 0; k < 3; ++k) {
        while (p < e && (text[p] == ' ' || text[p] == '\t')) ++p;
        size_t q = p;
        while (q < e && text[q] != ' ' && text[q] != '\t' && text[q] != '\r') ++q;
        size_t n = q - p < sizeof tok - 1 ? q - p : sizeof tok - 1;
        memcpy(tok, text + p, n); tok[n] = 0;
        v[k] = (float)atof(tok);
        p = q;
      }
      if (!grow((void**)&pos, &cap, 3 * (npos + 1), sizeof(float))) { rc = RT_ERR_INVALID; break; }
      pos[3 * npos] = v[0]; pos[3 * npos + 1] = v[1]; pos[3 * npos + 2] = v[2];
      ++npos;
    } else if (e - i > 2 && text[i] == 'f' && (text[i + 1] == ' ' || text[i + 1] == '\t')) {
      int idx[4], n = 0;
      size_t p = i + 1;
      while (p < e && n < 4) {
        while (p < e && (text[p] == ' ' || text[p] == '\t')) ++p;
        if (p >= e || text[p] == '\r') break;
        /* leading integer of v, v/vt, v//vn or v/vt/vn, read inside the line (the buffer need not be terminated) */
        size_t q = p;
        while (q < e && text[q] != ' ' && text[q] != '\t' && text[q] != '\r') ++q;
        size_t nt = q - p < sizeof tok - 1 ? q - p : sizeof tok - 1;
        memcpy(tok, text + p, nt); tok[nt] = 0;
        idx[n++] = atoi(tok);
        p = q;
      }
      if (n < 3) { rc = RT_ERR_UNSUPPORTED; break; }
      for (int k = 0; k < n; ++k)
        if (idx[k] < 1 || idx[k] > npos) rc = RT_ERR_INVALID; /* 1-based, already seen */
      if (rc != RT_OK) break;
      for (int t = 0; t + 2 < n; ++t) { /* (1,2,3) then (1,3,4) */
        const float* a = pos + 3 * (idx[0] - 1);
        const float* c = pos + 3 * (idx[t + 1] - 1);
        const float* d = pos + 3 * (idx[t + 2] - 1);
        double p1[3] = {a[0], a[1], a[2]}, p2[3] = {c[0], c[1], c[2]}, p3[3] = {d[0], d[1], d[2]};
        int r = rt_scene_add_triangle(b, p1, p2, p3, m);
        if (r < 0) { rc = r; break; }
      }
      if (rc != RT_OK) break;
    }
    i = e + 1;
  }
  free(pos);
  return rc;
}

int rt_scene_load_obj_file(rt_scene_builder* b, const char* path, const rt_material* m) {
  if (!b || !path) return RT_ERR_INVALID;
  FILE* f = fopen(path, "rb");
  if (!f) return RT_ERR_INVALID;
  long n = -1;
  if (fseek(f, 0, SEEK_END) == 0) n = ftell(f);
  if (n < 0 || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return RT_ERR_INVALID; }
  char* buf = malloc((size_t)n + 1);
  if (!buf) { fclose(f); return RT_ERR_INVALID; }
  size_t got = fread(buf, 1, (size_t)n, f);
  fclose(f);
  buf[got] = 0;
  int rc = rt_scene_load_obj(b, buf, got, m);
  free(buf);
  return rc;
}

/* --------------------------------------------------------------- finish -- */
int rt_scene_finish(rt_scene_builder* b, int bvh_leaf_size, rt_scene* out) {
  if (!b || !out) return RT_ERR_INVALID;
  if (!b->have_camera || !b->have_light) return RT_ERR_NOT_READY;
  if (bvh_leaf_size > 7) return RT_ERR_TOO_LARGE; /* the device's leaf encoding holds up to 7 triangles */
  free_outputs(b);
  memset(out, 0, sizeof *out);
  const int ns = b->n_spheres, nt = b->n_tris;
  b->sph_inv = malloc(sizeof(double) * 16 * (size_t)(ns ? ns : 1));
  b->sph_material = malloc(sizeof(int32_t) * (size_t)(ns ? ns : 1));
  b->sph_order = malloc(sizeof(int32_t) * (size_t)(ns ? ns : 1));
  b->tri_soa = malloc(sizeof(double) * 12 * (size_t)(nt ? nt : 1));
  b->tri_material = malloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
  b->tri_order = malloc(sizeof(int32_t) * (size_t)(nt ? nt : 1));
  if (!b->sph_inv || !b->sph_material || !b->sph_order || !b->tri_soa || !b->tri_material || !b->tri_order)
    return RT_ERR_INVALID;

  for (int i = 0; i < ns; ++i) {
    double* inv = b->sph_inv + 16 * (size_t)i;
    if (!rt_mat4_inverse(b->spheres[i].transform, inv)) return RT_ERR_UNSUPPORTED;
    /* the kernels drop the w lane: the transform must be affine */
    if (inv[12] != 0.0 || inv[13] != 0.0 || inv[14] != 0.0 || inv[15] != 1.0) return RT_ERR_UNSUPPORTED;
    b->sph_material[i] = b->spheres[i].material;
    b->sph_order[i] = b->spheres[i].order;
  }
  for (int i = 0; i < nt; ++i) {
    const tri_rec* t = &b->tris[i];
    double e1[3], e2[3], c[3], n[3];
    for (int k = 0; k < 3; ++k) { e1[k] = t->p2[k] - t->p1[k]; e2[k] = t->p3[k] - t->p1[k]; } /* :1004-1005 */
    cross3(e2, e1, c);                                                                          /* :1006 */
    normalize3(c, n);
    for (int k = 0; k < 3; ++k) {
      b->tri_soa[(size_t)(0 + k) * nt + i] = t->p1[k];
      b->tri_soa[(size_t)(3 + k) * nt + i] = e1[k];
      b->tri_soa[(size_t)(6 + k) * nt + i] = e2[k];
      b->tri_soa[(size_t)(9 + k) * nt + i] = n[k];
    }
    b->tri_material[i] = t->material;
    b->tri_order[i] = t->order;
  }

  out->camera = b->camera;
  out->light = b->light;
  out->n_materials = b->n_materials;
  out->materials = b->materials;
  out->n_spheres = ns;
  out->sph_inv = b->sph_inv;
  out->sph_material = b->sph_material;
  out->sph_order = b->sph_order;
  out->n_tris = nt;
  out->tri_soa = b->tri_soa;
  out->tri_material = b->tri_material;
  out->tri_order = b->tri_order;

  /* planes, cubes, cylinders; patterns and transparency */
  if (b->n_shapes > 0) {
    b->out_shapes = malloc(sizeof(rt_shape) * (size_t)b->n_shapes);
    if (!b->out_shapes) return RT_ERR_INVALID;
    for (int i = 0; i < b->n_shapes; ++i) {
      const shape_rec* r = &b->shapes[i];
      rt_shape* o = &b->out_shapes[i];
      memset(o, 0, sizeof *o);
      rt_mat4_identity(o->inv); /* mat4x4_set_ident before mat4x4_inverse, RayTracer.c:899-900 */
      if (!rt_mat4_inverse(r->transform, o->inv)) return RT_ERR_UNSUPPORTED;
      o->minimum = r->minimum;
      o->maximum = r->maximum;
      o->type = r->type;
      o->closed = r->closed;
      o->material = r->material;
      o->order = r->order;
    }
    out->n_shapes = b->n_shapes;
    out->shapes = b->out_shapes;
  }
  /* the second half of the materials travels only when some material needs it (the fast kernels do not read it) */
  b->any_ext = 0;
  for (int i = 0; i < b->n_materials; ++i) {
    const rt_material_ext* x = &b->exts[i].x;
    if (x->pattern_type != RT_PATTERN_NONE || x->transparency != 0.0 || x->refractive_index != 1.0) b->any_ext = 1;
  }
  if (b->any_ext && b->n_materials > 0) {
    b->out_exts = malloc(sizeof(rt_material_ext) * (size_t)b->n_materials);
    if (!b->out_exts) return RT_ERR_INVALID;
    for (int i = 0; i < b->n_materials; ++i) {
      b->out_exts[i] = b->exts[i].x;
      if (!rt_mat4_inverse(b->exts[i].pattern_transform, b->out_exts[i].pattern_inv)) return RT_ERR_UNSUPPORTED;
    }
    out->materials_ext = b->out_exts;
  }
  rt_mat4_identity(out->first_object_inv);
  if (b->n_objects > 0 && !rt_mat4_inverse(b->first_transform, out->first_object_inv)) return RT_ERR_UNSUPPORTED;

  if (bvh_leaf_size > 0 && nt > 0) {
    double* p123 = malloc(sizeof(double) * 9 * (size_t)nt);
    if (!p123) return RT_ERR_INVALID;
    for (int i = 0; i < nt; ++i) {
      memcpy(p123 + 9 * (size_t)i, b->tris[i].p1, 3 * sizeof(double));
      memcpy(p123 + 9 * (size_t)i + 3, b->tris[i].p2, 3 * sizeof(double));
      memcpy(p123 + 9 * (size_t)i + 6, b->tris[i].p3, 3 * sizeof(double));
    }
    int rc = rt_bvh_build(p123, nt, bvh_leaf_size, &b->bvh_nodes, &b->n_bvh_nodes, &b->bvh_tri);
    free(p123);
    if (rc != RT_OK) return rc;
    out->n_bvh_nodes = b->n_bvh_nodes;
    out->bvh_nodes = b->bvh_nodes;
    out->bvh_tri = b->bvh_tri;
  }
  return RT_OK;
}

/* ---------------------------------------------------------- stock scenes -- */
/* The mesh the reference ships in teapot.h (live definition, teapot.h:35-63):
 * 8 vertices, 12 triangles.  Decimal text there goes through (float)atof. */
static const char* const k_teapot_h_vertices[8][3] = {
    {"1.000000", "-1.000000", "-1.000000"},  {"1.000000", "-1.000000", "1.000000"},
    {"-1.000000", "-1.000000", "1.000000"},  {"-1.000000", "-1.000000", "-1.000000"},
    {"1.000000", "1.000000", "-0.999999"},   {"0.999999", "1.000000", "1.000001"},
    {"-1.000000", "1.000000", "1.000000"},   {"-1.000000", "1.000000", "-1.000000"}};
static const int k_teapot_h_faces[12][3] = {{2, 4, 1}, {8, 6, 5}, {5, 2, 1}, {6, 3, 2}, {3, 8, 4}, {1, 8, 5},
                                            {2, 3, 4}, {8, 7, 6}, {5, 6, 2}, {6, 7, 3}, {3, 7, 8}, {1, 4, 8}};

static int add_teapot_h_mesh(rt_scene_builder* b) {
  for (int f = 0; f < 12; ++f) {
    double p[3][3];
    for (int k = 0; k < 3; ++k)
      for (int c = 0; c < 3; ++c) p[k][c] = (float)atof(k_teapot_h_vertices[k_teapot_h_faces[f][k] - 1][c]);
    int rc = rt_scene_add_triangle(b, p[0], p[1], p[2], NULL);
    if (rc < 0) return rc;
  }
  return RT_OK;
}

static int add_complete_world_spheres(rt_scene_builder* b) { /* RayTracer.c:5852-6163 */
  double m[16], t[16], ry[16], rx[16], sc[16], id[16];
  rt_material mat;
  int rc;

  /* floor: scale(10, 0.01, 10) * identity, matte grey */
  rt_mat4_scale(10.0, 0.01, 10.0, sc);
  rt_mat4_identity(id);
  rt_mat4_mul(sc, id, m);
  mat = rt_material_default();
  mat.color[0] = mat.color[1] = mat.color[2] = 0.9;
  mat.specular = 0.0;
  if ((rc = rt_scene_add_sphere(b, m, &mat)) < 0) return rc;

  /* left wall: ends up with the BLUE matte material (RayTracer.c:6034, 6044) */
  rt_mat4_translate(0.0, 0.0, 5.0, t);
  rt_mat4_rotate_y(-M_PI / 4.0, ry);
  rt_mat4_rotate_x(M_PI / 2.0, rx);
  rt_mat4_identity(m);
  rt_mat4_mul(t, m, m);
  rt_mat4_mul(m, ry, m);
  rt_mat4_mul(m, rx, m);
  rt_mat4_mul(m, sc, m);
  mat.color[0] = 0.0; mat.color[1] = 0.0; mat.color[2] = 1.0;
  if ((rc = rt_scene_add_sphere(b, m, &mat)) < 0) return rc;

  /* right wall: keeps the default material */
  rt_mat4_rotate_y(M_PI / 4.0, ry);
  rt_mat4_identity(m);
  rt_mat4_mul(m, t, m);
  rt_mat4_mul(m, ry, m);
  rt_mat4_mul(m, rx, m);
  rt_mat4_mul(m, sc, m);
  if ((rc = rt_scene_add_sphere(b, m, NULL)) < 0) return rc;

  /* middle sphere */
  rt_mat4_translate(-0.5, 1.0, 0.5, m);
  mat = rt_material_default();
  mat.color[0] = 0.1; mat.color[1] = 1.0; mat.color[2] = 0.5;
  mat.diffuse = 0.7; mat.specular = 0.3;
  if ((rc = rt_scene_add_sphere(b, m, &mat)) < 0) return rc;

  /* right sphere */
  rt_mat4_translate(1.5, 0.5, -0.5, t);
  rt_mat4_scale(0.5, 0.5, 0.5, sc);
  rt_mat4_identity(m);
  rt_mat4_mul(m, t, m);
  rt_mat4_mul(m, sc, m);
  mat.color[0] = 0.5; mat.color[1] = 1.0; mat.color[2] = 0.1;
  if ((rc = rt_scene_add_sphere(b, m, &mat)) < 0) return rc;

  /* small sphere; 0.8f is a float literal in the reference (RayTracer.c:6151) */
  rt_mat4_translate(-1.5, 0.33, -0.75, t);
  rt_mat4_scale(0.33, 0.33, 0.33, sc);
  rt_mat4_identity(m);
  rt_mat4_mul(m, t, m);
  rt_mat4_mul(m, sc, m);
  mat.color[0] = 1.0; mat.color[1] = 0.8f; mat.color[2] = 0.1;
  if ((rc = rt_scene_add_sphere(b, m, &mat)) < 0) return rc;
  return RT_OK;
}

static rt_material_ext checkers(double from, double to) { /* checkers_pattern, RayTracer.c:1249 */
  rt_material_ext x = rt_material_ext_default();
  x.pattern_type = RT_PATTERN_CHECKER;
  x.pattern_from[0] = x.pattern_from[1] = x.pattern_from[2] = from;
  x.pattern_to[0] = x.pattern_to[1] = x.pattern_to[2] = to;
  return x;
}

/* T * S the way the scene functions build it: identity * translate, then * scale (e.g. RayTracer.c:6332-6335) */
static void ident_translate_scale(double tx, double ty, double tz, double sx, double sy, double sz, double m[16]) {
  double t[16], sc[16];
  rt_mat4_translate(tx, ty, tz, t);
  rt_mat4_scale(sx, sy, sz, sc);
  rt_mat4_identity(m);
  rt_mat4_mul(m, t, m);
  rt_mat4_mul(m, sc, m);
}

static int add_world_with_plane(rt_scene_builder* b) { /* render_complete_world_with_plane, RayTracer.c:6227-6396 */
  double m[16], t[16], rot[16], pt[16];
  int rc;
  const double ninf = -INFINITY, pinf = INFINITY;
  rt_material fm = rt_material_default(); /* floor_material, reused for the walls with its colour changed */
  fm.color[0] = fm.color[1] = fm.color[2] = 1.0;
  fm.specular = 0.0;
  fm.reflective = 0.0;
  rt_material_ext fx = checkers(0.15, 0.85);
  rt_mat4_identity(m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, m, &fm, &fx, NULL, ninf, pinf, 0)) < 0) return rc; /* floor */
  rt_mat4_rotate_z(M_PI / 2.0, rot);
  fm.color[0] = 0.0; fm.color[1] = 0.0; fm.color[2] = 1.0;
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, rot, &fm, &fx, NULL, ninf, pinf, 0)) < 0) return rc; /* right wall */
  fm.color[0] = 1.0; fm.color[1] = 0.0; fm.color[2] = 0.0;
  rt_mat4_rotate_x(-M_PI / 2.0, rot);
  rt_mat4_translate(0.0, 0.0, -2.75, t);
  rt_mat4_mul(rot, t, rot);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, rot, &fm, &fx, NULL, ninf, pinf, 0)) < 0) return rc; /* left wall */

  /* glass sphere: create_glass_sphere (:1017) then :6274-6277 */
  rt_material gm = rt_material_default();
  rt_material_ext gx = rt_material_ext_default();
  gm.diffuse = 0.0;
  gx.refractive_index = 1.5;
  gx.transparency = 0.9;
  gm.reflective = 0.2;
  rt_mat4_translate(-7.0, 1.0, -8.0, m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &gm, &gx, NULL)) < 0) return rc;

  /* middle sphere: purple stripes scaled by 0.175 */
  rt_material mm = rt_material_default();
  mm.color[0] = 0.1; mm.color[1] = 1.0; mm.color[2] = 0.5;
  mm.diffuse = 0.7; mm.specular = 0.3;
  rt_material_ext mx = rt_material_ext_default();
  mx.pattern_type = RT_PATTERN_STRIPE;
  mx.pattern_from[0] = mx.pattern_from[1] = mx.pattern_from[2] = 1.0;
  mx.pattern_to[0] = 0.439; mx.pattern_to[1] = 0.305; mx.pattern_to[2] = 0.827;
  rt_mat4_scale(0.175, 0.175, 0.175, pt);
  rt_mat4_translate(-3.5, 1.0, -3.0, m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &mm, &mx, pt)) < 0) return rc;

  /* right sphere: black mirror */
  rt_material rm = rt_material_default();
  rm.color[0] = rm.color[1] = rm.color[2] = 0.0;
  rm.diffuse = 0.7; rm.specular = 0.3; rm.reflective = 1.0;
  ident_translate_scale(-1.95, 1.0, -5.5, 0.5, 0.5, 0.5, m);
  if ((rc = rt_scene_add_sphere(b, m, &rm)) < 0) return rc;

  /* small sphere: dark stripes scaled by 0.07; 0.8f is a float literal (:6365) */
  rt_material sm = rt_material_default();
  sm.color[0] = 1.0; sm.color[1] = 0.8f; sm.color[2] = 0.1;
  sm.diffuse = 0.7; sm.specular = 0.3; sm.shininess = 100.0;
  rt_material_ext sx = rt_material_ext_default();
  sx.pattern_type = RT_PATTERN_STRIPE;
  sx.pattern_from[0] = sx.pattern_from[1] = sx.pattern_from[2] = 0.2;
  sx.pattern_to[0] = sx.pattern_to[1] = sx.pattern_to[2] = 0.0;
  rt_mat4_scale(0.07, 0.07, 0.07, pt);
  ident_translate_scale(-6.5, 0.33, -2.75, 0.33, 0.33, 0.33, m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &sm, &sx, pt)) < 0) return rc;

  /* origin sphere: default material, green */
  rt_material om = rt_material_default();
  om.color[0] = 0.0; om.color[2] = 0.0;
  rt_mat4_identity(m);
  if ((rc = rt_scene_add_sphere(b, m, &om)) < 0) return rc;
  return RT_OK;
}

static int add_dual_spheres(rt_scene_builder* b) { /* render_dual_spheres_refracting_on_floor, RayTracer.c:6432-6485 */
  double m[16], rot[16], pos[16];
  int rc;
  rt_material sm = rt_material_default();
  sm.ambient = 0.0;
  sm.color[0] = sm.color[1] = sm.color[2] = 1.0;
  sm.diffuse = 0.0; sm.specular = 0.9; sm.shininess = 300.0; sm.reflective = 0.9;
  rt_material_ext sx = rt_material_ext_default();
  sx.transparency = 0.9;
  sx.refractive_index = 1.0000034;
  rt_mat4_scale(0.5, 0.5, 0.5, m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &sm, &sx, NULL)) < 0) return rc; /* hollow centre */
  sx.refractive_index = 1.5;
  rt_mat4_identity(m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &sm, &sx, NULL)) < 0) return rc; /* outer sphere */
  rt_material wm = rt_material_default();
  wm.ambient = 0.8f; wm.diffuse = 0.2; wm.specular = 0.0;
  rt_material_ext wx = checkers(0.15, 0.85);
  rt_mat4_rotate_x(M_PI / 2.0, rot);
  rt_mat4_translate(0.0, 0.0, 10.0, pos);
  rt_mat4_mul(pos, rot, pos);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, pos, &wm, &wx, NULL, -INFINITY, INFINITY, 0)) < 0) return rc;
  return RT_OK;
}

static int add_refraction_scene(rt_scene_builder* b) { /* render_refraction_scene, RayTracer.c:6504-6710 */
  double m[16], t[16], r1[16], r2[16];
  int rc;
  const double ninf = -INFINITY, pinf = INFINITY;
  rt_material fm = rt_material_default();
  fm.ambient = 0.5; fm.diffuse = 0.4; fm.specular = 0.8; fm.reflective = 0.2;
  rt_material_ext cx = checkers(0.0, 0.75);
  rt_mat4_translate(0.0, 0.1, 0.0, t);
  rt_mat4_identity(m);
  rt_mat4_mul(m, t, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, m, &fm, &cx, NULL, ninf, pinf, 0)) < 0) return rc; /* floor */

  rt_material wm = rt_material_default();
  wm.specular = 0;
  rt_mat4_translate(5.0, 0.0, 0.0, t);
  rt_mat4_rotate_y(1.5708, r1);
  rt_mat4_rotate_z(1.5708, r2);
  rt_mat4_identity(m);
  rt_mat4_mul(r1, m, m);
  rt_mat4_mul(r2, m, m);
  rt_mat4_mul(t, m, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, m, &wm, &cx, NULL, ninf, pinf, 0)) < 0) return rc; /* right wall */
  rt_mat4_translate(0.0, 0.0, 5.0, t);
  rt_mat4_rotate_x(1.5708, r1);
  rt_mat4_identity(m);
  rt_mat4_mul(r1, m, m);
  rt_mat4_mul(t, m, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, m, &wm, &cx, NULL, ninf, pinf, 0)) < 0) return rc; /* left wall */
  rt_mat4_translate(0.0, 0.0, -5.0, t);
  rt_mat4_identity(m);
  rt_mat4_mul(r1, m, m);
  rt_mat4_mul(t, m, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_PLANE, m, &wm, &cx, NULL, ninf, pinf, 0)) < 0) return rc; /* wall behind */

  rt_material bm = rt_material_default();
  bm.color[0] = 0.8; bm.color[1] = 0.1; bm.color[2] = 0.3;
  rt_mat4_translate(4, 1, 4, m);
  if ((rc = rt_scene_add_sphere(b, m, &bm)) < 0) return rc; /* red ball */
  bm = rt_material_default();
  bm.color[0] = 0.2; bm.color[1] = 0.1; bm.color[2] = 0.8;
  bm.shininess = 10; bm.specular = 0.9;
  rt_mat4_translate(2.3, 0.6, 4.4, t);
  rt_mat4_scale(0.6, 0.6, 0.6, r1);
  rt_mat4_mul(t, r1, t);
  if ((rc = rt_scene_add_sphere(b, t, &bm)) < 0) return rc; /* blue ball */
  bm = rt_material_default();
  bm.color[0] = 0.1; bm.color[1] = 0.8; bm.color[2] = 0.2;
  bm.shininess = 200;
  rt_mat4_translate(4.6, 0.4, 2.9, t);
  rt_mat4_scale(0.4, 0.4, 0.4, r1);
  rt_mat4_mul(t, r1, t);
  if ((rc = rt_scene_add_sphere(b, t, &bm)) < 0) return rc; /* green ball */

  rt_material mm = rt_material_default(); /* mirror ball; refractive_index 0.0 as the scene sets it (:6648) */
  mm.color[0] = mm.color[1] = mm.color[2] = 0.0;
  mm.ambient = 0; mm.diffuse = 0.0; mm.specular = 0.9; mm.shininess = 300; mm.reflective = 1.0;
  rt_material_ext mx = rt_material_ext_default();
  mx.transparency = 0.0;
  mx.refractive_index = 0.0;
  ident_translate_scale(2.2, 0.6, 0.0, 0.5, 0.5, 0.5, m);
  if ((rc = rt_scene_add_sphere_ext(b, m, &mm, &mx, NULL)) < 0) return rc;

  ident_translate_scale(-0.60, 0.25, 0.80, 0.125, 0.125, 0.125, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_CUBE, m, NULL, NULL, NULL, ninf, pinf, 0)) < 0) return rc;

  rt_material cm = rt_material_default();
  cm.color[0] = 1.000; cm.color[1] = 0.766; cm.color[2] = 0.336;
  cm.shininess = 300; cm.reflective = 1.0; cm.ambient = 0.25;
  ident_translate_scale(.65, .21, -0.75, 0.175, 0.25, 0.175, m);
  if ((rc = rt_scene_add_shape(b, RT_SHAPE_CYLINDER, m, &cm, NULL, NULL, 0.0, 0.20, 0)) < 0) return rc;
  return RT_OK;
}

int rt_scene_stock(rt_scene_builder* b, const char* name, int hsize, int vsize) {
  if (!b || !name) return RT_ERR_INVALID;
  const double up[3] = {0.0, 1.0, 0.0};
  int rc;
  if (!strcmp(name, "world_with_plane")) {
    const double from[3] = {-10.0, 1.5, -10.0}, to[3] = {0.0, 0.0, 0.0};
    const double lp[3] = {-10.0, 10.0, -10.0}, li[3] = {1.0, 1.0, 1.0};
    if ((rc = add_world_with_plane(b)) != RT_OK) return rc;
    if ((rc = rt_scene_set_light(b, lp, li)) != RT_OK) return rc;
    return rt_scene_set_camera(b, hsize, vsize, M_PI / 3.0, from, to, up);
  }
  if (!strcmp(name, "dual_spheres")) {
    const double from[3] = {0.0, 0.0, -5.0}, to[3] = {0.0, 0.0, 0.0};
    const double lp[3] = {2.0, 10.0, -5.0}, li[3] = {0.9, 0.9, 0.9};
    if ((rc = rt_scene_set_camera(b, hsize, vsize, 0.45, from, to, up)) != RT_OK) return rc;
    if ((rc = rt_scene_set_light(b, lp, li)) != RT_OK) return rc;
    return add_dual_spheres(b);
  }
  if (!strcmp(name, "refraction")) {
    const double from[3] = {-4.5, 1.75, -4.7}, to[3] = {0.0, 0.85, 0.0};
    const double lp[3] = {-3.0, 3.0, 1}, li[3] = {1, 1, 1};
    if ((rc = rt_scene_set_camera(b, hsize, vsize, 0.5, from, to, up)) != RT_OK) return rc;
    if ((rc = rt_scene_set_light(b, lp, li)) != RT_OK) return rc;
    return add_refraction_scene(b);
  }
  if (!strcmp(name, "cube") || !strcmp(name, "lighthouse")) {
    const int lh = !strcmp(name, "lighthouse");
    const double from_c[3] = {10.0, 10.0, -5.0}, to_c[3] = {0.0, 0.0, 0.0};
    const double from_l[3] = {24.0, 18.0, -5.0}, to_l[3] = {0.0, 2.1, 0.0};
    const double lp[3] = {2.0, 10.0, -5.0}, li[3] = {0.9, 0.9, 0.9};
    if ((rc = rt_scene_set_camera(b, hsize, vsize, 0.5, lh ? from_l : from_c, lh ? to_l : to_c, up)) != RT_OK)
      return rc;
    if ((rc = rt_scene_set_light(b, lp, li)) != RT_OK) return rc;
    return lh ? RT_OK : add_teapot_h_mesh(b);
  }
  if (!strcmp(name, "spheres") || !strcmp(name, "cube_spheres")) {
    const double from[3] = {0.0, 1.5, -5.0}, to[3] = {0.0, 1.0, 0.0};
    const double lp[3] = {-10.0, 10.0, -10.0}, li[3] = {1.0, 1.0, 1.0};
    if ((rc = add_complete_world_spheres(b)) != RT_OK) return rc;
    if ((rc = rt_scene_set_light(b, lp, li)) != RT_OK) return rc;
    if ((rc = rt_scene_set_camera(b, hsize, vsize, M_PI / 3.0, from, to, up)) != RT_OK) return rc;
    return strcmp(name, "cube_spheres") ? RT_OK : add_teapot_h_mesh(b);
  }
  return RT_ERR_INVALID;
}

/* ------------------------------------------------------------------ PPM -- */
static size_t put_u8(char* d, unsigned v) { /* "%d " */
  size_t n = 0;
  if (v >= 100) d[n++] = (char)('0' + v / 100);
  if (v >= 10) d[n++] = (char)('0' + (v / 10) % 10);
  d[n++] = (char)('0' + v % 10);
  d[n++] = ' ';
  return n;
}

size_t rt_format_ppm(char* dst, size_t cap, const uint8_t* rgb, int width, int height) {
  char hdr[64];
  int h = snprintf(hdr, sizeof hdr, "P3\n%d %d\n255\n", width, height);
  size_t need = (size_t)h;
  const size_t npx = (size_t)width * height;
  for (size_t k = 0; k < npx * 3; ++k) need += rgb[k] >= 100 ? 4 : rgb[k] >= 10 ? 3 : 2;
  need += npx; /* '\n' after every pixel */
  if (!dst || cap < need) return need;
  memcpy(dst, hdr, (size_t)h);
  size_t o = (size_t)h;
  for (size_t p = 0; p < npx; ++p) {
    o += put_u8(dst + o, rgb[3 * p]);
    o += put_u8(dst + o, rgb[3 * p + 1]);
    o += put_u8(dst + o, rgb[3 * p + 2]);
    dst[o++] = '\n';
  }
  return o;
}

int rt_write_ppm(const char* path, const uint8_t* rgb, int width, int height) {
  if (!path || !rgb || width < 1 || height < 1) return RT_ERR_INVALID;
  size_t need = rt_format_ppm(NULL, 0, rgb, width, height);
  char* buf = malloc(need);
  if (!buf) return RT_ERR_INVALID;
  rt_format_ppm(buf, need, rgb, width, height);
  FILE* f = fopen(path, "wb");
  if (!f) { free(buf); return RT_ERR_INVALID; }
  size_t w = fwrite(buf, 1, need, f);
  fclose(f);
  free(buf);
  return w == need ? RT_OK : RT_ERR_INVALID;
}
