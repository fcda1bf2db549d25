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

struct rt_scene_builder {
  rt_camera camera;
  int have_camera, have_light;
  rt_light light;
  sphere_rec* spheres; int n_spheres, cap_spheres;
  tri_rec* tris; int n_tris, cap_tris;
  rt_material* materials; int n_materials, cap_materials;
  /* outputs of rt_scene_finish */
  double* sph_inv; int32_t* sph_material; int32_t* sph_order;
  double* tri_soa; int32_t* tri_material; int32_t* tri_order;
  rt_bvh_node* bvh_nodes; int32_t* bvh_tri; int n_bvh_nodes;
};

rt_scene_builder* rt_scene_create(void) { return calloc(1, sizeof(rt_scene_builder)); }

static void free_outputs(rt_scene_builder* b) {
  free(b->sph_inv); free(b->sph_material); free(b->sph_order);
  free(b->tri_soa); free(b->tri_material); free(b->tri_order);
  free(b->bvh_nodes); free(b->bvh_tri);
  b->sph_inv = b->tri_soa = NULL;
  b->sph_material = b->sph_order = b->tri_material = b->tri_order = b->bvh_tri = NULL;
  b->bvh_nodes = NULL;
  b->n_bvh_nodes = 0;
}

void rt_scene_destroy(rt_scene_builder* b) {
  if (!b) return;
  free_outputs(b);
  free(b->spheres); free(b->tris); free(b->materials);
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

static int intern_material(rt_scene_builder* b, const rt_material* m) {
  rt_material d = rt_material_default();
  if (!m) m = &d;
  for (int i = b->n_materials - 1; i >= 0 && i >= b->n_materials - 8; --i)
    if (memcmp(&b->materials[i], m, sizeof *m) == 0) return i;
  if (!grow((void**)&b->materials, &b->cap_materials, b->n_materials + 1, sizeof *m)) return -1;
  b->materials[b->n_materials] = *m;
  return b->n_materials++;
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

int rt_scene_add_sphere(rt_scene_builder* b, const double transform[16], const rt_material* m) {
  if (!b || !transform) return RT_ERR_INVALID;
  if (!grow((void**)&b->spheres, &b->cap_spheres, b->n_spheres + 1, sizeof(sphere_rec))) return RT_ERR_INVALID;
  int mi = intern_material(b, m);
  if (mi < 0) return RT_ERR_INVALID;
  sphere_rec* s = &b->spheres[b->n_spheres++];
  memcpy(s->transform, transform, sizeof s->transform);
  s->material = mi;
  s->order = b->n_spheres - 1 + b->n_tris;
  return s->order;
}

int rt_scene_add_triangle(rt_scene_builder* b, const double p1[3], const double p2[3], const double p3[3],
                          const rt_material* m) {
  if (!b || !p1 || !p2 || !p3) return RT_ERR_INVALID;
  if (!grow((void**)&b->tris, &b->cap_tris, b->n_tris + 1, sizeof(tri_rec))) return RT_ERR_INVALID;
  int mi = intern_material(b, m);
  if (mi < 0) return RT_ERR_INVALID;
  tri_rec* t = &b->tris[b->n_tris++];
  memcpy(t->p1, p1, sizeof t->p1);
  memcpy(t->p2, p2, sizeof t->p2);
  memcpy(t->p3, p3, sizeof t->p3);
  t->material = mi;
  t->order = b->n_spheres + b->n_tris - 1;
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
    m.reflective = tri_reflective;
    int mi = intern_material(b, &m);
    if (mi < 0) return RT_ERR_INVALID;
    b->tris[i].material = mi;
  }
  for (int i = 0; i < b->n_spheres; ++i) {
    rt_material m = b->materials[b->spheres[i].material];
    m.reflective = sph_reflective;
    int mi = intern_material(b, &m);
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
        idx[n++] = atoi(text + p); /* leading integer of v, v/vt, v//vn or v/vt/vn */
        while (p < e && text[p] != ' ' && text[p] != '\t') ++p;
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
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  rewind(f);
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
  free_outputs(b);
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

  memset(out, 0, sizeof *out);
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

int rt_scene_stock(rt_scene_builder* b, const char* name, int hsize, int vsize) {
  if (!b || !name) return RT_ERR_INVALID;
  const double up[3] = {0.0, 1.0, 0.0};
  int rc;
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
