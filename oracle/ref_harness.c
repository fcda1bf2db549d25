/*
 * ref_harness.c - TEST INFRASTRUCTURE.  Drives the UNMODIFIED reference
 * (RayTracer/RayTracer.c, found through -I at build time, never copied into
 * this repository) at any resolution, sub-pixel pattern and stock scene.
 *
 * The reference ties render()/canvas/write_canvas_to_file to two 300x300
 * macros (RayTracer.c:42-43, 255, 1574), but create_camera() takes the size at
 * run time (RayTracer.c:1135) and ray_for_pixel()/color_at() have external
 * linkage, so this translation unit includes the reference source as is
 * (main renamed) and runs the same per-pixel loop as render() into its own
 * buffer.  Nothing in the reference's arithmetic is touched.
 *
 * Two probes are attached from outside through a macro on the libc call
 * qsort(): intersect_world() sorts exactly once per ray (RayTracer.c:1052), so
 * counting qsort calls counts rays (one ray = one call of intersect_world), and the
 * array handed to qsort shows the list occupancy and exact-zero t values.
 *
 * Built only where the reference is present (oracle/Makefile -> oracle/_ref/).
 *
 * Second build, -DRT_REF_CAPTURE -fPIC -shared (librt_ref_capture.so): for the scene functions this file does
 * not repeat (planes, patterns, glass, cube, cylinder) the REFERENCE'S OWN function builds the world.  Compiled
 * position-independent, its call render(c, &w) goes through the PLT; the executable rt_ref_capture defines
 * render() and hands the camera and world it receives to rt_capture_render() below, which runs the same
 * per-pixel loop at the requested size with the reference's own field of view and view transform.
 */
#include <stdint.h>
#include <stdlib.h> /* before the macro below, so that only call sites are rewritten */
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static uint64_t g_rays;          /* intersect_world calls */
static uint64_t g_zero_t;        /* entries with t == 0.0 exactly (add_intersection quirk, RayTracer.c:168) */
static size_t g_max_list;        /* largest intersection count of any ray */
static void rt_probe_qsort(void* base, size_t n, size_t size, int (*cmp)(const void*, const void*)) {
  ++g_rays;
  if (n > g_max_list) g_max_list = n;
  for (size_t i = 0; i < n; ++i) {
    double t;
    memcpy(&t, (char*)base + i * size, sizeof t);
    if (t == 0.0) ++g_zero_t;
  }
  (qsort)(base, n, size, cmp);
}
#ifndef RT_REF_NO_PROBE /* the timing build leaves the libc call alone */
#define qsort(b, n, s, c) rt_probe_qsort((b), (n), (s), (c))
#endif

#define main rt_reference_main
#include "RayTracer.c"
#undef main
#undef qsort

/* ---------------------------------------------------------------- scenes -- */
/* Each builder repeats the statements of the reference's scene function with
 * the image size as a parameter; `--stock` runs the reference's own function
 * instead, and tests require both to give the same 300x300 canvas. */

typedef struct {
  world w;
  camera* c;
} scene_t;

static void append_teapot_h_triangles(world* w) { /* RayTracer.c:6739-6768 */
  struct objpar_data obj_data;
  size_t file_size = strlen(teapot);
  void* p_buffer = malloc(objpar_get_size(teapot, file_size));
  unsigned int ok = objpar(teapot, file_size, p_buffer, &obj_data);
  if (ok != 1 || obj_data.face_width != 3) { fprintf(stderr, "teapot.h parse failed\n"); exit(2); }
  int loc = 0;
  for (unsigned int i = 0; i < obj_data.face_count; i++) {
    tuple p[3];
    for (int k = 0; k < 3; ++k) {
      unsigned int f = *(obj_data.p_faces + loc + 3 * k);
      float x = *(obj_data.p_positions + f * obj_data.position_width - 3);
      float y = *(obj_data.p_positions + f * obj_data.position_width - 2);
      float z = *(obj_data.p_positions + f * obj_data.position_width - 1);
      p[k] = create_point(x, y, z);
    }
    add_shape_to_world(create_triangle(p[0], p[1], p[2]), w);
    loc += obj_data.face_width * 3;
  }
  free(p_buffer);
}

static scene_t scene_cube(int hs, int vs) { /* render_some_triangles, RayTracer.c:6727 */
  scene_t s;
  s.w = create_world();
  s.c = create_camera(hs, vs, 0.5);
  view_transform(create_point(10.0, 10.0, -5.0), create_point(0.0, 0.0, 0.0), create_vector(0.0, 1.0, 0.0),
                 s.c->view_transform);
  *s.w.lights = create_point_light(create_point(2.0, 10.0, -5.0), create_point(0.9, 0.9, 0.9));
  append_teapot_h_triangles(&s.w);
  return s;
}

/* The triangles load_model_file() would make, from a file of 9 floats per triangle (the positions it reads out of
 * objpar's float array, RayTracer.c:1657-1667, in face order): widened to double and handed to the reference's own
 * create_triangle / add_shape_to_world (RayTracer.c:1669), as load_model_file does.  For hosts where the OBJ text
 * is not present. */
static void append_triangles_f32(world* w, const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  float v[9];
  while (fread(v, sizeof(float), 9, f) == 9)
    add_shape_to_world(create_triangle(create_point(v[0], v[1], v[2]), create_point(v[3], v[4], v[5]),
                                       create_point(v[6], v[7], v[8])), w);
  fclose(f);
}

static scene_t scene_lighthouse(int hs, int vs, const char* obj, const char* tris_f32) { /* render_lighthouse_scene, RayTracer.c:6772 */
  scene_t s;
  s.w = create_world();
  s.c = create_camera(hs, vs, 0.5);
  view_transform(create_point(24.0, 18.0, -5.0), create_point(0.0, 2.1, 0.0), create_vector(0.0, 1.0, 0.0),
                 s.c->view_transform);
  *s.w.lights = create_point_light(create_point(2.0, 10.0, -5.0), create_point(0.9, 0.9, 0.9));
  if (tris_f32) append_triangles_f32(&s.w, tris_f32);
  else load_model_file((char*)obj, &s.w);
  return s;
}

static void append_complete_world_spheres(world* w) { /* render_complete_world, RayTracer.c:5852-6163 */
  shape* floor = create_shape(SPHERE);
  Mat4x4 floor_transform;
  gen_scale_matrix(10.0, 0.01, 10.0, floor_transform);
  mat4x4_mul_in_place(floor_transform, floor->transform, floor->transform);
  material floor_material = create_material_default();
  floor_material.color = create_point(0.9, 0.9, 0.9);
  floor_material.specular = 0.0;
  floor->material = floor_material;

  shape* left_wall = create_shape(SPHERE);
  Mat4x4 translate_left, rotate_y_left, rotate_x_left, scale_left, m;
  floor_material.color = create_point(1.0, 0.0, 0.0);
  left_wall->material = floor_material;
  gen_translate_matrix(0.0, 0.0, 5.0, translate_left);
  gen_rotate_matrix_Y(-M_PI / 4.0, rotate_y_left);
  gen_rotate_matrix_X(M_PI / 2.0, rotate_x_left);
  gen_scale_matrix(10.0, 0.01, 10.0, scale_left);
  mat4x4_set_ident(m);
  mat4x4_mul_in_place(translate_left, m, m);
  mat4x4_mul_in_place(m, rotate_y_left, m);
  mat4x4_mul_in_place(m, rotate_x_left, m);
  mat4x4_mul_in_place(m, scale_left, m);
  mat4x4_copy(m, left_wall->transform);

  shape* right_wall = create_shape(SPHERE); /* keeps the default material, RayTracer.c:6032-6044 */
  floor_material.color = create_point(0.0, 0.0, 1.0);
  Mat4x4 rotate_y_right;
  mat4x4_set_ident(m);
  mat4x4_mul_in_place(m, translate_left, m);
  gen_rotate_matrix_Y(M_PI / 4.0, rotate_y_right);
  mat4x4_mul_in_place(m, rotate_y_right, m);
  mat4x4_mul_in_place(m, rotate_x_left, m);
  mat4x4_mul_in_place(m, scale_left, m);
  mat4x4_copy(m, right_wall->transform);
  left_wall->material = floor_material; /* the blue material lands on the LEFT wall */

  shape* middle = create_shape(SPHERE);
  Mat4x4 t1;
  gen_translate_matrix(-0.5, 1.0, 0.5, t1);
  mat4x4_copy(t1, middle->transform);
  material mm = create_material_default();
  mm.color = create_point(0.1, 1.0, 0.5);
  mm.diffuse = 0.7;
  mm.specular = 0.3;
  middle->material = mm;

  shape* right = create_shape(SPHERE);
  Mat4x4 tr, sr;
  gen_translate_matrix(1.5, 0.5, -0.5, tr);
  gen_scale_matrix(0.5, 0.5, 0.5, sr);
  mat4x4_set_ident(m);
  mat4x4_mul_in_place(m, tr, m);
  mat4x4_mul_in_place(m, sr, m);
  mat4x4_copy(m, right->transform);
  material rm = create_material_default();
  rm.color = create_point(0.5, 1.0, 0.1);
  rm.diffuse = 0.7;
  rm.specular = 0.3;
  right->material = rm;

  shape* small = create_shape(SPHERE);
  Mat4x4 ts, ss;
  gen_translate_matrix(-1.5, 0.33, -0.75, ts);
  gen_scale_matrix(0.33, 0.33, 0.33, ss);
  mat4x4_set_ident(m);
  mat4x4_mul_in_place(m, ts, m);
  mat4x4_mul_in_place(m, ss, m);
  mat4x4_copy(m, small->transform);
  material sm = create_material_default();
  sm.color = create_point(1.0, 0.8f, 0.1);
  sm.diffuse = 0.7;
  sm.specular = 0.3;
  small->material = sm;

  add_shape_to_world(floor, w);
  add_shape_to_world(left_wall, w);
  add_shape_to_world(right_wall, w);
  add_shape_to_world(middle, w);
  add_shape_to_world(right, w);
  add_shape_to_world(small, w);
}

static scene_t scene_spheres(int hs, int vs) { /* render_complete_world, RayTracer.c:5848, 6205-6213 */
  scene_t s;
  s.w = create_world();
  append_complete_world_spheres(&s.w);
  *s.w.lights = create_point_light(create_point(-10.0, 10.0, -10.0), create_point(1.0, 1.0, 1.0));
  s.c = create_camera(hs, vs, M_PI / 3.0);
  view_transform(create_point(0.0, 1.5, -5.0), create_point(0.0, 1.0, 0.0), create_vector(0.0, 1.0, 0.0),
                 s.c->view_transform);
  return s;
}

/* teapot.h mesh + the six spheres in one world: spheres first, then triangles. */
static scene_t scene_cube_spheres(int hs, int vs) {
  scene_t s = scene_spheres(hs, vs);
  append_teapot_h_triangles(&s.w);
  return s;
}

/* ------------------------------------------------------------------ dump -- */
static void put(FILE* f, const void* p, size_t n) {
  if (fwrite(p, 1, n, f) != n) { perror("fwrite"); exit(2); }
}
static void put_tuple(FILE* f, tuple t) { put(f, &t, sizeof t); }

/* Scene description as the reference holds it, plus the loop invariants the
 * reference recomputes per ray, evaluated here by the reference's own
 * mat4x4_inverse (RayTracer.c:601).  Layout is read by tests/refdump.py. */
static void dump_scene(const char* path, scene_t* s) {
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); exit(2); }
  put(f, "RTSCENE1", 8);
  int32_t hv[2] = {(int32_t)s->c->hsize, (int32_t)s->c->vsize};
  put(f, hv, sizeof hv);
  double cam[4] = {s->c->field_of_view, s->c->pixel_size, s->c->half_width, s->c->half_height};
  put(f, cam, sizeof cam);
  put(f, s->c->view_transform, sizeof(Mat4x4));
  Mat4x4 inv;
  mat4x4_inverse(s->c->view_transform, inv);
  put(f, inv, sizeof inv);
  put_tuple(f, s->w.lights->position);
  put_tuple(f, s->w.lights->intensity);
  int32_t n = 0;
  for (shape* sp = s->w.objects; sp; sp = sp->next) ++n;
  int32_t hdr[2] = {n, 0};
  put(f, hdr, sizeof hdr);
  for (shape* sp = s->w.objects; sp; sp = sp->next) {
    int32_t ty[2] = {(int32_t)sp->type, 0};
    put(f, ty, sizeof ty);
    put(f, sp->transform, sizeof(Mat4x4));
    mat4x4_set_ident(inv);
    mat4x4_inverse(sp->transform, inv);
    put(f, inv, sizeof inv);
    if (sp->type == TRIANGLE) {
      put_tuple(f, sp->p1); put_tuple(f, sp->p2); put_tuple(f, sp->p3);
      put_tuple(f, sp->e1); put_tuple(f, sp->e2); put_tuple(f, sp->normal);
    } else {
      tuple z = {0, 0, 0, 0};
      for (int k = 0; k < 6; ++k) put_tuple(f, z);
    }
    put_tuple(f, sp->material.color);
    double mat[7] = {sp->material.ambient, sp->material.diffuse, sp->material.specular, sp->material.shininess,
                     sp->material.reflective, sp->material.transparency, sp->material.refractive_index};
    put(f, mat, sizeof mat);
  }
  /* second section: what planes, cubes, cylinders, patterns and glass add */
  put(f, "RTEXT001", 8);
  for (shape* sp = s->w.objects; sp; sp = sp->next) {
    double mm[2] = {sp->type == CYLINDER ? sp->minimum : 0.0, sp->type == CYLINDER ? sp->maximum : 0.0};
    put(f, mm, sizeof mm);
    int32_t fl[4] = {sp->type == CYLINDER ? (int32_t)sp->closed : 0, (int32_t)sp->material.has_pattern,
                     sp->material.has_pattern ? (int32_t)sp->material.pattern.type : -1, 0};
    put(f, fl, sizeof fl);
    tuple z = {0, 0, 0, 0};
    put_tuple(f, sp->material.has_pattern ? sp->material.pattern.from : z);
    put_tuple(f, sp->material.has_pattern ? sp->material.pattern.to : z);
    Mat4x4 pt, pinv;
    mat4x4_set_ident(pt);
    if (sp->material.has_pattern) mat4x4_copy(sp->material.pattern.transform, pt);
    put(f, pt, sizeof pt);
    mat4x4_inverse(pt, pinv);
    put(f, pinv, sizeof pinv);
  }
  fclose(f);
}

/* ------------------------------------------------------------------ main -- */
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void usage(void) {
  fprintf(stderr,
          "rt_ref_harness --scene lighthouse|cube|spheres|cube_spheres [--size W H] [--spp-grid G]\n"
          "   [--tri-reflective K] [--sph-reflective K] [--obj lighthouse.obj | --tris-f32 F] [--rows Y0 Y1]\n"
          "   [--stock] [--dump-scene F] [--out-f64 F] [--out-rgb F] [--out-ppm F] [--no-render]\n"
          "rt_ref_capture --capture --scene refraction|world_with_plane|dual_spheres|cube|spheres ... (same options)\n");
  exit(2);
}

/* options shared by main and the capture callback */
static struct {
  const char* scene;
  const char *dump, *out_f64, *out_rgb, *out_ppm;
  int W, H, G, no_render, y0, y1, stock;
  double tri_refl, sph_refl;
} g_opt;

static void write_outputs(const double* img, int W, int H) {
  if (img && g_opt.out_f64) {
    FILE* f = fopen(g_opt.out_f64, "wb");
    if (!f) { perror(g_opt.out_f64); exit(2); }
    put(f, img, sizeof(double) * 3 * W * H);
    fclose(f);
  }
  if (img && g_opt.out_rgb) {
    FILE* f = fopen(g_opt.out_rgb, "wb");
    if (!f) { perror(g_opt.out_rgb); exit(2); }
    for (size_t k = 0; k < (size_t)3 * W * H; ++k) {
      unsigned char b = (unsigned char)color_convert(img[k]);
      put(f, &b, 1);
    }
    fclose(f);
  }
  if (img && g_opt.out_ppm) { /* same bytes as write_canvas_to_file, RayTracer.c:1333-1348 */
    FILE* f = fopen(g_opt.out_ppm, "w");
    if (!f) { perror(g_opt.out_ppm); exit(2); }
    fprintf(f, "P3\n");
    fprintf(f, "%d %d\n255\n", W, H);
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        fprintf(f, "%d ", color_convert(img[(y * W + x) * 3 + 0]));
        fprintf(f, "%d ", color_convert(img[(y * W + x) * 3 + 1]));
        fprintf(f, "%d \n", color_convert(img[(y * W + x) * 3 + 2]));
      }
    fclose(f);
  }
}

static void report(double seconds) {
  printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp_grid\": %d, \"rows\": [%d, %d], \"stock\": %d, "
         "\"rays\": %llu, \"max_list\": %zu, \"exact_zero_t\": %llu, \"seconds\": %.6f}\n",
         g_opt.scene, g_opt.W, g_opt.H, g_opt.G, g_opt.y0, g_opt.y1, g_opt.stock, (unsigned long long)g_rays, g_max_list,
         (unsigned long long)g_zero_t, seconds);
  fflush(stdout);
}

/* the per-pixel loop of render() (RayTracer.c:1576-1587) at any size and sub-pixel grid, then the outputs */
static void trace_scene(scene_t* s) {
  const int W = g_opt.W, H = g_opt.H, G = g_opt.G;
  double* img = NULL;
  double seconds = 0.0;
  for (shape* sp = s->w.objects; sp; sp = sp->next) {
    if (sp->type == TRIANGLE && g_opt.tri_refl != 0.0) sp->material.reflective = g_opt.tri_refl;
    if (sp->type == SPHERE && g_opt.sph_refl != 0.0) sp->material.reflective = g_opt.sph_refl;
  }
  if (g_opt.dump) dump_scene(g_opt.dump, s);
  if (g_opt.y1 < 0 || g_opt.y1 > H) g_opt.y1 = H;
  if (!g_opt.no_render) {
    img = calloc((size_t)3 * W * H, sizeof(double));
    g_rays = 0;
    g_zero_t = 0;
    g_max_list = 0;
    double t0 = now_s();
    for (int y = g_opt.y0; y < g_opt.y1; ++y) {
      for (int x = 0; x < W; ++x) {
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (int j = 0; j < G; ++j) {
          for (int i = 0; i < G; ++i) {
            /* G == 1 gives px = x exactly: the reference's render() loop, RayTracer.c:1576-1587 */
            double px = x + (i + 0.5) / G - 0.5;
            double py = y + (j + 0.5) / G - 0.5;
            ray r = ray_for_pixel(s->c, px, py);
            tuple col = color_at(&s->w, &r, RECURSION_DEPTH);
            sx += col.x; sy += col.y; sz += col.z;
          }
        }
        double n = (double)(G * G);
        img[(y * W + x) * 3 + 0] = sx / n;
        img[(y * W + x) * 3 + 1] = sy / n;
        img[(y * W + x) * 3 + 2] = sz / n;
      }
    }
    seconds = now_s() - t0;
  }
  write_outputs(img, W, H);
  report(seconds);
  free(img);
}

#ifdef RT_REF_CAPTURE
/* called by the executable's render(): the world is the one the reference's scene function just built */
void rt_capture_render(camera* c, world* w) {
  scene_t s;
  s.w = *w;
  s.c = create_camera(g_opt.W, g_opt.H, c->field_of_view);
  mat4x4_copy(c->view_transform, s.c->view_transform);
  trace_scene(&s);
  free(s.c);
}
#endif

#ifdef RT_REF_CAPTURE
int rt_harness_main(int argc, char** argv) {
#else
int main(int argc, char** argv) {
#endif
  const char* obj = "lighthouse.obj";
  const char* tris_f32 = NULL;
  memset(&g_opt, 0, sizeof g_opt);
  g_opt.W = HORIZONTAL_SIZE; g_opt.H = VERTICAL_SIZE; g_opt.G = 1; g_opt.y1 = -1;
  int capture = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--scene") && i + 1 < argc) g_opt.scene = argv[++i];
    else if (!strcmp(argv[i], "--size") && i + 2 < argc) { g_opt.W = atoi(argv[++i]); g_opt.H = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--spp-grid") && i + 1 < argc) g_opt.G = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--tri-reflective") && i + 1 < argc) g_opt.tri_refl = atof(argv[++i]);
    else if (!strcmp(argv[i], "--sph-reflective") && i + 1 < argc) g_opt.sph_refl = atof(argv[++i]);
    else if (!strcmp(argv[i], "--obj") && i + 1 < argc) obj = argv[++i];
    else if (!strcmp(argv[i], "--tris-f32") && i + 1 < argc) tris_f32 = argv[++i];
    else if (!strcmp(argv[i], "--rows") && i + 2 < argc) { g_opt.y0 = atoi(argv[++i]); g_opt.y1 = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--stock")) g_opt.stock = 1;
    else if (!strcmp(argv[i], "--capture")) capture = 1;
    else if (!strcmp(argv[i], "--no-render")) g_opt.no_render = 1;
    else if (!strcmp(argv[i], "--dump-scene") && i + 1 < argc) g_opt.dump = argv[++i];
    else if (!strcmp(argv[i], "--out-f64") && i + 1 < argc) g_opt.out_f64 = argv[++i];
    else if (!strcmp(argv[i], "--out-rgb") && i + 1 < argc) g_opt.out_rgb = argv[++i];
    else if (!strcmp(argv[i], "--out-ppm") && i + 1 < argc) g_opt.out_ppm = argv[++i];
    else usage();
  }
  const char* scene = g_opt.scene;
  const int W = g_opt.W, H = g_opt.H, G = g_opt.G;
  if (!scene || W < 1 || H < 1 || G < 1) usage();
  contents = containers_create(); /* RayTracer.c:7048 */

  if (capture) {
#ifdef RT_REF_CAPTURE
    /* the reference's own scene function; its render() call lands in rt_capture_render */
    if (!strcmp(scene, "refraction")) render_refraction_scene();
    else if (!strcmp(scene, "world_with_plane")) render_complete_world_with_plane();
    else if (!strcmp(scene, "dual_spheres")) render_dual_spheres_refracting_on_floor();
    else if (!strcmp(scene, "cube")) render_some_triangles();
    else if (!strcmp(scene, "spheres")) render_complete_world();
    else usage();
    return 0;
#else
    usage();
#endif
  }
  if (g_opt.stock) {
    /* the reference's own scene function and its own render() into `canvas` */
    if (W != HORIZONTAL_SIZE || H != VERTICAL_SIZE || G != 1) usage();
    if (!strcmp(scene, "lighthouse")) {
      const char* slash = strrchr(obj, '/');
      if (slash) { /* load_model_file("lighthouse.obj") is CWD-relative, RayTracer.c:6784 */
        char dir[4096];
        snprintf(dir, sizeof dir, "%.*s", (int)(slash - obj), obj);
        if (chdir(dir) != 0) { perror(dir); return 2; }
      }
    }
    double t0 = now_s();
    if (!strcmp(scene, "lighthouse")) render_lighthouse_scene();
    else if (!strcmp(scene, "cube")) render_some_triangles();
    else if (!strcmp(scene, "spheres")) render_complete_world();
    else if (!strcmp(scene, "refraction")) render_refraction_scene();
    else if (!strcmp(scene, "world_with_plane")) render_complete_world_with_plane();
    else if (!strcmp(scene, "dual_spheres")) render_dual_spheres_refracting_on_floor();
    else usage();
    double seconds = now_s() - t0;
    double* img = malloc(sizeof(double) * 3 * W * H);
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        img[(y * W + x) * 3 + 0] = canvas[x][y].x;
        img[(y * W + x) * 3 + 1] = canvas[x][y].y;
        img[(y * W + x) * 3 + 2] = canvas[x][y].z;
      }
    g_opt.y0 = 0; g_opt.y1 = H;
    write_outputs(img, W, H);
    report(seconds);
    free(img);
    return 0;
  }
  scene_t s;
  if (!strcmp(scene, "lighthouse")) s = scene_lighthouse(W, H, obj, tris_f32);
  else if (!strcmp(scene, "cube")) s = scene_cube(W, H);
  else if (!strcmp(scene, "spheres")) s = scene_spheres(W, H);
  else if (!strcmp(scene, "cube_spheres")) s = scene_cube_spheres(W, H);
  else usage();
  trace_scene(&s);
  return 0;
}
