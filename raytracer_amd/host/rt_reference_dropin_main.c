/*
 * rt_reference_dropin_main.c - the reference's scene functions and PPM writer, unmodified, with render()
 * (RayTracer/RayTracer.c:1574) answered by the HIP library.
 *
 *   raytracer_dropin lighthouse|cube|spheres|refraction|world_with_plane|dual_spheres   -> ./canvas.ppm (300x300)
 *   raytracer_dropin --dump-scene FILE <scene>    no device: the flattened world the binding would upload, as
 *                                                 n_tris (int32), n_spheres (int32), then camera pixel_size,
 *                                                 half_width, half_height, inv_view[16], light position[3] and
 *                                                 intensity[3] (doubles), then tri_soa (12 * n_tris doubles)
 *
 * `render` below interposes the definition inside libraytracer_ref_dropin.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_hip.h"
#include "rt_scene.h"

int rt_render_hip(void* c, void* w);
int rt_binding_flatten(void* c, void* w, rt_scene_builder** builder, rt_scene* scene);
void render_lighthouse_scene(void);
void render_some_triangles(void);
void render_complete_world(void);
void render_refraction_scene(void);
void render_complete_world_with_plane(void);
void render_dual_spheres_refracting_on_floor(void);
void write_canvas_to_file(void);

static const char* g_dump = NULL;

void render(void* c, void* w) {
  if (g_dump) {
    rt_scene_builder* b = NULL;
    rt_scene s;
    int rc = rt_binding_flatten(c, w, &b, &s);
    FILE* f = rc == RT_OK ? fopen(g_dump, "wb") : NULL;
    if (!f) { fprintf(stderr, "--dump-scene: %d\n", rc); exit(1); }
    fwrite(&s.n_tris, sizeof(int32_t), 1, f);
    fwrite(&s.n_spheres, sizeof(int32_t), 1, f);
    fwrite(&s.camera.pixel_size, sizeof(double), 1, f);
    fwrite(&s.camera.half_width, sizeof(double), 1, f);
    fwrite(&s.camera.half_height, sizeof(double), 1, f);
    fwrite(s.camera.inv_view, sizeof(double), 16, f);
    fwrite(s.light.position, sizeof(double), 3, f);
    fwrite(s.light.intensity, sizeof(double), 3, f);
    if (s.n_tris) fwrite(s.tri_soa, sizeof(double), (size_t)12 * s.n_tris, f);
    fclose(f);
    rt_scene_destroy(b);
    exit(0);
  }
  int rc = rt_render_hip(c, w);
  if (rc != 0) {
    fprintf(stderr, "rt_render_hip failed: %d %s\n", rc, rt_hip_last_error());
    exit(1);
  }
}

int main(int argc, char** argv) {
  int i = 1;
  if (argc > 2 && !strcmp(argv[1], "--dump-scene")) { g_dump = argv[2]; i = 3; }
  const char* scene = argc > i ? argv[i] : "lighthouse";
  printf("Starting Render Process\n");
  if (!strcmp(scene, "lighthouse")) render_lighthouse_scene(); /* reads ./lighthouse.obj */
  else if (!strcmp(scene, "cube")) render_some_triangles();
  else if (!strcmp(scene, "spheres")) render_complete_world();
  else if (!strcmp(scene, "refraction")) render_refraction_scene();
  else if (!strcmp(scene, "world_with_plane")) render_complete_world_with_plane();
  else if (!strcmp(scene, "dual_spheres")) render_dual_spheres_refracting_on_floor();
  else return 2;
  printf("Writing Canvas To File\n");
  write_canvas_to_file();
  return 0;
}
