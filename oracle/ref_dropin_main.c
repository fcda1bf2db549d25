/*
 * ref_dropin_main.c - TEST INFRASTRUCTURE: the reference's scene functions and
 * PPM writer, unmodified, with render() answered by the HIP library.
 *
 *   raytracer_dropin lighthouse|cube|spheres|refraction|world_with_plane|dual_spheres   -> ./canvas.ppm (300x300)
 *
 * `render` below interposes the definition inside libraytracer_ref_dropin.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int rt_render_hip(void* c, void* w);
void render_lighthouse_scene(void);
void render_some_triangles(void);
void render_complete_world(void);
void render_refraction_scene(void);
void render_complete_world_with_plane(void);
void render_dual_spheres_refracting_on_floor(void);
void write_canvas_to_file(void);

void render(void* c, void* w) {
  int rc = rt_render_hip(c, w);
  if (rc != 0) {
    fprintf(stderr, "rt_render_hip failed: %d\n", rc);
    exit(1);
  }
}

int main(int argc, char** argv) {
  const char* scene = argc > 1 ? argv[1] : "lighthouse";
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
