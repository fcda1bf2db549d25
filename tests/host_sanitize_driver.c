/* Exercises the host-side scene builder, OBJ reader and BVH builder; built with -fsanitize=address,undefined by
 * test_host_sanitizers.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "rt_hip.h"
#include "rt_scene.h"
int main(void) {
  const char* names[] = {"spheres", "cube", "cube_spheres", "refraction", "world_with_plane", "dual_spheres", "lighthouse"};
  for (int k = 0; k < 7; ++k) {
    for (int leaf = 0; leaf <= 7; leaf += (leaf == 0 ? 1 : 3)) {
      rt_scene_builder* b = rt_scene_create();
      int rc = rt_scene_stock(b, names[k], 64, 48);
      if (rc != RT_OK) { printf("stock %s rc %d\n", names[k], rc); return 1; }
      if (!strcmp(names[k], "lighthouse")) {
        /* a random mesh + OBJ text with quads */
        float* v = malloc(sizeof(float) * 9 * 5000);
        for (int i = 0; i < 9 * 5000; ++i) v[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
        rc = rt_scene_add_triangles_f32(b, v, 5000, NULL);
        free(v);
        const char* obj = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\n# c\nf 1 2 3 4\nf 1/1/1 2/2/2 3/3/3\nf 1 2\nvn 0 0 1\n\n";
        int rc2 = rt_scene_load_obj(b, obj, strlen(obj), NULL);
        printf("obj rc %d\n", rc2);
      }
      rt_scene_set_reflective(b, 0.3, 0.2);
      rt_scene s;
      rc = rt_scene_finish(b, leaf, &s);
      printf("%s leaf %d: rc %d spheres %d tris %d shapes %d nodes %d\n", names[k], leaf, rc, s.n_spheres, s.n_tris, s.n_shapes, s.n_bvh_nodes);
      rc = rt_scene_finish(b, 4, &s); /* twice */
      rt_scene_destroy(b);
    }
  }
  { /* 70000 triangles: the BVH builder hands subtrees to worker threads */
    rt_scene_builder* b = rt_scene_create();
    if (rt_scene_stock(b, "lighthouse", 64, 48) != RT_OK) return 1;
    const int n = 70000;
    float* v = malloc(sizeof(float) * 9 * (size_t)n);
    for (int i = 0; i < 9 * n; ++i) v[i] = (float)rand() / RAND_MAX * 4.f - 2.f;
    int rc = rt_scene_add_triangles_f32(b, v, n, NULL);
    free(v);
    rt_scene s;
    if (rc == RT_OK) rc = rt_scene_finish(b, 4, &s);
    printf("threaded build: rc %d tris %d nodes %d\n", rc, s.n_tris, s.n_bvh_nodes);
    rt_scene_destroy(b);
    if (rc != RT_OK) return 1;
  }
  return 0;
}
