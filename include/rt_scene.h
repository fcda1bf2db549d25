/*
 * rt_scene.h - host-side C helpers that turn a scene description into the
 * flat `rt_scene` of rt_hip.h, using the reference's own arithmetic so that
 * every derived value (inverse matrices, triangle edges and normals, camera
 * constants) has the same bits the reference computes per ray.
 *
 * A host that already holds the reference's `world`/`camera` structs needs
 * only rt_scene_add_* and rt_scene_set_camera_view: it passes
 * shape.transform, shape.p1..p3, shape.material and camera.view_transform
 * straight through (raytracer_amd/host/rt_reference_binding.c is that loop).
 *
 * Reference counterparts (RayTracer/RayTracer.c):
 *   rt_mat4_mul            mat4x4_mul_in_place   :405
 *   rt_mat4_inverse        mat4x4_inverse        :601 (cofactors :520-594)
 *   rt_mat4_translate ...  gen_*_matrix          :620-653
 *   rt_view_transform      view_transform        :1501
 *   rt_scene_set_camera    create_camera         :1135
 *   rt_scene_add_sphere    create_shape(SPHERE) + set_transform  :968, :1057
 *   rt_scene_add_triangle  create_triangle       :998
 *   rt_scene_load_obj      load_model_file       :1621 (positions and faces only)
 *   rt_write_ppm           write_canvas_to_file  :1333
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 4x4 row-major matrices, reference operation order ---- */
void rt_mat4_identity(double m[16]);
void rt_mat4_translate(double x, double y, double z, double m[16]);
void rt_mat4_scale(double x, double y, double z, double m[16]);
void rt_mat4_rotate_x(double rad, double m[16]);
void rt_mat4_rotate_y(double rad, double m[16]);
void rt_mat4_rotate_z(double rad, double m[16]);
void rt_mat4_mul(const double a[16], const double b[16], double out[16]); /* out may alias a or b */
/* returns 0 and leaves `out` untouched when |det| < 1e-6 (mat4x4_invertable, RayTracer.c:596) */
int rt_mat4_inverse(const double a[16], double out[16]);
void rt_view_transform(const double from[3], const double to[3], const double up[3], double m[16]);

/* default material, create_material_default RayTracer.c:704 */
rt_material rt_material_default(void);

/* the pattern / transparency half of the default material (no pattern, opaque, refractive_index 1), :712-717 */
rt_material_ext rt_material_ext_default(void);

/* ---- scene builder ---- */
typedef struct rt_scene_builder rt_scene_builder;

rt_scene_builder* rt_scene_create(void);
void rt_scene_destroy(rt_scene_builder* b);

/* create_camera(hsize, vsize, fov) + view_transform(from, to, up) */
int rt_scene_set_camera(rt_scene_builder* b, int hsize, int vsize, double fov, const double from[3],
                        const double to[3], const double up[3]);
/* create_camera(hsize, vsize, fov) with a ready view_transform (camera.view_transform, RayTracer.c:96) */
int rt_scene_set_camera_view(rt_scene_builder* b, int hsize, int vsize, double fov, const double view[16]);
int rt_scene_set_light(rt_scene_builder* b, const double position[3], const double intensity[3]);

/* Append to the world list (add_shape_to_world, RayTracer.c:1078); the return
 * value is the position in the list (>= 0) or a negative rt_status. */
int rt_scene_add_sphere(rt_scene_builder* b, const double transform[16], const rt_material* m);
/* a sphere whose material has a pattern and/or transparency; pattern_transform (NULL = identity) is
 * pattern.transform as set_pattern_transform stores it (RayTracer.c:1061), x->pattern_inv is ignored */
int rt_scene_add_sphere_ext(rt_scene_builder* b, const double transform[16], const rt_material* m,
                            const rt_material_ext* x, const double* pattern_transform);
/* create_shape(PLANE | CUBE | CYLINDER) (RayTracer.c:968): type = RT_SHAPE_*; minimum / maximum / closed are the
 * cylinder's fields (-INFINITY, INFINITY, 0 by default, :989-993) and ignored for the other types */
int rt_scene_add_shape(rt_scene_builder* b, int type, const double transform[16], const rt_material* m,
                       const rt_material_ext* x, const double* pattern_transform, double minimum, double maximum,
                       int closed);
int rt_scene_add_triangle(rt_scene_builder* b, const double p1[3], const double p2[3], const double p3[3],
                          const rt_material* m);
int rt_scene_add_triangle_ext(rt_scene_builder* b, const double p1[3], const double p2[3], const double p3[3],
                              const rt_material* m, const rt_material_ext* x, const double* pattern_transform);
/* n triangles, 9 floats each (p1 p2 p3), widened to double as load_model_file does (RayTracer.c:1657-1669) */
int rt_scene_add_triangles_f32(rt_scene_builder* b, const float* xyz9, int n, const rt_material* m);
/* `v` and `f` records of a Wavefront OBJ text (`len` bytes, no terminator needed); 3- or 4-vertex faces (quads
 * fan-split, RayTracer.c:1696-1697).  Face indices are 1-based and refer to vertices already read, as in the
 * reference's files; relative (negative) indices are refused with RT_ERR_INVALID. */
int rt_scene_load_obj(rt_scene_builder* b, const char* text, size_t len, const rt_material* m);
int rt_scene_load_obj_file(rt_scene_builder* b, const char* path, const rt_material* m);

/* override material.reflective of every triangle / every sphere added so far */
int rt_scene_set_reflective(rt_scene_builder* b, double tri_reflective, double sph_reflective);

/* Derive inverses, build the BVH (leaf_size <= 0: no BVH; at most 7, RT_ERR_TOO_LARGE beyond) and fill *out.  The
 * arrays *out points at belong to the builder and live until
 * rt_scene_destroy / the next rt_scene_finish. */
int rt_scene_finish(rt_scene_builder* b, int bvh_leaf_size, rt_scene* out);

/* ---- the reference's stock scenes (RayTracer.c), any image size ---- */
/* "spheres": render_complete_world :5848;  "cube": render_some_triangles :6727 with the teapot.h mesh;
 * "lighthouse": render_lighthouse_scene :6772 (camera + light only, add the mesh yourself);
 * "cube_spheres": the six spheres followed by the teapot.h mesh, spheres camera and light;
 * "refraction": render_refraction_scene :6491 (planes, checkers, spheres, mirror ball, cube, open cylinder);
 * "world_with_plane": render_complete_world_with_plane :6227 (planes, patterns, glass sphere);
 * "dual_spheres": render_dual_spheres_refracting_on_floor :6419 (nested glass spheres before a checkered wall). */
int rt_scene_stock(rt_scene_builder* b, const char* name, int hsize, int vsize);

/* ---- image output ---- */
/* ASCII P3 with the exact bytes of write_canvas_to_file (RayTracer.c:1333-1348) */
int rt_write_ppm(const char* path, const uint8_t* rgb, int width, int height);
/* same bytes into memory; returns the length, or the needed length if cap is too small */
size_t rt_format_ppm(char* dst, size_t cap, const uint8_t* rgb, int width, int height);

#ifdef __cplusplus
}
#endif
#endif
