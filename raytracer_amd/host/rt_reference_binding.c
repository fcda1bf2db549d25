/*
 * rt_reference_binding.c - the binding a maintainer of the reference adds to
 * RayTracer.c: render() on the MI355X instead of the CPU loop.
 *
 * This file is written against the reference's OWN types (camera, world,
 * shape, material, tuple, Mat4x4: RayTracer/RayTracer.c:45-96) and is meant to
 * be compiled inside the reference's translation unit, after its definitions:
 *
 *     #include "rt_reference_binding.c"      // at the end of RayTracer.c
 *     ... and the body of render() (RayTracer.c:1574) becomes
 *     void render(camera* c, world* w) { rt_render_hip(c, w); }
 *
 * It walks world->objects once (the list the reference walks per ray,
 * RayTracer.c:1047-1051), hands transforms / vertices / materials to the scene
 * builder of rt_scene.h, renders, and stores the colours where write_pixel()
 * (RayTracer.c:257) would have put them, so write_canvas_to_file() emits the
 * same canvas.ppm.
 */
#include "rt_hip.h"
#include "rt_scene.h"

static rt_material rt_binding_material(const material* m) {
  rt_material o;
  o.color[0] = m->color.x; o.color[1] = m->color.y; o.color[2] = m->color.z;
  o.ambient = m->ambient; o.diffuse = m->diffuse; o.specular = m->specular;
  o.shininess = m->shininess; o.reflective = m->reflective;
  return o;
}

/* transparency, refractive index and pattern of a `material` (RayTracer.c:57); NULL-equivalent when all default */
static rt_material_ext rt_binding_material_ext(const material* m) {
  rt_material_ext x = rt_material_ext_default();
  x.transparency = m->transparency;
  x.refractive_index = m->refractive_index;
  if (m->has_pattern) {
    x.pattern_type = (int)m->pattern.type; /* RT_PATTERN_* carry the values of enum pattern_type */
    x.pattern_from[0] = m->pattern.from.x; x.pattern_from[1] = m->pattern.from.y; x.pattern_from[2] = m->pattern.from.z;
    x.pattern_to[0] = m->pattern.to.x; x.pattern_to[1] = m->pattern.to.y; x.pattern_to[2] = m->pattern.to.z;
  }
  return x;
}

/* The flat form of (camera, world): *scene points into *builder, which the caller destroys afterwards.
 * No device is touched. */
int rt_binding_flatten(camera* c, world* w, rt_scene_builder** builder, rt_scene* scene) {
  rt_scene_builder* b = rt_scene_create();
  *builder = b;
  if (!b) return RT_ERR_INVALID;
  int rc = rt_scene_set_camera_view(b, (int)c->hsize, (int)c->vsize, c->field_of_view, &c->view_transform[0][0]);
  const double lp[3] = {w->lights->position.x, w->lights->position.y, w->lights->position.z};
  const double li[3] = {w->lights->intensity.x, w->lights->intensity.y, w->lights->intensity.z};
  if (rc == RT_OK) rc = rt_scene_set_light(b, lp, li);
  for (shape* sp = w->objects; sp && rc >= 0; sp = sp->next) {
    rt_material m = rt_binding_material(&sp->material);
    rt_material_ext x = rt_binding_material_ext(&sp->material);
    const int plain = !sp->material.has_pattern && sp->material.transparency == 0.0 && sp->material.refractive_index == 1.0;
    const rt_material_ext* xp = plain ? NULL : &x;
    const double* pt = sp->material.has_pattern ? &sp->material.pattern.transform[0][0] : NULL;
    if (sp->type == SPHERE) {
      rc = rt_scene_add_sphere_ext(b, &sp->transform[0][0], &m, xp, pt);
    } else if (sp->type == TRIANGLE) {
      const double p1[3] = {sp->p1.x, sp->p1.y, sp->p1.z}, p2[3] = {sp->p2.x, sp->p2.y, sp->p2.z},
                   p3[3] = {sp->p3.x, sp->p3.y, sp->p3.z};
      rc = rt_scene_add_triangle_ext(b, p1, p2, p3, &m, xp, pt);
    } else if (sp->type == PLANE || sp->type == CUBE) {
      rc = rt_scene_add_shape(b, sp->type == PLANE ? RT_SHAPE_PLANE : RT_SHAPE_CUBE, &sp->transform[0][0], &m, xp, pt,
                              -INFINITY, INFINITY, 0);
    } else if (sp->type == CYLINDER) {
      rc = rt_scene_add_shape(b, RT_SHAPE_CYLINDER, &sp->transform[0][0], &m, xp, pt, sp->minimum, sp->maximum, sp->closed);
    } else {
      rc = RT_ERR_UNSUPPORTED;
    }
  }
  if (rc >= 0) rc = rt_scene_finish(b, 4, scene);
  return rc < 0 ? rc : RT_OK;
}

/* The 8-bit rows of a gathered frame into canvas[][]: the centre of each 8-bit step, which color_convert
 * (RayTracer.c:1322) maps back to the same byte, so write_canvas_to_file() emits the same file. */
static void rt_binding_store_rgb(const uint8_t* rgb) {
  for (int y = 0; y < VERTICAL_SIZE; ++y)
    for (int x = 0; x < HORIZONTAL_SIZE; ++x) {
      const uint8_t* px = rgb + ((size_t)y * HORIZONTAL_SIZE + x) * 3;
      tuple col = {(px[0] + 0.5) / 255.0, (px[1] + 0.5) / 255.0, (px[2] + 0.5) / 255.0, 1.0};
      write_pixel(x, y, col);
    }
}

/* returns RT_OK or a negative rt_status; on an error the caller keeps its CPU loop.
 * One device is the default: the reference's use is one process and one render() (RayTracer.c:7236-7251), and a
 * frame of HORIZONTAL_SIZE x VERTICAL_SIZE pixels is far below what several GPUs pay for
 * (RT_MULTI_GPU_MIN_SAMPLES).  RT_HIP_GPUS=n (n > 1) in the environment asks for n devices; if that path fails
 * for any reason (no RCCL, fewer devices, a failing exchange) the frame is rendered on device 0 after all.
 * One frame per process: the plan is not tuned (RT_RENDER_NO_TUNE). */
int rt_render_hip(camera* c, world* w) {
  if ((int)c->hsize != HORIZONTAL_SIZE || (int)c->vsize != VERTICAL_SIZE) return RT_ERR_UNSUPPORTED; /* canvas[][] is fixed */
  rt_scene_builder* b = NULL;
  rt_scene s;
  int rc = rt_binding_flatten(c, w, &b, &s);
  if (rc != RT_OK) { rt_scene_destroy(b); return rc; }
  rt_render_params p = {RECURSION_DEPTH, 1, RT_ACCEL_AUTO, 0, 1, 1, 0, RT_RENDER_NO_TUNE};
  const char* asked = getenv("RT_HIP_GPUS");
  const int gpus = asked ? atoi(asked) : 1;
  if (gpus > 1 && rt_hip_device_count() >= gpus) {
    /* every device renders its row bands, one RCCL gather brings the 8-bit rows to device 0 */
    static uint8_t rgb[HORIZONTAL_SIZE * VERTICAL_SIZE * 3];
    int mrc = rt_hip_multi_init(gpus);
    if (mrc == RT_OK) mrc = rt_hip_multi_upload_scene(&s);
    if (mrc == RT_OK) mrc = rt_hip_render_multi(&p, rgb, NULL);
    if (mrc == RT_OK) {
      rt_binding_store_rgb(rgb);
      rt_scene_destroy(b);
      return RT_OK;
    }
    fprintf(stderr, "rt_render_hip: %d GPUs failed (%s: %s); rendering on one\n", gpus, rt_hip_strerror(mrc), rt_hip_last_error());
  }
  rc = rt_hip_init(0);
  if (rc == RT_OK) rc = rt_hip_upload_scene(&s);
  if (rc == RT_OK) {
    static double colours[HORIZONTAL_SIZE * VERTICAL_SIZE * 3];
    rc = rt_hip_render_f64(&p, colours, NULL);
    if (rc == RT_OK)
      for (int y = 0; y < VERTICAL_SIZE; ++y)
        for (int x = 0; x < HORIZONTAL_SIZE; ++x) {
          const double* px = colours + ((size_t)y * HORIZONTAL_SIZE + x) * 3;
          tuple col = {px[0], px[1], px[2], 1.0};
          write_pixel(x, y, col);
        }
  }
  rt_scene_destroy(b);
  return rc;
}
