/*
 * raytracer_hip - the reference's main() flow (RayTracer/RayTracer.c:7046-7254)
 * on top of the HIP library: build a stock scene, render it, write canvas.ppm.
 *
 *   raytracer_hip [scene] [--size W H] [--obj mesh.obj | --tris-f32 file] [--out canvas.ppm]
 *                 [--spp-grid G] [--tri-reflective K] [--sph-reflective K] [--brute | --bvh] [--device N]
 *                 [--host-ppm] [--reference-list] [--device-bvh] [--gpus N] [--frames K] [--json] [--no-count] [--tune]
 *
 * scene: lighthouse (default, like the reference's main), cube, spheres, cube_spheres, refraction,
 * world_with_plane, dual_spheres.
 * With no options it does what the reference binary does: lighthouse, 300x300, ./canvas.ppm,
 * reading ./lighthouse.obj.  The P3 text of the file is formatted on the GPU (rt_hip_render_ppm) unless
 * --host-ppm asks for the C writer rt_write_ppm; both emit the bytes of write_canvas_to_file (:1333).
 * --device-bvh skips the host's SAH tree and builds one on the GPU (rt_hip_build_bvh_device).
 * --gpus N (0 = every visible device) renders the frame on N GPUs of this process (rt_hip_render_multi: row bands,
 * one RCCL gather to device 0; a frame of fewer than RT_MULTI_GPU_MIN_SAMPLES samples is rendered on one device unless
 * N is given explicitly as 2 or more); --frames K repeats the render K times and reports the mean wall time per frame;
 * --json prints one line of figures for scripts.  A single frame is rendered with the tile plan as it stands
 * (RT_RENDER_NO_TUNE: like the reference, one render() per process); --frames K > 1 or --tune tunes the plan first
 * (rt_hip_tune) and reports what that took.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rt_hip.h"
#include "rt_scene.h"

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static int fail(const char* what, int rc) {
  fprintf(stderr, "%s: %s (%d) %s\n", what, rt_hip_strerror(rc), rc, rt_hip_last_error());
  return 1;
}

int main(int argc, char** argv) {
  const char* scene = "lighthouse";
  const char* obj = "lighthouse.obj";
  const char* tris_f32 = NULL;
  const char* out = "canvas.ppm";
  int W = 300, H = 300, G = 1, device = 0, accel = RT_ACCEL_AUTO, host_ppm = 0, device_bvh = 0, gpus = -1, frames = 1, json = 0, count = 1, tune = 0; /* HORIZONTAL_SIZE x VERTICAL_SIZE, :42-43 */
  double tri_refl = 0.0, sph_refl = 0.0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--size") && i + 2 < argc) { W = atoi(argv[++i]); H = atoi(argv[++i]); }
    else if (!strcmp(argv[i], "--obj") && i + 1 < argc) obj = argv[++i];
    else if (!strcmp(argv[i], "--tris-f32") && i + 1 < argc) tris_f32 = argv[++i];
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
    else if (!strcmp(argv[i], "--spp-grid") && i + 1 < argc) G = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--tri-reflective") && i + 1 < argc) tri_refl = atof(argv[++i]);
    else if (!strcmp(argv[i], "--sph-reflective") && i + 1 < argc) sph_refl = atof(argv[++i]);
    else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--brute")) accel = RT_ACCEL_BRUTE;
    else if (!strcmp(argv[i], "--bvh")) accel = RT_ACCEL_BVH;
    else if (!strcmp(argv[i], "--reference-list")) accel = RT_ACCEL_REFERENCE;
    else if (!strcmp(argv[i], "--host-ppm")) host_ppm = 1;
    else if (!strcmp(argv[i], "--device-bvh")) device_bvh = 1;
    else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--json")) json = 1;
    else if (!strcmp(argv[i], "--tune")) tune = 1;
    else if (!strcmp(argv[i], "--no-count")) count = 0; /* the production kernel from the first call on (no ray counters) */
    else if (argv[i][0] != '-') scene = argv[i];
    else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
  }

  rt_scene_builder* b = rt_scene_create();
  if (!b) return fail("rt_scene_create", RT_ERR_INVALID);
  int rc = rt_scene_stock(b, scene, W, H);
  if (rc != RT_OK) return fail("rt_scene_stock", rc);
  if (!strcmp(scene, "lighthouse")) {
    if (tris_f32) {
      FILE* f = fopen(tris_f32, "rb");
      if (!f) { perror(tris_f32); return 1; }
      long n = -1;
      if (fseek(f, 0, SEEK_END) == 0) n = ftell(f);
      if (n < 0 || fseek(f, 0, SEEK_SET) != 0) { perror(tris_f32); return 1; }
      float* v = malloc((size_t)n + 1);
      if (!v || fread(v, 1, (size_t)n, f) != (size_t)n) { perror(tris_f32); return 1; }
      fclose(f);
      rc = rt_scene_add_triangles_f32(b, v, (int)(n / (9 * sizeof(float))), NULL);
      free(v);
    } else {
      rc = rt_scene_load_obj_file(b, obj, NULL); /* load_model_file("lighthouse.obj"), :6784 */
    }
    if (rc != RT_OK) return fail("loading the mesh", rc);
  }
  if (tri_refl != 0.0 || sph_refl != 0.0) rt_scene_set_reflective(b, tri_refl, sph_refl);
  rt_scene s;
  if ((rc = rt_scene_finish(b, device_bvh ? 0 : 4, &s)) != RT_OK) return fail("rt_scene_finish", rc);

  /* --gpus 0: every visible device if the frame is worth it, else one */
  const int multi = gpus >= 2 || gpus == 1 || (gpus == 0 && (long long)W * H * G * G >= RT_MULTI_GPU_MIN_SAMPLES && rt_hip_device_count() > 1);
  if (frames < 1) frames = 1;
  if (frames > 1) tune = 1;
  double t_init = now_s(), init_ms = 0.0, upload_ms = 0.0;
  if (multi) {
    host_ppm = 1; /* the gathered image comes back as 8-bit rows; the C writer emits the file */
    if ((rc = rt_hip_multi_init(gpus)) != RT_OK) return fail("rt_hip_multi_init", rc);
    init_ms = (now_s() - t_init) * 1e3;
    if ((rc = rt_hip_multi_upload_scene(&s)) != RT_OK) return fail("rt_hip_multi_upload_scene", rc);
  } else {
    if ((rc = rt_hip_init(device)) != RT_OK) return fail("rt_hip_init", rc);
    init_ms = (now_s() - t_init) * 1e3;
    if ((rc = rt_hip_upload_scene(&s)) != RT_OK) return fail("rt_hip_upload_scene", rc);
  }
  upload_ms = (now_s() - t_init) * 1e3 - init_ms;
  if (device_bvh && !multi) {
    rt_bvh_device_info info;
    if ((rc = rt_hip_build_bvh_device(4, &info)) != RT_OK) return fail("rt_hip_build_bvh_device", rc);
    printf("BVH built on the device: %d nodes, depth %d, %.3f ms\n", info.n_nodes, info.depth, info.build_ms);
  }

  const size_t cap = host_ppm ? (size_t)W * H * 3 : rt_hip_ppm_capacity(W, H);
  uint8_t* buf = malloc(cap ? cap : 1); /* the 8-bit image, or the text of the file */
  if (!buf) return 1;
  size_t text_len = 0;
  rt_render_params p = {5 /* RECURSION_DEPTH */, G, accel, count, 1, 1, 0, tune ? 0 : RT_RENDER_NO_TUNE};
  rt_stats st;
  double tune_ms = 0.0;
  if (tune && !multi) {
    const double tt = now_s();
    if ((rc = rt_hip_tune(&p)) != RT_OK) return fail("rt_hip_tune", rc);
    tune_ms = (now_s() - tt) * 1e3;
    printf("tile plan tuned in %.3f ms\n", tune_ms);
  }
  printf("Starting Render Process\n");
  /* the frame that is timed and written is rendered by the production kernel; the ray counts printed below come
   * from a second pass with the counting build (same image, tested), which is not part of "Render Took" */
  rt_render_params p_timed = p;
  p_timed.count_rays = 0;
  double t0 = now_s();
  if (multi) {
    if ((rc = rt_hip_render_multi(&p_timed, buf, &st)) != RT_OK) return fail("rt_hip_render_multi", rc);
  } else if (host_ppm) {
    if ((rc = rt_hip_render(&p_timed, buf, &st)) != RT_OK) return fail("rt_hip_render", rc);
  } else {
    if ((rc = rt_hip_render_ppm(&p_timed, (char*)buf, cap, &text_len, &st)) != RT_OK) return fail("rt_hip_render_ppm", rc);
  }
  double dt = now_s() - t0;
  printf("\nRender Took %f Seconds\n", dt);
  if (count) {
    uint8_t* scratch_image = malloc((size_t)W * H * 3 + 1);
    rt_stats counted;
    if (!scratch_image) return 1;
    rc = multi ? rt_hip_render_multi(&p, scratch_image, &counted) : rt_hip_render(&p, scratch_image, &counted);
    free(scratch_image);
    if (rc != RT_OK) return fail("counting pass", rc);
    st.rays_primary = counted.rays_primary;
    st.rays_shadow = counted.rays_shadow;
    st.rays_secondary = counted.rays_secondary;
  }
  unsigned long long rays = st.rays_primary + st.rays_shadow + st.rays_secondary;
  printf("%llu rays (%llu primary, %llu shadow, %llu reflected), kernel %.3f ms, %.1f Mrays/s, %s\n", rays,
         (unsigned long long)st.rays_primary, (unsigned long long)st.rays_shadow,
         (unsigned long long)st.rays_secondary, st.kernel_ms, rays / (st.kernel_ms * 1e3),
         st.accel_used == RT_ACCEL_BVH ? "BVH" : (st.accel_used == RT_ACCEL_REFERENCE ? "reference list" : "brute force"));
  double per_frame_ms = dt * 1e3, kernel_ms = st.kernel_ms, exchange_ms = st.exchange_ms;
  if (frames > 1) { /* steady state: the same frame again and again, wall time per frame (image to host memory) */
    rt_render_params q = p;
    q.count_rays = 0;
    rt_stats s2;
    double ksum = 0.0, xsum = 0.0;
    for (int k = 0; k < 3; ++k) rc = multi ? rt_hip_render_multi(&q, buf, &s2) : rt_hip_render(&q, buf, &s2);
    t0 = now_s();
    for (int k = 0; k < frames && rc == RT_OK; ++k) {
      rc = multi ? rt_hip_render_multi(&q, buf, &s2) : rt_hip_render(&q, buf, &s2);
      ksum += s2.kernel_ms;
      xsum += s2.exchange_ms;
    }
    if (rc != RT_OK) return fail("render", rc);
    per_frame_ms = (now_s() - t0) * 1e3 / frames;
    kernel_ms = ksum / frames;
    exchange_ms = xsum / frames;
    printf("%d frames: %.4f ms per frame to host memory (kernel %.4f ms, exchange %.4f ms)\n", frames, per_frame_ms, kernel_ms,
           exchange_ms);
  }
  if (json)
    printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp_grid\": %d, \"gpus\": %d, \"rays\": %llu, "
           "\"init_ms\": %.3f, \"upload_ms\": %.3f, \"tune_ms\": %.3f, \"first_call_ms\": %.4f, \"first_call_kernel_ms\": %.4f, \"frames\": %d, "
           "\"ms_per_frame\": %.5f, \"kernel_ms\": %.5f, \"exchange_ms\": %.5f}\n",
           scene, W, H, G, multi ? rt_hip_multi_devices() : 1, rays, init_ms, upload_ms, tune_ms, dt * 1e3, st.kernel_ms, frames, per_frame_ms,
           kernel_ms, exchange_ms);
  printf("Writing Canvas To File\n");
  if (host_ppm) {
    if ((rc = rt_write_ppm(out, buf, W, H)) != RT_OK) return fail("rt_write_ppm", rc);
  } else {
    FILE* f = fopen(out, "wb");
    if (!f || fwrite(buf, 1, text_len, f) != text_len || fclose(f) != 0) { perror(out); return 1; }
  }
  free(buf);
  rt_hip_shutdown();
  rt_scene_destroy(b);
  return 0;
}
