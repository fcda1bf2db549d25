/*
 * rt_api.hip - implementation of the C ABI in include/rt_hip.h.
 *
 * Replaces the call render(camera*, world*) of the reference
 * (RayTracer/RayTracer.c:1574): rt_hip_upload_scene takes the flattened world
 * once, rt_hip_render runs the trace kernels of rt_kernels.hip.inc over the
 * image and returns the quantised pixels.
 */
#include "rt_hip.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <chrono>
#include <thread>
#include <vector>

#include "rt_kernels.hip.inc"
#include "rt_general.hip.inc"
#include "rt_ppm.hip.inc"
#include "rt_lbvh.hip.inc"
#include "rt_sort.h"

/* Scratch of rt_hip_format_ppm_device_noalloc: the text length of every chunk of PPM_CHUNK pixels, then of every
 * group of chunks.  One copy per GPU, held until the process ends (256 KiB); its address on a device comes from
 * hipGetSymbolAddress with that device current (create_context), never from the name. */
constexpr int kNoAllocMaxChunks = 32768; /* x PPM_CHUNK pixels = 67,108,864 pixels, e.g. 8192 x 8192 */
static __device__ __align__(256) unsigned ppm_scratch[2 * (kNoAllocMaxChunks + 1)];

namespace {

/* What render_enqueue launched, for render_collect. */
struct Enqueued {
  int rows = 0;
  int accel_used = RT_ACCEL_BRUTE;
  bool timed = false, count = false;
};

struct Context {
  bool ready = false;
  bool have_scene = false;
  int device = -1;
  int n_cu = 256;
  int light_per_block = 20; /* background-only tiles per workgroup (filled five at a time) */
  bool sphere_masks = true; /* worlds with up to 64 spheres: masks for them as well */
  bool shadow_masks = true; /* ... and their shadow rays only the triangles that can lie between those and the light */
  bool tile_masks = true;   /* meshes of up to 64 triangles: camera rays test only the triangles that cover their tile */
  bool tile_lists = true;   /* larger meshes walked exhaustively: per-tile lists of those triangles instead of masks */
  bool shadow_lists = true; /* with the camera rays' lists: per block of tiles, the triangles that can lie between what those rays see and the light */
  /* Only the exhaustive loop uses these lists.  Frames that walk the tree were measured with them as well: the tiles
   * whose waves a frame waits for have the longest lists (lighthouse 1080p: up to 790 possible blockers for a block of
   * the lantern room, 140 on average); with lists of up to 64 .. 192 the 1080p frames stayed within 1 % and 4K gained
   * 5 %, with lists of up to 320 they took 75 % longer (a walk leaves out what lies behind the ray's start or beyond
   * the light box by box; in a list every triangle whose plane the ray's line crosses inside its outline costs the
   * double-precision routine), and the support for it cost the walking kernels 70 spilled scalar registers.  Lists per
   * single tile are no shorter where it matters (longest 686). */
  bool list_lds = true;     /* a mesh tile's workgroup copies its list to LDS (false: read through scalar loads) */
  /* Launches that walk a tree: the camera rays of a tile with at most this many triangles in its list run down the list
   * instead (0: never).  A launch of many mesh tiles is bound by throughput, and the list costs less work than the walk
   * up to a few hundred triangles (4K lighthouse 80 -> 65 us).  A launch whose mesh tiles are all resident at once lasts
   * as long as its slowest wave, which sits in the most crowded tile: there a long list takes as long as the walk
   * (about 300 cycles per triangle of the list against 70 node steps and 16 leaves), so only the short ones are used
   * (1080p lighthouse 57.5 us with no lists, 52 with lists of up to 64, 55.7 up to 128, 58 up to 256). */
  int list_max_walk = 256, list_max_walk_few = 256;
  bool cover_tiles = true;  /* find the mesh tiles by rasterising the triangles (false: by the rectangle of the mesh's bounds) */
  bool tile_plan = true;   /* order tiles mesh-first (scheduling hint) */
  bool use_lds = true;
  bool lds_stack = true;
  bool lds_nodes = true;  /* a small tree's inner nodes are copied to LDS (development switch RT_HIP_LDS_NODES) */
  int many_per_cu = 2;    /* the plain walk over a tree in LDS keeps its LDS to a third of a CU's from this many mesh workgroups per CU on
                             (development switch RT_HIP_MANY_PER_CU).  Since the walk fits the registers of three waves per SIMD in either
                             build, two workgroups per CU of mesh tiles are enough for the third to pay: 1080p lighthouse 28.8 -> 27.7 us,
                             1440p 36.0 -> 31.9 us with 2 instead of 4 (with 1: 720p 22.1 -> 22.9 us; always: 300x300 18.3 -> 19.1 us) */
  bool crowded_first = true; /* mesh tiles in descending order of the triangles that project into them */
  int split_tiles = 256, split_parts = 2, split_min_cost = 16, split_quarters = 8, split_quarters_reflect = 128, split_max_per_cu = 4; /* the most crowded tiles run as 2 or 4 workgroups with a half or a quarter of the lanes each */
  bool part_lists = true;   /* parts of split tiles get lists of their own (walks) */
  int part_min_lg = 2;      /* entries split into at least 2^this parts get lists of their own */
  int coop_min_lg = 2;      /* ... and from 2^this parts on whole waves render them (RenderArgs.coop_min_lg; 0: never) */
  int coop_cam_per_lane = 16;    /* such a part's camera rays get a list of up to this many triangles per lane */
  int coop_shadow_per_lane = 32; /* such a part's shadow rays run down a list of up to this many possible blockers per lane
                                    that holds one of its pixels (2^lg lanes), instead of walking */
  int walk_shadow_max = 0;  /* walks: a whole tile's shadow rays run down their block's list if it has at most this many triangles */
  int part_shadow_max = 0;  /* ... and lists for their shadow rays up to this length (0: none, the default: with lists of
                               up to 48 or 96 the lighthouse frames are the same to the microsecond, and making them costs
                               3-4 ms when the tiles are planned; from 256 up the frames get slower) */
  bool split_narrow = true; /* the refinement ends with rounds that split only the few slowest entries (refine_split) */
  double split_threshold = 0.85; /* the refinement splits the entries whose slowest wave ran longer than this share of the slowest of all */
  int split_max_lg = 6;     /* the refinement splits a tile into at most 2^this parts (6: one pixel per wave); 4K
                               lighthouse with mirrors 135 us with 4, 126 us with 5 or 6, 1080p the same with all three */
  bool split_resort = true; /* after the refinement: the winning list once more, in the order of its own measurement */
  int split_refine = 10;    /* launches that measure their waves and split the slowest tiles further */
  int xcd_tiles = -1;      /* mesh tiles in per-XCD runs: 0 never, 1 always, -1 when the tree is read from memory */
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr; /* per-call kernel time (rt_stats) */
  hipEvent_t tm0 = nullptr, tm1 = nullptr; /* caller-driven timer */
  DevScene scene{};
  bool reflective = false;
  bool bvh_single_leaf = false;
  rt_camera host_camera{};
  void* d_spheres = nullptr;
  void* d_tris = nullptr;
  void* d_normals = nullptr;
  void* d_materials = nullptr;
  void* d_bvh = nullptr;
  void* d_trif = nullptr;  /* single-precision forms of the triangles (same order as d_tris) and spheres */
  void* d_sphf = nullptr;
  double filter_radius = 0.0; /* every ray origin and vertex lies within this distance of the world origin */
  DevScene* d_scene = nullptr; /* device copy of `scene` */
  double mesh_lo[3] = {0, 0, 0}, mesh_hi[3] = {0, 0, 0}; /* world bounds of all triangles */
  double box_pad = 0.0; /* what every single-precision box is grown by */
  std::vector<float> tri_rect; /* pixel rectangle (x0 y0 x1 y1) of every triangle as the camera sees it; empty: unknown */
  std::vector<double> tri_world; /* meshes of up to 64 triangles: the corners in world space (9 per triangle) */
  std::vector<double> mesh_world; /* with host_trif (meshes of up to kMaxListTris triangles, no spheres): the same for every triangle */
  /* worlds of up to 64 spheres: rows 0..2 of every inverse transform, its Frobenius norm, and the eight corners of the
   * box that holds the sphere (the image of [-1, 1]^3; corner v has +1 where bit 0 / 1 / 2 of v is set) in world space
   * and as the camera sees them (pixels; NaN: a corner beside or behind the camera) */
  std::vector<double> sph_inv, sph_norm, sph_world, sph_xy;
  double host_light[3] = {0, 0, 0};
  std::vector<int32_t> sph_upload_index; /* device sphere k is the caller's sphere sph_upload_index[k] */
  std::vector<DevTriF> host_trif; /* meshes of up to kMaxListTris triangles without spheres: the single-precision forms as
                                     uploaded (the per-tile lists of plan_tiles hold copies of them) */
  std::vector<int32_t> host_tri_order; /* with host_trif: every triangle's position in the world's object list (the tie-break of hit()) */
  std::vector<float> host_tri_k;  /* with host_trif: (p1 - camera origin) . (e2 x e1) of every triangle (the numerator of t for a camera ray) */
  std::vector<float> tri_depth;   /* with tri_xy: how far in front of the camera the nearest corner of every triangle lies */
  std::vector<double> tri_xy;  /* where the camera sees the three corners of every triangle (x y x y x y, in pixels, pixel
                                  k covering [k, k + 1)); empty: unknown, or a corner beside or behind the camera */
  /* tile orders of recent launches and the parameters each was made for (scene, bands, kernel family, overlap
   * mode): a caller that alternates between a few kinds of launch does not re-plan every time */
  struct TilePlan {
    long long key[6] = {-1, -1, -1, -1, -1, -1};
    int32_t* d_list = nullptr;
    size_t cap = 0;
    int heavy = 0, tiles = 0, entries = 0;
    bool light_background = false; /* the light tiles lie outside the mesh's pixel rectangle */
    int light_per_block = 1;       /* light tiles per workgroup */
    size_t list_forms = 0, list_total = 0; /* RenderArgs.list_forms / list_total of launches with this plan (0: no lists) */
    std::vector<std::vector<int32_t>> tile_tris; /* by tile number: the triangles of its list (kept while refinements are pending) */
    int list_max = 0;                            /* tiles with longer lists have none (walks only) */
    int list_longest = 0;                        /* the longest list of a mesh tile */
    std::vector<std::vector<int32_t>> tile_shadow; /* by tile number: the triangles of its shadow rays' list (with tile_tris) */
    std::vector<unsigned char> tile_shadow_ok;     /* by tile number: 1 = tile_shadow holds every possible blocker, 0 = unknown */
    int shadow_max = 0, shadow_longest = 0;        /* as list_max / list_longest */
    struct PartLists { std::vector<int32_t> cam, shadow; bool own = false, has_shadow = false; long long start = -1, sstart = -1; };
    std::unordered_map<int32_t, PartLists> part_cache; /* by entry: the lists of parts made so far (while refinements are pending) */
    bool part_lists = false;       /* walks: a part of a tile split into four or more gets lists of its own rectangle */
    int part_shadow_max = 0;       /* ... and a shadow rays' list if that has at most this many triangles */
    int coop_min_lg = 0, coop_shadow_per_lane = 0; /* as in the context when the plan was made */
    int rows = 0, band_rows = 1, n_ranks = 1, rank = 0; /* the launch's packed rows and how they map to image rows */
    size_t mask_offset = 0;        /* d_list + mask_offset: four 64-bit masks per entry (0: none): the triangles a camera
                                      ray of the tile can hit, those a shadow ray from such a hit can, and the same for spheres */
    int refine_left = 0;           /* measured refinements of the split still to do (see refine_split) */
    bool refining = false;         /* refine_split is at work on this plan */
    std::vector<int32_t> host_list; /* the list as uploaded, kept while refinements are pending */
    std::vector<int32_t> debug_words; /* development build, host-only planning: what store_plan would have copied to the device */
    unsigned long long last_use = 0;
  };
  static constexpr int kPlans = 4;
  TilePlan plans[kPlans];
  unsigned long long plan_clock = 0;
  unsigned long long scene_serial = 0;
  const int32_t* launch_tile_list = nullptr; /* argument of the launch being issued */
  const unsigned long long* launch_tile_mask = nullptr; /* likewise: per-entry pairs of triangle masks, or nullptr */
  TilePlan* launch_plan = nullptr;           /* the plan it came from (plan_tiles), nullptr for an unplanned launch */
  size_t launch_lds = 0;                     /* dynamic LDS per workgroup of the last launch() */
  int launch_waves = 0;                      /* waves per SIMD its kernel is built for (walks; 0: not a walk) */
  unsigned long long* launch_counters = nullptr; /* counters argument of the launch being issued (nullptr: d_counters) */
  void* d_tri_primary = nullptr;
  void* d_sph_primary = nullptr;
  unsigned long long* d_counters = nullptr;
  unsigned char* d_rgb = nullptr;  /* library-owned output for rt_hip_render */
  size_t d_rgb_bytes = 0;
  double* d_canvas = nullptr;
  size_t d_canvas_bytes = 0;
  /* worlds with planes, cubes, cylinders, patterns or transparency: the general kernel's copy of the scene */
  bool general = false;
  int gen_list_cap = 100;
  int gen_hsize = 0, gen_vsize = 0;
  GenScene* d_gen_scene = nullptr;
  GenObject* d_gen_objects = nullptr;
  GenMaterial* d_gen_materials = nullptr;
  /* P3 formatter: per-chunk lengths and offsets, library-owned text buffer, pinned word for the length */
  unsigned* d_ppm_len = nullptr;
  unsigned* d_ppm_group = nullptr;
  size_t ppm_chunks_cap = 0;
  unsigned char* d_ppm_text = nullptr;
  size_t d_ppm_text_bytes = 0;
  unsigned long long* h_ppm_total = nullptr; /* pinned */
  unsigned long long* d_ppm_total_host = nullptr; /* the same word as the device addresses it */
  hipStream_t ppm_stream = nullptr;          /* stream of the last formatting call */
  bool ppm_pending = false;
  hipEvent_t ppm_ev = nullptr;               /* orders formatting calls on different streams: they share the arrays above */
  /* rt_hip_render_ppm_submit / _collect: two frames in flight, each with its image, its text on the device and in
   * pinned host memory, its length word and its event; the text travels on a stream of its own */
  struct PpmSlot {
    unsigned char* d_rgb = nullptr;
    unsigned char* d_text = nullptr;
    char* h_text = nullptr;                  /* pinned */
    unsigned long long* h_len = nullptr;     /* pinned */
    unsigned long long* d_len = nullptr;     /* the same word as the device addresses it */
    size_t rgb_bytes = 0, text_bytes = 0;
    hipEvent_t formatted = nullptr;
    bool busy = false;
    Enqueued enq;
    hipEvent_t k0 = nullptr, k1 = nullptr;   /* around the trace kernel, for rt_stats */
  };
  PpmSlot ppm_slot[2];
  hipStream_t copy_stream = nullptr;
  unsigned* ppm_scratch_ptr = nullptr;       /* this device's copy of ppm_scratch */
  hipStream_t noalloc_stream = nullptr;      /* stream of the last rt_hip_format_ppm_device_noalloc call */
  bool noalloc_pending = false;
};

/* One context per HIP device of this process; `cur` is the one the entry points act on (rt_hip_init and
 * rt_hip_set_device choose it). */
constexpr int kMaxDevices = 16;
Context g_ctx[kMaxDevices];
Context* cur = &g_ctx[0];
char g_err[512] = "";

bool hip_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return false;
}
#define HIP_TRY(call)                       \
  do {                                      \
    if (!hip_ok((call), #call)) return RT_ERR_HIP; \
  } while (0)

void free_general() {
  if (cur->d_gen_scene) (void)hipFree(cur->d_gen_scene);
  if (cur->d_gen_objects) (void)hipFree(cur->d_gen_objects);
  if (cur->d_gen_materials) (void)hipFree(cur->d_gen_materials);
  cur->d_gen_scene = nullptr;
  cur->d_gen_objects = nullptr;
  cur->d_gen_materials = nullptr;
  cur->general = false;
}

void free_scene() {
  free_general();
  if (cur->d_spheres) (void)hipFree(cur->d_spheres);
  if (cur->d_tris) (void)hipFree(cur->d_tris);
  if (cur->d_normals) (void)hipFree(cur->d_normals);
  if (cur->d_materials) (void)hipFree(cur->d_materials);
  if (cur->d_bvh) (void)hipFree(cur->d_bvh);
  if (cur->d_tri_primary) (void)hipFree(cur->d_tri_primary);
  if (cur->d_sph_primary) (void)hipFree(cur->d_sph_primary);
  if (cur->d_trif) (void)hipFree(cur->d_trif);
  if (cur->d_sphf) (void)hipFree(cur->d_sphf);
  cur->d_spheres = cur->d_tris = cur->d_normals = cur->d_materials = cur->d_bvh = cur->d_tri_primary = cur->d_sph_primary = nullptr;
  cur->d_trif = cur->d_sphf = nullptr;
  cur->have_scene = false;
}

/* Development build: rt_hip_debug_plan_host runs the host side of an upload and of the planner without a device
 * (nothing is allocated, copied or launched), so that what the planner decides can be tested where there is no GPU. */
bool g_host_only = false;

int upload(void** dst, const void* src, size_t bytes) {
  if (g_host_only) { *dst = nullptr; return RT_OK; }
  if (bytes == 0) bytes = 16;
  HIP_TRY(hipMalloc(dst, bytes));
  if (src) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return RT_OK;
}

/* The single-precision form of a sphere.  inv: rows 0..2 of inverse(transform); origin: the camera origin;
 * radius as for rt_make_trif.  The filter compares |o' x d'|^2 with |d'|^2 in object space rotated by Q^T (inv = Q U,
 * DevSphereF): norms and products are those of inv o + t and inv d, the bounds below hold for the shorter sums a fortiori.  With eps = 2^-24, Mn = the Frobenius norm
 * of the linear part of inv, tn = |translation|, Omax = Mn radius + tn >= |o'|, |d| = 1:
 *   |o'_f - o'| <= 6 eps Omax, |d'_f - d'| <= 5 eps Mn (four resp. three fused products of once-rounded values),
 *   |(o' x d')_f - o' x d'| <= 6 eps Omax Mn + 5 eps Omax Mn + 7 eps Omax Mn = 18 eps Omax Mn =: Ec.
 * |c| >= |c_f| - Ec and |d'| <= |d'_f| + 5 eps Mn, so |c_f|^2 > |d'_f|^2 + 2 Mn E + E^2 with
 * E = 24 eps max(Omax, 1) max(Mn, 1) >= Ec + 5 eps Mn proves |c| > |d'|; the kernel adds a relative 1e-5 for the
 * rounding of the two squared norms and of the reference's own discriminant. */
void make_spheref(const double* inv, const double* origin, double radius, DevSphereF* out) {
  double mn2 = 0.0, tn2 = 0.0;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) mn2 += inv[4 * r + c] * inv[4 * r + c];
    tn2 += inv[4 * r + 3] * inv[4 * r + 3];
  }
  const double mn = std::sqrt(mn2), tn = std::sqrt(tn2);
  const double omax = std::fmax(mn * radius + tn, 1.0);
  const double e = 24.0 * 5.9604644775390625e-8 * omax * std::fmax(mn, 1.0);
  const double k = (2.0 * mn * e + e * e) * 1.0001 + 1e-30;
  /* inv's linear part A = Q U by Gram-Schmidt on its columns (A is invertible: the upload has inverted it) */
  double a[3][3], q[3][3], u[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) a[c][r] = inv[4 * r + c]; /* a[c] = column c */
  bool qr_ok = true;
  for (int c = 0; c < 3; ++c) {
    double v[3] = {a[c][0], a[c][1], a[c][2]};
    for (int pass = 0; pass < 2; ++pass) /* twice: keeps Q orthogonal to rounding for the flattened spheres too */
      for (int j = 0; j < c; ++j) {
        const double p = q[j][0] * v[0] + q[j][1] * v[1] + q[j][2] * v[2];
        u[j][c] += p;
        for (int r = 0; r < 3; ++r) v[r] -= p * q[j][r];
      }
    const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(len > 1e-9 * mn) || !std::isfinite(len)) { qr_ok = false; break; }
    u[c][c] = len;
    for (int r = 0; r < 3; ++r) q[c][r] = v[r] / len;
  }
  if (qr_ok) { /* Q U must give A back and Q must be orthogonal, both far inside the test's margins */
    double worst = 0.0;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) {
        double back = 0.0, dot = 0.0;
        for (int j = 0; j < 3; ++j) { back += q[j][r] * u[j][c]; dot += q[c][j] * q[r][j]; }
        worst = std::fmax(worst, std::fabs(back - a[c][r]) / mn);
        worst = std::fmax(worst, std::fabs(dot - (c == r ? 1.0 : 0.0)));
      }
    qr_ok = worst < 1e-11;
  }
  double tq[3] = {0, 0, 0};
  if (qr_ok) {
    for (int j = 0; j < 3; ++j) tq[j] = q[j][0] * inv[3] + q[j][1] * inv[7] + q[j][2] * inv[11];
  }
  out->u[0] = (float)u[0][0]; out->u[1] = (float)u[0][1]; out->u[2] = (float)u[0][2];
  out->u[3] = (float)u[1][1]; out->u[4] = (float)u[1][2]; out->u[5] = (float)u[2][2];
  for (int r = 0; r < 3; ++r) {
    out->t[r] = (float)tq[r];
    out->po[r] = (float)(((origin[0] * u[r][0] + origin[1] * u[r][1]) + origin[2] * u[r][2]) + tq[r]);
  }
  out->k = (omax * mn < 1e15 && k < 1e30) ? (float)k : INFINITY; /* NaN compares false: infinite */
  /* for the "outside and leading away" test of sphere_filter: norms rounded up, and the camera origin's own c,
   * which the host knows in double precision (with the reference's expression, RayTracer.c:927) */
  out->mn = std::nextafterf((float)mn, INFINITY);
  out->tn = std::nextafterf((float)tn, INFINITY);
  double o[3];
  for (int r = 0; r < 3; ++r) o[r] = ((origin[0] * inv[4 * r] + origin[1] * inv[4 * r + 1]) + origin[2] * inv[4 * r + 2]) + inv[4 * r + 3];
  const double ss = (o[0] * o[0] + o[1] * o[1]) + o[2] * o[2];
  out->primary_outside = (ss - 1.0 > 2e-3 * ss && ss < 1e30) ? 1.0f : 0.0f;
  /* A shadow ray that starts on this sphere starts at inv (p + EPSILON n) = o' + EPSILON inv n with n = the unit
   * vector along inv^T o', i.e. c = 2 EPSILON |inv^T o'| + (rounding of the hit point, ~4e-16 Mn radius) with
   * |inv^T o'| >= the least singular value of inv >= |det inv| / Mn^2.  The rule "such a ray that leads away from
   * the centre meets nothing" is used only where the first term is a thousand times the second. */
  const double det = inv[0] * (inv[5] * inv[10] - inv[6] * inv[9]) - inv[1] * (inv[4] * inv[10] - inv[6] * inv[8]) +
                     inv[2] * (inv[4] * inv[9] - inv[5] * inv[8]);
  const double sigma_min = mn > 0 ? std::fabs(det) / (mn * mn) : 0.0;
  out->self_ok = (2.0 * RT_EPSILON * sigma_min > 1e3 * 4e-16 * mn * std::fmax(radius, 1.0) && sigma_min > 1e-6) ? 1.0f : 0.0f;
  if (!qr_ok) { /* every comparison of sphere_f32 is then false: the double-precision routine decides */
    out->k = INFINITY;
    out->mn = INFINITY;
    out->primary_outside = out->self_ok = 0.0f;
  }
}

int bvh_depth(const rt_bvh_node* nodes, int n, int i, int guard) {
  if (i < 0 || i >= n || guard > 4096) return 1 << 20;
  if (nodes[i].right < 0) return 1;
  if (nodes[i].left <= i || nodes[i].right <= i) return 1 << 20; /* children come after their parent */
  const int l = bvh_depth(nodes, n, nodes[i].left, guard + 1), r = bvh_depth(nodes, n, nodes[i].right, guard + 1);
  return 1 + (l > r ? l : r);
}

#if RT_BVH_WIDE
/* The caller's binary tree as four-wide nodes.  A wide node starts as the two children of a binary node; while it has
 * fewer than four slots, the inner child with the largest surface area is replaced by its own two children (the
 * child a ray is most likely to enter is the one worth opening).  Leaves stay as they are.  Boxes are the callers'
 * padded by `pad` and rounded outwards.  stack_need: the most subtrees a walk can have postponed at once (a step
 * postpones all but one of the children it enters).  Returns the index of the wide node made for binary node b. */
struct WideBuild {
  const rt_bvh_node* nodes;
  double pad;
  std::vector<DevBvh4>* out;
};
int wide_node(const WideBuild& w, int b, int& stack_need) {
  const rt_bvh_node* nodes = w.nodes;
  int slot[4] = {nodes[b].left, nodes[b].right, -1, -1};
  int k = 2;
  auto area = [&](int i) {
    const double dx = (double)nodes[i].hi[0] - nodes[i].lo[0], dy = (double)nodes[i].hi[1] - nodes[i].lo[1], dz = (double)nodes[i].hi[2] - nodes[i].lo[2];
    return dx * dy + dy * dz + dz * dx;
  };
  while (k < 4) {
    int open = -1;
    for (int j = 0; j < k; ++j)
      if (nodes[slot[j]].right >= 0 && (open < 0 || area(slot[j]) > area(slot[open]))) open = j;
    if (open < 0) break;
    const int c = slot[open];
    slot[open] = nodes[c].left;
    slot[k++] = nodes[c].right;
  }
  const int me = (int)w.out->size();
  w.out->push_back(DevBvh4{});
  float box[6][4];
  int32_t ref[4];
  int deepest = 0;
  for (int j = 0; j < 4; ++j) {
    if (j >= k) {
      for (int c = 0; c < 6; ++c) box[c][j] = NAN;
      ref[j] = RT_BVH_EMPTY_REF;
      continue;
    }
    const rt_bvh_node& n = nodes[slot[j]];
    for (int c = 0; c < 3; ++c) {
      box[c][j] = std::nextafterf((float)((double)n.lo[c] - w.pad), -INFINITY);
      box[3 + c][j] = std::nextafterf((float)((double)n.hi[c] + w.pad), INFINITY);
    }
    if (n.right >= 0) {
      int need = 0;
      ref[j] = wide_node(w, slot[j], need);
      deepest = std::max(deepest, need);
    } else {
      ref[j] = ~((n.left << 3) | (-n.right));
    }
  }
  DevBvh4& d = (*w.out)[(size_t)me];
  for (int c = 0; c < 6; ++c) d.q[c] = make_float4(box[c][0], box[c][1], box[c][2], box[c][3]);
  float f[4];
  memcpy(f, ref, sizeof f);
  d.q[6] = make_float4(f[0], f[1], f[2], f[3]);
  d.q[7] = make_float4(0.f, 0.f, 0.f, 0.f);
  stack_need = k - 1 + deepest;
  return me;
}

#endif

/* LDS staging limits: small meshes (brute force) and BVH node arrays that leave room for several
 * workgroups per CU */
constexpr int kMaxLdsTris = 320;      /* 320 * 64 B = 20 KiB of single-precision forms beside the 18 KiB the kernel keeps for values
                                         that wait out its loops: four workgroups still fit a CU */
constexpr int kMaxReferenceObjects = 1 << 16; /* worlds up to this size keep a second device copy for RT_ACCEL_REFERENCE */
constexpr size_t kMaxLdsPerWorkgroup = 79 * 1024; /* two workgroups per CU, beside the walks' 1 KiB of static LDS */
constexpr int kShadowBlock = 4;            /* shadow masks are worked out per block of this many tiles squared */
constexpr int kMaxLdsShadowForms = 320;    /* ... and a list of its shadow rays of up to this many (21 KiB) */
constexpr int kMaxLdsListForms = 384;      /* a mesh tile's workgroup keeps a list of up to this many forms in LDS (24 KiB) */
constexpr int kMaxListTris = 1 << 16;      /* meshes up to this size get per-tile triangle lists for their camera rays ... */
constexpr size_t kMaxListEntries = 1 << 20; /* ... unless the lists of a launch would hold more than this many forms (64 MiB) */
constexpr int kMaxCoverTris = 1 << 16;     /* meshes up to this size are rasterised into the tile grid to find the tiles they cover */
constexpr int kMaxTilePlanTris = 1 << 22;  /* meshes up to this size get their tiles ordered by how crowded they are */
constexpr int kAutoDeviceBvhTris = 4096; /* meshes from this size up never stay without a BVH */
constexpr int kMaxLdsBvhNodes = 48 * 1024 / (int)sizeof(DevBvhNode);  /* 48 KiB of inner nodes (~3000 triangles) */

template <bool COUNT, bool BVH, bool REFLECT, int STAGE, int KIND>
void launch_one(size_t lds, dim3 grid, hipStream_t s, const RenderArgs& a, unsigned char* rgb, double* canvas) {
  hipLaunchKernelGGL((rt_trace_kernel<COUNT, BVH, REFLECT, STAGE, KIND>), grid, dim3(RT_TILE_W * RT_TILE_H), lds, s, cur->d_scene,
                     (const DevSphere*)cur->d_spheres, (const DevSpherePrimary*)cur->d_sph_primary, (const DevTri*)cur->d_tris,
                     (const DevTriPrimary*)cur->d_tri_primary, (const double*)cur->d_normals, (const rt_material*)cur->d_materials,
                     (const DevBvhNode*)cur->d_bvh, (const DevTriF*)cur->d_trif, (const DevSphereF*)cur->d_sphf, a,
                     cur->launch_tile_list, cur->launch_tile_mask, rgb, canvas,
                     cur->launch_counters ? cur->launch_counters : cur->d_counters);
}
/* kernels specialised for the primitive kinds the scene holds (KIND 1: spheres only, 2: triangles only, 0: both) */
template <bool BVH, bool REFLECT, int STAGE>
void launch_kind(int kind, size_t lds, dim3 grid, hipStream_t s, const RenderArgs& a, unsigned char* rgb, double* canvas) {
  if (kind == 2) launch_one<false, BVH, REFLECT, STAGE, 2>(lds, grid, s, a, rgb, canvas);
  else launch_one<false, BVH, REFLECT, STAGE, 0>(lds, grid, s, a, rgb, canvas);
}
template <bool REFLECT>
void launch_family(bool bvh, int stage, int kind, size_t lds, dim3 grid, hipStream_t s, const RenderArgs& a, unsigned char* rgb,
                   double* canvas) {
  if (bvh) { /* a walk always has triangles */
    if (stage == 3) launch_kind<true, REFLECT, 3>(kind, lds, grid, s, a, rgb, canvas);
    else if (stage == 5) launch_kind<true, false, 5>(kind, lds, grid, s, a, rgb, canvas); /* without mirrors only */
    else if (stage == 4) launch_kind<true, REFLECT, 4>(kind, lds, grid, s, a, rgb, canvas);
    else launch_kind<true, REFLECT, 0>(kind, 0, grid, s, a, rgb, canvas);
  } else if (kind == 1) {
    launch_one<false, false, REFLECT, 0, 1>(0, grid, s, a, rgb, canvas);
  } else if (stage == 1) {
    launch_kind<false, REFLECT, 1>(kind, lds, grid, s, a, rgb, canvas);
  } else {
    launch_kind<false, REFLECT, 0>(kind, lds, grid, s, a, rgb, canvas);
  }
}
void launch(bool count, bool bvh, bool reflect, dim3 grid, hipStream_t s, const RenderArgs& a_in, unsigned char* rgb,
            double* canvas) {
  RenderArgs a = a_in;

  const size_t per_form = RT_LIST_LDS_BYTES, per_sform = RT_SLIST_LDS_BYTES;
  const size_t list_bytes = (size_t)a.list_lds * per_form + (size_t)a.slist_lds * per_sform;
  /* what fits behind `used` bytes of a workgroup's LDS budget `cap`: the camera rays' list first, then the shadow rays' */
  auto fit_lists = [&](size_t used, size_t cap) {
    a.list_lds = std::min(a.list_lds, (int)((cap - used) / per_form / 4 * 4));
    used += (size_t)a.list_lds * per_form;
    a.slist_lds = std::min(a.slist_lds, (int)((cap - used) / per_sform / 4 * 4));
    return used + (size_t)a.slist_lds * per_sform;
  };
  /* The counting build exists once per family: what it counts (rays, tests) does not depend on where the
   * primitives are staged, on the kinds present or on whether any material reflects. */
  if (count) {
    if (bvh) {
      a.list_lds = 0; /* no LDS in this build: the lists are read from memory */
      a.slist_lds = 0;
      launch_one<true, true, true, 0, 0>(0, grid, s, a, rgb, canvas);
    } else {
      launch_one<true, false, true, 0, 0>(list_bytes, grid, s, a, rgb, canvas);
    }
    return;
  }
  int stage = 0;
  size_t lds = 0;
  if (cur->use_lds) {
    if (!bvh && a.list_forms) {
      stage = 0; /* camera rays read their tile's list (from the workgroup's copy in LDS if it fits), the others the whole
                    array through scalar loads */
      /* three workgroups per CU (the large-mesh kernel runs three waves per SIMD) beside the kernel's own 18 KiB */
      lds = fit_lists(RT_PARK_BYTES, (size_t)160 * 1024 / RT_MIN_WAVES_EXHAUSTIVE_LARGE) - RT_PARK_BYTES;
    } else if (!bvh && cur->scene.n_tris > 0 && cur->scene.n_tris <= kMaxLdsTris) {
      stage = 1;
      lds = (size_t)cur->scene.n_tris * sizeof(DevTriF);
    } else if (bvh && cur->lds_stack && cur->scene.bvh_depth >= 1) {
      /* the traversal stacks, and in front of them the inner nodes if both fit while two workgroups still share
       * a CU's 160 KiB (the walk's register budget allows no more than two anyway) */
      const size_t stacks = (size_t)RT_TILE_W * RT_TILE_H * sizeof(int2) * (size_t)cur->scene.bvh_depth;
      const size_t nodes = (size_t)cur->scene.n_bvh_nodes * sizeof(DevBvhNode);
      /* a tree small enough for LDS has references of 16 bits and stack entries of 4 bytes (bvh_walk, LSTACK == 2) */
      const size_t stacks4 = stacks / 2;
      if (cur->lds_nodes && cur->scene.n_bvh_nodes <= kMaxLdsBvhNodes && cur->scene.n_tris < 4096 && nodes + stacks4 <= kMaxLdsPerWorkgroup) {
        stage = 3;
        lds = nodes + stacks4;
        /* the plain walk runs three workgroups per CU if its LDS lets it (longer lists are read from memory); with
         * mirrors the registers allow two */
        const bool many = !reflect && a.n_heavy > cur->many_per_cu * cur->n_cu && nodes + stacks4 <= (size_t)160 * 1024 / 3 - 5120;
        if (many) stage = 5; /* the same kernel held to the registers of three waves per SIMD */
        lds = fit_lists(lds, many ? (size_t)160 * 1024 / 3 - 1024 : kMaxLdsPerWorkgroup); /* 1 KiB of static LDS in every walk */
      } else if (stacks <= (reflect ? kMaxLdsPerWorkgroup : (size_t)160 * 1024 / 3 - 1024)) {
        /* with the nodes in global memory the kernels that need few registers run three workgroups per CU:
         * stacks that would cut that to two cost more than they save (1.31 M triangles, depth 27: +6 %); the
         * kernels with reflections run two per CU in any case (same mesh, reflective: -10 % with the stacks) */
        stage = 4;
        lds = stacks;
        lds = fit_lists(lds, reflect ? kMaxLdsPerWorkgroup : (size_t)160 * 1024 / 3 - 1024);
      }
    }
  }
  if ((bvh && stage == 0) || !cur->use_lds) {
    a.list_lds = 0;
    a.slist_lds = 0;
  }
  const int kind = (!bvh && cur->scene.n_tris == 0) ? 1 : (cur->scene.n_spheres == 0 ? 2 : 0);
  cur->launch_lds = lds;
  cur->launch_waves = !bvh ? 0 : (reflect || stage == 3 ? RT_MIN_WAVES_BVH : (stage == 5 ? RT_MIN_WAVES_BVH_LDS : RT_MIN_WAVES_BVH_PLAIN));
  if (reflect) launch_family<true>(bvh, stage, kind, lds, grid, s, a, rgb, canvas);
  else launch_family<false>(bvh, stage, kind, lds, grid, s, a, rgb, canvas);
}

#include "rt_plan.hip.inc"

#include "rt_general_api.hip.inc"

int check_params(const rt_render_params* p) {
  if (!p) return RT_ERR_INVALID;
  if (!cur->ready || !cur->have_scene) return RT_ERR_NOT_READY;
  if (p->max_depth < 0 || p->max_depth > RT_MAX_DEPTH) return RT_ERR_TOO_LARGE;
  if (p->spp_grid < 1) return RT_ERR_INVALID;
  if (p->spp_grid > RT_MAX_SPP_GRID) return RT_ERR_TOO_LARGE;
  if (p->n_ranks < 1 || p->rank < 0 || p->rank >= p->n_ranks) return RT_ERR_INVALID;
  if (p->n_ranks > 1 && p->band_rows < 1) return RT_ERR_INVALID;
  if (p->accel != RT_ACCEL_AUTO && p->accel != RT_ACCEL_BRUTE && p->accel != RT_ACCEL_BVH && p->accel != RT_ACCEL_REFERENCE)
    return RT_ERR_INVALID;
  if (p->accel == RT_ACCEL_BVH && cur->scene.n_bvh_nodes == 0 && cur->scene.n_tris > 0 && !cur->bvh_single_leaf)
    return RT_ERR_NOT_READY;
  return RT_OK;
}

/* Enqueue one render of the current context's scene on `s`.  timed: record the context's event pair around the
 * kernel; count: the counting build, counters zeroed first.
 * may_plan: a tile order that does not exist yet for these parameters may be made (allocates, copies, waits for
 * the copy); may_tune: a plan whose split has not been measured yet may be refined now (refine_split: dozens of
 * timed launches into d_rgb, milliseconds).  With both false the call only enqueues: no allocation, no event is
 * created, nothing is waited for, so it is legal on a stream that is being captured. */
int render_enqueue(const rt_render_params* p, unsigned char* d_rgb, double* d_canvas, hipStream_t s, bool timed, bool count,
                   bool may_plan, bool may_tune, Enqueued* out) {
  const int W = cur->scene.hsize, H = cur->scene.vsize;
  const int rows = rt_hip_rank_rows(H, p->band_rows, p->n_ranks, p->rank);
  RenderArgs a;
  a.probe = 0;
  a.max_depth = p->max_depth;
  a.spp_grid = p->spp_grid;
  a.band_rows = p->n_ranks > 1 ? p->band_rows : 1;
  a.n_ranks = p->n_ranks;
  a.rank = p->rank;
  a.rows = rows;
  a.ablate = (p->flags >> 8) & 0xff; /* development builds only (RT_ABLATE) */
  a.count_as_run = (p->flags & RT_RENDER_COUNT_AS_RUN) ? 1 : 0;
  a.coop_min_lg = cur->coop_min_lg;
  for (int i = 0; i < RT_MAX_SPP_GRID; ++i) a.sample_off[i] = (i + 0.5) / a.spp_grid; /* ref_harness / oracle: (i + 0.5) / G */
  const bool general = cur->general || p->accel == RT_ACCEL_REFERENCE;
  if (general && !cur->d_gen_scene) return RT_ERR_TOO_LARGE; /* too many objects for the every-ray-every-object kernel */
  bool bvh = cur->scene.n_bvh_nodes > 0 && cur->scene.n_tris > 0;
  if (p->accel == RT_ACCEL_BRUTE) bvh = false;
  if (p->accel == RT_ACCEL_AUTO && cur->scene.n_tris < 32) bvh = false; /* a handful of triangles: the walk costs more */
  const bool reflect = cur->reflective && p->max_depth >= 1;
  const bool overlap = (p->flags & RT_RENDER_FRAMES_OVERLAP) != 0;
  if (rows > 0) {
    if (count) HIP_TRY(hipMemsetAsync(cur->d_counters, 0, 10 * sizeof(unsigned long long), s));
    const long long n_tiles = (long long)((W + RT_TILE_W - 1) / RT_TILE_W) * ((rows + RT_TILE_H - 1) / RT_TILE_H);
    int n_blocks = 0;
    if (general) {
      a.n_heavy = (int)n_tiles;
      a.light_per_block = 1;
          a.ablate = 0; /* the timing cuts are the fast kernels'; this kernel reads the field as "probe launch" */
      a.n_entries = 0;
      a.light_is_background = 0;
      a.list_forms = 0;
      a.list_total = 0;
      a.list_lds = 0;
      a.slist_lds = 0;
  a.slist_lds = 0;
      n_blocks = (int)n_tiles;
      bvh = false;
      const int prc = plan_general_tiles(p, rows, may_plan, s, a);
      if (prc != RT_OK) return prc;
    } else {
      const int prc = plan_tiles(p, rows, bvh, reflect, overlap, may_plan, s, a, n_blocks);
      if (prc != RT_OK) return prc;
      if (bvh && !count && may_tune && !(p->flags & RT_RENDER_NO_TUNE) && cur->launch_plan && cur->launch_plan->refine_left > 0) {
        const int rrc = refine_split(cur->launch_plan, bvh, reflect, s, a, d_rgb, d_canvas, n_blocks);
        if (rrc != RT_OK) return rrc;
      }
    }
    dim3 grid((unsigned)n_blocks);
    if (timed) HIP_TRY(hipEventRecord(cur->ev0, s));
    if (general) launch_general(count, grid, s, a, d_rgb, d_canvas);
    else launch(count, bvh, reflect, grid, s, a, d_rgb, d_canvas);
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(cur->ev1, s));
  }
  out->rows = rows;
  out->accel_used = general ? RT_ACCEL_REFERENCE : (bvh ? RT_ACCEL_BVH : RT_ACCEL_BRUTE);
  out->timed = timed;
  out->count = count;
  return RT_OK;
}

/* Wait for what render_enqueue launched on the current context and fill `stats`. */
int render_collect(const Enqueued& e, hipStream_t s, rt_stats* stats) {
  memset(stats, 0, sizeof *stats);
  stats->rows_rendered = e.rows;
  stats->accel_used = e.accel_used;
  if (e.rows <= 0) return RT_OK;
  if (e.timed) {
    HIP_TRY(hipEventSynchronize(cur->ev1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, cur->ev0, cur->ev1));
    stats->kernel_ms = ms;
  }
  if (e.count) {
    unsigned long long c[10];
    HIP_TRY(hipMemcpyAsync(c, cur->d_counters, sizeof c, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stats->rays_primary = c[0];
    stats->rays_shadow = c[1];
    stats->rays_secondary = c[2];
    stats->prim_tests = c[3];
    stats->node_tests = c[4];
    stats->sphere_tests = c[5];
    stats->tri_exact = c[6];
    stats->sphere_exact = c[7];
    stats->zero_t_entries = c[8];
    stats->list_cap_rays = c[9];
  }
  return RT_OK;
}

/* enqueue one render on `s`; waits and fills stats only when stats != NULL.  synchronous: the entry point waits
 * for the frame anyway, so it may plan and tune first */
int render_common(const rt_render_params* p, unsigned char* d_rgb, double* d_canvas, hipStream_t s, rt_stats* stats, bool synchronous) {
  Enqueued e;
  const int rc = render_enqueue(p, d_rgb, d_canvas, s, stats != nullptr, p->count_rays != 0 && stats != nullptr, synchronous, synchronous, &e);
  if (rc != RT_OK || !stats) return rc;
  return render_collect(e, s, stats);
}

/* The first launch of a kernel pays for loading the library's code object onto the device and for the scratch
 * memory its waves may need (milliseconds, against a frame of tens of microseconds).  A caller that renders one
 * frame per process (the reference's main(): one render(), one canvas.ppm) would see that inside its render call,
 * so the upload ends with one launch of the kernel the default parameters select, over an image of no rows (a
 * single workgroup that finds no tile to render), after allocating the frame's output buffer and planning its
 * tile order. */
int warm_launch() {
  rt_render_params p = {5, 1, RT_ACCEL_AUTO, 0, 1, 1, 0, 0};
  RenderArgs a;
  memset(&a, 0, sizeof a);
  a.max_depth = p.max_depth;
  a.spp_grid = 1;
  a.band_rows = 1;
  a.n_ranks = 1;
  a.sample_off[0] = 0.5;
  const int H = cur->scene.vsize;
  /* the output image of rt_hip_render and the tile order of a whole-frame launch are made now as well */
  const size_t bytes = (size_t)H * cur->scene.hsize * 3;
  if (bytes > cur->d_rgb_bytes) {
    if (cur->d_rgb) (void)hipFree(cur->d_rgb);
    cur->d_rgb = nullptr;
    cur->d_rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_rgb, bytes));
    cur->d_rgb_bytes = bytes;
  }
  const bool bvh = cur->scene.n_bvh_nodes > 0 && cur->scene.n_tris >= 32;
  int n_blocks = 0;
  a.rows = H;
  const int prc = cur->general ? plan_general_tiles(&p, H, true, cur->stream, a) : plan_tiles(&p, H, bvh, cur->reflective, false, true, cur->stream, a, n_blocks);
  if (prc != RT_OK) return prc;
  a.rows = 0; /* ... but the launch itself renders nothing */
  a.n_heavy = 0;
  a.light_per_block = 1;
  a.n_entries = 0;
  cur->launch_tile_list = nullptr;
  cur->launch_tile_mask = nullptr;
  if (cur->general) launch_general(false, dim3(1), cur->stream, a, cur->d_rgb, nullptr);
  else launch(false, bvh, cur->reflective, dim3(1), cur->stream, a, cur->d_rgb, nullptr);
  HIP_TRY(hipGetLastError());
  /* the first device-to-host copy of some size sets up the runtime's staging buffers (5 ms, measured): done here
   * with the first 256 KiB of the (still unwritten) frame */
  const size_t probe = bytes < (256u << 10) ? bytes : (256u << 10);
  std::vector<unsigned char> sink(probe ? probe : 1);
  if (probe) HIP_TRY(hipMemcpyAsync(sink.data(), cur->d_rgb, probe, hipMemcpyDeviceToHost, cur->stream));
  HIP_TRY(hipStreamSynchronize(cur->stream));
  return RT_OK;
}

void multi_teardown(); /* rt_multi.hip.inc */

/* stream, events, counters and tuning knobs of the context `cur` points at; the caller has made `device` current */
int create_context(int device) {
  HIP_TRY(hipStreamCreateWithFlags(&cur->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&cur->ev0));
  HIP_TRY(hipEventCreate(&cur->ev1));
  HIP_TRY(hipEventCreate(&cur->tm0));
  HIP_TRY(hipEventCreate(&cur->tm1));
  HIP_TRY(hipEventCreateWithFlags(&cur->ppm_ev, hipEventDisableTiming));
  HIP_TRY(hipMalloc((void**)&cur->d_counters, (16 + (RT_STAMPS ? RT_STAMP_WORDS * (size_t)RT_STAMPS_MAX_WAVES : 0)) * sizeof(unsigned long long)));
  HIP_TRY(hipGetSymbolAddress((void**)&cur->ppm_scratch_ptr, HIP_SYMBOL(ppm_scratch)));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  cur->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  /* A/B switches of the development build (-DRT_DEV_SWITCHES: lib/librt_hip_dev.so, which the tests of the planner's
   * variants load); the shipping library reads no environment variable: a caller states what it wants through
   * rt_render_params */
#ifdef RT_DEV_SWITCHES
  if (const char* e = getenv("RT_HIP_CROWDED_FIRST")) cur->crowded_first = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_SPLIT_TILES")) cur->split_tiles = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_QUARTERS")) cur->split_quarters = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_MAX_PER_CU")) cur->split_max_per_cu = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_QUARTERS_REFLECT")) cur->split_quarters_reflect = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_PARTS")) cur->split_parts = atoi(e) == 4 ? 4 : (atoi(e) == 2 ? 2 : 1);
  if (const char* e = getenv("RT_HIP_SPLIT_MIN_COST")) cur->split_min_cost = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_REFINE")) cur->split_refine = atoi(e);
  if (const char* e = getenv("RT_HIP_SPLIT_RESORT")) cur->split_resort = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_WALK_SHADOW_MAX")) cur->walk_shadow_max = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_PART_MIN_LG")) cur->part_min_lg = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_SPLIT_NARROW")) cur->split_narrow = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_COOP")) cur->coop_min_lg = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_COOP_CAM")) cur->coop_cam_per_lane = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_COOP_SHADOW")) cur->coop_shadow_per_lane = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_PART_LISTS")) cur->part_lists = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_PART_SHADOW_MAX")) cur->part_shadow_max = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_SPLIT_THRESHOLD")) cur->split_threshold = std::fmax(0.1, std::fmin(0.99, atof(e)));
  if (const char* e = getenv("RT_HIP_SPLIT_MAX_LG")) cur->split_max_lg = std::max(0, std::min(6, atoi(e)));
  if (const char* e = getenv("RT_HIP_XCD_TILES")) cur->xcd_tiles = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("RT_HIP_LDS_STACK")) cur->lds_stack = atoi(e) != 0; /* 0 = traversal stack in scratch */
  if (const char* e = getenv("RT_HIP_LDS_NODES")) cur->lds_nodes = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_MANY_PER_CU")) cur->many_per_cu = atoi(e);
  if (const char* e = getenv("RT_HIP_LDS")) cur->use_lds = atoi(e) != 0; /* 0 = read primitives from global memory */
  if (const char* e = getenv("RT_HIP_LIGHT_PER_BLOCK")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) cur->light_per_block = v;
  }
  if (const char* e = getenv("RT_HIP_COVER_TILES")) cur->cover_tiles = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_TILE_MASKS")) cur->tile_masks = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_TILE_LISTS")) cur->tile_lists = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_LIST_MAX_WALK")) cur->list_max_walk = cur->list_max_walk_few = atoi(e);
  if (const char* e = getenv("RT_HIP_LIST_LDS")) cur->list_lds = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_SHADOW_LISTS")) cur->shadow_lists = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_SHADOW_MASKS")) cur->shadow_masks = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_SPHERE_MASKS")) cur->sphere_masks = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_TILE_PLAN")) cur->tile_plan = atoi(e) != 0; /* 0 = row-major tiles */
#endif
  cur->device = device;
  cur->ready = true;
  return RT_OK;
}

/* release whatever `c` holds (also a half-made context) and reset it; makes c's device current */
void destroy_context(Context* c) {
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  Context* const keep = cur;
  cur = c;
  free_scene();
  cur = keep;
  if (c->d_counters) (void)hipFree(c->d_counters);
  if (c->d_scene) (void)hipFree(c->d_scene);
  for (Context::TilePlan& t : c->plans)
    if (t.d_list) (void)hipFree(t.d_list);
  if (c->d_rgb) (void)hipFree(c->d_rgb);
  if (c->d_canvas) (void)hipFree(c->d_canvas);
  if (c->d_ppm_len) (void)hipFree(c->d_ppm_len);
  if (c->d_ppm_group) (void)hipFree(c->d_ppm_group);
  if (c->d_ppm_text) (void)hipFree(c->d_ppm_text);
  if (c->h_ppm_total) (void)hipHostFree(c->h_ppm_total);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->tm0) (void)hipEventDestroy(c->tm0);
  if (c->tm1) (void)hipEventDestroy(c->tm1);
  if (c->ppm_ev) (void)hipEventDestroy(c->ppm_ev);
  for (Context::PpmSlot& q : c->ppm_slot) {
    if (q.d_rgb) (void)hipFree(q.d_rgb);
    if (q.d_text) (void)hipFree(q.d_text);
    if (q.h_text) (void)hipHostFree(q.h_text);
    if (q.h_len) (void)hipHostFree(q.h_len);
    if (q.formatted) (void)hipEventDestroy(q.formatted);
    if (q.k0) (void)hipEventDestroy(q.k0);
    if (q.k1) (void)hipEventDestroy(q.k1);
  }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  *c = Context{};
}

}  // namespace

extern "C" {

int rt_hip_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    snprintf(g_err, sizeof g_err, "no HIP device");
    return RT_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n || device >= kMaxDevices) return RT_ERR_NO_DEVICE;
  HIP_TRY(hipSetDevice(device));
  cur = &g_ctx[device];
  if (cur->ready) return RT_OK;
  const int rc = create_context(device);
  if (rc != RT_OK) destroy_context(cur);
  return rc;
}

int rt_hip_set_device(int device) {
  if (device < 0 || device >= kMaxDevices || !g_ctx[device].ready) return RT_ERR_NOT_READY;
  HIP_TRY(hipSetDevice(device));
  cur = &g_ctx[device];
  return RT_OK;
}

int rt_hip_upload_scene(const rt_scene* sc) {
  if (!sc) return RT_ERR_INVALID;
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (sc->camera.hsize < 1 || sc->camera.vsize < 1 || sc->n_spheres < 0 || sc->n_tris < 0 || sc->n_materials < 0)
    return RT_ERR_INVALID;
  /* an empty world is legal (every pixel black); anything else needs its materials */
  if (sc->n_spheres + sc->n_tris > 0 && (sc->n_materials < 1 || !sc->materials)) return RT_ERR_INVALID;
  if ((sc->n_spheres && (!sc->sph_inv || !sc->sph_material || !sc->sph_order)) ||
      (sc->n_tris && (!sc->tri_soa || !sc->tri_material || !sc->tri_order)))
    return RT_ERR_INVALID;
  if ((long long)sc->camera.hsize * sc->camera.vsize > (1ll << 31) / 3) return RT_ERR_TOO_LARGE;
  if (sc->n_shapes != 0 || sc->materials_ext) return g_host_only ? (int)RT_ERR_UNSUPPORTED : upload_general(sc, true);
  const int ns = sc->n_spheres, nt = sc->n_tris;
  const int n_prims = ns + nt;

  /* validate: material indices, list positions form a permutation, affine inverses */
  {
    const double* v = sc->camera.inv_view; /* the kernels read rows 0-2: rays leave the camera as (point, vector) */
    if (v[12] != 0.0 || v[13] != 0.0 || v[14] != 0.0 || v[15] != 1.0) return RT_ERR_UNSUPPORTED;
  }
  std::vector<char> seen((size_t)(n_prims > 0 ? n_prims : 1), 0);
  for (int i = 0; i < ns; ++i) {
    const double* m = sc->sph_inv + 16 * (size_t)i;
    if (m[12] != 0.0 || m[13] != 0.0 || m[14] != 0.0 || m[15] != 1.0) return RT_ERR_UNSUPPORTED;
    if (sc->sph_material[i] < 0 || sc->sph_material[i] >= sc->n_materials) return RT_ERR_INVALID;
    const int k = sc->sph_order[i];
    if (k < 0 || k >= n_prims || seen[k]) return RT_ERR_INVALID;
    seen[k] = 1;
  }
  for (int i = 0; i < nt; ++i) {
    if (sc->tri_material[i] < 0 || sc->tri_material[i] >= sc->n_materials) return RT_ERR_INVALID;
    const int k = sc->tri_order[i];
    if (k < 0 || k >= n_prims || seen[k]) return RT_ERR_INVALID;
    seen[k] = 1;
  }
  bool use_bvh = sc->n_bvh_nodes > 0 && sc->bvh_nodes && sc->bvh_tri && nt > 0;
  int depth = 0;
  int n_bvh_inner = 0;
  cur->bvh_single_leaf = use_bvh && sc->bvh_nodes[0].right < 0;
  if (cur->bvh_single_leaf) use_bvh = false; /* nothing to walk: the exhaustive loop is the walk */
  if (use_bvh) {
    depth = bvh_depth(sc->bvh_nodes, sc->n_bvh_nodes, 0, 0);
    if (depth > 48) return RT_ERR_TOO_LARGE;
    if (nt >= (1 << 27)) return RT_ERR_TOO_LARGE;
    std::vector<char> used((size_t)nt, 0);
    for (int i = 0; i < sc->n_bvh_nodes; ++i) {
      const rt_bvh_node& n = sc->bvh_nodes[i];
      if (n.right < 0) {
        if (n.left < 0 || (long long)n.left - n.right > nt) return RT_ERR_INVALID;
        if (-n.right > 7) return RT_ERR_TOO_LARGE; /* the device leaf encoding holds up to 7 triangles */
      } else if (n.left <= 0 || n.left >= sc->n_bvh_nodes || n.right <= 0 || n.right >= sc->n_bvh_nodes) {
        return RT_ERR_INVALID;
      }
    }
    for (int i = 0; i < nt; ++i) {
      const int t = sc->bvh_tri[i];
      if (t < 0 || t >= nt || used[t]) return RT_ERR_INVALID;
      used[t] = 1;
    }
  }

  free_scene();
  /* Device order of the spheres: smallest first (by the extent of the forward transform).  hit() is decided by
   * (t, list position), whatever order the spheres are looked at in (every sphere carries its position), and a
   * small sphere that a ray hits usually lies in front of the large ones (floors, walls): once it has been solved,
   * the single-precision test rules those out as lying beyond the hit. */
  std::vector<int> sph_perm((size_t)ns);
  {
    std::vector<double> extent((size_t)ns, INFINITY);
    for (int i = 0; i < ns; ++i) {
      sph_perm[i] = i;
      double fwd[16];
      if (invert4(sc->sph_inv + 16 * (size_t)i, fwd)) {
        double e2 = 0.0;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) e2 += fwd[4 * r + c] * fwd[4 * r + c];
        if (std::isfinite(e2)) extent[i] = e2;
      }
    }
    std::stable_sort(sph_perm.begin(), sph_perm.end(), [&](int x, int y) { return extent[x] < extent[y]; });
  }
  cur->sph_upload_index.assign(sph_perm.begin(), sph_perm.end());
  std::vector<DevSphere> hs((size_t)(ns > 0 ? ns : 1));
  for (int k = 0; k < ns; ++k) {
    const int i = sph_perm[k];
    memcpy(hs[k].inv, sc->sph_inv + 16 * (size_t)i, 12 * sizeof(double));
    hs[k].material = sc->sph_material[i];
    hs[k].order = sc->sph_order[i];
  }
  std::vector<DevTri> ht((size_t)(nt > 0 ? nt : 1));
  std::vector<double> hn((size_t)(nt > 0 ? nt : 1) * 3);
  for (int k = 0; k < nt; ++k) {
    const int i = use_bvh ? sc->bvh_tri[k] : k; /* device order = BVH leaf order */
    for (int c = 0; c < 3; ++c) {
      ht[k].p1[c] = sc->tri_soa[(size_t)(0 + c) * nt + i];
      ht[k].e1[c] = sc->tri_soa[(size_t)(3 + c) * nt + i];
      ht[k].e2[c] = sc->tri_soa[(size_t)(6 + c) * nt + i];
      hn[3 * (size_t)k + c] = sc->tri_soa[(size_t)(9 + c) * nt + i];
    }
    ht[k].material = sc->tri_material[i];
    ht[k].order = sc->tri_order[i];
  }
  /* camera origin = inv_view * point(0,0,0), mat4x4_mul_tuple order (RayTracer.c:419, 1563) */
  double org[3];
  for (int r = 0; r < 3; ++r) {
    const double* m = sc->camera.inv_view + 4 * r;
    org[r] = ((0.0 * m[0] + 0.0 * m[1]) + 0.0 * m[2]) + 1.0 * m[3];
  }
  /* origin-only terms of the two intersection routines, in the reference's operation order
   * (host code is built without FMA contraction, like the reference binary) */
  std::vector<DevTriPrimary> htp((size_t)(nt > 0 ? nt : 1));
  double amax = 1.0, blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
  for (int k = 0; k < nt; ++k) {
    const DevTri& t = ht[k];
    double s[3], q[3];
    for (int c = 0; c < 3; ++c) s[c] = org[c] - t.p1[c];                 /* tuple_sub, :951 */
    q[0] = s[1] * t.e1[2] - s[2] * t.e1[1];                               /* tuple_cross(s, e1), :957 */
    q[1] = s[2] * t.e1[0] - s[0] * t.e1[2];
    q[2] = s[0] * t.e1[1] - s[1] * t.e1[0];
    for (int c = 0; c < 3; ++c) { htp[k].s[c] = s[c]; htp[k].q[c] = q[c]; }
    htp[k].tq = (t.e2[0] * q[0] + t.e2[1] * q[1]) + t.e2[2] * q[2];       /* tuple_dot(e2, q), :963 */
    htp[k].pad = 0.0;
    for (int c = 0; c < 3; ++c) {
      const double v[3] = {t.p1[c], t.p1[c] + t.e1[c], t.p1[c] + t.e2[c]};
      for (double x : v) {
        if (x < blo[c]) blo[c] = x;
        if (x > bhi[c]) bhi[c] = x;
        if (std::fabs(x) > amax) amax = std::fabs(x);
      }
    }
  }
  for (int c = 0; c < 3; ++c) if (std::fabs(org[c]) > amax) amax = std::fabs(org[c]);
  amax = std::fmax(amax, std::fmax(std::fabs(sc->light.position[0]),
                                   std::fmax(std::fabs(sc->light.position[1]), std::fabs(sc->light.position[2]))));
  /* rays that are walked through the tree also start on spheres (mirror and shadow rays): the single-precision
   * box tests are padded relative to the largest coordinate a ray origin can have */
  for (int i = 0; i < ns; ++i) {
    double fwd[16];
    if (!invert4(sc->sph_inv + 16 * (size_t)i, fwd)) continue;
    for (int c = 0; c < 3; ++c) {
      const double reach = std::fabs(fwd[4 * c + 3]) +
                           std::sqrt(fwd[4 * c] * fwd[4 * c] + fwd[4 * c + 1] * fwd[4 * c + 1] + fwd[4 * c + 2] * fwd[4 * c + 2]);
      if (std::isfinite(reach) && reach > amax) amax = reach;
    }
  }
  /* The single-precision first tests need a radius that holds every vertex and every ray origin (camera, points on
   * triangles and spheres): sqrt(3) x the largest coordinate.  A sphere whose extent is unknown leaves none. */
  bool radius_known = true;
  for (int i = 0; i < ns; ++i) {
    double fwd[16];
    if (!invert4(sc->sph_inv + 16 * (size_t)i, fwd)) radius_known = false;
  }
  const double radius = radius_known ? 1.7320508075688772 * amax * 1.0001 + 1e-5 : INFINITY;
  cur->filter_radius = radius;
  std::vector<DevTriF> htf((size_t)(nt > 0 ? nt : 1));
  for (int k = 0; k < nt; ++k) rt_make_trif(ht[k], radius, &htf[k]);
  std::vector<DevSphereF> hsf((size_t)(ns > 0 ? ns : 1));
  for (int i = 0; i < ns; ++i) make_spheref(hs[i].inv, org, radius, &hsf[i]);
  std::vector<DevSpherePrimary> hsp((size_t)(ns > 0 ? ns : 1));
  for (int i = 0; i < ns; ++i) {
    const double* m = hs[i].inv;
    double o[3];
    for (int r = 0; r < 3; ++r) o[r] = ((org[0] * m[4 * r] + org[1] * m[4 * r + 1]) + org[2] * m[4 * r + 2]) + m[4 * r + 3];
    for (int c = 0; c < 3; ++c) hsp[i].o[c] = o[c];
    hsp[i].c = ((o[0] * o[0] + o[1] * o[1]) + o[2] * o[2]) - 1.0;          /* tuple_dot - 1.0, :927 */
  }
  int rc;
  if ((rc = upload(&cur->d_tri_primary, htp.data(), sizeof(DevTriPrimary) * (size_t)nt)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_sph_primary, hsp.data(), sizeof(DevSpherePrimary) * (size_t)ns)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_spheres, hs.data(), sizeof(DevSphere) * (size_t)ns)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_tris, ht.data(), sizeof(DevTri) * (size_t)nt)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_trif, htf.data(), sizeof(DevTriF) * (size_t)nt)) != RT_OK) return rc;
  cur->host_trif.clear();
  cur->host_tri_k.clear();
  cur->host_tri_order.clear();
  cur->mesh_world.clear();
  if (ns == 0 && nt > 0 && nt <= kMaxListTris) {
    cur->host_trif.assign(htf.begin(), htf.begin() + nt);
    cur->host_tri_k.resize((size_t)nt);
    cur->host_tri_order.resize((size_t)nt);
    for (int k = 0; k < nt; ++k) cur->host_tri_order[k] = ht[k].order;
    cur->mesh_world.resize(9 * (size_t)nt);
    for (int k = 0; k < nt; ++k)
      for (int c = 0; c < 3; ++c) {
        cur->mesh_world[9 * (size_t)k + c] = ht[k].p1[c];
        cur->mesh_world[9 * (size_t)k + 3 + c] = ht[k].p1[c] + ht[k].e1[c];
        cur->mesh_world[9 * (size_t)k + 6 + c] = ht[k].p1[c] + ht[k].e2[c];
      }
    for (int k = 0; k < nt; ++k) {
      const double* p1 = ht[k].p1;
      const double* e1 = ht[k].e1;
      const double* e2 = ht[k].e2;
      const double n[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
      cur->host_tri_k[k] = (float)((p1[0] - org[0]) * n[0] + (p1[1] - org[1]) * n[1] + (p1[2] - org[2]) * n[2]);
    }
  }
  if ((rc = upload(&cur->d_sphf, hsf.data(), sizeof(DevSphereF) * (size_t)ns)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_normals, hn.data(), sizeof(double) * 3 * (size_t)nt)) != RT_OK) return rc;
  if ((rc = upload(&cur->d_materials, sc->materials, sizeof(rt_material) * (size_t)sc->n_materials)) != RT_OK) return rc;
  /* device BVH: inner nodes only, each with both children's boxes.  The host boxes get an extra pad of
   * ~500 float ulps of the largest coordinate in play (mesh, camera, light), far more than rounding the ray
   * origin, its reciprocal direction and the slab products to single precision can move a slab distance. */
#if RT_BVH_WIDE
  std::vector<DevBvh4> hb;
  if (use_bvh) {
    WideBuild wb = {sc->bvh_nodes, 3e-5 * amax, &hb};
    hb.reserve((size_t)sc->n_bvh_nodes / 3 + 1);
    int need = 0;
    (void)wide_node(wb, 0, need);
    if (need > RT_BVH_STACK) return RT_ERR_TOO_LARGE;
    depth = need; /* what the walk's stack has to hold */
    n_bvh_inner = (int)hb.size();
    if ((rc = upload(&cur->d_bvh, hb.data(), sizeof(DevBvh4) * hb.size())) != RT_OK) return rc;
  }
#else
  std::vector<DevBvh2> hb;
  if (use_bvh) {
    const double pad2 = 3e-5 * amax;
    std::vector<int> inner_index((size_t)sc->n_bvh_nodes, -1);
    int n_inner = 0;
    for (int i = 0; i < sc->n_bvh_nodes; ++i)
      if (sc->bvh_nodes[i].right >= 0) inner_index[i] = n_inner++;
    hb.resize((size_t)(n_inner > 0 ? n_inner : 1));
    auto child_ref = [&](int c) -> int32_t {
      const rt_bvh_node& n = sc->bvh_nodes[c];
      if (n.right >= 0) return inner_index[c];
      return ~((n.left << 3) | (-n.right));
    };
    for (int i = 0; i < sc->n_bvh_nodes; ++i) {
      const rt_bvh_node& n = sc->bvh_nodes[i];
      if (n.right < 0) continue;
      DevBvh2& d2 = hb[(size_t)inner_index[i]];
      const int ch[2] = {n.left, n.right};
      float lo[2][3], hi[2][3];
      for (int k = 0; k < 2; ++k)
        for (int c = 0; c < 3; ++c) {
          lo[k][c] = std::nextafterf((float)((double)sc->bvh_nodes[ch[k]].lo[c] - pad2), -INFINITY);
          hi[k][c] = std::nextafterf((float)((double)sc->bvh_nodes[ch[k]].hi[c] + pad2), INFINITY);
        }
      int32_t r0 = child_ref(ch[0]), r1 = child_ref(ch[1]);
      float f0, f1;
      memcpy(&f0, &r0, 4);
      memcpy(&f1, &r1, 4);
      d2.q[0] = make_float4(lo[0][0], lo[0][1], lo[0][2], f0);
      d2.q[1] = make_float4(hi[0][0], hi[0][1], hi[0][2], f1);
      d2.q[2] = make_float4(lo[1][0], lo[1][1], lo[1][2], 0.f);
      d2.q[3] = make_float4(hi[1][0], hi[1][1], hi[1][2], 0.f);
    }
    n_bvh_inner = n_inner;
    if ((rc = upload(&cur->d_bvh, hb.data(), sizeof(DevBvh2) * (size_t)n_inner)) != RT_OK) return rc;
  }
#endif

  DevScene& d = cur->scene;
  d.pixel_size = sc->camera.pixel_size;
  d.half_width = sc->camera.half_width;
  d.half_height = sc->camera.half_height;
  memcpy(d.inv_view, sc->camera.inv_view, 12 * sizeof(double));
  for (int r = 0; r < 3; ++r) d.origin[r] = org[r];
  /* bounds of all triangles for the single-precision early-out: padded by ~500 float ulps of the largest
   * coordinate in play, far more than single-precision rounding of origin, direction and slab products */
  const double pad = 3e-5 * amax;
  for (int c = 0; c < 3; ++c) {
    d.box_lo[c] = nt ? std::nextafterf((float)(blo[c] - pad), -INFINITY) : 0.f;
    d.box_hi[c] = nt ? std::nextafterf((float)(bhi[c] + pad), INFINITY) : 0.f;
  }
  for (int c = 0; c < 3; ++c) { cur->mesh_lo[c] = blo[c] - pad; cur->mesh_hi[c] = bhi[c] + pad; }
  /* where each triangle lands on the image: the tile order of a launch puts crowded tiles first */
  cur->tri_rect.clear();
  cur->tri_xy.clear();
  cur->tri_depth.clear();
  double view[16];
  if (nt > 0 && nt <= kMaxTilePlanTris && invert4(sc->camera.inv_view, view)) {
    cur->tri_rect.resize(4 * (size_t)nt);
    bool all_in_front = nt <= kMaxCoverTris;
    if (all_in_front) cur->tri_xy.resize(6 * (size_t)nt);
    cur->tri_depth.assign(all_in_front ? (size_t)nt : 0, 3.0e38f);
    const double W = sc->camera.hsize, H = sc->camera.vsize;
    for (int k = 0; k < nt; ++k) {
      const DevTri& t = ht[k];
      float x0 = 1e30f, y0 = 1e30f, x1 = -1e30f, y1 = -1e30f;
      for (int v = 0; v < 3; ++v) {
        double w[3];
        for (int c = 0; c < 3; ++c) w[c] = t.p1[c] + (v == 1 ? t.e1[c] : (v == 2 ? t.e2[c] : 0.0));
        double c3[3];
        for (int r = 0; r < 3; ++r) c3[r] = view[4 * r] * w[0] + view[4 * r + 1] * w[1] + view[4 * r + 2] * w[2] + view[4 * r + 3];
        if (!(c3[2] < -1e-6)) { x0 = 0.f; y0 = 0.f; x1 = (float)W; y1 = (float)H; all_in_front = false; break; } /* beside or behind the camera */
        const double px = (sc->camera.half_width - c3[0] / -c3[2]) / sc->camera.pixel_size;
        const double py = (sc->camera.half_height - c3[1] / -c3[2]) / sc->camera.pixel_size;
        if (all_in_front) {
          cur->tri_xy[6 * (size_t)k + 2 * v] = px;
          cur->tri_xy[6 * (size_t)k + 2 * v + 1] = py;
          cur->tri_depth[k] = std::fmin(cur->tri_depth[k], (float)-c3[2]);
        }
        x0 = std::fmin(x0, (float)px); x1 = std::fmax(x1, (float)px);
        y0 = std::fmin(y0, (float)py); y1 = std::fmax(y1, (float)py);
      }
      float* q = &cur->tri_rect[4 * (size_t)k];
      q[0] = x0 - 1.f; q[1] = y0 - 1.f; q[2] = x1 + 1.f; q[3] = y1 + 1.f;
    }
    if (!all_in_front) { cur->tri_xy.clear(); cur->tri_depth.clear(); }
  }
  cur->tri_world.clear();
  if (nt > 0 && nt <= 64) {
    cur->tri_world.resize(9 * (size_t)nt);
    for (int k = 0; k < nt; ++k)
      for (int c = 0; c < 3; ++c) {
        cur->tri_world[9 * (size_t)k + c] = ht[k].p1[c];
        cur->tri_world[9 * (size_t)k + 3 + c] = ht[k].p1[c] + ht[k].e1[c];
        cur->tri_world[9 * (size_t)k + 6 + c] = ht[k].p1[c] + ht[k].e2[c];
      }
  }
  for (int c = 0; c < 3; ++c) cur->host_light[c] = sc->light.position[c];
  /* the boxes of the spheres, in the world and on the image */
  cur->sph_inv.clear();
  cur->sph_norm.clear();
  cur->sph_world.clear();
  cur->sph_xy.clear();
  double view_s[16];
  if (ns > 0 && ns <= 64 && invert4(sc->camera.inv_view, view_s)) {
    bool ok = true;
    cur->sph_inv.resize(12 * (size_t)ns);
    cur->sph_norm.resize((size_t)ns);
    cur->sph_world.resize(24 * (size_t)ns);
    cur->sph_xy.resize(16 * (size_t)ns);
    for (int i = 0; i < ns && ok; ++i) {
      double inv4[16], fwd[16], n2 = 0.0;
      memcpy(inv4, hs[i].inv, 12 * sizeof(double));
      inv4[12] = inv4[13] = inv4[14] = 0.0;
      inv4[15] = 1.0;
      ok = invert4(inv4, fwd);
      memcpy(&cur->sph_inv[12 * (size_t)i], hs[i].inv, 12 * sizeof(double));
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) n2 += inv4[4 * r + c] * inv4[4 * r + c];
      cur->sph_norm[i] = std::sqrt(n2);
      for (int v = 0; v < 8 && ok; ++v) {
        const double u[3] = {(v & 1) ? 1.0 : -1.0, (v & 2) ? 1.0 : -1.0, (v & 4) ? 1.0 : -1.0};
        double w[3], c3[3];
        for (int r = 0; r < 3; ++r) w[r] = fwd[4 * r] * u[0] + fwd[4 * r + 1] * u[1] + fwd[4 * r + 2] * u[2] + fwd[4 * r + 3];
        for (int r = 0; r < 3; ++r) {
          cur->sph_world[24 * (size_t)i + 3 * v + r] = w[r];
          ok = ok && std::isfinite(w[r]);
          c3[r] = view_s[4 * r] * w[0] + view_s[4 * r + 1] * w[1] + view_s[4 * r + 2] * w[2] + view_s[4 * r + 3];
        }
        const bool in_front = c3[2] < -1e-6; /* NaN: a corner beside or behind the camera, the box has no pixel rectangle */
        cur->sph_xy[16 * (size_t)i + 2 * v] = in_front ? (sc->camera.half_width - c3[0] / -c3[2]) / sc->camera.pixel_size : NAN;
        cur->sph_xy[16 * (size_t)i + 2 * v + 1] = in_front ? (sc->camera.half_height - c3[1] / -c3[2]) / sc->camera.pixel_size : NAN;
      }
    }
    if (!ok) { cur->sph_inv.clear(); cur->sph_norm.clear(); cur->sph_world.clear(); cur->sph_xy.clear(); }
  }
  cur->box_pad = pad;
  cur->scene_serial++;
  memcpy(d.light_pos, sc->light.position, sizeof d.light_pos);
  memcpy(d.light_int, sc->light.intensity, sizeof d.light_int);
  d.hsize = sc->camera.hsize;
  d.vsize = sc->camera.vsize;
  d.n_spheres = ns;
  d.n_tris = nt;
  d.n_bvh_nodes = use_bvh ? n_bvh_inner : 0;
  d.bvh_depth = depth;
  cur->host_camera = sc->camera;
  cur->reflective = false;
  for (int i = 0; i < sc->n_materials; ++i)
    if (!(std::fabs(sc->materials[i].reflective - 0.0) < RT_EPSILON)) cur->reflective = true;
  cur->have_scene = true;
  if (g_host_only) return RT_OK;
  if (!cur->d_scene) HIP_TRY(hipMalloc((void**)&cur->d_scene, sizeof(DevScene)));
  HIP_TRY(hipMemcpy(cur->d_scene, &cur->scene, sizeof(DevScene), hipMemcpyHostToDevice));
  /* small worlds also get the general kernel's form, for RT_ACCEL_REFERENCE */
  if (n_prims <= kMaxReferenceObjects) {
    const int rc = upload_general(sc, false);
    if (rc != RT_OK) return rc;
  }
  /* A mesh this large without a tree would be walked exhaustively by every ray (seconds per frame): build one
   * here.  A tree that turns out too deep for the walk leaves the scene as it is. */
  if (!use_bvh && !cur->bvh_single_leaf && nt >= kAutoDeviceBvhTris) {
    /* Not fatal: a tree too deep for the walk (RT_ERR_TOO_LARGE), or no librt_hip_sort.so beside this library
     * (RT_ERR_UNSUPPORTED; rt_hip_last_error says so): the scene stays, rendered by the exhaustive loop.  Only the
     * explicit rt_hip_build_bvh_device call reports those. */
    const int rc = rt_hip_build_bvh_device(4, nullptr);
    if (rc != RT_OK && rc != RT_ERR_TOO_LARGE && rc != RT_ERR_UNSUPPORTED) return rc;
  }
  return warm_launch();
}

int rt_hip_rank_rows(int vsize, int band_rows, int n_ranks, int rank) {
  if (vsize < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return RT_ERR_INVALID;
  if (n_ranks == 1) return vsize;
  if (band_rows < 1) return RT_ERR_INVALID;
  const int stride = band_rows * n_ranks;
  const int full = vsize / stride, rem = vsize % stride;
  int rows = full * band_rows;
  const int start = rank * band_rows;
  if (rem > start) rows += (rem - start < band_rows) ? rem - start : band_rows;
  return rows;
}

int rt_hip_rank_row_to_y(int i, int band_rows, int n_ranks, int rank) {
  if (i < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return RT_ERR_INVALID;
  if (n_ranks == 1) return i;
  if (band_rows < 1) return RT_ERR_INVALID;
  const int band = i / band_rows;
  return (band * n_ranks + rank) * band_rows + i % band_rows;
}

int rt_hip_render_device(const rt_render_params* p, void* rgb_out_device, void* stream, rt_stats* stats) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  if (!rgb_out_device) return RT_ERR_INVALID;
  hipStream_t s = stream ? (hipStream_t)stream : cur->stream;
  return render_common(p, (unsigned char*)rgb_out_device, nullptr, s, stats, stats != nullptr);
}

int rt_hip_tune(const rt_render_params* p) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  const int rows = rt_hip_rank_rows(cur->scene.vsize, p->band_rows, p->n_ranks, p->rank);
  const size_t bytes = (size_t)rows * cur->scene.hsize * 3;
  if (bytes > cur->d_rgb_bytes) {
    HIP_TRY(hipStreamSynchronize(cur->stream));
    if (cur->d_rgb) (void)hipFree(cur->d_rgb);
    cur->d_rgb = nullptr;
    cur->d_rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_rgb, bytes));
    cur->d_rgb_bytes = bytes;
  }
  rt_render_params q = *p;
  q.count_rays = 0;
  q.flags &= ~(int32_t)RT_RENDER_NO_TUNE;
  Enqueued e;
  /* the measured launches (and one last frame) go into the library's own image */
  rc = render_enqueue(&q, cur->d_rgb, nullptr, cur->stream, false, false, true, true, &e);
  if (rc != RT_OK) return rc;
  HIP_TRY(hipStreamSynchronize(cur->stream));
  return RT_OK;
}

int rt_hip_render(const rt_render_params* p, uint8_t* rgb_out, rt_stats* stats) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  if (!rgb_out) return RT_ERR_INVALID;
  const int rows = rt_hip_rank_rows(cur->scene.vsize, p->band_rows, p->n_ranks, p->rank);
  const size_t bytes = (size_t)rows * cur->scene.hsize * 3;
  if (bytes > cur->d_rgb_bytes) {
    if (cur->d_rgb) (void)hipFree(cur->d_rgb);
    cur->d_rgb = nullptr;
    cur->d_rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_rgb, bytes));
    cur->d_rgb_bytes = bytes;
  }
  rc = render_common(p, cur->d_rgb, nullptr, cur->stream, stats, true);
  if (rc != RT_OK) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(rgb_out, cur->d_rgb, bytes, hipMemcpyDeviceToHost, cur->stream));
  HIP_TRY(hipStreamSynchronize(cur->stream));
  return RT_OK;
}

int rt_hip_render_f64(const rt_render_params* p, double* canvas_out, rt_stats* stats) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  if (!canvas_out) return RT_ERR_INVALID;
  const int rows = rt_hip_rank_rows(cur->scene.vsize, p->band_rows, p->n_ranks, p->rank);
  const size_t bytes = (size_t)rows * cur->scene.hsize * 3 * sizeof(double);
  if (bytes > cur->d_canvas_bytes) {
    if (cur->d_canvas) (void)hipFree(cur->d_canvas);
    cur->d_canvas = nullptr;
    cur->d_canvas_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_canvas, bytes));
    cur->d_canvas_bytes = bytes;
  }
  rc = render_common(p, nullptr, cur->d_canvas, cur->stream, stats, true);
  if (rc != RT_OK) return rc;
  if (bytes) HIP_TRY(hipMemcpyAsync(canvas_out, cur->d_canvas, bytes, hipMemcpyDeviceToHost, cur->stream));
  HIP_TRY(hipStreamSynchronize(cur->stream));
  return RT_OK;
}

#if RT_STAMPS
/* development builds: the per-wave time stamps of the last launch (rt_hip_debug_stamp_words() words per wave) */
int rt_hip_debug_stamp_words(void) { return RT_STAMP_WORDS; }
int rt_hip_debug_stamps(unsigned long long* out, int n_waves) {
  if (!cur->ready || n_waves < 0 || n_waves > RT_STAMPS_MAX_WAVES) return RT_ERR_INVALID;
  HIP_TRY(hipStreamSynchronize(cur->stream));
  HIP_TRY(hipMemcpy(out, cur->d_counters + 16, sizeof(unsigned long long) * RT_STAMP_WORDS * (size_t)n_waves, hipMemcpyDeviceToHost));
  return RT_OK;
}
#endif

int rt_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int rt_hip_event_record(int which, void* stream) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (which != 0 && which != 1) return RT_ERR_INVALID;
  hipStream_t s = stream ? (hipStream_t)stream : cur->stream;
  HIP_TRY(hipEventRecord(which == 0 ? cur->tm0 : cur->tm1, s));
  return RT_OK;
}

int rt_hip_event_elapsed_ms(double* ms_out) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (!ms_out) return RT_ERR_INVALID;
  HIP_TRY(hipEventSynchronize(cur->tm1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, cur->tm0, cur->tm1));
  *ms_out = ms;
  return RT_OK;
}

void rt_hip_shutdown(void) {
  multi_teardown();
  for (int d = 0; d < kMaxDevices; ++d)
    if (g_ctx[d].ready || g_ctx[d].stream) destroy_context(&g_ctx[d]);
  cur = &g_ctx[0];
}

const char* rt_hip_strerror(int status) {
  switch (status) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID: return "invalid argument";
    case RT_ERR_NO_DEVICE: return "no such HIP device";
    case RT_ERR_HIP: return "HIP runtime error";
    case RT_ERR_UNSUPPORTED: return "scene uses a feature outside the accelerated path";
    case RT_ERR_NOT_READY: return "rt_hip_init / rt_hip_upload_scene not called";
    case RT_ERR_TOO_LARGE: return "scene or image exceeds a library limit";
    default: return "unknown status";
  }
}

const char* rt_hip_last_error(void) { return g_err; }

#ifdef RT_DEV_SWITCHES
/* Development build only (no device needed): the tile plan the library would make for `sc` and `p`, as the words it
 * would copy to the device.  words[0 .. info[0]): entries (tile | part << 22 | log2 parts << 28), then what the
 * kernels read behind them: from word info[3] on four 64-bit masks per entry (info[3] == 0: none); with lists
 * (info[4] != 0) the first two of those are (offset | count << 32) of the entry's camera rays' and shadow rays'
 * triangle list, whose triangle numbers lie from word info[4] + 16 * info[5] + offset on (rt_kernels.hip.inc:
 * stage_list).  info: 0 words, 1 entries, 2 mesh entries, 3 mask offset, 4 list_forms, 5 list_total, 6 tiles per
 * image row, 7 light tiles are background, 8 BVH in use, 9 triangles are in the scene's BVH leaf order (1) or as
 * uploaded (0).  Behind the plan's words follow, one word per sphere, the caller's number of device sphere k (the
 * masks count spheres in device order: smallest first).  RT_ERR_TOO_LARGE: cap_words is too small (info[0] plus the
 * number of spheres are needed). */
int rt_hip_debug_plan_host(const rt_scene* sc, const rt_render_params* p, int32_t* words, size_t cap_words, int64_t info[10]) {
  if (!sc || !p || !info) return RT_ERR_INVALID;
  static Context scratch;
  Context* const keep = cur;
  scratch = Context{};
  scratch.ready = true;
  cur = &scratch;
  g_host_only = true;
  int rc = rt_hip_upload_scene(sc);
  RenderArgs a;
  memset(&a, 0, sizeof a);
  int n_blocks = 0;
  bool bvh = false;
  if (rc == RT_OK) rc = check_params(p);
  if (rc == RT_OK) {
    const int rows = rt_hip_rank_rows(cur->scene.vsize, p->band_rows, p->n_ranks, p->rank);
    bvh = cur->scene.n_bvh_nodes > 0 && cur->scene.n_tris > 0;
    if (p->accel == RT_ACCEL_BRUTE) bvh = false;
    if (p->accel == RT_ACCEL_AUTO && cur->scene.n_tris < 32) bvh = false;
    a.rows = rows;
    rc = plan_tiles(p, rows, bvh, cur->reflective && p->max_depth >= 1, (p->flags & RT_RENDER_FRAMES_OVERLAP) != 0, true, nullptr, a, n_blocks);
  }
  if (rc == RT_OK) {
    const Context::TilePlan* plan = cur->launch_plan;
    const std::vector<int32_t> none;
    const std::vector<int32_t>& w = plan ? plan->debug_words : none;
    info[0] = (int64_t)w.size();
    info[1] = plan ? plan->entries : 0;
    info[2] = plan ? plan->heavy : 0;
    info[3] = plan ? (int64_t)plan->mask_offset : 0;
    info[4] = a.list_forms;
    info[5] = a.list_total;
    info[6] = (cur->scene.hsize + RT_TILE_W - 1) / RT_TILE_W;
    info[7] = a.light_is_background;
    info[8] = bvh ? 1 : 0;
    info[9] = cur->scene.n_bvh_nodes > 0 ? 1 : 0;
    if (w.size() + cur->sph_upload_index.size() > cap_words) rc = RT_ERR_TOO_LARGE;
    else if (words) {
      if (!w.empty()) memcpy(words, w.data(), sizeof(int32_t) * w.size());
      for (size_t k = 0; k < cur->sph_upload_index.size(); ++k) words[w.size() + k] = cur->sph_upload_index[k];
    }
  }
  g_host_only = false;
  scratch = Context{};
  cur = keep;
  return rc;
}
#endif

}  // extern "C"

#include "rt_lbvh_api.hip.inc"
#include "rt_ppm_api.hip.inc"
#include "rt_multi.hip.inc"
