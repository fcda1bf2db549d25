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
  bool crowded_first = true; /* mesh tiles in descending order of the triangles that project into them */
  int split_tiles = 256, split_parts = 2, split_min_cost = 16, split_quarters = 8, split_quarters_reflect = 128, split_max_per_cu = 4; /* the most crowded tiles run as 2 or 4 workgroups with a half or a quarter of the lanes each */
  bool part_lists = true;   /* parts of split tiles get lists of their own (walks) */
  int part_min_lg = 2;      /* entries split into at least 2^this parts get lists of their own */
  int walk_shadow_max = 0;  /* walks: a whole tile's shadow rays run down their block's list if it has at most this many triangles */
  int part_shadow_max = 0;  /* ... and lists for their shadow rays up to this length (0: none, the default: with lists of
                               up to 48 or 96 the lighthouse frames are the same to the microsecond, and making them costs
                               3-4 ms when the tiles are planned; from 256 up the frames get slower) */
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
    int rows = 0, band_rows = 1, n_ranks = 1, rank = 0; /* the launch's packed rows and how they map to image rows */
    size_t mask_offset = 0;        /* d_list + mask_offset: four 64-bit masks per entry (0: none): the triangles a camera
                                      ray of the tile can hit, those a shadow ray from such a hit can, and the same for spheres */
    int refine_left = 0;           /* measured refinements of the split still to do (see refine_split) */
    bool refining = false;         /* refine_split is at work on this plan */
    std::vector<int32_t> host_list; /* the list as uploaded, kept while refinements are pending */
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

int upload(void** dst, const void* src, size_t bytes) {
  if (bytes == 0) bytes = 16;
  HIP_TRY(hipMalloc(dst, bytes));
  if (src) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return RT_OK;
}

/* The single-precision form of a sphere.  inv: rows 0..2 of inverse(transform); origin: the camera origin;
 * radius as for rt_make_trif.  The filter compares |o' x d'|^2 with |d'|^2.  With eps = 2^-24, Mn = the Frobenius norm
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
  for (int i = 0; i < 12; ++i) out->inv[i] = (float)inv[i];
  for (int r = 0; r < 3; ++r)
    out->po[r] = (float)(((origin[0] * inv[4 * r] + origin[1] * inv[4 * r + 1]) + origin[2] * inv[4 * r + 2]) + inv[4 * r + 3]);
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
constexpr size_t kMaxLdsPerWorkgroup = 80 * 1024;
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
      lds = fit_lists(RT_PARK_BYTES, (size_t)160 * 1024 / 3) - RT_PARK_BYTES;
    } else if (!bvh && cur->scene.n_tris > 0 && cur->scene.n_tris <= kMaxLdsTris) {
      stage = 1;
      lds = (size_t)cur->scene.n_tris * sizeof(DevTriF);
    } else if (bvh && cur->lds_stack && cur->scene.bvh_depth >= 1) {
      /* the traversal stacks, and in front of them the inner nodes if both fit while two workgroups still share
       * a CU's 160 KiB (the walk's register budget allows no more than two anyway) */
      const size_t stacks = (size_t)RT_TILE_W * RT_TILE_H * sizeof(int2) * (size_t)cur->scene.bvh_depth;
      const size_t nodes = (size_t)cur->scene.n_bvh_nodes * sizeof(DevBvhNode);
      if (cur->scene.n_bvh_nodes <= kMaxLdsBvhNodes && nodes + stacks <= kMaxLdsPerWorkgroup) {
        stage = 3;
        lds = nodes + stacks;
        lds = fit_lists(lds, kMaxLdsPerWorkgroup); /* longer lists are read from memory */
      } else if (stacks <= (reflect ? kMaxLdsPerWorkgroup : (size_t)160 * 1024 / 3)) {
        /* with the nodes in global memory the kernels that need few registers run three workgroups per CU:
         * stacks that would cut that to two cost more than they save (1.31 M triangles, depth 27: +6 %); the
         * kernels with reflections run two per CU in any case (same mesh, reflective: -10 % with the stacks) */
        stage = 4;
        lds = stacks;
        lds = fit_lists(lds, reflect ? kMaxLdsPerWorkgroup : (size_t)160 * 1024 / 3);
      }
    }
  }
  if ((bvh && stage == 0) || !cur->use_lds) {
    a.list_lds = 0;
    a.slist_lds = 0;
  }
  const int kind = (!bvh && cur->scene.n_tris == 0) ? 1 : (cur->scene.n_spheres == 0 ? 2 : 0);
  cur->launch_lds = lds;
  if (reflect) launch_family<true>(bvh, stage, kind, lds, grid, s, a, rgb, canvas);
  else launch_family<false>(bvh, stage, kind, lds, grid, s, a, rgb, canvas);
}

/* general 4x4 inverse (Gauss-Jordan, partial pivoting); only used for scheduling hints and padding scales */
bool invert4(const double* m, double* out) {
  double a[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { a[r][c] = m[4 * r + c]; a[r][4 + c] = r == c ? 1.0 : 0.0; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (std::fabs(a[piv][c]) < 1e-300) return false;
    for (int k = 0; k < 8; ++k) std::swap(a[c][k], a[piv][k]);
    const double d = a[c][c];
    for (int k = 0; k < 8; ++k) a[c][k] /= d;
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
    }
  }
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[4 * r + c] = a[r][4 + c];
  return true;
}

/* Pixel rectangle that contains every pixel whose camera rays can reach the mesh bounds: the bounds' eight
 * corners through the view transform and the canvas mapping of ray_for_pixel (RayTracer.c:1538-1544) run
 * backwards, plus a margin.  false = no useful rectangle (a corner beside or behind the camera). */
bool mesh_pixel_rect(const rt_camera& cam, double& x0, double& y0, double& x1, double& y1) {
  double view[16];
  if (!invert4(cam.inv_view, view)) return false;
  x0 = y0 = 1e300;
  x1 = y1 = -1e300;
  for (int k = 0; k < 8; ++k) {
    const double w[3] = {(k & 1) ? cur->mesh_hi[0] : cur->mesh_lo[0], (k & 2) ? cur->mesh_hi[1] : cur->mesh_lo[1],
                         (k & 4) ? cur->mesh_hi[2] : cur->mesh_lo[2]};
    double c[3];
    for (int r = 0; r < 3; ++r) c[r] = view[4 * r] * w[0] + view[4 * r + 1] * w[1] + view[4 * r + 2] * w[2] + view[4 * r + 3];
    if (!(c[2] < -1e-6)) return false;
    const double wx = c[0] / -c[2], wy = c[1] / -c[2];
    const double px = (cam.half_width - wx) / cam.pixel_size - 0.5, py = (cam.half_height - wy) / cam.pixel_size - 0.5;
    x0 = std::fmin(x0, px); x1 = std::fmax(x1, px);
    y0 = std::fmin(y0, py); y1 = std::fmax(y1, py);
  }
  x0 -= 3.0; y0 -= 3.0; x1 += 3.0; y1 += 3.0;
  return std::isfinite(x0) && std::isfinite(x1) && std::isfinite(y0) && std::isfinite(y1);
}

/* The cached tile plan for `key`, or the least recently used slot with `fresh` set.  Launches still in flight
 * (on any stream) may read the list of a slot that is about to be rewritten, so eviction waits for the device;
 * a slot that was never used needs no wait. */
Context::TilePlan* find_plan(const long long key[6], bool& fresh) {
  Context::TilePlan* victim = &cur->plans[0];
  for (Context::TilePlan& t : cur->plans) {
    if (memcmp(key, t.key, sizeof t.key) == 0) {
      fresh = false;
      t.last_use = ++cur->plan_clock;
      return &t;
    }
    if (t.last_use < victim->last_use) victim = &t;
  }
  fresh = true;
  return victim;
}
/* fill an evicted slot with `list` */
int store_plan(Context::TilePlan* t, const long long key[6], const std::vector<int32_t>& list, hipStream_t s) {
  if (t->last_use != 0) HIP_TRY(hipDeviceSynchronize());
  t->key[0] = -1;
  if (list.size() > t->cap) {
    if (t->d_list) (void)hipFree(t->d_list);
    t->d_list = nullptr;
    t->cap = 0;
    /* a plan whose split is still being refined is stored again with a longer list every round: room for that */
    const size_t want = t->refine_left > 0 || t->refining ? list.size() * 2 : list.size();
    HIP_TRY(hipMalloc((void**)&t->d_list, sizeof(int32_t) * want));
    t->cap = want;
  }
  HIP_TRY(hipMemcpyAsync(t->d_list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s)); /* the host vector goes away */
  memcpy(t->key, key, sizeof t->key);
  t->last_use = ++cur->plan_clock;
  return RT_OK;
}

/* Which tiles of the launch can a camera ray reach a triangle through?  Every triangle's projection (a triangle
 * again: all corners lie in front of the camera) is rasterised conservatively into the tile grid: a tile counts
 * as covered unless it lies outside the triangle's pixel rectangle or wholly on the outer side of one of its
 * edges, both with a margin of two pixels.  (The reference accepts a hit by u >= 0, v >= 0, u + v <= 1 in double
 * precision; its rounding moves an edge by parts in 1e11 of the triangle's size, the projection here is rounded
 * as finely, and a ray through a sub-pixel sample stays inside its pixel.)  A tile that is not covered shows
 * nothing of a world without spheres, and in a covered tile the camera rays need to be tested against the
 * triangles that cover it only (`masks`).  y_lo / y_hi: the image rows of every tile row (bands make them differ
 * from the packed rows).  false: unknown. */
bool covered_tiles(int tiles_x, int tiles_y, const std::vector<int>& y_lo, const std::vector<int>& y_hi, bool rows_are_y,
                   std::vector<unsigned char>& cover, std::vector<unsigned long long>* masks,
                   std::vector<std::vector<int32_t>>* lists = nullptr) {
  if (cur->tri_xy.empty()) return false;
  const double margin = 2.0;
  cover.assign((size_t)tiles_x * tiles_y, 0);
  const size_t nt = cur->tri_xy.size() / 6;
  if (masks) masks->assign((size_t)tiles_x * tiles_y, 0ull); /* bit k of a tile: triangle k may be seen through it (nt <= 64) */
  if (lists) lists->assign((size_t)tiles_x * tiles_y, std::vector<int32_t>()); /* the same as lists of triangle numbers */
  for (size_t k = 0; k < nt; ++k) {
    const double* q = &cur->tri_xy[6 * k];
    double bx0 = std::fmin(q[0], std::fmin(q[2], q[4])) - margin, bx1 = std::fmax(q[0], std::fmax(q[2], q[4])) + margin;
    double by0 = std::fmin(q[1], std::fmin(q[3], q[5])) - margin, by1 = std::fmax(q[1], std::fmax(q[3], q[5])) + margin;
    if (!(bx0 <= bx1 && by0 <= by1)) return false; /* NaN */
    if (bx1 < 0.0 || by1 < 0.0 || bx0 > (double)tiles_x * RT_TILE_W) continue;
    const int tx0 = std::max(0, (int)std::floor(bx0 / RT_TILE_W)), tx1 = std::min(tiles_x - 1, (int)std::floor(bx1 / RT_TILE_W));
    /* edge functions, oriented so that the inside is >= 0; a sliver's orientation is not trusted (rectangle only) */
    const double area2 = (q[2] - q[0]) * (q[5] - q[1]) - (q[3] - q[1]) * (q[4] - q[0]);
    const double extent = (bx1 - bx0) + (by1 - by0);
    const bool edges = std::fabs(area2) > 1e-6 * extent * extent;
    const double sg = area2 > 0 ? 1.0 : -1.0;
    int ty0 = 0, ty1 = tiles_y - 1;
    if (rows_are_y) {
      ty0 = std::max(0, (int)std::floor(by0 / RT_TILE_H));
      ty1 = std::min(tiles_y - 1, (int)std::floor(by1 / RT_TILE_H));
    }
    for (int ty = ty0; ty <= ty1; ++ty) {
      if ((double)y_hi[ty] + 1.0 < by0 || (double)y_lo[ty] > by1) continue;
      const double ylo = (double)y_lo[ty] - margin, yhi = (double)y_hi[ty] + 1.0 + margin;
      for (int tx = tx0; tx <= tx1; ++tx) {
        unsigned char& c = cover[(size_t)ty * tiles_x + tx];
        if (c && !masks && !lists) continue;
        bool out = false;
        if (edges) {
          const double xlo = (double)tx * RT_TILE_W - margin, xhi = (double)(tx + 1) * RT_TILE_W + margin;
          for (int e = 0; e < 3 && !out; ++e) {
            const double ax = q[2 * e], ay = q[2 * e + 1];
            const double dx = sg * (q[2 * ((e + 1) % 3)] - ax), dy = sg * (q[2 * ((e + 1) % 3) + 1] - ay);
            /* the corner of the (padded) tile that lies deepest on the inner side of the edge */
            const double cx = dy < 0 ? xhi : xlo, cy = dx > 0 ? yhi : ylo;
            out = dx * (cy - ay) - dy * (cx - ax) < 0.0;
          }
        }
        if (!out) {
          c = 1;
          if (masks) (*masks)[(size_t)ty * tiles_x + tx] |= 1ull << k;
          if (lists) (*lists)[(size_t)ty * tiles_x + tx].push_back((int32_t)k);
        }
      }
    }
  }
  /* nearest first (by the nearest corner): a ray that has its hit early can leave most of what lies behind it to the
   * single-precision test (list_nearest); the order decides nothing else */
  if (lists && cur->tri_depth.size() == nt)
    for (std::vector<int32_t>& l : *lists)
      std::stable_sort(l.begin(), l.end(), [](int32_t a_, int32_t b_) { return cur->tri_depth[(size_t)a_] < cur->tri_depth[(size_t)b_]; });
  return true;
}


/* The test of covered_tiles for any pixel rectangle [x0, x1) x [y0, y1) of the image: false only if no camera ray
 * through it can reach triangle k (same margins). */
bool tri_reaches_rect(size_t k, double x0, double x1, double y0, double y1) {
  const double margin = 2.0;
  const double* q = &cur->tri_xy[6 * k];
  const double bx0 = std::fmin(q[0], std::fmin(q[2], q[4])) - margin, bx1 = std::fmax(q[0], std::fmax(q[2], q[4])) + margin;
  const double by0 = std::fmin(q[1], std::fmin(q[3], q[5])) - margin, by1 = std::fmax(q[1], std::fmax(q[3], q[5])) + margin;
  if (!(bx0 <= bx1 && by0 <= by1)) return true; /* NaN */
  if (bx1 < x0 || bx0 > x1 || by1 < y0 || by0 > y1) return false;
  const double area2 = (q[2] - q[0]) * (q[5] - q[1]) - (q[3] - q[1]) * (q[4] - q[0]);
  const double extent = (bx1 - bx0) + (by1 - by0);
  if (!(std::fabs(area2) > 1e-6 * extent * extent)) return true; /* a sliver: the rectangle only */
  const double sg = area2 > 0 ? 1.0 : -1.0;
  const double xlo = x0 - margin, xhi = x1 + margin, ylo = y0 - margin, yhi = y1 + margin;
  for (int e = 0; e < 3; ++e) {
    const double ax = q[2 * e], ay = q[2 * e + 1];
    const double dx = sg * (q[2 * ((e + 1) % 3)] - ax), dy = sg * (q[2 * ((e + 1) % 3) + 1] - ay);
    const double cx = dy < 0 ? xhi : xlo, cy = dx > 0 ? yhi : ylo;
    if (dx * (cy - ay) - dy * (cx - ax) < 0.0) return false;
  }
  return true;
}

struct V3 {
  double x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double f) { return {a.x * f, a.y * f, a.z * f}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len(V3 a) { return std::sqrt(dot(a, a)); }

/* Which primitives can stand between the light and a point that a camera ray through the pixel rectangle
 * [x0, x1] x [y0, y1] sees on one of the triangles `seen_tri` or spheres `seen_sph`?  (A world of at most 64
 * triangles and 64 spheres, every corner of every triangle and of every sphere's box in front of the camera; a
 * sphere's box: the image of [-1, 1]^3 under its transform.)  The points such a ray can see on primitive j lie in
 * S = the part of j (a sphere: of its box) inside the rectangle's four-sided pyramid of rays; is_shadowed
 * (RayTracer.c:1410) sends its ray from a point EPSILON off the surface (:1371) to the light L and accepts hits on
 * that stretch only, so everything it can meet lies within `pad` (several EPSILON, plus rounding at the scale of
 * the scene) of the convex hull of S and L.  Primitive k is left out when a plane shows it to lie wholly outside
 * that padded hull.  The planes tried: for a triangle j, its own plane and the planes through L and an edge of S
 * (the hull is a pyramid over S with its apex in L); the plane of a triangle k; the six faces of the box of a
 * sphere k.  Anything doubtful (a sliver, L almost in the plane of j, an eye inside a box) only loses planes to
 * test with, or keeps everything. */
void shadow_blockers(double x0, double y0, double x1, double y1, unsigned long long seen_tri, unsigned long long seen_sph, double pad,
                     unsigned long long& block_tri, unsigned long long& block_sph) {
  const rt_camera& cam = cur->host_camera;
  const double* iv = cam.inv_view;
  const size_t nt = cur->tri_world.size() / 9, ns = cur->sph_world.size() / 24;
  const unsigned long long all_tri = nt >= 64 ? ~0ull : (1ull << nt) - 1ull, all_sph = ns >= 64 ? ~0ull : (1ull << ns) - 1ull;
  auto vert = [&](size_t k, int v) { const double* q = &cur->tri_world[9 * k + 3 * v]; return V3{q[0], q[1], q[2]}; };
  auto corner = [&](size_t k, int v) { const double* q = &cur->sph_world[24 * k + 3 * v]; return V3{q[0], q[1], q[2]}; };
  const V3 O = {iv[3], iv[7], iv[11]}, L = {cur->host_light[0], cur->host_light[1], cur->host_light[2]};
  V3 d[4], dc = {0, 0, 0};
  const double cx[4] = {x0, x1, x1, x0}, cy[4] = {y0, y0, y1, y1};
  for (int c = 0; c < 4; ++c) {
    const double wx = cam.half_width - cx[c] * cam.pixel_size, wy = cam.half_height - cy[c] * cam.pixel_size;
    d[c] = {iv[0] * wx + iv[1] * wy - iv[2], iv[4] * wx + iv[5] * wy - iv[6], iv[8] * wx + iv[9] * wy - iv[10]};
    dc = dc + d[c];
  }
  V3 side[4];
  for (int c = 0; c < 4; ++c) {
    side[c] = cross(d[c], d[(c + 1) % 4]);
    if (dot(side[c], dc) < 0) side[c] = side[c] * -1.0;
  }
  /* a convex polygon clipped by the four side planes of the pyramid (in place; returns the number of corners) */
  auto clip = [&](V3* poly, int np) {
    V3 tmp[16];
    for (int c = 0; c < 4 && np > 0; ++c) {
      int nq = 0;
      for (int i = 0; i < np; ++i) {
        const V3 a = poly[i], b = poly[(i + 1) % np];
        const double da = dot(side[c], a - O), db = dot(side[c], b - O);
        if (da >= 0) tmp[nq++] = a;
        if ((da >= 0) != (db >= 0)) tmp[nq++] = a + (b - a) * (da / (da - db));
      }
      np = nq;
      for (int i = 0; i < np; ++i) poly[i] = tmp[i];
    }
    return np;
  };
  struct Plane { V3 n; double o, lo, hi; }; /* the hull within lo <= n . x - o <= hi */
  /* everything a hull of `pts` (np points) and L cannot be told apart from, given planes that bound that hull */
  auto add_blockers = [&](const V3* pts, int np, const Plane* pl, int n_pl) {
    auto apart_by_planes = [&](const V3* kv, int nk) {
      for (int q = 0; q < n_pl; ++q) {
        bool below = true, above = true;
        for (int v = 0; v < nk; ++v) {
          const double sv = dot(pl[q].n, kv[v]) - pl[q].o;
          below = below && sv < pl[q].lo - pad;
          above = above && sv > pl[q].hi + pad;
        }
        if (below || above) return true;
      }
      return false;
    };
    for (size_t k = 0; k < nt; ++k) {
      if ((block_tri >> k) & 1ull) continue;
      const V3 kv[3] = {vert(k, 0), vert(k, 1), vert(k, 2)};
      bool apart = apart_by_planes(kv, 3);
      if (!apart) { /* the plane of k: the points and L all on one side of it */
        V3 nk = cross(kv[1] - kv[0], kv[2] - kv[0]);
        const double lk = len(nk);
        if (lk > 0) {
          nk = nk * (1.0 / lk);
          double lo = dot(nk, L - kv[0]), hi = lo;
          for (int v = 0; v < np; ++v) {
            const double sv = dot(nk, pts[v] - kv[0]);
            lo = std::fmin(lo, sv);
            hi = std::fmax(hi, sv);
          }
          apart = lo > pad || hi < -pad;
        }
      }
      if (!apart) block_tri |= 1ull << k;
    }
    for (size_t k = 0; k < ns; ++k) {
      if ((block_sph >> k) & 1ull) continue;
      V3 kv[8];
      for (int v = 0; v < 8; ++v) kv[v] = corner(k, v);
      bool apart = apart_by_planes(kv, 8);
      if (!apart) { /* the faces of k's box: the points and L all beyond one of them, in k's object space */
        const double* m = &cur->sph_inv[12 * k];
        const double padk = pad * cur->sph_norm[k];
        for (int r = 0; r < 3 && !apart; ++r) {
          double lo = m[4 * r] * L.x + m[4 * r + 1] * L.y + m[4 * r + 2] * L.z + m[4 * r + 3], hi = lo;
          for (int v = 0; v < np; ++v) {
            const double sv = m[4 * r] * pts[v].x + m[4 * r + 1] * pts[v].y + m[4 * r + 2] * pts[v].z + m[4 * r + 3];
            lo = std::fmin(lo, sv);
            hi = std::fmax(hi, sv);
          }
          apart = lo > 1.0 + padk || hi < -1.0 - padk;
        }
      }
      if (!apart) block_sph |= 1ull << k;
    }
  };
  for (size_t j = 0; j < nt; ++j) {
    if (!((seen_tri >> j) & 1ull)) continue;
    block_tri |= 1ull << j;
    V3 poly[16];
    for (int v = 0; v < 3; ++v) poly[v] = vert(j, v);
    const int np = clip(poly, 3);
    if (np < 1) continue; /* no camera ray of this tile reaches j after all */
    Plane pl[20];
    int n_pl = 0;
    const V3 p1 = vert(j, 0);
    V3 nj = cross(vert(j, 1) - p1, vert(j, 2) - p1);
    const double lj = len(nj);
    if (lj > 0) {
      nj = nj * (1.0 / lj);
      const double dl = dot(nj, L - p1);
      pl[n_pl++] = {nj, dot(nj, p1), std::fmin(0.0, dl), std::fmax(0.0, dl)};
    }
    for (int i = 0; i < np && np >= 3; ++i) {
      const V3 a = poly[i] - L, b = poly[(i + 1) % np] - L;
      V3 n = cross(a, b);
      const double ln = len(n);
      if (!(ln > 1e-12 * len(a) * len(b))) continue;
      n = n * (1.0 / ln);
      /* which side the rest of S lies on; an edge whose plane the other corners barely leave decides nothing */
      double smin = 0.0, smax = 0.0;
      for (int v = 0; v < np; ++v) {
        const double sv = dot(n, poly[v] - L);
        smin = std::fmin(smin, sv);
        smax = std::fmax(smax, sv);
      }
      if (smax > pad && smin >= -1e-9 * (1.0 + smax)) pl[n_pl++] = {n, dot(n, L), 0.0, 1e300};
      else if (smin < -pad && smax <= 1e-9 * (1.0 - smin)) pl[n_pl++] = {n * -1.0, -dot(n, L), 0.0, 1e300};
    }
    add_blockers(poly, np, pl, n_pl);
  }
  for (size_t j = 0; j < ns; ++j) {
    if (!((seen_sph >> j) & 1ull)) continue;
    block_sph |= 1ull << j;
    /* an eye inside the box sees its inside: no S worth the name */
    const double* m = &cur->sph_inv[12 * j];
    bool eye_inside = true;
    for (int r = 0; r < 3; ++r)
      eye_inside = eye_inside && std::fabs(m[4 * r] * O.x + m[4 * r + 1] * O.y + m[4 * r + 2] * O.z + m[4 * r + 3]) <= 1.0 + 1e-6;
    if (eye_inside) { block_tri = all_tri; block_sph = all_sph; return; }
    /* S: the corners of the six faces of the box, each face clipped by the pyramid */
    V3 pts[6 * 16];
    int np = 0;
    for (int axis = 0; axis < 3; ++axis)
      for (int side_ = 0; side_ < 2; ++side_) {
        const int b1 = (axis + 1) % 3, b2 = (axis + 2) % 3;
        V3 quad[16];
        const int loop[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
        for (int v = 0; v < 4; ++v) quad[v] = corner(j, (side_ << axis) | (loop[v][0] << b1) | (loop[v][1] << b2));
        const int nq = clip(quad, 4);
        for (int v = 0; v < nq; ++v) pts[np++] = quad[v];
      }
    if (np < 1) continue;
    add_blockers(pts, np, nullptr, 0);
  }
}

/* Which triangles can stand between the light and a point that a camera ray of a tile sees?  (A world without
 * spheres, every corner in front of the camera; cam_lists: per tile, the triangles its camera rays can hit.)
 * The points such a ray can see lie in S = the listed triangles clipped to the tile's pyramid of rays (two pixels of
 * margin, as in the rasterisation); is_shadowed (RayTracer.c:1410) sends its ray from a point EPSILON off the surface
 * (:1371) to the light L and accepts hits on that stretch only, so everything it can meet lies within `pad` of the
 * convex hull of S and L.  That hull is bounded from L's point of view: with m the coordinate axis along which S
 * lies farthest from L, every point x of the hull other than L has (x - L)_m of the sign of S's, at most S's largest,
 * and its central projection from L onto the plane (x - L)_m = +-1 falls into the rectangle that holds the
 * projections of S (padded).  A triangle wholly on that side of L projects to the triangle of its corners'
 * projections, so it can only meet the hull if the rectangle of those meets S's and its nearest corner is not beyond
 * S; a triangle wholly on the other side cannot; one with corners on both sides is kept.  The lists are made per
 * block of kListShadowBlock x kListShadowBlock tiles.  ok[tile] == 0: nothing known for this tile (S reaches around
 * the light): its shadow rays test everything. */
constexpr int kListShadowBlock = 2;
/* every triangle as the light L sees it along each axis: 0 = wholly on the + side, 1 = wholly on the - side, 2 = neither */
struct LightChart { double a0, a1, b0, b1, near; int state; };
void light_charts(double pad, std::vector<LightChart>& chart) {
  const size_t nt = cur->mesh_world.size() / 9;
  const V3 L = {cur->host_light[0], cur->host_light[1], cur->host_light[2]};
  chart.resize(3 * nt);
  for (size_t k = 0; k < nt; ++k)
    for (int m = 0; m < 3; ++m) {
      LightChart c = {1e300, -1e300, 1e300, -1e300, 1e300, 0};
      int pos = 0, neg = 0;
      for (int v = 0; v < 3; ++v) {
        const double* w = &cur->mesh_world[9 * k + 3 * (size_t)v];
        const double q[3] = {w[0] - L.x, w[1] - L.y, w[2] - L.z};
        const double qm = q[m];
        if (qm > pad) pos++;
        else if (qm < -pad) neg++;
        else continue;
        const double am = std::fabs(qm), a = q[(m + 1) % 3] / am, b = q[(m + 2) % 3] / am;
        c.a0 = std::fmin(c.a0, a); c.a1 = std::fmax(c.a1, a);
        c.b0 = std::fmin(c.b0, b); c.b1 = std::fmax(c.b1, b);
        c.near = std::fmin(c.near, am);
      }
      c.state = pos == 3 ? 0 : (neg == 3 ? 1 : 2);
      if (!(c.a0 <= c.a1 && c.b0 <= c.b1)) c.state = 2; /* NaN */
      chart[3 * k + (size_t)m] = c;
    }
}
/* the corners of the triangles `seen` clipped to the pyramid of camera rays through the pixel rectangle [x0, x1] x
 * [y0, y1] (margins included by the caller), appended to pts */
void clip_to_pixel_pyramid(double x0, double x1, double y0, double y1, const std::vector<int32_t>& seen, std::vector<V3>& pts) {
  const rt_camera& cam = cur->host_camera;
  const double* iv = cam.inv_view;
  const V3 O = {iv[3], iv[7], iv[11]};
  const double cx[4] = {x0, x1, x1, x0}, cy[4] = {y0, y0, y1, y1};
  V3 d[4], dc = {0, 0, 0}, side[4];
  for (int c = 0; c < 4; ++c) {
    const double wx = cam.half_width - cx[c] * cam.pixel_size, wy = cam.half_height - cy[c] * cam.pixel_size;
    d[c] = {iv[0] * wx + iv[1] * wy - iv[2], iv[4] * wx + iv[5] * wy - iv[6], iv[8] * wx + iv[9] * wy - iv[10]};
    dc = dc + d[c];
  }
  for (int c = 0; c < 4; ++c) {
    side[c] = cross(d[c], d[(c + 1) % 4]);
    if (dot(side[c], dc) < 0) side[c] = side[c] * -1.0;
  }
  for (int32_t j : seen) {
    V3 poly[16], tmp[16];
    int np = 3;
    for (int v = 0; v < 3; ++v) { const double* q = &cur->mesh_world[9 * (size_t)j + 3 * (size_t)v]; poly[v] = V3{q[0], q[1], q[2]}; }
    for (int c = 0; c < 4 && np > 0; ++c) {
      int nq = 0;
      for (int i = 0; i < np; ++i) {
        const V3 a = poly[i], b = poly[(i + 1) % np];
        const double da = dot(side[c], a - O), db = dot(side[c], b - O);
        if (da >= 0) tmp[nq++] = a;
        if ((da >= 0) != (db >= 0)) tmp[nq++] = a + (b - a) * (da / (da - db));
      }
      np = nq;
      for (int i = 0; i < np; ++i) poly[i] = tmp[i];
    }
    for (int i = 0; i < np; ++i) pts.push_back(poly[i]);
  }
}
/* the triangles that can meet the padded hull of the points pts and the light; false: unknown (the points reach around it) */
bool blockers_of_points(const std::vector<V3>& pts, const std::vector<LightChart>& chart, double pad, std::vector<int32_t>& blockers) {
  const size_t nt = chart.size() / 3;
  const V3 L = {cur->host_light[0], cur->host_light[1], cur->host_light[2]};
  blockers.clear();
  if (pts.empty()) return true;
  V3 sum = {0, 0, 0};
  for (const V3& p : pts) sum = sum + (p - L);
  const double sa[3] = {sum.x, sum.y, sum.z};
  int m = 0;
  for (int c = 1; c < 3; ++c) if (std::fabs(sa[c]) > std::fabs(sa[m])) m = c;
  const double sg = sa[m] < 0 ? -1.0 : 1.0;
  double a0 = 1e300, a1 = -1e300, b0 = 1e300, b1 = -1e300, far = 0.0;
  for (const V3& p : pts) {
    const double q[3] = {p.x - L.x, p.y - L.y, p.z - L.z};
    const double qm = sg * q[m];
    if (!(qm > 4.0 * pad)) return false;
    far = std::fmax(far, qm + pad);
    for (int e = 0; e < 4; ++e) { /* the point moved by up to pad along every axis */
      const double dq = (e & 1) ? pad : -pad, dm = (e & 2) ? pad : -pad;
      const double a = (q[(m + 1) % 3] + dq) / (qm + dm), b = (q[(m + 2) % 3] + dq) / (qm + dm);
      a0 = std::fmin(a0, a); a1 = std::fmax(a1, a);
      b0 = std::fmin(b0, b); b1 = std::fmax(b1, b);
    }
  }
  const double ea = 1e-9 * (1.0 + std::fmax(std::fabs(a0), std::fabs(a1))), eb = 1e-9 * (1.0 + std::fmax(std::fabs(b0), std::fabs(b1)));
  a0 -= ea; a1 += ea; b0 -= eb; b1 += eb;
  if (!(a0 <= a1 && b0 <= b1)) return false;
  for (size_t k = 0; k < nt; ++k) {
    const LightChart& c = chart[3 * k + (size_t)m];
    if (c.state == 2) { blockers.push_back((int32_t)k); continue; }
    if ((c.state == 0) != (sg > 0)) continue;
    if (c.a1 >= a0 && c.a0 <= a1 && c.b1 >= b0 && c.b0 <= b1 && c.near <= far) blockers.push_back((int32_t)k);
  }
  return true;
}
/* what is_shadowed's ray can stray from the hull by: several EPSILON plus rounding at the scale of the scene */
double shadow_pad() {
  double reach = 1.0;
  for (double v : cur->mesh_world) reach = std::fmax(reach, std::fabs(v));
  for (int c = 0; c < 3; ++c) reach = std::fmax(reach, std::fmax(std::fabs(cur->host_light[c]), std::fabs(cur->host_camera.inv_view[4 * c + 3])));
  return 8.0 * RT_EPSILON + 1e-9 * reach;
}
void shadow_tile_lists(int tiles_x, int tiles_y, const std::vector<int>& y_lo, const std::vector<int>& y_hi,
                       const std::vector<std::vector<int32_t>>& cam_lists, double pad, std::vector<std::vector<int32_t>>& out,
                       std::vector<unsigned char>& ok) {
  out.assign((size_t)tiles_x * tiles_y, std::vector<int32_t>());
  ok.assign((size_t)tiles_x * tiles_y, 0);
  std::vector<LightChart> chart;
  light_charts(pad, chart);
  std::vector<V3> pts;
  for (int by = 0; by < tiles_y; by += kListShadowBlock)
    for (int bx = 0; bx < tiles_x; bx += kListShadowBlock) {
      pts.clear();
      for (int ty = by; ty < std::min(tiles_y, by + kListShadowBlock); ++ty)
        for (int tx = bx; tx < std::min(tiles_x, bx + kListShadowBlock); ++tx) {
          const std::vector<int32_t>& seen = cam_lists[(size_t)ty * tiles_x + tx];
          if (seen.empty()) continue;
          /* the tile's pyramid of rays: the directions through the corners of its (padded) pixel rectangle */
          clip_to_pixel_pyramid(tx * (double)RT_TILE_W - 2.0, (tx + 1) * (double)RT_TILE_W + 2.0, (double)y_lo[ty] - 2.0,
                                (double)y_hi[ty] + 3.0, seen, pts);
        }
      std::vector<int32_t> blockers;
      const bool known = blockers_of_points(pts, chart, pad, blockers);
      for (int ty = by; ty < std::min(tiles_y, by + kListShadowBlock); ++ty)
        for (int tx = bx; tx < std::min(tiles_x, bx + kListShadowBlock); ++tx) {
          const size_t tile = (size_t)ty * tiles_x + tx;
          ok[tile] = known ? 1 : 0;
          if (known && !cam_lists[tile].empty()) out[tile] = blockers;
        }
    }
}

/* Behind the `entries` entries of `list`: four 64-bit words per entry, as for the masks, of which the first says where
 * the tile's triangles start in the arrays that follow (low half) and how many they are (high half; all ones: the
 * tile has more than max_per_tile and gets no list); then, from a 64-byte boundary on, copies of those triangles'
 * single-precision forms, tile after tile in the order of their first entry (the parts of a split tile share one
 * copy), so that a tile's camera rays read consecutive memory; then the numbers of the same triangles; then, for each,
 * the numerator of a camera ray's t (host_tri_k); then their positions in the world's object list.
 * false: nothing appended (no triangles at all, or more than kMaxListEntries forms). */
bool append_tile_lists(std::vector<int32_t>& list, int entries, int heavy, const std::vector<std::vector<int32_t>>& tile_tris,
                       int max_per_tile, Context::TilePlan* plan, int tiles_x) {
  std::vector<long long> start(tile_tris.size(), -1);
  size_t total = 0;
  /* A part of a tile split into four or more is one rectangle of the tile (rt_part_rect): it gets the triangles of
   * the tile's list that its own camera rays can reach, and, if they are few, the triangles that can stand between
   * what those rays see and the light (a walk leaves out more than the list of a whole tile's shadow rays does, but a
   * few pixels see little, and a short list in LDS costs less than a walk's chain of dependent steps). */
  typedef Context::TilePlan::PartLists PartLists;
  std::vector<PartLists> parts;
  std::vector<int32_t> part_of((size_t)entries, -1);
  const bool by_part = plan->part_lists && tiles_x > 0 && cur->tri_xy.size() == 6 * (size_t)cur->scene.n_tris &&
                       cur->mesh_world.size() == 9 * (size_t)cur->scene.n_tris;
  std::vector<LightChart> chart;
  std::vector<V3> pts;
  const double pad = by_part ? shadow_pad() : 0.0;
  for (int e = 0; e < entries; ++e) {
    const size_t tile = (size_t)(list[e] & RT_TILE_MASK);
    if (tile >= tile_tris.size()) return false;
    const int lg = RT_ENTRY_LG(list[e]);
    if (by_part && e < heavy && lg >= cur->part_min_lg && !tile_tris[tile].empty()) {
      int rx, ry, rw, rh;
      rt_part_rect(lg, RT_ENTRY_PART(list[e]), rx, ry, rw, rh);
      const int r0 = (int)(tile / (size_t)tiles_x) * RT_TILE_H + ry, r1 = std::min(plan->rows, r0 + rh) - 1;
      auto cached = plan->part_cache.find(list[e]);
      if (cached == plan->part_cache.end()) {
        PartLists fresh;
        fresh.own = true;
        if (r1 >= r0) {
          const double x0 = (double)((int)(tile % (size_t)tiles_x) * RT_TILE_W + rx), x1 = x0 + rw;
          const double y0 = (double)rt_hip_rank_row_to_y(r0, plan->band_rows, plan->n_ranks, plan->rank);
          const double y1 = (double)rt_hip_rank_row_to_y(r1, plan->band_rows, plan->n_ranks, plan->rank) + 1.0;
          for (int32_t k : tile_tris[tile])
            if (tri_reaches_rect((size_t)k, x0, x1, y0, y1)) fresh.cam.push_back(k);
          fresh.own = fresh.cam.size() <= (size_t)max_per_tile;
          if (fresh.own && plan->part_shadow_max > 0 && !fresh.cam.empty()) {
            if (chart.empty()) light_charts(pad, chart);
            pts.clear();
            clip_to_pixel_pyramid(x0 - 2.0, x1 + 2.0, y0 - 2.0, y1 + 2.0, fresh.cam, pts);
            fresh.has_shadow = blockers_of_points(pts, chart, pad, fresh.shadow) && fresh.shadow.size() <= (size_t)plan->part_shadow_max;
            if (!fresh.has_shadow) fresh.shadow.clear();
          }
          if (!fresh.own) fresh.cam.clear();
        }
        cached = plan->part_cache.emplace(list[e], std::move(fresh)).first;
      }
      PartLists pl = cached->second;
      const bool own = pl.own;
      if (own && total + pl.cam.size() + pl.shadow.size() <= kMaxListEntries / 2) {
        pl.start = (long long)total;
        total += pl.cam.size();
        if (pl.has_shadow) { pl.sstart = (long long)total; total += pl.shadow.size(); }
        part_of[(size_t)e] = (int32_t)parts.size();
        parts.push_back(std::move(pl));
        continue;
      }
    }
    if (start[tile] >= 0 || tile_tris[tile].size() > (size_t)max_per_tile) continue;
    start[tile] = (long long)total;
    total += tile_tris[tile].size();
  }
  plan->mask_offset = 0;
  plan->list_forms = 0;
  plan->list_total = 0;
  plan->list_max = max_per_tile;
  plan->list_longest = 0;
  plan->shadow_longest = 0;
  if (total == 0 || total > kMaxListEntries) return false;
  for (int e = 0; e < heavy && e < entries; ++e) {
    const size_t tile = (size_t)(list[e] & RT_TILE_MASK);
    if (part_of[(size_t)e] >= 0) {
      const PartLists& pl = parts[(size_t)part_of[(size_t)e]];
      plan->list_longest = std::max(plan->list_longest, (int)pl.cam.size());
      plan->shadow_longest = std::max(plan->shadow_longest, (int)pl.shadow.size());
    } else if (start[tile] >= 0) {
      plan->list_longest = std::max(plan->list_longest, (int)tile_tris[tile].size());
    }
  }
  /* the shadow rays' lists of whole tiles: one per block of tiles (the tiles of a block share it), behind the camera rays' */
  const bool shadows = plan->tile_shadow.size() == tile_tris.size() && plan->tile_shadow_ok.size() == tile_tris.size() && tiles_x > 0;
  const int bx_n = (tiles_x + kListShadowBlock - 1) / kListShadowBlock;
  std::vector<long long> shadow_start_of_block;
  std::vector<long long> shadow_start(tile_tris.size(), -1);
  if (shadows) {
    const int tiles_y = (int)(tile_tris.size() / (size_t)tiles_x);
    shadow_start_of_block.assign((size_t)bx_n * ((tiles_y + kListShadowBlock - 1) / kListShadowBlock + 1), -1);
    for (int e = 0; e < entries; ++e) {
      const size_t tile = (size_t)(list[e] & RT_TILE_MASK);
      if (part_of[(size_t)e] >= 0 || shadow_start[tile] >= 0 || !plan->tile_shadow_ok[tile] || tile_tris[tile].empty()) continue;
      const std::vector<int32_t>& sl = plan->tile_shadow[tile];
      if (sl.size() > (size_t)plan->shadow_max || total + sl.size() > kMaxListEntries) continue;
      const size_t blk = (size_t)((int)(tile / (size_t)tiles_x) / kListShadowBlock) * bx_n + (size_t)((int)(tile % (size_t)tiles_x) / kListShadowBlock);
      if (shadow_start_of_block[blk] < 0) {
        shadow_start_of_block[blk] = (long long)total;
        total += sl.size();
      }
      shadow_start[tile] = shadow_start_of_block[blk];
      if (e < heavy) plan->shadow_longest = std::max(plan->shadow_longest, (int)sl.size());
    }
  }
  if (getenv("RT_HIP_VERBOSE")) {
    int longest_any = 0, n_over[5] = {0, 0, 0, 0, 0}, with_shadow = 0, n_parts = 0, parts_with_shadow = 0;
    size_t shadow_sum = 0, part_sum = 0, part_shadow_sum = 0;
    for (int e = 0; e < heavy && e < entries; ++e) {
      const size_t tile = (size_t)(list[e] & RT_TILE_MASK);
      if (part_of[(size_t)e] >= 0) {
        const PartLists& pl = parts[(size_t)part_of[(size_t)e]];
        n_parts++;
        part_sum += pl.cam.size();
        if (pl.has_shadow) { parts_with_shadow++; part_shadow_sum += pl.shadow.size(); }
        continue;
      }
      const int n = (int)tile_tris[tile].size();
      longest_any = std::max(longest_any, n);
      n_over[n > 512 ? 4 : (n > 256 ? 3 : (n > 128 ? 2 : (n > 64 ? 1 : 0)))]++;
      if (shadows && shadow_start[tile] >= 0) { with_shadow++; shadow_sum += plan->tile_shadow[tile].size(); }
    }
    fprintf(stderr, "tile lists: %zu forms, longest list in use %d (of any mesh tile: %d), mesh entries with <=64/128/256/512/more: %d %d %d %d %d; "
                    "shadow lists for %d mesh entries, mean %.1f, longest %d; %d parts with lists of their own (mean %.1f), %d of them with shadow lists (mean %.1f)\n",
            total, plan->list_longest, longest_any, n_over[0], n_over[1], n_over[2], n_over[3], n_over[4], with_shadow,
            with_shadow ? (double)shadow_sum / with_shadow : 0.0, plan->shadow_longest, n_parts, n_parts ? (double)part_sum / n_parts : 0.0,
            parts_with_shadow, parts_with_shadow ? (double)part_shadow_sum / parts_with_shadow : 0.0);
  }
  list.resize((size_t)entries);
  if (list.size() % 2) list.push_back(0);
  plan->mask_offset = list.size();
  for (int e = 0; e < entries; ++e) {
    const size_t tile = (size_t)(list[e] & RT_TILE_MASK);
    if (part_of[(size_t)e] >= 0) {
      const PartLists& pl = parts[(size_t)part_of[(size_t)e]];
      list.push_back((int32_t)(uint32_t)pl.start);
      list.push_back((int32_t)pl.cam.size());
      list.push_back(pl.has_shadow ? (int32_t)(uint32_t)pl.sstart : 0);
      list.push_back(pl.has_shadow ? (int32_t)pl.shadow.size() : -1);
    } else {
      const bool has = start[tile] >= 0;
      list.push_back(has ? (int32_t)(uint32_t)start[tile] : 0);
      list.push_back(has ? (int32_t)tile_tris[tile].size() : -1);
      const bool hs = shadows && shadow_start[tile] >= 0;
      list.push_back(hs ? (int32_t)(uint32_t)shadow_start[tile] : 0);
      list.push_back(hs ? (int32_t)plan->tile_shadow[tile].size() : -1);
    }
    for (int k = 0; k < 4; ++k) list.push_back(0);
  }
  while (list.size() % 16) list.push_back(0);
  plan->list_forms = list.size();
  plan->list_total = total;
  const size_t words = sizeof(DevTriF) / 4;
  list.resize(plan->list_forms + total * words + 3 * total, 0);
  auto put = [&](size_t at, int32_t tri, bool camera) {
    memcpy(&list[plan->list_forms + at * words], &cur->host_trif[(size_t)tri], sizeof(DevTriF));
    list[plan->list_forms + total * words + at] = tri;
    if (camera) {
      memcpy(&list[plan->list_forms + total * words + total + at], &cur->host_tri_k[(size_t)tri], sizeof(float));
      list[plan->list_forms + total * words + 2 * total + at] = cur->host_tri_order[(size_t)tri];
    }
  };
  for (size_t tile = 0; tile < tile_tris.size(); ++tile) {
    if (start[tile] >= 0) {
      const std::vector<int32_t>& l = tile_tris[tile];
      for (size_t k = 0; k < l.size(); ++k) put((size_t)start[tile] + k, l[k], true);
    }
    if (shadows && shadow_start[tile] >= 0) { /* the tiles of a block write the same entries */
      const std::vector<int32_t>& l = plan->tile_shadow[tile];
      for (size_t k = 0; k < l.size(); ++k) put((size_t)shadow_start[tile] + k, l[k], false);
    }
  }
  for (const PartLists& pl : parts) {
    for (size_t k = 0; k < pl.cam.size(); ++k) put((size_t)pl.start + k, pl.cam[k], true);
    for (size_t k = 0; k < pl.shadow.size(); ++k) put((size_t)pl.sstart + k, pl.shadow[k], false);
  }
  return true;
}

/* Workgroups of a launch with `plan`: one per mesh entry, then the light tiles, light_per_block to a workgroup. */
int plan_blocks(const Context::TilePlan* plan) {
  return plan->heavy + (plan->entries - plan->heavy + plan->light_per_block - 1) / plan->light_per_block;
}

/* The tile plan of a world with spheres (at most 64, and at most 64 triangles, no BVH in use): tiles in raster
 * order, those that no triangle and no sphere's box reaches last (they show nothing), and four masks per entry:
 * the triangles and the spheres a camera ray of the tile can hit, and the triangles and the spheres that can
 * stand between such a hit and the light (shadow_blockers).  false: no such plan for this world. */
bool masked_world(bool bvh) {
  const int ns = cur->scene.n_spheres, nt = cur->scene.n_tris;
  return cur->tile_plan && cur->tile_masks && cur->sphere_masks && !bvh && ns >= 1 && ns <= 64 && nt <= 64 && !cur->sph_world.empty() &&
         !cur->sph_xy.empty() && (nt == 0 || (!cur->tri_xy.empty() && !cur->tri_world.empty()));
}
int plan_masked_world(const rt_render_params* p, int rows, bool reflect, bool may_plan, hipStream_t s, RenderArgs& a, int& n_blocks) {
  const int W = cur->scene.hsize, ns = cur->scene.n_spheres, nt = cur->scene.n_tris;
  const int tiles_x = (W + RT_TILE_W - 1) / RT_TILE_W, tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
  const long long key[6] = {(long long)cur->scene_serial, rows, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank,
                            (1ll << 20) + cur->light_per_block * 8 + (reflect ? 4 : 0)};
  bool fresh = false;
  Context::TilePlan* const plan = find_plan(key, fresh);
  if (fresh && !may_plan) return RT_ERR_INVALID; /* the caller launches in raster order, without masks */
  if (fresh) {
    std::vector<int> y_lo((size_t)tiles_y, 1 << 30), y_hi((size_t)tiles_y, -1);
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int r = ty * RT_TILE_H; r < (ty + 1) * RT_TILE_H && r < rows; ++r) {
        const int y = rt_hip_rank_row_to_y(r, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank);
        y_lo[ty] = std::min(y_lo[ty], y);
        y_hi[ty] = std::max(y_hi[ty], y);
      }
    std::vector<unsigned char> cover;
    std::vector<unsigned long long> tri_mask((size_t)tiles_x * tiles_y, 0ull), sph_mask((size_t)tiles_x * tiles_y, 0ull);
    if (nt > 0 && !covered_tiles(tiles_x, tiles_y, y_lo, y_hi, p->n_ranks <= 1, cover, &tri_mask)) return RT_ERR_INVALID;
    const double margin = 2.0;
    for (int j = 0; j < ns; ++j) { /* the pixel rectangle of the box of sphere j */
      const double* q = &cur->sph_xy[16 * (size_t)j];
      double bx0 = 1e300, by0 = 1e300, bx1 = -1e300, by1 = -1e300;
      bool whole = false;
      for (int v = 0; v < 8; ++v) {
        whole = whole || !(std::isfinite(q[2 * v]) && std::isfinite(q[2 * v + 1]));
        bx0 = std::fmin(bx0, q[2 * v]); bx1 = std::fmax(bx1, q[2 * v]);
        by0 = std::fmin(by0, q[2 * v + 1]); by1 = std::fmax(by1, q[2 * v + 1]);
      }
      if (whole) { /* a box that reaches beside or behind the camera: any pixel may see it */
        bx0 = by0 = -1e300;
        bx1 = by1 = 1e300;
      }
      bx0 -= margin; by0 -= margin; bx1 += margin; by1 += margin;
      const int tx0 = (int)std::fmax(0.0, std::floor(bx0 / RT_TILE_W)), tx1 = (int)std::fmin((double)tiles_x - 1, std::floor(bx1 / RT_TILE_W));
      for (int ty = 0; ty < tiles_y; ++ty) {
        if ((double)y_hi[ty] + 1.0 < by0 || (double)y_lo[ty] > by1) continue;
        for (int tx = tx0; tx <= tx1; ++tx) sph_mask[(size_t)ty * tiles_x + tx] |= 1ull << j;
      }
    }
    std::vector<int32_t> list, light;
    for (int t = 0; t < tiles_x * tiles_y; ++t) (tri_mask[t] | sph_mask[t] ? list : light).push_back(t);
    plan->heavy = (int)list.size();
    plan->tiles = tiles_x * tiles_y;
    plan->light_background = !light.empty();
    plan->light_per_block = cur->light_per_block;
    plan->refine_left = 0;
    plan->host_list.clear();
    list.insert(list.end(), light.begin(), light.end());
    plan->entries = (int)list.size();
    if (list.size() % 2) list.push_back(0);
    plan->mask_offset = list.size();
    double reach = 1.0;
    for (double v : cur->tri_world) reach = std::fmax(reach, std::fabs(v));
    for (double v : cur->sph_world) reach = std::fmax(reach, std::fabs(v));
    for (int c = 0; c < 3; ++c) reach = std::fmax(reach, std::fmax(std::fabs(cur->host_light[c]), std::fabs(cur->host_camera.inv_view[4 * c + 3])));
    const double pad = 8.0 * RT_EPSILON + 1e-9 * reach;
    const unsigned long long all_tri = nt >= 64 ? ~0ull : (1ull << nt) - 1ull, all_sph = ns >= 64 ? ~0ull : (1ull << ns) - 1ull;
    /* the shadow masks are worked out for blocks of kShadowBlock x kShadowBlock tiles (what any tile of the block
     * sees, the pyramid of the whole block): a sixteenth of the work for masks that are hardly any wider */
    const int bx_n = (tiles_x + kShadowBlock - 1) / kShadowBlock, by_n = (tiles_y + kShadowBlock - 1) / kShadowBlock;
    std::vector<unsigned long long> block_tri((size_t)bx_n * by_n, all_tri), block_sph((size_t)bx_n * by_n, all_sph);
    if (cur->shadow_masks)
      for (int by = 0; by < by_n; ++by)
        for (int bx = 0; bx < bx_n; ++bx) {
          const int tx0 = bx * kShadowBlock, tx1 = std::min(tiles_x, tx0 + kShadowBlock), ty0 = by * kShadowBlock, ty1 = std::min(tiles_y, ty0 + kShadowBlock);
          unsigned long long seen_tri = 0ull, seen_sph = 0ull;
          int ylo = 1 << 30, yhi = -1;
          for (int ty = ty0; ty < ty1; ++ty) {
            ylo = std::min(ylo, y_lo[ty]);
            yhi = std::max(yhi, y_hi[ty]);
            for (int tx = tx0; tx < tx1; ++tx) { seen_tri |= tri_mask[(size_t)ty * tiles_x + tx]; seen_sph |= sph_mask[(size_t)ty * tiles_x + tx]; }
          }
          unsigned long long bt = 0ull, bs = 0ull;
          shadow_blockers(tx0 * (double)RT_TILE_W - margin, (double)ylo - margin, tx1 * (double)RT_TILE_W + margin, (double)yhi + 1.0 + margin, seen_tri,
                          seen_sph, pad, bt, bs);
          block_tri[(size_t)by * bx_n + bx] = bt;
          block_sph[(size_t)by * bx_n + bx] = bs;
        }
    for (int e = 0; e < plan->entries; ++e) {
      const int tile = list[e], tx = tile % tiles_x, ty = tile / tiles_x;
      const size_t blk = (size_t)(ty / kShadowBlock) * bx_n + tx / kShadowBlock;
      const unsigned long long m[4] = {tri_mask[tile], block_tri[blk], sph_mask[tile], block_sph[blk]};
      for (int k = 0; k < 4; ++k) {
        list.push_back((int32_t)(m[k] & 0xffffffffull));
        list.push_back((int32_t)(m[k] >> 32));
      }
    }
    const int src = store_plan(plan, key, list, s);
    if (src != RT_OK) return src;
  }
  cur->launch_plan = plan;
  cur->launch_tile_list = plan->d_list;
  cur->launch_tile_mask = reinterpret_cast<const unsigned long long*>(plan->d_list + plan->mask_offset);
  a.n_heavy = plan->heavy;
  a.light_per_block = plan->light_per_block;
  a.n_entries = plan->entries;
  a.light_is_background = plan->light_background ? 1 : 0;
  a.list_forms = 0;
  a.list_total = 0;
  a.list_lds = 0;
  a.slist_lds = 0;
  n_blocks = plan_blocks(plan);
  return RT_OK;
}

/* Tile order for a launch: tiles that can see the mesh first (one per workgroup), the rest after (several
 * per workgroup).  Kept until the scene or the band layout changes.
 * may_plan == false (an entry point that has promised only to enqueue): a plan that exists is used, a plan that
 * does not is not made (making one allocates, copies and waits for the copy); the launch then takes its tiles in
 * raster order, which changes how long it runs and nothing else. */
int plan_tiles(const rt_render_params* p, int rows, bool bvh, bool reflect, bool overlap, bool may_plan, hipStream_t s, RenderArgs& a, int& n_blocks) {
  const int W = cur->scene.hsize;
  const int tiles_x = (W + RT_TILE_W - 1) / RT_TILE_W, tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
  const int n_tiles = tiles_x * tiles_y;
  cur->launch_tile_list = nullptr;
  cur->launch_tile_mask = nullptr;
  cur->launch_plan = nullptr;
  a.n_heavy = n_tiles;
  a.light_per_block = 1;
  a.n_entries = 0;
  a.light_is_background = 0;
  a.list_forms = 0;
  a.list_total = 0;
  a.list_lds = 0;
  a.slist_lds = 0;
  n_blocks = n_tiles;
  double x0, y0, x1, y1;
  if (n_tiles > RT_TILE_MASK) return RT_OK; /* an entry of a tile order has 22 bits for the tile: larger launches go in raster order */
  if (masked_world(bvh) && cur->light_per_block > 1) {
    const int mrc = plan_masked_world(p, rows, reflect, may_plan, s, a, n_blocks);
    if (mrc != RT_ERR_INVALID) return mrc; /* RT_ERR_INVALID: not this way after all */
    cur->launch_tile_list = nullptr;
    cur->launch_tile_mask = nullptr;
    cur->launch_plan = nullptr;
  }
  /* with spheres in the world any pixel may be expensive; without triangles there is nothing to sort by */
  if (!cur->tile_plan || cur->scene.n_tris == 0 || cur->light_per_block <= 1) return RT_OK;
  const bool tree_in_memory = cur->scene.n_bvh_nodes > kMaxLdsBvhNodes;
  const bool by_xcd_wanted = cur->xcd_tiles == 1 || (cur->xcd_tiles < 0 && tree_in_memory);
  const bool have_rect = cur->scene.n_spheres == 0 && mesh_pixel_rect(cur->host_camera, x0, y0, x1, y1);
  /* no rectangle (the camera is inside the bounds, say): every tile counts as a mesh tile */
  if (!have_rect && !by_xcd_wanted && (cur->tri_rect.empty() || !cur->crowded_first)) return RT_OK;
  const long long key[6] = {(long long)cur->scene_serial, rows, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank,
                            cur->light_per_block * 8 + (bvh ? 1 : 0) + (overlap ? 2 : 0) + (reflect ? 4 : 0)};
  bool fresh = false;
  Context::TilePlan* const plan = find_plan(key, fresh);
  if (fresh && !may_plan) return RT_OK;
  if (fresh) {
    std::vector<int32_t> heavy, light;
    /* image rows covered by every tile row (bands make packed rows non-contiguous in y) */
    std::vector<int> y_lo((size_t)tiles_y, 1 << 30), y_hi((size_t)tiles_y, -1);
    for (int ty = 0; ty < tiles_y; ++ty)
      for (int r = ty * RT_TILE_H; r < (ty + 1) * RT_TILE_H && r < rows; ++r) {
        const int y = rt_hip_rank_row_to_y(r, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank);
        y_lo[ty] = std::min(y_lo[ty], y);
        y_hi[ty] = std::max(y_hi[ty], y);
      }
    /* the tiles the triangles cover, or (unknown) the tiles of the pixel rectangle of the mesh's bounds */
    std::vector<unsigned char> cover;
    std::vector<unsigned long long> masks;
    const bool want_masks = !bvh && cur->tile_masks && cur->scene.n_tris <= 64;
    std::vector<std::vector<int32_t>> tri_lists;
    bool want_lists = cur->tile_lists && (bvh ? cur->list_max_walk > 0 || cur->list_max_walk_few > 0 : cur->scene.n_tris > 64) &&
                      cur->host_trif.size() == (size_t)cur->scene.n_tris && cur->host_tri_k.size() == cur->host_trif.size() &&
                      cur->host_tri_order.size() == cur->host_trif.size();
    const bool have_cover = have_rect && cur->cover_tiles &&
                            covered_tiles(tiles_x, tiles_y, y_lo, y_hi, p->n_ranks <= 1, cover, want_masks ? &masks : nullptr,
                                          want_lists ? &tri_lists : nullptr);
    for (int ty = 0; ty < tiles_y; ++ty) {
      const bool row_hit = !have_rect || (y_hi[ty] >= y0 && y_lo[ty] <= y1);
      for (int tx = 0; tx < tiles_x; ++tx) {
        bool hit = row_hit && (!have_rect || ((tx + 1) * RT_TILE_W - 1 >= x0 && tx * RT_TILE_W <= x1));
        if (have_cover) hit = hit && cover[(size_t)ty * tiles_x + tx] != 0;
        (hit ? heavy : light).push_back(ty * tiles_x + tx);
      }
    }
    /* Longest first: the tiles that many triangles project into hold the waves that run longest, and a frame
     * ends when its last wave does.  Count, per 16 x 16 cell of the image, the triangles whose pixel rectangle
     * touches it; the mesh tiles start in descending order of that count. */
    std::vector<int> cost, cost_of_tile; /* by position in `heavy` (raster order); by tile number */
    if (cur->crowded_first && !cur->tri_rect.empty() && heavy.size() > 1) {
      const int cells_y = (cur->scene.vsize + RT_TILE_H - 1) / RT_TILE_H;
      std::vector<int> crowd((size_t)tiles_x * cells_y, 0);
      const size_t n_rect = cur->tri_rect.size() / 4;
      for (size_t k = 0; k < n_rect; ++k) {
        const float* q = &cur->tri_rect[4 * k];
        if (!(q[2] >= 0.f && q[3] >= 0.f && q[0] < (float)W && q[1] < (float)cur->scene.vsize)) continue;
        const int cx0 = std::max(0, (int)(q[0] / RT_TILE_W)), cx1 = std::min(tiles_x - 1, (int)(q[2] / RT_TILE_W));
        const int cy0 = std::max(0, (int)(q[1] / RT_TILE_H)), cy1 = std::min(cells_y - 1, (int)(q[3] / RT_TILE_H));
        if ((long long)(cx1 - cx0 + 1) * (cy1 - cy0 + 1) > 4096) continue; /* a triangle across the whole image says nothing */
        for (int cy = cy0; cy <= cy1; ++cy)
          for (int cx = cx0; cx <= cx1; ++cx) crowd[(size_t)cy * tiles_x + cx]++;
      }
      cost.resize(heavy.size());
      for (size_t i = 0; i < heavy.size(); ++i) {
        const int tx = heavy[i] % tiles_x, ty = heavy[i] / tiles_x;
        const int y = rt_hip_rank_row_to_y(ty * RT_TILE_H, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank);
        const int cy = std::min(cells_y - 1, std::max(0, y / RT_TILE_H));
        cost[i] = crowd[(size_t)cy * tiles_x + tx];
      }
      cost_of_tile.assign((size_t)n_tiles, 0);
      for (size_t i = 0; i < heavy.size(); ++i) cost_of_tile[heavy[i]] = cost[i];
    }
    /* positions [from, to) of `heavy` (still in raster order) by descending cost; equally crowded tiles: nearer
     * the middle of the mesh's pixel rectangle first (more of their pixels are covered) */
    const double ccx = 0.5 * (x0 + x1), ccy = 0.5 * (y0 + y1);
    auto by_cost = [&](size_t from, size_t to) {
      std::vector<int32_t> order(to - from);
      for (size_t i = from; i < to; ++i) order[i - from] = (int32_t)i;
      if (!cost.empty()) {
        auto dist2 = [&](int32_t i) {
          const double dx = (heavy[i] % tiles_x + 0.5) * RT_TILE_W - ccx, dy = (heavy[i] / tiles_x + 0.5) * RT_TILE_H - ccy;
          return dx * dx + dy * dy;
        };
        std::stable_sort(order.begin(), order.end(), [&](int32_t a_, int32_t b_) {
          if (cost[a_] != cost[b_]) return cost[a_] > cost[b_];
          return have_rect ? dist2(a_) < dist2(b_) : false;
        });
      }
      for (int32_t& i : order) i = heavy[i];
      return order;
    };
    if (by_xcd_wanted && heavy.size() >= 64) {
      /* Workgroups b and b + 8 run on the same XCD and share its L2: give each XCD one contiguous run of the
       * mesh tiles instead of every eighth, so that what its L2 holds of the tree is what its tiles walk
       * (81,920 triangles -10 %, 327,680 -18 %, 1.31 M -9 %; a tree that sits in LDS gains nothing, +5 %).
       * Inside its run every XCD starts with the most crowded tiles. */
      const size_t nh = heavy.size(), per = (nh + 7) / 8;
      std::vector<int32_t> run[8];
      for (size_t x = 0; x < 8; ++x) run[x] = by_cost(std::min(nh, x * per), std::min(nh, (x + 1) * per));
      std::vector<int32_t> by_xcd;
      by_xcd.reserve(nh);
      for (size_t b = 0; b < per * 8; ++b)
        if (b / 8 < run[b % 8].size()) by_xcd.push_back(run[b % 8][b / 8]);
      heavy.swap(by_xcd);
    } else if (!cost.empty()) {
      std::vector<int32_t> sorted = by_cost(0, heavy.size());
      heavy.swap(sorted);
    }
    /* Split the most crowded tiles (BVH walks only: in the exhaustive loop every wave does the same work
     * whatever its lanes).  A launch whose mesh tiles are resident within two rounds of workgroups lasts as
     * long as its slowest wave, and a wave with fewer pixels walks less: 1080p lighthouse 65.9 -> 61.0 us,
     * 720p 82.7 -> 76.8 us, reflective 203 -> 192 us.  A launch with more tiles than that is bound by
     * throughput and pays for the idle lanes (4K lighthouse 150 -> 163 us, with mirror bounces 223 -> 285 us,
     * 1440p 219 -> 241 us): no split there.  Four parts for more
     * than a few tiles push the crowded tiles into a second round (1080p: 79 us).  Frames that overlap on
     * several streams fill the chip between them and are bound by throughput as well (lighthouse over 4
     * streams: 39.7 us per frame whole, 48.0 split): no split while the caller alternates streams.  Tiles that
     * fewer than 16 triangles project into are left whole: their walks are short, and halving them only costs
     * lanes (lighthouse with a floor of 8, 16, 24: 61.0, 61.5, 61.2 us; reflective 192, 195, 204 us). */
    if (bvh && cur->split_parts > 1 && cur->split_tiles > 0 && !cost.empty() && !(by_xcd_wanted && heavy.size() >= 64) &&
        n_tiles <= RT_TILE_MASK && heavy.size() <= (size_t)cur->split_max_per_cu * cur->n_cu && !overlap) {
      std::vector<int> sorted_cost(cost);
      std::sort(sorted_cost.begin(), sorted_cost.end(), std::greater<int>());
      const int floor_cost = std::max(cur->split_min_cost, sorted_cost[std::min(sorted_cost.size() - 1, (size_t)cur->split_tiles - 1)]);
      /* the very first of them in four parts: few enough to stay in the first round (1080p lighthouse 61.9 ->
       * 58.9 us, 720p 77.5 -> 69.0 us with 8; from 64 on the gain is gone).  With mirror bounces a wave lasts
       * several times longer and the extra workgroups weigh less: 128 there (720p / 900p / 1080p: 181 / 220 /
       * 197 us with 8, 176 / 176 / 188 us with 128; whole tiles: 203 / 237 / 208 us). */
      const int quarters = reflect ? cur->split_quarters_reflect : cur->split_quarters;
      const int floor_cost4 = quarters > 0 && cur->split_parts == 2
                                  ? std::max(cur->split_min_cost, sorted_cost[std::min(sorted_cost.size() - 1, (size_t)quarters - 1)])
                                  : 0x7fffffff;
      std::vector<int32_t> split;
      split.reserve(heavy.size() + (size_t)cur->split_tiles * 4);
      for (int32_t tile : heavy) {
        if (cost_of_tile[tile] >= floor_cost) {
          const int parts = cost_of_tile[tile] >= floor_cost4 ? 4 : cur->split_parts;
          for (int part = 0; part < parts; ++part) split.push_back(RT_ENTRY(tile, part, parts == 4 ? 2 : 1));
        } else {
          split.push_back(tile);
        }
      }
      heavy.swap(split);
    }
    plan->heavy = (int)heavy.size();
    plan->tiles = n_tiles;
    plan->light_background = have_rect && !light.empty();
    plan->light_per_block = cur->light_per_block;
    /* the split above goes by how many triangles project into a tile; the first launches with this plan measure
     * how long every wave really runs and split the slowest further (refine_split) */
    const bool was_split = bvh && cur->split_parts > 1 && cur->split_tiles > 0 && !cost.empty() &&
                           !(by_xcd_wanted && heavy.size() >= 64) && n_tiles <= RT_TILE_MASK && !overlap &&
                           heavy.size() <= (size_t)(cur->split_max_per_cu + 2) * cur->n_cu;
    plan->refine_left = was_split ? cur->split_refine : 0;
    heavy.insert(heavy.end(), light.begin(), light.end());
    plan->entries = (int)heavy.size();
    plan->mask_offset = 0;
    plan->list_forms = 0;
    plan->list_total = 0;
    std::vector<int32_t> entries_only;
    if (have_cover && want_lists) {
      entries_only = heavy;
      const bool few_tiles = (size_t)plan->heavy <= (size_t)(cur->split_max_per_cu + 2) * cur->n_cu && !overlap;
      plan->tile_shadow.clear();
      plan->tile_shadow_ok.clear();
      plan->shadow_max = bvh ? cur->walk_shadow_max : 0x7fffffff;
      plan->part_cache.clear();
      plan->part_lists = bvh && cur->part_lists;
      plan->part_shadow_max = cur->shadow_lists ? cur->part_shadow_max : 0;
      plan->rows = rows;
      plan->band_rows = p->n_ranks > 1 ? p->band_rows : 1;
      plan->n_ranks = p->n_ranks;
      plan->rank = p->rank;
      if (cur->shadow_lists && plan->shadow_max > 0 && cur->mesh_world.size() == 9 * (size_t)cur->scene.n_tris) {
        shadow_tile_lists(tiles_x, tiles_y, y_lo, y_hi, tri_lists, shadow_pad(), plan->tile_shadow, plan->tile_shadow_ok);
      }
      if (!append_tile_lists(heavy, plan->entries, plan->heavy, tri_lists, bvh ? (few_tiles ? cur->list_max_walk_few : cur->list_max_walk) : 0x7fffffff, plan, tiles_x))
        want_lists = false;
    }
    if (have_cover && want_lists) {
      plan->tile_tris.swap(tri_lists);
    } else if (have_cover && want_masks) {
      /* behind the entries (8-byte aligned): for every entry the triangles its camera rays have to be tested against */
      if (heavy.size() % 2) heavy.push_back(0);
      plan->mask_offset = heavy.size();
      double reach = 1.0;
      for (double v : cur->tri_world) reach = std::fmax(reach, std::fabs(v));
      for (int c = 0; c < 3; ++c) reach = std::fmax(reach, std::fmax(std::fabs(cur->host_light[c]), std::fabs(cur->host_camera.inv_view[4 * c + 3])));
      const double pad = 8.0 * RT_EPSILON + 1e-9 * reach;
      /* the shadow masks are worked out for blocks of kShadowBlock x kShadowBlock tiles (what any tile of the block
       * sees, the pyramid of the whole block): a sixteenth of the work for masks that are hardly any wider */
      const unsigned long long all = cur->scene.n_tris >= 64 ? ~0ull : (1ull << cur->scene.n_tris) - 1ull;
      const int bx_n = (tiles_x + kShadowBlock - 1) / kShadowBlock, by_n = (tiles_y + kShadowBlock - 1) / kShadowBlock;
      std::vector<unsigned long long> block_mask((size_t)bx_n * by_n, all);
      if (cur->shadow_masks && !cur->tri_world.empty())
        for (int by = 0; by < by_n; ++by)
          for (int bx = 0; bx < bx_n; ++bx) {
            const int tx0 = bx * kShadowBlock, tx1 = std::min(tiles_x, tx0 + kShadowBlock), ty0 = by * kShadowBlock, ty1 = std::min(tiles_y, ty0 + kShadowBlock);
            unsigned long long seen = 0ull, sh = 0ull, none = 0ull;
            int ylo = 1 << 30, yhi = -1;
            for (int ty = ty0; ty < ty1; ++ty) {
              ylo = std::min(ylo, y_lo[ty]);
              yhi = std::max(yhi, y_hi[ty]);
              for (int tx = tx0; tx < tx1; ++tx) seen |= masks[(size_t)ty * tiles_x + tx];
            }
            /* the same two pixels of margin as the rasterisation that made the masks */
            if (seen) shadow_blockers(tx0 * (double)RT_TILE_W - 2.0, (double)ylo - 2.0, tx1 * (double)RT_TILE_W + 2.0, (double)yhi + 3.0, seen, 0ull, pad, sh, none);
            block_mask[(size_t)by * bx_n + bx] = sh;
          }
      for (int e = 0; e < plan->entries; ++e) {
        const int tile = heavy[e] & RT_TILE_MASK, tx = tile % tiles_x, ty = tile / tiles_x;
        const unsigned long long m = masks[(size_t)tile], sh = block_mask[(size_t)(ty / kShadowBlock) * bx_n + tx / kShadowBlock];
        heavy.push_back((int32_t)(m & 0xffffffffull));
        heavy.push_back((int32_t)(m >> 32));
        heavy.push_back((int32_t)(sh & 0xffffffffull));
        heavy.push_back((int32_t)(sh >> 32));
        for (int k = 0; k < 4; ++k) heavy.push_back(0); /* the two masks of the spheres: there are none */
      }
    }
    const int src = store_plan(plan, key, heavy, s);
    if (src != RT_OK) return src;
    if (plan->refine_left > 0) plan->host_list = entries_only.empty() ? heavy : entries_only;
    else plan->host_list.clear();
    if (plan->refine_left <= 0 || !plan->list_forms) {
      plan->part_cache.clear();
      plan->tile_tris.clear(); plan->tile_tris.shrink_to_fit();
      plan->tile_shadow.clear(); plan->tile_shadow.shrink_to_fit();
      plan->tile_shadow_ok.clear();
    }
  }
  cur->launch_plan = plan;
  cur->launch_tile_list = plan->d_list;
  cur->launch_tile_mask = plan->mask_offset ? reinterpret_cast<const unsigned long long*>(plan->d_list + plan->mask_offset) : nullptr;
  a.n_heavy = plan->heavy;
  a.light_per_block = plan->light_per_block;
  a.n_entries = plan->entries;
  a.light_is_background = plan->light_background ? 1 : 0;
  a.list_forms = (int32_t)plan->list_forms;
  a.list_total = (int32_t)plan->list_total;
  a.list_lds = plan->list_forms && cur->list_lds ? (std::min(plan->list_longest, kMaxLdsListForms) + 3) / 4 * 4 : 0;
  a.slist_lds = plan->list_forms && cur->list_lds ? (std::min(plan->shadow_longest, kMaxLdsShadowForms) + 3) / 4 * 4 : 0;
  n_blocks = plan_blocks(plan);
  return RT_OK;
}

/* Rows 0-2 of an inverse transform for the general kernel, and in place of its last row, which has to be
 * exactly 0 0 0 1 (every matrix the reference's constructors and inverse() produce for a shape is), the signed
 * zeros 0.0 * m[3], 0.0 * m[7], 0.0 * m[11] that transform() adds to a transformed vector. */
bool general_rows(double* dst, const double* inv) {
  if (inv[12] != 0.0 || inv[13] != 0.0 || inv[14] != 0.0 || inv[15] != 1.0) return false;
  memcpy(dst, inv, 12 * sizeof(double));
  for (int r = 0; r < 3; ++r) dst[12 + r] = 0.0 * inv[4 * r + 3];
  dst[15] = 1.0;
  return true;
}

/* Largest eigenvalue of the symmetric 3x3 matrix s (closed form, Smith 1961). */
double largest_eigenvalue_sym3(const double s[3][3]) {
  const double p1 = s[0][1] * s[0][1] + s[0][2] * s[0][2] + s[1][2] * s[1][2];
  if (p1 == 0.0) return std::fmax(s[0][0], std::fmax(s[1][1], s[2][2]));
  const double q = (s[0][0] + s[1][1] + s[2][2]) / 3.0;
  const double p2 = (s[0][0] - q) * (s[0][0] - q) + (s[1][1] - q) * (s[1][1] - q) + (s[2][2] - q) * (s[2][2] - q) + 2.0 * p1;
  const double p = std::sqrt(p2 / 6.0);
  double b[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) b[i][j] = (s[i][j] - (i == j ? q : 0.0)) / p;
  const double det = b[0][0] * (b[1][1] * b[2][2] - b[1][2] * b[2][1]) - b[0][1] * (b[1][0] * b[2][2] - b[1][2] * b[2][0]) +
                     b[0][2] * (b[1][0] * b[2][1] - b[1][1] * b[2][0]);
  const double r = std::fmin(1.0, std::fmax(-1.0, det / 2.0));
  return q + 2.0 * p * std::cos(std::acos(r) / 3.0);
}

/* World-space bounding sphere of a shape given inverse(transform): a unit sphere or cube (half-diagonal
 * `reach` = 1 or sqrt 3) or a cylinder segment around `local_centre` on the y axis.  Rounded outwards; radius
 * INFINITY (= never culled) when the transform is too ill-conditioned to invert here. */
void general_bound(GenObject* o, const double* inv, double local_y, double reach) {
  o->bound[0] = o->bound[1] = o->bound[2] = 0.f;
  o->bound[3] = INFINITY;
  o->centre_sq = 0.f;
  const double a[3][3] = {{inv[0], inv[1], inv[2]}, {inv[4], inv[5], inv[6]}, {inv[8], inv[9], inv[10]}};
  const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                     a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
  if (!(std::fabs(det) > 1e-300) || !std::isfinite(det) || !std::isfinite(reach) || !std::isfinite(local_y)) return;
  double f[3][3]; /* the forward linear part: inverse of a */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
      f[i][j] = (a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]) / det;
    }
  /* forward(p) = f * (p - t) with t = the inverse's translation column: inverse(x) = a * x + t */
  const double local[3] = {0.0 - inv[3], local_y - inv[7], 0.0 - inv[11]};
  double centre[3], ftf[3][3];
  for (int i = 0; i < 3; ++i) centre[i] = f[i][0] * local[0] + f[i][1] * local[1] + f[i][2] * local[2];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) ftf[i][j] = f[0][i] * f[0][j] + f[1][i] * f[1][j] + f[2][i] * f[2][j];
  const double stretch = std::sqrt(std::fmax(0.0, largest_eigenvalue_sym3(ftf))) * (1.0 + 1e-6);
  /* check the inversion: a * f must be the identity to well below the margins used here */
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double v = a[i][0] * f[0][j] + a[i][1] * f[1][j] + a[i][2] * f[2][j];
      if (!(std::fabs(v - (i == j ? 1.0 : 0.0)) < 1e-9)) return;
    }
  const double radius = stretch * reach * (1.0 + 1e-6);
  const double r2 = radius * radius * (1.0 + 1e-5);
  double c2 = 0.0;
  for (int i = 0; i < 3; ++i) c2 += centre[i] * centre[i];
  if (!std::isfinite(r2) || !std::isfinite(c2) || r2 > 1e30 || c2 > 1e30) return;
  for (int i = 0; i < 3; ++i) o->bound[i] = (float)centre[i];
  o->bound[3] = std::nextafterf((float)r2, INFINITY);
  o->centre_sq = std::nextafterf((float)c2, INFINITY);
}

/* A world that needs the general kernel: one object array in world-list order, materials with their pattern and
 * transparency halves. */
int upload_general(const rt_scene* sc, bool only_form) {
  const int ns = sc->n_spheres, nt = sc->n_tris, nsh = sc->n_shapes;
  const long long n_ll = (long long)ns + nt + nsh;
  if (nsh < 0 || (nsh > 0 && !sc->shapes)) return RT_ERR_INVALID;
  if (n_ll > (1 << 20)) return RT_ERR_TOO_LARGE; /* every ray visits every object: this path is for small worlds */
  const int n = (int)n_ll;
  if (n > 0 && (sc->n_materials < 1 || !sc->materials)) return RT_ERR_INVALID;
  {
    const double* v = sc->camera.inv_view; /* rays leave the camera as (point, vector) */
    if (v[12] != 0.0 || v[13] != 0.0 || v[14] != 0.0 || v[15] != 1.0) return RT_ERR_UNSUPPORTED;
  }
  std::vector<GenObject> objs((size_t)(n > 0 ? n : 1));
  std::vector<char> seen((size_t)(n > 0 ? n : 1), 0);
  long long max_entries = 0; /* the most intersections one ray can append */
  auto place = [&](int order, int material) -> GenObject* {
    if (order < 0 || order >= n || seen[order]) return nullptr;
    if (material < 0 || material >= sc->n_materials) return nullptr;
    seen[order] = 1;
    GenObject* o = &objs[order];
    memset(o, 0, sizeof *o);
    o->material = material;
    return o;
  };
  for (int i = 0; i < ns; ++i) {
    GenObject* o = place(sc->sph_order[i], sc->sph_material[i]);
    if (!o) return RT_ERR_INVALID;
    o->kind = GEN_SPHERE;
    if (!general_rows(o->m, sc->sph_inv + 16 * (size_t)i)) return RT_ERR_UNSUPPORTED;
    general_bound(o, sc->sph_inv + 16 * (size_t)i, 0.0, 1.0);
    max_entries += 2;
  }
  for (int i = 0; i < nt; ++i) {
    GenObject* o = place(sc->tri_order[i], sc->tri_material[i]);
    if (!o) return RT_ERR_INVALID;
    o->kind = GEN_TRIANGLE;
    o->bound[3] = INFINITY;
    for (int r = 0; r < 12; ++r) o->m[r] = sc->tri_soa[(size_t)r * nt + i];
    max_entries += 1;
  }
  for (int i = 0; i < nsh; ++i) {
    const rt_shape& sh = sc->shapes[i];
    GenObject* o = place(sh.order, sh.material);
    if (!o) return RT_ERR_INVALID;
    if (sh.type == RT_SHAPE_PLANE) { o->kind = GEN_PLANE; max_entries += 1; }
    else if (sh.type == RT_SHAPE_CUBE) { o->kind = GEN_CUBE; max_entries += 2; }
    else if (sh.type == RT_SHAPE_CYLINDER) { o->kind = GEN_CYLINDER; max_entries += 4; }
    else return RT_ERR_UNSUPPORTED;
    if (!general_rows(o->m, sh.inv)) return RT_ERR_UNSUPPORTED;
    o->bound[3] = INFINITY; /* a plane has none */
    if (sh.type == RT_SHAPE_CUBE) general_bound(o, sh.inv, 0.0, std::sqrt(3.0));
    if (sh.type == RT_SHAPE_CYLINDER) { /* the segment minimum..maximum of the unit cylinder, if it is one */
      const double half = (sh.maximum - sh.minimum) / 2.0;
      general_bound(o, sh.inv, (sh.maximum + sh.minimum) / 2.0, std::sqrt(1.0 + half * half));
    }
    o->minimum = sh.minimum;
    o->maximum = sh.maximum;
    o->closed = sh.closed != 0;
  }
  std::vector<GenMaterial> mats((size_t)(sc->n_materials > 0 ? sc->n_materials : 1));
  for (int i = 0; i < sc->n_materials; ++i) {
    GenMaterial& m = mats[i];
    memset(&m, 0, sizeof m);
    const rt_material& a = sc->materials[i];
    memcpy(m.color, a.color, sizeof m.color);
    m.ambient = a.ambient; m.diffuse = a.diffuse; m.specular = a.specular; m.shininess = a.shininess;
    m.reflective = a.reflective;
    m.transparency = 0.0;
    m.refractive_index = 1.0; /* create_material_default, RayTracer.c:713 */
    m.pattern_type = RT_PATTERN_NONE;
    if (sc->materials_ext) {
      const rt_material_ext& x = sc->materials_ext[i];
      if (x.pattern_type < RT_PATTERN_NONE || x.pattern_type > RT_PATTERN_CHECKER) return RT_ERR_INVALID;
      m.transparency = x.transparency;
      m.refractive_index = x.refractive_index;
      memcpy(m.pattern_from, x.pattern_from, sizeof m.pattern_from);
      memcpy(m.pattern_to, x.pattern_to, sizeof m.pattern_to);
      memcpy(m.pattern_inv, x.pattern_inv, sizeof m.pattern_inv);
      m.pattern_type = x.pattern_type;
    }
  }
  GenScene hd;
  memset(&hd, 0, sizeof hd);
  hd.pixel_size = sc->camera.pixel_size;
  hd.half_width = sc->camera.half_width;
  hd.half_height = sc->camera.half_height;
  memcpy(hd.inv_view, sc->camera.inv_view, sizeof hd.inv_view);
  memcpy(hd.light_pos, sc->light.position, sizeof hd.light_pos);
  memcpy(hd.light_int, sc->light.intensity, sizeof hd.light_int);
  memcpy(hd.first_inv, sc->first_object_inv, sizeof hd.first_inv);
  hd.hsize = sc->camera.hsize;
  hd.vsize = sc->camera.vsize;
  hd.n_objects = n;
  for (const GenMaterial& m : mats)
    if (!(m.transparency == 0.0)) hd.any_transparent = 1; /* the test prepare_computations' walk is behind */

  if (only_form) free_scene();
  else free_general();
  HIP_TRY(hipMalloc((void**)&cur->d_gen_scene, sizeof hd));
  HIP_TRY(hipMemcpy(cur->d_gen_scene, &hd, sizeof hd, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&cur->d_gen_objects, sizeof(GenObject) * objs.size()));
  HIP_TRY(hipMemcpy(cur->d_gen_objects, objs.data(), sizeof(GenObject) * objs.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc((void**)&cur->d_gen_materials, sizeof(GenMaterial) * mats.size()));
  HIP_TRY(hipMemcpy(cur->d_gen_materials, mats.data(), sizeof(GenMaterial) * mats.size(), hipMemcpyHostToDevice));
  /* a short list when no ray can fill it; otherwise the reference's 100 slots, overflow behaviour included */
  cur->gen_list_cap = max_entries <= 32 ? 32 : 100;
  if (!only_form) return RT_OK; /* the scene also has its sphere / triangle form; RT_ACCEL_REFERENCE selects this one */
  cur->general = true;
  cur->scene = DevScene{};
  cur->scene.hsize = sc->camera.hsize;
  cur->scene.vsize = sc->camera.vsize;
  cur->host_camera = sc->camera;
  cur->reflective = false;
  cur->have_scene = true;
  ++cur->scene_serial;
  return RT_OK;
}

template <bool COUNT>
void launch_general_cap(dim3 grid, hipStream_t s, const RenderArgs& a, const int32_t* tile_list, int32_t* probe_cost,
                        unsigned char* rgb, double* canvas) {
  const dim3 block(RT_TILE_W * RT_TILE_H);
  if (cur->gen_list_cap <= 32)
    hipLaunchKernelGGL((rt_general_kernel<COUNT, 32>), grid, block, 0, s, (const GenScene*)cur->d_gen_scene,
                       (const GenObject*)cur->d_gen_objects, (const GenMaterial*)cur->d_gen_materials, a, tile_list, probe_cost, rgb,
                       canvas, cur->d_counters);
  else
    hipLaunchKernelGGL((rt_general_kernel<COUNT, 100>), grid, block, 0, s, (const GenScene*)cur->d_gen_scene,
                       (const GenObject*)cur->d_gen_objects, (const GenMaterial*)cur->d_gen_materials, a, tile_list, probe_cost, rgb,
                       canvas, cur->d_counters);
}
void launch_general(bool count, dim3 grid, hipStream_t s, const RenderArgs& a, unsigned char* rgb, double* canvas) {
  if (count) launch_general_cap<true>(grid, s, a, cur->launch_tile_list, nullptr, rgb, canvas);
  else launch_general_cap<false>(grid, s, a, cur->launch_tile_list, nullptr, rgb, canvas);
}

/* Tile order for launches of the general kernel.  What a tile costs there is the size of its pixels' ray trees
 * (mirrors and glass, also on planes that no pixel rectangle bounds), so it is measured: one small launch traces
 * four pixels of every tile and counts their rays, and the tiles then start, block by block, in descending order
 * of that count.
 * Kept in the same plan cache as plan_tiles' lists, keyed by scene, band layout and depth. */
int plan_general_tiles(const rt_render_params* p, int rows, bool may_plan, hipStream_t s, const RenderArgs& a) {
  cur->launch_tile_list = nullptr;
  if (!cur->tile_plan) return RT_OK;
  const int tiles_x = (cur->scene.hsize + RT_TILE_W - 1) / RT_TILE_W, tiles_y = (rows + RT_TILE_H - 1) / RT_TILE_H;
  const long long n_tiles_ll = (long long)tiles_x * tiles_y;
  if (n_tiles_ll < 64 || n_tiles_ll > (1 << 28)) return RT_OK; /* a launch this small has no tail worth ordering */
  const int n_tiles = (int)n_tiles_ll;
  const long long key[6] = {(long long)cur->scene_serial, rows, p->n_ranks > 1 ? p->band_rows : 1, p->n_ranks, p->rank,
                            -2 - (long long)p->max_depth - ((long long)(p->accel == RT_ACCEL_REFERENCE) << 20)};
  bool fresh = false;
  Context::TilePlan* const plan = find_plan(key, fresh);
  if (fresh && !may_plan) return RT_OK; /* raster order */
  if (fresh) {
    int32_t* d_cost = nullptr;
    HIP_TRY(hipMalloc((void**)&d_cost, sizeof(int32_t) * (size_t)n_tiles));
    std::vector<int32_t> cost((size_t)n_tiles), order((size_t)n_tiles);
    RenderArgs pa = a;
    pa.ablate = RT_GEN_PROBE;
    pa.spp_grid = 1;
    pa.sample_off[0] = 0.5;
    hipError_t e = hipMemsetAsync(d_cost, 0, sizeof(int32_t) * (size_t)n_tiles, s);
    if (e == hipSuccess) {
      const unsigned threads = RT_TILE_W * RT_TILE_H;
      launch_general_cap<true>(dim3((unsigned)(((size_t)n_tiles * 4 + threads - 1) / threads)), s, pa, nullptr, d_cost,
                                      nullptr, nullptr);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(cost.data(), d_cost, sizeof(int32_t) * (size_t)n_tiles, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_cost);
    if (!hip_ok(e, "tile cost probe")) return RT_ERR_HIP;
    for (int t = 0; t < n_tiles; ++t) order[t] = t;
    /* whole blocks of 8 x 8 tiles, by the summed estimate, tiles in raster order inside a block: measured better
     * than tile by tile (refraction frame 0.418 ms unordered, 0.394 tile by tile, 0.375 in blocks) */
    const int blk = 8;
    const int bx = (tiles_x + blk - 1) / blk, by = (tiles_y + blk - 1) / blk;
    std::vector<long long> block_cost((size_t)bx * by, 0);
    auto block_of = [&](int32_t t) { return (size_t)((t / tiles_x) / blk) * bx + (size_t)((t % tiles_x) / blk); };
    for (int t = 0; t < n_tiles; ++t) block_cost[block_of(t)] += cost[t];
    std::stable_sort(order.begin(), order.end(), [&](int32_t a_, int32_t b_) {
      const size_t ba = block_of(a_), bb = block_of(b_);
      if (ba == bb) return false;
      if (block_cost[ba] != block_cost[bb]) return block_cost[ba] > block_cost[bb];
      return ba < bb;
    });
    const int src = store_plan(plan, key, order, s);
    if (src != RT_OK) return src;
  }
  cur->launch_tile_list = plan->d_list;
  return RT_OK;
}

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

/* The split of plan_tiles goes by a static estimate (triangles that project into a tile).  A frame whose mesh tiles
 * are all resident at once lasts as long as its slowest wave, so the first launch with a new plan measures instead:
 * every wave of a mesh tile's workgroup reports its cycle count, the entries whose slowest wave runs longer than the
 * launch could take if its work were spread evenly, and longer than split_threshold of the slowest of all, are split
 * into twice as many parts (down to one pixel per wave), and the list is re-ordered longest first by what was
 * measured.  Only the few slowest are split in a round, and there are up to split_refine rounds: a part costs a
 * workgroup with its staging whatever it renders, and halving the pixels of a wave shortens it by far less than half
 * (the wave walks the union of its lanes' paths), so splitting everything that is slower than half the slowest, in
 * three rounds, ended with more work than it saved (1080p lighthouse 40.5 us; 31.4 us this way).  Every candidate
 * list is a real render into the caller's image, timed with events; the list with the shortest launch is the one the
 * plan keeps (a later one has to win by 2 %), and three rounds in a row that do not win end the search. */
int refine_split(Context::TilePlan* plan, bool bvh, bool reflect, hipStream_t s, RenderArgs& a, unsigned char* d_rgb,
                 double* d_canvas, int& n_blocks) {
  int slots = cur->n_cu * 4 * (reflect ? RT_MIN_WAVES_BVH : RT_MIN_WAVES_BVH_PLAIN);
  const int rounds = plan->refine_left;
  plan->refine_left = 0;
  plan->refining = true;
  const int n_light = plan->entries - plan->heavy;
  std::vector<int32_t> list = plan->host_list, best_list = list;
  plan->host_list.clear();
  plan->host_list.shrink_to_fit();
  if (list.empty()) { plan->refining = false; return RT_OK; }
  long long key[6];
  memcpy(key, plan->key, sizeof key);
  hipEvent_t e0, e1;
  if (!hip_ok(hipEventCreate(&e0), "hipEventCreate")) { plan->refining = false; return RT_ERR_HIP; }
  if (!hip_ok(hipEventCreate(&e1), "hipEventCreate")) { (void)hipEventDestroy(e0); plan->refining = false; return RT_ERR_HIP; }
  float best_ms = 1e30f;
  std::vector<unsigned> best_cost; /* the slowest wave of every mesh entry of best_list, as measured with that list */
  bool current_is_best = true;
  int stale = 0; /* rounds in a row that did not win */
  int rc = RT_OK;
  bool with_lists = plan->list_forms != 0 && !plan->tile_tris.empty();
  auto adopt = [&](const std::vector<int32_t>& l) {
    std::vector<int32_t> full;
    if (with_lists) { /* the entries changed: so do the words behind them that say where each entry's triangle list lies */
      full = l;
      if (!append_tile_lists(full, (int)l.size(), (int)l.size() - n_light, plan->tile_tris, plan->list_max, plan, (cur->scene.hsize + RT_TILE_W - 1) / RT_TILE_W))
        return (int)RT_ERR_INVALID;
    }
    const int src = store_plan(plan, key, with_lists ? full : l, s);
    if (src != RT_OK) return src;
    plan->heavy = (int)l.size() - n_light;
    plan->entries = (int)l.size();
    cur->launch_tile_list = plan->d_list;
    cur->launch_tile_mask = plan->mask_offset ? reinterpret_cast<const unsigned long long*>(plan->d_list + plan->mask_offset) : nullptr;
    a.list_forms = (int32_t)plan->list_forms;
    a.list_total = (int32_t)plan->list_total;
    a.list_lds = plan->list_forms && cur->list_lds ? (std::min(plan->list_longest, kMaxLdsListForms) + 3) / 4 * 4 : 0;
    a.slist_lds = plan->list_forms && cur->list_lds ? (std::min(plan->shadow_longest, kMaxLdsShadowForms) + 3) / 4 * 4 : 0;
    a.n_heavy = plan->heavy;
    a.n_entries = plan->entries;
    n_blocks = plan_blocks(plan);
    return (int)RT_OK;
  };
  unsigned* d_probe = nullptr; /* four words per mesh entry; made once, with room for the entries that the rounds add */
  size_t probe_cap = 0;
  for (int round = 0; round <= rounds && rc == RT_OK; ++round) {
    const int nh = (int)list.size() - n_light;
    if ((size_t)nh > probe_cap) {
      if (d_probe) (void)hipFree(d_probe);
      d_probe = nullptr;
      probe_cap = (size_t)nh * 2 + 64;
      if (!hip_ok(hipMalloc((void**)&d_probe, sizeof(unsigned) * 4 * probe_cap), "hipMalloc (split probe)")) { rc = RT_ERR_HIP; break; }
    }
    std::vector<unsigned> cyc((size_t)nh * 4, 0u);
    float ms = 1e30f;
    hipError_t e = hipMemsetAsync(d_probe, 0, sizeof(unsigned) * 4 * (size_t)nh, s);
    for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) { /* the first run warms up, the shortest of the others counts */
      a.probe = 1;
      cur->launch_counters = reinterpret_cast<unsigned long long*>(d_probe);
      e = hipEventRecord(e0, s);
      launch(false, bvh, reflect, dim3((unsigned)n_blocks), s, a, d_rgb, d_canvas);
      cur->launch_counters = nullptr;
      a.probe = 0;
      if (e == hipSuccess) e = hipGetLastError();
      if (e == hipSuccess) e = hipEventRecord(e1, s);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float one = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&one, e0, e1);
      if (rep > 0 && one < ms) ms = one;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(cyc.data(), d_probe, sizeof(unsigned) * cyc.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (!hip_ok(e, "split refinement probe")) { rc = RT_ERR_HIP; break; }
    std::vector<unsigned> cost((size_t)nh);
    unsigned long long total = 0;
    unsigned slowest = 0;
    for (int i = 0; i < nh; ++i) {
      unsigned m = 0;
      for (int w = 0; w < 4; ++w) { m = std::max(m, cyc[(size_t)i * 4 + w]); total += cyc[(size_t)i * 4 + w]; }
      cost[i] = m;
      slowest = std::max(slowest, m);
    }
    if (ms < best_ms * (round == 0 ? 1.0f : 0.98f)) {
      best_ms = ms;
      best_list = list;
      best_cost = cost;
      current_is_best = true;
      stale = 0;
    } else {
      current_is_best = false;
      stale++;
    }
    /* wave slots of the chip for this kernel: a workgroup is one wave per SIMD, and the LDS it asks for may allow
     * fewer workgroups per CU than the registers do */
    {
      const int by_regs = reflect ? RT_MIN_WAVES_BVH : RT_MIN_WAVES_BVH_PLAIN;
      const int by_lds = cur->launch_lds ? (int)((size_t)160 * 1024 / cur->launch_lds) : by_regs;
      slots = cur->n_cu * 4 * std::max(1, std::min(by_regs, by_lds));
    }
    const double even = (double)total / slots; /* cycles of a launch whose waves fill every slot to the end */
    if (getenv("RT_HIP_VERBOSE")) {
      int by_lg[5] = {0, 0, 0, 0, 0};
      for (int i = 0; i < nh; ++i) by_lg[std::min(4, RT_ENTRY_LG(list[i]))]++;
      fprintf(stderr, "refine_split round %d: %.1f us, %d mesh entries (parts 1/2/4/8/16: %d %d %d %d %d), slowest wave %u cycles, even %.0f\n",
              round, ms * 1e3, nh, by_lg[0], by_lg[1], by_lg[2], by_lg[3], by_lg[4], slowest, even);
      std::vector<int> idx((size_t)nh);
      for (int i = 0; i < nh; ++i) idx[i] = i;
      std::partial_sort(idx.begin(), idx.begin() + std::min(nh, 5), idx.end(), [&](int x, int y) { return cost[x] > cost[y]; });
      for (int q = 0; q < std::min(nh, 5); ++q) {
        const int i = idx[q], tile = list[i] & RT_TILE_MASK;
        fprintf(stderr, "  slow entry at position %d: tile %d (x %d, y %d), part %d of 2^%d, list of %zu, waves %u %u %u %u cycles\n", i, tile,
                tile % ((cur->scene.hsize + RT_TILE_W - 1) / RT_TILE_W), tile / ((cur->scene.hsize + RT_TILE_W - 1) / RT_TILE_W), RT_ENTRY_PART(list[i]),
                RT_ENTRY_LG(list[i]), (size_t)tile < plan->tile_tris.size() ? plan->tile_tris[(size_t)tile].size() : (size_t)0, cyc[(size_t)i * 4],
                cyc[(size_t)i * 4 + 1], cyc[(size_t)i * 4 + 2], cyc[(size_t)i * 4 + 3]);
      }
    }
    if (round == rounds || (double)slowest <= 1.15 * even || stale >= 3) break;
    const double threshold = std::max(even, cur->split_threshold * (double)slowest);
    struct Item { int32_t entry; double cost; };
    std::vector<Item> items;
    items.reserve((size_t)nh * 2);
    bool any = false;
    for (int i = 0; i < nh; ++i) {
      const int32_t entry = list[i];
      const int lg = RT_ENTRY_LG(entry), part = RT_ENTRY_PART(entry);
      if ((double)cost[i] > threshold && lg < cur->split_max_lg) {
        const int32_t tile = entry & RT_TILE_MASK;
        for (int h = 0; h < 2; ++h) items.push_back({RT_ENTRY(tile, 2 * part + h, lg + 1), 0.6 * cost[i]});
        any = true;
      } else {
        items.push_back({entry, (double)cost[i]});
      }
    }
    if (!any) break;
    std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.cost > y.cost; });
    std::vector<int32_t> next;
    next.reserve(items.size() + (size_t)n_light);
    for (const Item& it : items) next.push_back(it.entry);
    next.insert(next.end(), list.begin() + nh, list.end());
    list.swap(next);
    rc = adopt(list);
    current_is_best = false;
  }
  /* a candidate whose triangle lists would not fit (RT_ERR_INVALID from adopt) ends the search; the best list so
   * far is what the plan keeps */
  if (rc == RT_ERR_INVALID) rc = RT_OK;
  /* The list that won was put in order by what the round before it had measured, with estimates for the entries it
   * had just split.  Its own measurement is at hand: the same entries, longest first by that, are tried as well
   * (an entry that was guessed too cheap starts in a later round of workgroups, and the frame then waits for it). */
  if (rc == RT_OK && cur->split_resort && best_cost.size() + (size_t)n_light == best_list.size() && best_cost.size() > 1) {
    const size_t nh = best_cost.size();
    std::vector<int32_t> order(nh);
    for (size_t i = 0; i < nh; ++i) order[i] = (int32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return best_cost[(size_t)x] > best_cost[(size_t)y]; });
    std::vector<int32_t> resorted;
    resorted.reserve(best_list.size());
    for (int32_t i : order) resorted.push_back(best_list[(size_t)i]);
    resorted.insert(resorted.end(), best_list.begin() + (long)nh, best_list.end());
    if (resorted != best_list && (rc = adopt(resorted)) == RT_ERR_INVALID) {
      rc = RT_OK;
      current_is_best = false;
    } else if (resorted != best_list && rc == RT_OK) {
      float ms = 1e30f;
      hipError_t e = hipSuccess;
      for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) {
        e = hipEventRecord(e0, s);
        launch(false, bvh, reflect, dim3((unsigned)n_blocks), s, a, d_rgb, d_canvas);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float one = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&one, e0, e1);
        if (rep > 0 && one < ms) ms = one;
      }
      if (!hip_ok(e, "split refinement (re-ordered list)")) rc = RT_ERR_HIP;
      if (getenv("RT_HIP_VERBOSE")) fprintf(stderr, "refine_split re-ordered by its own measurement: %.1f us (best so far %.1f us)\n", ms * 1e3, best_ms * 1e3);
      if (rc == RT_OK && ms < best_ms * 0.99f) {
        best_list.swap(resorted);
        current_is_best = true;
      } else {
        current_is_best = false;
      }
    }
  }
  if (rc == RT_OK && !current_is_best) {
    rc = adopt(best_list);
    if (rc == RT_ERR_INVALID) { /* cannot happen for a list that was adopted before; if it does: the entries without lists */
      with_lists = false;
      plan->list_forms = 0;
      plan->list_total = 0;
      rc = adopt(best_list);
    }
  }
  if (d_probe) (void)hipFree(d_probe);
  plan->refining = false;
  plan->part_cache.clear();
  plan->tile_tris.clear();
  plan->tile_tris.shrink_to_fit();
  plan->tile_shadow.clear();
  plan->tile_shadow.shrink_to_fit();
  plan->tile_shadow_ok.clear();
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return rc;
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

/* The two formatting launches on `s`: text lengths of the chunks and groups, then the text.  chunk_len / group_len:
 * at least n_chunks + 1 words each; total_out: where the kernel stores the length of the file (device memory or the
 * device address of pinned host memory).  An empty image is one workgroup that finds no pixels and writes the
 * header and the length. */
int ppm_enqueue(const void* rgb_device, int width, int height, void* text_device, unsigned* chunk_len, unsigned* group_len,
                unsigned long long* total_out, hipStream_t s) {
  const long long n_px = (long long)width * height;
  const int n_chunks = (int)((n_px + PPM_CHUNK - 1) / PPM_CHUNK);
  PpmHeader hd;
  memset(&hd, 0, sizeof hd);
  hd.len = snprintf(hd.text, sizeof hd.text, "P3\n%d %d\n255\n", width, height); /* RayTracer.c:1335-1336 */
  /* groups: about a thousand of them keep both the length pass parallel and a chunk's sum over groups short */
  int per_group = n_chunks / 1024;
  per_group = per_group < 1 ? 1 : (per_group > PPM_MAX_GROUP ? PPM_MAX_GROUP : per_group);
  const int n_groups = (n_chunks + per_group - 1) / per_group;
  if (n_chunks > 0)
    hipLaunchKernelGGL(ppm_len_kernel, dim3((unsigned)n_groups), dim3(PPM_THREADS), 0, s, (const unsigned char*)rgb_device, n_px,
                       n_chunks, per_group, chunk_len, group_len);
  hipLaunchKernelGGL(ppm_write_kernel, dim3((unsigned)(n_chunks > 0 ? n_chunks : 1)), dim3(PPM_THREADS), 0, s,
                     (const unsigned char*)rgb_device, n_px, (const unsigned*)chunk_len, (const unsigned*)group_len, per_group, hd,
                     (unsigned char*)text_device, total_out);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

/* room for n_chunks + 1 words in each of the library-owned length arrays (every formatting call on this context
 * shares them; calls are ordered by stream or by ppm_ev) */
int ppm_reserve(int n_chunks) {
  if ((size_t)n_chunks + 1 <= cur->ppm_chunks_cap) return RT_OK;
  if (cur->ppm_pending) HIP_TRY(hipStreamSynchronize(cur->ppm_stream));
  HIP_TRY(hipStreamSynchronize(cur->stream));
  if (cur->d_ppm_len) (void)hipFree(cur->d_ppm_len);
  if (cur->d_ppm_group) (void)hipFree(cur->d_ppm_group);
  cur->d_ppm_len = nullptr;
  cur->d_ppm_group = nullptr;
  cur->ppm_chunks_cap = 0;
  HIP_TRY(hipMalloc((void**)&cur->d_ppm_len, sizeof(unsigned) * ((size_t)n_chunks + 1)));
  HIP_TRY(hipMalloc((void**)&cur->d_ppm_group, sizeof(unsigned) * ((size_t)n_chunks + 1)));
  cur->ppm_chunks_cap = (size_t)n_chunks + 1;
  return RT_OK;
}

/* ---- several GPUs of one process: RCCL, loaded when first asked for ----
 * The library does not link against librccl (half a gigabyte that a one-GPU host never needs): the handful of
 * entry points the gather uses are looked up with dlopen/dlsym at rt_hip_multi_init. */
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int g_multi_n = 0;                  /* devices 0 .. g_multi_n - 1 form the group */
ncclComm_t g_comms[kMaxDevices] = {};
unsigned char* g_gather = nullptr;  /* device 0: every rank's packed rows, rank after rank */
unsigned char* g_final = nullptr;   /* device 0: the rows in image order */
size_t g_gather_bytes = 0;
constexpr int kMaxChunks = 4;       /* a device's rows travel in up to this many parts, each as soon as it is rendered */
int g_multi_chunks = 0;             /* rt_hip_multi_set_chunks: 0 = chosen per frame */
hipEvent_t g_multi_ev[kMaxDevices][kMaxChunks] = {}; /* device r has rendered its part c (recorded on its stream) */
hipStream_t g_multi_xfer[kMaxDevices] = {};          /* device r's transfers: beside its kernels, not behind them */
hipEvent_t g_multi_done[kMaxChunks] = {};            /* device 0: part c of every device is in place */
hipStream_t g_multi_copy = nullptr;                  /* device 0: the placed rows on their way to the host */
hipEvent_t g_multi_t0 = nullptr, g_multi_t1 = nullptr; /* device 0: exchange + placement */

bool rccl_ok(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return true;
  snprintf(g_err, sizeof g_err, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return false;
}
#define RCCL_TRY(call)                              \
  do {                                              \
    if (!rccl_ok((call), #call)) return RT_ERR_HIP; \
  } while (0)

/* librt_hip_sort.so (rt_sort.h), looked for beside this library first */
struct SortLib {
  void* lib = nullptr;
  decltype(&rt_sort_pairs_u64_u32) pairs = nullptr;
  decltype(&rt_exclusive_sum_u32) sum = nullptr;
};
SortLib g_sort;
int load_sort() {
  if (g_sort.lib) return RT_OK;
  void* lib = nullptr;
  Dl_info here;
  if (dladdr((const void*)&rt_hip_strerror, &here) && here.dli_fname) {
    std::string path(here.dli_fname);
    const size_t slash = path.rfind('/');
    path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "librt_hip_sort.so";
    lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  }
  if (!lib) lib = dlopen("librt_hip_sort.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    snprintf(g_err, sizeof g_err, "the device BVH build needs librt_hip_sort.so beside librt_hip.so: %s", dlerror());
    return RT_ERR_UNSUPPORTED;
  }
  SortLib l;
  l.lib = lib;
  l.pairs = (decltype(l.pairs))dlsym(lib, "rt_sort_pairs_u64_u32");
  l.sum = (decltype(l.sum))dlsym(lib, "rt_exclusive_sum_u32");
  if (!l.pairs || !l.sum) {
    snprintf(g_err, sizeof g_err, "librt_hip_sort.so lacks rt_sort_pairs_u64_u32 / rt_exclusive_sum_u32");
    dlclose(lib);
    return RT_ERR_UNSUPPORTED;
  }
  g_sort = l;
  return RT_OK;
}

int load_rccl() {
  if (g_rccl.lib) return RT_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    snprintf(g_err, sizeof g_err, "dlopen librccl: %s", dlerror());
    return RT_ERR_HIP;
  }
  Rccl r;
  r.lib = lib;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(lib, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
  r.Send = (decltype(r.Send))dlsym(lib, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(lib, "ncclRecv");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
    snprintf(g_err, sizeof g_err, "librccl lacks an entry point the gather needs");
    dlclose(lib);
    return RT_ERR_HIP;
  }
  g_rccl = r;
  return RT_OK;
}

void multi_teardown() {
  for (int d = 0; d < g_multi_n; ++d) {
    if (g_comms[d] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_comms[d]);
    g_comms[d] = nullptr;
    (void)hipSetDevice(d);
    for (int c = 0; c < kMaxChunks; ++c)
      if (g_multi_ev[d][c]) { (void)hipEventDestroy(g_multi_ev[d][c]); g_multi_ev[d][c] = nullptr; }
    if (g_multi_xfer[d]) { (void)hipStreamDestroy(g_multi_xfer[d]); g_multi_xfer[d] = nullptr; }
  }
  if (g_multi_n > 0) (void)hipSetDevice(0);
  for (int c = 0; c < kMaxChunks; ++c)
    if (g_multi_done[c]) { (void)hipEventDestroy(g_multi_done[c]); g_multi_done[c] = nullptr; }
  if (g_multi_copy) { (void)hipStreamDestroy(g_multi_copy); g_multi_copy = nullptr; }
  if (g_gather) (void)hipFree(g_gather);
  if (g_final) (void)hipFree(g_final);
  if (g_multi_t0) (void)hipEventDestroy(g_multi_t0);
  if (g_multi_t1) (void)hipEventDestroy(g_multi_t1);
  g_gather = g_final = nullptr;
  g_gather_bytes = 0;
  g_multi_t0 = g_multi_t1 = nullptr;
  g_multi_n = 0;
}

/* Rows of the gathered buffer (rank after rank, each rank's rows packed in increasing y) into image order:
 * out row y = rank (y / band_rows) % n_ranks's packed row.  16 bytes per thread where the row length allows. */
__global__ __launch_bounds__(256) void place_bands_kernel(const unsigned char* __restrict__ gathered, unsigned char* __restrict__ out,
                                                          int row_bytes, int vsize, int band_rows, int n_ranks, int rank_lo, int rank_hi) {
  const int y = blockIdx.y;
  const int band = y / band_rows, rank = band % n_ranks;
  if (rank < rank_lo || rank >= rank_hi) return; /* only the rows of ranks rank_lo .. rank_hi - 1 (one part of a gather that arrives in parts) */
  const int packed = (band / n_ranks) * band_rows + (y - band * band_rows);
  /* rows of the ranks before `rank`: full rounds of bands, plus what the last, partial round gives them */
  const int stride = band_rows * n_ranks, full = vsize / stride, rem = vsize % stride;
  long long before = (long long)full * band_rows * rank;
  before += min(rem, rank * band_rows);
  const unsigned char* __restrict__ src = gathered + (before + packed) * (long long)row_bytes;
  unsigned char* __restrict__ dst = out + (long long)y * row_bytes;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 16;
  if (x + 16 <= row_bytes && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    *reinterpret_cast<uint4*>(dst + x) = *reinterpret_cast<const uint4*>(src + x);
  } else {
    for (int k = x; k < min(x + 16, row_bytes); ++k) dst[k] = src[k];
  }
}

/* device allocations of one build: freed on every exit unless handed over */
struct BuildBuffers {
  std::vector<void*> owned;
  ~BuildBuffers() { for (void* p : owned) if (p) (void)hipFree(p); }
  template <class T> bool get(T** out, size_t count) {
    void* p = nullptr;
    if (!hip_ok(hipMalloc(&p, sizeof(T) * (count ? count : 1)), "hipMalloc (bvh build)")) return false;
    owned.push_back(p);
    *out = (T*)p;
    return true;
  }
  void release(void* p) { for (void*& q : owned) if (q == p) q = nullptr; }
};

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
  /* A/B switches for development builds; a caller states what it wants through rt_render_params */
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
  if (const char* e = getenv("RT_HIP_PART_LISTS")) cur->part_lists = atoi(e) != 0;
  if (const char* e = getenv("RT_HIP_PART_SHADOW_MAX")) cur->part_shadow_max = std::max(0, atoi(e));
  if (const char* e = getenv("RT_HIP_SPLIT_THRESHOLD")) cur->split_threshold = std::fmax(0.1, std::fmin(0.99, atof(e)));
  if (const char* e = getenv("RT_HIP_SPLIT_MAX_LG")) cur->split_max_lg = std::max(0, std::min(6, atoi(e)));
  if (const char* e = getenv("RT_HIP_XCD_TILES")) cur->xcd_tiles = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("RT_HIP_LDS_STACK")) cur->lds_stack = atoi(e) != 0; /* 0 = traversal stack in scratch */
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
  if (sc->n_shapes != 0 || sc->materials_ext) return upload_general(sc, true);
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
  std::vector<DevSphere> hs((size_t)(ns > 0 ? ns : 1));
  for (int i = 0; i < ns; ++i) {
    memcpy(hs[i].inv, sc->sph_inv + 16 * (size_t)i, 12 * sizeof(double));
    hs[i].material = sc->sph_material[i];
    hs[i].order = sc->sph_order[i];
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
  if (!cur->d_scene) HIP_TRY(hipMalloc((void**)&cur->d_scene, sizeof(DevScene)));
  HIP_TRY(hipMemcpy(cur->d_scene, &cur->scene, sizeof(DevScene), hipMemcpyHostToDevice));
  cur->have_scene = true;
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

int rt_hip_build_bvh_device(int leaf_size, rt_bvh_device_info* info) {
  if (!cur->ready || !cur->have_scene) return RT_ERR_NOT_READY;
  if (leaf_size < 1 || leaf_size > 7) return RT_ERR_INVALID;
  if (cur->general) return RT_ERR_UNSUPPORTED; /* worlds of the general kernel walk the reference's list */
  const int n = cur->scene.n_tris;
  if (n >= (1 << 27)) return RT_ERR_TOO_LARGE;
  if (info) memset(info, 0, sizeof *info);
  {
    const int src = load_sort();
    if (src != RT_OK) return src;
  }
  if (n <= leaf_size) { /* nothing to walk: the exhaustive loop is the walk */
    if (cur->d_bvh) (void)hipFree(cur->d_bvh);
    cur->d_bvh = nullptr;
    cur->scene.n_bvh_nodes = 0;
    cur->scene.bvh_depth = 0;
    cur->bvh_single_leaf = n > 0;
    HIP_TRY(hipMemcpy(cur->d_scene, &cur->scene, sizeof(DevScene), hipMemcpyHostToDevice));
    cur->scene_serial++;
    return RT_OK;
  }
  /* what the scene keeps afterwards: one allocation each; everything else: one allocation carved up */
  BuildBuffers b;
  DevTri* tris_out;
  DevTriF* trif_out;
  DevTriPrimary* prim_out;
  double* normals_out;
  DevBvhNode* nodes = nullptr;
  const size_t nn = (size_t)n, ni = nn - 1;
  if (!b.get(&tris_out, nn) || !b.get(&trif_out, nn) || !b.get(&prim_out, nn) || !b.get(&normals_out, 3 * nn)) return RT_ERR_HIP;
  /* four-wide nodes are allocated once their number is known */
  if (!RT_BVH_WIDE && !b.get(&nodes, ni)) return RT_ERR_HIP;
  size_t sort_bytes = 0, scan_bytes = 0;
  HIP_TRY((hipError_t)g_sort.pairs(nullptr, &sort_bytes, nullptr, nullptr, nullptr, nullptr, n, cur->stream));
  HIP_TRY((hipError_t)g_sort.sum(nullptr, &scan_bytes, nullptr, nullptr, (int)ni, cur->stream));
  const size_t space_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes; /* both library calls work in it, in turn */
  size_t arena_bytes = 0;
  auto reserve = [&](size_t bytes) { const size_t at = arena_bytes; arena_bytes += (bytes + 255) & ~(size_t)255; return at; };
  const size_t at_keys = reserve(8 * nn), at_keys_sorted = reserve(8 * nn), at_index = reserve(4 * nn), at_index_sorted = reserve(4 * nn);
  const size_t at_visits = reserve(4 * ni), at_walked = reserve(4 * ni), at_slot = reserve(4 * ni);
  const size_t at_leaf_lo = reserve(12 * nn), at_leaf_hi = reserve(12 * nn), at_node_lo = reserve(12 * ni), at_node_hi = reserve(12 * ni);
  const size_t at_left = reserve(4 * ni), at_right = reserve(4 * ni), at_range_lo = reserve(4 * ni), at_range_hi = reserve(4 * ni);
  const size_t at_parent_of_node = reserve(4 * ni), at_parent_of_leaf = reserve(4 * nn), at_node_depth = reserve(4 * ni);
  const size_t at_space = reserve(space_bytes);
  unsigned char* arena;
  if (!b.get(&arena, arena_bytes)) return RT_ERR_HIP;
  unsigned long long* keys = (unsigned long long*)(arena + at_keys);
  unsigned long long* keys_sorted = (unsigned long long*)(arena + at_keys_sorted);
  unsigned* index = (unsigned*)(arena + at_index);
  unsigned* index_sorted = (unsigned*)(arena + at_index_sorted);
  unsigned* visits = (unsigned*)(arena + at_visits);
  unsigned* walked = (unsigned*)(arena + at_walked);
  unsigned* slot = (unsigned*)(arena + at_slot);
  float* leaf_lo = (float*)(arena + at_leaf_lo);
  float* leaf_hi = (float*)(arena + at_leaf_hi);
  float* node_lo = (float*)(arena + at_node_lo);
  float* node_hi = (float*)(arena + at_node_hi);
  int* left = (int*)(arena + at_left);
  int* right = (int*)(arena + at_right);
  int* range_lo = (int*)(arena + at_range_lo);
  int* range_hi = (int*)(arena + at_range_hi);
  int* parent_of_node = (int*)(arena + at_parent_of_node);
  int* parent_of_leaf = (int*)(arena + at_parent_of_leaf);
  int* node_depth = (int*)(arena + at_node_depth);
  unsigned char* sort_space = arena + at_space;

  double scale[3];
  for (int c = 0; c < 3; ++c) {
    const double extent = cur->mesh_hi[c] - cur->mesh_lo[c];
    scale[c] = extent > 0.0 ? 2097152.0 / extent : 0.0;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  if (!hip_ok(hipEventCreate(&e1), "hipEventCreate")) { (void)hipEventDestroy(e0); return RT_ERR_HIP; }
  const dim3 per_tri((unsigned)((nn + LBVH_THREADS - 1) / LBVH_THREADS)), per_node((unsigned)((ni + LBVH_THREADS - 1) / LBVH_THREADS));
  hipStream_t s = cur->stream;
  hipError_t e = hipEventRecord(e0, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lbvh_keys_kernel, per_tri, dim3(LBVH_THREADS), 0, s, (const DevTri*)cur->d_tris, n, cur->mesh_lo[0], cur->mesh_lo[1],
                       cur->mesh_lo[2], scale[0], scale[1], scale[2], keys, index);
    e = (hipError_t)g_sort.pairs(sort_space, &sort_bytes, keys, keys_sorted, index, index_sorted, n, s);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lbvh_gather_kernel, per_tri, dim3(LBVH_THREADS), 0, s, (const DevTri*)cur->d_tris,
                       (const DevTriPrimary*)cur->d_tri_primary, (const double*)cur->d_normals, (const unsigned*)index_sorted, n,
                       cur->box_pad, cur->filter_radius, trif_out, tris_out, prim_out, normals_out, leaf_lo, leaf_hi);
    hipLaunchKernelGGL(lbvh_hierarchy_kernel, per_node, dim3(LBVH_THREADS), 0, s, (const unsigned long long*)keys_sorted, n, leaf_size,
                       walked, left, right, range_lo, range_hi, parent_of_node, parent_of_leaf);
    e = hipMemsetAsync(visits, 0, sizeof(unsigned) * ni, s);
  }
#if !RT_BVH_WIDE
  if (e == hipSuccess) {
    size_t bytes = space_bytes;
    e = (hipError_t)g_sort.sum(sort_space, &bytes, walked, slot, (int)ni, s);
  }
#endif
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lbvh_fit_kernel, dim3((unsigned)((nn + LBVH_FIT_THREADS - 1) / LBVH_FIT_THREADS)), dim3(LBVH_FIT_THREADS), 0, s, n, leaf_size, (const int*)left, (const int*)right,
                       (const int*)range_lo, (const int*)range_hi, (const int*)parent_of_node, (const int*)parent_of_leaf,
                       (const float*)leaf_lo, (const float*)leaf_hi, node_lo, node_hi, node_depth, visits);
#if RT_BVH_WIDE
    /* which walked nodes become four-wide nodes (`visits` has served the fit and holds the flags now), and where */
    unsigned* const wide = visits;
    hipLaunchKernelGGL(lbvh_level_kernel, per_node, dim3(LBVH_THREADS), 0, s, n, (const int*)parent_of_node, (const unsigned*)walked, wide);
    e = hipGetLastError();
    if (e == hipSuccess) {
      size_t bytes = space_bytes;
      e = (hipError_t)g_sort.sum(sort_space, &bytes, wide, slot, (int)ni, s);
    }
#else
    hipLaunchKernelGGL(lbvh_emit_kernel, per_node, dim3(LBVH_THREADS), 0, s, n, leaf_size, (const int*)left, (const int*)right,
                       (const int*)range_lo, (const int*)range_hi, (const float*)leaf_lo, (const float*)leaf_hi,
                       (const float*)node_lo, (const float*)node_hi, (const unsigned*)walked, (const unsigned*)slot, nodes);
    e = hipGetLastError();
#endif
  }
  int packed = 0;
  unsigned last_slot = 0, last_flag = 0;
  float ms = 0.f;
#if RT_BVH_WIDE
  /* the number of four-wide nodes is known now: room for them, then the nodes themselves */
  hipEvent_t e_mid = nullptr;
  if (e == hipSuccess) e = hipEventCreate(&e_mid);
  if (e == hipSuccess) e = hipEventRecord(e_mid, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&last_slot, slot + (ni - 1), sizeof last_slot, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&last_flag, visits + (ni - 1), sizeof last_flag, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipMemcpyAsync(&packed, node_depth, sizeof packed, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  float ms_a = 0.f, ms_b = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_a, e0, e_mid);
  const int n_walked = (int)(last_slot + last_flag);
  if (e == hipSuccess && !b.get(&nodes, (size_t)(n_walked > 0 ? n_walked : 1))) e = hipErrorOutOfMemory;
  if (e == hipSuccess) e = hipEventRecord(e_mid, s);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(lbvh_emit4_kernel, per_node, dim3(LBVH_THREADS), 0, s, n, leaf_size, (const int*)left, (const int*)right,
                       (const int*)range_lo, (const int*)range_hi, (const float*)leaf_lo, (const float*)leaf_hi,
                       (const float*)node_lo, (const float*)node_hi, (const unsigned*)visits, (const unsigned*)slot, nodes);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipEventRecord(e1, s);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_b, e_mid, e1);
  ms = ms_a + ms_b; /* the kernels; the allocation in between is the caller's wall time */
  if (e_mid) (void)hipEventDestroy(e_mid);
  const int depth = (packed >> 8) & 255; /* subtrees a four-wide walk can have postponed at once */
  const int levels = packed & 255;
#else
  if (e == hipSuccess) e = hipEventRecord(e1, s);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e == hipSuccess) e = hipMemcpy(&packed, node_depth, sizeof packed, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&last_slot, slot + (ni - 1), sizeof last_slot, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&last_flag, walked + (ni - 1), sizeof last_flag, hipMemcpyDeviceToHost);
  const int n_walked = (int)(last_slot + last_flag);
  const int depth = packed & 255, levels = depth;
#endif
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (!hip_ok(e, "device BVH build")) return RT_ERR_HIP;
  if (depth < 1 || depth > RT_BVH_STACK || levels >= 255) {
    snprintf(g_err, sizeof g_err, "device BVH: %d levels, a walk may postpone %d subtrees; its stack holds %d", levels, depth, RT_BVH_STACK);
    return RT_ERR_TOO_LARGE;
  }
  /* hand over: the sorted triangle arrays and the nodes replace the scene's */
  if (cur->d_tris) (void)hipFree(cur->d_tris);
  if (cur->d_tri_primary) (void)hipFree(cur->d_tri_primary);
  if (cur->d_normals) (void)hipFree(cur->d_normals);
  if (cur->d_bvh) (void)hipFree(cur->d_bvh);
  cur->d_tris = tris_out; b.release(tris_out);
  if (cur->d_trif) (void)hipFree(cur->d_trif);
  cur->d_trif = trif_out; b.release(trif_out);
  cur->d_tri_primary = prim_out; b.release(prim_out);
  cur->d_normals = normals_out; b.release(normals_out);
  cur->d_bvh = nodes; b.release(nodes);
  cur->scene.n_bvh_nodes = n_walked;
  cur->scene.bvh_depth = depth;
  cur->bvh_single_leaf = false;
  /* the triangles are in another order now: what the host knows about them by number no longer applies (tile masks
   * and lists; the pixel rectangles only ever count triangles per tile, whatever their numbers) */
  cur->tri_xy.clear();
  cur->tri_world.clear();
  cur->host_trif.clear();
  cur->host_tri_k.clear();
  cur->host_tri_order.clear();
  cur->mesh_world.clear();
  cur->tri_depth.clear();
  HIP_TRY(hipMemcpy(cur->d_scene, &cur->scene, sizeof(DevScene), hipMemcpyHostToDevice));
  cur->scene_serial++;
  if (info) { info->n_nodes = n_walked; info->depth = levels; info->build_ms = ms; }
  return RT_OK;
}

size_t rt_hip_ppm_capacity(int width, int height) {
  if (width < 0 || height < 0) return 0;
  return 64 + (size_t)PPM_MAX_PX_TEXT * (size_t)width * (size_t)height;
}

int rt_hip_format_ppm_device(const void* rgb_device, int width, int height, void* text_device, size_t capacity,
                             size_t* bytes_out, void* stream) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (!rgb_device || !text_device || width < 0 || height < 0) return RT_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(rgb_device) & 3u) != 0 || (reinterpret_cast<uintptr_t>(text_device) & 15u) != 0)
    return RT_ERR_INVALID;
  if (capacity < rt_hip_ppm_capacity(width, height)) return RT_ERR_TOO_LARGE;
  const long long n_chunks_ll = ((long long)width * height + PPM_CHUNK - 1) / PPM_CHUNK;
  if (n_chunks_ll > (1ll << 30)) return RT_ERR_TOO_LARGE;
  hipStream_t s = stream ? (hipStream_t)stream : cur->stream;
  if (!cur->h_ppm_total) {
    HIP_TRY(hipHostMalloc((void**)&cur->h_ppm_total, sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**)&cur->d_ppm_total_host, cur->h_ppm_total, 0));
  }
  int rc = ppm_reserve((int)n_chunks_ll);
  if (rc != RT_OK) return rc;
  /* every call shares the length arrays and the pinned length word: a call on another stream than the last one
   * starts after that one has finished */
  if (cur->ppm_pending && cur->ppm_stream != s) {
    HIP_TRY(hipEventRecord(cur->ppm_ev, cur->ppm_stream));
    HIP_TRY(hipStreamWaitEvent(s, cur->ppm_ev, 0));
  }
  if ((rc = ppm_enqueue(rgb_device, width, height, text_device, cur->d_ppm_len, cur->d_ppm_group, cur->d_ppm_total_host, s)) != RT_OK)
    return rc;
  cur->ppm_stream = s; /* the write kernel has put the length into *h_ppm_total once the stream has drained */
  cur->ppm_pending = true;
  if (bytes_out) return rt_hip_ppm_last_length(bytes_out);
  return RT_OK;
}

int rt_hip_ppm_last_length(size_t* bytes_out) {
  if (!cur->ready || !cur->h_ppm_total) return RT_ERR_NOT_READY;
  if (!bytes_out) return RT_ERR_INVALID;
  if (cur->ppm_pending) {
    HIP_TRY(hipStreamSynchronize(cur->ppm_stream));
    cur->ppm_pending = false;
  }
  *bytes_out = (size_t)*cur->h_ppm_total;
  return RT_OK;
}

int rt_hip_format_ppm_device_noalloc(const void* rgb_device, int width, int height, void* text_device, size_t capacity,
                                     unsigned long long* length_device, void* stream) {
  if (!cur->ready || !cur->ppm_scratch_ptr) return RT_ERR_NOT_READY;
  if (!rgb_device || !text_device || !length_device || width < 0 || height < 0) return RT_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(rgb_device) & 3u) != 0 || (reinterpret_cast<uintptr_t>(text_device) & 15u) != 0 ||
      (reinterpret_cast<uintptr_t>(length_device) & 7u) != 0)
    return RT_ERR_INVALID;
  if (capacity < rt_hip_ppm_capacity(width, height)) return RT_ERR_TOO_LARGE;
  const long long n_px = (long long)width * height;
  const long long n_chunks_ll = (n_px + PPM_CHUNK - 1) / PPM_CHUNK;
  if (n_chunks_ll > kNoAllocMaxChunks) {
    snprintf(g_err, sizeof g_err, "rt_hip_format_ppm_device_noalloc: %lld pixels; the fixed scratch holds %lld", n_px,
             (long long)kNoAllocMaxChunks * PPM_CHUNK);
    return RT_ERR_TOO_LARGE;
  }
  hipStream_t s = stream ? (hipStream_t)stream : cur->stream;
  unsigned* const chunk_len = cur->ppm_scratch_ptr;
  unsigned* const group_len = cur->ppm_scratch_ptr + (kNoAllocMaxChunks + 1);
  /* the scratch is one per GPU: a call on another stream than the last one starts after that one has finished.
   * Nothing in it has to start at zero: the length pass writes every entry the write pass reads. */
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capturing) != hipSuccess) { (void)hipGetLastError(); capturing = hipStreamCaptureStatusNone; }
  if (capturing != hipStreamCaptureStatusNone) {
    /* Under capture nothing outside the capture may be waited for (an event recorded on another stream would
     * invalidate it), and when the graph will run is not known here: the two kernels are recorded as they are, and
     * the caller orders the graph's launches against every other user of this GPU's scratch. */
    return ppm_enqueue(rgb_device, width, height, text_device, chunk_len, group_len, length_device, s);
  }
  if (cur->noalloc_pending && cur->noalloc_stream != s) {
    /* the earlier stream must still exist (see the header); once it has drained there is nothing to order */
    const hipError_t busy = hipStreamQuery(cur->noalloc_stream);
    (void)hipGetLastError(); /* "not ready" is an answer, not a failure */
    if (busy == hipErrorNotReady) {
      HIP_TRY(hipEventRecord(cur->ppm_ev, cur->noalloc_stream));
      HIP_TRY(hipStreamWaitEvent(s, cur->ppm_ev, 0));
    }
    cur->noalloc_pending = false;
  }
  const int erc = ppm_enqueue(rgb_device, width, height, text_device, chunk_len, group_len, length_device, s);
  if (erc != RT_OK) return erc;
  cur->noalloc_stream = s;
  cur->noalloc_pending = true;
  return RT_OK;
}

int rt_hip_format_ppm(const void* rgb_device, int width, int height, char* text_host, size_t capacity,
                      size_t* bytes_out) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (!text_host && capacity > 0) return RT_ERR_INVALID;
  const size_t need = rt_hip_ppm_capacity(width, height);
  if (need > cur->d_ppm_text_bytes) {
    if (cur->ppm_pending) HIP_TRY(hipStreamSynchronize(cur->ppm_stream));
    if (cur->d_ppm_text) (void)hipFree(cur->d_ppm_text);
    cur->d_ppm_text = nullptr;
    cur->d_ppm_text_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_ppm_text, need));
    cur->d_ppm_text_bytes = need;
  }
  size_t len = 0;
  const int rc = rt_hip_format_ppm_device(rgb_device, width, height, cur->d_ppm_text, cur->d_ppm_text_bytes, &len, cur->stream);
  if (rc != RT_OK) return rc;
  if (bytes_out) *bytes_out = len;
  if (len > capacity) return RT_ERR_TOO_LARGE;
  if (len) HIP_TRY(hipMemcpyAsync(text_host, cur->d_ppm_text, len, hipMemcpyDeviceToHost, cur->stream));
  HIP_TRY(hipStreamSynchronize(cur->stream));
  return RT_OK;
}

int rt_hip_render_ppm(const rt_render_params* p, char* text_host, size_t capacity, size_t* bytes_out, rt_stats* stats) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  if (p->n_ranks != 1) return RT_ERR_INVALID; /* the file is one whole frame */
  const size_t bytes = (size_t)cur->scene.vsize * cur->scene.hsize * 3;
  if (bytes > cur->d_rgb_bytes) {
    if (cur->d_rgb) (void)hipFree(cur->d_rgb);
    cur->d_rgb = nullptr;
    cur->d_rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&cur->d_rgb, bytes ? bytes : 16));
    cur->d_rgb_bytes = bytes;
  }
  rc = render_common(p, cur->d_rgb, nullptr, cur->stream, stats, true);
  if (rc != RT_OK) return rc;
  return rt_hip_format_ppm(cur->d_rgb, cur->scene.hsize, cur->scene.vsize, text_host, capacity, bytes_out);
}

int rt_hip_render_ppm_submit(const rt_render_params* p, int slot) {
  int rc = check_params(p);
  if (rc != RT_OK) return rc;
  if (p->n_ranks != 1) return RT_ERR_INVALID; /* the file is one whole frame */
  if (slot < 0 || slot > 1) return RT_ERR_INVALID;
  Context::PpmSlot& q = cur->ppm_slot[slot];
  if (q.busy) return RT_ERR_NOT_READY; /* collect the slot's previous frame first */
  const int W = cur->scene.hsize, H = cur->scene.vsize;
  const size_t bytes = (size_t)W * H * 3, cap = rt_hip_ppm_capacity(W, H);
  if (!cur->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&cur->copy_stream, hipStreamNonBlocking));
  if (!q.formatted) {
    HIP_TRY(hipEventCreateWithFlags(&q.formatted, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&q.k0));
    HIP_TRY(hipEventCreate(&q.k1));
    HIP_TRY(hipHostMalloc((void**)&q.h_len, sizeof(unsigned long long), hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**)&q.d_len, q.h_len, 0));
  }
  if (bytes > q.rgb_bytes) {
    if (q.d_rgb) (void)hipFree(q.d_rgb);
    q.d_rgb = nullptr;
    q.rgb_bytes = 0;
    HIP_TRY(hipMalloc((void**)&q.d_rgb, bytes ? bytes : 16));
    q.rgb_bytes = bytes;
  }
  if (cap > q.text_bytes) {
    if (q.d_text) (void)hipFree(q.d_text);
    if (q.h_text) (void)hipHostFree(q.h_text);
    q.d_text = nullptr;
    q.h_text = nullptr;
    q.text_bytes = 0;
    HIP_TRY(hipMalloc((void**)&q.d_text, cap));
    HIP_TRY(hipHostMalloc((void**)&q.h_text, cap, hipHostMallocDefault));
    q.text_bytes = cap;
  }
  const long long n_chunks_ll = ((long long)W * H + PPM_CHUNK - 1) / PPM_CHUNK;
  if (n_chunks_ll > (1ll << 30)) return RT_ERR_TOO_LARGE;
  if ((rc = ppm_reserve((int)n_chunks_ll)) != RT_OK) return rc;
  hipStream_t s = cur->stream;
  if (cur->ppm_pending && cur->ppm_stream != s) { /* a formatting call of the caller's on another stream shares the length arrays */
    HIP_TRY(hipEventRecord(cur->ppm_ev, cur->ppm_stream));
    HIP_TRY(hipStreamWaitEvent(s, cur->ppm_ev, 0));
  }
  /* render and format on the library's stream; the kernel time goes to the slot's own event pair */
  hipEvent_t keep0 = cur->ev0, keep1 = cur->ev1;
  cur->ev0 = q.k0;
  cur->ev1 = q.k1;
  rc = render_enqueue(p, q.d_rgb, nullptr, s, true, p->count_rays != 0, true, false, &q.enq); /* plans, never tunes: rt_hip_tune does */
  cur->ev0 = keep0;
  cur->ev1 = keep1;
  if (rc != RT_OK) return rc;
  if ((rc = ppm_enqueue(q.d_rgb, W, H, q.d_text, cur->d_ppm_len, cur->d_ppm_group, q.d_len, s)) != RT_OK) return rc;
  HIP_TRY(hipEventRecord(q.formatted, s));
  cur->ppm_stream = s;
  cur->ppm_pending = true;
  q.busy = true;
  return RT_OK;
}

int rt_hip_render_ppm_collect(int slot, const char** text_host, size_t* bytes_out, rt_stats* stats) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (slot < 0 || slot > 1 || !text_host || !bytes_out) return RT_ERR_INVALID;
  Context::PpmSlot& q = cur->ppm_slot[slot];
  if (!q.busy) return RT_ERR_NOT_READY;
  /* the length is known once the frame is formatted; only then can the copy of exactly that many bytes be issued.
   * It runs on the copy stream, beside whatever the caller has submitted to the other slot meanwhile. */
  HIP_TRY(hipEventSynchronize(q.formatted));
  const size_t len = (size_t)*q.h_len;
  if (len > q.text_bytes) return RT_ERR_HIP;
  if (len) HIP_TRY(hipMemcpyAsync(q.h_text, q.d_text, len, hipMemcpyDeviceToHost, cur->copy_stream));
  HIP_TRY(hipStreamSynchronize(cur->copy_stream));
  q.busy = false;
  *text_host = q.h_text;
  *bytes_out = len;
  if (stats) {
    hipEvent_t keep0 = cur->ev0, keep1 = cur->ev1;
    cur->ev0 = q.k0;
    cur->ev1 = q.k1;
    Enqueued e = q.enq;
    if (e.count) e.count = false; /* the counters belong to the context and a later frame may have reset them */
    const int rc = render_collect(e, cur->stream, stats);
    cur->ev0 = keep0;
    cur->ev1 = keep1;
    if (rc != RT_OK) return rc;
  }
  return RT_OK;
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

size_t rt_hip_gather_offset(int hsize, int vsize, int band_rows, int n_ranks, int rank) {
  if (hsize < 0 || vsize < 0 || n_ranks < 1 || rank < 0 || rank > n_ranks || (n_ranks > 1 && band_rows < 1)) return 0;
  size_t rows = 0;
  for (int k = 0; k < rank; ++k) rows += (size_t)rt_hip_rank_rows(vsize, band_rows, n_ranks, k);
  return rows * (size_t)hsize * 3;
}

int rt_hip_place_bands_device(const void* gathered_device, void* rgb_out_device, int hsize, int vsize, int band_rows, int n_ranks,
                              void* stream) {
  if (!cur->ready) return RT_ERR_NOT_READY;
  if (!gathered_device || !rgb_out_device || hsize < 1 || vsize < 1 || n_ranks < 1 || band_rows < 1) return RT_ERR_INVALID;
  if ((long long)hsize * 3 > (1ll << 30)) return RT_ERR_TOO_LARGE;
  hipStream_t s = stream ? (hipStream_t)stream : cur->stream;
  const int row_bytes = hsize * 3;
  const dim3 grid((unsigned)((row_bytes + 256 * 16 - 1) / (256 * 16)), (unsigned)vsize);
  if (vsize > 65535) return RT_ERR_TOO_LARGE;
  hipLaunchKernelGGL(place_bands_kernel, grid, dim3(256), 0, s, (const unsigned char*)gathered_device, (unsigned char*)rgb_out_device,
                     row_bytes, vsize, n_ranks > 1 ? band_rows : vsize, n_ranks, 0, n_ranks);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

int rt_hip_multi_init(int n_devices) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
    snprintf(g_err, sizeof g_err, "no HIP device");
    return RT_ERR_NO_DEVICE;
  }
  if (n_devices == 0) n_devices = n; /* every visible device */
  if (n_devices < 1 || n_devices > n || n_devices > kMaxDevices) return RT_ERR_NO_DEVICE;
  if (g_multi_n == n_devices) return RT_OK;
  multi_teardown();
  int rc = load_rccl();
  if (rc != RT_OK) return rc;
  for (int d = 0; d < n_devices; ++d)
    if ((rc = rt_hip_init(d)) != RT_OK) return rc;
  int devs[kMaxDevices];
  for (int d = 0; d < n_devices; ++d) devs[d] = d;
  RCCL_TRY(g_rccl.CommInitAll(g_comms, n_devices, devs));
  g_multi_n = n_devices;
  for (int d = 0; d < n_devices; ++d) {
    HIP_TRY(hipSetDevice(d));
    for (int c = 0; c < kMaxChunks; ++c) HIP_TRY(hipEventCreateWithFlags(&g_multi_ev[d][c], hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&g_multi_xfer[d], hipStreamNonBlocking));
  }
  HIP_TRY(hipSetDevice(0));
  for (int c = 0; c < kMaxChunks; ++c) HIP_TRY(hipEventCreateWithFlags(&g_multi_done[c], hipEventDisableTiming));
  HIP_TRY(hipStreamCreateWithFlags(&g_multi_copy, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&g_multi_t0));
  HIP_TRY(hipEventCreate(&g_multi_t1));
  return rt_hip_set_device(0);
}

int rt_hip_multi_upload_scene(const rt_scene* scene) {
  if (g_multi_n < 1) return RT_ERR_NOT_READY;
  for (int d = 0; d < g_multi_n; ++d) {
    int rc = rt_hip_set_device(d);
    if (rc == RT_OK) rc = rt_hip_upload_scene(scene);
    if (rc != RT_OK) { (void)rt_hip_set_device(0); return rc; }
  }
  return rt_hip_set_device(0);
}

int rt_hip_multi_set_chunks(int chunks) {
  if (chunks < 0 || chunks > kMaxChunks) return RT_ERR_INVALID;
  g_multi_chunks = chunks;
  return RT_OK;
}

/* The frame over the group.  Every device's rows are dealt out once more into `chunks` parts (a device is `chunks`
 * ranks of a round-robin over devices x chunks bands: rank c * n + d is part c of device d), one launch per part.
 * All launches of all devices are enqueued first; nothing here waits for a device before the others have their work.
 * Part c of device d leaves for device 0 on d's transfer stream as soon as d has rendered it - beside d's next part,
 * not behind it - and device 0 puts the parts of every device in place and sends them on to the host one part after
 * the other, so the gather and the copy to the host run under the tail of the rendering instead of after it. */
int rt_hip_render_multi(const rt_render_params* p, uint8_t* rgb_out, rt_stats* stats) {
  if (g_multi_n < 1) return RT_ERR_NOT_READY;
  if (!p || !rgb_out) return RT_ERR_INVALID;
  if (p->n_ranks != 1 || p->rank != 0) return RT_ERR_INVALID; /* the library deals the bands out itself */
  const int n = g_multi_n;
  int rc = rt_hip_set_device(0);
  if (rc != RT_OK) return rc;
  if ((rc = check_params(p)) != RT_OK) return rc;
  const int W = cur->scene.hsize, H = cur->scene.vsize;
  if (H > 65535) return RT_ERR_TOO_LARGE;
  const bool want = stats != nullptr, count = want && p->count_rays != 0;
  /* parts: a frame of a millisecond or more per device hides its gather under its own tail in two parts; a short
   * frame would only pay for the extra launches; the counting build keeps one set of counters per device */
  const long long samples = (long long)W * H * p->spp_grid * p->spp_grid;
  int chunks = g_multi_chunks > 0 ? g_multi_chunks : (samples >= RT_MULTI_GPU_MIN_SAMPLES ? 2 : 1);
  if (count) chunks = 1;
  /* one tile row per band unless the caller says otherwise; fewer rows per band when the image is short, so
   * that every part of every device gets some (bands of 16 rows balance the compact lit region of mesh scenes) */
  int band = p->band_rows > 1 ? p->band_rows : RT_TILE_H;
  while (chunks > 1 && (H + band - 1) / band < n * chunks) --chunks;
  const int V = n * chunks;
  while (band > 1 && (H + band - 1) / band < 2 * V) band /= 2;
  const size_t row_bytes = (size_t)W * 3, image_bytes = row_bytes * H;
  if (image_bytes > g_gather_bytes) {
    HIP_TRY(hipDeviceSynchronize());
    if (g_gather) (void)hipFree(g_gather);
    if (g_final) (void)hipFree(g_final);
    g_gather = g_final = nullptr;
    g_gather_bytes = 0;
    HIP_TRY(hipMalloc((void**)&g_gather, image_bytes));
    HIP_TRY(hipMalloc((void**)&g_final, image_bytes));
    g_gather_bytes = image_bytes;
  }
  rt_render_params q = *p;
  q.band_rows = band;
  q.n_ranks = V;
  Enqueued enq[kMaxDevices];
  size_t bytes[kMaxDevices][kMaxChunks], offset[kMaxDevices][kMaxChunks], local[kMaxDevices][kMaxChunks];
  auto fail = [&](int code) { (void)rt_hip_set_device(0); return code; };
  /* every device renders its parts: device 0 straight into its places of the gathered buffer */
  for (int d = 0; d < n; ++d) {
    if ((rc = rt_hip_set_device(d)) != RT_OK) return fail(rc);
    if (!cur->have_scene || cur->scene.hsize != W || cur->scene.vsize != H) return fail(RT_ERR_NOT_READY);
    size_t mine = 0;
    for (int c = 0; c < chunks; ++c) {
      const int v = c * n + d;
      bytes[d][c] = (size_t)rt_hip_rank_rows(H, band, V, v) * row_bytes;
      offset[d][c] = rt_hip_gather_offset(W, H, band, V, v);
      local[d][c] = mine;
      mine += bytes[d][c];
    }
    if (d > 0 && mine > cur->d_rgb_bytes) {
      HIP_TRY(hipStreamSynchronize(cur->stream));
      if (cur->d_rgb) (void)hipFree(cur->d_rgb);
      cur->d_rgb = nullptr;
      cur->d_rgb_bytes = 0;
      if (!hip_ok(hipMalloc((void**)&cur->d_rgb, mine ? mine : 16), "hipMalloc (rows of a device)")) return fail(RT_ERR_HIP);
      cur->d_rgb_bytes = mine;
    }
    if (want && !hip_ok(hipEventRecord(cur->ev0, cur->stream), "hipEventRecord")) return fail(RT_ERR_HIP);
    for (int c = 0; c < chunks; ++c) {
      q.rank = c * n + d;
      unsigned char* const target = d == 0 ? g_gather + offset[0][c] : cur->d_rgb + local[d][c];
      /* plans, never tunes here: no device waits for another's measurements (they are made after the frame, below) */
      if ((rc = render_enqueue(&q, target, nullptr, cur->stream, false, count, true, false, &enq[d])) != RT_OK) return fail(rc);
      if (!hip_ok(hipEventRecord(g_multi_ev[d][c], cur->stream), "hipEventRecord")) return fail(RT_ERR_HIP);
    }
    if (want && !hip_ok(hipEventRecord(cur->ev1, cur->stream), "hipEventRecord")) return fail(RT_ERR_HIP);
    enq[d].timed = want;
    enq[d].rows = rt_hip_rank_rows(H, band, n, d) /* what the device rendered in all */;
  }
  if ((rc = rt_hip_set_device(0)) != RT_OK) return rc;
  hipStream_t x0 = g_multi_xfer[0];
  if (want) HIP_TRY(hipEventRecord(g_multi_t0, cur->stream)); /* device 0's own rows are rendered */
  for (int c = 0; c < chunks; ++c) {
    /* one exchange step per part: every other device sends that part to device 0 over its own link (a grouped
     * send/recv: the gather has a fixed root and unequal parts, which no single RCCL collective expresses) */
    HIP_TRY(hipStreamWaitEvent(x0, g_multi_ev[0][c], 0));
    if (n > 1) {
      for (int d = 1; d < n; ++d) {
        if (!hip_ok(hipSetDevice(d), "hipSetDevice")) return fail(RT_ERR_HIP);
        if (!hip_ok(hipStreamWaitEvent(g_multi_xfer[d], g_multi_ev[d][c], 0), "hipStreamWaitEvent")) return fail(RT_ERR_HIP);
      }
      if (!hip_ok(hipSetDevice(0), "hipSetDevice")) return fail(RT_ERR_HIP);
      RCCL_TRY(g_rccl.GroupStart());
      bool group_ok = true; /* a group that was opened is closed whatever happens inside it */
      for (int d = 1; d < n && group_ok; ++d) {
        if (bytes[d][c] == 0) continue;
        group_ok = rccl_ok(g_rccl.Send(g_ctx[d].d_rgb + local[d][c], bytes[d][c], ncclUint8, 0, g_comms[d], g_multi_xfer[d]), "ncclSend") &&
                   rccl_ok(g_rccl.Recv(g_gather + offset[d][c], bytes[d][c], ncclUint8, d, g_comms[0], x0), "ncclRecv");
      }
      const ncclResult_t ended = g_rccl.GroupEnd();
      (void)rt_hip_set_device(0);
      if (!group_ok) return RT_ERR_HIP;
      if (!rccl_ok(ended, "ncclGroupEnd")) return RT_ERR_HIP;
    }
    const dim3 grid((unsigned)((row_bytes + 256 * 16 - 1) / (256 * 16)), (unsigned)H);
    hipLaunchKernelGGL(place_bands_kernel, grid, dim3(256), 0, x0, (const unsigned char*)g_gather, g_final, (int)row_bytes, H,
                       V > 1 ? band : H, V, c * n, (c + 1) * n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(g_multi_done[c], x0));
  }
  if (want) HIP_TRY(hipEventRecord(g_multi_t1, x0));
  /* to the host part by part: part c is the n bands from band c * n on out of every V, i.e. runs of n * band rows
   * every V * band rows - one strided copy, and one more for a run the bottom of the image cuts short */
  for (int c = 0; c < chunks; ++c) {
    HIP_TRY(hipStreamWaitEvent(g_multi_copy, g_multi_done[c], 0));
    const size_t run_rows = (size_t)n * band, period_rows = (size_t)V * band, first_row = (size_t)c * n * band;
    size_t full = 0;
    while (first_row + full * period_rows + run_rows <= (size_t)H) ++full;
    if (full > 0)
      HIP_TRY(hipMemcpy2DAsync(rgb_out + first_row * row_bytes, period_rows * row_bytes, g_final + first_row * row_bytes,
                               period_rows * row_bytes, run_rows * row_bytes, full, hipMemcpyDeviceToHost, g_multi_copy));
    const size_t tail_row = first_row + full * period_rows;
    if (tail_row < (size_t)H)
      HIP_TRY(hipMemcpyAsync(rgb_out + tail_row * row_bytes, g_final + tail_row * row_bytes, ((size_t)H - tail_row) * row_bytes,
                             hipMemcpyDeviceToHost, g_multi_copy));
  }
  HIP_TRY(hipStreamSynchronize(g_multi_copy));
  if (want) {
    rt_stats sum;
    memset(&sum, 0, sizeof sum);
    for (int d = 0; d < n; ++d) {
      rt_stats one;
      if ((rc = rt_hip_set_device(d)) != RT_OK) return rc;
      if ((rc = render_collect(enq[d], cur->stream, &one)) != RT_OK) return fail(rc);
      sum.rays_primary += one.rays_primary; sum.rays_shadow += one.rays_shadow; sum.rays_secondary += one.rays_secondary;
      sum.prim_tests += one.prim_tests; sum.node_tests += one.node_tests; sum.sphere_tests += one.sphere_tests;
      sum.tri_exact += one.tri_exact; sum.sphere_exact += one.sphere_exact;
      sum.zero_t_entries += one.zero_t_entries; sum.list_cap_rays += one.list_cap_rays;
      if (one.kernel_ms > sum.kernel_ms) sum.kernel_ms = one.kernel_ms; /* the devices run side by side */
      sum.rows_rendered += one.rows_rendered;
      sum.accel_used = one.accel_used;
    }
    if ((rc = rt_hip_set_device(0)) != RT_OK) return rc;
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, g_multi_t0, g_multi_t1));
    sum.exchange_ms = ms;
    *stats = sum;
  }
  /* the frame is out; a caller that will render more (no RT_RENDER_NO_TUNE) gets every device's plans tuned now */
  if (!(p->flags & RT_RENDER_NO_TUNE) && !count) {
    for (int d = 0; d < n; ++d) {
      if (rt_hip_set_device(d) != RT_OK) break;
      bool pending = false;
      for (const Context::TilePlan& t : cur->plans) pending = pending || (t.refine_left > 0 && t.key[0] == (long long)cur->scene_serial);
      for (int c = 0; pending && c < chunks; ++c) {
        q.rank = c * n + d;
        if ((rc = rt_hip_tune(&q)) != RT_OK) return fail(rc);
      }
    }
    if ((rc = rt_hip_set_device(0)) != RT_OK) return rc;
  }
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

int rt_hip_multi_devices(void) { return g_multi_n; }

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

}  // extern "C"
