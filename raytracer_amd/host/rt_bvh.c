/*
 * rt_bvh.c - binned surface-area-heuristic BVH build on the host.
 *
 * Boxes are stored as float, rounded outwards, after padding every triangle's
 * exact double bounds by 1e-6 x (largest |coordinate| in the mesh, at least 1).
 * The pad is ~1000x larger than anything that rounding can do to the
 * reference's double-precision Moeller-Trumbore acceptance test
 * (intersect_triangle, RayTracer/RayTracer.c:943-966) or to the kernels'
 * double-precision slab test, so a box test can only report false positives.
 */
#include "rt_bvh.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NBINS 16

typedef struct { double lo[3], hi[3]; } box_t;
typedef struct { box_t b; double c[3]; int tri; } prim_t;

typedef struct {
  prim_t* prims;
  rt_bvh_node* nodes;
  int n_nodes, cap_nodes;
  int leaf_size;
  double pad;
} build_t;

static void box_empty(box_t* b) {
  for (int k = 0; k < 3; ++k) { b->lo[k] = DBL_MAX; b->hi[k] = -DBL_MAX; }
}
static void box_grow(box_t* b, const box_t* o) {
  for (int k = 0; k < 3; ++k) {
    if (o->lo[k] < b->lo[k]) b->lo[k] = o->lo[k];
    if (o->hi[k] > b->hi[k]) b->hi[k] = o->hi[k];
  }
}
static double box_area(const box_t* b) {
  double dx = b->hi[0] - b->lo[0], dy = b->hi[1] - b->lo[1], dz = b->hi[2] - b->lo[2];
  if (dx < 0) return 0.0;
  return dx * dy + dy * dz + dz * dx;
}
static float f_down(double v) { float f = (float)v; return (double)f > v ? nextafterf(f, -INFINITY) : f; }
static float f_up(double v) { float f = (float)v; return (double)f < v ? nextafterf(f, INFINITY) : f; }

static int new_node(build_t* B) {
  if (B->n_nodes == B->cap_nodes) {
    int nc = B->cap_nodes ? B->cap_nodes * 2 : 256;
    rt_bvh_node* q = realloc(B->nodes, sizeof(rt_bvh_node) * (size_t)nc);
    if (!q) return -1;
    B->nodes = q; B->cap_nodes = nc;
  }
  return B->n_nodes++;
}

static int build_range(build_t* B, int first, int count) {
  int me = new_node(B);
  if (me < 0) return -1;
  box_t nb, cb;
  box_empty(&nb); box_empty(&cb);
  for (int i = first; i < first + count; ++i) {
    box_grow(&nb, &B->prims[i].b);
    for (int k = 0; k < 3; ++k) {
      if (B->prims[i].c[k] < cb.lo[k]) cb.lo[k] = B->prims[i].c[k];
      if (B->prims[i].c[k] > cb.hi[k]) cb.hi[k] = B->prims[i].c[k];
    }
  }
  for (int k = 0; k < 3; ++k) {
    B->nodes[me].lo[k] = f_down(nb.lo[k] - B->pad);
    B->nodes[me].hi[k] = f_up(nb.hi[k] + B->pad);
  }
  int axis = 0;
  for (int k = 1; k < 3; ++k)
    if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
  const double ext = cb.hi[axis] - cb.lo[axis];
  if (count <= B->leaf_size || !(ext > 0.0)) {
    if (count > B->leaf_size && !(ext > 0.0)) {
      /* coincident centroids: split in the middle to respect the leaf limit */
      int mid = first + count / 2;
      int l = build_range(B, first, mid - first);
      int r = build_range(B, mid, first + count - mid);
      if (l < 0 || r < 0) return -1;
      B->nodes[me].left = l; B->nodes[me].right = r;
      return me;
    }
    B->nodes[me].left = first;
    B->nodes[me].right = -count;
    return me;
  }
  /* binned SAH: the cheapest of the NBINS - 1 splits of each of the three axes */
  int best = -1, best_axis = axis;
  double best_cost = DBL_MAX;
  for (int ax = 0; ax < 3; ++ax) {
    const double ext_a = cb.hi[ax] - cb.lo[ax];
    if (!(ext_a > 0.0)) continue;
    box_t bin_box[NBINS];
    int bin_cnt[NBINS];
    for (int i = 0; i < NBINS; ++i) { box_empty(&bin_box[i]); bin_cnt[i] = 0; }
    const double scale_a = NBINS / ext_a;
    for (int i = first; i < first + count; ++i) {
      int bi = (int)((B->prims[i].c[ax] - cb.lo[ax]) * scale_a);
      if (bi < 0) bi = 0;
      if (bi >= NBINS) bi = NBINS - 1;
      bin_cnt[bi]++;
      box_grow(&bin_box[bi], &B->prims[i].b);
    }
    double right_area[NBINS];
    int right_cnt[NBINS];
    box_t acc;
    box_empty(&acc);
    int cnt = 0;
    for (int i = NBINS - 1; i > 0; --i) {
      if (bin_cnt[i]) box_grow(&acc, &bin_box[i]);
      cnt += bin_cnt[i];
      right_area[i] = box_area(&acc);
      right_cnt[i] = cnt;
    }
    box_empty(&acc);
    cnt = 0;
    for (int i = 1; i < NBINS; ++i) { /* split between bin i-1 and bin i */
      if (bin_cnt[i - 1]) box_grow(&acc, &bin_box[i - 1]);
      cnt += bin_cnt[i - 1];
      if (cnt == 0 || right_cnt[i] == 0) continue;
      double cost = box_area(&acc) * cnt + right_area[i] * right_cnt[i];
      if (cost < best_cost) { best_cost = cost; best = i; best_axis = ax; }
    }
  }
  axis = best_axis;
  const double scale = NBINS / (cb.hi[axis] - cb.lo[axis]);
  int mid;
  if (best < 0) {
    mid = first + count / 2;
  } else {
    /* stable partition by bin: keeps the original triangle order inside each side */
    prim_t* tmp = malloc(sizeof(prim_t) * (size_t)count);
    if (!tmp) return -1;
    int nl = 0;
    for (int i = first; i < first + count; ++i) {
      int bi = (int)((B->prims[i].c[axis] - cb.lo[axis]) * scale);
      if (bi < 0) bi = 0;
      if (bi >= NBINS) bi = NBINS - 1;
      if (bi < best) tmp[nl++] = B->prims[i];
    }
    int k = nl;
    for (int i = first; i < first + count; ++i) {
      int bi = (int)((B->prims[i].c[axis] - cb.lo[axis]) * scale);
      if (bi < 0) bi = 0;
      if (bi >= NBINS) bi = NBINS - 1;
      if (bi >= best) tmp[k++] = B->prims[i];
    }
    mid = first + nl;
    memcpy(B->prims + first, tmp, sizeof(prim_t) * (size_t)count);
    free(tmp);
    if (mid == first || mid == first + count) mid = first + count / 2;
  }
  int l = build_range(B, first, mid - first);
  int r = build_range(B, mid, first + count - mid);
  if (l < 0 || r < 0) return -1;
  B->nodes[me].left = l;
  B->nodes[me].right = r;
  return me;
}

int rt_bvh_build(const double* p123, int n, int leaf_size, rt_bvh_node** nodes, int* n_nodes, int32_t** tri_index) {
  if (!p123 || n < 1 || leaf_size < 1 || !nodes || !n_nodes || !tri_index) return RT_ERR_INVALID;
  build_t B;
  memset(&B, 0, sizeof B);
  B.leaf_size = leaf_size;
  B.prims = malloc(sizeof(prim_t) * (size_t)n);
  if (!B.prims) return RT_ERR_INVALID;
  double amax = 1.0;
  for (int i = 0; i < n; ++i) {
    prim_t* p = &B.prims[i];
    box_empty(&p->b);
    for (int v = 0; v < 3; ++v)
      for (int k = 0; k < 3; ++k) {
        double x = p123[9 * (size_t)i + 3 * v + k];
        if (!isfinite(x)) { free(B.prims); return RT_ERR_UNSUPPORTED; }
        if (x < p->b.lo[k]) p->b.lo[k] = x;
        if (x > p->b.hi[k]) p->b.hi[k] = x;
        if (fabs(x) > amax) amax = fabs(x);
      }
    for (int k = 0; k < 3; ++k) p->c[k] = 0.5 * (p->b.lo[k] + p->b.hi[k]);
    p->tri = i;
  }
  B.pad = 1e-6 * amax;
  if (build_range(&B, 0, n) < 0) { free(B.prims); free(B.nodes); return RT_ERR_INVALID; }
  int32_t* idx = malloc(sizeof(int32_t) * (size_t)n);
  if (!idx) { free(B.prims); free(B.nodes); return RT_ERR_INVALID; }
  for (int i = 0; i < n; ++i) idx[i] = B.prims[i].tri;
  free(B.prims);
  *nodes = B.nodes;
  *n_nodes = B.n_nodes;
  *tri_index = idx;
  return RT_OK;
}

static int depth_of(const rt_bvh_node* nodes, int i) {
  if (nodes[i].right < 0) return 1;
  int l = depth_of(nodes, nodes[i].left), r = depth_of(nodes, nodes[i].right);
  return 1 + (l > r ? l : r);
}
int rt_bvh_depth(const rt_bvh_node* nodes, int n_nodes) {
  if (!nodes || n_nodes < 1) return 0;
  return depth_of(nodes, 0);
}
