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
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#define NBINS 16

typedef struct { double lo[3], hi[3]; } box_t;
typedef struct { box_t b; double c[3]; int tri; } prim_t;

/* a subtree left for a worker thread: the node that stands for it in the top of the tree, and its triangles */
typedef struct { int node, first, count; } task_t;

typedef struct {
  prim_t* prims;
  rt_bvh_node* nodes;
  int n_nodes, cap_nodes;
  int leaf_size;
  double pad;
  /* top of a parallel build only: ranges of at most defer_count triangles become tasks instead of subtrees */
  int defer_count;
  task_t* tasks;
  int n_tasks, cap_tasks;
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
  if (B->defer_count > 0 && count <= B->defer_count && count > B->leaf_size) {
    if (B->n_tasks == B->cap_tasks) {
      int nc = B->cap_tasks ? B->cap_tasks * 2 : 64;
      task_t* q = realloc(B->tasks, sizeof(task_t) * (size_t)nc);
      if (!q) return -1;
      B->tasks = q; B->cap_tasks = nc;
    }
    B->tasks[B->n_tasks].node = me;
    B->tasks[B->n_tasks].first = first;
    B->tasks[B->n_tasks].count = count;
    B->n_tasks++;
    B->nodes[me].left = first; /* a leaf for now; the subtree's root replaces it */
    B->nodes[me].right = -1;
    return me;
  }
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

/* ---- subtrees on worker threads ----
 * The ranges of different tasks are disjoint parts of `prims`, every worker has a node array of its own, and
 * tasks are handed out by a counter under a mutex; the subtrees are spliced into the top afterwards, in task
 * order, so the tree does not depend on the number of threads or on their timing. */
typedef struct {
  const build_t* top;
  build_t* sub;           /* one per task */
  int* status;            /* build_range's result per task */
  int next;
  pthread_mutex_t lock;
} pool_t;

static void* pool_worker(void* arg) {
  pool_t* P = arg;
  for (;;) {
    pthread_mutex_lock(&P->lock);
    const int t = P->next < P->top->n_tasks ? P->next++ : -1;
    pthread_mutex_unlock(&P->lock);
    if (t < 0) return NULL;
    build_t* S = &P->sub[t];
    memset(S, 0, sizeof *S);
    S->prims = P->top->prims;
    S->leaf_size = P->top->leaf_size;
    S->pad = P->top->pad;
    P->status[t] = build_range(S, P->top->tasks[t].first, P->top->tasks[t].count);
  }
}

static int host_threads(void) {
  long n = sysconf(_SC_NPROCESSORS_ONLN);
  const char* e = getenv("RT_HOST_THREADS");
  if (e && atoi(e) > 0) n = atoi(e);
  if (n < 1) n = 1;
  if (n > 16) n = 16;
  return (int)n;
}

/* build the deferred subtrees and put them behind the top nodes: node j > 0 of task t becomes node
 * base_t + j - 1, its root replaces the placeholder (children keep following their parent in the array) */
static int build_tasks(build_t* B, int n_threads) {
  pool_t P;
  memset(&P, 0, sizeof P);
  P.top = B;
  P.sub = calloc((size_t)B->n_tasks, sizeof(build_t));
  P.status = calloc((size_t)B->n_tasks, sizeof(int));
  pthread_t* th = calloc((size_t)n_threads, sizeof(pthread_t));
  int ok = P.sub && P.status && th && pthread_mutex_init(&P.lock, NULL) == 0;
  int started = 0;
  if (ok) {
    for (int i = 0; i < n_threads - 1; ++i)
      if (pthread_create(&th[started], NULL, pool_worker, &P) == 0) started++;
    pool_worker(&P); /* this thread works too: the build completes even if no thread could be started */
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&P.lock);
    for (int t = 0; t < B->n_tasks && ok; ++t) ok = P.status[t] == 0;
  }
  size_t extra = 0;
  for (int t = 0; t < B->n_tasks && ok; ++t) extra += (size_t)P.sub[t].n_nodes - 1;
  if (ok && extra > 0) {
    if ((size_t)B->n_nodes + extra > (size_t)0x7fffffff) ok = 0;
    rt_bvh_node* q = ok ? realloc(B->nodes, sizeof(rt_bvh_node) * ((size_t)B->n_nodes + extra)) : NULL;
    if (!q) ok = 0;
    else { B->nodes = q; B->cap_nodes = B->n_nodes + (int)extra; }
  }
  for (int t = 0; t < B->n_tasks && ok; ++t) {
    const build_t* S = &P.sub[t];
    const int base = B->n_nodes;
    for (int j = 0; j < S->n_nodes; ++j) {
      rt_bvh_node nd = S->nodes[j];
      if (nd.right >= 0) { nd.left += base - 1; nd.right += base - 1; } /* inner: children are nodes >= 1 of the subtree */
      B->nodes[j == 0 ? B->tasks[t].node : base + j - 1] = nd;
    }
    B->n_nodes += S->n_nodes - 1;
  }
  if (P.sub) for (int t = 0; t < B->n_tasks; ++t) free(P.sub[t].nodes);
  free(P.sub); free(P.status); free(th);
  return ok ? 0 : -1;
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
  const int n_threads = host_threads();
  /* two or more subtrees per thread: finer tasks balance better but lengthen the single-threaded top of the build */
  if (n_threads > 1 && n >= 65536) B.defer_count = n / (2 * n_threads);
  int rc = build_range(&B, 0, n) < 0 ? -1 : 0;
  B.defer_count = 0;
  if (rc == 0 && B.n_tasks > 0) rc = build_tasks(&B, n_threads);
  free(B.tasks);
  if (rc < 0) { free(B.prims); free(B.nodes); return RT_ERR_INVALID; }
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
