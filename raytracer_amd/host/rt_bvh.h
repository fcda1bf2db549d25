/*
 * rt_bvh.h - host-built bounding volume hierarchy over the triangle list.
 * The reference has no acceleration structure: intersect_world walks every
 * shape for every ray (RayTracer/RayTracer.c:1043-1055).  The hierarchy only
 * prunes that walk; it must never hide a triangle the full walk would hit.
 */
#ifndef RT_BVH_H
#define RT_BVH_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* p123: n triangles x 9 doubles (p1 p2 p3).  On success *nodes (malloc'd,
 * *n_nodes entries, root at 0, depth-first order) and *tri_index (malloc'd, n
 * entries: triangle numbers in leaf order).  Deterministic: same input, same
 * tree. */
int rt_bvh_build(const double* p123, int n, int leaf_size, rt_bvh_node** nodes, int* n_nodes, int32_t** tri_index);

/* depth of the deepest leaf (root = 1); the traversal stack must hold it */
int rt_bvh_depth(const rt_bvh_node* nodes, int n_nodes);

#ifdef __cplusplus
}
#endif
#endif
