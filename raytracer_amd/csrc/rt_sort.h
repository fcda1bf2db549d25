/* The two library primitives the device BVH build needs, kept in a library of their own (librt_hip_sort.so):
 * the radix sort and the scan of rocPRIM instantiate hundreds of kernels, a megabyte of code object that every
 * process would load at start-up although only scenes of thousands of triangles without a host-built tree use
 * them.  librt_hip.so loads that library when rt_hip_build_bvh_device first runs.
 * Both follow the hipCUB convention: temp == NULL only reports the temporary storage needed in *temp_bytes.
 * Return value: a hipError_t. */
#ifndef RT_SORT_H
#define RT_SORT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* (key, value) pairs by bits 0..62 of the key (the build's codes leave bit 63 clear), stable */
int rt_sort_pairs_u64_u32(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                          const unsigned* values_in, unsigned* values_out, int n, void* stream);
/* out[i] = in[0] + ... + in[i - 1] */
int rt_exclusive_sum_u32(void* temp, size_t* temp_bytes, const unsigned* in, unsigned* out, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
