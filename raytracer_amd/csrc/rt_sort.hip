/* librt_hip_sort.so: see rt_sort.h. */
#include "rt_sort.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

extern "C" int rt_sort_pairs_u64_u32(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                                     const unsigned* values_in, unsigned* values_out, int n, void* stream) {
  return (int)hipcub::DeviceRadixSort::SortPairs(temp, *temp_bytes, keys_in, keys_out, values_in, values_out, n, 0, 63,
                                                 (hipStream_t)stream);
}

extern "C" int rt_exclusive_sum_u32(void* temp, size_t* temp_bytes, const unsigned* in, unsigned* out, int n, void* stream) {
  return (int)hipcub::DeviceScan::ExclusiveSum(temp, *temp_bytes, in, out, n, (hipStream_t)stream);
}
