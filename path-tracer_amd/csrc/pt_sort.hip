// Radix sort of (key, value) pairs of 32-bit words on a stream - the shadow queue's order by light-grid cell
// (pt_gpu.hip, shadow stage).  A translation unit of its own: rocPRIM's device sort is a few hundred template
// instantiations that the integrator's kernels need not be recompiled with.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "pt_sort.h"

size_t pt_sort_temp_bytes(size_t n, int begin_bit, int end_bit) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, n, begin_bit, end_bit);
    return bytes;
}

hipError_t pt_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                         uint32_t* vals_out, size_t n, int begin_bit, int end_bit, hipStream_t stream) {
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
