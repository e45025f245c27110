// (key, value) radix sort on a stream (pt_sort.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

size_t pt_sort_temp_bytes(size_t n, int begin_bit, int end_bit);
hipError_t pt_sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                         uint32_t* vals_out, size_t n, int begin_bit, int end_bit, hipStream_t stream);
