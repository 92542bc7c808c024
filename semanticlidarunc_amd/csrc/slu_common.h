// Shared helpers for the libslu_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "slu.h"

#define SLU_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return SLU_ELAUNCH; \
    return SLU_OK;                                           \
  } while (0)

static inline hipStream_t slu_stream(slu_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// 64-lane wavefront sum (all lanes receive the total).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
