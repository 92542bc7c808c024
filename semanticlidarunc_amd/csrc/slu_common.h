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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one of these per kernel instantiation remembers
// what has been granted on each device of the process (one process may drive several GPUs).  Benign race: the call is idempotent.
struct SluLdsGrant {
  size_t granted[32] = {};
};
static inline int slu_grant_dynamic_lds(const void* kern, size_t lds, SluLdsGrant& g) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return SLU_ELAUNCH;
  if (lds > g.granted[dev]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SLU_ELAUNCH;
    g.granted[dev] = lds;
  }
  return SLU_OK;
}

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
