// Register-resident v_mfma_f32_32x32x16_f16 loop: what the matrix cores sustain on this box (clock under load included).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  half8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(threadIdx.x * 0.001f + k); b[k] = (_Float16)(k * 0.5f - threadIdx.x * 0.002f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.0f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 20000;
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 1000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 16 * (double)NACC * iters * blocks * 4;
  printf("NACC=%d waves/SIMD=%d: %.1f ms  %.1f TFLOP/s\n", NACC, waves_per_simd, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  run<4>(1); run<4>(2); run<8>(1); run<8>(2); run<2>(2); run<4>(4);
  return 0;
}
