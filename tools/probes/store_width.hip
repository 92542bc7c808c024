// What a mixed read + write stream sustains on this box as a function of the bytes a lane moves per instruction:
// a grid-stride copy of a 537 MB buffer (the size of a 32-channel 64-image full-resolution h8 tensor) with 16-byte loads
// and stores of 4 / 8 / 16 bytes per lane, the 8-byte form written the way the conv epilogues write (lanes l and l ^ 32
// own the two halves of a 16-byte record, a wave instruction covers 512 contiguous bytes).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/store_width.hip -o /tmp/store_width && /tmp/store_width
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>      // 0: 16 B loads + 16 B stores;  1: 16 B loads + 2 x 8 B stores (half-record pattern);  2: 16 B loads + 4 x 4 B stores;  3: stores only, 8 B half records;  4: stores only, 16 B
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nrec) {
  const int lane = threadIdx.x & 63, jj = lane & 31, hh = lane >> 5;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t base = wave * 64; base + 64 <= nrec; base += nwave * 64) {
    if (MODE == 0) {
      dst[base + lane] = src[base + lane];
    } else if (MODE == 1) {
      const uint4 v = src[base + lane];
      uint2* d = reinterpret_cast<uint2*>(dst);
      // two store instructions, each covering 32 records x one half: [rec jj][half hh] and [rec 32 + jj][half hh]
      d[(base + jj) * 2 + hh] = make_uint2(v.x, v.y);
      d[(base + 32 + jj) * 2 + hh] = make_uint2(v.z, v.w);
    } else if (MODE == 2) {
      const uint4 v = src[base + lane];
      unsigned* d = reinterpret_cast<unsigned*>(dst) + base * 4;
      d[lane] = v.x; d[64 + lane] = v.y; d[128 + lane] = v.z; d[192 + lane] = v.w;
    } else if (MODE == 3) {
      uint2* d = reinterpret_cast<uint2*>(dst);
      d[(base + jj) * 2 + hh] = make_uint2(lane, 1u);
      d[(base + 32 + jj) * 2 + hh] = make_uint2(lane, 2u);
    } else {
      dst[base + lane] = make_uint4(lane, 1u, 2u, 3u);
    }
  }
}

template <int MODE>
void run(const char* what, const uint4* src, uint4* dst, size_t nrec, double bytes) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(256 * 8), dim3(256), 0, 0, src, dst, nrec);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(copy_kernel<MODE>, dim3(256 * 8), dim3(256), 0, 0, src, dst, nrec);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %8.1f us  %6.2f TB/s\n", what, ms * 100.0, bytes * 10 / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t nrec = (size_t)537 * 1000 * 1000 / 16;
  uint4 *src, *dst;
  (void)hipMalloc(&src, nrec * 16);
  (void)hipMalloc(&dst, nrec * 16);
  (void)hipMemset(src, 1, nrec * 16);
  run<0>("copy, 16 B loads + 16 B stores (r + w)", src, dst, nrec, 2.0 * nrec * 16);
  run<1>("copy, 16 B loads + 8 B half-record stores (r + w)", src, dst, nrec, 2.0 * nrec * 16);
  run<2>("copy, 16 B loads + 4 B stores (r + w)", src, dst, nrec, 2.0 * nrec * 16);
  run<4>("fill, 16 B stores", src, dst, nrec, 1.0 * nrec * 16);
  run<3>("fill, 8 B half-record stores", src, dst, nrec, 1.0 * nrec * 16);
  return 0;
}
