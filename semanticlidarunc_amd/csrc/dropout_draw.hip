// All Dropout2d multipliers of one MC-dropout evaluation in ONE launch (reference nn.Dropout2d sites: baselines/SalsaNext/SalsaNext.py:98,106,
// 145,149,168; utils/mc_dropout.py:13-34 toggles them).  Dropout2d zeroes whole (sample, channel) planes with probability p and scales the
// survivors by 1 / (1 - p), so a site is an [N, C] multiplier; consecutive sites of the U-Net's decoder reach a conv as PRODUCTS of such
// multipliers (UpBlock: the producer's deferred dropout3 x PixelShuffle'd dropout1 x dropout2, SalsaNext.py:141-149).  The draws are a
// stateless function of (seed, offset, site, sample, channel) -- Philox4x32-10 keyed with the seed and counted by offset + element / 4, the
// construction torch's own CUDA generator uses -- so every output element computes the (at most three) draws it depends on by itself and
// no pass over intermediate masks exists.  The host takes seed / offset from torch's CUDA generator and advances the offset.
#include "slu_common.h"

namespace {

struct uint4x { unsigned x, y, z, w; };

__device__ __forceinline__ uint4x philox4x32_10(unsigned long long seed, unsigned long long counter, unsigned long long subseq) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)counter, c1 = (unsigned)(counter >> 32), c2 = (unsigned)subseq, c3 = (unsigned)(subseq >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return uint4x{c0, c1, c2, c3};
}

// multiplier of (site, element e of its [N][C] table): 0 with probability p, else 1 / (1 - p); 1 for an inactive site
__device__ __forceinline__ float site_mult(const slu_dropout_site* sites, int site, long long e, unsigned long long seed, unsigned long long offset) {
  const slu_dropout_site s = sites[site];
  if (!s.active) return 1.0f;
  const long long g = s.begin + e;                     // position in the global element order of this call's draws
  const uint4x r = philox4x32_10(seed, offset + (unsigned long long)(g >> 2), 0ull);
  const unsigned bits = (g & 3) == 0 ? r.x : ((g & 3) == 1 ? r.y : ((g & 3) == 2 ? r.z : r.w));
  const float u = (float)(bits >> 8) * (1.0f / 16777216.0f);      // uniform in [0, 1)
  return u < s.p ? 0.0f : 1.0f / (1.0f - s.p);
}

__global__ __launch_bounds__(256) void dropout_draw_kernel(const slu_dropout_site* __restrict__ sites, const slu_dropout_out* __restrict__ outs, int nout,
                                                           int N, unsigned long long seed, unsigned long long offset, float* __restrict__ buf,
                                                           long long total) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    int o = 0;
    while (o + 1 < nout && e >= outs[o + 1].begin) ++o;         // (a handful of outputs: linear search)
    const slu_dropout_out d = outs[o];
    const long long r = e - d.begin;
    const int c = (int)(r % d.C);
    const long long n = r / d.C;
    float v = 1.0f;
    if (d.site_a >= 0) v *= site_mult(sites, d.site_a, n * sites[d.site_a].C + d.off_a + c, seed, offset);
    // b and c act on the PixelShuffle'd tensor: stored channel c feeds shuffled channel c / 4
    if (d.site_b >= 0) v *= site_mult(sites, d.site_b, n * sites[d.site_b].C + d.off_b + (d.shuffled ? c / 4 : c), seed, offset);
    if (d.site_c >= 0) v *= site_mult(sites, d.site_c, n * sites[d.site_c].C + d.off_c + (d.shuffled ? c / 4 : c), seed, offset);
    buf[e] = v;
  }
}

}  // namespace

extern "C" int slu_dropout_draw(const slu_dropout_site* sites, int nsites, const slu_dropout_out* outs, int nout, int N, unsigned long long seed,
                                unsigned long long offset, float* buf, long long total, slu_stream_t stream) {
  if (!sites || !outs || !buf || nsites <= 0 || nout <= 0 || N <= 0 || total <= 0) return SLU_EINVAL;
  const long long nb = (total + 255) / 256;
  hipLaunchKernelGGL(dropout_draw_kernel, dim3((unsigned)(nb > 4096 ? 4096 : nb)), dim3(256), 0, slu_stream(stream), sites, outs, nout, N, seed, offset,
                     buf, total);
  SLU_CHECK_LAUNCH();
}
