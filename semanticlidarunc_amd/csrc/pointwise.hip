// HBM-bound per-pixel kernels: BatchNorm fold, 3x3/s2 average pool, MC-dropout reduction
// (softmax over C, running mean over T, predictive entropy, mutual information, argmax) and the
// single-pass softmax/entropy map.  One lane owns one pixel; lanes of a wave are azimuth-adjacent,
// so every load/store instruction is a 256-byte contiguous row segment; the class axis (C <= 32,
// 20 for SemanticKITTI) lives in registers, never in LDS.
#include "slu_common.h"

namespace {

__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, int C,
                               float* __restrict__ a, float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float s = gamma[c] / sqrtf(var[c] + eps);
    a[c] = s;
    b[c] = beta[c] - mean[c] * s;
  }
}

// y[n,c,oy,ox] = (1/9) sum_{i,j in -1..1} s * x[n,c,2oy+i,2ox+j]  (zeros outside; divisor always 9)
__global__ void avgpool3s2_kernel(const float* __restrict__ x, const float* __restrict__ scale, float* __restrict__ y,
                                  int NC, int H, int W, int OH, int OW, int C, int in_batch) {
  const size_t total = (size_t)NC * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t nc = r / OH;
    const float s = scale ? scale[nc] : 1.0f;
    const size_t nc_in = in_batch ? ((nc / C) % in_batch) * C + nc % C : nc;   // broadcast source image
    const float* p = x + nc_in * (size_t)H * W;
    float acc = 0.0f;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      const int iy = 2 * oy + i;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const int ix = 2 * ox + j;
        if (ix < 0 || ix >= W) continue;
        acc += p[(size_t)iy * W + ix] * s;
      }
    }
    y[e] = acc / 9.0f;
  }
}

// probs_t = exp(log_softmax(x_t)); accumulates over t in registers; never materialises [T,B,C,HW].
template <int CMAX>
__global__ __launch_bounds__(256) void mc_reduce_kernel(const float* __restrict__ logits, int T, int B, int C, int HW,
                                                        float eps, float lnC, float* __restrict__ p_bar,
                                                        float* __restrict__ h_norm, float* __restrict__ mi_norm,
                                                        int64_t* __restrict__ preds) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  float psum[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) psum[c] = 0.0f;
  float hsum = 0.0f;
  for (int t = 0; t < T; ++t) {
    const float* src = logits + ((size_t)t * B + b) * C * (size_t)HW + hw;
    float x[CMAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      x[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
      m = fmaxf(m, x[c]);
    }
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) se += expf(x[c] - m);
    const float lse = logf(se);
    float ht = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float p = expf(x[c] - m - lse);
        psum[c] += p;
        const float pc = fmaxf(p, eps);
        ht -= pc * logf(pc);
      }
    hsum += ht;
  }
  float hb = 0.0f, best = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      const float p = psum[c] / (float)T;
      p_bar[((size_t)b * C + c) * HW + hw] = p;
      if (p > best) { best = p; arg = c; }
      const float pc = fmaxf(p, eps);
      hb -= pc * logf(pc);
    }
  h_norm[pix] = hb / lnC;
  mi_norm[pix] = fmaxf((hb - hsum / (float)T) / lnC, 0.0f);
  preds[pix] = arg;
}

template <int CMAX>
__global__ __launch_bounds__(256) void softmax_entropy_kernel(const float* __restrict__ logits, int B, int C, int HW,
                                                              float eps, float lnC, float* __restrict__ probs,
                                                              float* __restrict__ h_norm, int64_t* __restrict__ preds) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  const float* src = logits + (size_t)b * C * (size_t)HW + hw;
  float x[CMAX];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    x[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
    m = fmaxf(m, x[c]);
  }
  float se = 0.0f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) se += expf(x[c] - m);
  const float lse = logf(se);
  float h = 0.0f, best = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      const float p = expf(x[c] - m - lse);
      if (probs) probs[((size_t)b * C + c) * HW + hw] = p;
      if (p > best) { best = p; arg = c; }
      h -= p * logf(fmaxf(p, eps));
    }
  if (h_norm) h_norm[pix] = h / lnC;
  if (preds) preds[pix] = arg;
}

}  // namespace

extern "C" int slu_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C,
                           float* a, float* b, slu_stream_t stream) {
  if (!gamma || !beta || !mean || !var || !a || !b || C <= 0) return SLU_EINVAL;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, slu_stream(stream), gamma, beta, mean, var, eps, C, a, b);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_avgpool3s2_fwd(const float* x, const float* scale, float* y, int N, int C, int H, int W,
                                  slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const size_t total = (size_t)N * C * OH * OW;
  const size_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(avgpool3s2_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, slu_stream(stream), x, scale, y,
                     N * C, H, W, OH, OW, C, 0);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_avgpool3s2_bcast_fwd(const float* x, const float* scale, float* y, int N, int in_batch, int C, int H, int W,
                                        slu_stream_t stream) {
  if (!x || !y || N <= 0 || in_batch <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const size_t total = (size_t)N * C * OH * OW;
  const size_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(avgpool3s2_kernel, dim3((unsigned)(nb > 16384 ? 16384 : nb)), dim3(256), 0, slu_stream(stream), x, scale, y,
                     N * C, H, W, OH, OW, C, in_batch);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_mc_reduce(const float* logits, int T, int B, int C, int HW, float eps, float* p_bar, float* h_norm,
                             float* mi_norm, int64_t* preds, slu_stream_t stream) {
  if (!logits || !p_bar || !h_norm || !mi_norm || !preds || T <= 0 || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const size_t npix = (size_t)B * HW;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  const float lnC = (float)log((double)C);
  if (C <= 20)
    hipLaunchKernelGGL(mc_reduce_kernel<20>, dim3(nb), dim3(256), 0, slu_stream(stream), logits, T, B, C, HW, eps, lnC, p_bar,
                       h_norm, mi_norm, preds);
  else
    hipLaunchKernelGGL(mc_reduce_kernel<32>, dim3(nb), dim3(256), 0, slu_stream(stream), logits, T, B, C, HW, eps, lnC, p_bar,
                       h_norm, mi_norm, preds);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_softmax_entropy(const float* logits, int B, int C, int HW, float eps, float* probs, float* h_norm,
                                   int64_t* preds, slu_stream_t stream) {
  if (!logits || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const size_t npix = (size_t)B * HW;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  const float lnC = (float)log((double)C);
  if (C <= 20)
    hipLaunchKernelGGL(softmax_entropy_kernel<20>, dim3(nb), dim3(256), 0, slu_stream(stream), logits, B, C, HW, eps, lnC, probs,
                       h_norm, preds);
  else
    hipLaunchKernelGGL(softmax_entropy_kernel<32>, dim3(nb), dim3(256), 0, slu_stream(stream), logits, B, C, HW, eps, lnC, probs,
                       h_norm, preds);
  SLU_CHECK_LAUNCH();
}
