// Per-pixel NLL / cross-entropy (forward + backward) and the softmax backward of the fused
// "SalsaNext" loss  (reference: src/models/trainer.py:508-516, src/models/losses.py:50-73).
// One lane per pixel, class axis in registers, lanes azimuth-adjacent: HBM-bound.
#include "slu_common.h"

namespace {

enum NllKind { kLogits = 0, kProbsClamp = 1, kProbsEps = 2, kLogProbs = 3 };

template <int CMAX, int KIND>
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int B, int C,
                                                      int HW, float param, int64_t ignore, double* __restrict__ nll_sum,
                                                      unsigned long long* __restrict__ count) {
  __shared__ double s_part[4];
  __shared__ unsigned s_cnt[4];
  const size_t npix = (size_t)B * HW;
  double local = 0.0;
  unsigned n = 0;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int64_t lab = labels[pix];
    if (lab == ignore || lab < 0 || lab >= C) continue;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)b * HW);
    const float* src = x + (size_t)b * C * (size_t)HW + hw;
    float v;
    if constexpr (KIND == kLogits) {
      float xs[CMAX];
      float m = -INFINITY, xy = 0.0f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        xs[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
        m = fmaxf(m, xs[c]);
        if ((int64_t)c == lab) xy = xs[c];
      }
      float se = 0.0f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) se += expf(xs[c] - m);
      v = (m + logf(se)) - xy;
    } else {
      const float t = src[(size_t)lab * HW];
      if constexpr (KIND == kProbsClamp) v = -logf(fmaxf(t, param));
      else if constexpr (KIND == kProbsEps) v = -logf(t + param);
      else v = -t;
    }
    local += (double)v;
    ++n;
  }
  local = wave_sum(local);
  const unsigned nw = (unsigned)wave_sum((float)n);
  if ((threadIdx.x & 63) == 0) { s_part[threadIdx.x >> 6] = local; s_cnt[threadIdx.x >> 6] = nw; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    unsigned c = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { t += s_part[w]; c += s_cnt[w]; }
    if (c) { atomicAdd(nll_sum, t); atomicAdd(count, (unsigned long long)c); }
  }
}

// grad_x = gscale[0] * d(sum of nll)/dx   (caller folds 1/count and the upstream gradient into gscale)
template <int CMAX, int KIND>
__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int B, int C,
                                                      int HW, float param, int64_t ignore,
                                                      const float* __restrict__ gscale, float* __restrict__ grad_x) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const float s = gscale[0];
  const int64_t lab = labels[pix];
  const bool valid = !(lab == ignore || lab < 0 || lab >= C);
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  const float* src = x + (size_t)b * C * (size_t)HW + hw;
  float* dst = grad_x + (size_t)b * C * (size_t)HW + hw;
  if constexpr (KIND == kLogits) {
    float xs[CMAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      xs[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
      m = fmaxf(m, xs[c]);
    }
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) { xs[c] = expf(xs[c] - m); se += xs[c]; }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) dst[(size_t)c * HW] = valid ? s * (xs[c] / se - ((int64_t)c == lab ? 1.0f : 0.0f)) : 0.0f;
  } else {
    const float t = valid ? src[(size_t)lab * HW] : 1.0f;
    float g = 0.0f;
    if (valid) {
      if constexpr (KIND == kProbsClamp) g = t >= param ? -s / t : 0.0f;
      else if constexpr (KIND == kProbsEps) g = -s / (t + param);
      else g = -s;
    }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) dst[(size_t)c * HW] = ((int64_t)c == lab) ? g : 0.0f;
  }
}

// grad_logits = softmax backward of  g_c = gout * ( w_dense * dense_c  -  [c == y && p_y >= clamp] * w_nll / p_y )
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_loss_bwd_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ dense, float w_dense, float w_nll,
                                                               float clampv, const float* __restrict__ gout, int B, int C, int HW,
                                                               float* __restrict__ grad_logits) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const float go = gout ? gout[0] : 1.0f;
  const int64_t lab = labels ? labels[pix] : -1;
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  const size_t base = (size_t)b * C * (size_t)HW + hw;
  float p[CMAX], g[CMAX];
  float dot = 0.0f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      p[c] = probs[base + (size_t)c * HW];
      float gc = dense ? w_dense * dense[base + (size_t)c * HW] : 0.0f;
      if ((int64_t)c == lab && p[c] >= clampv) gc -= w_nll / p[c];
      g[c] = gc * go;
      dot += g[c] * p[c];
    }
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) grad_logits[base + (size_t)c * HW] = p[c] * (g[c] - dot);
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

template <int KIND>
int launch_nll_fwd(const float* x, const int64_t* labels, int B, int C, int HW, float param, int64_t ign, double* acc,
                   unsigned long long* cnt, hipStream_t st) {
  const size_t nb = blocks_for((size_t)B * HW);
  const unsigned g = (unsigned)(nb > 2048 ? 2048 : nb);
  if (C <= 20)
    hipLaunchKernelGGL((nll_fwd_kernel<20, KIND>), dim3(g), dim3(256), 0, st, x, labels, B, C, HW, param, ign, acc, cnt);
  else
    hipLaunchKernelGGL((nll_fwd_kernel<32, KIND>), dim3(g), dim3(256), 0, st, x, labels, B, C, HW, param, ign, acc, cnt);
  SLU_CHECK_LAUNCH();
}

template <int KIND>
int launch_nll_bwd(const float* x, const int64_t* labels, int B, int C, int HW, float param, int64_t ign, const float* gscale,
                   float* grad_x, hipStream_t st) {
  const unsigned g = blocks_for((size_t)B * HW);
  if (C <= 20)
    hipLaunchKernelGGL((nll_bwd_kernel<20, KIND>), dim3(g), dim3(256), 0, st, x, labels, B, C, HW, param, ign, gscale, grad_x);
  else
    hipLaunchKernelGGL((nll_bwd_kernel<32, KIND>), dim3(g), dim3(256), 0, st, x, labels, B, C, HW, param, ign, gscale, grad_x);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_nll_fwd(const float* x, const int64_t* labels, int B, int C, int HW, int kind, float param, int64_t ignore_index,
                           double* nll_sum, int64_t* count, slu_stream_t stream) {
  if (!x || !labels || !nll_sum || !count || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(count);
  hipStream_t st = slu_stream(stream);
  switch (kind) {
    case kLogits: return launch_nll_fwd<kLogits>(x, labels, B, C, HW, param, ignore_index, nll_sum, cnt, st);
    case kProbsClamp: return launch_nll_fwd<kProbsClamp>(x, labels, B, C, HW, param, ignore_index, nll_sum, cnt, st);
    case kProbsEps: return launch_nll_fwd<kProbsEps>(x, labels, B, C, HW, param, ignore_index, nll_sum, cnt, st);
    case kLogProbs: return launch_nll_fwd<kLogProbs>(x, labels, B, C, HW, param, ignore_index, nll_sum, cnt, st);
  }
  return SLU_EUNSUPPORTED;
}

extern "C" int slu_nll_bwd(const float* x, const int64_t* labels, int B, int C, int HW, int kind, float param, int64_t ignore_index,
                           const float* gscale, float* grad_x, slu_stream_t stream) {
  if (!x || !labels || !gscale || !grad_x || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  hipStream_t st = slu_stream(stream);
  switch (kind) {
    case kLogits: return launch_nll_bwd<kLogits>(x, labels, B, C, HW, param, ignore_index, gscale, grad_x, st);
    case kProbsClamp: return launch_nll_bwd<kProbsClamp>(x, labels, B, C, HW, param, ignore_index, gscale, grad_x, st);
    case kProbsEps: return launch_nll_bwd<kProbsEps>(x, labels, B, C, HW, param, ignore_index, gscale, grad_x, st);
    case kLogProbs: return launch_nll_bwd<kLogProbs>(x, labels, B, C, HW, param, ignore_index, gscale, grad_x, st);
  }
  return SLU_EUNSUPPORTED;
}

extern "C" int slu_softmax_loss_bwd(const float* probs, const int64_t* labels, const float* dense, float w_dense, float w_nll,
                                    float clampv, const float* gout, int B, int C, int HW, float* grad_logits, slu_stream_t stream) {
  if (!probs || !grad_logits || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const unsigned g = blocks_for((size_t)B * HW);
  if (C <= 20)
    hipLaunchKernelGGL(softmax_loss_bwd_kernel<20>, dim3(g), dim3(256), 0, slu_stream(stream), probs, labels, dense, w_dense, w_nll, clampv, gout, B, C, HW, grad_logits);
  else
    hipLaunchKernelGGL(softmax_loss_bwd_kernel<32>, dim3(g), dim3(256), 0, slu_stream(stream), probs, labels, dense, w_dense, w_nll, clampv, gout, B, C, HW, grad_logits);
  SLU_CHECK_LAUNCH();
}
