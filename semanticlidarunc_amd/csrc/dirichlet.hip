// Dirichlet head of the reference's default ("Dirichlet") loss path, SURVEY row a15 (probability_helper.py:89-136,148-153):
//   alpha = 1 + softplus(scale / T) * softmax(shape) + eps            (to_alpha_concentrations_from_shape_and_scale)
//   alpha0 = sum_c alpha + eps;  p_hat = alpha / alpha0                (trainer.py:537-538)
//   H  = -sum_c p_hat log(p_hat + eps)                                 (get_predictive_entropy)
//   AU = -sum_c p_hat (digamma(alpha + 1) - digamma(alpha0 + 1))       (get_aleatoric_uncertainty);  EU = H - AU
// One lane owns one pixel (lanes of a wave are azimuth-adjacent: every load / store is a 256-byte row segment), the class
// axis lives in registers, every input byte is read once: HBM-bound, 4 (C+1) bytes in and up to 8 C + 16 bytes out per pixel.
#include "slu_common.h"

namespace {

// digamma for x >= 1 (alpha >= 1 always): recurrence up to x >= 6, then the asymptotic series; |err| < 2e-7 relative
__device__ __forceinline__ float digamma_ge1(float x) {
  float r = 0.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    if (x < 6.0f) {
      r -= 1.0f / x;
      x += 1.0f;
    }
  }
  const float inv = 1.0f / x, inv2 = inv * inv;
  return r + logf(x) - 0.5f * inv - inv2 * (1.0f / 12.0f - inv2 * (1.0f / 120.0f - inv2 * (1.0f / 252.0f)));
}

__device__ __forceinline__ float softplus_torch(float x) { return x > 20.0f ? x : log1pf(expf(x)); }   // beta = 1, threshold = 20

template <int CMAX, bool FROM_LOGITS>
__global__ __launch_bounds__(256) void dirichlet_kernel(const float* __restrict__ in, long long in_bs, const float* __restrict__ scale,
                                                        long long scale_bs, int B, int C, int HW, float inv_t, float eps,
                                                        float* __restrict__ alpha_out, float* __restrict__ p_hat, float* __restrict__ h,
                                                        float* __restrict__ au, int64_t* __restrict__ preds) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  const float* src = in + (size_t)b * in_bs + hw;
  float al[CMAX];
  if constexpr (FROM_LOGITS) {
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      al[c] = c < C ? src[(size_t)c * HW] : -INFINITY;
      m = fmaxf(m, al[c]);
    }
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        al[c] = expf(al[c] - m);
        se += al[c];
      }
    const float s = softplus_torch(scale[(size_t)b * scale_bs + hw] * inv_t);
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) al[c] = 1.0f + s * (al[c] / se) + eps;
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) al[c] = c < C ? src[(size_t)c * HW] : 0.0f;
  }
  float a0 = 0.0f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) a0 += al[c];
  a0 += eps;
  const float dg0 = au ? digamma_ge1(a0 + 1.0f) : 0.0f;
  float hh = 0.0f, aa = 0.0f, best = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      const float p = al[c] / a0;
      if (alpha_out) alpha_out[((size_t)b * C + c) * HW + hw] = al[c];
      if (p_hat) p_hat[((size_t)b * C + c) * HW + hw] = p;
      if (al[c] > best) { best = al[c]; arg = c; }
      hh -= p * logf(p + eps);
      if (au) aa -= p * (digamma_ge1(al[c] + 1.0f) - dg0);
    }
  if (h) h[pix] = hh;
  if (au) au[pix] = aa;
  if (preds) preds[pix] = arg;
}

template <bool FROM_LOGITS>
int launch(const float* in, long long in_bs, const float* scale, long long scale_bs, int B, int C, int HW, float inv_t, float eps, float* alpha,
           float* p_hat, float* h, float* au, int64_t* preds, hipStream_t st) {
  const size_t npix = (size_t)B * HW;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  if (C <= 20)
    hipLaunchKernelGGL((dirichlet_kernel<20, FROM_LOGITS>), dim3(nb), dim3(256), 0, st, in, in_bs, scale, scale_bs, B, C, HW, inv_t, eps, alpha, p_hat,
                       h, au, preds);
  else
    hipLaunchKernelGGL((dirichlet_kernel<32, FROM_LOGITS>), dim3(nb), dim3(256), 0, st, in, in_bs, scale, scale_bs, B, C, HW, inv_t, eps, alpha, p_hat,
                       h, au, preds);
  SLU_CHECK_LAUNCH();
}

}  // namespace

extern "C" int slu_dirichlet_head(const float* shape_logits, long long shape_batch_stride, const float* scale_logits, long long scale_batch_stride,
                                  int B, int C, int HW, float temperature, float eps, float* alpha, float* p_hat, float* entropy, float* aleatoric,
                                  int64_t* preds, slu_stream_t stream) {
  if (!shape_logits || !scale_logits || B <= 0 || C <= 0 || HW <= 0 || !(temperature > 0.0f)) return SLU_EINVAL;
  if (shape_batch_stride < (long long)C * HW || scale_batch_stride < HW) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  return launch<true>(shape_logits, shape_batch_stride, scale_logits, scale_batch_stride, B, C, HW, 1.0f / temperature, eps, alpha, p_hat, entropy,
                      aleatoric, preds, slu_stream(stream));
}

extern "C" int slu_dirichlet_uncertainty(const float* alpha, int B, int C, int HW, float eps, float* p_hat, float* entropy, float* aleatoric,
                                         int64_t* preds, slu_stream_t stream) {
  if (!alpha || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  return launch<false>(alpha, (long long)C * HW, nullptr, 0, B, C, HW, 1.0f, eps, nullptr, p_hat, entropy, aleatoric, preds, slu_stream(stream));
}
