// Tversky loss of the reference (src/models/losses.py:74-128; the 'Tversky' loss branch of trainer.py:497-503).
//   valid = 0 <= y < C and y != ignore;   p = softmax(x) | x | exp(x)  by model_act;   over valid pixels, per class c:
//   TP_c = sum p_c [y = c],  FP_c = sum p_c [y != c] = S_c - TP_c,  FN_c = sum (1 - p_c) [y = c] = N_c - TP_c
//   tversky_c = (TP_c + s) / (TP_c + alpha FP_c + beta FN_c + s),   loss = reduce_c (1 - tversky_c)
// Forward: one pass, per-workgroup LDS partial sums of (S_c, TP_c, N_c), fp64 atomics.  Backward: one pass,
//   d tversky_c / d p_c,i = ([y_i = c] D_c - A_c (alpha + [y_i = c] (1 - alpha - beta))) / D_c^2   (A = TP + s, D = denominator),
// chained through softmax / exp for logits / log-probs.  One lane per pixel, class axis in registers: HBM-bound.
#include "slu_common.h"

namespace {

enum { kActLogits = 0, kActProbs = 1, kActLogProbs = 2 };

template <int CMAX>
__device__ __forceinline__ void load_probs(const float* src, int C, size_t HW, int act, float (&p)[CMAX]) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    p[c] = c < C ? src[(size_t)c * HW] : (act == kActLogits ? -INFINITY : 0.0f);
    m = fmaxf(m, p[c]);
  }
  if (act == kActLogits) {
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) { p[c] = expf(p[c] - m); se += p[c]; }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = c < C ? p[c] / se : 0.0f;
  } else if (act == kActLogProbs) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = c < C ? expf(p[c]) : 0.0f;
  }
}

template <int CMAX>
__global__ __launch_bounds__(256) void tversky_sums_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int B, int C, int HW, int act,
                                                           int has_ignore, int64_t ignore, double* __restrict__ sums /* [3][C]: S, TP, N */) {
  __shared__ float s_s[32], s_tp[32];
  __shared__ unsigned s_n[32];
  if (threadIdx.x < 32) { s_s[threadIdx.x] = 0.0f; s_tp[threadIdx.x] = 0.0f; s_n[threadIdx.x] = 0u; }
  __syncthreads();
  const size_t npix = (size_t)B * HW;
  float ls[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) ls[c] = 0.0f;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int64_t y = labels[pix];
    if (!(y >= 0 && y < C) || (has_ignore && y == ignore)) continue;
    const int b = (int)(pix / HW);
    float p[CMAX];
    load_probs<CMAX>(x + (size_t)b * C * HW + (pix - (size_t)b * HW), C, (size_t)HW, act, p);
    float py = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      ls[c] += p[c];
      py = (c == (int)y) ? p[c] : py;
    }
    atomicAdd(&s_tp[(int)y], py);
    atomicAdd(&s_n[(int)y], 1u);
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    const float t = wave_sum(ls[c]);
    if (c < C && (threadIdx.x & 63) == 0 && t != 0.0f) atomicAdd(&s_s[c], t);
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    if (s_s[threadIdx.x] != 0.0f) atomicAdd(&sums[threadIdx.x], (double)s_s[threadIdx.x]);
    if (s_n[threadIdx.x]) {
      atomicAdd(&sums[C + threadIdx.x], (double)s_tp[threadIdx.x]);
      atomicAdd(&sums[2 * C + threadIdx.x], (double)s_n[threadIdx.x]);
    }
  }
}

// loss value and the per-class backward coefficients: coef[0][c] = A_c / D_c^2, coef[1][c] = 1 / D_c  (both times -w_c)
__global__ void tversky_finalize_kernel(const double* __restrict__ sums, int C, float alpha, float beta, float smooth, int reduction,
                                        float* __restrict__ loss /* [1] or [C] */, float* __restrict__ coef /* [2][C] */, float* __restrict__ any_valid) {
  if (threadIdx.x != 0) return;
  double total = 0.0, nvalid = 0.0;
  for (int c = 0; c < C; ++c) {
    const double S = sums[c], TP = sums[C + c], N = sums[2 * C + c];
    nvalid += N;
    const double A = TP + smooth, D = TP + alpha * (S - TP) + beta * (N - TP) + smooth;
    const double lc = 1.0 - A / D;
    total += lc;
    if (reduction == 2) loss[c] = (float)lc;
    const double w = reduction == 0 ? 1.0 / C : 1.0;           // d loss / d loss_c for mean / sum / none (none: scaled by grad_out[c] later)
    coef[c] = (float)(w * A / (D * D));
    coef[C + c] = (float)(w / D);
  }
  *any_valid = nvalid > 0.0 ? 1.0f : 0.0f;
  if (reduction == 0) loss[0] = nvalid > 0.0 ? (float)(total / C) : 0.0f;      // "no valid pixels: zero loss" (losses.py:101-103)
  else if (reduction == 1) loss[0] = nvalid > 0.0 ? (float)total : 0.0f;
  else if (nvalid == 0.0) for (int c = 0; c < C; ++c) loss[c] = 0.0f;
}

template <int CMAX>
__global__ __launch_bounds__(256) void tversky_bwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels, int B, int C, int HW, int act,
                                                          int has_ignore, int64_t ignore, float alpha, float beta, const float* __restrict__ coef,
                                                          const float* __restrict__ gout, int gout_per_class, float* __restrict__ grad_x) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / HW);
  const size_t hw = pix - (size_t)b * HW;
  float* dst = grad_x + (size_t)b * C * HW + hw;
  const int64_t y = labels[pix];
  const bool valid = (y >= 0 && y < C) && !(has_ignore && y == ignore);
  if (!valid) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) dst[(size_t)c * HW] = 0.0f;
    return;
  }
  float p[CMAX], g[CMAX];
  load_probs<CMAX>(x + (size_t)b * C * HW + hw, C, (size_t)HW, act, p);
  float dot = 0.0f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    if (c < C) {
      const float go = gout ? gout[gout_per_class ? c : 0] : 1.0f;
      const bool fg = c == (int)y;
      // d loss / d p_c = -w ( [fg] / D - A (alpha + [fg] (1 - alpha - beta)) / D^2 )
      g[c] = -go * ((fg ? coef[C + c] : 0.0f) - coef[c] * (alpha + (fg ? 1.0f - alpha - beta : 0.0f)));
      dot += g[c] * p[c];
    } else {
      g[c] = 0.0f;
    }
  }
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) {
      float r = g[c];
      if (act == kActLogits) r = p[c] * (g[c] - dot);
      else if (act == kActLogProbs) r = g[c] * p[c];
      dst[(size_t)c * HW] = r;
    }
}

}  // namespace

extern "C" int slu_tversky_fwd(const float* x, const int64_t* labels, int B, int C, int HW, int model_act, int has_ignore, int64_t ignore_index,
                               float alpha, float beta, float smooth, int reduction, double* sums, float* coef, float* loss, float* any_valid,
                               slu_stream_t stream) {
  if (!x || !labels || !sums || !coef || !loss || !any_valid || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (model_act < 0 || model_act > 2 || reduction < 0 || reduction > 2) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  hipStream_t st = slu_stream(stream);
  if (hipMemsetAsync(sums, 0, (size_t)3 * C * sizeof(double), st) != hipSuccess) return SLU_ELAUNCH;
  const size_t npix = (size_t)B * HW;
  size_t nb = (npix + 255) / 256;
  if (nb > 2048) nb = 2048;
  if (C <= 20)
    hipLaunchKernelGGL(tversky_sums_kernel<20>, dim3((unsigned)nb), dim3(256), 0, st, x, labels, B, C, HW, model_act, has_ignore, ignore_index, sums);
  else
    hipLaunchKernelGGL(tversky_sums_kernel<32>, dim3((unsigned)nb), dim3(256), 0, st, x, labels, B, C, HW, model_act, has_ignore, ignore_index, sums);
  hipLaunchKernelGGL(tversky_finalize_kernel, dim3(1), dim3(64), 0, st, sums, C, alpha, beta, smooth, reduction, loss, coef, any_valid);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_tversky_bwd(const float* x, const int64_t* labels, int B, int C, int HW, int model_act, int has_ignore, int64_t ignore_index,
                               float alpha, float beta, const float* coef, const float* grad_out, int grad_out_per_class, float* grad_x,
                               slu_stream_t stream) {
  if (!x || !labels || !coef || !grad_x || B <= 0 || C <= 0 || HW <= 0 || model_act < 0 || model_act > 2) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const size_t npix = (size_t)B * HW;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  if (C <= 20)
    hipLaunchKernelGGL(tversky_bwd_kernel<20>, dim3(nb), dim3(256), 0, slu_stream(stream), x, labels, B, C, HW, model_act, has_ignore, ignore_index,
                       alpha, beta, coef, grad_out, grad_out_per_class, grad_x);
  else
    hipLaunchKernelGGL(tversky_bwd_kernel<32>, dim3(nb), dim3(256), 0, slu_stream(stream), x, labels, B, C, HW, model_act, has_ignore, ignore_index,
                       alpha, beta, coef, grad_out, grad_out_per_class, grad_x);
  SLU_CHECK_LAUNCH();
}
