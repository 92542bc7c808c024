// Segmentation head + MC-dropout reduction in one pass (half-precision path):
//   logits_t = W x_t + b                         (SalsaNext.logits, a 1x1 conv over the 32-channel decoder output; SalsaNext.py:213)
//   p_t      = exp(log_softmax(logits_t));  p_bar = mean_t p_t;  H = -sum clamp(p_bar) log clamp(p_bar) / ln C
//   MI       = max((H_bar - mean_t H[p_t]) / ln C, 0);  preds = argmax_c p_bar          (trainer.py:1143-1154)
// Unfused, the head writes T*B fp32 logit maps (80 B per pixel and pass) that the reduction kernel reads back; here a wave
// owns a block of 32 pixels of one scan, walks its T passes (two 16-byte loads per lane and pass, two MFMAs against the
// weight fragments it keeps in registers) and keeps the softmax statistics in registers.  The 32x32 MFMA result leaves the
// classes of a pixel in TWO lanes (lane l and l ^ 32 hold classes 8q + 4h + k, h = l >> 5), so the per-pixel max / sum /
// entropy / argmax reductions finish with one cross-half shuffle each.  fp32 throughout after the fp16 x fp16 products.
#include <math.h>
#include "slu_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

template <int NKS>
__global__ __launch_bounds__(256) void head_mc_h8_kernel(const uint4* __restrict__ x, int T, int B, int HW, const uint4* __restrict__ wpack,
                                                        const float* __restrict__ bias, int C, float eps, float lnC, float* __restrict__ p_bar,
                                                        float* __restrict__ h_norm, float* __restrict__ mi_norm, int64_t* __restrict__ preds) {
  constexpr int G = 2 * NKS;
  const int lane = threadIdx.x & 63, hh = lane >> 5, jj = lane & 31;
  const long long nblk = (long long)B * (HW / 32);
  const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
  half8 af[NKS];
#pragma unroll
  for (int k = 0; k < NKS; ++k) af[k] = __builtin_bit_cast(half8, wpack[k * 64 + lane]);
  float bs[16];
  bool ok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = 8 * (r >> 2) + 4 * hh + (r & 3);
    ok[r] = c < C;
    bs[r] = (ok[r] && bias) ? bias[c] : 0.0f;
  }
  const float invT = 1.0f / (float)T, eps_log_eps = eps > 0.0f ? eps * logf(eps) : 0.0f;
  for (long long blk = wave0; blk < nblk; blk += nwave) {
    const int b = (int)(blk / (HW / 32));
    const size_t pix = (size_t)(blk - (long long)b * (HW / 32)) * 32 + jj;
    float psum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) psum[r] = 0.0f;
    float hsum = 0.0f;                                   // this lane's share of sum_t H[p_t]
    for (int t = 0; t < T; ++t) {
      const uint4* src = x + ((size_t)(t * B + b) * G + hh) * HW + pix;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int k = 0; k < NKS; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[k], __builtin_bit_cast(half8, src[(size_t)2 * k * HW]), acc, 0, 0, 0);
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[r] += bs[r];
        if (ok[r]) m = fmaxf(m, acc[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      // one exp per class and pass: p = e / sum(e), and log p = (z - m) - log(sum e) is already known (the clamp at eps, which the
      // reference applies before the log, only matters for p < eps: there the term is the constant eps log eps)
      float e[16], se = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        e[r] = ok[r] ? expf(acc[r] - m) : 0.0f;
        se += e[r];
      }
      se += __shfl_xor(se, 32, 64);
      const float lse = logf(se), rse = 1.0f / se;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (ok[r]) {
          const float p = e[r] * rse;
          psum[r] += p;
          hsum -= p >= eps ? p * (acc[r] - m - lse) : eps_log_eps;
        }
    }
    float hb = 0.0f, best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ok[r]) {
        const int c = 8 * (r >> 2) + 4 * hh + (r & 3);
        const float p = psum[r] * invT;
        p_bar[((size_t)b * C + c) * HW + pix] = p;
        if (p > best) { best = p; arg = c; }             // classes ascend with r: the first maximum of this lane's share
        const float pc = fmaxf(p, eps);
        hb -= pc * logf(pc);
      }
    hb += __shfl_xor(hb, 32, 64);
    hsum += __shfl_xor(hsum, 32, 64);
    const float ob = __shfl_xor(best, 32, 64);
    const int oa = __shfl_xor(arg, 32, 64);
    if (ob > best || (ob == best && oa < arg)) arg = oa;  // first maximum over all classes, like argmax
    if (hh == 0) {
      const size_t o = (size_t)b * HW + pix;
      h_norm[o] = hb / lnC;
      mi_norm[o] = fmaxf((hb - hsum * invT) / lnC, 0.0f);
      preds[o] = arg;
    }
  }
}

}  // namespace

extern "C" int slu_head_mc_h8(const void* x, int T, int B, int G, int HW, const void* wpack, const float* bias, int C, float eps, float* p_bar,
                              float* h_norm, float* mi_norm, int64_t* preds, slu_stream_t stream) {
  if (!x || !wpack || !p_bar || !h_norm || !mi_norm || !preds || T <= 0 || B <= 0 || G <= 0 || HW <= 0 || C <= 0) return SLU_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)wpack) & 15)) return SLU_EINVAL;
  if (C > 32 || HW % 32 || (G != 2 && G != 4 && G != 8)) return SLU_EUNSUPPORTED;
  const long long nblk = (long long)B * (HW / 32);
  long long nb = (nblk + 3) / 4;
  if (nb > 256 * 8) nb = 256 * 8;                                     // grid-stride: the chip sweeps the T*B images as one front per pass
  const float lnC = (float)log((double)C);
  auto xs = reinterpret_cast<const uint4*>(x);
  auto ws = reinterpret_cast<const uint4*>(wpack);
  hipStream_t st = slu_stream(stream);
  if (G == 2) hipLaunchKernelGGL(head_mc_h8_kernel<1>, dim3((unsigned)nb), dim3(256), 0, st, xs, T, B, HW, ws, bias, C, eps, lnC, p_bar, h_norm, mi_norm, preds);
  else if (G == 4) hipLaunchKernelGGL(head_mc_h8_kernel<2>, dim3((unsigned)nb), dim3(256), 0, st, xs, T, B, HW, ws, bias, C, eps, lnC, p_bar, h_norm, mi_norm, preds);
  else hipLaunchKernelGGL(head_mc_h8_kernel<4>, dim3((unsigned)nb), dim3(256), 0, st, xs, T, B, HW, ws, bias, C, eps, lnC, p_bar, h_norm, mi_norm, preds);
  SLU_CHECK_LAUNCH();
}
