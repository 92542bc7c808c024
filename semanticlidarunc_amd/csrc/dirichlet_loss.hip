// Per-pixel Dirichlet losses of the reference's default ("Dirichlet") loss path, SURVEY section 8(f-1): each is the mean over the
// valid pixels (label != ignore_index) of a function of alpha[:, pixel] and the label, forward and d/d alpha in one pass each.
//   kind 0  NLLDirichletCategorical  losses/dirichlet_losses.py:73-119    -(log(a_y + eps) - log(a0 + eps))
//   kind 1  DigammaDirichletCE       :122-167                             psi(a0) - psi(a_y)
//   kind 2  BrierDirichlet           :174-221                             sum_i E[p_i^2] - 2 p_y + 1   (param = s_ref, < 0: use a0)
//   kind 3  DirichletMSELoss         :317-385                             sum_c (y_c - p_c)^2 + alpha_c (a0 - alpha_c) / ((a0^2 + eps)(a0 + 1))
//   kind 4  KL_offClasses_to_uniform losses/regularizers.py:291-389       KL(Dir(alpha~) || Dir(1)), alpha~ = alpha with the true class set to 1
//   kind 5  ComplementKLUniform      losses/dirichlet_losses.py:228-314   w(p_y) KL(p_off / (1 - p_y) || U) [/ ln(C-1)]; params gamma, tau, sigma,
//                                                                         s_target (< 0: none), normalize, detach_uncert
//   kind 7  KL_offClasses_to_uniform(with_conf_weighting=True)  :375-385     kind 4 times the detached weight (1 - p_y)^gamma, averaged over
//                                                                         max(sum of the weights, 1); param gamma
//   kind 6  WrongLowEvidence         losses/regularizers.py:218-289       gate * relu(ln a0 - ln(C + s_low + eps))^2 on wrong pixels, averaged over
//                                                                         sum(gate) (second accumulator); params s_low, margin, soft_margin_k
// One lane per pixel, class axis in registers, fp32 per pixel, fp64 sums: HBM-bound (4 C bytes in; backward 4 C in + 4 C out).
#include "slu_common.h"

namespace {

__device__ __forceinline__ float digamma_pos(float x) {       // x > 0: recurrence to x >= 6, then the asymptotic series
  float r = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (x < 6.0f) { r -= 1.0f / x; x += 1.0f; }
  const float inv = 1.0f / x, inv2 = inv * inv;
  return r + logf(x) - 0.5f * inv - inv2 * (1.0f / 12.0f - inv2 * (1.0f / 120.0f - inv2 * (1.0f / 252.0f)));
}

__device__ __forceinline__ float trigamma_pos(float x) {      // psi'(x) = psi'(x + 1) + 1 / x^2; asymptotic series for x >= 6
  float r = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (x < 6.0f) { r += 1.0f / (x * x); x += 1.0f; }
  const float inv = 1.0f / x, inv2 = inv * inv;
  return r + inv * (1.0f + 0.5f * inv + inv2 * (1.0f / 6.0f - inv2 * (1.0f / 30.0f - inv2 * (1.0f / 42.0f))));
}

struct LossParams { float p[6]; };

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int CMAX, bool BWD>
__global__ __launch_bounds__(256) void dirichlet_loss_kernel(const float* __restrict__ alpha, const int64_t* __restrict__ labels, int B, int C, int HW,
                                                             int kind, LossParams prm, float eps, int has_ignore, int64_t ignore,
                                                             double* __restrict__ sum, double* __restrict__ sum2, unsigned long long* __restrict__ count,
                                                             const float* __restrict__ gscale, float* __restrict__ grad) {
  const float param = prm.p[0];
  __shared__ double s_sum[4], s_sum2[4];
  __shared__ unsigned s_cnt[4];
  const size_t npix = (size_t)B * HW;
  double lsum = 0.0, lsum2 = 0.0;
  unsigned lcnt = 0;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(pix / HW);
    const size_t hw = pix - (size_t)b * HW;
    const int64_t y64 = labels[pix];
    const bool valid = !(has_ignore && y64 == ignore) && y64 >= 0 && y64 < C;
    float* dst = BWD ? grad + (size_t)b * C * HW + hw : nullptr;
    if (!valid) {
      if constexpr (BWD) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) dst[(size_t)c * HW] = 0.0f;
      }
      continue;
    }
    const int y = (int)y64;
    const float* src = alpha + (size_t)b * C * HW + hw;
    float a[CMAX];
    float a0 = 0.0f, ay = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      a[c] = c < C ? src[(size_t)c * HW] : 0.0f;
      a0 += a[c];
      ay = (c == y) ? a[c] : ay;
    }
    const float gs = BWD ? gscale[0] : 0.0f;
    float v = 0.0f;
    if (kind == 0) {
      v = -(logf(ay + eps) - logf(a0 + eps));
      if constexpr (BWD) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) dst[(size_t)c * HW] = gs * (1.0f / (a0 + eps) - (c == y ? 1.0f / (ay + eps) : 0.0f));
      }
    } else if (kind == 1) {
      v = digamma_pos(a0) - digamma_pos(ay);
      if constexpr (BWD) {
        const float t0 = trigamma_pos(a0), ty = trigamma_pos(ay);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) dst[(size_t)c * HW] = gs * (t0 - (c == y ? ty : 0.0f));
      }
    } else if (kind == 2) {
      const float d = a0 + eps;
      float q = 0.0f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) q += (a[c] / d) * (a[c] / d);
      const float py = ay / d;
      const bool ref = param >= 0.0f;
      const float s = ref ? param : a0;
      v = (s * q + 1.0f) / (s + 1.0f) - 2.0f * py + 1.0f;
      if constexpr (BWD) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            const float pc = a[c] / d;
            const float dq = 2.0f * (pc - q) / d;
            const float df = ref ? s / (s + 1.0f) * dq : (q - 1.0f + a0 * (a0 + 1.0f) * dq) / ((a0 + 1.0f) * (a0 + 1.0f));
            dst[(size_t)c * HW] = gs * (df - 2.0f * ((c == y ? 1.0f : 0.0f) - py) / d);
          }
      }
    } else if (kind == 3) {
      const float d = a0 + eps;
      const float G = (a0 * a0 + eps) * (a0 + 1.0f);
      float sq = 0.0f, nn = 0.0f, cross = 0.0f;       // sum (y - p)^2, sum alpha (a0 - alpha), sum (y - p) p
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
          const float pc = a[c] / d, e = (c == y ? 1.0f : 0.0f) - pc;
          sq += e * e;
          nn += a[c] * (a0 - a[c]);
          cross += e * pc;
        }
      v = sq + nn / G;
      if constexpr (BWD) {
        const float dG = 2.0f * a0 * (a0 + 1.0f) + (a0 * a0 + eps);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            const float pc = a[c] / d, e = (c == y ? 1.0f : 0.0f) - pc;
            // d nn / d alpha_c = sum_j d[alpha_j (a0 - alpha_j)] = (a0 - alpha_c) - alpha_c + sum_{j} alpha_j = 2 a0 - 2 alpha_c
            dst[(size_t)c * HW] = gs * (-2.0f * (e - cross) / d + ((2.0f * a0 - 2.0f * a[c]) * G - nn * dG) / (G * G));
          }
      }
    } else if (kind == 5) {
      const float gamma = prm.p[0], tau = prm.p[1], sigma = prm.p[2], s_t = prm.p[3];
      const bool normalize = prm.p[4] != 0.0f, detach = prm.p[5] != 0.0f;
      if (C > 2) {
        const float d = a0 + eps;                                  // a0 of the reference (eps included)
        const float py_raw = ay / d, py = fmaxf(py_raw, eps);
        const float den_raw = 1.0f - py, den = fmaxf(den_raw, eps);
        const float lc1 = logf((float)(C - 1));
        float kl = lc1, gt = 0.0f;                                  // gt = sum_c g_c tilde_c  (g_c = d kl / d tilde_c)
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C && c != y) {
            const float tl = (a[c] / d) / den;
            const float lt = logf(fmaxf(tl, eps));
            kl += tl * lt;
            gt += (lt + (tl >= eps ? 1.0f : 0.0f)) * tl;
          }
        const float nrm = normalize ? 1.0f / lc1 : 1.0f;
        kl *= nrm;
        const float sg = sigmoidf((tau - py) / sigma);
        const float wu = powf(1.0f - py, gamma) * sg;
        const float we = s_t >= 0.0f ? s_t / (d + s_t) : 1.0f;
        v = wu * we * kl;
        if constexpr (BWD) {
          const float w = wu * we * nrm;
          // d/d p_y: through denom = 1 - py (both clamps must be inactive), plus the gate when it is not detached
          float Gy = (py_raw >= eps && den_raw >= eps) ? w * gt / den : 0.0f;
          if (!detach && py_raw >= eps) {
            const float dwu = -gamma * powf(1.0f - py, gamma - 1.0f) * sg - powf(1.0f - py, gamma) * sg * (1.0f - sg) / sigma;
            Gy += dwu * we * kl;
          }
          float dot = Gy * py_raw;                                  // sum_c G_c p_c
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (c < C && c != y) {
              const float pc = a[c] / d, tl = pc / den;
              dot += w * (logf(fmaxf(tl, eps)) + (tl >= eps ? 1.0f : 0.0f)) / den * pc;
            }
#pragma unroll
          for (int c = 0; c < CMAX; ++c)
            if (c < C) {
              const float pc = a[c] / d, tl = pc / den;
              const float Gc = c == y ? Gy : w * (logf(fmaxf(tl, eps)) + (tl >= eps ? 1.0f : 0.0f)) / den;
              dst[(size_t)c * HW] = gs * (Gc - dot) / d;
            }
        }
      } else if constexpr (BWD) {
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) dst[(size_t)c * HW] = 0.0f;
      }
    } else if (kind == 6) {
      const float s_low = prm.p[0], margin = prm.p[1], kk = prm.p[2];
      const float d = fmaxf(a0, eps);
      float pmax = -1.0f;
      int pred = 0;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
          const float pc = a[c] / d;
          if (pc > pmax) { pmax = pc; pred = c; }                  // first maximum, like argmax
        }
      const float m = fmaxf(pmax, eps) - fmaxf(ay / d, eps);
      float gate = pred != y ? 1.0f : 0.0f;
      if (margin > 0.0f) gate *= kk > 0.0f ? sigmoidf((m - margin) / kk) : (m > margin ? 1.0f : 0.0f);
      const float h = fmaxf(logf(d) - logf((float)C + s_low + eps), 0.0f);
      v = h * h * gate;
      lsum2 += (double)gate;
      if constexpr (BWD) {
        const float g = a0 >= eps ? gs * gate * 2.0f * h / d : 0.0f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) dst[(size_t)c * HW] = g;
      }
    } else {
      float S = 0.0f, slg = 0.0f, t2 = 0.0f, sm1 = 0.0f;
      float at[CMAX];
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        at[c] = c < C ? fmaxf(c == y ? 1.0f : a[c], eps) : 0.0f;
        S += at[c];
      }
      const float dS = digamma_pos(S);
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) {
          slg += lgammaf(at[c]);
          t2 += (at[c] - 1.0f) * (digamma_pos(at[c]) - dS);
          sm1 += at[c] - 1.0f;
        }
      v = lgammaf(S) - slg + t2;
      // kind 7: the confidence-weighted form (regularizers.py:375-385): w = clamp(1 - alpha_y / (alpha0 + eps), 0, 1)^gamma, detached;
      // the mean then runs over sum(w) (second accumulator), the gradient is w * d KL / d alpha
      float wconf = 1.0f;
      if (kind == 7) {
        const float om = fminf(fmaxf(1.0f - ay / (a0 + eps), 0.0f), 1.0f);
        wconf = powf(om, prm.p[0]);
        v *= wconf;
        lsum2 += (double)wconf;
      }
      if constexpr (BWD) {
        const float tS = trigamma_pos(S);
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
          if (c < C) {
            // the true class is replaced by the constant 1; a clamped entry (alpha < eps) has zero gradient as well
            const bool live = c != y && a[c] >= eps;
            dst[(size_t)c * HW] = live ? gs * wconf * ((at[c] - 1.0f) * trigamma_pos(at[c]) - tS * sm1) : 0.0f;
          }
      }
    }
    lsum += (double)v;
    ++lcnt;
  }
  if constexpr (!BWD) {
    lsum = wave_sum(lsum);
    lsum2 = wave_sum(lsum2);
    lcnt = (unsigned)wave_sum((float)lcnt);
    if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = lsum; s_sum2[threadIdx.x >> 6] = lsum2; s_cnt[threadIdx.x >> 6] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0, t2 = 0.0;
      unsigned long long n = 0;
      for (int w = 0; w < 4; ++w) { t += s_sum[w]; t2 += s_sum2[w]; n += s_cnt[w]; }
      if (n) { atomicAdd(sum, t); atomicAdd(count, n); }
      if (n && sum2) atomicAdd(sum2, t2);
    }
  }
}

template <bool BWD>
int launch(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, const LossParams& prm, float eps, int has_ignore, int64_t ignore,
           double* sum, double* sum2, int64_t* count, const float* gscale, float* grad, hipStream_t st) {
  const size_t npix = (size_t)B * HW;
  size_t nb = (npix + 255) / 256;
  if (!BWD && nb > 2048) nb = 2048;
  if (BWD && nb > 65535u * 16) nb = 65535u * 16;
  auto cnt = reinterpret_cast<unsigned long long*>(count);
  if (C <= 20)
    hipLaunchKernelGGL((dirichlet_loss_kernel<20, BWD>), dim3((unsigned)nb), dim3(256), 0, st, alpha, labels, B, C, HW, kind, prm, eps, has_ignore,
                       ignore, sum, sum2, cnt, gscale, grad);
  else
    hipLaunchKernelGGL((dirichlet_loss_kernel<32, BWD>), dim3((unsigned)nb), dim3(256), 0, st, alpha, labels, B, C, HW, kind, prm, eps, has_ignore,
                       ignore, sum, sum2, cnt, gscale, grad);
  SLU_CHECK_LAUNCH();
}

bool fill_params(int kind, const float* params, int nparams, LossParams& prm) {
  const int need = kind == 5 ? 6 : (kind == 6 ? 3 : ((kind == 2 || kind == 7) ? 1 : 0));
  if (nparams < need || (need && !params) || nparams > 6) return false;
  for (int i = 0; i < 6; ++i) prm.p[i] = i < nparams ? params[i] : 0.0f;
  if (kind == 5 && !(prm.p[2] > 0.0f)) return false;      // sigma
  return true;
}

}  // namespace

extern "C" int slu_dirichlet_loss_fwd(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, float param, float eps,
                                      int has_ignore, int64_t ignore_index, double* sum, int64_t* count, slu_stream_t stream) {
  if (!alpha || !labels || !sum || !count || B <= 0 || C <= 0 || HW <= 0 || kind < 0 || kind > 4) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  hipStream_t st = slu_stream(stream);
  if (hipMemsetAsync(sum, 0, sizeof(double), st) != hipSuccess || hipMemsetAsync(count, 0, sizeof(int64_t), st) != hipSuccess) return SLU_ELAUNCH;
  const LossParams prm{{param, 0, 0, 0, 0, 0}};
  return launch<false>(alpha, labels, B, C, HW, kind, prm, eps, has_ignore, ignore_index, sum, nullptr, count, nullptr, nullptr, st);
}

extern "C" int slu_dirichlet_loss_bwd(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, float param, float eps,
                                      int has_ignore, int64_t ignore_index, const float* gscale, float* grad_alpha, slu_stream_t stream) {
  if (!alpha || !labels || !gscale || !grad_alpha || B <= 0 || C <= 0 || HW <= 0 || kind < 0 || kind > 4) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const LossParams prm{{param, 0, 0, 0, 0, 0}};
  return launch<true>(alpha, labels, B, C, HW, kind, prm, eps, has_ignore, ignore_index, nullptr, nullptr, nullptr, gscale, grad_alpha, slu_stream(stream));
}

extern "C" int slu_dirichlet_loss_fwd_ex(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, const float* params, int nparams,
                                         float eps, int has_ignore, int64_t ignore_index, double* sums2, int64_t* count, slu_stream_t stream) {
  LossParams prm;
  if (!alpha || !labels || !sums2 || !count || B <= 0 || C <= 0 || HW <= 0 || kind < 0 || kind > 7 || !fill_params(kind, params, nparams, prm)) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  hipStream_t st = slu_stream(stream);
  if (hipMemsetAsync(sums2, 0, 2 * sizeof(double), st) != hipSuccess || hipMemsetAsync(count, 0, sizeof(int64_t), st) != hipSuccess) return SLU_ELAUNCH;
  return launch<false>(alpha, labels, B, C, HW, kind, prm, eps, has_ignore, ignore_index, sums2, sums2 + 1, count, nullptr, nullptr, st);
}

extern "C" int slu_dirichlet_loss_bwd_ex(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, const float* params, int nparams,
                                         float eps, int has_ignore, int64_t ignore_index, const float* gscale, float* grad_alpha, slu_stream_t stream) {
  LossParams prm;
  if (!alpha || !labels || !gscale || !grad_alpha || B <= 0 || C <= 0 || HW <= 0 || kind < 0 || kind > 7 || !fill_params(kind, params, nparams, prm)) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  return launch<true>(alpha, labels, B, C, HW, kind, prm, eps, has_ignore, ignore_index, nullptr, nullptr, nullptr, gscale, grad_alpha, slu_stream(stream));
}
