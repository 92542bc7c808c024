// Training-side per-pixel kernels around the conv stack (HBM-bound):
//   train-mode BatchNorm statistics / apply, BatchNorm + LeakyReLU backward, per-channel reductions,
//   NCHW -> NHWC transposes (the weight-gradient kernel wants channels on the lanes), gradient split of a
//   concatenated / pixel-shuffled / dropout-scaled conv input, average-pool backward.
// Reference semantics: nn.BatchNorm2d (biased batch variance for normalisation, SalsaNext.py:32,...),
// nn.LeakyReLU(0.01), nn.AvgPool2d(3,2,1), nn.PixelShuffle(2), nn.Dropout2d as a per-(n,c) multiplier.
#include "slu_common.h"

namespace {

// ---- per-channel reductions over (N, HW): 2-D grid (chunk of pixels, channel) ----------------------
// mode 0: sum y, sum y^2            (batch statistics)
// mode 1: sum g, sum g * (y - mean[c]) * invstd[c]   (BatchNorm backward)
// VEC = 4: HW % 4 == 0 and 16-byte aligned tensors -- one 16-byte load per stream and lane (the scalar form moved 4 bytes per lane and
// spent a 64-bit division per element: 42 % of the HBM rate on the full-resolution layers)
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          int N, int C, int HW, double* __restrict__ o1, double* __restrict__ o2) {
  const int c = blockIdx.y;
  const int hwv = HW / VEC;
  const size_t per_c = (size_t)N * hwv;
  double a1 = 0.0, a2 = 0.0;
  const float mu = MODE == 1 ? mean[c] : 0.0f, is = MODE == 1 ? invstd[c] : 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_c; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hwv, hw = (i - n * hwv) * VEC;
    const size_t idx = (n * C + c) * HW + hw;
    float pv[VEC], qv[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p + idx);
      pv[0] = t.x; pv[1] = t.y; pv[2] = t.z; pv[3] = t.w;
      if (MODE == 1) {
        const float4 u = *reinterpret_cast<const float4*>(q + idx);
        qv[0] = u.x; qv[1] = u.y; qv[2] = u.z; qv[3] = u.w;
      }
    } else {
      pv[0] = p[idx];
      if (MODE == 1) qv[0] = q[idx];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (MODE == 0) {
        a1 += (double)pv[k];
        a2 += (double)pv[k] * (double)pv[k];
      } else {
        a1 += (double)pv[k];
        a2 += (double)(pv[k] * ((qv[k] - mu) * is));
      }
    }
  }
  __shared__ double s1[4], s2[4];
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = a1; s2[threadIdx.x >> 6] = a2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&o1[c], s1[0] + s1[1] + s1[2] + s1[3]);
    atomicAdd(&o2[c], s2[0] + s2[1] + s2[2] + s2[3]);
  }
}

// Per-channel coefficient math of train/eval BatchNorm in one launch (fp64 inside), replacing dozens of [C]-sized
// tensor ops per layer.
//   train: mean = sum/M, var = sumsq/M - mean^2 (biased), running <- (1-mom) running + mom {mean, var*M/(M-1)}
//   eval : mean/var = running stats
//   a = gamma * invstd, b = beta - mean * a
__global__ void bn_coeffs_fwd_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq, double M, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, float eps, float momentum, int train, float* __restrict__ running_mean,
                                     float* __restrict__ running_var, int C, float* __restrict__ mean, float* __restrict__ invstd,
                                     float* __restrict__ a, float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mu, var;
  if (train) {
    mu = sum[c] / M;
    var = sumsq[c] / M - mu * mu;
    var = var > 0.0 ? var : 0.0;
    if (running_mean) {
      const double unbiased = var * (M / (M > 1.0 ? M - 1.0 : 1.0));
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  } else {
    mu = (double)running_mean[c];
    var = (double)running_var[c];
  }
  const double is = 1.0 / sqrt(var + (double)eps);
  const double g = (double)gamma[c];
  mean[c] = (float)mu;
  invstd[c] = (float)is;
  a[c] = (float)(g * is);
  b[c] = (float)((double)beta[c] - mu * g * is);
}

// backward coefficients: da = (k1*dz + k2 + k3*y) * act'(y);  dgamma = s2, dbeta = s1
//   k1 = gamma*invstd;  train: k3 = -gamma*invstd^2*s2/M, k2 = -gamma*invstd*s1/M - k3*mean;  eval: k2 = k3 = 0
__global__ void bn_coeffs_bwd_kernel(const double* __restrict__ s1, const double* __restrict__ s2, double M, const float* __restrict__ gamma,
                                     const float* __restrict__ mean, const float* __restrict__ invstd, int train, int C,
                                     float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double g = (double)gamma[c], is = (double)invstd[c];
  const double q3 = train ? -g * is * is * s2[c] / M : 0.0;
  k1[c] = (float)(g * is);
  k3[c] = (float)q3;
  k2[c] = train ? (float)(-g * is * s1[c] / M - q3 * (double)mean[c]) : 0.0f;
  dgamma[c] = (float)s2[c];
  dbeta[c] = (float)s1[c];
}

// z = a[c] * y + b[c] (+ resid)
template <int VEC>
__global__ __launch_bounds__(256) void affine_kernel(const float* __restrict__ y, const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ resid, float* __restrict__ z, int C, int HW, size_t total) {
  const int hwv = HW / VEC;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total / VEC; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((e / hwv) % C);
    const float ac = a ? a[c] : 1.0f, bc = b ? b[c] : 0.0f;
    if constexpr (VEC == 4) {
      const float4 v = reinterpret_cast<const float4*>(y)[e];
      float4 o = make_float4(v.x * ac + bc, v.y * ac + bc, v.z * ac + bc, v.w * ac + bc);
      if (resid) {
        const float4 r = reinterpret_cast<const float4*>(resid)[e];
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      reinterpret_cast<float4*>(z)[e] = o;
    } else {
      float v = y[e] * ac + bc;
      if (resid) v += resid[e];
      z[e] = v;
    }
  }
}

// da = (k1[c] * dz + k2[c] + k3[c] * y) * leaky'(y)      and      dbias[c] += sum da
template <int VEC>
__global__ __launch_bounds__(256) void act_affine_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                             const float* __restrict__ k1, const float* __restrict__ k2,
                                                             const float* __restrict__ k3, float slope, int has_act, int N, int C,
                                                             int HW, float* __restrict__ da, double* __restrict__ dbias) {
  const int c = blockIdx.y;
  const int hwv = HW / VEC;
  const size_t per_c = (size_t)N * hwv;
  const float c1 = k1 ? k1[c] : 1.0f, c2 = k2 ? k2[c] : 0.0f, c3 = k3 ? k3[c] : 0.0f;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_c; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hwv, hw = (i - n * hwv) * VEC;
    const size_t idx = (n * C + c) * HW + hw;
    float yv[VEC], gv[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(dz + idx);
      gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
      if (y) {
        const float4 u = *reinterpret_cast<const float4*>(y + idx);
        yv[0] = u.x; yv[1] = u.y; yv[2] = u.z; yv[3] = u.w;
      } else {
        yv[0] = yv[1] = yv[2] = yv[3] = 0.0f;
      }
    } else {
      gv[0] = dz[idx];
      yv[0] = y ? y[idx] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float g = c1 * gv[k] + c2 + c3 * yv[k];
      if (has_act && !(yv[k] > 0.0f)) g *= slope;
      gv[k] = g;
      acc += (double)g;
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(da + idx) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    else da[idx] = gv[0];
  }
  if (dbias) {
    __shared__ double s[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&dbias[c], s[0] + s[1] + s[2] + s[3]);
  }
}


// ---- the two per-layer BatchNorm passes with their coefficient kernels folded in (one launch each instead of two + a dtype conversion) ----
// z = BatchNorm(y) [+ resid]: every workgroup of channel c derives mean / invstd / a / b from the fp64 sums itself (a dozen fp64 operations);
// workgroup 0 of the channel also publishes mean / invstd for the backward and updates the running statistics.
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ y, const double* __restrict__ sum, const double* __restrict__ sumsq,
                                                           double M, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float momentum, int train, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, const float* __restrict__ resid, float* __restrict__ z,
                                                           int N, int C, int HW, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
  const int c = blockIdx.y;
  __shared__ float s_ab[2];
  if (threadIdx.x == 0) {
    double mu, var;
    if (train) {
      mu = sum[c] / M;
      var = sumsq[c] / M - mu * mu;
      var = var > 0.0 ? var : 0.0;
    } else {
      mu = (double)running_mean[c];
      var = (double)running_var[c];
    }
    const double is = 1.0 / sqrt(var + (double)eps), g = (double)gamma[c];
    s_ab[0] = (float)(g * is);
    s_ab[1] = (float)((double)beta[c] - mu * g * is);
    if (blockIdx.x == 0) {
      mean_out[c] = (float)mu;
      invstd_out[c] = (float)is;
      if (train && running_mean) {
        const double unbiased = var * (M / (M > 1.0 ? M - 1.0 : 1.0));
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
      }
    }
  }
  __syncthreads();
  const float ac = s_ab[0], bc = s_ab[1];
  const int hwv = HW / VEC;
  const size_t per_c = (size_t)N * hwv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_c; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hwv, hw = (i - n * hwv) * VEC;
    const size_t idx = (n * C + c) * HW + hw;
    if constexpr (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(y + idx);
      float4 o = make_float4(v.x * ac + bc, v.y * ac + bc, v.z * ac + bc, v.w * ac + bc);
      if (resid) {
        const float4 r = *reinterpret_cast<const float4*>(resid + idx);
        o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
      }
      *reinterpret_cast<float4*>(z + idx) = o;
    } else {
      float v = y[idx] * ac + bc;
      if (resid) v += resid[idx];
      z[idx] = v;
    }
  }
}

// da = (k1 dz + k2 + k3 y) * leaky'(y) with k1, k2, k3 derived per workgroup from the reduction sums s1 / s2 (has_bn) or 1, 0, 0; workgroup 0 of
// a channel writes dgamma / dbeta; the bias gradient is summed in fp64 (acc64, zeroed by the caller) and the LAST workgroup of the channel to
// arrive (ticket counter, zeroed by the caller) rounds it to fp32 -- no conversion launch.
template <int VEC>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ y, const double* __restrict__ s1,
                                                         const double* __restrict__ s2, double M, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd, int has_bn, int train,
                                                         float slope, int has_act, int N, int C, int HW, float* __restrict__ da,
                                                         double* __restrict__ acc64, unsigned* __restrict__ ticket, float* __restrict__ dbias,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.y;
  __shared__ float s_k[3];
  if (threadIdx.x == 0) {
    float k1 = 1.0f, k2 = 0.0f, k3 = 0.0f;
    if (has_bn) {
      const double g = (double)gamma[c], is = (double)invstd[c];
      const double q3 = train ? -g * is * is * s2[c] / M : 0.0;
      k1 = (float)(g * is);
      k3 = (float)q3;
      k2 = train ? (float)(-g * is * s1[c] / M - q3 * (double)mean[c]) : 0.0f;
      if (blockIdx.x == 0) {
        dgamma[c] = (float)s2[c];
        dbeta[c] = (float)s1[c];
      }
    }
    s_k[0] = k1; s_k[1] = k2; s_k[2] = k3;
  }
  __syncthreads();
  const float c1 = s_k[0], c2 = s_k[1], c3 = s_k[2];
  const int hwv = HW / VEC;
  const size_t per_c = (size_t)N * hwv;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_c; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / hwv, hw = (i - n * hwv) * VEC;
    const size_t idx = (n * C + c) * HW + hw;
    float yv[VEC], gv[VEC];
    if constexpr (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(dz + idx);
      gv[0] = t.x; gv[1] = t.y; gv[2] = t.z; gv[3] = t.w;
      if (y) {
        const float4 u = *reinterpret_cast<const float4*>(y + idx);
        yv[0] = u.x; yv[1] = u.y; yv[2] = u.z; yv[3] = u.w;
      } else {
        yv[0] = yv[1] = yv[2] = yv[3] = 0.0f;
      }
    } else {
      gv[0] = dz[idx];
      yv[0] = y ? y[idx] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float g = c1 * gv[k] + c2 + c3 * yv[k];
      if (has_act && !(yv[k] > 0.0f)) g *= slope;
      gv[k] = g;
      acc += (double)g;
    }
    if constexpr (VEC == 4) *reinterpret_cast<float4*>(da + idx) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    else da[idx] = gv[0];
  }
  if (dbias) {
    __shared__ double s[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      // Ordering without a fence (a __threadfence() here writes the L2 back at the end of EVERY workgroup of a kernel that has just filled it
      // with `da`: +2.8 ms per training step, measured): all three operations are device-scope read-modify-writes executed at the memory side;
      // the RETURNED value of the add is consumed before the ticket is taken, so the add has been performed by then, and the last workgroup
      // reads the sum with another atomic.
      const double before = atomicAdd(&acc64[c], s[0] + s[1] + s[2] + s[3]);
      unsigned one = 1u;
      asm volatile("" : "+v"(one) : "v"(before));
      if (atomicAdd(&ticket[c], one) == gridDim.x - 1) dbias[c] = (float)atomicAdd(&acc64[c], 0.0);
    }
  }
}

// ---- [N,C,HW] -> [N,HW,Cp] (Cp = C rounded up to 32, zero filled) through a 32x64 LDS tile ----------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int Cp, int HW, float* __restrict__ dst) {
  __shared__ float tile[32][65];
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int c = c0 + ty + 4 * r, p = p0 + tx;
    tile[ty + 4 * r][tx] = (c < C && p < HW) ? src[((size_t)n * C + c) * HW + p] : 0.0f;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int p = p0 + py + 8 * r;
    if (p < HW) dst[((size_t)n * HW + p) * Cp + c0 + cx] = tile[cx][py + 8 * r];
  }
}

struct GatherArgs {
  const float* ptr[SLU_MAX_SRC];
  const float* scale[SLU_MAX_SRC];
  int C[SLU_MAX_SRC], ps[SLU_MAX_SRC], cbeg[SLU_MAX_SRC], ccount[SLU_MAX_SRC];
  int nsrc, H, W, Cin, Cp;
};

// the conv's actual input (concat of sources, PixelShuffle, multipliers) as [N,H*W,Cp]
__global__ __launch_bounds__(256) void gather_nhwc_kernel(const GatherArgs a, float* __restrict__ dst) {
  __shared__ float tile[32][65];
  const int HW = a.H * a.W;
  const int n = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int cg = c0 + ty + 4 * r, p = p0 + tx;
    float v = 0.0f;
    if (cg < a.Cin && p < HW) {
      const int gy = p / a.W, gx = p - gy * a.W;
#pragma unroll
      for (int s = 0; s < SLU_MAX_SRC; ++s) {
        const int cl = cg - a.cbeg[s];
        if (s < a.nsrc && cl >= 0 && cl < a.ccount[s]) {
          int cs;
          if (!a.ps[s]) {
            cs = cl;
            v = a.ptr[s][(((size_t)n * a.C[s] + cs) * a.H + gy) * a.W + gx];
          } else {
            cs = cl * 4 + ((gy & 1) << 1) + (gx & 1);
            v = a.ptr[s][(((size_t)n * a.C[s] + cs) * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)];
          }
          if (a.scale[s]) v *= a.scale[s][(size_t)n * a.C[s] + cs];
        }
      }
    }
    tile[ty + 4 * r][tx] = v;
  }
  __syncthreads();
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int p = p0 + py + 8 * r;
    if (p < HW) dst[((size_t)n * HW + p) * a.Cp + c0 + cx] = tile[cx][py + 8 * r];
  }
}

// gradient of one source of a conv input: dsrc[n,cs,..] = dcat[n, cbeg + f(cs), ..] * scale[n,cs]
// plain source, H * W % 4 == 0: 16 bytes per lane
__global__ __launch_bounds__(256) void split_grad_vec4_kernel(const float* __restrict__ dcat, int Ccat, int cbeg, int HW, int Csrc,
                                                              const float* __restrict__ scale, float* __restrict__ dsrc, size_t total4) {
  const int hwv = HW / 4;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const size_t r = e / hwv;
    const int hw4 = (int)(e - r * hwv);
    const int cs = (int)(r % Csrc);
    const size_t n = r / Csrc;
    float4 v = reinterpret_cast<const float4*>(dcat + (n * Ccat + cbeg + cs) * (size_t)HW)[hw4];
    if (scale) {
      const float sc = scale[n * Csrc + cs];
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    }
    reinterpret_cast<float4*>(dsrc)[e] = v;
  }
}

__global__ __launch_bounds__(256) void split_grad_kernel(const float* __restrict__ dcat, int Ccat, int cbeg, int H, int W, int Csrc, int ps,
                                                         const float* __restrict__ scale, float* __restrict__ dsrc, size_t total) {
  const int h = ps ? H >> 1 : H, w = ps ? W >> 1 : W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(e % w);
    size_t r = e / w;
    const int y = (int)(r % h);
    r /= h;
    const int cs = (int)(r % Csrc);
    const size_t n = r / Csrc;
    int c, gy, gx;
    if (ps) { c = cs >> 2; gy = 2 * y + ((cs >> 1) & 1); gx = 2 * x + (cs & 1); }
    else    { c = cs; gy = y; gx = x; }
    float v = dcat[((n * Ccat + cbeg + c) * H + gy) * W + gx];
    if (scale) v *= scale[n * Csrc + cs];
    dsrc[e] = v;
  }
}

// dx of y = avgpool3s2(x * s): dx[n,c,iy,ix] = s/9 * sum of dy over the (<= 2x2) windows covering (iy,ix)
__global__ __launch_bounds__(256) void avgpool3s2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ scale,
                                                             float* __restrict__ dx, int NC, int H, int W, int OH, int OW) {
  const size_t total = (size_t)NC * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(e % W);
    const size_t r = e / W;
    const int iy = (int)(r % H);
    const size_t nc = r / H;
    const float* p = dy + nc * (size_t)OH * OW;
    float acc = 0.0f;
    const int oy0 = iy >> 1, ox0 = ix >> 1;
    const int ny = (iy & 1) ? 2 : 1, nx = (ix & 1) ? 2 : 1;
    for (int i = 0; i < ny; ++i) {
      const int oy = oy0 + i;
      if (oy >= OH) continue;
      for (int j = 0; j < nx; ++j) {
        const int ox = ox0 + j;
        if (ox < OW) acc += p[(size_t)oy * OW + ox];
      }
    }
    dx[e] = acc / 9.0f * (scale ? scale[nc] : 1.0f);
  }
}

inline unsigned cap(size_t n, unsigned c) { return (unsigned)(n > c ? c : (n ? n : 1)); }

}  // namespace

extern "C" int slu_bn_stats(const float* y, int N, int C, int HW, double* sum, double* sumsq, slu_stream_t stream) {
  if (!y || !sum || !sumsq || N <= 0 || C <= 0 || HW <= 0 || C > 65535) return SLU_EINVAL;
  const unsigned gx = cap(((size_t)N * HW + 8191) / 8192, 64);      // one fp64 atomic (pair) per workgroup and channel: keep the chains short
  if (HW % 4 == 0 && !((uintptr_t)y & 15))
    hipLaunchKernelGGL((chan_reduce_kernel<0, 4>), dim3(gx, C), dim3(256), 0, slu_stream(stream), y, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, C, HW, sum, sumsq);
  else
    hipLaunchKernelGGL((chan_reduce_kernel<0, 1>), dim3(gx, C), dim3(256), 0, slu_stream(stream), y, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, C, HW, sum, sumsq);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_bn_bwd_reduce(const float* dz, const float* y, const float* mean, const float* invstd, int N, int C, int HW,
                                 double* s1, double* s2, slu_stream_t stream) {
  if (!dz || !y || !mean || !invstd || !s1 || !s2 || N <= 0 || C <= 0 || HW <= 0 || C > 65535) return SLU_EINVAL;
  const unsigned gx = cap(((size_t)N * HW + 8191) / 8192, 64);      // one fp64 atomic (pair) per workgroup and channel: keep the chains short
  if (HW % 4 == 0 && !(((uintptr_t)dz | (uintptr_t)y) & 15))
    hipLaunchKernelGGL((chan_reduce_kernel<1, 4>), dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, mean, invstd, N, C, HW, s1, s2);
  else
    hipLaunchKernelGGL((chan_reduce_kernel<1, 1>), dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, mean, invstd, N, C, HW, s1, s2);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_bn_coeffs_fwd(const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float eps,
                                 float momentum, int train, float* running_mean, float* running_var, int C, float* mean, float* invstd,
                                 float* a, float* b, slu_stream_t stream) {
  if (!gamma || !beta || !mean || !invstd || !a || !b || C <= 0) return SLU_EINVAL;
  if (train ? (!sum || !sumsq || count <= 0) : (!running_mean || !running_var)) return SLU_EINVAL;
  hipLaunchKernelGGL(bn_coeffs_fwd_kernel, dim3((C + 255) / 256), dim3(256), 0, slu_stream(stream), sum, sumsq, count, gamma, beta, eps,
                     momentum, train, running_mean, running_var, C, mean, invstd, a, b);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_bn_coeffs_bwd(const double* s1, const double* s2, double count, const float* gamma, const float* mean, const float* invstd,
                                 int train, int C, float* k1, float* k2, float* k3, float* dgamma, float* dbeta, slu_stream_t stream) {
  if (!s1 || !s2 || !gamma || !mean || !invstd || !k1 || !k2 || !k3 || !dgamma || !dbeta || C <= 0 || count <= 0) return SLU_EINVAL;
  hipLaunchKernelGGL(bn_coeffs_bwd_kernel, dim3((C + 255) / 256), dim3(256), 0, slu_stream(stream), s1, s2, count, gamma, mean, invstd,
                     train, C, k1, k2, k3, dgamma, dbeta);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_affine_fwd(const float* y, const float* a, const float* b, const float* resid, float* z, int N, int C, int HW,
                              slu_stream_t stream) {
  if (!y || !z || N <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  const size_t total = (size_t)N * C * HW;
  if (HW % 4 == 0 && !(((uintptr_t)y | (uintptr_t)z | (uintptr_t)resid) & 15))
    hipLaunchKernelGGL(affine_kernel<4>, dim3(cap((total / 4 + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), y, a, b, resid, z, C, HW, total);
  else
    hipLaunchKernelGGL(affine_kernel<1>, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), y, a, b, resid, z, C, HW, total);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_act_affine_bwd(const float* dz, const float* y, const float* k1, const float* k2, const float* k3, float slope,
                                  int has_act, int N, int C, int HW, float* da, double* dbias, slu_stream_t stream) {
  if (!dz || !da || N <= 0 || C <= 0 || HW <= 0 || C > 65535) return SLU_EINVAL;
  if ((has_act || k3) && !y) return SLU_EINVAL;
  const unsigned gx = cap(((size_t)N * HW + 8191) / 8192, 64);      // one fp64 atomic (pair) per workgroup and channel: keep the chains short
  if (HW % 4 == 0 && !(((uintptr_t)dz | (uintptr_t)y | (uintptr_t)da) & 15))
    hipLaunchKernelGGL(act_affine_bwd_kernel<4>, dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, k1, k2, k3, slope, has_act, N, C, HW, da, dbias);
  else
    hipLaunchKernelGGL(act_affine_bwd_kernel<1>, dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, k1, k2, k3, slope, has_act, N, C, HW, da, dbias);
  SLU_CHECK_LAUNCH();
}


extern "C" int slu_bn_apply_fwd(const float* y, const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float eps,
                                float momentum, int train, float* running_mean, float* running_var, const float* resid, float* z, int N, int C,
                                int HW, float* mean, float* invstd, slu_stream_t stream) {
  if (!y || !z || !gamma || !beta || !mean || !invstd || N <= 0 || C <= 0 || HW <= 0 || C > 65535) return SLU_EINVAL;
  if (train ? (!sum || !sumsq || count <= 0) : (!running_mean || !running_var)) return SLU_EINVAL;
  const unsigned gx = cap(((size_t)N * HW + 8191) / 8192, 64);
  if (HW % 4 == 0 && !(((uintptr_t)y | (uintptr_t)z | (uintptr_t)resid) & 15))
    hipLaunchKernelGGL(bn_apply_fwd_kernel<4>, dim3(gx, C), dim3(256), 0, slu_stream(stream), y, sum, sumsq, count, gamma, beta, eps, momentum, train,
                       running_mean, running_var, resid, z, N, C, HW, mean, invstd);
  else
    hipLaunchKernelGGL(bn_apply_fwd_kernel<1>, dim3(gx, C), dim3(256), 0, slu_stream(stream), y, sum, sumsq, count, gamma, beta, eps, momentum, train,
                       running_mean, running_var, resid, z, N, C, HW, mean, invstd);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_bn_act_bwd(const float* dz, const float* y, const double* s1, const double* s2, double count, const float* gamma, const float* mean,
                              const float* invstd, int has_bn, int train, float slope, int has_act, int N, int C, int HW, float* da, double* acc64,
                              unsigned* ticket, float* dbias, float* dgamma, float* dbeta, slu_stream_t stream) {
  if (!dz || !da || N <= 0 || C <= 0 || HW <= 0 || C > 65535) return SLU_EINVAL;
  if (has_bn && (!s1 || !s2 || !gamma || !mean || !invstd || !dgamma || !dbeta || count <= 0)) return SLU_EINVAL;
  if ((has_act || (has_bn && train)) && !y) return SLU_EINVAL;
  if (dbias && (!acc64 || !ticket)) return SLU_EINVAL;
  const unsigned gx = cap(((size_t)N * HW + 8191) / 8192, 64);      // one fp64 atomic per workgroup and channel: keep the chains short
  if (HW % 4 == 0 && !(((uintptr_t)dz | (uintptr_t)y | (uintptr_t)da) & 15))
    hipLaunchKernelGGL(bn_act_bwd_kernel<4>, dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, s1, s2, count, gamma, mean, invstd, has_bn, train,
                       slope, has_act, N, C, HW, da, acc64, ticket, dbias, dgamma, dbeta);
  else
    hipLaunchKernelGGL(bn_act_bwd_kernel<1>, dim3(gx, C), dim3(256), 0, slu_stream(stream), dz, y, s1, s2, count, gamma, mean, invstd, has_bn, train,
                       slope, has_act, N, C, HW, da, acc64, ticket, dbias, dgamma, dbeta);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_nchw_to_nhwc(const float* src, int N, int C, int HW, float* dst, slu_stream_t stream) {
  if (!src || !dst || N <= 0 || C <= 0 || HW <= 0 || N > 65535) return SLU_EINVAL;
  const int Cp = (C + 31) / 32 * 32;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 63) / 64, Cp / 32, N), dim3(256), 0, slu_stream(stream), src, C, Cp, HW, dst);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_gather_nhwc(const slu_conv_src* src, int nsrc, int N, int H, int W, float* dst, slu_stream_t stream) {
  if (!src || !dst || nsrc < 1 || nsrc > SLU_MAX_SRC || N <= 0 || H <= 0 || W <= 0 || N > 65535) return SLU_EINVAL;
  GatherArgs a{};
  int c = 0;
  for (int s = 0; s < nsrc; ++s) {
    if (!src[s].ptr || src[s].C <= 0) return SLU_EINVAL;
    if (src[s].pixel_shuffle && ((src[s].C & 3) || (H & 1) || (W & 1))) return SLU_EINVAL;
    a.ptr[s] = src[s].ptr; a.scale[s] = src[s].scale; a.C[s] = src[s].C; a.ps[s] = src[s].pixel_shuffle ? 1 : 0;
    a.cbeg[s] = c; a.ccount[s] = src[s].pixel_shuffle ? src[s].C / 4 : src[s].C;
    c += a.ccount[s];
  }
  a.nsrc = nsrc; a.H = H; a.W = W; a.Cin = c; a.Cp = (c + 31) / 32 * 32;
  hipLaunchKernelGGL(gather_nhwc_kernel, dim3((H * W + 63) / 64, a.Cp / 32, N), dim3(256), 0, slu_stream(stream), a, dst);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_split_grad(const float* dcat, int N, int Ccat, int cbeg, int H, int W, int Csrc, int pixel_shuffle, const float* scale,
                              float* dsrc, slu_stream_t stream) {
  if (!dcat || !dsrc || N <= 0 || Ccat <= 0 || Csrc <= 0 || H <= 0 || W <= 0 || cbeg < 0) return SLU_EINVAL;
  const int contributed = pixel_shuffle ? Csrc / 4 : Csrc;
  if (cbeg + contributed > Ccat || (pixel_shuffle && ((Csrc & 3) || (H & 1) || (W & 1)))) return SLU_EINVAL;
  const size_t total = (size_t)N * Csrc * (pixel_shuffle ? (H >> 1) * (size_t)(W >> 1) : (size_t)H * W);
  if (!pixel_shuffle && ((size_t)H * W) % 4 == 0 && !(((uintptr_t)dcat | (uintptr_t)dsrc) & 15)) {
    hipLaunchKernelGGL(split_grad_vec4_kernel, dim3(cap((total / 4 + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), dcat, Ccat, cbeg, H * W, Csrc,
                       scale, dsrc, total / 4);
    SLU_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(split_grad_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), dcat, Ccat, cbeg, H, W, Csrc,
                     pixel_shuffle ? 1 : 0, scale, dsrc, total);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_avgpool3s2_bwd(const float* dy, const float* scale, float* dx, int N, int C, int H, int W, slu_stream_t stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  const size_t total = (size_t)N * C * H * W;
  hipLaunchKernelGGL(avgpool3s2_bwd_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), dy, scale, dx, N * C, H, W,
                     (H + 1) / 2, (W + 1) / 2);
  SLU_CHECK_LAUNCH();
}
