// Data-movement / per-pixel kernels of the ResNet-FPN model (reference src/models/semanticFCN.py), all HBM-bound:
//   max-pool 3x3/s2/p1 (:149 stem), nearest down-sampling of the meta channels (:283-285), space-to-depth (turns
//   the stride-2 3x3 convs of the ResNet stages into stride-1 2x2 convs), depth-to-space with optional ELU+1
//   (ConvTranspose2d as conv + rearrangement, :230-232,:244-245,:352), attention weights softmax over azimuth
//   multiplied into the value map (:32-38).
#include "slu_common.h"

namespace {

inline unsigned cap(size_t n, unsigned c) { return (unsigned)(n > c ? c : (n ? n : 1)); }

// nn.MaxPool2d(3, stride 2, padding 1): padding never wins (-inf)
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W, int OH, int OW) {
  const size_t total = (size_t)NC * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const float* p = x + (r / OH) * (size_t)H * W;
    float m = -INFINITY;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      const int iy = 2 * oy + i;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const int ix = 2 * ox + j;
        if (ix >= 0 && ix < W) m = fmaxf(m, p[(size_t)iy * W + ix]);
      }
    }
    y[e] = m;
  }
}

// F.interpolate(mode='nearest', scale_factor=1/f): y[oy][ox] = x[oy*f][ox*f]
__global__ __launch_bounds__(256) void nearest_down_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W, int f) {
  const int OH = H / f, OW = W / f;
  const size_t total = (size_t)NC * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    y[e] = x[((r / OH) * H + (size_t)oy * f) * W + (size_t)ox * f];
  }
}

// y[n, (p*2+q)*C + c, oy, ox] = x[n, c, 2*oy+p, 2*ox+q]   (phase-major so each phase is a contiguous channel group)
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W) {
  const int OH = H / 2, OW = W / 2;
  const size_t total = (size_t)N * 4 * C * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    size_t r = e / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int cc = (int)(r % (4 * C));
    const size_t n = r / (4 * C);
    const int ph = cc / C, c = cc - ph * C;
    y[e] = x[((n * C + c) * H + 2 * oy + (ph >> 1)) * W + 2 * ox + (ph & 1)];
  }
}

// space-to-depth of the channel concatenation cat(a[:, :ca], b)  (a has Ca >= ca channels, b has Cb; Cc = ca + Cb):
// y[n, (2p+q)*Cc + c, oy, ox] = (c < ca ? a[n,c,..] : b[n,c-ca,..])[2*oy+p, 2*ox+q]     (semanticFCN.py:309-313 feeding a stride-2 stage)
__global__ __launch_bounds__(256) void space_to_depth2_cat_kernel(const float* __restrict__ a, int Ca, int ca, const float* __restrict__ b, int Cb,
                                                                  float* __restrict__ y, int N, int H, int W) {
  const int OH = H / 2, OW = W / 2, Cc = ca + Cb;
  const size_t total = (size_t)N * 4 * Cc * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    size_t r = e / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int cc = (int)(r % (4 * Cc));
    const size_t n = r / (4 * Cc);
    const int ph = cc / Cc, c = cc - ph * Cc;
    const size_t pos = (size_t)(2 * oy + (ph >> 1)) * W + 2 * ox + (ph & 1);
    y[e] = c < ca ? a[(n * Ca + c) * (size_t)H * W + pos] : b[(n * Cb + (c - ca)) * (size_t)H * W + pos];
  }
}

// nn.PixelShuffle(r) (+ optional ELU(alpha=1) + 1): y[n,c,h*r+i,w*r+j] = f(x[n, c*r*r + i*r + j, h, w])
__global__ __launch_bounds__(256) void depth_to_space_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int Cout, int H, int W, int r,
                                                             int elu_plus_one, int c_off, int Ctot) {
  const int OH = H * r, OW = W * r;
  const size_t total = (size_t)N * Cout * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    size_t q = e / OW;
    const int oy = (int)(q % OH);
    q /= OH;
    const int c = (int)(q % Cout);
    const size_t n = q / Cout;
    const int cs = c * r * r + (oy % r) * r + (ox % r);
    float v = x[((n * (size_t)Cout * r * r + cs) * H + oy / r) * W + ox / r];
    if (elu_plus_one) v = (v > 0.0f ? v : expm1f(v)) + 1.0f;
    y[((n * Ctot + c_off + c) * OH + oy) * (size_t)OW + ox] = v;
  }
}

// out[n,c,y,:] = value[n,c,y,:] * softmax_x(score[n,0,y,:])      one workgroup per (n, y) row
__global__ __launch_bounds__(256) void row_softmax_mul_kernel(const float* __restrict__ score, const float* __restrict__ value,
                                                              float* __restrict__ out, int C, int H, int W) {
  __shared__ float s_red[4];
  __shared__ float s_w[4096];
  const int n = blockIdx.x / H, y = blockIdx.x % H;
  const float* srow = score + ((size_t)n * H + y) * W;
  float m = -INFINITY;
  for (int x = threadIdx.x; x < W; x += blockDim.x) m = fmaxf(m, srow[x]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const float e = expf(srow[x] - m);
    s_w[x] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  for (int c = 0; c < C; ++c) {
    const size_t base = (((size_t)n * C + c) * H + y) * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) out[base + x] = value[base + x] * (s_w[x] * inv);
  }
}

}  // namespace

extern "C" int slu_maxpool3s2_fwd(const float* x, float* y, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const size_t total = (size_t)N * C * OH * OW;
  hipLaunchKernelGGL(maxpool3s2_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), x, y, N * C, H, W, OH, OW);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_nearest_down(const float* x, float* y, int N, int C, int H, int W, int factor, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || factor <= 0) return SLU_EINVAL;
  if (H % factor || W % factor) return SLU_EUNSUPPORTED;
  const size_t total = (size_t)N * C * (H / factor) * (W / factor);
  hipLaunchKernelGGL(nearest_down_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), x, y, N * C, H, W, factor);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_space_to_depth2(const float* x, float* y, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  if ((H & 1) || (W & 1)) return SLU_EUNSUPPORTED;
  const size_t total = (size_t)N * C * H * W;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), x, y, N, C, H, W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_space_to_depth2_cat(const float* a, int Ca, int ca, const float* b, int Cb, float* y, int N, int H, int W, slu_stream_t stream) {
  if (!a || !b || !y || N <= 0 || Ca <= 0 || Cb <= 0 || ca <= 0 || ca > Ca || H <= 0 || W <= 0) return SLU_EINVAL;
  if ((H & 1) || (W & 1)) return SLU_EUNSUPPORTED;
  const size_t total = (size_t)N * (ca + Cb) * H * W;
  hipLaunchKernelGGL(space_to_depth2_cat_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), a, Ca, ca, b, Cb, y, N, H, W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_depth_to_space(const float* x, float* y, int N, int Cout, int H, int W, int r, int elu_plus_one, int c_off, int Ctot,
                                  slu_stream_t stream) {
  if (!x || !y || N <= 0 || Cout <= 0 || H <= 0 || W <= 0 || r <= 0 || c_off < 0 || c_off + Cout > Ctot) return SLU_EINVAL;
  const size_t total = (size_t)N * Cout * H * W * r * r;
  hipLaunchKernelGGL(depth_to_space_kernel, dim3(cap((total + 255) / 256, 16384)), dim3(256), 0, slu_stream(stream), x, y, N, Cout, H, W, r, elu_plus_one, c_off, Ctot);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_row_softmax_mul(const float* score, const float* value, float* out, int N, int C, int H, int W, slu_stream_t stream) {
  if (!score || !value || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  if (W > 4096 || (long long)N * H > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL(row_softmax_mul_kernel, dim3((unsigned)(N * H)), dim3(256), 0, slu_stream(stream), score, value, out, C, H, W);
  SLU_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// Pieces of the `semanticFCN_opt` variant (baselines/Reichert/semanticFCN_opt.py):
//   bilinear_up_kernel        F.interpolate(scale_factor = s, mode = 'bilinear', align_corners = False)       UpsampleBlock :24-27
//   groupnorm_stats_kernel    per (sample, group) mean and 1 / sqrt(var + eps) over (C / G) * H * W           nn.GroupNorm :20,66-70
//   groupnorm_apply_kernel    y = (x - mean) * rstd * gamma[c] + beta[c]  [-> ReLU]
//   spatial_softmax_stats / spatial_gate    w = softmax(score over H * W);  out = x * w + x                    SpatialAttention :80-85
// ---------------------------------------------------------------------------------------------------------------------
namespace {

// source index of ATen's area_pixel_compute_source_index (align_corners = False, not cubic): max(0, (dst + 0.5) / s - 0.5)
__device__ __forceinline__ void bilinear_taps(int dst, int s, int n_in, int& i0, int& i1, float& l0, float& l1) {
  float src = ((float)dst + 0.5f) * (1.0f / (float)s) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.0f - l1;
}

__global__ __launch_bounds__(256) void bilinear_up_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, int H, int W, int s) {
#pragma clang fp contract(off)
  const int OW = W * s, OH = H * s;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t nc = r / OH;
    int y0, y1, x0, x1;
    float hl0, hl1, wl0, wl1;
    bilinear_taps(oy, s, H, y0, y1, hl0, hl1);
    bilinear_taps(ox, s, W, x0, x1, wl0, wl1);
    const float* p = x + nc * (size_t)H * W;
    y[e] = hl0 * (wl0 * p[(size_t)y0 * W + x0] + wl1 * p[(size_t)y0 * W + x1]) + hl1 * (wl0 * p[(size_t)y1 * W + x0] + wl1 * p[(size_t)y1 * W + x1]);
  }
}

// one workgroup per (sample, group): the group's (C / G) * HW values are contiguous in NCHW
__global__ __launch_bounds__(1024) void groupnorm_stats_kernel(const float* __restrict__ x, size_t per_group, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd) {
  __shared__ double s_a[16], s_b[16];
  const float* p = x + (size_t)blockIdx.x * per_group;
  double a = 0.0, b = 0.0;
  for (size_t i = threadIdx.x; i < per_group; i += blockDim.x) {
    const double v = (double)p[i];
    a += v;
    b += v * v;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ta += s_a[w]; tb += s_b[w]; }
    const double m = ta / (double)per_group;
    double var = tb / (double)per_group - m * m;            // biased variance, as nn.GroupNorm
    var = var < 0.0 ? 0.0 : var;
    mean[blockIdx.x] = (float)m;
    rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                                              size_t total, int C, int cpg, size_t HW, int relu) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t nc = e / HW;
    const int c = (int)(nc % C);
    const size_t ng = (nc / C) * (size_t)(C / cpg) + c / cpg;
    float v = (x[e] - mean[ng]) * rstd[ng] * (gamma ? gamma[c] : 1.0f) + (beta ? beta[c] : 0.0f);
    y[e] = relu ? fmaxf(v, 0.0f) : v;
  }
}

// per sample: max and 1 / sum(exp(s - max)) of the score map [HW]
__global__ __launch_bounds__(1024) void spatial_softmax_stats_kernel(const float* __restrict__ score, size_t HW, float* __restrict__ stats) {
  __shared__ float s_r[16];
  const float* p = score + (size_t)blockIdx.x * HW;
  float m = -INFINITY;
  for (size_t i = threadIdx.x; i < HW; i += blockDim.x) m = fmaxf(m, p[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = m;
  __syncthreads();
  m = s_r[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, s_r[w]);
  __syncthreads();
  float sum = 0.0f;
  for (size_t i = threadIdx.x; i < HW; i += blockDim.x) sum += expf(p[i] - m);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_r[w];
    stats[2 * blockIdx.x] = m;
    stats[2 * blockIdx.x + 1] = 1.0f / t;
  }
}

__global__ __launch_bounds__(256) void spatial_gate_kernel(const float* __restrict__ x, const float* __restrict__ score, const float* __restrict__ stats,
                                                           float* __restrict__ out, size_t total, int C, size_t HW) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = e % HW, n = e / HW / C;
    const float w = expf(score[n * HW + hw] - stats[2 * n]) * stats[2 * n + 1];
    const float v = x[e];
    out[e] = v * w + v;
  }
}

inline unsigned grid1d(size_t total) {
  const size_t nb = (total + 255) / 256;
  return (unsigned)(nb > 65535 ? 65535 : (nb ? nb : 1));
}

}  // namespace

extern "C" int slu_bilinear_upsample(const float* x, float* y, int N, int C, int H, int W, int scale, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || scale < 1) return SLU_EINVAL;
  const size_t total = (size_t)N * C * H * W * scale * scale;
  hipLaunchKernelGGL(bilinear_up_kernel, dim3(grid1d(total)), dim3(256), 0, slu_stream(stream), x, y, total, H, W, scale);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_groupnorm_fwd(const float* x, const float* gamma, const float* beta, int N, int C, int HW, int groups, float eps, int relu,
                                 float* mean, float* rstd, float* y, slu_stream_t stream) {
  if (!x || !y || !mean || !rstd || N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups || !(eps >= 0.0f)) return SLU_EINVAL;
  if ((long long)N * groups > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  const int cpg = C / groups;
  hipStream_t st = slu_stream(stream);
  hipLaunchKernelGGL(groupnorm_stats_kernel, dim3((unsigned)(N * groups)), dim3(1024), 0, st, x, (size_t)cpg * HW, eps, mean, rstd);
  const size_t total = (size_t)N * C * HW;
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(grid1d(total)), dim3(256), 0, st, x, mean, rstd, gamma, beta, y, total, C, cpg, (size_t)HW, relu);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_spatial_softmax_gate(const float* x, const float* score, float* stats, float* out, int N, int C, int HW, slu_stream_t stream) {
  if (!x || !score || !stats || !out || N <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  hipStream_t st = slu_stream(stream);
  hipLaunchKernelGGL(spatial_softmax_stats_kernel, dim3((unsigned)N), dim3(1024), 0, st, score, (size_t)HW, stats);
  const size_t total = (size_t)N * C * HW;
  hipLaunchKernelGGL(spatial_gate_kernel, dim3(grid1d(total)), dim3(256), 0, st, x, score, stats, out, total, C, (size_t)HW);
  SLU_CHECK_LAUNCH();
}
