// Training path of the ResNet-FPN models (reference src/models/semanticFCN.py:266-354, src/baselines/Reichert/semanticFCN_opt.py:366-455;
// the trainer calls loss.backward() on them, src/models/trainer.py:783-786): the per-pixel / data-movement kernels the autograd nodes of
// semanticlidarunc_amd/fpn_autograd.py need besides the conv / BatchNorm machinery of backward.hip and wgrad.hip.  All HBM-bound, fp32.
//   pointwise activations (ReLU / LeakyReLU, tanh, ELU + 1; SiLU) forward and backward (from the OUTPUT, so no input is kept; SiLU: from the input)
//   MaxPool2d(3, 2, 1) backward (gather form: every input pixel re-derives the arg-max of the <= 4 windows that contain it)
//   nearest down-sampling backward, channel-tail replacement (x[:, -m:] = meta) forward / backward
//   softmax-over-azimuth x value (AttentionModule) backward, depth-to-space (ConvTranspose2d as conv + rearrangement) backward
//   bilinear up-sampling backward, GroupNorm forward-with-statistics / backward, spatial-softmax gate backward (semanticFCN_opt)
#include "slu_common.h"

namespace {

inline unsigned grid_for(size_t total, unsigned cap = 65536) {
  const size_t nb = (total + 255) / 256;
  return (unsigned)(nb > cap ? cap : (nb ? nb : 1));
}

enum { OP_LEAKY = 0, OP_TANH = 1, OP_ELU1 = 2, OP_SILU = 3 };

__global__ __launch_bounds__(256) void pointwise_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, int op, float slope) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = x[e];
    float r;
    if (op == OP_LEAKY) r = v > 0.0f ? v : v * slope;
    else if (op == OP_TANH) r = tanhf(v);
    else if (op == OP_SILU) r = v / (1.0f + expf(-v));
    else r = (v > 0.0f ? v : expm1f(v)) + 1.0f;      // nn.ELU(alpha = 1) followed by the reference's "+ 1" (semanticFCN.py:352)
    y[e] = r;
  }
}

// dx = dy * f'(x) expressed through y = f(x): leaky: y > 0 ? 1 : slope (slope >= 0 keeps the sign); tanh: 1 - y^2; ELU + 1: y > 1 ? 1 : y.
// SiLU is not invertible from its output: there `y` is the forward's INPUT x, f'(x) = s (1 + x (1 - s)), s = sigmoid(x)
__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n,
                                                            int op, float slope) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float g = dy[e], v = y[e];
    float d;
    if (op == OP_LEAKY) d = v > 0.0f ? 1.0f : slope;
    else if (op == OP_TANH) d = 1.0f - v * v;
    else if (op == OP_SILU) { const float sg = 1.0f / (1.0f + expf(-v)); d = sg * (1.0f + v * (1.0f - sg)); }
    else d = v > 1.0f ? 1.0f : v;
    dx[e] = g * d;
  }
}

// first maximum of the 3x3 / stride 2 / pad 1 window of output (oy, ox) in row-major order (ATen keeps the first: `val > maxval`)
__device__ __forceinline__ int window_argmax(const float* __restrict__ p, int H, int W, int oy, int ox) {
  float m = -INFINITY;
  int arg = -1;
#pragma unroll
  for (int i = -1; i <= 1; ++i) {
    const int iy = 2 * oy + i;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int j = -1; j <= 1; ++j) {
      const int ix = 2 * ox + j;
      if (ix < 0 || ix >= W) continue;
      const float v = p[(size_t)iy * W + ix];
      if (v > m || v != v) m = v, arg = iy * W + ix;
    }
  }
  return arg;
}

__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int NC,
                                                             int H, int W, int OH, int OW) {
  const size_t total = (size_t)NC * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(e % W);
    const size_t r = e / W;
    const int iy = (int)(r % H);
    const size_t nc = r / H;
    const float* p = x + nc * (size_t)H * W;
    const float* g = dy + nc * (size_t)OH * OW;
    // windows that contain (iy, ix): oy with 2 oy - 1 <= iy <= 2 oy + 1
    const int oy0 = iy / 2, oy1 = (iy + 1) / 2, ox0 = ix / 2, ox1 = (ix + 1) / 2;
    float acc = 0.0f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (oy >= OH) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (ox >= OW) continue;
        if (window_argmax(p, H, W, oy, ox) == iy * W + ix) acc += g[(size_t)oy * OW + ox];
      }
    }
    dx[e] = acc;
  }
}

// backward of y[oy][ox] = x[oy f][ox f]: the gradient lands on the sampled pixels, zero elsewhere
__global__ __launch_bounds__(256) void nearest_down_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int H, int W, int f) {
  const int OH = H / f, OW = W / f;
  const size_t total = (size_t)NC * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(e % W);
    const size_t r = e / W;
    const int iy = (int)(r % H);
    const bool hit = iy % f == 0 && ix % f == 0 && iy / f < OH && ix / f < OW;
    dx[e] = hit ? dy[((r / H) * OH + iy / f) * (size_t)OW + ix / f] : 0.0f;
  }
}

// out = cat(x[:, :C-m], meta)   (the reference's `torch.cat([x1[:, 0:-m], meta_k], 1)`, semanticFCN.py:309-313)
__global__ __launch_bounds__(256) void replace_tail_fwd_kernel(const float* __restrict__ x, const float* __restrict__ meta, float* __restrict__ out, int N,
                                                               int C, int m, size_t HW) {
  const size_t total = (size_t)N * C * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = e % HW;
    const size_t r = e / HW;
    const int c = (int)(r % C);
    const size_t n = r / C;
    out[e] = c < C - m ? x[e] : meta[(n * m + (c - (C - m))) * HW + pix];
  }
}

__global__ __launch_bounds__(256) void replace_tail_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, float* __restrict__ dmeta, int N, int C,
                                                               int m, size_t HW) {
  const size_t total = (size_t)N * C * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = e % HW;
    const size_t r = e / HW;
    const int c = (int)(r % C);
    const size_t n = r / C;
    const float g = dout[e];
    if (c < C - m) {
      if (dx) dx[e] = g;
    } else {
      if (dx) dx[e] = 0.0f;
      if (dmeta) dmeta[(n * m + (c - (C - m))) * HW + pix] = g;
    }
  }
}

// out = value * softmax_x(score) per (n, y) row.  dvalue = dout * p;  dp[x] = sum_c dout * value;  dscore = p * (dp - sum_x p dp)
__global__ __launch_bounds__(256) void row_softmax_mul_bwd_kernel(const float* __restrict__ score, const float* __restrict__ value,
                                                                  const float* __restrict__ dout, float* __restrict__ dscore, float* __restrict__ dvalue,
                                                                  int C, int H, int W) {
  __shared__ float s_red[4];
  __shared__ float s_p[4096];
  __shared__ float s_dp[4096];
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
    s_p[x] = e;
    s_dp[x] = 0.0f;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (s_red[0] + s_red[1] + s_red[2] + s_red[3]);
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const size_t base = (((size_t)n * C + c) * H + y) * W;
    for (int x = threadIdx.x; x < W; x += blockDim.x) {      // a thread owns the same columns in every pass: no race on s_dp
      const float g = dout[base + x];
      if (dvalue) dvalue[base + x] = g * (s_p[x] * inv);
      s_dp[x] += g * value[base + x];
    }
  }
  float dot = 0.0f;
  for (int x = threadIdx.x; x < W; x += blockDim.x) dot += s_p[x] * inv * s_dp[x];
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  if (dscore)
    for (int x = threadIdx.x; x < W; x += blockDim.x) dscore[((size_t)n * H + y) * W + x] = s_p[x] * inv * (s_dp[x] - dot);
}

// backward of depth_to_space_kernel (fpn_ops.hip): dx[n, c r r + i r + j, h, w] = dy[n, c_off + c, h r + i, w r + j]
__global__ __launch_bounds__(256) void depth_to_space_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int Cout, int H, int W, int r,
                                                                 int c_off, int Ctot) {
  const int OH = H * r, OW = W * r;
  const size_t total = (size_t)N * Cout * r * r * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(e % W);
    size_t q = e / W;
    const int h = (int)(q % H);
    q /= H;
    const int cs = (int)(q % ((size_t)Cout * r * r));
    const size_t n = q / ((size_t)Cout * r * r);
    const int c = cs / (r * r), ij = cs - c * r * r, i = ij / r, j = ij - i * r;
    dx[e] = dy[((n * Ctot + c_off + c) * OH + (size_t)h * r + i) * (size_t)OW + (size_t)w * r + j];
  }
}

// ---- semanticFCN_opt: F.interpolate(scale_factor = s, mode = 'bilinear', align_corners = False) backward, gather form --------------------
// forward (fpn_ops.hip: bilinear_upsample_kernel): src = max((o + 0.5) / s - 0.5, 0); i0 = floor(src), i1 = min(i0 + 1, n - 1), l = src - i0
__device__ __forceinline__ void bilin_src(int o, int s, int n, int& i0, int& i1, float& l) {
  float src = ((float)o + 0.5f) * (1.0f / (float)s) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l = src - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_upsample_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int H, int W, int s) {
  const int OH = H * s, OW = W * s;
  const size_t total = (size_t)NC * H * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(e % W);
    const size_t r = e / W;
    const int iy = (int)(r % H);
    const float* g = dy + (r / H) * (size_t)OH * OW;
    // output rows / columns whose two source taps can include iy / ix
    const int oy_lo = max(0, (iy - 1) * s), oy_hi = min(OH - 1, (iy + 2) * s - 1);
    const int ox_lo = max(0, (ix - 1) * s), ox_hi = min(OW - 1, (ix + 2) * s - 1);
    float acc = 0.0f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      int y0, y1;
      float ly;
      bilin_src(oy, s, H, y0, y1, ly);
      float wy = 0.0f;
      if (y0 == iy) wy += 1.0f - ly;
      if (y1 == iy) wy += ly;
      if (wy == 0.0f) continue;
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx;
        bilin_src(ox, s, W, x0, x1, lx);
        float wx = 0.0f;
        if (x0 == ix) wx += 1.0f - lx;
        if (x1 == ix) wx += lx;
        if (wx != 0.0f) acc += wy * wx * g[(size_t)oy * OW + ox];
      }
    }
    dx[e] = acc;
  }
}

// GroupNorm backward, one workgroup per (sample, group): x_hat = (x - mean) rstd;  dx = rstd (g - mean(g) - x_hat mean(g x_hat)) with
// g = dy gamma (dy masked by the ReLU that followed when relu != 0: y > 0);  dgamma += sum dy x_hat, dbeta += sum dy (fp64 atomics over samples)
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
                                                            float* __restrict__ dx, double* __restrict__ dgamma, double* __restrict__ dbeta, int C, int HW, int groups, int relu) {
  __shared__ double s_red[2][4];
  const int n = blockIdx.x / groups, gidx = blockIdx.x % groups, cpg = C / groups;
  const float mean = mean_a[blockIdx.x], rstd = rstd_a[blockIdx.x];
  const size_t base = ((size_t)n * C + (size_t)gidx * cpg) * HW;
  const size_t cnt = (size_t)cpg * HW;
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < cpg; ++c) {
    const float ga = gamma ? gamma[gidx * cpg + c] : 1.0f;
    double a1 = 0.0, a2 = 0.0;
    for (size_t i = threadIdx.x; i < (size_t)HW; i += blockDim.x) {
      const size_t e = base + (size_t)c * HW + i;
      float g = dy[e];
      if (relu && !(y[e] > 0.0f)) g = 0.0f;
      const float xh = (x[e] - mean) * rstd;
      a1 += (double)g;
      a2 += (double)g * xh;
    }
    a1 = wave_sum(a1);
    a2 = wave_sum(a2);
    if ((threadIdx.x & 63) == 0) s_red[0][threadIdx.x >> 6] = a1, s_red[1][threadIdx.x >> 6] = a2;
    __syncthreads();
    const double t1 = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3], t2 = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
    __syncthreads();
    if (threadIdx.x == 0) {
      if (dbeta) atomicAdd(&dbeta[gidx * cpg + c], t1);
      if (dgamma) atomicAdd(&dgamma[gidx * cpg + c], t2);
    }
    s1 += t1 * ga;
    s2 += t2 * ga;
  }
  const float m1 = (float)(s1 / (double)cnt), m2 = (float)(s2 / (double)cnt);
  for (int c = 0; c < cpg; ++c) {
    const float ga = gamma ? gamma[gidx * cpg + c] : 1.0f;
    for (size_t i = threadIdx.x; i < (size_t)HW; i += blockDim.x) {
      const size_t e = base + (size_t)c * HW + i;
      float g = dy[e];
      if (relu && !(y[e] > 0.0f)) g = 0.0f;
      const float xh = (x[e] - mean) * rstd;
      dx[e] = rstd * (g * ga - m1 - xh * m2);
    }
  }
}

// SpatialAttention gate (semanticFCN_opt.py:73-85): p = softmax over H W of score[n, 0]; out = x p + x.  dx = dout (p + 1);
// dp = sum_c dout x;  dscore = p (dp - sum p dp).  stats[n] = (max, 1 / sum exp) from the forward.  Pass A (all pixels in parallel): dx and dp;
// pass B (one workgroup per sample): the dot product and dscore.
__global__ __launch_bounds__(256) void spatial_gate_bwd_a_kernel(const float* __restrict__ x, const float* __restrict__ score, const float* __restrict__ stats,
                                                                 const float* __restrict__ dout, float* __restrict__ dx, float* __restrict__ dp, int N, int C,
                                                                 size_t HW) {
  const size_t total = (size_t)N * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t i = e % HW, n = e / HW;
    const float p = expf(score[e] - stats[2 * n]) * stats[2 * n + 1];
    float acc = 0.0f;
    for (int c = 0; c < C; ++c) {
      const size_t k = (n * C + c) * HW + i;
      const float g = dout[k];
      acc += g * x[k];
      dx[k] = g * (p + 1.0f);
    }
    dp[e] = acc;
  }
}

__global__ __launch_bounds__(1024) void spatial_gate_bwd_b_kernel(const float* __restrict__ score, const float* __restrict__ stats, const float* __restrict__ dp,
                                                                  float* __restrict__ dscore, size_t HW) {
  __shared__ double s_red[16];
  const size_t n = blockIdx.x;
  const float mx = stats[2 * n], inv = stats[2 * n + 1];
  const float* s = score + n * HW;
  const float* d = dp + n * HW;
  double dot = 0.0;
  for (size_t i = threadIdx.x; i < HW; i += blockDim.x) dot += (double)(expf(s[i] - mx) * inv) * (double)d[i];
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = dot;
  __syncthreads();
  double tot = 0.0;
  for (int k = 0; k < (int)(blockDim.x >> 6); ++k) tot += s_red[k];
  for (size_t i = threadIdx.x; i < HW; i += blockDim.x) dscore[n * HW + i] = expf(s[i] - mx) * inv * (d[i] - (float)tot);
}

}  // namespace

extern "C" int slu_pointwise_fwd(const float* x, float* y, size_t n, int op, float slope, slu_stream_t stream) {
  if (!x || !y || n == 0 || op < 0 || op > 3 || (op == OP_LEAKY && !(slope >= 0.0f))) return SLU_EINVAL;
  hipLaunchKernelGGL(pointwise_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, slu_stream(stream), x, y, n, op, slope);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_pointwise_bwd(const float* dy, const float* y, float* dx, size_t n, int op, float slope, slu_stream_t stream) {
  if (!dy || !y || !dx || n == 0 || op < 0 || op > 3 || (op == OP_LEAKY && !(slope >= 0.0f))) return SLU_EINVAL;
  hipLaunchKernelGGL(pointwise_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, slu_stream(stream), dy, y, dx, n, op, slope);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !dy || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, slu_stream(stream), x, dy, dx, N * C, H, W, OH, OW);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_nearest_down_bwd(const float* dy, float* dx, int N, int C, int H, int W, int factor, slu_stream_t stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0 || factor < 1 || H % factor || W % factor) return SLU_EINVAL;
  hipLaunchKernelGGL(nearest_down_bwd_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, slu_stream(stream), dy, dx, N * C, H, W, factor);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_replace_tail_fwd(const float* x, const float* meta, float* out, int N, int C, int m, int H, int W, slu_stream_t stream) {
  if (!x || !meta || !out || N <= 0 || C <= 0 || m <= 0 || m > C || H <= 0 || W <= 0) return SLU_EINVAL;
  hipLaunchKernelGGL(replace_tail_fwd_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, slu_stream(stream), x, meta, out, N, C, m,
                     (size_t)H * W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_replace_tail_bwd(const float* dout, float* dx, float* dmeta, int N, int C, int m, int H, int W, slu_stream_t stream) {
  if (!dout || (!dx && !dmeta) || N <= 0 || C <= 0 || m <= 0 || m > C || H <= 0 || W <= 0) return SLU_EINVAL;
  hipLaunchKernelGGL(replace_tail_bwd_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, slu_stream(stream), dout, dx, dmeta, N, C, m,
                     (size_t)H * W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_row_softmax_mul_bwd(const float* score, const float* value, const float* dout, float* dscore, float* dvalue, int N, int C, int H,
                                       int W, slu_stream_t stream) {
  if (!score || !value || !dout || (!dscore && !dvalue) || N <= 0 || C <= 0 || H <= 0 || W <= 0 || W > 4096) return SLU_EINVAL;
  hipLaunchKernelGGL(row_softmax_mul_bwd_kernel, dim3((unsigned)(N * H)), dim3(256), 0, slu_stream(stream), score, value, dout, dscore, dvalue, C, H, W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_depth_to_space_bwd(const float* dy, float* dx, int N, int Cout, int H, int W, int r, int c_off, int Ctot, slu_stream_t stream) {
  if (!dy || !dx || N <= 0 || Cout <= 0 || H <= 0 || W <= 0 || r < 1 || c_off < 0 || c_off + Cout > Ctot) return SLU_EINVAL;
  hipLaunchKernelGGL(depth_to_space_bwd_kernel, dim3(grid_for((size_t)N * Cout * r * r * H * W)), dim3(256), 0, slu_stream(stream), dy, dx, N, Cout, H, W,
                     r, c_off, Ctot);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_bilinear_upsample_bwd(const float* dy, float* dx, int N, int C, int H, int W, int scale, slu_stream_t stream) {
  if (!dy || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0 || scale < 1) return SLU_EINVAL;
  hipLaunchKernelGGL(bilinear_upsample_bwd_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, slu_stream(stream), dy, dx, N * C, H, W, scale);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_groupnorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                                 double* dgamma, double* dbeta, int N, int C, int HW, int groups, int relu, slu_stream_t stream) {
  if (!x || !dy || !mean || !rstd || !dx || N <= 0 || C <= 0 || HW <= 0 || groups <= 0 || C % groups || (relu && !y)) return SLU_EINVAL;
  hipLaunchKernelGGL(groupnorm_bwd_kernel, dim3((unsigned)(N * groups)), dim3(256), 0, slu_stream(stream), x, y, dy, gamma, mean, rstd, dx, dgamma, dbeta,
                     C, HW, groups, relu);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_spatial_softmax_gate_bwd(const float* x, const float* score, const float* stats, const float* dout, float* dx, float* dscore,
                                            float* workspace, int N, int C, int HW, slu_stream_t stream) {
  if (!x || !score || !stats || !dout || !dx || !dscore || !workspace || N <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  hipStream_t st = slu_stream(stream);
  hipLaunchKernelGGL(spatial_gate_bwd_a_kernel, dim3(grid_for((size_t)N * HW)), dim3(256), 0, st, x, score, stats, dout, dx, workspace, N, C, (size_t)HW);
  hipLaunchKernelGGL(spatial_gate_bwd_b_kernel, dim3((unsigned)N), dim3(1024), 0, st, score, stats, workspace, dscore, (size_t)HW);
  SLU_CHECK_LAUNCH();
}
