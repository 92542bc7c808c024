// EfficientNetV2 pieces of the semanticFCN_opt encoder (reference baselines/Reichert/semanticFCN_opt.py:170-180,238-247,396-404 builds
// torchvision's efficientnet_v2_{s,m,l} and uses features[0], [2], [3], [4]: FusedMBConv and MBConv blocks).  The dense convs (3x3 expansions,
// 1x1 expansions / projections, with folded BatchNorm, SiLU, per-(sample, channel) SE multipliers and the residual) are launches of the fused
// conv kernel; what is new here is HBM-bound:
//   dwconv3x3_kernel      depthwise 3x3 (stride 1 / 2) + folded BatchNorm + SiLU: one lane per output pixel, lanes azimuth-adjacent
//   global_avgpool_kernel SqueezeExcitation's AdaptiveAvgPool2d(1): one wave per (sample, channel) plane
//   se_gate_kernel        fc1 -> SiLU -> fc2 -> sigmoid on the pooled vector: one workgroup per sample (<= 1536 channels x <= 96 squeezed)
//   dwconv3x3_wgrad_kernel  training: dL/dw[c][tap] = sum over (sample, pixel) of dy * shifted x -- one workgroup per channel, nine fp64 block sums
//                           (the data gradient of the stride-1 depthwise conv is the forward kernel with the taps reversed)
#include "slu_common.h"

namespace {

__device__ __forceinline__ float silu(float v) { return v / (1.0f + expf(-v)); }

template <int STRIDE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int NC, int C, int H, int W, int OH, int OW, int act) {
  const size_t total = (size_t)NC * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    const size_t r = e / OW;
    const int oy = (int)(r % OH);
    const size_t nc = r / OH;
    const int c = (int)(nc % C);
    const float* p = x + nc * (size_t)H * W;
    const float* wc = w + (size_t)c * 9;
    float acc = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int iy = oy * STRIDE + i - 1;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ix = ox * STRIDE + j - 1;
        if (ix >= 0 && ix < W) acc = fmaf(wc[i * 3 + j], p[(size_t)iy * W + ix], acc);
      }
    }
    y[e] = act == 3 ? silu(acc) : acc;
  }
}

__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* __restrict__ x, float* __restrict__ out, int NC, int HW) {
  const int plane = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= NC) return;
  const float* p = x + (size_t)plane * HW;
  double s = 0.0;
  for (int i = threadIdx.x & 63; i < HW; i += 64) s += (double)p[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) out[plane] = (float)(s / (double)HW);
}

__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ avg, const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ scale, int C, int S) {
  extern __shared__ float s_mem[];      // [C] pooled vector | [S] squeezed activations
  float* s_avg = s_mem;
  float* s_hid = s_mem + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) s_avg[c] = avg[(size_t)n * C + c];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int s = wave; s < S; s += nw) {      // one wave per squeezed unit: a dot product over C
    float acc = 0.0f;
    for (int c = lane; c < C; c += 64) acc = fmaf(w1[(size_t)s * C + c], s_avg[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) s_hid[s] = silu(acc + (b1 ? b1[s] : 0.0f));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = b2 ? b2[c] : 0.0f;
    for (int s = 0; s < S; ++s) acc = fmaf(w2[(size_t)c * S + s], s_hid[s], acc);
    scale[(size_t)n * C + c] = 1.0f / (1.0f + expf(-acc));
  }
}

__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, int N,
                                                              int C, int H, int W) {
  const int c = blockIdx.x;
  const size_t HW = (size_t)H * W;
  double acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* px = x + ((size_t)n * C + c) * HW;
    const float* pg = dy + ((size_t)n * C + c) * HW;
    float part[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) part[t] = 0.0f;
    for (size_t e = threadIdx.x; e < HW; e += blockDim.x) {
      const int xx = (int)(e % W), yy = (int)(e / W);
      const float g = pg[e];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int iy = yy + i - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int ix = xx + j - 1;
          if (ix >= 0 && ix < W) part[i * 3 + j] = fmaf(g, px[(size_t)iy * W + ix], part[i * 3 + j]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] += (double)part[t];      // fp32 within one plane and lane (<= HW / 256 terms), fp64 across planes and lanes
  }
  __shared__ double s_red[4][9];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const double v = wave_sum(acc[t]);
    if (lane == 0) s_red[wave][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 9) dw[(size_t)c * 9 + threadIdx.x] = (float)(s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

inline unsigned grid1d(size_t total) {
  const size_t nb = (total + 255) / 256;
  return (unsigned)(nb > 65535 ? 65535 : (nb ? nb : 1));
}

}  // namespace

extern "C" int slu_dwconv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int stride, int act,
                                 slu_stream_t stream) {
  if (!x || !w || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (stride != 1 && stride != 2) || (act != 0 && act != 3)) return SLU_EINVAL;
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;      // floor((H + 2 - 3) / stride) + 1
  const size_t total = (size_t)N * C * OH * OW;
  if (stride == 1)
    hipLaunchKernelGGL(dwconv3x3_kernel<1>, dim3(grid1d(total)), dim3(256), 0, slu_stream(stream), x, w, bias, y, N * C, C, H, W, OH, OW, act);
  else
    hipLaunchKernelGGL(dwconv3x3_kernel<2>, dim3(grid1d(total)), dim3(256), 0, slu_stream(stream), x, w, bias, y, N * C, C, H, W, OH, OW, act);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !dy || !dw || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SLU_EINVAL;
  hipLaunchKernelGGL(dwconv3x3_wgrad_kernel, dim3((unsigned)C), dim3(256), 0, slu_stream(stream), x, dy, dw, N, C, H, W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_global_avgpool(const float* x, float* out, int N, int C, int HW, slu_stream_t stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  const long long planes = (long long)N * C;
  hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, slu_stream(stream), x, out, (int)planes, HW);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_se_gate(const float* avg, const float* w1, const float* b1, const float* w2, const float* b2, float* scale, int N, int C, int S,
                           slu_stream_t stream) {
  if (!avg || !w1 || !w2 || !scale || N <= 0 || C <= 0 || S <= 0 || (size_t)(C + S) * 4 > 64 * 1024) return SLU_EINVAL;
  hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)N), dim3(256), (size_t)(C + S) * 4, slu_stream(stream), avg, w1, b1, w2, b2, scale, C, S);
  SLU_CHECK_LAUNCH();
}
