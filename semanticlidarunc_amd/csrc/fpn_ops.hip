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
