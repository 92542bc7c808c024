// Spherical projection of a LiDAR point cloud into the range image (src/dataset/utils.py:61-67,288-349).
//   phi = atan2(y, x), theta = -atan2(sqrt(x^2 + y^2), z) + pi/2, r = sqrt(x^2 + y^2 + z^2)              (float64, as the reference)
//   rows:  bins_h = linspace(theta_min, theta_max, H)[::-1],  idx_h = digitize(theta, bins_h) - 1  (-1 wraps to the last row)
//   cols:  bins_w = linspace(-pi, pi, W)[::-1],               idx_w = digitize(phi, bins_w) - 1
//   the reference writes the points in descending range order, so the NEAREST point of a pixel survives; here the winner is found
//   with a 64-bit atomicMin on the range bits (second pass: smallest point index among equal ranges), then its channels are copied.
// Three passes over N points (~1.2e5 per scan): latency-bound, microseconds; fp64 so that bin assignment matches numpy's.
#include "slu_common.h"

namespace {

constexpr double kPi = 3.141592653589793238462643383279502884;

// numpy.linspace(start, stop, n)[i]: arange(n) * step + start with step = (stop - start) / (n - 1); the last sample is `stop` exactly
__device__ __forceinline__ double linspace_at(double start, double stop, int n, int i) {
  if (n == 1) return start;
  if (i == n - 1) return stop;
  const double step = (stop - start) / (double)(n - 1);
  return __dadd_rn(__dmul_rn((double)i, step), start);          // no fused multiply-add: numpy rounds twice
}

// numpy.digitize(v, bins) - 1 for the DECREASING bins  b[k] = linspace(lo, hi, n)[n - 1 - k]  (right = False):
//   digitize = smallest i with v >= b[i] (n if none); a -1 result indexes the last row / column, as numpy's negative index does
__device__ __forceinline__ int digitize_desc(double v, double lo, double hi, int n) {
  int a = 0, b = n;                       // invariant: v < bins[k] for k < a, v >= bins[k] for k >= b
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (v >= linspace_at(lo, hi, n, n - 1 - mid)) b = mid; else a = mid + 1;
  }
  const int idx = a - 1;
  return idx < 0 ? n - 1 : idx;
}

__device__ __forceinline__ void angles(const double* p, double& phi, double& theta, double& r) {
  const double x = p[0], y = p[1], z = p[2];
  const double xy = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
  r = sqrt(__dadd_rn(xy, __dmul_rn(z, z)));
  phi = atan2(y, x);
  theta = -atan2(sqrt(xy), z) + kPi / 2;
}

__device__ __forceinline__ unsigned long long orderable(double v) {         // monotone map double -> uint64
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(256) void theta_minmax_kernel(const double* __restrict__ pc, int N, int C, unsigned long long* __restrict__ mm) {
  unsigned long long lo = ~0ull, hi = 0ull;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    double phi, theta, r;
    angles(pc + (size_t)i * C, phi, theta, r);
    const unsigned long long k = orderable(theta);
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if ((threadIdx.x & 63) == 0) { atomicMin(&mm[0], lo); atomicMax(&mm[1], hi); }
}

__device__ __forceinline__ double from_orderable(unsigned long long k) {
  const unsigned long long u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  return __longlong_as_double((long long)u);
}

// numpy.digitize(v, bins) - 1 for an explicit monotone array (right = False): decreasing bins -> #{bins > v} - 1, increasing bins ->
// #{bins <= v} - 1 (both counts are prefixes of the array); -1 wraps to the last row like numpy's negative index
__device__ __forceinline__ int digitize_bins(double v, const double* __restrict__ bins, int n, int increasing) {
  int a = 0, b = n;                       // the count lies in [a, b]
  while (a < b) {
    const int mid = (a + b) >> 1;
    const bool in_prefix = increasing ? (bins[mid] <= v) : (bins[mid] > v);
    if (in_prefix) a = mid + 1; else b = mid;
  }
  const int idx = a - 1;
  return idx < 0 ? n - 1 : idx;
}

// pass 1: pixel of every point, nearest (or, keep_farthest, farthest) range per pixel
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ pc, int N, int C, int H, int W, int use_data_range, double tmin, double tmax,
                                                      const unsigned long long* __restrict__ mm, const double* __restrict__ bins_h, int bins_increasing,
                                                      int keep_farthest, int* __restrict__ pixel, unsigned long long* __restrict__ best_r) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (use_data_range) { tmin = from_orderable(mm[0]); tmax = from_orderable(mm[1]); }
  double phi, theta, r;
  angles(pc + (size_t)i * C, phi, theta, r);
  const int row = bins_h ? digitize_bins(theta, bins_h, H, bins_increasing) : digitize_desc(theta, tmin, tmax, H);
  const int col = digitize_desc(phi, -kPi, kPi, W);
  const int px = row * W + col;
  pixel[i] = px;
  const unsigned long long rb = (unsigned long long)__double_as_longlong(r);      // r >= 0: the bit pattern is monotone
  if (keep_farthest) atomicMax(&best_r[px], rb); else atomicMin(&best_r[px], rb);
}

// pass 2: among the points at the winning range of their pixel, the smallest index (nearest mode: the reference's descending argsort
// keeps an unspecified one of them) / the largest index (farthest mode)
__global__ __launch_bounds__(256) void winner_kernel(const double* __restrict__ pc, int N, int C, const int* __restrict__ pixel,
                                                     const unsigned long long* __restrict__ best_r, int keep_farthest, int* __restrict__ winner) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double phi, theta, r;
  angles(pc + (size_t)i * C, phi, theta, r);
  if ((unsigned long long)__double_as_longlong(r) == best_r[pixel[i]]) {
    if (keep_farthest) atomicMax(&winner[pixel[i]], i); else atomicMin(&winner[pixel[i]], i);
  }
}

// pass 3: copy the winner's channels (float32 image, zeros where no point fell).  flip: the horizontal-flip augmentation of the
// dataloaders (dataloader_semantic_KITTI.py:72-74: columns reversed, y negated) applied while writing.
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ pc, int C, int H, int W, const int* __restrict__ winner, int none,
                                                     int flip, float* __restrict__ img) {
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= H * W) return;
  const int w = winner[px];
  const int row = px / W, col = px - row * W;
  float* o = img + ((size_t)row * W + (flip ? W - 1 - col : col)) * C;
  for (int c = 0; c < C; ++c) {
    float v = w == none ? 0.0f : (float)pc[(size_t)w * C + c];
    if (flip && c == 1) v = -v;
    o[c] = v;
  }
}

__global__ void theta_range_kernel(const unsigned long long* __restrict__ mm, int use_data_range, double tmin, double tmax, double* __restrict__ out) {
  out[0] = use_data_range ? from_orderable(mm[0]) : tmin;
  out[1] = use_data_range ? from_orderable(mm[1]) : tmax;
}

size_t carve_proj(char* base, int N, int HW, unsigned long long** mm, int** pixel, unsigned long long** best, int** winner) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
  char* m = take(2 * sizeof(unsigned long long));
  char* px = take((size_t)N * sizeof(int));
  char* b = take((size_t)HW * sizeof(unsigned long long));
  char* w = take((size_t)HW * sizeof(int));
  if (base) { *mm = (unsigned long long*)m; *pixel = (int*)px; *best = (unsigned long long*)b; *winner = (int*)w; }
  return off;
}

}  // namespace

extern "C" size_t slu_spherical_projection_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return carve_proj(nullptr, N, H * W, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int slu_spherical_projection_ex(const double* pc, int N, int C, int H, int W, int use_data_theta_range, double theta_min, double theta_max,
                                           const double* bins_h, int bins_increasing, int keep_farthest, int flip, void* workspace,
                                           size_t workspace_bytes, float* img, double* theta_range_out, slu_stream_t stream) {
  if (!pc || !workspace || !img || N <= 0 || C < 3 || H <= 0 || W <= 0 || (long long)H * W > 0x7ffffffe) return SLU_EINVAL;
  unsigned long long *mm, *best;
  int *pixel, *winner;
  if (carve_proj((char*)workspace, N, H * W, &mm, &pixel, &best, &winner) > workspace_bytes) return SLU_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return SLU_EINVAL;
  hipStream_t st = slu_stream(stream);
  const int HW = H * W;
  // theta min -> all ones, theta max -> 0; winning range -> all ones (nearest: above any finite range) / 0 (farthest);
  // winner -> INT_MAX (nearest) / -1 (farthest) = "no point"
  const int none = keep_farthest ? -1 : 0x7fffffff;
  if (hipMemsetAsync(mm, 0xff, sizeof(unsigned long long), st) != hipSuccess || hipMemsetAsync(mm + 1, 0, sizeof(unsigned long long), st) != hipSuccess ||
      hipMemsetAsync(best, keep_farthest ? 0x00 : 0xff, (size_t)HW * sizeof(unsigned long long), st) != hipSuccess)
    return SLU_ELAUNCH;
  const unsigned nbp = (unsigned)((N + 255) / 256), nbx = (unsigned)((HW + 255) / 256);
  if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(winner), none, (size_t)HW, st) != hipSuccess) return SLU_ELAUNCH;
  const int data_range = use_data_theta_range && !bins_h;
  if (use_data_theta_range)
    hipLaunchKernelGGL(theta_minmax_kernel, dim3(nbp > 256 ? 256 : nbp), dim3(256), 0, st, pc, N, C, mm);
  hipLaunchKernelGGL(project_kernel, dim3(nbp), dim3(256), 0, st, pc, N, C, H, W, data_range, theta_min, theta_max, mm, bins_h, bins_increasing,
                     keep_farthest, pixel, best);
  hipLaunchKernelGGL(winner_kernel, dim3(nbp), dim3(256), 0, st, pc, N, C, pixel, best, keep_farthest, winner);
  hipLaunchKernelGGL(gather_kernel, dim3(nbx), dim3(256), 0, st, pc, C, H, W, winner, none, flip, img);
  if (theta_range_out)
    hipLaunchKernelGGL(theta_range_kernel, dim3(1), dim3(1), 0, st, mm, use_data_theta_range, theta_min, theta_max, theta_range_out);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_spherical_projection(const double* pc, int N, int C, int H, int W, int use_data_theta_range, double theta_min, double theta_max,
                                        void* workspace, size_t workspace_bytes, float* img, double* theta_range_out, slu_stream_t stream) {
  return slu_spherical_projection_ex(pc, N, C, H, W, use_data_theta_range, theta_min, theta_max, nullptr, 0, 0, 0, workspace, workspace_bytes, img,
                                     theta_range_out, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// The two ends of the dataloader's __getitem__ around the projection (dataloader_semantic_KITTI.py:31-99):
//   decode   .bin float32 [N][4] (x, y, z, intensity) + .label uint32 [N] -> float64 [N][5] (x, y, z, i, id_map[label & 0xFFFF]),
//            optional yaw rotation (rotate_z, dataset/utils.py:4-18: points @ R, float64)                                      (:36-56)
//   split    projected image [H][W][C >= 5] (+ normals [H][W][3]) -> range = |xyz| (float32, (x^2 + y^2) + z^2 as numpy sums it),
//            reflectivity, xyz and normals channel-first, semantics int64                                                       (:83-99)
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void kitti_decode_kernel(const float* __restrict__ xyzi, const unsigned* __restrict__ label, int N,
                                                           const int* __restrict__ lut, int lut_size, int rotate, double ca, double sa,
                                                           double* __restrict__ pc, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float4 p = reinterpret_cast<const float4*>(xyzi)[i];
  double x = (double)p.x, y = (double)p.y, z = (double)p.z;
  if (rotate) {                           // [x y z] @ [[c, -s, 0], [s, c, 0], [0, 0, 1]]: plain products and sums, no fused multiply-add
    const double xr = __dadd_rn(__dadd_rn(__dmul_rn(x, ca), __dmul_rn(y, sa)), __dmul_rn(z, 0.0));
    const double yr = __dadd_rn(__dadd_rn(__dmul_rn(x, -sa), __dmul_rn(y, ca)), __dmul_rn(z, 0.0));
    x = xr;
    y = yr;
  }
  const unsigned sem = label[i] & 0xFFFFu;
  int cls = sem < (unsigned)lut_size ? lut[sem] : -1;
  if (cls < 0) {                          // the reference's dict lookup raises KeyError: counted here, raised by the host wrapper
    atomicAdd(bad, 1);
    cls = 0;
  }
  double* o = pc + (size_t)i * 5;
  o[0] = x; o[1] = y; o[2] = z; o[3] = (double)p.w; o[4] = (double)cls;
}

__global__ __launch_bounds__(256) void range_image_split_kernel(const float* __restrict__ img, const float* __restrict__ normals, int HW, int C,
                                                                float* __restrict__ range, float* __restrict__ refl, float* __restrict__ xyz,
                                                                float* __restrict__ normals_chw, int64_t* __restrict__ labels) {
#pragma clang fp contract(off)
  const int px = blockIdx.x * blockDim.x + threadIdx.x;
  if (px >= HW) return;
  const float* p = img + (size_t)px * C;
  const float x = p[0], y = p[1], z = p[2];
  range[px] = sqrtf((x * x + y * y) + z * z);
  refl[px] = p[3];
  xyz[px] = x; xyz[HW + px] = y; xyz[2 * (size_t)HW + px] = z;
  labels[px] = (int64_t)p[4];
  if (normals) {
    normals_chw[px] = normals[(size_t)px * 3];
    normals_chw[HW + px] = normals[(size_t)px * 3 + 1];
    normals_chw[2 * (size_t)HW + px] = normals[(size_t)px * 3 + 2];
  }
}

// cv2.resize(img, (OW, OH), interpolation=INTER_NEAREST) of an [H][W][C] image (dataloader_semantic_KITTI.py:61-62): OpenCV's nearest
// neighbour takes source index min(floor(dst * (src_size / dst_size)), src_size - 1) per axis (the ratio in double precision); flip = the
// dataloader's augmentation applied AFTER the resize (:71-73: columns reversed, y negated)
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ img, int H, int W, int C, float* __restrict__ out, int OH, int OW,
                                                             double fy, double fx, int flip) {
  const size_t total = (size_t)OH * OW * C;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % C);
    const size_t px = e / C;
    int ox = (int)(px % OW);
    const int oy = (int)(px / OW);
    if (flip) ox = OW - 1 - ox;
    int sy = (int)floor((double)oy * fy), sx = (int)floor((double)ox * fx);
    sy = sy < H - 1 ? sy : H - 1;
    sx = sx < W - 1 ? sx : W - 1;
    float v = img[((size_t)sy * W + sx) * C + c];
    if (flip && c == 1) v = -v;
    out[e] = v;
  }
}

}  // namespace

extern "C" int slu_resize_nearest_hwc(const float* img, int H, int W, int C, float* out, int OH, int OW, int flip, slu_stream_t stream) {
  if (!img || !out || H <= 0 || W <= 0 || C <= 0 || OH <= 0 || OW <= 0 || (flip && C < 2)) return SLU_EINVAL;
  const size_t total = (size_t)OH * OW * C;
  const size_t nb = (total + 255) / 256;
  hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)(nb > 65535 ? 65535 : nb)), dim3(256), 0, slu_stream(stream), img, H, W, C, out, OH, OW,
                     (double)H / (double)OH, (double)W / (double)OW, flip ? 1 : 0);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_kitti_decode(const float* xyzi, const uint32_t* label, int N, const int32_t* lut, int lut_size, int rotate, double cos_a, double sin_a,
                                double* pc, int32_t* bad_count, slu_stream_t stream) {
  if (!xyzi || !label || !lut || !pc || !bad_count || N <= 0 || lut_size <= 0 || ((uintptr_t)xyzi & 15)) return SLU_EINVAL;
  hipLaunchKernelGGL(kitti_decode_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, slu_stream(stream), xyzi, label, N, lut, lut_size, rotate,
                     cos_a, sin_a, pc, bad_count);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_range_image_split(const float* img, const float* normals, int H, int W, int C, float* range, float* refl, float* xyz,
                                     float* normals_chw, int64_t* labels, slu_stream_t stream) {
  if (!img || !range || !refl || !xyz || !labels || H <= 0 || W <= 0 || C < 5 || (normals && !normals_chw) || (long long)H * W > 0x7ffffffe) return SLU_EINVAL;
  const int HW = H * W;
  hipLaunchKernelGGL(range_image_split_kernel, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, slu_stream(stream), img, normals, HW, C, range, refl,
                     xyz, normals_chw, labels);
  SLU_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// Surface normals of a staggered range image (dataset/utils.py:30-58 build_normal_xyz, called by every
// dataloader and by inference_ouster.py:70).  Per coordinate plane two 3x3 Scharr derivatives -- OpenCV's cv2.Scharr
// (opencv-python 4.11.0.86 in docker/requirements.txt: taps [-1 0 1] x [3 10 3], border BORDER_REFLECT_101, result times
// `scale`) -- then n = -(d/dcol x d/drow) per pixel, divided by (|n| + 1e-10).  One lane per pixel; the 27 neighbours come
// through L1/L2 (a 64x2048x3 image is 1.5 MB).
// ---------------------------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int reflect101(int i, int n) {       // gfedcb|abcdefgh|gfedcba ; n == 1: always 0
  if (n == 1) return 0;
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ xyz, int H, int W, int stride_px, float scale,
                                                      float* __restrict__ out) {
  // no FMA contraction here: where the two tangent vectors are parallel the reference's a*b - b*a is exactly 0 (-> the zero
  // normal), while fma(a, b, -(b*a)) leaves the rounding error of one product, which the normalisation would blow up to length 1
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int xm = reflect101(x - 1, W), xp = reflect101(x + 1, W), ym = reflect101(y - 1, H), yp = reflect101(y + 1, H);
  float dcol[3], drow[3];                                        // d/dx (along a row) and d/dy (along a column) of the X, Y, Z planes
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    auto at = [&](int yy, int xx) { return xyz[((size_t)yy * W + xx) * stride_px + c]; };
    // separable, rows first, in float32 (the order OpenCV's FilterEngine uses)
    // (symmetric taps summed first, like OpenCV's symmetric small-kernel filters; `scale` sits in the smoothing taps there)
    const float r_m = at(ym, xp) - at(ym, xm), r_0 = at(y, xp) - at(y, xm), r_p = at(yp, xp) - at(yp, xm);
    dcol[c] = (10.0f * scale) * r_0 + (3.0f * scale) * (r_m + r_p);
    const float s_m = (10.0f * scale) * at(ym, x) + (3.0f * scale) * (at(ym, xm) + at(ym, xp));
    const float s_p = (10.0f * scale) * at(yp, x) + (3.0f * scale) * (at(yp, xm) + at(yp, xp));
    drow[c] = s_p - s_m;
  }
  // Sxx = dcol[0], Sxy = drow[0], Syx = dcol[1], Syy = drow[1], Szx = dcol[2], Szy = drow[2]
  float n0 = -(dcol[1] * drow[2] - dcol[2] * drow[1]);
  float n1 = -(dcol[2] * drow[0] - drow[2] * dcol[0]);
  float n2 = -(dcol[0] * drow[1] - dcol[1] * drow[0]);
  const float len = sqrtf(n0 * n0 + n1 * n1 + n2 * n2) + 1e-10f;
  float* o = out + ((size_t)y * W + x) * 3;
  o[0] = n0 / len;
  o[1] = n1 / len;
  o[2] = n2 / len;
}

}  // namespace

extern "C" int slu_build_normals(const float* xyz, int H, int W, int channels, float norm_factor, float* normals, slu_stream_t stream) {
  if (!xyz || !normals || H <= 0 || W <= 0 || channels < 3 || !(norm_factor > 0.0f)) return SLU_EINVAL;
  if (H > 65535) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL(normals_kernel, dim3((W + 255) / 256, H), dim3(256), 0, slu_stream(stream), xyz, H, W, channels, 1.0f / norm_factor, normals);
  SLU_CHECK_LAUNCH();
}
