// Lovasz-Softmax on gfx950 (replaces src/losses/lovasz.py:12-88 of the reference).
//
//   for every class c present among the valid labels:
//       err_i = | 1[y_i == c] - p_c,i |            (valid pixels; ignored pixels are dropped)
//       sort err descending, fg in the same order, J_k = 1 - (G - cumfg_k) / (G + k - cumfg_k)
//       loss_c = sum_k err_(k) * (J_k - J_{k-1})
//   loss = mean over present classes
//
// Device formulation.  Ignored pixels are kept in place with err = 0 / fg = 0: they sort behind every
// positive error, so they change no J_k that multiplies a non-zero error -- the value is identical and
// no compaction pass is needed.  All classes are sorted at once: a batched LSD radix sort (4 x 8 bit)
// over [C][N] (key = 0x3F800000 - bits(err), i.e. ascending key == descending err in [0,1]; value =
// pixel index | fg << 31).  Per pass: per-tile digit histograms -> one exclusive scan per class ->
// stable scatter (the rank of an element inside its wave comes from 8 ballots over the digit bits,
// 64-lane waves; per-wave digit counters in LDS).  Classes with no foreground exit every kernel at once.
// After the sort one kernel scans the fg bits, forms J_k in fp32 exactly like the reference
// (1 - inter/union, both exact integers in fp32), accumulates loss_c in fp64 and scatters
// d loss / d p_c,i = (J_k - J_{k-1}) * (fg ? -1 : +1) back to pixel order (0 where err == 0).
//
// HBM-bound: ~ 4 passes x (8 B read for the histogram + 8 B read + 8 B written by the scatter) per key.
#include "slu_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kItems = 8;                       // keys per thread per tile
constexpr int kTile = kThreads * kItems;        // 2048 keys per workgroup
constexpr unsigned kOne = 0x3F800000u;          // bits(1.0f)

struct Ws {
  unsigned* keys[2];
  unsigned* vals[2];
  unsigned* hist;      // [C][256][nblk]
  unsigned* G;         // [C] foreground count per class (valid pixels)
  unsigned* blkfg;     // [C][nblk]
  double* loss_c;      // [C]
  unsigned* act;       // [C] 1 = the class is summed whether or not it is present (classes='all' / an explicit list); unused for 'present'
};

__host__ __device__ inline int nblk_of(long long n) { return (int)((n + kTile - 1) / kTile); }

// ---- keys / values / per-class foreground counts -------------------------------------------------
__global__ __launch_bounds__(kThreads) void keygen_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels,
                                                          int B, int C, int HW, int64_t ignore, unsigned* __restrict__ keys,
                                                          unsigned* __restrict__ vals, unsigned* __restrict__ G) {
  __shared__ unsigned s_cnt[32];
  const int c = blockIdx.y;
  if (threadIdx.x < 32) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long N = (long long)B * HW;
  unsigned local = 0;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < N; i += (long long)gridDim.x * kThreads) {
    const int64_t y = labels[i];
    const int b = (int)(i / HW);
    const int hw = (int)(i - (long long)b * HW);
    const float p = probs[((size_t)b * C + c) * HW + hw];
    const bool valid = y != ignore;
    const bool fg = valid && y == c;
    float err = valid ? fabsf((fg ? 1.0f : 0.0f) - p) : 0.0f;
    err = fminf(fmaxf(err, 0.0f), 1.0f);
    keys[(size_t)c * N + i] = kOne - __float_as_uint(err);
    vals[(size_t)c * N + i] = (unsigned)i | (fg ? 0x80000000u : 0u);
    local += fg ? 1u : 0u;
  }
  local = (unsigned)wave_sum((float)local);     // <= 64 * iterations, exact in fp32 for any sane grid
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(&s_cnt[0], local);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt[0]) atomicAdd(&G[c], s_cnt[0]);
}

// ---- radix pass: histogram ------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void hist_kernel(const unsigned* __restrict__ keys, long long N, int nblk, int shift,
                                                        const unsigned* __restrict__ G, unsigned* __restrict__ hist) {
  const int c = blockIdx.y, blk = blockIdx.x;
  if (G[c] == 0) return;
  __shared__ unsigned s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const long long base = (long long)blk * kTile;
  const unsigned* k = keys + (size_t)c * N;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * kThreads + threadIdx.x;
    if (i < N) atomicAdd(&s_h[(k[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[((size_t)c * nblk + blk) * 256 + threadIdx.x] = s_h[threadIdx.x];      // [class][tile][digit]: the scan reads 1 KB rows
}

// ---- radix pass: exclusive scan over (digit, tile) of one class (one workgroup per class) ---------
__global__ __launch_bounds__(256) void scan_kernel(unsigned* __restrict__ hist, int nblk, const unsigned* __restrict__ G) {
  const int c = blockIdx.x;
  if (G[c] == 0) return;
  __shared__ unsigned s_tot[256];
  // thread = digit; the tiles of a digit are 1 KB apart, the 256 digits of a tile contiguous: every iteration is one coalesced row and the
  // rows are independent loads (the [digit][tile] layout made each thread walk its own 4-byte-strided column: 98 us per pass)
  unsigned* col = hist + (size_t)c * nblk * 256 + threadIdx.x;
  unsigned tot = 0;
#pragma unroll 8
  for (int b = 0; b < nblk; ++b) tot += col[(size_t)b * 256];
  s_tot[threadIdx.x] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {          // 256 values: a serial scan is a few hundred cycles
    unsigned run = 0;
    for (int d = 0; d < 256; ++d) { const unsigned t = s_tot[d]; s_tot[d] = run; run += t; }
  }
  __syncthreads();
  unsigned run = s_tot[threadIdx.x];
#pragma unroll 8
  for (int b = 0; b < nblk; ++b) { const unsigned t = col[(size_t)b * 256]; col[(size_t)b * 256] = run; run += t; }
}

// ---- radix pass: stable scatter --------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void scatter_kernel(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ vals_in,
                                                           unsigned* __restrict__ keys_out, unsigned* __restrict__ vals_out,
                                                           long long N, int nblk, int shift, const unsigned* __restrict__ G,
                                                           const unsigned* __restrict__ hist) {
  const int c = blockIdx.y, blk = blockIdx.x;
  if (G[c] == 0) return;
  constexpr int kWaves = kThreads / 64;
  __shared__ unsigned s_cnt[kWaves][256];       // running count of each digit inside each wave's slice
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < kWaves * 256; i += kThreads) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  // a wave owns kItems consecutive 64-key rows of the tile: key order == (wave, item, lane)
  const long long base = (long long)blk * kTile + (long long)wave * (kItems * 64);
  const size_t seg = (size_t)c * N;
  unsigned key[kItems], val[kItems], rank[kItems];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * 64 + lane;
    const bool in = i < N;
    key[it] = in ? keys_in[seg + i] : 0xFFFFFFFFu;
    val[it] = in ? vals_in[seg + i] : 0u;
    const unsigned d = (key[it] >> shift) & 255u;
    unsigned long long peers = __ballot(in);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const unsigned before = (unsigned)__popcll(peers & lt);
    const unsigned old = in ? s_cnt[wave][d] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (in && before == 0) s_cnt[wave][d] = old + (unsigned)__popcll(peers);    // lowest lane of each digit group
    __builtin_amdgcn_wave_barrier();
    rank[it] = old + before;
  }
  __syncthreads();
  // exclusive offsets of this wave's slice inside the tile, per digit (kWaves is 4: unrolled sum)
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * 64 + lane;
    if (i < N) {
      const unsigned d = (key[it] >> shift) & 255u;
      unsigned off = hist[((size_t)c * nblk + blk) * 256 + d];
#pragma unroll
      for (int w = 0; w < kWaves; ++w)
        if (w < wave) off += s_cnt[w][d];
      const size_t o = seg + off + rank[it];
      keys_out[o] = key[it];
      vals_out[o] = val[it];
    }
  }
}

// ---- after the sort: per-tile foreground counts, then Jaccard steps / loss / gradient scatter -------
__global__ __launch_bounds__(kThreads) void fgcount_kernel(const unsigned* __restrict__ vals, long long N, int nblk,
                                                           const unsigned* __restrict__ G, unsigned* __restrict__ blkfg) {
  const int c = blockIdx.y, blk = blockIdx.x;
  if (G[c] == 0) return;
  __shared__ unsigned s_w[kThreads / 64];
  const long long base = (long long)blk * kTile;
  unsigned n = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * kThreads + threadIdx.x;
    if (i < N) n += vals[(size_t)c * N + i] >> 31;
  }
  n = (unsigned)wave_sum((float)n);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_w[w];
    blkfg[(size_t)c * nblk + blk] = t;
  }
}

__global__ __launch_bounds__(256) void fgscan_kernel(unsigned* __restrict__ blkfg, int nblk, const unsigned* __restrict__ G) {
  const int c = blockIdx.x;                    // serial exclusive scan over <= a few hundred tiles
  if (G[c] == 0 || threadIdx.x != 0) return;
  unsigned run = 0;
  unsigned* row = blkfg + (size_t)c * nblk;
  for (int b = 0; b < nblk; ++b) { const unsigned t = row[b]; row[b] = run; run += t; }
}

__device__ __forceinline__ float jaccard(unsigned G, unsigned long long k, unsigned cum) {
  // lovasz.py:31-33 : 1 - (gts - cumsum(fg)) / (gts + cumsum(1 - fg)), all exact integers in fp32
  return 1.0f - (float)(G - cum) / (float)(G + (unsigned)k - cum);
}

__global__ __launch_bounds__(kThreads) void lovasz_steps_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals,
                                                                long long N, int nblk, const unsigned* __restrict__ G,
                                                                const unsigned* __restrict__ act, const unsigned* __restrict__ blkfg,
                                                                double* __restrict__ loss_c, float* __restrict__ grad /* [C][N] pixel order, or null */) {
  const int c = blockIdx.y, blk = blockIdx.x;
  // an ABSENT class that is summed anyway (g = 0, lovasz.py:66-69 with classes='all'): the Jaccard steps degenerate to (1, 0, 0, ...), the term is
  // the largest error = max p_c and its gradient +1 at that pixel -- the formulas below give exactly that with g = 0
  const unsigned g = G[c];
  if (act[c] == 0) return;
  __shared__ unsigned s_wave[kThreads / 64];
  __shared__ double s_part[kThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread t owns kItems CONSECUTIVE sorted positions, so its prefix is one block scan of its fg total
  const long long first = (long long)blk * kTile + (long long)threadIdx.x * kItems;
  const size_t seg = (size_t)c * N;
  unsigned k_[kItems], v_[kItems];
  unsigned mine = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = first + it;
    k_[it] = i < N ? keys[seg + i] : kOne;
    v_[it] = i < N ? vals[seg + i] : 0u;
    mine += v_[it] >> 31;
  }
  // exclusive scan of `mine` over the workgroup (wave shuffle scan + wave totals)
  unsigned incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned cum = blkfg[(size_t)c * nblk + blk] + incl - mine;
  for (int w = 0; w < wave; ++w) cum += s_wave[w];
  double part = 0.0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = first + it;
    if (i < N) {
      const unsigned fg = v_[it] >> 31;
      const float j_prev = (i == 0) ? 0.0f : jaccard(g, (unsigned long long)i, cum);
      cum += fg;
      const float step = jaccard(g, (unsigned long long)i + 1, cum) - j_prev;
      const float err = __uint_as_float(kOne - k_[it]);
      part += (double)(err * step);
      if (grad) grad[seg + (v_[it] & 0x7FFFFFFFu)] = err > 0.0f ? (fg ? -step : step) : 0.0f;
    }
  }
  part = wave_sum(part);
  if (lane == 0) s_part[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_part[w];
    atomicAdd(&loss_c[c], t);
  }
}

__global__ void lovasz_set_active_kernel(unsigned* __restrict__ act, int C, unsigned mask) {
  if ((int)threadIdx.x < C) act[threadIdx.x] = (mask >> threadIdx.x) & 1u;
}

// loss = mean over the summed classes (G = the activity flags: the foreground counts for 'present'); grad_probs[b][c][hw] = grad[c][b*HW+hw] / n (0 for the others)
__global__ void lovasz_finalize_kernel(const unsigned* __restrict__ G, const double* __restrict__ loss_c, int C,
                                       float* __restrict__ loss_out, float* __restrict__ n_present_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    int n = 0;
    for (int c = 0; c < C; ++c)
      if (G[c]) { s += loss_c[c]; ++n; }
    loss_out[0] = n ? (float)(s / n) : 0.0f;
    n_present_out[0] = (float)n;
  }
}

__global__ __launch_bounds__(256) void lovasz_grad_layout_kernel(const float* __restrict__ grad, const unsigned* __restrict__ G,
                                                                 const float* __restrict__ n_present, int B, int C, int HW,
                                                                 float* __restrict__ grad_probs) {
  const size_t total = (size_t)B * C * HW;
  const float inv = n_present[0] > 0.0f ? 1.0f / n_present[0] : 0.0f;
  const size_t N = (size_t)B * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int hw = (int)(e % HW);
    const size_t r = e / HW;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    grad_probs[e] = G[c] ? grad[(size_t)c * N + (size_t)b * HW + hw] * inv : 0.0f;
  }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t carve(char* base, long long N, int C, Ws* ws) {
  const int nblk = nblk_of(N);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  // zero-initialised block first (G, loss_c): one memset covers it
  char* g = take((size_t)C * sizeof(unsigned));
  char* l = take((size_t)C * sizeof(double));
  char* k0 = take((size_t)C * N * 4); char* k1 = take((size_t)C * N * 4);
  char* v0 = take((size_t)C * N * 4); char* v1 = take((size_t)C * N * 4);
  char* h = take((size_t)C * 256 * nblk * 4);
  char* bf = take((size_t)C * nblk * 4);
  char* ac = take((size_t)C * sizeof(unsigned));
  if (ws) {
    ws->G = (unsigned*)g; ws->loss_c = (double*)l; ws->act = (unsigned*)ac;
    ws->keys[0] = (unsigned*)k0; ws->keys[1] = (unsigned*)k1; ws->vals[0] = (unsigned*)v0; ws->vals[1] = (unsigned*)v1;
    ws->hist = (unsigned*)h; ws->blkfg = (unsigned*)bf;
  }
  return off;
}

}  // namespace

extern "C" size_t slu_lovasz_workspace_bytes(int B, int C, int HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return 0;
  return carve(nullptr, (long long)B * HW, C, nullptr);
}

extern "C" int slu_lovasz_fwd(const float* probs, const int64_t* labels, int B, int C, int HW, int64_t ignore_index, unsigned class_mask,
                              void* workspace, size_t workspace_bytes, float* loss, float* n_present, float* grad_probs,
                              slu_stream_t stream) {
  if (!probs || !labels || !workspace || !loss || !n_present || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  const long long N = (long long)B * HW;
  if (N >= (1ll << 31) || C > 32) return SLU_EUNSUPPORTED;
  Ws ws;
  if (carve((char*)workspace, N, C, &ws) > workspace_bytes) return SLU_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return SLU_EINVAL;
  hipStream_t st = slu_stream(stream);
  const int nblk = nblk_of(N);
  if (hipMemsetAsync(ws.G, 0, align256((size_t)C * sizeof(unsigned)) + (size_t)C * sizeof(double), st) != hipSuccess) return SLU_ELAUNCH;
  const unsigned kg = (unsigned)((N + kThreads - 1) / kThreads > 1024 ? 1024 : (N + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(keygen_kernel, dim3(kg, C), dim3(kThreads), 0, st, probs, labels, B, C, HW, ignore_index, ws.keys[0], ws.vals[0], ws.G);
  // which classes are sorted and summed: the present ones (flag = foreground count), or the caller's set whether present or not
  const unsigned* act = ws.G;
  if (class_mask) {
    hipLaunchKernelGGL(lovasz_set_active_kernel, dim3(1), dim3(64), 0, st, ws.act, C, class_mask);
    act = ws.act;
  }
  int cur = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(hist_kernel, dim3(nblk, C), dim3(kThreads), 0, st, ws.keys[cur], N, nblk, shift, act, ws.hist);
    hipLaunchKernelGGL(scan_kernel, dim3(C), dim3(256), 0, st, ws.hist, nblk, act);
    hipLaunchKernelGGL(scatter_kernel, dim3(nblk, C), dim3(kThreads), 0, st, ws.keys[cur], ws.vals[cur], ws.keys[cur ^ 1],
                       ws.vals[cur ^ 1], N, nblk, shift, act, ws.hist);
    cur ^= 1;
  }
  hipLaunchKernelGGL(fgcount_kernel, dim3(nblk, C), dim3(kThreads), 0, st, ws.vals[cur], N, nblk, act, ws.blkfg);
  hipLaunchKernelGGL(fgscan_kernel, dim3(C), dim3(256), 0, st, ws.blkfg, nblk, act);
  // per-pixel gradient (class-major, pixel order) reuses the idle value buffer
  float* gtmp = grad_probs ? reinterpret_cast<float*>(ws.keys[cur ^ 1]) : nullptr;
  hipLaunchKernelGGL(lovasz_steps_kernel, dim3(nblk, C), dim3(kThreads), 0, st, ws.keys[cur], ws.vals[cur], N, nblk, ws.G, act, ws.blkfg,
                     ws.loss_c, gtmp);
  hipLaunchKernelGGL(lovasz_finalize_kernel, dim3(1), dim3(64), 0, st, act, ws.loss_c, C, loss, n_present);
  if (grad_probs) {
    const size_t total = (size_t)B * C * HW;
    const unsigned g = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(lovasz_grad_layout_kernel, dim3(g), dim3(256), 0, st, gtmp, act, n_present, B, C, HW, grad_probs);
  }
  SLU_CHECK_LAUNCH();
}

// =====================================================================================================================
// AUROC of error detection (replaces src/metrics/auroc.py:36-78 of the reference).
//
//   score kernel   per pixel: probabilities by mode (alpha / logits / probs, auroc.py:36-45), argmax, the uncertainty score
//                  (entropy, entropy_norm, 1-maxprob; Dirichlet mutual information for mode alpha, :47-63), validity.
//   AUROC          the reference sorts the samples by score (descending), takes cumulative sums and integrates TPR over FPR
//                  with the trapezoid rule (:65-78).  A negative sample at sorted position i advances FPR by 1/N at height
//                  TPR_i = (#positives before i) / P and a positive one advances nothing, so
//                      AUROC = sum over negatives of (#positives ranked before it) / (P N)        -- exact integers.
//                  The samples go through the same batched LSD radix sort as the Lovasz loss (one "class"); the order among
//                  equal scores is arbitrary, as it is for numpy's argsort in the reference.
// =====================================================================================================================
namespace {

__device__ __forceinline__ float digamma_ge1_f(float x) {      // x >= 1: recurrence to x >= 6, then the asymptotic series
  float r = 0.0f;
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if (x < 6.0f) { r -= 1.0f / x; x += 1.0f; }
  const float inv = 1.0f / x, inv2 = inv * inv;
  return r + logf(x) - 0.5f * inv - inv2 * (1.0f / 12.0f - inv2 * (1.0f / 120.0f - inv2 * (1.0f / 252.0f)));
}

// mode: 0 alpha, 1 logits, 2 probs.  kind: 0 entropy, 1 entropy_norm, 2 mi, 3 mi_norm, 4 1-maxprob.
// flag: 0 correct, 1 error, 2 not valid (label == ignore_index)
template <int CMAX>
__global__ __launch_bounds__(256) void auroc_score_kernel(const float* __restrict__ preds, const int64_t* __restrict__ labels,
                                                          const float* __restrict__ score_override, int B, int C, int HW, int mode, int kind,
                                                          int has_ignore, int64_t ignore, float eps, float* __restrict__ score_out,
                                                          uint8_t* __restrict__ flag_out) {
  const size_t npix = (size_t)B * HW;
  const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = (int)(pix / HW);
  const int hw = (int)(pix - (size_t)b * HW);
  const float* src = preds + (size_t)b * C * HW + hw;
  float x[CMAX], p[CMAX];
  float m = -INFINITY, sum = 0.0f;
#pragma unroll
  for (int c = 0; c < CMAX; ++c) {
    x[c] = c < C ? src[(size_t)c * HW] : (mode == 1 ? -INFINITY : 0.0f);
    m = fmaxf(m, x[c]);
  }
  if (mode == 1) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) { p[c] = expf(x[c] - m); sum += p[c]; }
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = c < C ? p[c] / sum : 0.0f;
  } else if (mode == 0) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) sum += x[c];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = c < C ? x[c] / (sum + eps) : 0.0f;
  } else {
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) { p[c] = fmaxf(x[c], 0.0f); sum += p[c]; }
    const float d = fmaxf(sum, eps);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) p[c] = c < C ? p[c] / d : 0.0f;
  }
  float best = -INFINITY;
  int arg = 0;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C && p[c] > best) { best = p[c]; arg = c; }
  float score;
  if (score_override) {
    score = score_override[pix];
  } else if (kind == 4) {
    score = 1.0f - best;
  } else if (mode == 0 && (kind == 2 || kind == 3)) {
    // Dirichlet mutual information (auroc.py:55-63): note alpha0 carries eps here, unlike the probabilities above
    const float a0 = sum + eps;
    const float dg0 = digamma_ge1_f(a0 + 1.0f);
    float h = 0.0f, eh = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float q = x[c] / a0;
        const float qc = fmaxf(q, eps);
        h -= qc * logf(qc);
        eh -= q * (digamma_ge1_f(x[c] + 1.0f) - dg0);
      }
    score = h - eh;
    if (kind == 3) score /= logf((float)C);
  } else {
    float h = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float qc = fmaxf(p[c], eps);
        h -= qc * logf(qc);
      }
    score = (kind == 1) ? h / logf((float)C) : h;       // every other kind falls through to the plain entropy (auroc.py:53)
  }
  const int64_t lab = labels[pix];
  const bool valid = !has_ignore || lab != ignore;
  score_out[pix] = score;
  flag_out[pix] = valid ? (uint8_t)(arg != lab ? 1 : 0) : (uint8_t)2;
}

// sort key: descending score == ascending key, any finite or infinite float (NaN sorts first)
__device__ __forceinline__ unsigned desc_key(float s) {
  const unsigned u = __float_as_uint(s);
  const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}

__global__ __launch_bounds__(kThreads) void auroc_keygen_kernel(const float* __restrict__ scores, const uint8_t* __restrict__ is_err, long long n,
                                                                unsigned* __restrict__ keys, unsigned* __restrict__ vals, unsigned* __restrict__ G) {
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    keys[i] = desc_key(scores[i]);
    vals[i] = (unsigned)i | (is_err[i] ? 0x80000000u : 0u);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) G[0] = 1u;          // the sort kernels skip "classes" whose count is 0
}

// after the sort: sum over negatives of the number of positives ranked before them (exact, 64-bit)
__global__ __launch_bounds__(kThreads) void auroc_sum_kernel(const unsigned* __restrict__ vals, long long n, const unsigned* __restrict__ blkfg,
                                                             unsigned long long* __restrict__ total) {
  constexpr int kWaves = kThreads / 64;
  __shared__ unsigned s_w[kWaves];
  __shared__ unsigned long long s_sum[kWaves];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, blk = blockIdx.x;
  // a wave owns kItems consecutive 64-element rows of the tile (sorted order == (wave, item, lane))
  const long long base = (long long)blk * kTile + (long long)wave * (kItems * 64);
  unsigned fg[kItems];
  unsigned cnt = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * 64 + lane;
    fg[it] = i < n ? (vals[i] >> 31) : 0u;
    cnt += (unsigned)__popcll(__ballot(fg[it]));
  }
  if (lane == 0) s_w[wave] = cnt;
  __syncthreads();
  unsigned before = blkfg[blk];
  for (int w = 0; w < wave; ++w) before += s_w[w];
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned long long acc = 0;
#pragma unroll
  for (int it = 0; it < kItems; ++it) {
    const long long i = base + it * 64 + lane;
    const unsigned long long m = __ballot(fg[it]);
    if (i < n && !fg[it]) acc += before + (unsigned)__popcll(m & lt);
    before += (unsigned)__popcll(m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) s_sum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < kWaves; ++w) t += s_sum[w];
    if (t) atomicAdd(total, t);
  }
}

__global__ void auroc_finalize_kernel(const unsigned long long* __restrict__ total, const unsigned* __restrict__ blkfg, const unsigned* __restrict__ vals,
                                      long long n, int nblk, double* __restrict__ out) {
  // positives = exclusive prefix of the last tile + its own count
  unsigned long long pos = blkfg[nblk - 1];
  for (long long i = (long long)(nblk - 1) * kTile; i < n; ++i) pos += vals[i] >> 31;
  const double P = (double)pos, N = (double)(n - (long long)pos);
  out[1] = P;
  out[2] = N;
  out[0] = (pos == 0 || (long long)pos == n) ? (double)NAN : (double)*total / (P * N);
}

__global__ __launch_bounds__(kThreads) void auroc_unsort_kernel(const unsigned* __restrict__ keys, const unsigned* __restrict__ vals, long long n,
                                                                float* __restrict__ scores, uint8_t* __restrict__ is_err) {
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
    const unsigned asc = ~keys[i];
    const unsigned m = (unsigned)((int)asc >> 31);                   // all ones for a non-negative score
    reinterpret_cast<unsigned*>(scores)[i] = asc ^ ~(m & 0x7FFFFFFFu);
    is_err[i] = (uint8_t)(vals[i] >> 31);
  }
}

size_t auroc_carve(char* base, long long n, Ws* ws, unsigned long long** total) {
  const int nblk = nblk_of(n);
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  char* g = take(sizeof(unsigned));
  char* t = take(sizeof(unsigned long long));
  char* k0 = take((size_t)n * 4); char* k1 = take((size_t)n * 4);
  char* v0 = take((size_t)n * 4); char* v1 = take((size_t)n * 4);
  char* h = take((size_t)256 * nblk * 4);
  char* bf = take((size_t)nblk * 4);
  if (ws) {
    ws->G = (unsigned*)g; ws->loss_c = nullptr;
    ws->keys[0] = (unsigned*)k0; ws->keys[1] = (unsigned*)k1; ws->vals[0] = (unsigned*)v0; ws->vals[1] = (unsigned*)v1;
    ws->hist = (unsigned*)h; ws->blkfg = (unsigned*)bf;
    *total = (unsigned long long*)t;
  }
  return off;
}

}  // namespace

extern "C" int slu_auroc_scores(const float* preds, const int64_t* labels, const float* score_override, int B, int C, int HW, int mode, int score_kind,
                                int has_ignore, int64_t ignore_index, float eps, float* scores, uint8_t* flags, slu_stream_t stream) {
  if (!preds || !labels || !scores || !flags || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (mode < 0 || mode > 2 || score_kind < 0 || score_kind > 4) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const size_t npix = (size_t)B * HW;
  const unsigned nb = (unsigned)((npix + 255) / 256);
  if (C <= 20)
    hipLaunchKernelGGL(auroc_score_kernel<20>, dim3(nb), dim3(256), 0, slu_stream(stream), preds, labels, score_override, B, C, HW, mode, score_kind,
                       has_ignore, ignore_index, eps, scores, flags);
  else
    hipLaunchKernelGGL(auroc_score_kernel<32>, dim3(nb), dim3(256), 0, slu_stream(stream), preds, labels, score_override, B, C, HW, mode, score_kind,
                       has_ignore, ignore_index, eps, scores, flags);
  SLU_CHECK_LAUNCH();
}

extern "C" size_t slu_auroc_workspace_bytes(long long n) {
  if (n <= 0) return 0;
  return auroc_carve(nullptr, n, nullptr, nullptr);
}

extern "C" int slu_auroc_compute(const float* scores, const uint8_t* is_error, long long n, void* workspace, size_t workspace_bytes, double* out3,
                                 float* sorted_scores, uint8_t* sorted_is_error, slu_stream_t stream) {
  if (!scores || !is_error || !workspace || !out3 || n <= 0) return SLU_EINVAL;
  if (n >= (1ll << 31)) return SLU_EUNSUPPORTED;
  if ((sorted_scores == nullptr) != (sorted_is_error == nullptr)) return SLU_EINVAL;
  Ws ws;
  unsigned long long* total = nullptr;
  if (auroc_carve((char*)workspace, n, &ws, &total) > workspace_bytes) return SLU_EINVAL;
  if (reinterpret_cast<uintptr_t>(workspace) & 255) return SLU_EINVAL;
  hipStream_t st = slu_stream(stream);
  const int nblk = nblk_of(n);
  if (hipMemsetAsync(ws.G, 0, align256(sizeof(unsigned)) + sizeof(unsigned long long), st) != hipSuccess) return SLU_ELAUNCH;
  const unsigned kg = (unsigned)((n + kThreads - 1) / kThreads > 1024 ? 1024 : (n + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(auroc_keygen_kernel, dim3(kg), dim3(kThreads), 0, st, scores, is_error, n, ws.keys[0], ws.vals[0], ws.G);
  int cur = 0;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    hipLaunchKernelGGL(hist_kernel, dim3(nblk, 1), dim3(kThreads), 0, st, ws.keys[cur], n, nblk, shift, ws.G, ws.hist);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), 0, st, ws.hist, nblk, ws.G);
    hipLaunchKernelGGL(scatter_kernel, dim3(nblk, 1), dim3(kThreads), 0, st, ws.keys[cur], ws.vals[cur], ws.keys[cur ^ 1], ws.vals[cur ^ 1], n, nblk,
                       shift, ws.G, ws.hist);
    cur ^= 1;
  }
  hipLaunchKernelGGL(fgcount_kernel, dim3(nblk, 1), dim3(kThreads), 0, st, ws.vals[cur], n, nblk, ws.G, ws.blkfg);
  hipLaunchKernelGGL(fgscan_kernel, dim3(1), dim3(256), 0, st, ws.blkfg, nblk, ws.G);
  hipLaunchKernelGGL(auroc_sum_kernel, dim3(nblk), dim3(kThreads), 0, st, ws.vals[cur], n, ws.blkfg, total);
  hipLaunchKernelGGL(auroc_finalize_kernel, dim3(1), dim3(1), 0, st, total, ws.blkfg, ws.vals[cur], n, nblk, out3);
  if (sorted_scores) {
    // the sorted sample list (descending score) for ROC curves: undo the key transform, keep the error flag
    hipLaunchKernelGGL(auroc_unsort_kernel, dim3(kg), dim3(kThreads), 0, st, ws.keys[cur], ws.vals[cur], n, sorted_scores, sorted_is_error);
  }
  SLU_CHECK_LAUNCH();
}
