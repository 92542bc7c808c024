// On-device metric accumulators: confusion matrix (IoU) and top-label calibration bins (ECE),
// plus the softmax + NLL reduction of the SalsaNext loss.  Integer counts are exact (LDS integer
// histograms -> one 64-bit global atomic per non-empty bin per workgroup).
#include "slu_common.h"

namespace {

// models/evaluator.py:45-53 : idx = t*C + p ; bincount
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ preds, const int64_t* __restrict__ targets,
                                                        int64_t n, int C, unsigned long long* __restrict__ cm) {
  __shared__ unsigned int hist[32 * 32];
  const int cc = C * C;
  for (int i = threadIdx.x; i < cc; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = targets[e], p = preds[e];
    if (t >= 0 && t < C && p >= 0 && p < C) atomicAdd(&hist[(int)t * C + (int)p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < cc; i += blockDim.x)
    if (hist[i]) atomicAdd(&cm[i], (unsigned long long)hist[i]);
}

// metrics/ece.py:55-84 ('probs' mode) + :136-140 (np.histogram on float32 linspace edges)
template <int CMAX>
__global__ __launch_bounds__(256) void ece_kernel(const float* __restrict__ probs, const int64_t* __restrict__ labels, int B, int C,
                                                  int HW, int64_t ignore_index, int n_bins,
                                                  unsigned long long* __restrict__ count, double* __restrict__ sum_correct,
                                                  double* __restrict__ sum_conf) {
  __shared__ unsigned int s_n[64], s_ok[64];
  __shared__ float s_conf[64];
  __shared__ float s_edge[65];
  for (int i = threadIdx.x; i < 64; i += blockDim.x) { s_n[i] = 0; s_ok[i] = 0; s_conf[i] = 0.0f; }
  for (int i = threadIdx.x; i <= n_bins; i += blockDim.x)
    s_edge[i] = (i == n_bins) ? 1.0f : (float)((double)i * (1.0 / (double)n_bins));  // np.linspace(0,1,n+1,float32)
  __syncthreads();
  const size_t npix = (size_t)B * HW;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int64_t lab = labels[pix];
    if (lab == ignore_index) continue;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)b * HW);
    const float* src = probs + (size_t)b * C * (size_t)HW + hw;
    float s = 0.0f, best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float p = fmaxf(src[(size_t)c * HW], 0.0f);
        s += p;
        if (p > best) { best = p; arg = c; }
      }
    float conf = best / fmaxf(s, 1e-12f);
    conf = fminf(fmaxf(conf, 0.0f), 1.0f);
    int bin = (int)(conf * (float)n_bins);
    bin = bin < 0 ? 0 : (bin > n_bins - 1 ? n_bins - 1 : bin);
    while (bin > 0 && conf < s_edge[bin]) --bin;
    while (bin < n_bins - 1 && conf >= s_edge[bin + 1]) ++bin;
    atomicAdd(&s_n[bin], 1u);
    if ((int64_t)arg == lab) atomicAdd(&s_ok[bin], 1u);
    atomicAdd(&s_conf[bin], conf);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_bins; i += blockDim.x)
    if (s_n[i]) {
      atomicAdd(&count[i], (unsigned long long)s_n[i]);
      atomicAdd(&sum_correct[i], (double)s_ok[i]);
      atomicAdd(&sum_conf[i], (double)s_conf[i]);
    }
}

// models/trainer.py:511-514 : probs = softmax(logits); -log(max(probs[y], clamp)) summed over pixels
template <int CMAX>
__global__ __launch_bounds__(256) void softmax_nll_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                          int B, int C, int HW, float clampv, float* __restrict__ probs,
                                                          double* __restrict__ nll_sum) {
  __shared__ double s_part[4];
  const size_t npix = (size_t)B * HW;
  double local = 0.0;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)b * HW);
    const float* src = logits + (size_t)b * C * (size_t)HW + hw;
    float x[CMAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      x[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
      m = fmaxf(m, x[c]);
    }
    float se = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) { x[c] = expf(x[c] - m); se += x[c]; }
    const int64_t lab = labels[pix];
    float py = 1.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float p = x[c] / se;
        if (probs) probs[((size_t)b * C + c) * HW + hw] = p;
        if ((int64_t)c == lab) py = p;
      }
    if (lab >= 0 && lab < C) local += (double)(-logf(fmaxf(py, clampv)));
  }
  local = wave_sum(local);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_part[w];
    atomicAdd(nll_sum, tot);
  }
}

inline unsigned grid_for(size_t n, unsigned cap) {
  const size_t nb = (n + 255) / 256;
  return (unsigned)(nb > cap ? cap : (nb ? nb : 1));
}


// models/evaluator.py:659-673 (UncertaintyAccuracyAggregator.update): u = clamp(uncertainty, 0, 1), flag = label == pred, pixels whose
// label is in ignore_ids dropped.  flag: 0 wrong, 1 correct, 2 ignored.
__global__ __launch_bounds__(256) void ua_samples_kernel(const int64_t* __restrict__ labels, const int64_t* __restrict__ preds,
                                                         const float* __restrict__ unc, size_t n, const int64_t* __restrict__ ignore_ids,
                                                         int n_ignore, float* __restrict__ u_out, uint8_t* __restrict__ flag_out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t lab = labels[i];
    bool ign = false;
    for (int k = 0; k < n_ignore; ++k) ign |= lab == ignore_ids[k];
    u_out[i] = fminf(fmaxf(unc[i], 0.0f), 1.0f);
    flag_out[i] = ign ? (uint8_t)2 : (uint8_t)(lab == preds[i] ? 1 : 0);
  }
}

// models/evaluator.py:733-739: np.histogram(u, bins=edges) and the same with weights = correct, for arbitrary strictly increasing
// float32 edges: bin i = [e_i, e_{i+1}), the last one closed, values outside [e_0, e_K] dropped.  K <= 256.
__global__ __launch_bounds__(256) void binned_counts_kernel(const float* __restrict__ u, const uint8_t* __restrict__ correct, size_t n,
                                                            const float* __restrict__ edges, int K, unsigned long long* __restrict__ count,
                                                            unsigned long long* __restrict__ n_correct) {
  __shared__ unsigned s_n[256], s_ok[256];
  __shared__ float s_edge[257];
  s_n[threadIdx.x] = 0;
  s_ok[threadIdx.x] = 0;
  for (int i = threadIdx.x; i <= K; i += blockDim.x) s_edge[i] = edges[i];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    if (!(v >= s_edge[0] && v <= s_edge[K])) continue;
    int lo = 0, hi = K;                     // largest b with e_b <= v (b < K), the right edge folded into the last bin
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (v >= s_edge[mid]) lo = mid; else hi = mid;
    }
    atomicAdd(&s_n[lo], 1u);
    if (correct[i]) atomicAdd(&s_ok[lo], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < K && s_n[threadIdx.x]) {
    atomicAdd(&count[threadIdx.x], (unsigned long long)s_n[threadIdx.x]);
    atomicAdd(&n_correct[threadIdx.x], (unsigned long long)s_ok[threadIdx.x]);
  }
}

// metrics/ece.py:55-84 (ECEAggregator._to_probs + update): per pixel the top-label confidence of p (by mode: 0 alpha a / (sum a + eps),
// 1 logits softmax, 2 probs clamp >= 0 then / max(sum, eps)), clamped to [0, 1], and flag = 1 prediction == label / 0 otherwise /
// 2 label == ignore_index.  Feeds the capped sample buffers of the reservoir mode (:93-111).
template <int CMAX>
__global__ __launch_bounds__(256) void ece_samples_kernel(const float* __restrict__ preds, const int64_t* __restrict__ labels, int B, int C, int HW,
                                                          int mode, int has_ignore, int64_t ignore_index, float eps, float* __restrict__ conf_out,
                                                          uint8_t* __restrict__ flag_out) {
  const size_t npix = (size_t)B * HW;
  for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)b * HW);
    const float* src = preds + (size_t)b * C * (size_t)HW + hw;
    float x[CMAX];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
      x[c] = (c < C) ? src[(size_t)c * HW] : -INFINITY;
      if (mode == 2 && c < C) x[c] = fmaxf(x[c], 0.0f);
      m = fmaxf(m, x[c]);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        if (mode == 1) x[c] = expf(x[c] - m);
        s += x[c];
      }
    const float den = mode == 0 ? s + eps : (mode == 1 ? s : fmaxf(s, eps));
    float best = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < C) {
        const float p = x[c] / den;
        if (p > best) { best = p; arg = c; }
      }
    const int64_t lab = labels[pix];
    conf_out[pix] = fminf(fmaxf(best, 0.0f), 1.0f);
    flag_out[pix] = (has_ignore && lab == ignore_index) ? (uint8_t)2 : (uint8_t)((int64_t)arg == lab ? 1 : 0);
  }
}

// np.histogram(u, bins=edges) with weights None / correct / u (metrics/ece.py:136-140): binned_counts_kernel plus the per-bin sum of u.
__global__ __launch_bounds__(256) void binned_stats_kernel(const float* __restrict__ u, const uint8_t* __restrict__ correct, size_t n,
                                                           const float* __restrict__ edges, int K, unsigned long long* __restrict__ count,
                                                           unsigned long long* __restrict__ n_correct, double* __restrict__ sum_u) {
  __shared__ unsigned s_n[256], s_ok[256];
  __shared__ float s_u[256];
  __shared__ float s_edge[257];
  s_n[threadIdx.x] = 0;
  s_ok[threadIdx.x] = 0;
  s_u[threadIdx.x] = 0.0f;
  for (int i = threadIdx.x; i <= K; i += blockDim.x) s_edge[i] = edges[i];
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = u[i];
    if (!(v >= s_edge[0] && v <= s_edge[K])) continue;
    int lo = 0, hi = K;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (v >= s_edge[mid]) lo = mid; else hi = mid;
    }
    atomicAdd(&s_n[lo], 1u);
    if (correct[i]) atomicAdd(&s_ok[lo], 1u);
    atomicAdd(&s_u[lo], v);
  }
  __syncthreads();
  if ((int)threadIdx.x < K && s_n[threadIdx.x]) {
    atomicAdd(&count[threadIdx.x], (unsigned long long)s_n[threadIdx.x]);
    atomicAdd(&n_correct[threadIdx.x], (unsigned long long)s_ok[threadIdx.x]);
    atomicAdd(&sum_u[threadIdx.x], (double)s_u[threadIdx.x]);
  }
}

}  // namespace

extern "C" int slu_confusion_update(const int64_t* preds, const int64_t* targets, int64_t n, int C, int64_t* confmat,
                                    slu_stream_t stream) {
  if (!preds || !targets || !confmat || n < 0 || C <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  if (n == 0) return SLU_OK;
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for((size_t)n, 1024)), dim3(256), 0, slu_stream(stream), preds, targets, n, C,
                     reinterpret_cast<unsigned long long*>(confmat));
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_ece_update(const float* probs, const int64_t* labels, int B, int C, int HW, int64_t ignore_index, int n_bins,
                              int64_t* count, double* sum_correct, double* sum_conf, slu_stream_t stream) {
  if (!probs || !labels || !count || !sum_correct || !sum_conf || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32 || n_bins < 2 || n_bins > 64) return SLU_EUNSUPPORTED;
  const unsigned g = grid_for((size_t)B * HW, 1024);
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(count);
  if (C <= 20)
    hipLaunchKernelGGL(ece_kernel<20>, dim3(g), dim3(256), 0, slu_stream(stream), probs, labels, B, C, HW, ignore_index, n_bins, cnt,
                       sum_correct, sum_conf);
  else
    hipLaunchKernelGGL(ece_kernel<32>, dim3(g), dim3(256), 0, slu_stream(stream), probs, labels, B, C, HW, ignore_index, n_bins, cnt,
                       sum_correct, sum_conf);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_softmax_nll_fwd(const float* logits, const int64_t* labels, int B, int C, int HW, float clampv, float* probs,
                                   double* nll_sum, slu_stream_t stream) {
  if (!logits || !labels || !nll_sum || B <= 0 || C <= 0 || HW <= 0) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const unsigned g = grid_for((size_t)B * HW, 2048);
  if (C <= 20)
    hipLaunchKernelGGL(softmax_nll_kernel<20>, dim3(g), dim3(256), 0, slu_stream(stream), logits, labels, B, C, HW, clampv, probs,
                       nll_sum);
  else
    hipLaunchKernelGGL(softmax_nll_kernel<32>, dim3(g), dim3(256), 0, slu_stream(stream), logits, labels, B, C, HW, clampv, probs,
                       nll_sum);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_ua_samples(const int64_t* labels, const int64_t* preds, const float* uncertainty, long long n, const int64_t* ignore_ids,
                              int n_ignore, float* u_out, uint8_t* flags, slu_stream_t stream) {
  if (!labels || !preds || !uncertainty || !u_out || !flags || n <= 0 || n_ignore < 0 || (n_ignore > 0 && !ignore_ids)) return SLU_EINVAL;
  hipLaunchKernelGGL(ua_samples_kernel, dim3(grid_for((size_t)n, 4096)), dim3(256), 0, slu_stream(stream), labels, preds, uncertainty, (size_t)n,
                     ignore_ids, n_ignore, u_out, flags);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_binned_counts(const float* u, const uint8_t* correct, long long n, const float* edges, int n_bins, int64_t* count,
                                 int64_t* n_correct, slu_stream_t stream) {
  if (!u || !correct || !edges || !count || !n_correct || n <= 0 || n_bins <= 0) return SLU_EINVAL;
  if (n_bins > 256) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL(binned_counts_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, slu_stream(stream), u, correct, (size_t)n, edges, n_bins,
                     reinterpret_cast<unsigned long long*>(count), reinterpret_cast<unsigned long long*>(n_correct));
  SLU_CHECK_LAUNCH();
}


extern "C" int slu_ece_samples(const float* preds, const int64_t* labels, int B, int C, int HW, int mode, int has_ignore, int64_t ignore_index,
                               float eps, float* conf, uint8_t* flags, slu_stream_t stream) {
  if (!preds || !labels || !conf || !flags || B <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 2) return SLU_EINVAL;
  if (C > 32) return SLU_EUNSUPPORTED;
  const unsigned g = grid_for((size_t)B * HW, 4096);
  if (C <= 20)
    hipLaunchKernelGGL(ece_samples_kernel<20>, dim3(g), dim3(256), 0, slu_stream(stream), preds, labels, B, C, HW, mode, has_ignore, ignore_index,
                       eps, conf, flags);
  else
    hipLaunchKernelGGL(ece_samples_kernel<32>, dim3(g), dim3(256), 0, slu_stream(stream), preds, labels, B, C, HW, mode, has_ignore, ignore_index,
                       eps, conf, flags);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_binned_stats(const float* u, const uint8_t* correct, long long n, const float* edges, int n_bins, int64_t* count,
                                int64_t* n_correct, double* sum_u, slu_stream_t stream) {
  if (!u || !correct || !edges || !count || !n_correct || !sum_u || n <= 0 || n_bins <= 0) return SLU_EINVAL;
  if (n_bins > 256) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL(binned_stats_kernel, dim3(grid_for((size_t)n, 2048)), dim3(256), 0, slu_stream(stream), u, correct, (size_t)n, edges, n_bins,
                     reinterpret_cast<unsigned long long*>(count), reinterpret_cast<unsigned long long*>(n_correct), sum_u);
  SLU_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// Samples grouped by class, stable (models/evaluator.py:211-232 UncertaintyPerClassAggregator.update: the
// reference pulls `uncertainty[labels == c]` for every class from host copies).  A wave owns a run of kRun consecutive
// samples; lane c of the wave keeps the count / write cursor of class c, ranks inside a 64-sample step come from ballots.
//   1. group_count_kernel:  counts[unit][c]
//   2. group_scan_kernel:   counts -> exclusive offsets, class-major (class c's segment starts after all samples of classes < c)
//   3. group_scatter_kernel: out[offset(unit, c) + rank] = value
// ---------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kRun = 1024;      // samples per wave

__global__ __launch_bounds__(256) void group_count_kernel(const int64_t* __restrict__ labels, long long n, int C, long long nunit,
                                                          unsigned* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= nunit) return;
  unsigned mine = 0;
  for (int it = 0; it < kRun / 64; ++it) {
    const long long i = unit * kRun + it * 64 + lane;
    const int64_t y = i < n ? labels[i] : -1;
    for (int c = 0; c < C; ++c) {
      const unsigned k = (unsigned)__popcll(__ballot(y == c));
      if (lane == c) mine += k;
    }
  }
  if (lane < C) counts[unit * 32 + lane] = mine;
}

// one workgroup; thread t walks units t, t + 1024, ...; per class a block-wide exclusive scan with a running carry
__global__ __launch_bounds__(1024) void group_scan_kernel(unsigned* __restrict__ counts, long long nunit, int C, long long* __restrict__ totals) {
  __shared__ unsigned long long s_w[16];
  __shared__ unsigned long long s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int c = 0; c < C; ++c) {
    const unsigned long long class_begin = s_carry;
    for (long long base = 0; base < nunit; base += 1024) {
      const long long u = base + tid;
      const unsigned v = u < nunit ? counts[u * 32 + c] : 0u;
      unsigned long long x = v;                                   // inclusive scan inside the wave
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long y = __shfl_up(x, d);
        if (lane >= d) x += y;
      }
      if (lane == 63) s_w[wave] = x;
      __syncthreads();
      unsigned long long before = s_carry;
      for (int w = 0; w < wave; ++w) before += s_w[w];
      if (u < nunit) counts[u * 32 + c] = (unsigned)(before + x - v);      // exclusive offset (n < 2^32 checked by the host wrapper)
      __syncthreads();
      if (tid == 1023) s_carry = before + x;
      __syncthreads();
    }
    if (tid == 0) totals[c] = (long long)(s_carry - class_begin);
  }
}

__global__ __launch_bounds__(256) void group_scatter_kernel(const int64_t* __restrict__ labels, const float* __restrict__ values, long long n, int C,
                                                            long long nunit, const unsigned* __restrict__ offsets, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long unit = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= nunit) return;
  unsigned cursor = lane < C ? offsets[unit * 32 + lane] : 0u;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int it = 0; it < kRun / 64; ++it) {
    const long long i = unit * kRun + it * 64 + lane;
    const int64_t y = i < n ? labels[i] : -1;
    unsigned dst = 0;
    for (int c = 0; c < C; ++c) {
      const unsigned long long m = __ballot(y == c);
      const unsigned start = __shfl(cursor, c);
      if (y == c) dst = start + (unsigned)__popcll(m & lt);
      if (lane == c) cursor += (unsigned)__popcll(m);
    }
    if (y >= 0 && y < C) out[dst] = values[i];
  }
}

}  // namespace

extern "C" size_t slu_group_by_class_workspace_bytes(long long n) {
  const long long nunit = (n + kRun - 1) / kRun;
  return (size_t)(nunit > 0 ? nunit : 1) * 32 * sizeof(unsigned);
}

extern "C" int slu_group_by_class(const int64_t* labels, const float* values, long long n, int C, float* out_values, int64_t* counts, void* workspace,
                                  size_t workspace_bytes, slu_stream_t stream) {
  if (!labels || !values || !out_values || !counts || !workspace || n <= 0 || C <= 0) return SLU_EINVAL;
  if (C > 32 || n >= (1ll << 32)) return SLU_EUNSUPPORTED;
  if (workspace_bytes < slu_group_by_class_workspace_bytes(n)) return SLU_EINVAL;
  const long long nunit = (n + kRun - 1) / kRun;
  unsigned* cnt = reinterpret_cast<unsigned*>(workspace);
  hipStream_t st = slu_stream(stream);
  const unsigned nb = (unsigned)((nunit + 3) / 4);
  hipLaunchKernelGGL(group_count_kernel, dim3(nb), dim3(256), 0, st, labels, n, C, nunit, cnt);
  hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(1024), 0, st, cnt, nunit, C, reinterpret_cast<long long*>(counts));
  hipLaunchKernelGGL(group_scatter_kernel, dim3(nb), dim3(256), 0, st, labels, values, n, C, nunit, cnt, out_values);
  SLU_CHECK_LAUNCH();
}
