// Convolution weight gradient on the fp32 matrix cores (gfx950).
//
//   dW[co][ci][ti][tj] = sum_{n,y,x} da[n,co,y,x] * in[n,ci, y - pad + ti*dil, x - pad + tj*dil]
//
// GEMM view: M = output channels, N = input channels, K = pixels.  v_mfma_f32_32x32x2_f32 wants the M / N
// index on the lane and two K values per instruction, so both operands are read from channel-last copies
// (da_t [N,HW,Cop], in_t [N,HW,Cip], Cop/Cip = channel counts padded to 32; written by slu_nchw_to_nhwc /
// slu_gather_nhwc): lane l loads channel (l & 31) of pixel (x + (l >> 5)) -- two 128-byte segments per load,
// no LDS.  One wave accumulates a 32 x 32 x (KS*KS taps) block of dW over a strided set of pixel runs; the
// tap-shifted B operands of neighbouring K-steps overlap and are served by L1/L2.  Partial blocks are added
// to a tap-major image dWp[co][tap][ci] with float atomics (contiguous 128-byte segments per instruction;
// summation order is not reproducible bit for bit), then unpacked to OIHW.
#include "slu_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kRun = 32;   // pixel pairs per run (64 azimuth-adjacent pixels)

template <int KS, int DIL, int PAD>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const float* __restrict__ da_t, const float* __restrict__ in_t, int N, int H,
                                                       int W, int Cop, int Cip, float* __restrict__ dWp) {
  constexpr int T = KS * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  const int cob = blockIdx.y, cib = blockIdx.z;
  const long long HW = (long long)H * W;
  const long long npairs = (long long)N * HW / 2;
  const long long nruns = (npairs + kRun - 1) / kRun;
  const long long worker = (long long)blockIdx.x * 4 + wave, nworkers = (long long)gridDim.x * 4;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const float* abase = da_t + (size_t)cob * 32 + jj;
  const float* bbase = in_t + (size_t)cib * 32 + jj;

  // tap offsets in elements of in_t (wave-uniform); both operands are channel-last, so the flat pixel index is linear
  // across rows and images and every K-step is one pointer bump
  long long tapoff[T];
#pragma unroll
  for (int t = 0; t < T; ++t) tapoff[t] = ((long long)(-PAD + (t / KS) * DIL) * W + (-PAD + (t % KS) * DIL)) * Cip;

  for (long long run = worker; run < nruns; run += nworkers) {
    const long long q0 = run * kRun;
    const long long q1 = (q0 + kRun < npairs) ? q0 + kRun : npairs;
    // position of the first pixel of the run (wave-uniform; kept in scalar registers and advanced incrementally)
    const long long pix0 = 2 * q0;
    const long long n0 = pix0 / HW;
    const int rem0 = (int)(pix0 - n0 * HW);
    int y = rem0 / W, x = rem0 - y * W;
    const float* pa = abase + (size_t)(pix0 + hh) * Cop;
    const float* pb = bbase + (size_t)(pix0 + hh) * Cip;
    // four K-steps (pixel pairs) per iteration: all 4 * (1 + T) loads are issued before the first MFMA, so the
    // L2/HBM latency of one batch hides behind the 4 * T MFMAs of the previous one across the 2 waves per SIMD
    constexpr int U = (T >= 9) ? 2 : 4;           // 3x3: 144 accumulator registers leave room for two batches of operands
    for (long long q = q0; q < q1; q += U) {
      float a[U], b[U][T];
      int yu = y, xu = x;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = q + u < q1;
        a[u] = live ? pa[(size_t)u * 2 * Cop] : 0.0f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int yy = yu - PAD + (t / KS) * DIL, xx = xu + hh - PAD + (t % KS) * DIL;
          const bool ok = live && yy >= 0 && yy < H && xx >= 0 && xx < W;
          const float v = pb[ok ? (long long)u * 2 * Cip + tapoff[t] : 0];
          b[u][t] = ok ? v : 0.0f;
        }
        xu += 2;
        if (xu >= W) { xu = 0; ++yu; if (yu >= H) yu = 0; }      // W is even: a pair never straddles a row
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][t], acc[t], 0, 0, 0);
      pa += (size_t)U * 2 * Cop;
      pb += (size_t)U * 2 * Cip;
      y = yu;
      x = xu;
    }
  }

  // The four waves of the workgroup worked on the same (cob, cib) block: their partial sums meet in LDS tap by tap and ONE wave adds
  // each tap's total to dWp -- float atomics execute at the memory side at ~one 256-byte wave-instruction per 50 ns per CU
  // (MI355X_MICROARCH.md), so a wave-private flush of T x 16 of them cost more than the MFMAs of a short run.
  // D[i = co][j = ci]: lane & 31 = ci, register/half = co
  __shared__ float s_red[4][16][64];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[t][r];
    __syncthreads();
    if (wave == (t & 3)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane]);
        const int co = cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        atomicAdd(&dWp[((size_t)co * T + t) * Cip + cib * 32 + jj], v);
      }
    }
    __syncthreads();
  }
}

// 1x1 convs: a plain GEMM dW[co][ci] = sum_pix da[pix][co] * in[pix][ci].  One (32 x 32) block per wave needs two loads per MFMA and is
// latency-bound; here a wave owns MT x NT blocks (MT + NT loads per MT * NT MFMAs) and U K-steps of operands are in flight.
template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(const float* __restrict__ da_t, const float* __restrict__ in_t, long long npix, int Cop,
                                                          int Cip, float* __restrict__ dWp) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  const int cob0 = blockIdx.y * MT, cib0 = blockIdx.z * NT;
  const int ncob = Cop / 32, ncib = Cip / 32;
  const long long npairs = npix / 2;
  const long long nruns = (npairs + kRun - 1) / kRun;
  const long long worker = (long long)blockIdx.x * 4 + wave, nworkers = (long long)gridDim.x * 4;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // blocks past the last one read block 0 and are never stored
  const float* abase[MT];
  const float* bbase[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) abase[i] = da_t + (size_t)(cob0 + i < ncob ? cob0 + i : 0) * 32 + jj;
#pragma unroll
  for (int j = 0; j < NT; ++j) bbase[j] = in_t + (size_t)(cib0 + j < ncib ? cib0 + j : 0) * 32 + jj;

  for (long long run = worker; run < nruns; run += nworkers) {
    const long long q0 = run * kRun;
    const long long q1 = (q0 + kRun < npairs) ? q0 + kRun : npairs;
    for (long long q = q0; q < q1; q += U) {
      float a[U][MT], b[U][NT];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = q + u < q1;
        const size_t pix = (size_t)(2 * (live ? q + u : q0) + hh);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float v = abase[i][pix * Cop];
          a[u][i] = live ? v : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float v = bbase[j][pix * Cip];
          b[u][j] = live ? v : 0.0f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
  }

  __shared__ float s_red[4][16][64];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[i][j][r];
      __syncthreads();
      if (wave == ((i * NT + j) & 3) && cob0 + i < ncob && cib0 + j < ncib) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane]);
          const int co = (cob0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          atomicAdd(&dWp[(size_t)co * Cip + (cib0 + j) * 32 + jj], v);
        }
      }
      __syncthreads();
    }
}

// 1x1 convs straight from the NCHW tensors (no channel-last copies of da / the inputs): lane (channel i, half k) loads FOUR consecutive
// pixels of its channel with one 16-byte load -- pixels 8 u + 4 k .. + 3 of unit u -- and the four components feed four K-steps; the pixel
// order inside the sum is permuted, which a sum does not care about, as long as A and B use the same permutation.  Sources of a concatenated
// input are separate base pointers per 32-channel block (every concat of the network is 32-aligned).  MT x NT blocks per wave as above.
struct W1Args {
  const float* da;               // [N][Cout][HW]
  const float* src[SLU_MAX_SRC]; // [N][Cs][HW]
  int cs[SLU_MAX_SRC], cbeg[SLU_MAX_SRC];
  int nsrc, N, HW, Cout, Cin;
  float* dW;                     // [Cout][Cin], zeroed
};

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void wgrad1x1_nchw_kernel(const W1Args a) {
  // unit = 32 pixels: lane (channel i, half k) owns pixels 32 u + 16 k .. + 15 of its channel = 64 contiguous bytes (4 x dwordx4), so the two
  // halves of a channel cover one whole 128-byte line exactly once; 16 K-steps per unit, the next unit's loads in flight under them
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  // 1-D grid, the tile index fastest: the workgroups that read the same pixels (other channel tiles) are dispatched together, so the
  // re-reads of da / the inputs by the other tiles are cache hits, not HBM reads
  const int ncob = (a.Cout + 31) / 32, ncib = (a.Cin + 31) / 32;
  const int gy = (ncob + MT - 1) / MT, gz = (ncib + NT - 1) / NT;
  const int tile = blockIdx.x % (gy * gz), bx = blockIdx.x / (gy * gz);
  const int cob0 = (tile / gz) * MT, cib0 = (tile % gz) * NT;
  const long long upi = a.HW / 32;                         // units per image
  const long long nunits = (long long)a.N * upi;
  const long long worker = (long long)bx * 4 + wave, nworkers = (long long)(gridDim.x / (gy * gz)) * 4;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float* abase[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {   // lanes past the last channel read channel 0: their rows / columns of the product are never stored
    const int co = (cob0 + i) * 32 + jj;
    abase[i] = a.da + (size_t)((cob0 + i < ncob && co < a.Cout) ? co : 0) * a.HW + 16 * hh;
  }
  const float* bbase[NT];
  long long bimg[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ci = (cib0 + j) * 32 + jj;
    int s = 0;
#pragma unroll
    for (int t = 1; t < SLU_MAX_SRC; ++t)
      if (t < a.nsrc && ci >= a.cbeg[t]) s = t;
    const int cl = (cib0 + j < ncib && ci < a.Cin) ? ci - a.cbeg[s] : 0;
    bbase[j] = a.src[s] + (size_t)cl * a.HW + 16 * hh;
    bimg[j] = (long long)a.cs[s] * a.HW;
  }
  const long long aimg = (long long)a.Cout * a.HW;

  // register budget (256 at two waves per SIMD): 16 (MT + NT) floats per buffered unit; the 2 x 2 tile buffers HALF units (Q = 2 loads per
  // block) and alternates the two halves of one unit between the buffers, so a line is still consumed by one wave back to back
  constexpr int Q = (MT + NT > 3) ? 2 : 4;
  float4 av[2][MT][Q], bv[2][NT][Q];
  auto load = [&](int buf, long long u, int part) {
    const long long n = u / upi;
    const long long p = (u - n * upi) * 32 + part * (4 * Q);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float4* q = reinterpret_cast<const float4*>(abase[i] + n * aimg + p);
#pragma unroll
      for (int t = 0; t < Q; ++t) av[buf][i][t] = q[t];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4* q = reinterpret_cast<const float4*>(bbase[j] + n * bimg[j] + p);
#pragma unroll
      for (int t = 0; t < Q; ++t) bv[buf][j][t] = q[t];
    }
  };
  auto mac = [&](int buf) {
#pragma unroll
    for (int t = 0; t < Q; ++t)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][i][t].x, bv[buf][j][t].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][i][t].y, bv[buf][j][t].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][i][t].z, bv[buf][j][t].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][i][t].w, bv[buf][j][t].w, acc[i][j], 0, 0, 0);
        }
  };
  long long u = worker;
  if (u < nunits) load(0, u, 0);
  if constexpr (Q == 4) {
    while (u < nunits) {                                    // two units per trip so the buffer index is a compile-time constant
      const long long u1 = u + nworkers, u2 = u1 + nworkers;
      if (u1 < nunits) load(1, u1, 0);
      mac(0);
      if (u1 >= nunits) break;
      if (u2 < nunits) load(0, u2, 0);
      mac(1);
      u = u2;
    }
  } else {
    while (u < nunits) {
      const long long u1 = u + nworkers;
      load(1, u, 1);
      mac(0);
      if (u1 < nunits) load(0, u1, 0);
      mac(1);
      u = u1;
    }
  }

  __shared__ float s_red[4][16][64];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[i][j][r];
      __syncthreads();
      if (wave == ((i * NT + j) & 3)) {
        const int ci = (cib0 + j) * 32 + jj;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane]);
          const int co = (cob0 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (cob0 + i < ncob && cib0 + j < ncib && co < a.Cout && ci < a.Cin) atomicAdd(&a.dW[(size_t)co * a.Cin + ci], v);
        }
      }
      __syncthreads();
    }
}

// k x k convs straight from the NCHW tensors.  Lane (channel jj, half hh) owns 8 consecutive pixels of one row per unit of 16: the A operand
// is two 16-byte loads of da, the B operands of ALL taps come from one window of 8 + L + R input pixels per tap row (two 16-byte loads + the L
// pixels before and the R after), so a unit costs 2 + KS * 4 load instructions for 8 * KS * KS MFMAs (the channel-last form: 1 + KS * KS
// four-byte loads per KS * KS MFMAs), and neither da nor the inputs are copied to channel-last first.  The K pairing (pixel e of half 0 with
// pixel e of half 1) is the same on both operands.  Zero padding: rows outside the image contribute zeros (wave-uniform), the pixels before /
// after a row are zeroed at the row's ends (units never straddle rows: W % 16 == 0).  Sources of a concatenated input are per-32-channel-block
// base pointers (as in wgrad1x1_nchw_kernel).  Partial sums leave through the same LDS reduction + atomics into the tap-major image dWp.
struct WkArgs {
  const float* da;               // [N][Cout][H][W]
  const float* src[SLU_MAX_SRC]; // [N][Cs][H][W], or (PixelShuffle) [N][Cs][H/2][W/2] contributing Cs/4 channels
  const float* scale[SLU_MAX_SRC];   // [N][Cs] multiplier per stored channel, or nullptr
  int cs[SLU_MAX_SRC], cbeg[SLU_MAX_SRC], ps[SLU_MAX_SRC];
  int nsrc, N, H, W, Cout, Cin, Cip;
  float* dWp;                    // [Cout][T][Cip], zeroed
};

template <int KS, int DIL, int PAD>
__global__ __launch_bounds__(256, 2) void wgradk_nchw_kernel(const WkArgs a) {
  constexpr int T = KS * KS, L = PAD, R = (KS - 1) * DIL - PAD, WIN = 8 + L + R;
  static_assert(L >= 1 && L <= 2 && R >= 1 && R <= 2, "window edges are one or two pixels");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  const int cob = blockIdx.y, cib = blockIdx.z;
  const long long HW = (long long)a.H * a.W;
  const int upr = a.W / 16;                                  // units per row
  const long long nunits = (long long)a.N * a.H * upr;
  constexpr int RUN = 4;                                     // consecutive units per visit: 64 pixels of a row
  const long long nruns = (nunits + RUN - 1) / RUN;
  const long long worker = (long long)blockIdx.x * 4 + wave, nworkers = (long long)gridDim.x * 4;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // lanes past the last channel read channel 0: their rows / columns of the product are never stored
  const int co = cob * 32 + jj, ci = cib * 32 + jj;
  const float* abase = a.da + (size_t)(co < a.Cout ? co : 0) * HW + 8 * hh;
  int s = 0;                                                 // per lane: a 32-channel block may straddle two sources
#pragma unroll
  for (int t = 1; t < SLU_MAX_SRC; ++t)
    if (t < a.nsrc && ci >= a.cbeg[t]) s = t;
  const int cl = ci < a.Cin ? ci - a.cbeg[s] : 0;            // channel inside the source (of the shuffled tensor for a PixelShuffle source)
  const bool lps = a.ps[s] != 0;
  const float* sbase = a.src[s];
  const float* bbase = sbase + (size_t)cl * HW + 8 * hh;     // plain sources
  const float* scl = a.scale[s];
  const int csrc = a.cs[s];
  const long long aimg = (long long)a.Cout * HW, bimg = (long long)csrc * HW;
  const int W2 = a.W >> 1;
  const long long HW4 = HW >> 2, bimg4 = (long long)csrc * HW4;

  for (long long run = worker; run < nruns; run += nworkers) {
    const long long u0 = run * RUN;
    const long long u1 = (u0 + RUN < nunits) ? u0 + RUN : nunits;
    for (long long u = u0; u < u1; ++u) {
      const long long row = u / upr;                         // n * H + y
      const int xu = (int)(u - row * upr);
      const int n = (int)(row / a.H), y = (int)(row - (long long)n * a.H);
      const int x0 = xu * 16 + 8 * hh;
      const float* pa = abase + (size_t)n * aimg + (size_t)y * a.W + xu * 16;
      const float4 a0 = *reinterpret_cast<const float4*>(pa), a1 = *reinterpret_cast<const float4*>(pa + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      float win[KS][WIN];
#pragma unroll
      for (int ti = 0; ti < KS; ++ti) {
        const int yy = y - PAD + ti * DIL;
        if (yy >= 0 && yy < a.H) {                           // wave-uniform
          const bool lok = x0 > 0, rok = x0 + 8 < a.W;       // the pixels before / after this lane's 8 exist in the row
          float own[8], lft[2] = {0.0f, 0.0f}, rgt[2] = {0.0f, 0.0f};
          if (!lps) {
            const float* pb = bbase + (size_t)n * bimg + (size_t)yy * a.W + xu * 16;
            const float4 b0 = *reinterpret_cast<const float4*>(pb), b1 = *reinterpret_cast<const float4*>(pb + 4);
            const float sc = scl ? scl[(size_t)n * csrc + cl] : 1.0f;
            own[0] = b0.x * sc; own[1] = b0.y * sc; own[2] = b0.z * sc; own[3] = b0.w * sc;
            own[4] = b1.x * sc; own[5] = b1.y * sc; own[6] = b1.z * sc; own[7] = b1.w * sc;
            if constexpr (L == 2) {
              const float2 v = *reinterpret_cast<const float2*>(lok ? pb - 2 : pb);
              lft[0] = lok ? v.x * sc : 0.0f; lft[1] = lok ? v.y * sc : 0.0f;
            } else {
              const float v = *(lok ? pb - 1 : pb);
              lft[0] = lok ? v * sc : 0.0f;
            }
            if constexpr (R == 2) {
              const float2 v = *reinterpret_cast<const float2*>(rok ? pb + 8 : pb);
              rgt[0] = rok ? v.x * sc : 0.0f; rgt[1] = rok ? v.y * sc : 0.0f;
            } else {
              const float v = *(rok ? pb + 8 : pb);
              rgt[0] = rok ? v * sc : 0.0f;
            }
          } else {
            // PixelShuffle(2): in[c][yy][x] = stored[4 c + 2 (yy & 1) + (x & 1)][yy / 2][x / 2]: the 8 pixels interleave 4 half-resolution
            // pixels of two stored channels; the multiplier is per STORED channel
            const int c0 = 4 * cl + 2 * (yy & 1);
            const float* p0 = sbase + (size_t)n * bimg4 + (size_t)c0 * HW4 + (size_t)(yy >> 1) * W2 + (x0 >> 1);
            const float* p1 = p0 + HW4;
            const float4 e0 = *reinterpret_cast<const float4*>(p0), e1 = *reinterpret_cast<const float4*>(p1);
            const float s0 = scl ? scl[(size_t)n * csrc + c0] : 1.0f, s1 = scl ? scl[(size_t)n * csrc + c0 + 1] : 1.0f;
            own[0] = e0.x * s0; own[1] = e1.x * s1; own[2] = e0.y * s0; own[3] = e1.y * s1;
            own[4] = e0.z * s0; own[5] = e1.z * s1; own[6] = e0.w * s0; own[7] = e1.w * s1;
            const float l0 = *(lok ? p0 - 1 : p0), l1 = *(lok ? p1 - 1 : p1);       // pixels x0 - 2 (even) and x0 - 1 (odd)
            const float r0 = *(rok ? p0 + 4 : p0), r1 = *(rok ? p1 + 4 : p1);       // pixels x0 + 8 (even) and x0 + 9 (odd)
            if constexpr (L == 2) { lft[0] = lok ? l0 * s0 : 0.0f; lft[1] = lok ? l1 * s1 : 0.0f; }
            else lft[0] = lok ? l1 * s1 : 0.0f;
            rgt[0] = rok ? r0 * s0 : 0.0f;
            if constexpr (R == 2) rgt[1] = rok ? r1 * s1 : 0.0f;
          }
#pragma unroll
          for (int j = 0; j < L; ++j) win[ti][j] = lft[j];
#pragma unroll
          for (int j = 0; j < 8; ++j) win[ti][L + j] = own[j];
#pragma unroll
          for (int j = 0; j < R; ++j) win[ti][L + 8 + j] = rgt[j];
        } else {
#pragma unroll
          for (int j = 0; j < WIN; ++j) win[ti][j] = 0.0f;
        }
      }
      // output pixel e with tap (ti, tj) reads input pixel x0 + e - PAD + tj * DIL = window index e + tj * DIL
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int ti = 0; ti < KS; ++ti)
#pragma unroll
          for (int tj = 0; tj < KS; ++tj)
            acc[ti * KS + tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], win[ti][e + tj * DIL], acc[ti * KS + tj], 0, 0, 0);
    }
  }

  __shared__ float s_red[4][16][64];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_red[wave][r][lane] = acc[t][r];
    __syncthreads();
    if (wave == (t & 3)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (s_red[0][r][lane] + s_red[1][r][lane]) + (s_red[2][r][lane] + s_red[3][r][lane]);
        const int c = cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (c < a.Cout) atomicAdd(&a.dWp[((size_t)c * T + t) * a.Cip + cib * 32 + jj], v);
      }
    }
    __syncthreads();
  }
}

__global__ void wgrad_unpack_kernel(const float* __restrict__ dWp, int Cout, int Cin, int T, int Cip, float* __restrict__ dW, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(e % T);
    const size_t r = e / T;
    const int ci = (int)(r % Cin);
    const int co = (int)(r / Cin);
    dW[e] = dWp[((size_t)co * T + t) * Cip + ci];
  }
}

// weights of the data-gradient conv: Wd[ci][co][KS-1-ti][KS-1-tj] = W[co][ci][ti][tj]   (all four kernel
// families are symmetric: 2*pad == (KS-1)*dil, so dgrad is the same conv family with these weights)
__global__ void dgrad_weight_kernel(const float* __restrict__ w, int Cout, int Cin, int KS, float* __restrict__ wd, size_t total) {
  const int T = KS * KS;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(e % T);
    const size_t r = e / T;
    const int co = (int)(r % Cout);
    const int ci = (int)(r / Cout);
    wd[e] = w[((size_t)co * Cin + ci) * T + (T - 1 - t)];
  }
}

template <int KS, int DIL, int PAD>
int launch_wgrad(const float* da_t, const float* in_t, int N, int H, int W, int Cop, int Cip, float* dWp, hipStream_t st) {
  const int nb = (Cop / 32) * (Cip / 32);
  const long long nruns = ((long long)N * H * W / 2 + kRun - 1) / kRun;
  // two resident workgroups per CU over all channel-block pairs (2 waves per SIMD): every wave then walks many pixel runs before its
  // partial sums are flushed, instead of four queued waves of workgroups that each flush after a couple of runs
  long long gx = 512 / nb;
  if (gx < 1) gx = 1;
  if (gx * 4 > nruns) gx = (nruns + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((wgrad_kernel<KS, DIL, PAD>), dim3((unsigned)gx, Cop / 32, Cip / 32), dim3(256), 0, st, da_t, in_t, N, H, W, Cop, Cip, dWp);
  SLU_CHECK_LAUNCH();
}

template <int MT, int NT>
int launch_wgrad1x1_t(const float* da_t, const float* in_t, long long npix, int Cop, int Cip, float* dWp, hipStream_t st) {
  const int gy = (Cop / 32 + MT - 1) / MT, gz = (Cip / 32 + NT - 1) / NT;
  const long long nruns = (npix / 2 + kRun - 1) / kRun;
  long long gx = 1024 / ((long long)gy * gz);            // ~4 workgroups per CU: these waves are light on registers and live on occupancy
  if (gx < 1) gx = 1;
  if (gx * 4 > nruns) gx = (nruns + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((wgrad1x1_kernel<MT, NT>), dim3((unsigned)gx, gy, gz), dim3(256), 0, st, da_t, in_t, npix, Cop, Cip, dWp);
  SLU_CHECK_LAUNCH();
}

int launch_wgrad1x1(const float* da_t, const float* in_t, long long npix, int Cop, int Cip, float* dWp, hipStream_t st) {
  const int ncob = Cop / 32, ncib = Cip / 32;
  if (ncob >= 2 && ncib >= 2) return launch_wgrad1x1_t<2, 2>(da_t, in_t, npix, Cop, Cip, dWp, st);
  if (ncob >= 2) return launch_wgrad1x1_t<2, 1>(da_t, in_t, npix, Cop, Cip, dWp, st);
  if (ncib >= 2) return launch_wgrad1x1_t<1, 2>(da_t, in_t, npix, Cop, Cip, dWp, st);
  return launch_wgrad1x1_t<1, 1>(da_t, in_t, npix, Cop, Cip, dWp, st);
}

template <int MT, int NT>
int launch_w1_nchw_t(const W1Args& a, hipStream_t st) {
  const int ncob = (a.Cout + 31) / 32, ncib = (a.Cin + 31) / 32;
  const int gy = (ncob + MT - 1) / MT, gz = (ncib + NT - 1) / NT;
  const long long nunits = (long long)a.N * (a.HW / 32);
  const long long tiles = (long long)gy * gz;
  long long gx = nunits / 32;                              // ~8 units per wave: the epilogue (LDS reduction + atomics) is paid per workgroup
  const long long hi = 1024 / tiles > 1 ? 1024 / tiles : 1, lo = (256 + tiles - 1) / tiles;
  if (gx > hi) gx = hi;
  if (gx < lo) gx = lo;
  if (gx * 4 > nunits) gx = (nunits + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((wgrad1x1_nchw_kernel<MT, NT>), dim3((unsigned)(gx * tiles)), dim3(256), 0, st, a);
  SLU_CHECK_LAUNCH();
}

template <int KS, int DIL, int PAD>
int launch_wk_nchw(const WkArgs& a, hipStream_t st) {
  const int ncob = (a.Cout + 31) / 32, ncib = a.Cip / 32;
  const long long nruns = ((long long)a.N * a.H * (a.W / 16) + 3) / 4;
  long long gx = 512 / ((long long)ncob * ncib);            // two resident workgroups per CU over all channel-block pairs
  if (gx < 1) gx = 1;
  if (gx * 4 > nruns) gx = (nruns + 3) / 4;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL((wgradk_nchw_kernel<KS, DIL, PAD>), dim3((unsigned)gx, ncob, ncib), dim3(256), 0, st, a);
  SLU_CHECK_LAUNCH();
}

}  // namespace

extern "C" int slu_conv2d_wgrad_nchw(const float* da, const slu_conv_src* src, int nsrc, int N, int H, int W, int Cout, int ksize, int dil, int pad,
                                     float* dWp, float* dW, int prezeroed, slu_stream_t stream) {
  if (!da || !src || !dWp || !dW || nsrc < 1 || nsrc > SLU_MAX_SRC || N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return SLU_EINVAL;
  if (W % 16 || ((uintptr_t)da & 15) || Cout > 65535 * 32) return SLU_EUNSUPPORTED;
  WkArgs a{};
  int c = 0;
  for (int s = 0; s < nsrc; ++s) {
    if (!src[s].ptr || src[s].C <= 0) return SLU_EINVAL;
    if (src[s].nbatch || src[s].cuse || ((uintptr_t)src[s].ptr & 15)) return SLU_EUNSUPPORTED;
    if (src[s].pixel_shuffle && ((src[s].C & 3) || (H & 1))) return SLU_EINVAL;
    a.src[s] = src[s].ptr; a.scale[s] = src[s].scale; a.cs[s] = src[s].C; a.cbeg[s] = c; a.ps[s] = src[s].pixel_shuffle ? 1 : 0;
    c += src[s].pixel_shuffle ? src[s].C / 4 : src[s].C;
  }
  if (c > 65535 * 32) return SLU_EUNSUPPORTED;
  a.da = da; a.nsrc = nsrc; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.Cin = c; a.Cip = (c + 31) / 32 * 32; a.dWp = dWp;
  hipStream_t st = slu_stream(stream);
  if (!prezeroed && hipMemsetAsync(dWp, 0, slu_wgrad_packed_floats(Cout, c, ksize) * sizeof(float), st) != hipSuccess) return SLU_ELAUNCH;
  int rc;
  if (ksize == 3 && dil == 1 && pad == 1) rc = launch_wk_nchw<3, 1, 1>(a, st);
  else if (ksize == 3 && dil == 2 && pad == 2) rc = launch_wk_nchw<3, 2, 2>(a, st);
  else if (ksize == 2 && dil == 2 && pad == 1) rc = launch_wk_nchw<2, 2, 1>(a, st);
  else return SLU_EUNSUPPORTED;
  if (rc != SLU_OK) return rc;
  const size_t total = (size_t)Cout * c * ksize * ksize;
  const unsigned g = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(g), dim3(256), 0, st, dWp, Cout, c, ksize * ksize, a.Cip, dW, total);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_conv1x1_wgrad_nchw(const float* da, const slu_conv_src* src, int nsrc, int N, int HW, int Cout, float* dW, int prezeroed,
                                      slu_stream_t stream) {
  if (!da || !src || !dW || nsrc < 1 || nsrc > SLU_MAX_SRC || N <= 0 || HW <= 0 || Cout <= 0) return SLU_EINVAL;
  if (HW % 32 || ((uintptr_t)da & 15)) return SLU_EUNSUPPORTED;
  W1Args a{};
  int c = 0;
  for (int s = 0; s < nsrc; ++s) {
    if (!src[s].ptr || src[s].C <= 0) return SLU_EINVAL;
    if (src[s].pixel_shuffle || src[s].scale || src[s].nbatch || src[s].cuse || ((uintptr_t)src[s].ptr & 15)) return SLU_EUNSUPPORTED;
    if (s + 1 < nsrc && (src[s].C % 32)) return SLU_EUNSUPPORTED;          // a 32-channel block must not straddle two tensors
    a.src[s] = src[s].ptr; a.cs[s] = src[s].C; a.cbeg[s] = c;
    c += src[s].C;
  }
  a.da = da; a.nsrc = nsrc; a.N = N; a.HW = HW; a.Cout = Cout; a.Cin = c; a.dW = dW;
  hipStream_t st = slu_stream(stream);
  if (!prezeroed && hipMemsetAsync(dW, 0, (size_t)Cout * c * sizeof(float), st) != hipSuccess) return SLU_ELAUNCH;
  const int ncob = (Cout + 31) / 32, ncib = (c + 31) / 32;
  if (ncob >= 2 && ncib >= 2) return launch_w1_nchw_t<2, 2>(a, st);
  if (ncob >= 2) return launch_w1_nchw_t<2, 1>(a, st);
  if (ncib >= 2) return launch_w1_nchw_t<1, 2>(a, st);
  return launch_w1_nchw_t<1, 1>(a, st);
}

extern "C" size_t slu_wgrad_packed_floats(int cout, int cin, int ksize) {
  if (cout <= 0 || cin <= 0 || ksize <= 0) return 0;
  return (size_t)((cout + 31) / 32 * 32) * ksize * ksize * ((cin + 31) / 32 * 32);
}

extern "C" int slu_conv2d_wgrad(const float* da_t, const float* in_t, int N, int H, int W, int Cout, int Cin, int ksize, int dil, int pad,
                                float* dWp, float* dW, slu_stream_t stream) {
  if (!da_t || !in_t || !dWp || !dW || N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0) return SLU_EINVAL;
  if (W & 1) return SLU_EUNSUPPORTED;
  if (Cout > 65535 * 32 || Cin > 65535 * 32) return SLU_EUNSUPPORTED;
  const int Cop = (Cout + 31) / 32 * 32, Cip = (Cin + 31) / 32 * 32;
  hipStream_t st = slu_stream(stream);
  if (hipMemsetAsync(dWp, 0, slu_wgrad_packed_floats(Cout, Cin, ksize) * sizeof(float), st) != hipSuccess) return SLU_ELAUNCH;
  int rc;
  if (ksize == 1 && dil == 1 && pad == 0) rc = launch_wgrad1x1(da_t, in_t, (long long)N * H * W, Cop, Cip, dWp, st);
  else if (ksize == 3 && dil == 1 && pad == 1) rc = launch_wgrad<3, 1, 1>(da_t, in_t, N, H, W, Cop, Cip, dWp, st);
  else if (ksize == 3 && dil == 2 && pad == 2) rc = launch_wgrad<3, 2, 2>(da_t, in_t, N, H, W, Cop, Cip, dWp, st);
  else if (ksize == 2 && dil == 2 && pad == 1) rc = launch_wgrad<2, 2, 1>(da_t, in_t, N, H, W, Cop, Cip, dWp, st);
  else return SLU_EUNSUPPORTED;
  if (rc != SLU_OK) return rc;
  const size_t total = (size_t)Cout * Cin * ksize * ksize;
  const unsigned g = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(wgrad_unpack_kernel, dim3(g), dim3(256), 0, st, dWp, Cout, Cin, ksize * ksize, Cip, dW, total);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_dgrad_weight(const float* w, int cout, int cin, int ksize, float* wd, slu_stream_t stream) {
  if (!w || !wd || cout <= 0 || cin <= 0 || ksize <= 0) return SLU_EINVAL;
  const size_t total = (size_t)cout * cin * ksize * ksize;
  const unsigned g = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(dgrad_weight_kernel, dim3(g), dim3(256), 0, slu_stream(stream), w, cout, cin, ksize, wd, total);
  SLU_CHECK_LAUNCH();
}
