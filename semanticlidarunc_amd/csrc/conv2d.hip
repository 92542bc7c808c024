// Fused direct convolution for 2-D range images on gfx950 (MI355X), fp32 in / fp32 out, exact fp32
// arithmetic on the matrix cores (v_mfma_f32_32x32x2_f32 == a k-ordered fmaf chain).
//
// Replaces, per call, the chain  torch.cat / nn.PixelShuffle(2) / nn.Dropout2d -> nn.Conv2d ->
// nn.LeakyReLU -> nn.BatchNorm2d(eval) -> residual add  of the reference's SalsaNext blocks
// (src/baselines/SalsaNext/SalsaNext.py:25-39, :73-109, :142-170, :213).
//
// Mapping (implicit GEMM, one 64-lane wave = one or more 32x32 MFMA tiles):
//   M = output channels   (A operand = weights, pre-packed in fragment order by slu_pack_conv_weight)
//   N = 32 azimuth-adjacent pixels of one image row (B operand, read from an LDS input tile)
//   K = (input channel, tap) pairs, two input channels per MFMA (lane>>5 selects the channel)
// The D fragment has the pixel on the lane (col = lane&31) and the output channel on the register,
// so every store instruction writes two 128-byte row segments (coalesced along azimuth).
//
// Work decomposition: a workgroup owns TH rows x 64 columns of one image and WM*MB*32 output
// channels; per K-chunk (CK input channels) it stages the (TH+2*pad) x (64+2*pad) halo tile of
// each channel and the matching weight fragments into LDS, then every wave runs KS*KS*CK/2 K-steps.
#include <stdlib.h>

#include "conv_common.h"

using namespace slu_conv;

#ifdef SLU_CONV_PROF      // development aid: per-phase shader-clock totals of wave 0 of every workgroup of conv_fwd_kernel
__device__ unsigned long long g_conv_prof[8];
#define CONV_PROF_MARK(i)                                          \
  {                                                                \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
    asm volatile("" ::: "memory");                                 \
    prof_acc[i] += t_now - prof_t;                                 \
    prof_t = t_now;                                                \
  }
#else
#define CONV_PROF_MARK(i)
#endif

namespace {

// GEN = false: plain sources (no PixelShuffle, no multipliers) that start on chunk boundaries, W % 4 == 0 -- every conv except UpBlock.conv1.
template <int KS, int DIL, int PAD, int CK, int MB, int WM, int WN, int RPW, bool GEN>
__global__ __launch_bounds__(64 * WM * WN, (MB * RPW >= 4) ? 2 : ((MB * RPW >= 2) ? 3 : 4)) void conv_fwd_kernel(const ConvArgs a, const float* __restrict__ resid, float* __restrict__ out) {
  constexpr int NT = 64 * WM * WN;
  constexpr bool _PIN = true;
  constexpr int TW = 64, TH = WN * RPW, NB = 2 * RPW;
  constexpr int XO = PAD ? 4 : 0;                 // the LDS tile starts XO (16-byte aligned) columns left of x0
  constexpr int LW = TW + 2 * XO, LH = TH + 2 * PAD, LW4 = LW / 4;
  constexpr int PLANE = LH * LW;
  constexpr int HALF = CK / 2;
  constexpr int KSTEPS = KS * KS * HALF;
  constexpr int MBLK = WM * MB;
  constexpr int NIV = CK * LH * LW4, NI = (NIV + NT - 1) / NT;              // float4 items of the input tile
  constexpr int NWV = MBLK * KSTEPS * 16, NW = (NWV + NT - 1) / NT;          // float4 items of the weight tile

  __shared__ __attribute__((aligned(16))) float s_in[CK * PLANE];
  __shared__ __attribute__((aligned(16))) float s_w[MBLK * KSTEPS * 64];
  __shared__ float s_epi[3 * MBLK * 32];          // bias | bn_a | bn_b of the workgroup's output channels
  __shared__ double s_stat[2 * MBLK * 32];        // sum | sum of squares of the stored values per output channel (a.stats); fp64 so that
                                                  // the order in which the waves arrive cannot change the result at fp32 level

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // Workgroups are dealt round-robin over the 8 XCDs: give every XCD one contiguous run of tiles so that
  // the halo rows/columns neighbouring tiles share are served by that XCD's own L2 (speed only).
  int t = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = t & 7, qq = nwg >> 3, rr = nwg & 7;
    t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (t >> 3);
  }
  const int tx = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int n = t / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int mblk0 = blockIdx.y * MBLK;
  const SrcImg im = src_images(a, n);

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.0f;

  if (tid < MBLK * 32) {   // visible to everyone after the first barrier of the chunk loop
    const int co = mblk0 * 32 + tid;
    const bool ok = co < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[co] : 0.0f;
    s_epi[MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_a[co] : 1.0f;
    s_epi[2 * MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_b[co] : 0.0f;
    s_stat[tid] = 0.0;
    s_stat[MBLK * 32 + tid] = 0.0;
  }

  const int hh = lane >> 5, jj = lane & 31;
  const int bbase = hh * PLANE + (wn * RPW) * LW + jj + (XO - PAD);
  const int abase = (wm * MB) * KSTEPS * 64 + lane;

#ifdef SLU_CONV_PROF      // phases: barrier before staging | staged registers -> LDS (waits for the loads) | barrier | next chunk's loads issued | MFMA phase | epilogue
  unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
  // Software pipeline over the K-chunks: the global loads of chunk q + 1 (weight fragments + halo tile, ~10 float4 per lane) are ISSUED before the
  // MFMA phase of chunk q and written to LDS after it, so their latency (an L2 / HBM round trip, which was a third of every wave's clocks when
  // each chunk loaded, waited, wrote and only then multiplied) hides behind 18 000 cycles of matrix work.  One LDS buffer: the writes wait for
  // the barrier that ends the phase reading it.
  float4 sw[NW];
  Item<GEN> st[NI];
  unsigned okm = 0, psm = 0;
  // GEN = false (plain sources whose channel ranges start on chunk boundaries, W % 4 == 0: every layer of SalsaNext but UpBlock.conv1): what a
  // lane loads differs from chunk to chunk only by a wave-uniform base, so the per-lane part -- a 32-bit element offset per item and the
  // validity bits -- is computed ONCE here.  Recomputing the full 64-bit index chains per chunk (to keep them out of registers across the MFMA
  // phase) cost 25-35 % of every wave's clocks in integer multiplies (measured with -DSLU_CONV_PROF), whatever the memory did.
  int in_off[GEN ? 1 : NI], w_off[GEN ? 1 : NW];
  unsigned sp_ok = 0, w_ok = 0, ci_pack[GEN ? 1 : (NI + 7) / 8];
  if constexpr (!GEN) {
    const int plane_i = a.H * a.W;
#pragma unroll
    for (int k = 0; k < (NI + 7) / 8; ++k) ci_pack[k] = 0;
    // (compile-time loops: with `#pragma unroll` these arrays were indexed dynamically and lived in scratch memory)
    slu_static_for<NI>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + i * NT;
      const int ci = e / (LH * LW4);
      const int rem = e - ci * (LH * LW4);
      const int r = rem / LW4;
      const int c4 = rem - r * LW4;
      const int gy = y0 + r - PAD, gx4 = x0 - XO + 4 * c4;
      const bool ok = (NIV % NT == 0 || e < NIV) && gy >= 0 && gy < a.H && gx4 >= 0 && gx4 < a.W;
      in_off[i] = ok ? ci * plane_i + gy * a.W + gx4 : 0;
      sp_ok |= (ok ? 1u : 0u) << i;
      ci_pack[i / 8] |= (unsigned)(ci & 15) << (4 * (i % 8));
    });
    slu_static_for<NW>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      const int e = tid + i * NT;
      const int m = e / (KSTEPS * 16);
      const int r = e - m * (KSTEPS * 16);
      const bool ok = (NWV % NT == 0 || e < NWV) && mblk0 + m < a.nmblk;
      w_off[i] = ok ? m * a.nchunks * (KSTEPS * 64) + 4 * r : 0;
      w_ok |= (ok ? 1u : 0u) << i;
    });
  }
  static_assert(GEN || CK <= 16, "ci_pack holds 4 bits per item");
  auto fetch = [&](int q) __attribute__((always_inline)) {
    if constexpr (!GEN) {
      const float* wq = a.wpack + ((size_t)mblk0 * a.nchunks + q) * (KSTEPS * 64);
      slu_static_for<NW>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        sw[i] = *reinterpret_cast<const float4*>(wq + w_off[i]);
      });
      // the source this chunk lies in (wave-uniform: the host routes layers whose sources do not start on chunk boundaries to GEN = true)
      const int c0 = q * CK;
      const SrcPick sp = pick_src(a, im, c0);       // (a chain of overwrites: written as "find the index, then index" the source table went to scratch memory)
      const float* base = sp.ptr + ((size_t)sp.ns * sp.C + sp.cl) * ((size_t)a.H * a.W);
      const int lim = a.Cin - c0;                 // channels of this chunk that exist (the last chunk of the last source may be short)
      okm = 0;
      slu_static_for<NI>([&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        const int ci = (int)((ci_pack[i / 8] >> (4 * (i % 8))) & 15u);
        const bool ok = ((sp_ok >> i) & 1u) && ci < lim;
        st[i].v = *reinterpret_cast<const float4*>(base + (ok ? in_off[i] : 0));
        okm |= (ok ? 1u : 0u) << i;
      });
      return;
    }
    // keep the (chunk-invariant) staging address arithmetic from being hoisted out of the chunk loop and held in registers across the MFMA
    // phase: recompute it per chunk from an opaque copy of tid
    int tq = tid;
    asm volatile("" : "+v"(tq));
    // ---- A fragments of the chunk (L2-resident, MBLK contiguous runs of KSTEPS*64 floats) ----
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tq + i * NT;
      int m = e / (KSTEPS * 16);
      const int r = e - m * (KSTEPS * 16);
      const bool ok = (NWV % NT == 0 || e < NWV) && mblk0 + m < a.nmblk;
      // out-of-range fragments read the packed weights' first record (always mapped) and are zeroed when they are written to LDS
      const size_t off = ok ? ((size_t)(mblk0 + m) * a.nchunks + q) * (KSTEPS * 64) + 4 * r : 0;
      sw[i] = *reinterpret_cast<const float4*>(a.wpack + off);
    }
    // ---- halo tile of CK input channels (zero outside the image / beyond Cin): all loads of a thread are issued back to back
    //      (branch-free: clamped address, the select happens at the LDS write) ----
    okm = psm = 0;
    if (a.vec) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int e = tq + i * NT;
        const int ci = e / (LH * LW4);
        const int rem = e - ci * (LH * LW4);
        const int r = rem / LW4;
        const int c4 = rem - r * LW4;
        const int gy = y0 + r - PAD, gx4 = x0 - XO + 4 * c4, cg = q * CK + ci;
        const bool ok = (NIV % NT == 0 || e < NIV) && cg < a.Cin && gy >= 0 && gy < a.H && gx4 >= 0 && gx4 < a.W;
        bool ps;
        fetch_item<GEN>(a, im, n, ok ? cg : 0, gy, gx4, ok, st[i], ps);
        okm |= (ok ? 1u : 0u) << i;
        psm |= (ps ? 1u : 0u) << i;
      }
    }
  };
  auto commit = [&](int q) __attribute__((always_inline)) {
    int tq = tid;
    asm volatile("" : "+v"(tq));
    if (!GEN || a.vec) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int e = tq + i * NT;
        if (NIV % NT == 0 || e < NIV)
          reinterpret_cast<float4*>(s_in)[e] = item_value<GEN>(st[i], (okm >> i) & 1u, (psm >> i) & 1u);
      }
    } else {      // W % 4 != 0 (no layer of the models): element by element, not pipelined
      for (int e = tq; e < CK * PLANE; e += NT) {
        const int ci = e / PLANE;
        const int rem = e - ci * PLANE;
        const int r = rem / LW;
        const int c = rem - r * LW;
        const int gy = y0 + r - PAD, gx = x0 + c - XO, cg = q * CK + ci;
        float v = 0.0f;
        if (cg < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = load_input(a, im, n, cg, gy, gx);
        s_in[e] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tq + i * NT;
      bool ok;
      if constexpr (GEN) ok = (NWV % NT == 0 || e < NWV) && mblk0 + e / (KSTEPS * 16) < a.nmblk;
      else ok = (w_ok >> i) & 1u;
      if (NWV % NT == 0 || e < NWV) reinterpret_cast<float4*>(s_w)[e] = ok ? sw[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  fetch(0);
  for (int q = 0; q < a.nchunks; ++q) {
    __syncthreads();                       // every wave has finished reading chunk q - 1 from LDS
    CONV_PROF_MARK(0)
    commit(q);
#ifdef SLU_CONV_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    CONV_PROF_MARK(1)
    __syncthreads();
    CONV_PROF_MARK(2)
    if (q + 1 < a.nchunks) fetch(q + 1);   // in flight during the MFMA phase below
    CONV_PROF_MARK(3)
    // ---- K-steps: the MB + NB operand reads of step s + 1 are issued before the MB * NB MFMAs of step s, in THIS order (sched_barrier pins
    //      it; left alone the compiler emits read, wait, MFMA, read, wait, ... and every 64-cycle MFMA starts with an LDS round trip) ----
    float av[2][MB], bv[2][NB];
    auto read_step = [&](int s, int set) {
      const int tap = s / HALF, pp = s % HALF;
      const int dy = (tap / KS) * DIL, dx = (tap % KS) * DIL;
#pragma unroll
      for (int i = 0; i < MB; ++i) av[set][i] = s_w[abase + (i * KSTEPS + s) * 64];
#pragma unroll
      for (int b = 0; b < NB; ++b) bv[set][b] = s_in[bbase + (2 * pp) * PLANE + ((b >> 1) + dy) * LW + (b & 1) * 32 + dx];
    };
    read_step(0, 0);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      if (s + 1 < KSTEPS) read_step(s + 1, (s + 1) & 1);
      if (_PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][i], bv[s & 1][b], acc[i][b], 0, 0, 0);
      if (_PIN) __builtin_amdgcn_sched_barrier(0);
    }
#ifdef SLU_CONV_PROF
    asm volatile("s_nop 0" ::"v"(acc[MB - 1][NB - 1][0]));      // the last MFMA has retired
#endif
    CONV_PROF_MARK(4)
  }

  // ---- epilogue: bias, LeakyReLU, folded BatchNorm, residual, store (conv_common.h) ----
  const bool want_stats = a.stats != nullptr;      // wave-uniform
  conv_epilogue<MB, NB, MBLK, RPW>(a, acc, s_epi, s_stat, resid, out, n, y0, x0, mblk0, wm, wn, hh, jj);
  if (want_stats) {
    __syncthreads();
    if (tid < MBLK * 32) {
      const int co = mblk0 * 32 + tid;
      if (co < a.Cout) {
        atomicAdd(&a.stats[co], s_stat[tid]);
        atomicAdd(&a.stats[a.Cout + co], s_stat[MBLK * 32 + tid]);
      }
    }
  }
#ifdef SLU_CONV_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CONV_PROF_MARK(5)
  if (tid == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_conv_prof[i], prof_acc[i]);
    atomicAdd(&g_conv_prof[6], 1ull);
  }
#endif
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int ks, int ck, int nchunks,
                                   size_t total, float* __restrict__ out) {
  const int half = ck / 2, ksteps = ks * ks * half;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    size_t r = e >> 6;
    const int s = (int)(r % ksteps);
    r /= ksteps;
    const int q = (int)(r % nchunks);
    const int m = (int)(r / nchunks);
    const int tap = s / half, pp = s % half;
    const int ci = q * ck + 2 * pp + (lane >> 5);
    const int co = m * 32 + (lane & 31);
    float v = 0.0f;
    if (co < cout && ci < cin) v = w[((size_t)co * cin + ci) * (ks * ks) + tap];
    out[e] = v;
  }
}

// All conv weights of a model in ONE launch (the training step repacks every weight after every optimizer step: 51 forward images + 50
// data-gradient images were 151 launches of ~4.5 us).  A job packs one weight in the layout of pack_weight_kernel; dgrad = 1 packs the weights
// of the data-gradient conv instead (channels swapped, taps mirrored: slu_dgrad_weight followed by slu_pack_conv_weight, without the copy).
__global__ void pack_weight_multi_kernel(const slu_pack_job* __restrict__ jobs, int njobs, size_t total) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs - 1;                            // last job whose first element is <= e
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].begin <= e) lo = mid; else hi = mid - 1;
    }
    const slu_pack_job j = jobs[lo];
    const size_t le = e - j.begin;
    const int co_n = j.dgrad ? j.cin : j.cout, ci_n = j.dgrad ? j.cout : j.cin;      // dimensions of the packed conv
    const int ks = j.ksize, T = ks * ks, half = j.ck / 2, ksteps = T * half, nchunks = (ci_n + j.ck - 1) / j.ck;
    const int lane = (int)(le & 63);
    size_t r = le >> 6;
    const int s = (int)(r % ksteps);
    r /= ksteps;
    const int q = (int)(r % nchunks);
    const int m = (int)(r / nchunks);
    const int tap = s / half, pp = s % half;
    const int ci = q * j.ck + 2 * pp + (lane >> 5);
    const int co = m * 32 + (lane & 31);
    float v = 0.0f;
    if (co < co_n && ci < ci_n)
      v = j.dgrad ? j.w[((size_t)ci * j.cin + co) * T + (T - 1 - tap)] : j.w[((size_t)co * j.cin + ci) * T + tap];
    j.out[le] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------------
template <int KS, int DIL, int PAD, int CK, int MB, int WM, int WN, int RPW, bool GEN>
int launch_cfg(ConvArgs& a, hipStream_t st) {
  constexpr int TH = WN * RPW, MBLK = WM * MB;
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long gx = (long long)a.tiles_x * a.tiles_y * a.N;
  const int gy = (a.nmblk + MBLK - 1) / MBLK;
  if (gx <= 0 || gx > 0x7fffffffLL || gy > 65535) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL((conv_fwd_kernel<KS, DIL, PAD, CK, MB, WM, WN, RPW, GEN>), dim3((unsigned)gx, (unsigned)gy), dim3(64 * WM * WN),
                     0, st, a, a.resid, a.out);
  SLU_CHECK_LAUNCH();
}

template <int KS, int DIL, int PAD, int CK, bool GEN>
int launch_tiles(ConvArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case M32_TH8:  return launch_cfg<KS, DIL, PAD, CK, 1, 1, 4, 2, GEN>(a, st);
    case M64_TH8:  return launch_cfg<KS, DIL, PAD, CK, 2, 1, 4, 2, GEN>(a, st);
    case M128_TH4: return launch_cfg<KS, DIL, PAD, CK, 2, 2, 2, 2, GEN>(a, st);
    case M32_TH4:  return launch_cfg<KS, DIL, PAD, CK, 1, 1, 4, 1, GEN>(a, st);
    case M64_TH4:  return launch_cfg<KS, DIL, PAD, CK, 2, 1, 4, 1, GEN>(a, st);
  }
  return SLU_EUNSUPPORTED;
}

template <int KS, int DIL, int PAD, int CK>
int launch_family(ConvArgs& a, int cfg, hipStream_t st) {
  return a.gen ? launch_tiles<KS, DIL, PAD, CK, true>(a, cfg, st) : launch_tiles<KS, DIL, PAD, CK, false>(a, cfg, st);
}

}  // namespace

extern "C" int slu_conv_ck(int ksize) { return ksize == 1 ? 16 : 8; }

extern "C" size_t slu_packed_weight_floats(int cout, int cin, int ksize, int ck) {
  if (cout <= 0 || cin <= 0 || ksize <= 0 || ck <= 0 || (ck & 1)) return 0;
  const size_t nmblk = (cout + 31) / 32, nchunks = (cin + ck - 1) / ck;
  return nmblk * nchunks * (size_t)(ksize * ksize * ck / 2) * 64;
}

extern "C" int slu_pack_conv_weight(const float* w, int cout, int cin, int ksize, int ck, float* out, slu_stream_t stream) {
  if (!w || !out) return SLU_EINVAL;
  const size_t total = slu_packed_weight_floats(cout, cin, ksize, ck);
  if (total == 0) return SLU_EINVAL;
  const int nchunks = (cin + ck - 1) / ck;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, slu_stream(stream), w, cout, cin, ksize, ck, nchunks, total, out);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_pack_conv_weights_multi(const slu_pack_job* jobs_dev, int njobs, size_t total, slu_stream_t stream) {
  if (!jobs_dev || njobs <= 0 || total == 0) return SLU_EINVAL;
  const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(blocks), dim3(256), 0, slu_stream(stream), jobs_dev, njobs, total);
  SLU_CHECK_LAUNCH();
}

#ifdef SLU_CONV_PROF
extern "C" int slu_conv_prof_read(unsigned long long* out8) {      // copies and clears the phase clocks (synchronises the device)
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_prof), sizeof(g_conv_prof)) != hipSuccess) return SLU_ELAUNCH;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_conv_prof), z, sizeof(z)) == hipSuccess ? SLU_OK : SLU_ELAUNCH;
}
#endif

int slu_conv2d_fwd_f16x3_impl(const slu_conv_desc* d, hipStream_t st);   // conv2d_f16x3.hip

extern "C" int slu_conv2d_fwd(const slu_conv_desc* d, slu_stream_t stream) {
  if (d && d->precision == SLU_CONV_F16X3) return slu_conv2d_fwd_f16x3_impl(d, slu_stream(stream));
  ConvArgs a{};
  const int rc = fill_args(d, a);
  if (rc != SLU_OK) return rc;
  const int cfg = choose_cfg(a);
  hipStream_t st = slu_stream(stream);
  if (d->ksize == 1 && d->dil == 1 && d->pad == 0) return launch_family<1, 1, 0, 16>(a, cfg, st);
  if (d->ksize == 3 && d->dil == 1 && d->pad == 1) return launch_family<3, 1, 1, 8>(a, cfg, st);
  if (d->ksize == 3 && d->dil == 2 && d->pad == 2) return launch_family<3, 2, 2, 8>(a, cfg, st);
  if (d->ksize == 2 && d->dil == 2 && d->pad == 1) return launch_family<2, 2, 1, 8>(a, cfg, st);
  if (d->ksize == 2 && d->dil == 1 && d->pad == 1) return launch_family<2, 1, 1, 8>(a, cfg, st);
  return SLU_EUNSUPPORTED;
}

// Template arguments of the instantiation slu_conv2d_fwd would launch for this descriptor, in the
// order rocprofv3 prints them: "conv_fwd_kernel<KS, DIL, PAD, CK, MB, WM, WN, RPW>".
extern "C" int slu_conv2d_kernel_name(const slu_conv_desc* d, char* buf, size_t buflen) {
  if (!buf || buflen < 64) return SLU_EINVAL;
  ConvArgs a{};
  const int rc = fill_args(d, a);
  if (rc != SLU_OK) return rc;
  static const int kTile[5][4] = {{1, 1, 4, 2}, {2, 1, 4, 2}, {2, 2, 2, 2}, {1, 1, 4, 1}, {2, 1, 4, 1}};
  int cfg = choose_cfg(a);
  if (d->precision == SLU_CONV_F16X3 && cfg == M32_TH8) cfg = M32_TH4;
  if (d->precision == SLU_CONV_F16X3 && cfg == M64_TH8) cfg = M64_TH4;
  const int* t = kTile[cfg];
  if (d->precision == SLU_CONV_F16X3 && d->ksize == 1 && !a.gen && a.nmblk <= 8 && ((long long)a.H * a.W) % 32 == 0 &&
      (a.nsrc < 2 || a.src[0].ccount % 16 == 0) && (a.nsrc < 3 || a.src[1].ccount % 16 == 0)) {
    const int mb = a.nmblk == 1 ? 1 : (a.nmblk == 2 ? 2 : (a.nmblk <= 4 ? 4 : 8));
    snprintf(buf, buflen, "conv1x1_f16x3_kernel<%d, %d>", mb, 1);
    return SLU_OK;
  }
  if (d->precision == SLU_CONV_F16X3)
    snprintf(buf, buflen, "conv_f16x3_kernel<%d, %d, %d, %d, %d, %d, %d, %s>", d->ksize, d->dil, d->pad, t[0], t[1], t[2], t[3], a.gen ? "true" : "false");
  else
    snprintf(buf, buflen, "conv_fwd_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %s>", d->ksize, d->dil, d->pad, d->ck, t[0], t[1], t[2], t[3], a.gen ? "true" : "false");
  return SLU_OK;
}
