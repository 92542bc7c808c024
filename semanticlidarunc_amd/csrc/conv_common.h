// Definitions shared by the conv kernels (fp32-exact and split-fp16): launch arguments, input addressing
// (concatenated sources, PixelShuffle, folded-dropout multipliers), tile-shape choice.
#pragma once
#include <utility>
#include <stdio.h>
#include <stdlib.h>

#include "slu_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace slu_conv {

struct SrcDev {
  const float* ptr;
  const float* scale;
  int C;       // channels of the stored tensor
  int ps;      // pixel-shuffle source
  int cbeg;    // first conv-input channel contributed
  int ccount;  // number of conv-input channels contributed
  int nb;      // images held by the tensor (output image n reads image n % nb); 0 = N
};

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}) -- for bodies too large for `#pragma unroll`
// whose index must stay a constant (register arrays indexed by it would otherwise move to scratch memory)
template <class F, int... Is>
__device__ __forceinline__ void slu_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void slu_static_for(F&& f) {
  slu_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

struct ConvArgs {
  SrcDev src[SLU_MAX_SRC];
  int nsrc;
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y, nchunks, nmblk;
  const float* wpack;
  const float* bias;
  const float* bn_a;
  const float* bn_b;
  const float* resid;
  float* out;
  float slope;
  int has_act;
  int vec;  // W % 4 == 0 and every base pointer 16-byte aligned: float4 staging path
  int gen;  // some source is read through PixelShuffle or carries a multiplier -- or the layer does not fit the plain kernels' fast addressing
            // (float4 path, every source starting on a K-chunk boundary, 31-bit element offsets inside one chunk of one image)
  double* stats;  // nullable: [2][Cout] per-channel sum / sum of squares of the stored output, accumulated (train-mode BatchNorm)
};

// Which source feeds conv-input channel cg (sources are channel-concatenated).
struct SrcPick {
  const float* ptr;
  const float* scale;
  int C, cl, ps, ns;       // ns = image of the source tensor this workgroup reads
};
// Source image per source for output image n (wave-uniform; computed once per workgroup, not per element).
struct SrcImg {
  int v[SLU_MAX_SRC];
};
__device__ __forceinline__ SrcImg src_images(const ConvArgs& a, int n) {
  SrcImg r;
#pragma unroll
  for (int s = 0; s < SLU_MAX_SRC; ++s) r.v[s] = (s < a.nsrc && a.src[s].nb) ? n % a.src[s].nb : n;
  return r;
}

__device__ __forceinline__ SrcPick pick_src(const ConvArgs& a, const SrcImg& im, int cg) {
  SrcPick p{a.src[0].ptr, a.src[0].scale, a.src[0].C, cg, a.src[0].ps, im.v[0]};
#pragma unroll
  for (int s = 1; s < SLU_MAX_SRC; ++s)
    if (s < a.nsrc && cg >= a.src[s].cbeg) p = SrcPick{a.src[s].ptr, a.src[s].scale, a.src[s].C, cg - a.src[s].cbeg, a.src[s].ps, im.v[s]};
  return p;
}

// one element (any W): used only when W % 4 != 0
__device__ __forceinline__ float load_input(const ConvArgs& a, const SrcImg& im, int n, int cg, int gy, int gx) {
  const SrcPick p = pick_src(a, im, cg);
  const int ns = p.ns;                          // image of the source tensor
  float v;
  int cs;
  if (!p.ps) {
    cs = p.cl;
    v = p.ptr[(((size_t)ns * p.C + cs) * a.H + gy) * a.W + gx];
  } else {
    cs = p.cl * 4 + ((gy & 1) << 1) + (gx & 1);
    v = p.ptr[(((size_t)ns * p.C + cs) * (a.H >> 1) + (gy >> 1)) * (a.W >> 1) + (gx >> 1)];
  }
  if (p.scale) v *= p.scale[(size_t)n * p.C + cs];
  return v;
}

// One float4 item of the input tile, held raw between its (unconditional, clamped-address) loads and the
// LDS write so that no wait sits between the loads of different items.
template <bool GEN>
struct Item;
template <>
struct Item<false> {
  float4 v;
};
template <>
struct Item<true> {
  float4 v;      // plain: 4 adjacent pixels; PixelShuffle: {even0, even1, odd0, odd1}
  float ku, kw;  // folded-dropout multipliers (1 when the source has none)
};

// gx4 % 4 == 0, W % 4 == 0.  `ok` false -> address clamped to element 0 of the source (always mapped).
template <bool GEN>
__device__ __forceinline__ void fetch_item(const ConvArgs& a, const SrcImg& im, int n, int cg, int gy, int gx4, bool ok, Item<GEN>& it, bool& is_ps) {
  const SrcPick p = pick_src(a, im, cg);
  const int ns = p.ns;                          // image of the source tensor (multipliers stay per output image)
  if constexpr (!GEN) {
    const size_t idx = ok ? (((size_t)ns * p.C + p.cl) * a.H + gy) * a.W + gx4 : 0;
    it.v = *reinterpret_cast<const float4*>(p.ptr + idx);
    is_ps = false;
  } else {
    const size_t hp = (size_t)(a.H >> 1) * (a.W >> 1);
    const int cs = p.ps ? p.cl * 4 + ((gy & 1) << 1) : p.cl;
    const size_t base = ((size_t)n * p.C + cs);          // multiplier index
    const size_t dbase = ((size_t)ns * p.C + cs);        // data index
    size_t i0 = p.ps ? dbase * hp + (size_t)(gy >> 1) * (a.W >> 1) + (gx4 >> 1) : (dbase * a.H + gy) * a.W + gx4;
    size_t i1 = p.ps ? i0 + hp : i0 + 2;      // second 8-byte half: next stored channel / next two pixels
    if (!ok) { i0 = 0; i1 = 0; }
    const float2 u = *reinterpret_cast<const float2*>(p.ptr + i0);
    const float2 w = *reinterpret_cast<const float2*>(p.ptr + i1);
    it.v = make_float4(u.x, u.y, w.x, w.y);
    const bool hs = ok && p.scale != nullptr;
    const float* sp = hs ? p.scale + base : a.wpack;   // any mapped address when there is no multiplier
    const float k0 = sp[0];
    const float k1 = sp[(hs && p.ps) ? 1 : 0];
    it.ku = hs ? k0 : 1.0f;
    it.kw = hs ? k1 : 1.0f;
    is_ps = p.ps != 0;
  }
}

template <bool GEN>
__device__ __forceinline__ float4 item_value(const Item<GEN>& it, bool ok, bool is_ps) {
  float4 r;
  if constexpr (!GEN) {
    r = it.v;
  } else {
    r = is_ps ? make_float4(it.v.x * it.ku, it.v.z * it.kw, it.v.y * it.ku, it.v.w * it.kw)
              : make_float4(it.v.x * it.ku, it.v.y * it.ku, it.v.z * it.ku, it.v.w * it.ku);
  }
  return ok ? r : make_float4(0.f, 0.f, 0.f, 0.f);
}


// Epilogue of the fp32 conv kernel (conv2d.hip): the accumulators of one wave (MB channel blocks x NB pixel blocks of the
// workgroup's tile at (n, y0, x0), channel blocks from mblk0 + wm * MB) -> bias, activation, folded BatchNorm, residual, store, batch statistics
// into the workgroup's LDS accumulators s_stat (flushed to a.stats by the caller after a barrier).
template <int MB, int NB, int MBLK, int RPW>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[MB][NB], const float* s_epi, double* s_stat, const float* __restrict__ resid,
                                              float* __restrict__ out, int n, int y0, int x0, int mblk0, int wm, int wn, int hh, int jj) {
  // ---- epilogue: bias, LeakyReLU, folded BatchNorm, residual, store (2 x 128 B per instruction).
  //      Per-channel constants come from LDS; `resid` / `out` are __restrict__ kernel arguments (the ABI
  //      forbids out aliasing an input), so residual loads are scheduled ahead of the stores instead of
  //      each waiting behind the previous store. ----
  const size_t plane = (size_t)a.H * a.W;
  // activation as two leaky slopes (1.0 = identity): before BatchNorm/residual, or (has_act & 4) after them
  const int act_kind = a.has_act & 3;
  const bool act_late = (a.has_act & 4) != 0, act_tanh = act_kind == 2, act_silu = act_kind == 3;
  const float slope_pre = (act_kind == 1 && !act_late) ? a.slope : 1.0f;
  const float slope_post = (act_kind == 1 && act_late) ? a.slope : 1.0f;
  const bool want_stats = a.stats != nullptr;      // wave-uniform
  // The accumulator indices must be COMPILE-TIME constants: with a plain `#pragma unroll` over i the body (NB x 16 stores + the statistics
  // butterfly) was too large for the unroller in the MB = 2, NB = 4 instantiations ("loop not unrolled"), acc[i][b] became a dynamic index and
  // the whole accumulator array lived in scratch memory -- 360 scratch loads / stores around the 288 MFMAs of the K loop of the largest tiles.
  // The rare wave-uniform options (tanh / SiLU, activation after the residual) take a generic per-element form; the common one -- bias ->
  // LeakyReLU -> folded BatchNorm [-> + residual] -- is a tight loop with the residual loads of an accumulator tile issued 8 at a time: as
  // per-element branches the options cost ~60 instructions and, with a residual, one load + s_waitcnt vmcnt(0) per stored element (128 memory
  // round trips per lane in a row).
  const bool special = act_tanh || act_silu || slope_post != 1.0f;
  slu_static_for<MB>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int ml = wm * MB + i;               // channel block inside the workgroup tile
    float ssum[16], ssq[16];                  // this lane's share of the batch statistics (its NB pixels of 16 channels)
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[r] = ssq[r] = 0.0f;
    slu_static_for<NB>([&](auto bc) __attribute__((always_inline)) {
      constexpr int b = decltype(bc)::value;
      const int gy = y0 + wn * RPW + (b >> 1), gx = x0 + (b & 1) * 32 + jj;
      const bool pix_ok = gy < a.H && gx < a.W;
      const size_t pix = (size_t)gy * a.W + gx;
      if (special) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const int co = mblk0 * 32 + cl;
          const bool ok = pix_ok && co < a.Cout;
          const size_t o = ok ? ((size_t)n * a.Cout + co) * plane + pix : 0;
          float v = acc[i][b][r] + s_epi[cl];
          v = v > 0.0f ? v : v * slope_pre;
          if (act_tanh) v = tanhf(v);
          if (act_silu) v = v / (1.0f + expf(-v));      // nn.SiLU (EfficientNetV2 blocks)
          v = v * s_epi[MBLK * 32 + cl] + s_epi[2 * MBLK * 32 + cl];
          if (resid) v += resid[o];
          v = v > 0.0f ? v : v * slope_post;
          if (ok) out[o] = v;
          if (want_stats && ok) { ssum[r] += v; ssq[r] += v * v; }
        }
      } else {
        // vmcnt counts loads AND stores in issue order: a wait for a residual load placed between two stores also waits for every older store
        // to be acknowledged (measured: the stores of a tile then went out ~8 per memory round trip, a third of the kernel on the 32-channel
        // full-resolution layers).  So: all residual loads of the tile, then all arithmetic (results back into the accumulator registers), then
        // all stores with no wait between them.
        float rv[16];
        if (resid) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = mblk0 * 32 + ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            rv[r] = resid[(pix_ok && co < a.Cout) ? ((size_t)n * a.Cout + co) * plane + pix : 0];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) rv[r] = 0.0f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          const bool ok = pix_ok && mblk0 * 32 + cl < a.Cout;
          float v = acc[i][b][r] + s_epi[cl];
          v = v > 0.0f ? v : v * slope_pre;
          v = v * s_epi[MBLK * 32 + cl] + s_epi[2 * MBLK * 32 + cl] + rv[r];
          acc[i][b][r] = v;
          if (want_stats && ok) { ssum[r] += v; ssq[r] += v * v; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = mblk0 * 32 + ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (pix_ok && co < a.Cout) out[((size_t)n * a.Cout + co) * plane + pix] = acc[i][b][r];
        }
      }
    });
    if (want_stats) {
      // the 32 lanes of a half hold 32 pixels of the same 16 channels: butterfly over them, lane 0 of each half adds the tile's share
      // to the workgroup's LDS accumulators (float), which go out as one double atomic per channel at the end
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s1 = ssum[r], s2 = ssq[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if (jj == 0) {
          const int cl = ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          atomicAdd(&s_stat[cl], (double)s1);
          atomicAdd(&s_stat[MBLK * 32 + cl], (double)s2);
        }
      }
    }
  });
}

enum TileCfg { M32_TH8 = 0, M64_TH8, M128_TH4, M32_TH4, M64_TH4 };

inline long long wg_count(const ConvArgs& a, int th, int mblk) {
  return (long long)a.N * ((a.H + th - 1) / th) * ((a.W + 63) / 64) * ((a.nmblk + mblk - 1) / mblk);
}

// Tile choice: the biggest tile that still gives >= 2 workgroups per CU (256 CUs); below that,
// shrink rows first (TH 8 -> 4), then the channel tile, so small feature maps still fill the chip.
inline int choose_cfg(const ConvArgs& a) {
  static const int forced = [] { const char* e = getenv("SLU_CONV_CFG"); return e ? atoi(e) : -1; }();      // development override: 0..4 = TileCfg
  if (forced >= 0 && forced <= 4 && !(forced == M128_TH4 && a.nmblk < 4) && !((forced == M64_TH8 || forced == M64_TH4) && a.nmblk < 2) &&
      !((forced == M32_TH8 || forced == M64_TH8) && a.H < 8))
    return forced;
  const long long want = 512;
  if (a.nmblk >= 4) {
    if (wg_count(a, 4, 4) >= want) return M128_TH4;
    if (wg_count(a, 4, 2) >= want) return M64_TH4;
    return M32_TH4;
  }
  if (a.nmblk >= 2) {
    if (a.H >= 8 && wg_count(a, 8, 2) >= want) return M64_TH8;
    if (wg_count(a, 4, 2) >= want) return M64_TH4;
    return M32_TH4;
  }
  // 32 output channels: the 8-row tile runs 3 workgroups per CU (768 slots), the 4-row tile 4 (1 024 slots).  With only a round or two of tiles
  // (a training batch of 4 scans at 64x2048 is 1 024 eight-row tiles = 1.33 rounds) the last, partly filled round costs more than the taller
  // tile saves in halo: measured at B = 4 (tools/conv_layer_time.py, SLU_CONV_CFG): 32->32 3x3 134 -> 116 us, 64->32 3x3 217 -> 188, 96->32 1x1
  // 76 -> 67.  Many rounds (inference batches): the taller tile.
  if (a.H >= 8 && wg_count(a, 8, 1) >= 4 * 768) return M32_TH8;
  return M32_TH4;
}


inline int fill_args(const slu_conv_desc* d, ConvArgs& a) {
  if (!d || !d->out || !d->wpack || d->nsrc < 1 || d->nsrc > SLU_MAX_SRC) return SLU_EINVAL;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return SLU_EINVAL;
  if (d->bn_a && !d->bn_b) return SLU_EINVAL;
  int c = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    const slu_conv_src& S = d->src[s];
    if (!S.ptr || S.C <= 0) return SLU_EINVAL;
    if (S.pixel_shuffle && ((S.C & 3) || (d->H & 1) || (d->W & 1))) return SLU_EINVAL;
    a.src[s].ptr = S.ptr;
    a.src[s].scale = S.scale;
    a.src[s].C = S.C;
    a.src[s].ps = S.pixel_shuffle ? 1 : 0;
    a.src[s].cbeg = c;
    a.src[s].ccount = S.pixel_shuffle ? S.C / 4 : (S.cuse > 0 ? S.cuse : S.C);
    // a channel prefix (cuse) is only honoured on the LAST source: the kernels bound it by Cin, not per source
    if (S.cuse < 0 || S.cuse > S.C || (S.cuse && (S.pixel_shuffle || s != d->nsrc - 1))) return SLU_EINVAL;
    a.src[s].nb = S.nbatch > 0 ? S.nbatch : 0;
    c += a.src[s].ccount;
  }
  if (c != d->Cin) return SLU_EINVAL;
  if (d->precision != SLU_CONV_FP32 && d->precision != SLU_CONV_F16X3) return SLU_EINVAL;
  if (d->ck != (d->precision == SLU_CONV_F16X3 ? 16 : slu_conv_ck(d->ksize))) return SLU_EINVAL;
  a.nsrc = d->nsrc;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
  a.nchunks = (d->Cin + d->ck - 1) / d->ck;
  a.nmblk = (d->Cout + 31) / 32;
  a.wpack = d->wpack; a.bias = d->bias; a.bn_a = d->bn_a; a.bn_b = d->bn_b; a.resid = d->resid; a.out = d->out; a.stats = d->stats;
  if (d->has_act < 0 || d->has_act > 5 || d->has_act == 4) return SLU_EINVAL;   // 0 none, 1 leaky, 2 tanh, 3 SiLU, 5 = leaky after the residual (tanh / SiLU have no late form)
  a.slope = d->slope; a.has_act = d->has_act;
  a.vec = (d->W % 4 == 0);
  a.gen = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    if (reinterpret_cast<uintptr_t>(d->src[s].ptr) & 15) a.vec = 0;
    if (d->src[s].pixel_shuffle || d->src[s].scale) a.gen = 1;
    if (a.src[s].cbeg % d->ck) a.gen = 1;
  }
  if (!a.vec || (long long)(d->ck + 1) * d->H * d->W >= 0x7fffffffLL || (long long)a.nmblk * a.nchunks * 64 * 9 * 8 >= 0x7fffffffLL) a.gen = 1;
  return SLU_OK;
}


}  // namespace slu_conv
