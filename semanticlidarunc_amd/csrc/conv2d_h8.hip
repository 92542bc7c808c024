// Half-precision storage path ("h8"): activations live in HBM as fp16 in channel blocks of 8,
//     x[N][G = C/8][H][W][8]         (16 bytes per (pixel, block); azimuth-adjacent pixels are adjacent records)
// and every conv multiplies fp16 x fp16 on v_mfma_f32_32x32x16_f16 with fp32 accumulation and an fp32 epilogue
// (bias, LeakyReLU, folded BatchNorm, residual) before rounding the result to fp16 once.  The layout makes ONE
// 16-byte record = ONE MFMA B operand of one lane (lane (pixel r, half h) holds the 8 channels of block 2k+h), so
// staging a tile is a plain 16-byte copy (no conversion, no transpose), a wave's global loads / stores are 1 KB
// contiguous along the azimuth, and HBM traffic is half of the fp32 path.  BASELINE.json configs[2],[4] name this
// storage precision ("bf16"); fp16 is used instead because the activations of the range-image stack are O(1..100)
// and fp16's 11-bit mantissa keeps the logits within the 1e-3 parity bar (bf16 does not).
//
//   conv_h8_kernel      3x3 / dilated / 2x2 convs: LDS tile [blocks][rows+halo][cols+halo] of 16-byte records,
//                       register prefetch of the next channel chunk during the MFMA phase
//   conv1x1_h8_kernel   1x1 convs: no halo => B operands straight from global memory, weights through LDS
//   plus layout / pooling / pixel-shuffle helpers at the end of the file.
#include <stdio.h>
#include "slu_common.h"
#include <cstdlib>
#include <utility>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

// scheduling options of the 8-wave 128-channel configuration (conv_h8_kernel's OPT), as measured with tools/h8_ab.py (one process,
// interleaved rounds, N = 64, single-source form): 15 is +1 ... +3 % on the 3x3 layers and +3 ... +7 % on the 2x2-dilated ones; two K-steps
// per barrier paid (+6 ... +8 %) only while the per-chunk set-up was expensive and costs 2 ... 4 % since it is one scalar multiply-add
#ifndef H8_M128_OPT_3X3
#define H8_M128_OPT_3X3 15
#endif
#ifndef H8_M128_OPT_2X2
#define H8_M128_OPT_2X2 15
#endif
#ifndef H8_M128_KPC2_DEFAULT
#define H8_M128_KPC2_DEFAULT 0      // 2x2-dilated 128 / 256-channel layers: two K-steps per barrier (kept as an A/B switch, SLU_H8_KPC2=1)
#endif

namespace {

struct H8Src {
  const uint4* ptr;    // [nimg][G][H][W] records
  const float* scale;  // [N][8 G] fp32 multiplier per (output image, channel) or nullptr
  int G;               // channel blocks of the stored tensor
  int gbeg;            // first block of this source in the concatenated input
  int nb;              // 0: holds N images; k > 0: holds k images, output image n reads n % k
};

struct H8Args {
  H8Src src[SLU_MAX_SRC];
  int nsrc;
  int N, H, W, Gin, Cout, Gout, nmblk, nks;   // nks = 16-channel K-steps = ceil(Gin / 2)
  const uint4* wpack;
  const float *bias, *bn_a, *bn_b;
  int has_act;
  float slope;
  int out_f32;         // 1: `out` is fp32 NCHW [N][Cout][H][W] (the logits head); 0: h8
  int tiles_x, tiles_y;
  int order;                 // 0: each workgroup walks a contiguous run of tiles; 1: tiles interleaved across workgroups
  int dbg;                   // development switches of gemm1x1_h8_kernel (SLU_GEMM_DBG): 1 no input DMA, 2 no weight DMA, 4 no MFMA
};

struct SrcSel {
  const uint4* ptr;
  const float* scale;
  int G, gl, ns;
};

__device__ __forceinline__ SrcSel select_src(const H8Args& a, const int (&img)[SLU_MAX_SRC], int g) {
  SrcSel p{a.src[0].ptr, a.src[0].scale, a.src[0].G, g, img[0]};
#pragma unroll
  for (int s = 1; s < SLU_MAX_SRC; ++s)
    if (s < a.nsrc && g >= a.src[s].gbeg) p = SrcSel{a.src[s].ptr, a.src[s].scale, a.src[s].G, g - a.src[s].gbeg, img[s]};
  return p;
}

// 8 halves * 8 fp32 multipliers (rounded to fp16 first, then packed multiplies)
__device__ __forceinline__ uint4 scale_record(uint4 v, const float* sp) {
  const float4 s0 = *reinterpret_cast<const float4*>(sp);
  const float4 s1 = *reinterpret_cast<const float4*>(sp + 4);
  half8 h = __builtin_bit_cast(half8, v);
  h[0] *= (_Float16)s0.x; h[1] *= (_Float16)s0.y; h[2] *= (_Float16)s0.z; h[3] *= (_Float16)s0.w;
  h[4] *= (_Float16)s1.x; h[5] *= (_Float16)s1.y; h[6] *= (_Float16)s1.z; h[7] *= (_Float16)s1.w;
  return __builtin_bit_cast(uint4, h);
}

__device__ __forceinline__ unsigned pack2(float x, float y) {
  half2v h;
  h[0] = (_Float16)x;      // round to nearest even
  h[1] = (_Float16)y;
  return __builtin_bit_cast(unsigned, h);
}

// Epilogue shared by both kernels: one 32-channel x 32-pixel accumulator tile of a lane -> 4 x (4 channels)
//   chan0: first channel of the 32-block; se: LDS constants bias | bn_a | bn_b indexed by `cl0 + ...`
template <int STRIDE>
__device__ __forceinline__ void store_tile(const H8Args& a, const f32x16& acc, const float* se, int cl0, int co0, int hh, bool pix_ok,
                                           size_t n, size_t pix, size_t HW, const void* __restrict__ resid, void* __restrict__ out,
                                           float slope_pre) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cl = cl0 + 8 * q + 4 * hh + k;
      float t = acc[4 * q + k] + se[cl];
      t = t > 0.0f ? t : t * slope_pre;
      v[k] = t * se[STRIDE + cl] + se[2 * STRIDE + cl];
    }
    const int co = co0 + 8 * q + 4 * hh;           // first of this lane's 4 channels
    if (a.out_f32) {
      float* o = reinterpret_cast<float*>(out);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (pix_ok && co + k < a.Cout) o[(n * a.Cout + co + k) * HW + pix] = v[k];
    } else {
      const int go = co >> 3;
      const bool ok = pix_ok && go < a.Gout;
      const size_t idx = ok ? ((n * a.Gout + go) * HW + pix) * 2 + hh : 0;   // 8-byte half records
      if (resid) {
        const uint2 rv = reinterpret_cast<const uint2*>(resid)[idx];
        const half2v r0 = __builtin_bit_cast(half2v, rv.x), r1 = __builtin_bit_cast(half2v, rv.y);
        v[0] += (float)r0[0]; v[1] += (float)r0[1]; v[2] += (float)r1[0]; v[3] += (float)r1[1];
      }
      if (ok) reinterpret_cast<uint2*>(out)[idx] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
    }
  }
}

// the source of every out-of-range / padding record of an LDS-DMA copy (never written)
__device__ uint4 g_zero_rec;
// where the lanes of a border tile that lie outside the image store (so that every lane of every tile issues its stores)
__device__ uint4 g_trash_rec;

// Epilogue of the persistent kernel (h8 output): EVERY lane issues its 4 stores (and its 4 residual loads when they were
// not prefetched) -- lanes outside the image / past the last channel block read the zero record and store to a scratch
// record -- so the number of vector-memory operations per tile is a compile-time constant the kernel's counted waits rely on.
template <int STRIDE, bool PRE>
__device__ __forceinline__ void store_tile_full(const H8Args& a, const f32x16& acc, const float* se, int cl0, int go0, int hh, bool pix_ok, size_t n,
                                                size_t pix, size_t HW, const uint2* __restrict__ resid, const uint2 (&rv)[4],
                                                uint2* __restrict__ out, float slope_pre, uintptr_t zero_addr, uintptr_t trash_addr) {
  // packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma: two channels per instruction); LeakyReLU as max(t, slope t),
  // exact for 0 <= slope <= 1 (slope_pre = 1 means "no activation"); the per-channel constants come as 16-byte LDS reads
  const float4* se4 = reinterpret_cast<const float4*>(se);
  const float2v sl = {slope_pre, slope_pre};
  const size_t plane2 = HW * 2;
  const size_t idx0 = ((n * a.Gout + go0) * HW + pix) * 2 + hh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c4 = (cl0 + 8 * q) / 4 + hh;
    const float4 bi = se4[c4], ba = se4[STRIDE / 4 + c4], bb = se4[2 * STRIDE / 4 + c4];
    float2v t0 = {acc[4 * q], acc[4 * q + 1]}, t1 = {acc[4 * q + 2], acc[4 * q + 3]};
    t0 += float2v{bi.x, bi.y};
    t1 += float2v{bi.z, bi.w};
    t0 = __builtin_elementwise_max(t0, t0 * sl);
    t1 = __builtin_elementwise_max(t1, t1 * sl);
    t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
    t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
    const bool ok = pix_ok && go0 + q < a.Gout;
    const size_t idx = idx0 + q * plane2;
    if (resid) {
      const uint2 r = PRE ? rv[q] : *(ok ? resid + idx : reinterpret_cast<const uint2*>(zero_addr));
      t0 += __builtin_convertvector(__builtin_bit_cast(half2v, r.x), float2v);
      t1 += __builtin_convertvector(__builtin_bit_cast(half2v, r.y), float2v);
    }
    *(ok ? out + idx : reinterpret_cast<uint2*>(trash_addr)) =
        make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v)), __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v)));
  }
}

// The same epilogue with whole 16-byte records on the way out: the accumulator holds half of each record per lane (4 channels of
// block q in lane half hh); v_permlane32_swap trades halves between lanes jj and jj + 32 so that lane half 0 stores the record of
// block 2 pr and lane half 1 that of block 2 pr + 1 -- 2 store instructions per accumulator tile instead of 4.  The residual (rare on
// the layers this serves) is added before the exchange from 8-byte loads of the lane's own channels.
template <int STRIDE>
__device__ __forceinline__ void store_tile_swap16(const H8Args& a, const f32x16& acc, const float* se, int cl0, int go0, int hh, bool pix_ok, size_t n,
                                                  size_t pix, size_t HW, const uint2* __restrict__ resid, uint4* __restrict__ out, float slope_pre,
                                                  uintptr_t zero_addr, uintptr_t trash_addr) {
  const float4* se4 = reinterpret_cast<const float4*>(se);
  const float2v sl = {slope_pre, slope_pre};
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    unsigned hw[4];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int q = 2 * pr + q2;
      const int c4 = (cl0 + 8 * q) / 4 + hh;
      const float4 bi = se4[c4], ba = se4[STRIDE / 4 + c4], bb = se4[2 * STRIDE / 4 + c4];
      float2v t0 = {acc[4 * q], acc[4 * q + 1]}, t1 = {acc[4 * q + 2], acc[4 * q + 3]};
      t0 += float2v{bi.x, bi.y};
      t1 += float2v{bi.z, bi.w};
      t0 = __builtin_elementwise_max(t0, t0 * sl);
      t1 = __builtin_elementwise_max(t1, t1 * sl);
      t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
      t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
      if (resid) {
        const bool okq = pix_ok && go0 + q < a.Gout;
        const uint2 r = *(okq ? resid + (((n * a.Gout + go0 + q) * HW + pix) * 2 + hh) : reinterpret_cast<const uint2*>(zero_addr));
        t0 += __builtin_convertvector(__builtin_bit_cast(half2v, r.x), float2v);
        t1 += __builtin_convertvector(__builtin_bit_cast(half2v, r.y), float2v);
      }
      hw[2 * q2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v));
      hw[2 * q2 + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v));
    }
    const auto s0 = __builtin_amdgcn_permlane32_swap(hw[0], hw[2], false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(hw[1], hw[3], false, false);
    const int go = go0 + 2 * pr + hh;
    uint4* dst = (pix_ok && go < a.Gout) ? out + ((n * a.Gout + go) * HW + pix) : reinterpret_cast<uint4*>(trash_addr);
    *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
  }
}

#ifdef SLU_H8_PROF      // development aid: per-phase shader-clock totals of wave 0 of every workgroup of the tiled kernel
__device__ unsigned long long g_h8_prof[8];
#define H8_PROF_MARK(i)                                         \
  {                                                             \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime(); \
    asm volatile("" ::: "memory");                              \
    prof_acc[i] += t_now - prof_t;                              \
    prof_t = t_now;                                             \
  }
#else
#define H8_PROF_MARK(i)
#endif

// one global_load_lds_dwordx4: lane l copies the 16 bytes at its own `gsrc` to LDS address `ldst_wave_base + 16 l`
#define SLU_GLDS16(gsrc, ldst_wave_base)                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),                 \
                                   (__attribute__((address_space(3))) void*)(ldst_wave_base), 16, 0, 0)

// -----------------------------------------------------------------------------------------------------------
// Tiled kernel, persistent, LDS-DMA staged.  Workgroup = WM x WN waves; output tile = TH rows x 64 columns x
// (32 WM MB) channels.  Tiles are dealt to the resident workgroups round-robin (tile = w + i * #workgroups, the 32
// workgroups of an XCD side by side along the azimuth), so at any moment the chip works on one compact band of the
// image: neighbouring halos meet in L2 and DRAM sees long contiguous rows.  The unit of staging is a chunk = one
// K-step (16 channels): the input tile (2 channel blocks, halo included) and the weight fragments are copied
// global -> LDS by global_load_lds (no staging registers), into the buffer the previous chunk is not using, piece
// by piece BETWEEN the taps of the current chunk's MFMA phase; the chunk after a tile's last one is the first
// chunk of the NEXT tile, so loads stay in flight across the epilogue.
// One barrier per chunk.  WRES: the weight fragments of ALL K-steps stay in LDS for the whole kernel (small
// layers).  SCALED: per-(image, channel) multipliers (Dropout2d on a concatenated input) are applied to the B
// fragments after the LDS read.
// -----------------------------------------------------------------------------------------------------------
// KPC: K-steps (16 channels each) per chunk = per barrier (2 for the 2x2-dilated 128-channel layers, whose 4 taps per K-step are too
// little MFMA work per barrier).  OPT (bit mask, the 8-wave 128-channel configuration): 1 = waves 4..7 (the SIMD partners of waves
// 0..3) issue a tap's LDS-DMA pieces BEFORE its MFMAs, waves 0..3 after them, so the two waves of a SIMD stop stalling on the
// address pipe at the same moment; 2 = s_setprio 1 around the MFMA cluster; 4 = the next chunk's staging set-up runs before the
// wait + barrier instead of after; 8 = whole 16-byte records per lane on the way out (v_permlane32_swap).
// ONE: the layer has ONE plain source (no concatenation, no batch broadcast): the staging set-up of a chunk is one 64-bit multiply-add on
// the scalar unit instead of the source-selection chains of the general form.
template <int KS, int DIL, int PAD, int MB, int WM, int WN, int RPW, bool SCALED, bool WRES, bool F32OUT, int KPC = 1, int OPT = 0, bool ONE = false>
__global__ __launch_bounds__(64 * WM * WN, (MB * RPW >= 8) ? 1 : ((WM * WN >= 8 || MB >= 2 || RPW >= 2) ? 2 : 3)) void conv_h8_kernel(const H8Args a, const void* __restrict__ resid,
                                                                                                       void* __restrict__ out) {
  constexpr int NWAVE = WM * WN;
  constexpr int T = KS * KS, TS = KPC * T;          // taps per K-step, tap-steps per chunk
  constexpr int TW = 64, TH = WN * RPW, NB = 2 * RPW;
  constexpr int LW = TW + 2 * PAD, LH = TH + 2 * PAD;
  constexpr int REC = LH * LW;                      // records per channel block
  constexpr int MBLK = WM * MB;
  constexpr int NREC_B = KPC * 2 * REC, NBLK_B = (NREC_B + 63) / 64;    // 64-record pieces of the input tile of a chunk
  constexpr int NB_ALLOC = NBLK_B * 64;
  constexpr int NREC_A = MBLK * TS * 64, NBLK_A = MBLK * TS;            // weight fragments of a chunk
  constexpr int NIB = (NBLK_B + NWAVE - 1) / NWAVE, NIA = (NBLK_A + NWAVE - 1) / NWAVE;
  static_assert(KPC == 1 || (!WRES && !SCALED && MB == 2 && WM == 2 && WN == 4), "multi-K-step chunks: the 8-wave 128-channel configuration only");
  static_assert(OPT == 0 || (MB == 2 && WM == 2 && WN == 4 && !F32OUT), "OPT: the 8-wave 128-channel configuration only");
  static_assert(KPC <= 2, "pc_rc carries 2 bits of channel block");
  static_assert(KPC == 1 || ONE, "multi-K-step chunks: single-source layers only");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_epi = reinterpret_cast<float*>(smem);                        // bias | bn_a | bn_b
  uint4* s_scale = reinterpret_cast<uint4*>(s_epi + 3 * MBLK * 32);     // [2][64] fp16 multipliers per channel block (SCALED)
  uint4* s_b = s_scale + (SCALED ? 128 : 0);                            // [2][NB_ALLOC]
  uint4* s_a = s_b + 2 * NB_ALLOC;                                      // WRES: [MBLK][nks][T][64]; else [2][MBLK][KPC][T][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool late_half = wave >= NWAVE / 2;      // waves 4..7 share their SIMDs with waves 0..3
  const int mblk0 = blockIdx.y * MBLK;
  // contiguous run of tiles of this workgroup; workgroups that share an XCD (blockIdx.x % 8) get neighbouring runs
  int t_beg, t_end, t_step = 1;
  {
    const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
    const int w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
    if (a.order) {      // interleaved: at any moment the resident workgroups cover a compact band of the image
      t_step = nwg;
      t_beg = w;
      t_end = w < nt ? w + (int)((nt - w + nwg - 1) / nwg) * nwg : w;
    } else {
      t_beg = (int)(nt * w / nwg);
      t_end = (int)(nt * (w + 1) / nwg);
    }
  }
  t_beg = __builtin_amdgcn_readfirstlane(t_beg);
  t_end = __builtin_amdgcn_readfirstlane(t_end);
  t_step = __builtin_amdgcn_readfirstlane(t_step);
  if (t_beg >= t_end) return;

  if (tid < MBLK * 32) {
    const int co = mblk0 * 32 + tid;
    const bool ok = co < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[co] : 0.0f;
    s_epi[MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_a[co] : 1.0f;
    s_epi[2 * MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_b[co] : 0.0f;
  }

  const int hh = lane >> 5, jj = lane & 31;
  const size_t HW = (size_t)a.H * a.W;
  const float slope_pre = (a.has_act & 3) == 1 ? a.slope : 1.0f;
  const int nks = a.nks;
  const int nchunk = (nks + KPC - 1) / KPC;
  const int a_stride = WRES ? nks * T * 64 : TS * 64;                   // uint4 per channel block in s_a
  const int abase = (wm * MB) * a_stride + lane;
  const int bbase = hh * REC + (wn * RPW) * LW + jj;                    // + (rr + dy)*LW + cb*32 + dx

  // Tile bookkeeping stays on the scalar unit: the position of the first tile comes from one division, every later one from adding the
  // (pre-divided) tile stride with carries.  Integer division runs on the vector ALU even for uniform operands; left to itself hipcc
  // kept the whole per-chunk staging set-up that depends on it in VGPRs (and in scratch, reloaded behind a vmcnt(0) that also drained
  // the DMA queue) -- readfirstlane pins the results to SGPRs.
  struct TilePos { int tx, ty, x0, y0, n, i0, i1, i2; };      // i_s: the image of source s that output image n reads (once per tile, not per chunk)
  auto rfl = [](int v) __attribute__((always_inline)) { return __builtin_amdgcn_readfirstlane(v); };
  const int step_tx = rfl(t_step % a.tiles_x), step_ty = rfl((t_step / a.tiles_x) % a.tiles_y), step_n = rfl(t_step / (a.tiles_x * a.tiles_y));
  auto finish = [&](TilePos& p) __attribute__((always_inline)) {
    p.x0 = p.tx * TW;
    p.y0 = p.ty * TH;
    p.i0 = p.i1 = p.i2 = p.n;
    if constexpr (!ONE) {
      if (a.src[0].nb) p.i0 = rfl(p.n % a.src[0].nb);
      if (a.nsrc > 1 && a.src[1].nb) p.i1 = rfl(p.n % a.src[1].nb);
      if (a.nsrc > 2 && a.src[2].nb) p.i2 = rfl(p.n % a.src[2].nb);
    }
  };
  auto decode = [&](int t) __attribute__((always_inline)) {
    TilePos p;
    p.tx = rfl(t % a.tiles_x);
    t /= a.tiles_x;
    p.ty = rfl(t % a.tiles_y);
    p.n = rfl(t / a.tiles_y);
    finish(p);
    return p;
  };
  auto advance = [&](TilePos p) __attribute__((always_inline)) {      // the tile t_step after p
    p.tx += step_tx;
    if (p.tx >= a.tiles_x) p.tx -= a.tiles_x, p.ty += 1;
    p.ty += step_ty;
    if (p.ty >= a.tiles_y) p.ty -= a.tiles_y, p.n += 1;
    p.n += step_n;
    finish(p);
    return p;
  };
  static_assert(SLU_MAX_SRC == 3, "TilePos carries one image index per source");
  // The address of the zero record, once, in an SGPR pair the compiler cannot rematerialise: left alone it re-loaded the address from the
  // GOT for every piece (s_getpc + s_load_dwordx2 + s_waitcnt lgkmcnt(0): a scalar-memory round trip in every tap's staging slot).
  uintptr_t zero_addr = reinterpret_cast<uintptr_t>(&g_zero_rec);
  asm volatile("" : "+s"(zero_addr));
  uintptr_t trash_addr = reinterpret_cast<uintptr_t>(&g_trash_rec);
  asm volatile("" : "+s"(trash_addr));
  // Per-lane description of the input-tile pieces this wave copies (the same for every chunk and tile): piece i covers
  // records [64 (i NWAVE + wave), +64) of the [2 KPC][LH][LW] tile image; pc_rc = row | col << 8 | block << 16 | inside << 20.
  int pc_rc[NIB], pc_off[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int e = (i * NWAVE + wave) * 64 + lane;
    const int g2 = e / REC;
    const int rem = e - g2 * REC;
    const int r = rem / LW;
    const int c = rem - r * LW;
    pc_rc[i] = r | (c << 8) | ((g2 & (2 * KPC - 1)) << 16) | ((e < NREC_B ? 1 : 0) << 20);
    pc_off[i] = r * a.W + c + (g2 & (2 * KPC - 1)) * (int)HW;      // block g of the chunk starts g planes after block 0 (same source)
  }
  // LDS-DMA of chunk q of tile tp into input buffer `buf` (and, unless WRES, its weight fragments into weight buffer `buf`),
  // in NPIECE pieces per wave: stage_begin fixes the wave-uniform part, stage_piece(i) issues one global_load_lds.  The
  // pieces of chunk c+1 are issued BETWEEN the taps of chunk c (see the tap loops): when all of them came in one burst
  // right after the barrier, the eight waves queued on the CU's single address pipe while the matrix cores idled
  // (measured with -DSLU_H8_PROF: 25 % of the kernel in that burst, another 20 % in the barrier behind it).
  constexpr int NPIECE = NIB + (WRES ? 0 : NIA);
  constexpr int PPT = (NPIECE + TS - 1) / TS;        // pieces issued after each tap-step
  // st_b0: byte address of the record at tile-image position (0, 0) of the chunk's first block; st_b1: the same for its second block
  // MINUS one plane (pc_off carries the plane offset of the block), which equals st_b0 unless the K-step straddles two sources
  uintptr_t st_b0 = 0, st_b1 = 0;
  int st_g0 = 0;                                       // first channel block of the chunk (a block g is live while st_g0 + g < Gin)
  bool st_on = false;
  int st_x0 = 0, st_y0 = 0, st_q = 0, st_wave = wave;
  uint4 *st_db = s_b, *st_da = s_a;
  auto stage_begin = [&](const TilePos& tp, int q, int buf) __attribute__((always_inline)) {
    const int img[SLU_MAX_SRC] = {tp.i0, tp.i1, tp.i2};
    const long long org = (long long)(tp.y0 - PAD) * a.W + (tp.x0 - PAD);
    st_g0 = 2 * KPC * q;
    if constexpr (ONE) {
      st_b0 = reinterpret_cast<uintptr_t>(a.src[0].ptr) + 16 * ((long long)(((size_t)tp.n * a.src[0].G + st_g0) * HW) + org);
      st_b1 = st_b0;
    } else {
      const SrcSel p0 = select_src(a, img, st_g0 < a.Gin ? st_g0 : 0), p1 = select_src(a, img, st_g0 + 1 < a.Gin ? st_g0 + 1 : 0);
      st_b0 = reinterpret_cast<uintptr_t>(p0.ptr) + 16 * ((long long)(((size_t)p0.ns * p0.G + p0.gl) * HW) + org);
      st_b1 = reinterpret_cast<uintptr_t>(p1.ptr) + 16 * ((long long)(((size_t)p1.ns * p1.G + p1.gl) * HW) + org - (long long)HW);
    }
    st_x0 = tp.x0 - PAD;
    st_y0 = tp.y0 - PAD;
    st_q = q;
    st_db = s_b + buf * NB_ALLOC;
    st_da = s_a + buf * NREC_A;
    st_on = true;
    // an opaque copy of the wave number per chunk: otherwise every piece's LDS address and bounds test is hoisted out of the tile loop
    // as a loop invariant, ~100 SGPRs live across it, spilled to VGPR lanes (and, in the two-K-step form, to scratch)
    st_wave = wave;
    asm volatile("" : "+s"(st_wave));
  };
  auto stage_piece = [&](int i) __attribute__((always_inline)) {
    if (i < NIB) {
      const int blk = i * NWAVE + st_wave;
      if (NBLK_B % NWAVE == 0 || blk < NBLK_B) {
        const int rc = pc_rc[i];
        const int gy = st_y0 + (rc & 255), gx = st_x0 + ((rc >> 8) & 255);
        const int gsel = (rc >> 16) & 3;
        const uintptr_t gb = (!ONE && gsel == 1) ? st_b1 : st_b0;
        const bool ok = (rc >> 20) && st_g0 + gsel < a.Gin && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const uintptr_t src = ok ? gb + 16 * (long long)pc_off[i] : zero_addr;
        SLU_GLDS16(reinterpret_cast<const uint4*>(src), st_db + blk * 64);
      }
    } else if constexpr (!WRES) {
      const int blk = (i - NIB) * NWAVE + st_wave;                      // = (m * KPC + j) * T + tap
      if (NBLK_A % NWAVE == 0 || blk < NBLK_A) {
        const int m = blk / TS, r = blk - m * TS;                         // r = j * T + tap: K-step j of the chunk
        const bool live = mblk0 + m < a.nmblk && (KPC == 1 || KPC * st_q + r / T < nks);
        const uint4* src = live ? a.wpack + (((size_t)(mblk0 + m) * nks + KPC * st_q) * T + r) * 64 + lane : reinterpret_cast<const uint4*>(zero_addr);
        SLU_GLDS16(src, st_da + blk * 64);
      }
    }
  };
  // the pieces that go with tap-step `tap` (compile-time indices once the tap loop is unrolled)
  auto stage_after_tap = [&](int tap) __attribute__((always_inline)) {
    if (st_on) {
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        if (tap * PPT + k < NPIECE) stage_piece(tap * PPT + k);
    }
  };
  // per-channel multipliers of image n as fp16, one record per channel block (SCALED)
  auto stage_scales = [&](int n, int par) __attribute__((always_inline)) {
    if (tid < 64) {
      half8 h;
#pragma unroll
      for (int k = 0; k < 8; ++k) h[k] = (_Float16)1.0f;
      if (tid < a.Gin) {
        int img[SLU_MAX_SRC] = {0, 0, 0};
        const SrcSel p = select_src(a, img, tid);
        if (p.scale) {
          const float* sp = p.scale + ((size_t)n * p.G + p.gl) * 8;
          const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
          h[0] = (_Float16)s0.x; h[1] = (_Float16)s0.y; h[2] = (_Float16)s0.z; h[3] = (_Float16)s0.w;
          h[4] = (_Float16)s1.x; h[5] = (_Float16)s1.y; h[6] = (_Float16)s1.z; h[7] = (_Float16)s1.w;
        }
      }
      s_scale[par * 64 + tid] = __builtin_bit_cast(uint4, h);
    }
  };

  if constexpr (WRES) {      // all weight fragments of this channel-block group, once
    const int per_m = nks * T;
    for (int blk = wave; blk < MBLK * per_m; blk += NWAVE) {
      const int m = blk / per_m;
      const uint4* src = mblk0 + m < a.nmblk ? a.wpack + ((size_t)(mblk0 + m) * per_m + (blk - m * per_m)) * 64 + lane : &g_zero_rec;
      SLU_GLDS16(src, s_a + blk * 64);
    }
  }
  TilePos cur = decode(t_beg), nxt = cur;
  stage_begin(cur, 0, 0);
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) stage_piece(i);
  int buf = 0;
  constexpr bool SWAP16 = (OPT & 8) != 0;            // whole 16-byte records per lane on the way out
  constexpr int NST = MB * NB * (SWAP16 ? 2 : 4);    // stores of a tile's epilogue, per wave (h8 output: every lane stores)
  constexpr bool PRE = MB == 1 && !F32OUT;           // residual of the tile prefetched before its last MFMA phase
  const uint2* resid2 = reinterpret_cast<const uint2*>(resid);
  uint2 rv[PRE ? NB : 1][4];
  // the staging set-up of the chunk after (tile, q): the next chunk of this tile, or the first of the next tile
  auto setup_next = [&](int tile, int q) __attribute__((always_inline)) {
    st_on = false;
    if (q + 1 < nchunk) {
      stage_begin(cur, q + 1, buf ^ 1);
    } else if (tile + t_step < t_end) {
      nxt = advance(cur);
      stage_begin(nxt, 0, buf ^ 1);
    }
  };

#ifdef SLU_H8_PROF
  unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
  int tile_no = 0;
  for (int tile = t_beg; tile < t_end; tile += t_step, ++tile_no) {
    f32x16 acc[MB][NB];                                // per tile (not carried around the loop: keeps it in the MFMA registers)
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.0f;
    const int spar = tile_no & 1;
    if constexpr (SCALED) stage_scales(cur.n, spar);
    for (int q = 0; q < nchunk; ++q) {
      // (OPT & 4) every piece of chunk (tile, q) was issued during the previous MFMA phase, so the staging state may move on to the
      // chunk after it while this wave waits for its DMA and for the other waves
      if constexpr ((OPT & 4) != 0) setup_next(tile, q);
      // Chunk (tile, q) has landed and nobody reads the other buffer any more.  vmcnt counts loads, LDS-DMA and stores in
      // issue order: at a tile's first chunk the only operations younger than the DMA we wait for are the NST stores of the
      // previous tile's epilogue, which may stay in flight (waiting for them would expose the HBM write latency per tile).
      if (!F32OUT && q == 0 && tile != t_beg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST < 63 ? NST : 63) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      H8_PROF_MARK(0)                                    // waiting for the chunk's DMA (and, at q = 0, the epilogue before it)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      H8_PROF_MARK(1)                                    // barrier
      if constexpr ((OPT & 4) == 0) setup_next(tile, q);
      if constexpr (PRE) {
        if (resid && q == nchunk - 1) {
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj;
            const bool pix_ok = gy < a.H && gx < a.W;
            const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int go = (mblk0 + wm) * 4 + k;
              rv[b][k] = *((pix_ok && go < a.Gout) ? resid2 + (((size_t)cur.n * a.Gout + go) * HW + pix) * 2 + hh
                                                   : reinterpret_cast<const uint2*>(zero_addr));
            }
          }
        }
      }
      H8_PROF_MARK(2)                                    // issuing the next chunk's DMA (+ residual prefetch)
      const uint4* sb = s_b + buf * NB_ALLOC + bbase;
      const uint4* sa = s_a + abase + (WRES ? q * T * 64 : buf * NREC_A);
      half8 sc;
      if constexpr (SCALED) sc = __builtin_bit_cast(half8, s_scale[spar * 64 + 2 * q + hh]);
      {
        if constexpr (MB == 2 && WM == 2 && WN == 4) {
          // the 8-wave 128-channel configuration (MFMA-bound layers): fragments of tap-step t+1 are read before the MFMAs of tap-step t
          // issue, in THIS order -- sched_barrier pins it; with sched_group_barrier hints (below) the compiler still emits read, wait,
          // MFMA, read, ... (+3 % on these layers; the other configurations spill with a second fragment set)
          half8 af[2][MB], bf[2][NB];
          auto read_frags = [&](int set, int ts) __attribute__((always_inline)) {
            const int j = ts / T, tap = ts % T;          // K-step of the chunk, tap
            const int dy = (tap / KS) * DIL, dx = (tap % KS) * DIL;
#ifdef SLU_H8_FAKE_LDS      // experiment (WRONG results): a quarter of the fragment reads, the same MFMAs -- does LDS traffic limit these layers?
            af[set][0] = __builtin_bit_cast(half8, sa[ts * 64]);
            bf[set][0] = __builtin_bit_cast(half8, sb[j * 2 * REC + dy * LW + dx]);
#pragma unroll
            for (int i = 1; i < MB; ++i) af[set][i] = af[set][0];
#pragma unroll
            for (int b = 1; b < NB; ++b) bf[set][b] = bf[set][0];
#else
#pragma unroll
            for (int i = 0; i < MB; ++i) af[set][i] = __builtin_bit_cast(half8, sa[i * a_stride + ts * 64]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              bf[set][b] = __builtin_bit_cast(half8, sb[j * 2 * REC + ((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
              if constexpr (SCALED) bf[set][b] *= sc;
            }
#endif
          };
          read_frags(0, 0);
#pragma unroll
          for (int ts = 0; ts < TS; ++ts) {
            if (ts + 1 < TS) read_frags((ts + 1) & 1, ts + 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((OPT & 1) != 0) {
              if (late_half) stage_after_tap(ts);
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr ((OPT & 2) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ts & 1][i], bf[ts & 1][b], acc[i][b], 0, 0, 0);
            if constexpr ((OPT & 2) != 0) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((OPT & 1) != 0) {
              if (!late_half) stage_after_tap(ts);
            } else {
              stage_after_tap(ts);
            }
          }
        } else if constexpr (MB == 1) {
          // fragments of tap t+1 are read from LDS while the MFMAs of tap t issue (two register sets, one DS read per MFMA
          // slot); with MB = 2 the second set does not fit in 256 VGPRs next to the 128 accumulator registers
          half8 af[2][MB], bf[2][NB];
          auto read_frags = [&](int set, int tap) __attribute__((always_inline)) {
            const int dy = (tap / KS) * DIL, dx = (tap % KS) * DIL;
#pragma unroll
            for (int i = 0; i < MB; ++i) af[set][i] = __builtin_bit_cast(half8, sa[i * a_stride + tap * 64]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              bf[set][b] = __builtin_bit_cast(half8, sb[((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
              if constexpr (SCALED) bf[set][b] *= sc;
            }
          };
          read_frags(0, 0);
#pragma unroll
          for (int tap = 0; tap < T; ++tap) {
            if (tap + 1 < T) read_frags((tap + 1) & 1, tap + 1);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[tap & 1][i], bf[tap & 1][b], acc[i][b], 0, 0, 0);
            if (tap + 1 < T) {
#pragma unroll
              for (int k = 0; k < MB * NB; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, (MB + NB + MB * NB - 1) / (MB * NB), 0);
              }
            }
            stage_after_tap(tap);
          }
        } else {
#pragma unroll
          for (int tap = 0; tap < T; ++tap) {
            const int dy = (tap / KS) * DIL, dx = (tap % KS) * DIL;
            half8 af[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = __builtin_bit_cast(half8, sa[i * a_stride + tap * 64]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              half8 bf = __builtin_bit_cast(half8, sb[((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
              if constexpr (SCALED) bf *= sc;
#pragma unroll
              for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf, acc[i][b], 0, 0, 0);
            }
            // ask for an MFMA / LDS-read interleave: each tap's fragment reads are spread between the previous tap's MFMAs
#pragma unroll
            for (int k = 0; k < MB * NB; ++k) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, (MB + NB + MB * NB - 1) / (MB * NB), 0);
            }
            stage_after_tap(tap);
          }
        }
      }
      buf ^= 1;
      H8_PROF_MARK(3)                                    // LDS reads + MFMAs of the chunk
    }
    {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int ml = wm * MB + i;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj;
          const bool pix_ok = gy < a.H && gx < a.W;
          const size_t pix = pix_ok ? (size_t)gy * a.W + gx : 0;
          if constexpr (F32OUT)
            store_tile<MBLK * 32>(a, acc[i][b], s_epi, ml * 32, (mblk0 + ml) * 32, hh, pix_ok, (size_t)cur.n, pix, HW, resid, out, slope_pre);
          else if constexpr (SWAP16)
            store_tile_swap16<MBLK * 32>(a, acc[i][b], s_epi, ml * 32, (mblk0 + ml) * 4, hh, pix_ok, (size_t)cur.n, pix, HW, resid2,
                                         reinterpret_cast<uint4*>(out), slope_pre, zero_addr, trash_addr);
          else
            store_tile_full<MBLK * 32, PRE>(a, acc[i][b], s_epi, ml * 32, (mblk0 + ml) * 4, hh, pix_ok, (size_t)cur.n, pix, HW, resid2,
                                            rv[PRE ? b : 0], reinterpret_cast<uint2*>(out), slope_pre, zero_addr, trash_addr);
          __builtin_amdgcn_sched_barrier(0);     // one accumulator tile at a time (register pressure)
        }
      }
    }
    H8_PROF_MARK(4)                                      // epilogue
    cur = nxt;
  }
#ifdef SLU_H8_PROF
  if (tid == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_h8_prof[i], prof_acc[i]);
    atomicAdd(&g_h8_prof[5], 1ull);
  }
#endif
}

// -----------------------------------------------------------------------------------------------------------
// 1x1 convs, streaming: a wave owns NBW blocks of 32 consecutive pixels and ALL output channels (MB blocks of 32).
// Its B operands are 16-byte global loads (lane (r, h): block 2k+h of pixel r; per wave two 512-byte runs), the
// input is read exactly once, the output written once; only the weight fragments go through LDS.
// Needs H*W % 32 == 0 and no multipliers.
// -----------------------------------------------------------------------------------------------------------
template <int MB, int NBW>
__global__ __launch_bounds__(256, (MB * NBW >= 8) ? 2 : ((MB * NBW >= 4) ? 3 : 4)) void conv1x1_h8_kernel(const H8Args a, const void* __restrict__ resid,
                                                                                                         void* __restrict__ out) {
  constexpr int KSPC = 4;                               // K-steps (16 channels each) per weight chunk
  constexpr int NWV = MB * KSPC * 64, NW = (NWV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_a = reinterpret_cast<uint4*>(smem);          // [MB][KSPC][64]
  float* s_epi = reinterpret_cast<float*>(s_a + MB * KSPC * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  const long long HW = (long long)a.H * a.W;
  const long long nblocks = (long long)a.N * HW / 32;
  const long long pb0 = ((long long)blockIdx.x * 4 + wave) * NBW;

  if (tid < MB * 32) {
    const bool ok = tid < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[tid] : 0.0f;
    s_epi[MB * 32 + tid] = (ok && a.bn_a) ? a.bn_a[tid] : 1.0f;
    s_epi[2 * MB * 32 + tid] = (ok && a.bn_a) ? a.bn_b[tid] : 0.0f;
  }

  f32x16 acc[MB][NBW];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.0f;

  int nimg[NBW];
  long long hw[NBW];
  bool live[NBW];
  int simg[NBW][SLU_MAX_SRC];
#pragma unroll
  for (int b = 0; b < NBW; ++b) {
    const long long pb = pb0 + b;
    live[b] = pb < nblocks;
    const long long pix = (live[b] ? pb : 0) * 32 + jj;
    nimg[b] = (int)(pix / HW);
    hw[b] = pix - nimg[b] * HW;
#pragma unroll
    for (int s = 0; s < SLU_MAX_SRC; ++s) simg[b][s] = (s < a.nsrc && a.src[s].nb) ? nimg[b] % a.src[s].nb : nimg[b];
  }

  const int nq = (a.nks + KSPC - 1) / KSPC;
#ifdef SLU_H8_PROF      // phases: input loads issued + first barrier | weight staging | second barrier | MFMAs (incl. waiting for the inputs) | epilogue
  unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif
  for (int q = 0; q < nq; ++q) {
    uint4 x[KSPC][NBW];
#pragma unroll
    for (int ks = 0; ks < KSPC; ++ks) {
      const int g = 2 * (q * KSPC + ks) + hh;           // this lane half's channel block
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        const bool ok = live[b] && g < a.Gin;
        const SrcSel p = select_src(a, simg[b], ok ? g : 0);
        const size_t idx = ok ? ((size_t)p.ns * p.G + p.gl) * HW + hw[b] : 0;
        const uint4 v = p.ptr[idx];
        x[ks][b] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      }
    }
    __syncthreads();
    H8_PROF_MARK(0)
    uint4 sw[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 256;
      const int m = e / (KSPC * 64);
      const int r = e - m * (KSPC * 64);
      const int ks = r >> 6;
      const bool ok = (NWV % 256 == 0 || e < NWV) && m < a.nmblk && q * KSPC + ks < a.nks;
      const size_t off = ok ? ((size_t)m * a.nks + q * KSPC) * 64 + r : 0;
      sw[i] = a.wpack[off];
      if (!ok) sw[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 256;
      if (NWV % 256 == 0 || e < NWV) s_a[e] = sw[i];
    }
#ifdef SLU_H8_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    H8_PROF_MARK(1)
    __syncthreads();
    H8_PROF_MARK(2)
#pragma unroll
    for (int ks = 0; ks < KSPC; ++ks)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const half8 af = __builtin_bit_cast(half8, s_a[(i * KSPC + ks) * 64 + lane]);
#pragma unroll
        for (int b = 0; b < NBW; ++b)
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(half8, x[ks][b]), acc[i][b], 0, 0, 0);
      }
#ifdef SLU_H8_PROF
    asm volatile("s_nop 0" ::"v"(acc[MB - 1][NBW - 1][0]));      // the last MFMA has retired
#endif
    H8_PROF_MARK(3)
  }

  const float slope_pre = (a.has_act & 3) == 1 ? a.slope : 1.0f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int b = 0; b < NBW; ++b)
      store_tile<MB * 32>(a, acc[i][b], s_epi, i * 32, i * 32, hh, live[b], (size_t)nimg[b], (size_t)hw[b], (size_t)HW, resid, out, slope_pre);
#ifdef SLU_H8_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  H8_PROF_MARK(4)
  if (tid == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_h8_prof[i], prof_acc[i]);
    atomicAdd(&g_h8_prof[5], 1ull);
  }
#endif
}

// -----------------------------------------------------------------------------------------------------------
// 1x1 convs with few output channels (MB <= 2 blocks of 32) and few input channels (NKS K-steps, all of them resident
// in LDS as weight fragments): pure streaming.  No barrier after the prologue: a wave takes one block of 32 pixels at
// a time (grid-stride, so the chip sweeps memory as one front), issues ALL its NKS input loads (16 B per lane each, two
// 512-byte runs per wave) and the residual loads at once, and multiplies as they arrive.  Memory-level parallelism
// comes from occupancy (4-5 waves per SIMD, up to NKS KB in flight per wave); LDS only serves the A fragments.
// -----------------------------------------------------------------------------------------------------------
template <int MB, int NKS>
__global__ __launch_bounds__(256, (MB * NKS >= 12) ? 2 : ((MB * NKS >= 2) ? 3 : 4)) void conv1x1_h8_res_kernel(const H8Args a, const void* __restrict__ resid, void* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_epi = reinterpret_cast<float*>(smem);                 // bias | bn_a | bn_b
  uint4* s_a = reinterpret_cast<uint4*>(s_epi + 3 * MB * 32);    // [MB][NKS][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, jj = lane & 31;
  const long long HW = (long long)a.H * a.W;
  const long long nblocks = (long long)a.N * HW / 32;

  if (tid < MB * 32) {
    const bool ok = tid < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[tid] : 0.0f;
    s_epi[MB * 32 + tid] = (ok && a.bn_a) ? a.bn_a[tid] : 1.0f;
    s_epi[2 * MB * 32 + tid] = (ok && a.bn_a) ? a.bn_b[tid] : 0.0f;
  }
  for (int e = tid; e < MB * NKS * 64; e += 256) {
    const int m = e / (NKS * 64);
    const int r = e - m * (NKS * 64);
    const bool ok = m < a.nmblk && (r >> 6) < a.nks;
    s_a[e] = ok ? a.wpack[(size_t)m * a.nks * 64 + r] : make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  const float slope_pre = (a.has_act & 3) == 1 ? a.slope : 1.0f;
  const uint2* resid2 = reinterpret_cast<const uint2*>(resid);
  const long long stride = (long long)gridDim.x * 4;
  for (long long pb = (long long)blockIdx.x * 4 + wave; pb < nblocks; pb += stride) {
    const long long pix = pb * 32 + jj;
    const int n = (int)(pix / HW);
    const size_t hw = (size_t)(pix - n * HW);
    int img[SLU_MAX_SRC];
#pragma unroll
    for (int s = 0; s < SLU_MAX_SRC; ++s) img[s] = (s < a.nsrc && a.src[s].nb) ? n % a.src[s].nb : n;
    uint4 x[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int g = 2 * ks + hh;                        // this lane half's channel block
      const bool ok = g < a.Gin;
      const SrcSel p = select_src(a, img, ok ? g : 0);
      const uint4 v = p.ptr[ok ? ((size_t)p.ns * p.G + p.gl) * HW + hw : 0];
      x[ks] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
    uint2 rv[MB][4];
    if (resid) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int go = i * 4 + q;
          rv[i][q] = go < a.Gout ? resid2[(((size_t)n * a.Gout + go) * HW + hw) * 2 + hh] : make_uint2(0u, 0u);
        }
    }
    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int i = 0; i < MB; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, s_a[(i * NKS + ks) * 64 + lane]), __builtin_bit_cast(half8, x[ks]),
                                                        acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      if (a.out_f32) {
        store_tile<MB * 32>(a, acc[i], s_epi, i * 32, i * 32, hh, true, (size_t)n, hw, (size_t)HW, nullptr, out, slope_pre);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cl = i * 32 + 8 * q + 4 * hh + k;
            float t = acc[i][4 * q + k] + s_epi[cl];
            t = t > 0.0f ? t : t * slope_pre;
            v[k] = t * s_epi[MB * 32 + cl] + s_epi[2 * MB * 32 + cl];
          }
          if (resid) {
            const half2v r0 = __builtin_bit_cast(half2v, rv[i][q].x), r1 = __builtin_bit_cast(half2v, rv[i][q].y);
            v[0] += (float)r0[0]; v[1] += (float)r0[1]; v[2] += (float)r1[0]; v[3] += (float)r1[1];
          }
          const int go = i * 4 + q;
          if (go < a.Gout)
            reinterpret_cast<uint2*>(out)[(((size_t)n * a.Gout + go) * HW + hw) * 2 + hh] = make_uint2(pack2(v[0], v[1]), pack2(v[2], v[3]));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------------
// Wide 1x1 convs with 256 output channels (768 -> 256 concat convs and the 128 / 256 -> 256 shortcuts of the U-Net's lower levels) as a plain
// GEMM, out[Cout][pixels] = W[Cout][Cin] x[Cin][pixels]: BOTH operands go global -> LDS by global_load_lds into a ring of D chunks of KC K-steps,
// D - 1 in flight, issued as one burst behind the barrier that frees the slot; one barrier per chunk; every wave issues the same number of DMAs
// at every position (beyond the end: the zero record), so "chunk c has landed" is the constant vmcnt((D - 2) NPIECE).  A workgroup of 8 waves
// (2 along the channels x 4 along the pixels) owns 256 consecutive pixels of one image plane and all channels: the weights stream once per 256
// pixels (the streaming conv1x1_h8_kernel re-reads them per 128 and stages them through registers + two barriers per 64 channels).
// Measured (N = 64, tools/h8_1x1_bench.py): 768 -> 256 at 16x512 504 -> 459 us, 128 -> 256 at 16x512 117 -> 96, 256 -> 256 at 8x256 46 -> 38; the
// 128-output instantiation is 17 % SLOWER than the streaming kernel (544 -> 634 us at 32x1024) and is not dispatched.  What the ablation
// switches (H8Args.dbg / SLU_GEMM_DBG) showed on 768 -> 256: no input DMA 399 us, no weight DMA 413, no MFMA 354, none of the three STILL 306 of
// 481 -- the time is in the per-chunk skeleton (32 DMA instructions per CU and chunk, barrier, fragment reads) and the epilogue, not in HBM,
// L2 or the matrix cores; a deeper ring of smaller chunks (KC 2, D 4) was slower (484) than two 64-channel chunks (459).
// With a residual the epilogue's loads drain the ring once per tile (the compiler's vmcnt(0)); without, the stores stay in flight.
// Needs: Cout = 64 MB, H W % 256 == 0, nks % KC == 0, every source a whole number of chunks, no multipliers / batch broadcast.
// -----------------------------------------------------------------------------------------------------------
template <int MB, int KC, int D>
__global__ __launch_bounds__(512, 2) void gemm1x1_h8_kernel(const H8Args a, const void* __restrict__ resid, void* __restrict__ out) {
  constexpr int NWAVE = 8, MBLK = 2 * MB, TP = 256, NB = 2;
  constexpr int NREC_A = MBLK * KC * 64, NREC_B = 2 * KC * TP;           // records per chunk: weight fragments, input tile [2 KC blocks][256 px]
  constexpr int NPA = MBLK * KC, NPB = 2 * KC * 4;                       // 64-record pieces
  constexpr int NIA = (NPA + NWAVE - 1) / NWAVE, NIB = NPB / NWAVE;
  constexpr int NPIECE = NIA + NIB;
  static_assert(NPB % NWAVE == 0 && NPA % NWAVE == 0 && D >= 2 && (D - 2) * NPIECE <= 63, "pieces divide over the waves; vmcnt is 6 bits");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_epi = reinterpret_cast<float*>(smem);                         // bias | bn_a | bn_b
  uint4* s_b = reinterpret_cast<uint4*>(s_epi + 3 * MBLK * 32);          // [D][NREC_B]
  uint4* s_a = s_b + D * NREC_B;                                         // [D][MBLK][KC][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, hh = lane >> 5, jj = lane & 31;
  const size_t HW = (size_t)a.H * a.W;
  const int tiles_per_img = (int)(HW / TP);
  const int ntiles = tiles_per_img * a.N, nch = a.nks / KC;
  const int t_step = gridDim.x, t_beg = (int)blockIdx.x;
  if (t_beg >= ntiles) return;

  if (tid < MBLK * 32) {
    const bool ok = tid < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[tid] : 0.0f;
    s_epi[MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_a[tid] : 1.0f;
    s_epi[2 * MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_b[tid] : 0.0f;
  }
  uintptr_t zero_addr = reinterpret_cast<uintptr_t>(&g_zero_rec), trash_addr = reinterpret_cast<uintptr_t>(&g_trash_rec);
  asm volatile("" : "+s"(zero_addr));
  asm volatile("" : "+s"(trash_addr));

  // the staging cursor runs D - 1 chunks ahead of the compute cursor: (s_tile, s_q) = the next chunk to copy, into ring slot s_slot
  int s_tile = t_beg, s_q = 0, s_slot = 0;
  auto stage_next = [&]() __attribute__((always_inline)) {
    const bool valid = s_tile < ntiles;
    int wv = wave;
    asm volatile("" : "+s"(wv));                                        // (keeps the per-piece address arithmetic out of the loop-invariant set)
    uintptr_t base = zero_addr;
    if (valid) {
      const int n = s_tile / tiles_per_img, p0 = (s_tile - n * tiles_per_img) * TP;
      const int g = 2 * KC * s_q;                                        // first channel block of the chunk: one source holds the whole chunk
      const uint4* ptr = a.src[0].ptr;
      int G = a.src[0].G, gl = g;
#pragma unroll
      for (int s = 1; s < SLU_MAX_SRC; ++s)
        if (s < a.nsrc && g >= a.src[s].gbeg) ptr = a.src[s].ptr, G = a.src[s].G, gl = g - a.src[s].gbeg;
      base = reinterpret_cast<uintptr_t>(ptr) + 16 * (((size_t)n * G + gl) * HW + p0);
    }
    uint4* db = s_b + s_slot * NREC_B;
    uint4* da = s_a + s_slot * NREC_A;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int p = i * NWAVE + wv;                                      // block = p / 4, pixel quarter = p % 4
      const uintptr_t src = (valid && !(a.dbg & 1)) ? base + 16 * ((size_t)(p >> 2) * HW + (size_t)((p & 3) * 64 + lane)) : zero_addr;
      SLU_GLDS16(reinterpret_cast<const uint4*>(src), db + p * 64);
    }
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int p = i * NWAVE + wv;                                      // = m * KC + ks
      const int m = p / KC, ks = p - m * KC;
      const uint4* src = (valid && !(a.dbg & 2)) ? a.wpack + ((size_t)m * a.nks + KC * s_q + ks) * 64 + lane : reinterpret_cast<const uint4*>(zero_addr);
      SLU_GLDS16(src, da + p * 64);
    }
    asm volatile("" ::: "memory");
    s_slot = s_slot + 1 == D ? 0 : s_slot + 1;
    if (++s_q == nch) s_q = 0, s_tile += t_step;
  };
#pragma unroll
  for (int c = 0; c < D - 1; ++c) stage_next();

  int r_slot = 0;
  const float slope_pre = (a.has_act & 3) == 1 ? a.slope : 1.0f;
  const uint2* resid2 = reinterpret_cast<const uint2*>(resid);
  const int abase = (wm * MB) * KC * 64 + lane;
  const int bbase = hh * TP + wn * 64 + jj;

  for (int tile = t_beg; tile < ntiles; tile += t_step) {
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.0f;
    for (int q = 0; q < nch; ++q) {
      // the oldest chunk in flight has landed once at most the (D - 2) younger chunks' pieces are outstanding (stores of an epilogue in
      // between only make the wait stricter); then every wave is past its reads of the slot that is re-filled next
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NPIECE) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      stage_next();
      const uint4* sb = s_b + r_slot * NREC_B + bbase;
      const uint4* sa = s_a + r_slot * NREC_A + abase;
      r_slot = r_slot + 1 == D ? 0 : r_slot + 1;
      half8 af[2][MB], bf[2][NB];
      auto read_frags = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MB; ++i) af[set][i] = __builtin_bit_cast(half8, sa[(i * KC + ks) * 64]);
#pragma unroll
        for (int b = 0; b < NB; ++b) bf[set][b] = __builtin_bit_cast(half8, sb[2 * ks * TP + b * 32]);
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        if (ks + 1 < KC) read_frags((ks + 1) & 1, ks + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.dbg & 4)) {
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[ks & 1][i], bf[ks & 1][b], acc[i][b], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < MB; ++i) asm volatile("" ::"v"(af[ks & 1][i]));
#pragma unroll
          for (int b = 0; b < NB; ++b) asm volatile("" ::"v"(bf[ks & 1][b]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      const int n = tile / tiles_per_img, p0 = (tile - n * tiles_per_img) * TP;
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int ml = wm * MB + i;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const size_t pix = (size_t)p0 + wn * 64 + b * 32 + jj;
          store_tile_swap16<MBLK * 32>(a, acc[i][b], s_epi, ml * 32, ml * 4, hh, true, (size_t)n, pix, HW, resid2, reinterpret_cast<uint4*>(out), slope_pre,
                                       zero_addr, trash_addr);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // zero-record DMAs issued beyond the last chunk: LDS must not be released under them
}

// wpack[mblk][kstep][tap][lane][8]: lane (r, h) holds W[co = 32 mblk + r][ci = 16 kstep + 8 h + j][tap], j = 0..7, as fp16
__global__ void pack_h8_kernel(const float* __restrict__ w, int cout, int cin, int ks, int nks, size_t total, uint4* __restrict__ out) {
  const int T = ks * ks;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    size_t r = e >> 6;
    const int tap = (int)(r % T);
    r /= T;
    const int k = (int)(r % nks);
    const int m = (int)(r / nks);
    const int co = m * 32 + (lane & 31);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = k * 16 + 8 * (lane >> 5) + j;
      x[j] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * T + tap] : 0.0f;
    }
    out[e] = make_uint4(pack2(x[0], x[1]), pack2(x[2], x[3]), pack2(x[4], x[5]), pack2(x[6], x[7]));
  }
}

// ---- layout / pooling helpers -------------------------------------------------------------------------------
// fp32 NCHW [N][C][H][W] -> h8 [N][ceil(C/8)][H][W][8] (pad channels = 0), optional per-(n, c) multiplier
__global__ __launch_bounds__(256) void nchw_to_h8_kernel(const float* __restrict__ x, const float* __restrict__ scale, uint4* __restrict__ y, int N,
                                                         int C, int G, size_t HW) {
  const size_t total = (size_t)N * G * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = e % HW;
    const size_t ng = e / HW;
    const int g = (int)(ng % G);
    const size_t n = ng / G;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      v[k] = c < C ? x[(n * C + c) * HW + pix] * (scale ? scale[n * C + c] : 1.0f) : 0.0f;
    }
    y[e] = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
  }
}

__global__ __launch_bounds__(256) void h8_to_nchw_kernel(const uint4* __restrict__ x, float* __restrict__ y, int N, int C, int G, size_t HW) {
  const size_t total = (size_t)N * G * HW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = e % HW;
    const size_t ng = e / HW;
    const int g = (int)(ng % G);
    const size_t n = ng / G;
    const half8 h = __builtin_bit_cast(half8, x[e]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = g * 8 + k;
      if (c < C) y[(n * C + c) * HW + pix] = (float)h[k];
    }
  }
}

// AvgPool2d(3, stride 2, pad 1, count_include_pad) of x * scale[n, c]; x may hold `in_batch` images shared by all n
__global__ __launch_bounds__(256) void avgpool3s2_h8_kernel(const uint4* __restrict__ x, const float* __restrict__ scale, uint4* __restrict__ y,
                                                            int N, int G, int H, int W, int OH, int OW, int in_batch) {
  const size_t total = (size_t)N * G * OH * OW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % OW);
    size_t r = e / OW;
    const int oy = (int)(r % OH);
    r /= OH;
    const int g = (int)(r % G);
    const size_t n = r / G;
    const size_t ni = in_batch ? n % in_batch : n;
    const uint4* p = x + (ni * G + g) * (size_t)H * W;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      const int iy = 2 * oy + i;
#pragma unroll
      for (int j = -1; j <= 1; ++j) {
        const int ix = 2 * ox + j;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const half8 h = __builtin_bit_cast(half8, p[ok ? (size_t)iy * W + ix : 0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += ok ? (float)h[k] : 0.0f;
      }
    }
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = scale ? scale[(n * G + g) * 8 + k] : 1.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] * s[k] / 9.0f;
    y[e] = make_uint4(pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7]));
  }
}

// y[n, c, 2h+i, 2w+j] = x[n, 4c+2i+j, h, w] * s_in[n, 4c+2i+j] * s_out[n, c]   (nn.PixelShuffle(2) + both Dropout2d multipliers)
// x: h8 with Gi blocks at HxW; y: h8 with Go = ceil(2 Gi / 8) blocks at 2Hx2W.  One thread takes the four input records (blocks
// 4 go .. 4 go + 3) of one input pixel -- 32 stored channels = 8 output channels x 4 sub-pixels -- and writes the four complete
// output records of the 2x2 output patch: 16-byte loads and stores only, consecutive threads = consecutive azimuth.
__global__ __launch_bounds__(256) void pixel_shuffle_h8_kernel(const uint4* __restrict__ x, const float* __restrict__ s_in,
                                                               const float* __restrict__ s_out, uint4* __restrict__ y, int N, int Gi, int Go, int H,
                                                               int W) {
  const size_t total = (size_t)N * Go * H * W;
  const size_t HWi = (size_t)H * W;
  const int OW = 2 * W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int w = (int)(e % W);
    size_t r = e / W;
    const int h = (int)(r % H);
    r /= H;
    const int go = (int)(r % Go);
    const size_t n = r / Go;
    // v[sub][k]: output channel 8 go + k at sub-pixel sub = 2 i + j  <-  stored channel 32 go + 4 k + sub = block 4 go + (k >> 1), slot 4 (k & 1) + sub
    float v[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int gi = 4 * go + q;
      const bool ok = gi < Gi;
      const half8 rec = __builtin_bit_cast(half8, ok ? x[(n * Gi + gi) * HWi + (size_t)h * W + w] : make_uint4(0u, 0u, 0u, 0u));
#pragma unroll
      for (int slot = 0; slot < 8; ++slot) {
        float t = (float)rec[slot];
        if (ok && s_in) t *= s_in[n * Gi * 8 + gi * 8 + slot];
        v[slot & 3][2 * q + (slot >> 2)] = t;
      }
    }
    if (s_out) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int co = 8 * go + k;
        const float so = co < 2 * Gi ? s_out[n * (Gi * 2) + co] : 0.0f;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) v[sub][k] *= so;
      }
    }
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const size_t o = ((n * Go + go) * (size_t)(2 * H) + (2 * h + (sub >> 1))) * OW + 2 * w + (sub & 1);
      y[o] = make_uint4(pack2(v[sub][0], v[sub][1]), pack2(v[sub][2], v[sub][3]), pack2(v[sub][4], v[sub][5]), pack2(v[sub][6], v[sub][7]));
    }
  }
}

inline unsigned grid_for(size_t total) {
  const size_t nb = (total + 255) / 256;
  return (unsigned)(nb > 32768 ? 32768 : (nb ? nb : 1));
}

int fill_h8(const slu_conv_h8_desc* d, H8Args& a) {
  if (!d || !d->out || !d->wpack || d->nsrc < 1 || d->nsrc > SLU_MAX_SRC) return SLU_EINVAL;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0) return SLU_EINVAL;
  if (d->bn_a && !d->bn_b) return SLU_EINVAL;
  if (d->has_act != 0 && d->has_act != 1) return SLU_EINVAL;
  if (d->has_act && !(d->slope >= 0.0f && d->slope <= 1.0f)) return SLU_EINVAL;      // LeakyReLU is evaluated as max(v, slope v)
  int g = 0;
  for (int s = 0; s < d->nsrc; ++s) {
    const slu_h8_src& S = d->src[s];
    if (!S.ptr || S.G <= 0 || S.nbatch < 0) return SLU_EINVAL;
    if (((uintptr_t)S.ptr & 15) || ((uintptr_t)S.scale & 15)) return SLU_EINVAL;
    a.src[s] = H8Src{reinterpret_cast<const uint4*>(S.ptr), S.scale, S.G, g, S.nbatch};
    g += S.G;
  }
  for (int s = d->nsrc; s < SLU_MAX_SRC; ++s) a.src[s] = H8Src{nullptr, nullptr, 0, 0x7fffffff, 0};
  if (((uintptr_t)d->out & 15) || ((uintptr_t)d->resid & 15) || ((uintptr_t)d->wpack & 15)) return SLU_EINVAL;
  if (d->resid && d->out_f32_nchw) return SLU_EINVAL;
  a.nsrc = d->nsrc;
  a.N = d->N; a.H = d->H; a.W = d->W;
  a.Gin = g;
  a.Cout = d->Cout;
  a.Gout = (d->Cout + 7) / 8;
  a.nmblk = (d->Cout + 31) / 32;
  a.nks = (g + 1) / 2;
  a.wpack = reinterpret_cast<const uint4*>(d->wpack);
  a.bias = d->bias; a.bn_a = d->bn_a; a.bn_b = d->bn_b;
  a.has_act = d->has_act; a.slope = d->slope;
  a.out_f32 = d->out_f32_nchw ? 1 : 0;
  a.tiles_x = a.tiles_y = 0;
  a.dbg = [] { const char* e = getenv("SLU_GEMM_DBG"); return e ? atoi(e) : 0; }();
  return SLU_OK;
}

template <int KS, int DIL, int PAD, int MB, int WM, int WN, int RPW, bool SCALED, bool WRES, bool F32OUT = false, int KPC = 1, int OPT = 0, bool ONE = false>
int launch_h8_k(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  constexpr int TH = WN * RPW, MBLK = WM * MB, T = KS * KS, NWAVE = WM * WN;
  constexpr int WAVES_PER_SIMD = (MB * RPW >= 8) ? 1 : ((NWAVE >= 8 || MB >= 2 || RPW >= 2) ? 2 : 3);       // the kernel's __launch_bounds__
  constexpr size_t nb_alloc = (size_t)((KPC * 2 * (TH + 2 * PAD) * (64 + 2 * PAD) + 63) / 64) * 64;
  const size_t lds = (size_t)3 * MBLK * 32 * 4 + (SCALED ? 2048 : 0) + 2 * nb_alloc * 16 + (size_t)MBLK * (WRES ? a.nks : 2 * KPC) * T * 64 * 16;
  if (lds > 160 * 1024) return SLU_EUNSUPPORTED;
  if (SCALED && a.Gin > 64) return SLU_EUNSUPPORTED;
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  static const int order = [] { const char* e = getenv("SLU_H8_ORDER"); return e ? atoi(e) : 1; }();      // 0 is kept for A/B runs
  a.order = order;
  const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
  const int gy = (a.nmblk + MBLK - 1) / MBLK;
  if (nt <= 0 || nt > 0x7fffffffLL || gy > 65535) return SLU_EUNSUPPORTED;
  // persistent grid: as many workgroups as fit on the 256 CUs at once (registers / LDS), never more than tiles
  long long per_cu = WAVES_PER_SIMD * 4 / NWAVE;
  const long long by_lds = (long long)(160 * 1024 / lds);
  if (by_lds < per_cu) per_cu = by_lds;
  if (per_cu < 1) per_cu = 1;
  long long gx = (256 * per_cu + gy - 1) / gy;
  if (gx < 8) gx = 8;
  if (gx > nt) gx = nt;
  auto kern = conv_h8_kernel<KS, DIL, PAD, MB, WM, WN, RPW, SCALED, WRES, F32OUT, KPC, OPT, ONE>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64 * NWAVE), lds, st, a, d->resid, d->out);
  SLU_CHECK_LAUNCH();
}

// -----------------------------------------------------------------------------------------------------------
// 3x3 convs of the full-resolution layers with 32 / 64 input and output channels, ONE plain source: the deep-ring form.
// conv_h8_kernel keeps one 16-channel chunk (~21-43 KB) of DMA in flight per CU while it multiplies the previous one; at
// 5.5 TB/s x ~2 us of loaded latency a CU needs ~43 KB in flight ALL the time, and these layers have too little MFMA work
// per chunk to cover one round trip.  Here (the structure of tail2_h8_kernel, conv_tail_h8.hip):
//   * all weights resident in LDS for the life of the persistent workgroup (<= 72 KB);
//   * the rest of the LDS is a ring of D input chunks (16 channels with halo), D - 1 of them in flight;
//   * one wave owns ALL output channels of its pixels (MB blocks) and RPW rows; fragments double-buffered in source order;
//   * counted vmcnt waits: every wave issues the same VM operations at every position of every tile (surplus DMA slots
//     copy the zero record to a trash block, beyond the last tile whole chunks do);
//   * whole 16-byte records per lane on the way out (v_permlane32_swap between lanes jj and jj + 32).
// One tile = NKS positions: wait(chunk s) | barrier | DMA(chunk s + P) | 9 taps of chunk s ; after the last: epilogue, NST stores.
// -----------------------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct RingArgs {
  const uint4 *x, *x1;         // h8 [N][G0][H][W] (+ a second plain source [N][G1][H][W], G0 + G1 = 2 NKS, G0 even: the concatenation)
  int G0, G1;
  const uint4* wpack;          // [MB][NKS][9][64]
  const float *bias, *bn_a, *bn_b;
  float slope;                 // 1 = no activation
  uint4* out;                  // h8 [N][4 MB][H][W]
  int N, H, W, tiles_x, tiles_y;
};

template <int NKS, int NIB, int P, int NST>
constexpr int ring3_younger(int c) {             // VM operations issued after the DMA of chunk c and before the top of position c
  const int s0 = ((c - P) % NKS + NKS) % NKS;
  int n = s0 == NKS - 1 ? NST : 0;
  for (int d = 1; d < P; ++d) n += NIB + ((s0 + d) % NKS == NKS - 1 ? NST : 0);
  return n;
}
template <class F, int... Cs>
__device__ __forceinline__ void ring3_static_for(F&& f, std::integer_sequence<int, Cs...>) {
  (f(std::integral_constant<int, Cs>{}), ...);
}

template <int DIL, int MB, int NKS, int RPW, int D>
__global__ __launch_bounds__(512, 2) void ring3_h8_kernel(const RingArgs a) {
  constexpr int NWAVE = 8, KS = 3, T = 9, PAD = DIL, P = D - 1;
  constexpr int C = 32 * MB, GO = 4 * MB;
  constexpr int TW = 64, TH = NWAVE * RPW, NB = 2 * RPW;
  constexpr int LW = TW + 2 * PAD, LH = TH + 2 * PAD, REC = LH * LW;
  constexpr int NBLK_B = (2 * REC + 63) / 64, NIB = (NBLK_B + NWAVE - 1) / NWAVE;
  constexpr int BUFREC = NBLK_B * 64;
  constexpr int NST = MB * NB * 2;
  static_assert(P >= 1 && P <= NKS, "ring depth");

  // static array + native vector loads: see tail2_h8_kernel (reads without a TBAA tag are guarded with vmcnt(0) by the compiler)
  __shared__ __attribute__((aligned(16))) float s_epi[3 * C];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_w = reinterpret_cast<uint4*>(smem);                      // [MB][NKS][9][64]
  uint4* s_ring = s_w + MB * NKS * T * 64;                          // [D][BUFREC]
  uint4* s_trash = s_ring + D * BUFREC;                             // [64]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave;
  const int hh = lane >> 5, jj = lane & 31;
  const size_t HW = (size_t)a.H * a.W;

  int t_beg, t_end, t_step;
  {
    const int nwg = gridDim.x, b = blockIdx.x, xcd = b & 7, qq = nwg >> 3, rr = nwg & 7;
    const int w = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (b >> 3);
    const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
    t_step = nwg;
    t_beg = w;
    t_end = w < nt ? w + (int)((nt - w + nwg - 1) / nwg) * nwg : w;
  }
  if (t_beg >= t_end) return;

  if (tid < C) {
    s_epi[tid] = a.bias ? a.bias[tid] : 0.0f;
    s_epi[C + tid] = a.bn_a ? a.bn_a[tid] : 1.0f;
    s_epi[2 * C + tid] = a.bn_a ? a.bn_b[tid] : 0.0f;
  }
  for (int blk = wave; blk < MB * NKS * T; blk += NWAVE) SLU_GLDS16(a.wpack + (size_t)blk * 64 + lane, s_w + blk * 64);

  struct TilePos { int x0, y0, n; };
  auto decode = [&](int t) {
    TilePos p;
    const int tx = t % a.tiles_x;
    t /= a.tiles_x;
    p.x0 = tx * TW;
    p.y0 = (t % a.tiles_y) * TH;
    p.n = t / a.tiles_y;
    return p;
  };
  int pc_rc[NIB], pc_off[NIB];
#pragma unroll
  for (int i = 0; i < NIB; ++i) {
    const int e = (i * NWAVE + wave) * 64 + lane;
    const int g2 = e / REC, rem = e - g2 * REC, r = rem / LW, c = rem - r * LW;
    pc_rc[i] = r | (c << 8) | ((g2 & 1) << 16) | ((e < 2 * REC ? 1 : 0) << 17);
    pc_off[i] = (r * a.W + c) * 16;
  }
  auto stage = [&](const TilePos& tp, int c, int slot, bool valid) {
    uint4* db = s_ring + slot * BUFREC;
    const uintptr_t zero = reinterpret_cast<uintptr_t>(&g_zero_rec);
    const bool second = 2 * c >= a.G0;                  // a K-step never straddles the two sources (G0 is even)
    const uint4* src = second ? a.x1 : a.x;
    const int gs = second ? a.G1 : a.G0, g = second ? 2 * c - a.G0 : 2 * c;
    const uintptr_t base0 = reinterpret_cast<uintptr_t>(src) +
                            16 * ((long long)(((size_t)tp.n * gs + g) * HW) + (long long)(tp.y0 - PAD) * a.W + (tp.x0 - PAD));
    const uintptr_t base1 = base0 + 16 * (long long)HW;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int blk = i * NWAVE + wave;
      const int rc = pc_rc[i];
      const int gy = tp.y0 - PAD + (rc & 255), gx = tp.x0 - PAD + ((rc >> 8) & 255);
      const bool ok = valid && (rc >> 17) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const uintptr_t p = ok ? (((rc >> 16) & 1) ? base1 : base0) + (long long)pc_off[i] : zero;
      SLU_GLDS16(reinterpret_cast<const uint4*>(p), (NBLK_B % NWAVE == 0 || blk < NBLK_B) ? db + blk * 64 : s_trash);
    }
    asm volatile("" ::: "memory");
  };

  const int bbase = hh * REC + (wn * RPW) * LW + jj;
  TilePos cur = decode(t_beg), nxt = cur;
  bool has_next = t_beg + t_step < t_end;
  if (has_next) nxt = decode(t_beg + t_step);
#pragma unroll
  for (int c = 0; c < P; ++c) stage(cur, c, c, true);
  int rslot = 0, wslot = P % D;
  bool first = true;
  const float2v sl = {a.slope, a.slope};
  const f32x4v* se4p = reinterpret_cast<const f32x4v*>(s_epi) + hh;

  for (int tile = t_beg; tile < t_end; tile += t_step) {
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][b][r] = 0.0f;

    auto position = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int YOUNG = ring3_younger<NKS, NIB, P, NST>(c);
      static_assert(YOUNG <= 63, "vmcnt is a 6-bit counter");
      if (first && c < P) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first tile's prologue (and the resident weights)
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        constexpr int cn = (c + P) % NKS;
        if (c + P < NKS) stage(cur, cn, wslot, true);
        else stage(nxt, cn, wslot, has_next);
        wslot = wslot + 1 == D ? 0 : wslot + 1;
      }
      const uint4* sb = s_ring + rslot * BUFREC + bbase;
      rslot = rslot + 1 == D ? 0 : rslot + 1;
      half8 fa[2][MB], fb[2][NB];
      auto rd = [&](int tap, int buf) {
        const int dy = (tap / KS) * DIL, dx = (tap % KS) * DIL;
#pragma unroll
        for (int i = 0; i < MB; ++i) fa[buf][i] = __builtin_bit_cast(half8, s_w[((i * NKS + c) * T + tap) * 64 + lane]);
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[buf][b] = __builtin_bit_cast(half8, sb[((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
      };
      rd(0, 0);
#pragma unroll
      for (int tap = 0; tap < T; ++tap) {
        if (tap + 1 < T) rd(tap + 1, (tap + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int i = 0; i < MB; ++i) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[tap & 1][i], fb[tap & 1][b], acc[i][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    ring3_static_for(position, std::make_integer_sequence<int, NKS>{});
    first = false;

#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj;
        const bool ok = gy < a.H && gx < a.W;
        const size_t idx0 = ok ? ((size_t)cur.n * GO + i * 4 + hh) * HW + (size_t)gy * a.W + gx : 0;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          unsigned hw[4];
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const int q = 2 * pr + q2;
            const int c4 = (i * 32 + 8 * q) / 4;
            const f32x4v bi = se4p[c4], ba = se4p[C / 4 + c4], bb = se4p[2 * C / 4 + c4];
            float2v t0 = {acc[i][b][4 * q], acc[i][b][4 * q + 1]}, t1 = {acc[i][b][4 * q + 2], acc[i][b][4 * q + 3]};
            t0 += float2v{bi.x, bi.y};
            t1 += float2v{bi.z, bi.w};
            t0 = __builtin_elementwise_max(t0, t0 * sl);
            t1 = __builtin_elementwise_max(t1, t1 * sl);
            t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
            t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
            hw[2 * q2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v));
            hw[2 * q2 + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v));
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(hw[0], hw[2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(hw[1], hw[3], false, false);
          uint4* dst = ok ? a.out + idx0 + (size_t)(2 * pr) * HW : &g_trash_rec;
          *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    asm volatile("" ::: "memory");
    cur = nxt;
    has_next = tile + 2 * t_step < t_end;
    if (has_next) nxt = decode(tile + 2 * t_step);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // zero-record DMAs issued for the tile after the last: LDS must not be released under them
}

template <int DIL, int MB, int NKS, int RPW, int D>
int launch_ring3(const H8Args& h, const slu_conv_h8_desc* d, hipStream_t st) {
  constexpr int TH = 8 * RPW, PAD = DIL;
  constexpr size_t nblk_b = (size_t)(2 * (TH + 2 * PAD) * (64 + 2 * PAD) + 63) / 64;
  constexpr size_t lds = ((size_t)MB * NKS * 9 * 64 + (size_t)D * nblk_b * 64 + 64) * 16;      // + 3 * 32 MB floats static
  static_assert(lds + 3 * 32 * MB * 4 <= 160 * 1024, "ring does not fit in LDS");
  RingArgs a{};
  a.x = h.src[0].ptr; a.G0 = h.src[0].G;
  a.x1 = h.nsrc > 1 ? h.src[1].ptr : h.src[0].ptr; a.G1 = h.nsrc > 1 ? h.src[1].G : 0;
  a.wpack = h.wpack; a.bias = h.bias; a.bn_a = h.bn_a; a.bn_b = h.bn_b;
  a.slope = (h.has_act & 3) == 1 ? h.slope : 1.0f;
  a.out = reinterpret_cast<uint4*>(d->out);
  a.N = h.N; a.H = h.H; a.W = h.W;
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
  if (nt <= 0 || nt > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  long long gx = 256;
  if (gx > nt) gx = nt;
  auto kern = ring3_h8_kernel<DIL, MB, NKS, RPW, D>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, st, a);
  SLU_CHECK_LAUNCH();
}

// the layers ring3_h8_kernel covers: 3x3 (dil 1 / 2), one plain source of exactly 32 / 64 channels, 32 / 64 output channels, h8 output,
// no residual, activation before BN only, and enough tiles to fill the chip.  Returns the instantiation's name, or nullptr.
const char* ring3_name(const slu_conv_h8_desc* d, const H8Args& a, char* buf, size_t n) {
  static const bool off = [] { const char* e = getenv("SLU_H8_RING3"); return e && e[0] == '0'; }();      // A/B switch
  if (off || d->ksize != 3 || d->pad != d->dil || (d->dil != 1 && d->dil != 2) || a.nsrc < 1 || a.nsrc > 2) return nullptr;
  int gsum = 0;
  for (int s = 0; s < a.nsrc; ++s) {
    if (a.src[s].scale || a.src[s].nb) return nullptr;
    gsum += a.src[s].G;
  }
  if (gsum != a.Gin || (a.nsrc == 2 && (a.src[0].G & 1)) || a.out_f32 || d->resid || (a.has_act & ~1)) return nullptr;
  if ((a.has_act & 1) && !(a.slope >= 0.0f && a.slope <= 1.0f)) return nullptr;      // LeakyReLU as max(t, slope t)
  const int mb = a.Cout / 32, nks = a.Gin / 2;
  const bool shape = (a.Cout == 32 || a.Cout == 64) && ((a.Gin == 4 || a.Gin == 8) || (a.Gin == 10 && a.Cout == 32 && d->dil == 1));
  if (!shape) return nullptr;
  const int rpw = (mb * nks >= 5 || (d->dil == 2 && mb * nks == 4)) ? 1 : 2;      // 16-row tiles where weights + the ring fit in LDS
  if ((long long)a.N * ((a.H + 8 * rpw - 1) / (8 * rpw)) * ((a.W + 63) / 64) < 256) return nullptr;
  snprintf(buf, n, "ring3_h8_kernel<%d, %d, %d, %d, %d>", d->dil, mb, nks, rpw, nks == 5 ? 4 : 3);
  return buf;
}

int launch_ring3_any(const H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  const int mb = a.Cout / 32, nks = a.Gin / 2;
  if (nks == 5) return launch_ring3<1, 1, 5, 1, 4>(a, d, st);                       // 80 -> 32: PixelShuffle output | skip (UpBlock.conv1, full resolution)
  if (mb == 2 && nks == 4) return d->dil == 1 ? launch_ring3<1, 2, 4, 1, 3>(a, d, st) : launch_ring3<2, 2, 4, 1, 3>(a, d, st);
  if (mb == 2 && nks == 2) return d->dil == 1 ? launch_ring3<1, 2, 2, 2, 3>(a, d, st) : launch_ring3<2, 2, 2, 1, 3>(a, d, st);
  if (mb == 1 && nks == 4) return d->dil == 1 ? launch_ring3<1, 1, 4, 2, 3>(a, d, st) : launch_ring3<2, 1, 4, 1, 3>(a, d, st);
  return d->dil == 1 ? launch_ring3<1, 1, 2, 2, 3>(a, d, st) : launch_ring3<2, 1, 2, 2, 3>(a, d, st);
}

constexpr size_t WRES_MAX_BYTES = 24 * 1024;

// weights of all K-steps stay resident in LDS when they are small (full-resolution 32-channel layers)
template <int KS, int DIL, int PAD, int MB, int WM, int WN, int RPW, bool SCALED>
int launch_h8(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  const size_t wbytes = (size_t)WM * MB * a.nks * KS * KS * 64 * 16;
  if (wbytes <= WRES_MAX_BYTES) return launch_h8_k<KS, DIL, PAD, MB, WM, WN, RPW, SCALED, true>(a, d, st);
  return launch_h8_k<KS, DIL, PAD, MB, WM, WN, RPW, SCALED, false>(a, d, st);
}

inline long long wg_count(const H8Args& a, int th, int mblk) {
  return (long long)a.N * ((a.H + th - 1) / th) * ((a.W + 63) / 64) * ((a.nmblk + mblk - 1) / mblk);
}

// tile configurations {MB, WM, WN, RPW}: 8-wave workgroups (16 or 8 rows) when the layer has enough tiles for
// every CU, 4-wave ones with 8 / 4 rows for the small feature maps at the bottom of the U-Net
enum { CFG_M32_TH16 = 0, CFG_M64_TH16, CFG_M128_TH8, CFG_M32_TH8, CFG_M64_TH8, CFG_M128_TH4, CFG_M32_TH4, CFG_M64_TH4, CFG_COUNT };
const int CFG_TABLE[CFG_COUNT][4] = {{1, 1, 8, 2}, {2, 1, 8, 2}, {2, 2, 4, 2}, {1, 1, 4, 2}, {2, 1, 4, 2}, {2, 2, 2, 2}, {1, 1, 4, 1}, {2, 1, 4, 1}};

int choose_h8(const H8Args& a) {
  const long long want = 256;
  static const int forced = [] { const char* e = getenv("SLU_H8_CFG"); return e ? atoi(e) : -1; }();      // development aid
  if (forced >= 0 && forced < CFG_COUNT) return forced;
  if (a.nmblk >= 4) {
    if (a.H >= 8 && wg_count(a, 8, 4) >= want) return CFG_M128_TH8;
    if (wg_count(a, 4, 4) >= want) return CFG_M128_TH4;
    if (wg_count(a, 4, 2) >= want) return CFG_M64_TH4;
    return CFG_M32_TH4;
  }
  if (a.nmblk >= 2) {
    if (a.H >= 16 && wg_count(a, 16, 2) >= want) return CFG_M64_TH16;
    if (a.H >= 8 && wg_count(a, 8, 2) >= want) return CFG_M64_TH8;
    if (wg_count(a, 4, 2) >= want) return CFG_M64_TH4;
    return CFG_M32_TH4;
  }
  if (a.H >= 16 && wg_count(a, 16, 1) >= want) return CFG_M32_TH16;
  if (a.H >= 8 && wg_count(a, 8, 1) >= want) return CFG_M32_TH8;
  return CFG_M32_TH4;
}

// The 8-wave 128-channel configuration (the MFMA-bound 128 / 256-channel layers of the U-Net's lower levels) with its scheduling
// options (conv_h8_kernel's OPT / KPC).  SLU_H8_OPT / SLU_H8_KPC2 (environment) and, in -DSLU_H8_AB builds, slu_h8_dev_set_opt()
// are A/B switches of the development tools.
int g_h8_opt = [] { const char* e = getenv("SLU_H8_OPT"); return e ? atoi(e) : -1; }();
int g_h8_kpc2 = [] { const char* e = getenv("SLU_H8_KPC2"); return e ? atoi(e) : H8_M128_KPC2_DEFAULT; }();

constexpr size_t h8_m128_weight_bytes(int nks, int T) { return (size_t)4 * nks * T * 64 * 16; }

inline bool h8_one_plain_source(const H8Args& a) { return a.nsrc == 1 && a.src[0].nb == 0 && !a.src[0].scale; }

template <int KS, int DIL, int PAD, bool SCALED>
int launch_h8_m128(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  if (h8_m128_weight_bytes(a.nks, KS * KS) <= WRES_MAX_BYTES) return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, SCALED, true>(a, d, st);
  if constexpr (!SCALED) {
    if (h8_one_plain_source(a)) {
      if constexpr (KS == 2) {
        if (g_h8_kpc2 && a.nks >= 2) {
#ifdef SLU_H8_AB
          if (g_h8_opt == 0) return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 2, 0, true>(a, d, st);
#endif
          return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 2, H8_M128_OPT_2X2, true>(a, d, st);
        }
      }
#ifdef SLU_H8_AB
      switch (g_h8_opt) {
        case 0: return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 1, 0, true>(a, d, st);
        case 8: return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 1, 8, true>(a, d, st);
        case 15: return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 1, 15, true>(a, d, st);
        case 100: return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 1, 0, false>(a, d, st);      // the general form, for comparison
      }
#endif
      return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, false, false, false, 1, KS == 2 ? H8_M128_OPT_2X2 : H8_M128_OPT_3X3, true>(a, d, st);
    }
  }
  return launch_h8_k<KS, DIL, PAD, 2, 2, 4, 2, SCALED, false>(a, d, st);
}

template <int KS, int DIL, int PAD, bool SCALED>
int launch_h8_tiles(H8Args& a, const slu_conv_h8_desc* d, int cfg, hipStream_t st) {
  switch (cfg) {
    case CFG_M32_TH16: return launch_h8<KS, DIL, PAD, 1, 1, 8, 2, SCALED>(a, d, st);
    case CFG_M64_TH16: return launch_h8<KS, DIL, PAD, 2, 1, 8, 2, SCALED>(a, d, st);
    case CFG_M128_TH8: return launch_h8_m128<KS, DIL, PAD, SCALED>(a, d, st);
    case CFG_M32_TH8:  return launch_h8<KS, DIL, PAD, 1, 1, 4, 2, SCALED>(a, d, st);
    case CFG_M64_TH8:  return launch_h8<KS, DIL, PAD, 2, 1, 4, 2, SCALED>(a, d, st);
    case CFG_M128_TH4: return launch_h8<KS, DIL, PAD, 2, 2, 2, 2, SCALED>(a, d, st);
    case CFG_M32_TH4:  return launch_h8<KS, DIL, PAD, 1, 1, 4, 1, SCALED>(a, d, st);
    case CFG_M64_TH4:  return launch_h8<KS, DIL, PAD, 2, 1, 4, 1, SCALED>(a, d, st);
  }
  return SLU_EUNSUPPORTED;
}

template <int KS, int DIL, int PAD>
int launch_h8_family(H8Args& a, const slu_conv_h8_desc* d, int cfg, bool scaled, hipStream_t st) {
  return scaled ? launch_h8_tiles<KS, DIL, PAD, true>(a, d, cfg, st) : launch_h8_tiles<KS, DIL, PAD, false>(a, d, cfg, st);
}

template <int MB, int NBW>
int launch_h8_1x1(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  constexpr size_t lds = (size_t)MB * 4 * 64 * 16 + (size_t)3 * MB * 32 * 4;
  const long long nblocks = (long long)a.N * a.H * a.W / 32;
  const long long gx = (nblocks + 4 * NBW - 1) / (4 * NBW);
  if (gx <= 0 || gx > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL((conv1x1_h8_kernel<MB, NBW>), dim3((unsigned)gx), dim3(256), lds, st, a, d->resid, d->out);
  SLU_CHECK_LAUNCH();
}

template <int MB, int NKS>
int launch_h8_1x1_res(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  constexpr size_t lds = (size_t)MB * NKS * 64 * 16 + (size_t)3 * MB * 32 * 4;
  const long long nblocks = (long long)a.N * a.H * a.W / 32;
  long long gx = 256 * ((MB * NKS >= 12) ? 2 : ((MB * NKS >= 2) ? 3 : 4));      // as many 4-wave workgroups per CU as the registers allow
  if (gx * 4 > nblocks) gx = (nblocks + 3) / 4;
  if (gx <= 0) return SLU_EUNSUPPORTED;
  hipLaunchKernelGGL((conv1x1_h8_res_kernel<MB, NKS>), dim3((unsigned)gx), dim3(256), lds, st, a, d->resid, d->out);
  SLU_CHECK_LAUNCH();
}

// resident-weight streaming form: <= 64 output channels and 1 / 2 / 6 / 12 K-steps (the 1x1 convs of the full- and
// half-resolution blocks); returns -1 when the shape is not covered
template <int MB>
int launch_h8_1x1_res_nks(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  switch (a.nks) {
    case 1:  return launch_h8_1x1_res<MB, 1>(a, d, st);
    case 2:  return launch_h8_1x1_res<MB, 2>(a, d, st);
    case 5: case 6:   return launch_h8_1x1_res<MB, 6>(a, d, st);
    case 10: case 12: return launch_h8_1x1_res<MB, 12>(a, d, st);
  }
  return -1;
}

const char* res_1x1_name(const H8Args& a, char* buf, size_t n) {
  const int nks = a.nks <= 2 ? a.nks : (a.nks == 5 || a.nks == 6 ? 6 : ((a.nks == 10 || a.nks == 12) ? 12 : 0));
  if (a.nmblk > 2 || nks == 0) return nullptr;
  snprintf(buf, n, "conv1x1_h8_res_kernel<%d, %d>", a.nmblk, nks);
  return buf;
}

// the layers gemm1x1_h8_kernel covers; SLU_H8_GEMM1X1=0 is the A/B switch back to the streaming kernels
constexpr int GEMM1X1_KC = 4, GEMM1X1_D256 = 2, GEMM1X1_D128 = 3;      // chunk = 64 channels; ring: 2 x 64 KB (256 outputs), 3 x 48 KB (128)
bool gemm1x1_ok(const slu_conv_h8_desc* d, const H8Args& a) {
  static const bool off = [] { const char* e = getenv("SLU_H8_GEMM1X1"); return e && e[0] == '0'; }();
  // 256 output channels only: measured (tools/h8_1x1_bench.py, N = 64) 5 ... 18 % faster than the streaming kernel there (768->256 at 16x512:
  // 504 -> 459 us) and 17 % SLOWER for 128 outputs (384->128 at 32x1024: 544 -> 634 us), whose instantiation stays available to the tests
  static const bool all = [] { const char* e = getenv("SLU_H8_GEMM1X1"); return e && e[0] == '2'; }();
  if (off || d->ksize != 1 || d->pad != 0 || a.out_f32 || (a.Cout != 256 && !(all && a.Cout == 128))) return false;
  if (((long long)a.H * a.W) % 256 || a.nks % GEMM1X1_KC || a.Gin != 2 * a.nks) return false;
  if ((a.has_act & ~1) || ((a.has_act & 1) && !(a.slope >= 0.0f && a.slope <= 1.0f))) return false;
  for (int s = 0; s < a.nsrc; ++s)
    if (a.src[s].scale || a.src[s].nb || a.src[s].G % (2 * GEMM1X1_KC)) return false;
  return (long long)a.N * a.H * a.W / 256 <= 0x7fffffffLL;
}

template <int MB, int D>
int launch_gemm1x1(H8Args& a, const slu_conv_h8_desc* d, hipStream_t st) {
  constexpr int KC = GEMM1X1_KC, MBLK = 2 * MB;
  constexpr size_t lds = (size_t)3 * MBLK * 32 * 4 + (size_t)D * (2 * KC * 256) * 16 + (size_t)D * (MBLK * KC * 64) * 16;
  static_assert(lds <= 160 * 1024, "gemm1x1 LDS");
  const long long nt = (long long)a.N * a.H * a.W / 256;
  const long long gx = nt < 256 ? nt : 256;
  auto kern = gemm1x1_h8_kernel<MB, KC, D>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, st, a, d->resid, d->out);
  SLU_CHECK_LAUNCH();
}

bool any_scale(const slu_conv_h8_desc* d) {
  for (int s = 0; s < d->nsrc; ++s)
    if (d->src[s].scale) return true;
  return false;
}

bool stream_ok(const slu_conv_h8_desc* d, const H8Args& a) {
  return d->ksize == 1 && d->pad == 0 && a.nmblk <= 8 && ((long long)a.H * a.W) % 32 == 0 && !any_scale(d);
}

}  // namespace

extern "C" size_t slu_packed_weight_bytes_h8(int cout, int cin, int ksize) {
  if (cout <= 0 || cin <= 0 || ksize <= 0) return 0;
  const size_t nmblk = (cout + 31) / 32, nks = (cin + 15) / 16;
  return nmblk * nks * (size_t)(ksize * ksize) * 64 * 16;
}

extern "C" int slu_pack_conv_weight_h8(const float* w, int cout, int cin, int ksize, void* out, slu_stream_t stream) {
  if (!w || !out) return SLU_EINVAL;
  const size_t bytes = slu_packed_weight_bytes_h8(cout, cin, ksize);
  if (bytes == 0) return SLU_EINVAL;
  const size_t total = bytes / 16;
  hipLaunchKernelGGL(pack_h8_kernel, dim3(grid_for(total)), dim3(256), 0, slu_stream(stream), w, cout, cin, ksize, (cin + 15) / 16, total,
                     reinterpret_cast<uint4*>(out));
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_conv2d_h8_fwd(const slu_conv_h8_desc* d, slu_stream_t stream) {
  H8Args a{};
  const int rc = fill_h8(d, a);
  if (rc != SLU_OK) return rc;
  hipStream_t st = slu_stream(stream);
  if (gemm1x1_ok(d, a)) return a.Cout == 128 ? launch_gemm1x1<2, GEMM1X1_D128>(a, d, st) : launch_gemm1x1<4, GEMM1X1_D256>(a, d, st);
  if (stream_ok(d, a)) {
    if (a.nmblk <= 2) {
      const int rc2 = a.nmblk == 1 ? launch_h8_1x1_res_nks<1>(a, d, st) : launch_h8_1x1_res_nks<2>(a, d, st);
      if (rc2 != -1) return rc2;
    }
    if (a.nmblk == 1) return launch_h8_1x1<1, 2>(a, d, st);
    if (a.nmblk == 2) return launch_h8_1x1<2, 2>(a, d, st);
    if (a.nmblk <= 4) return launch_h8_1x1<4, 1>(a, d, st);
    return launch_h8_1x1<8, 1>(a, d, st);
  }
  {
    char nm[96];
    if (ring3_name(d, a, nm, sizeof nm)) return launch_ring3_any(a, d, st);
  }
  const bool sc = any_scale(d);
  if (a.out_f32) {      // fp32 NCHW output outside the streaming kernel's reach (odd H*W): 1x1 head only
    if (d->ksize != 1 || d->dil != 1 || d->pad != 0 || sc) return SLU_EUNSUPPORTED;
    return launch_h8_k<1, 1, 0, 1, 1, 4, 1, false, false, true>(a, d, st);
  }
  const int cfg = choose_h8(a);
  if (d->ksize == 1 && d->dil == 1 && d->pad == 0) return launch_h8_family<1, 1, 0>(a, d, cfg, sc, st);
  if (d->ksize == 3 && d->dil == 1 && d->pad == 1) return launch_h8_family<3, 1, 1>(a, d, cfg, sc, st);
  if (d->ksize == 3 && d->dil == 2 && d->pad == 2) return launch_h8_family<3, 2, 2>(a, d, cfg, sc, st);
  if (d->ksize == 2 && d->dil == 2 && d->pad == 1) return launch_h8_family<2, 2, 1>(a, d, cfg, sc, st);
  return SLU_EUNSUPPORTED;
}

// name of the kernel instantiation slu_conv2d_h8_fwd launches for `d` (as rocprofv3 prints it), for per-kernel accounting
extern "C" int slu_conv2d_h8_kernel_name(const slu_conv_h8_desc* d, char* buf, size_t n) {
  H8Args a{};
  const int rc = fill_h8(d, a);
  if (rc != SLU_OK || !buf || n == 0) return rc != SLU_OK ? rc : SLU_EINVAL;
  if (gemm1x1_ok(d, a)) {
    snprintf(buf, n, "gemm1x1_h8_kernel<%d, %d, %d>", a.Cout / 64, GEMM1X1_KC, a.Cout == 128 ? GEMM1X1_D128 : GEMM1X1_D256);
    return SLU_OK;
  }
  if (stream_ok(d, a)) {
    if (res_1x1_name(a, buf, n)) return SLU_OK;
    const int mb = a.nmblk == 1 ? 1 : (a.nmblk == 2 ? 2 : (a.nmblk <= 4 ? 4 : 8));
    snprintf(buf, n, "conv1x1_h8_kernel<%d, %d>", mb, mb <= 2 ? 2 : 1);
    return SLU_OK;
  }
  if (ring3_name(d, a, buf, n)) return SLU_OK;
  if (a.out_f32) {
    snprintf(buf, n, "conv_h8_kernel<1, 1, 0, 1, 1, 4, 1, false, false, true, 1, 0, false>");
    return SLU_OK;
  }
  const int cfg = choose_h8(a);
  const int* c = CFG_TABLE[cfg];
  const bool wres = (size_t)c[0] * c[1] * a.nks * d->ksize * d->ksize * 64 * 16 <= WRES_MAX_BYTES;
  int kpc = 1, opt = 0;
  bool one = false;
  if (cfg == CFG_M128_TH8 && !wres && !any_scale(d) && h8_one_plain_source(a)) {      // launch_h8_m128's choice
    one = true;
    kpc = (d->ksize == 2 && g_h8_kpc2 && a.nks >= 2) ? 2 : 1;
    opt = d->ksize == 2 ? H8_M128_OPT_2X2 : H8_M128_OPT_3X3;
#ifdef SLU_H8_AB
    if (kpc == 2 ? g_h8_opt == 0 : (g_h8_opt == 0 || g_h8_opt == 8 || g_h8_opt == 15)) opt = g_h8_opt;
    if (kpc == 1 && g_h8_opt == 100) opt = 0, one = false;
#endif
  }
  snprintf(buf, n, "conv_h8_kernel<%d, %d, %d, %d, %d, %d, %d, %s, %s, false, %d, %d, %s>", d->ksize, d->dil, d->pad, c[0], c[1], c[2], c[3],
           any_scale(d) ? "true" : "false", wres ? "true" : "false", kpc, opt, one ? "true" : "false");
  return SLU_OK;
}

#ifdef SLU_H8_AB
extern "C" void slu_h8_dev_set_opt(int opt, int kpc2) { g_h8_opt = opt; g_h8_kpc2 = kpc2; }
#endif

#ifdef SLU_H8_PROF
extern "C" int slu_h8_prof_read(unsigned long long* out8) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_h8_prof), sizeof(unsigned long long) * 8) != hipSuccess) return SLU_ELAUNCH;
  unsigned long long z[8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_h8_prof), z, sizeof(z)) == hipSuccess ? SLU_OK : SLU_ELAUNCH;
}
#endif

extern "C" int slu_nchw_to_h8(const float* x, const float* scale, void* y, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ((uintptr_t)y & 15)) return SLU_EINVAL;
  const int G = (C + 7) / 8;
  const size_t total = (size_t)N * G * H * W;
  hipLaunchKernelGGL(nchw_to_h8_kernel, dim3(grid_for(total)), dim3(256), 0, slu_stream(stream), x, scale, reinterpret_cast<uint4*>(y), N, C, G,
                     (size_t)H * W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_h8_to_nchw(const void* x, float* y, int N, int C, int H, int W, slu_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || ((uintptr_t)x & 15)) return SLU_EINVAL;
  const int G = (C + 7) / 8;
  const size_t total = (size_t)N * G * H * W;
  hipLaunchKernelGGL(h8_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, slu_stream(stream), reinterpret_cast<const uint4*>(x), y, N, C, G,
                     (size_t)H * W);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_avgpool3s2_h8(const void* x, const float* scale, void* y, int N, int in_batch, int G, int H, int W, slu_stream_t stream) {
  if (!x || !y || N <= 0 || in_batch < 0 || G <= 0 || H <= 0 || W <= 0 || (((uintptr_t)x | (uintptr_t)y) & 15)) return SLU_EINVAL;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const size_t total = (size_t)N * G * OH * OW;
  hipLaunchKernelGGL(avgpool3s2_h8_kernel, dim3(grid_for(total)), dim3(256), 0, slu_stream(stream), reinterpret_cast<const uint4*>(x), scale,
                     reinterpret_cast<uint4*>(y), N, G, H, W, OH, OW, in_batch);
  SLU_CHECK_LAUNCH();
}

extern "C" int slu_pixel_shuffle_h8(const void* x, const float* scale_in, const float* scale_out, void* y, int N, int Gin, int H, int W,
                                    slu_stream_t stream) {
  if (!x || !y || N <= 0 || Gin <= 0 || H <= 0 || W <= 0 || (((uintptr_t)x | (uintptr_t)y) & 15)) return SLU_EINVAL;
  const int Go = (Gin * 2 + 7) / 8;                 // Cin/4 = 2 Gin output channels
  const size_t total = (size_t)N * Go * H * W;      // one thread per (input pixel, output block)
  hipLaunchKernelGGL(pixel_shuffle_h8_kernel, dim3(grid_for(total)), dim3(256), 0, slu_stream(stream), reinterpret_cast<const uint4*>(x), scale_in,
                     scale_out, reinterpret_cast<uint4*>(y), N, Gin, Go, H, W);
  SLU_CHECK_LAUNCH();
}
