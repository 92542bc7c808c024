// Fused tail of a SalsaNext block on the half-precision ("h8") path:
//
//     a3  = bnA * leaky(conv2x2_dil2(a2) + biasA) + bnA_b                (ResBlock.conv4 / UpBlock.conv3, SalsaNext.py:59-62,157-160)
//     out = [resid +] bnB * leaky(conv1x1(cat(a1, a2, a3)) + biasB) + bnB_b   (ResBlock.conv5 / UpBlock.conv4, :64-68,162-167)
//
// Unfused these are two kernels moving 7 tensor passes (read a2, write a3; read a1, a2, a3, resid, write out); here a3 never
// leaves the CU: 4 passes (a1, a2, resid, out), both layers HBM-bound at full resolution.  Structure = the persistent LDS-DMA
// kernel of conv2d_h8.hip (round-robin tiles, one K-step = 16 channels per chunk, double-buffered, the next tile's first chunk
// in flight across the epilogues):
//   chunks 0 .. nks-1      : a2 tile (halo 1) -> 4 dilated taps into acc3  +  the centre tap (1x1 over a2) into acc_out
//   chunks nks .. 2 nks-1  : a1 tile          -> the centre tap (1x1 over a1) into acc_out
//   epilogue A             : acc3 -> bias, LeakyReLU, BN -> fp16 -> LDS image of a3 in B-operand layout [block][row][64]
//   stage 3                : 1x1 over a3 from LDS (its weights stay resident in LDS) into acc_out
//   epilogue B             : acc_out -> bias, LeakyReLU, BN, + resid -> h8 store
// C = 32 MB WM channels, 8 waves = WM x WN, a wave owns MB channel blocks and RPW output rows: (C, TH) = (32, 16), (64, 8), (128, 4)
// -- the a3 image takes C * TH * 128 B = 64 KB in each.  W3RES: the 1x1 weights over a3 stay resident in LDS (C <= 64); otherwise
// they stream through the chunk pipeline as NKS more (weights-only) chunks, for which the LDS has no room at C = 128.
#include "slu_common.h"
#include <cstdlib>
#include <utility>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

namespace {

__device__ uint4 g_zero_rec_t;      // source of every out-of-image record of an LDS-DMA copy (never written)
__device__ uint4 g_trash_rec_t;     // where lanes outside the image store

#define SLU_GLDS16_T(gsrc, ldst_wave_base)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),                 \
                                   (__attribute__((address_space(3))) void*)(ldst_wave_base), 16, 0, 0)

struct TailArgs {
  const uint4 *a1, *a2;        // h8 [N][G][H][W]
  const uint4 *w2, *w1;        // packed 2x2 weights [MB][nks][4][64]; packed 1x1 weights over 3 C inputs [MB][3 nks][1][64]
  const float *biasA, *bnA_a, *bnA_b, *biasB, *bnB_a, *bnB_b;
  float slopeA, slopeB;        // 1 = no activation
  const uint2* resid;          // h8 or nullptr
  const uint4* sc_x;           // shortcut mode (tail2, C = 64): h8 [N][4][H][W], the block's input; resid = act(conv1x1(sc_x) + sc_bias) computed here
  const uint4* sc_w;           // packed [C][32][1][1] weights
  const float* sc_bias;
  float slopeS;
  uint2* out;
  int N, H, W, G;              // G = C / 8 channel blocks
  int tiles_x, tiles_y;
  int dbg;                     // development: 1 = ring DMAs copy the zero record (no input traffic), 2 = no MFMA work
};

template <int MB, int WM, int WN, int RPW, bool W3RES>
__global__ __launch_bounds__(64 * WM * WN, 2) void tail_h8_kernel(const TailArgs a) {
  constexpr int NWAVE = WM * WN, T = 4, PAD = 1, DIL = 2;
  constexpr int MBLK = MB * WM, C = 32 * MBLK, NKS = 2 * MBLK;
  constexpr int NCH = W3RES ? 2 * NKS : 3 * NKS;                       // chunks per tile
  constexpr int TW = 64, TH = WN * RPW, NB = 2 * RPW;
  constexpr int LW = TW + 2 * PAD, LH = TH + 2 * PAD, REC = LH * LW;
  constexpr int NREC_B = 2 * REC, NBLK_B = (NREC_B + 63) / 64, NB_ALLOC = NBLK_B * 64;
  constexpr int NBLK_A = MBLK * (T + 1), NREC_A = NBLK_A * 64;     // per chunk: 4 tap fragments + 1 centre fragment per channel block
  constexpr int NIB = (NBLK_B + NWAVE - 1) / NWAVE, NIA = (NBLK_A + NWAVE - 1) / NWAVE;
  constexpr int NST = MB * NB * 4;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_epi = reinterpret_cast<float*>(smem);                    // A: bias | bn_a | bn_b ; B: bias | bn_a | bn_b   (6 C floats)
  uint4* s_b = reinterpret_cast<uint4*>(s_epi + 6 * C);             // [2][NB_ALLOC] input tiles
  uint4* s_a = s_b + 2 * NB_ALLOC;                                  // [2][NREC_A] weight fragments of a chunk
  uint4* s_w3 = s_a + 2 * NREC_A;                                   // W3RES: [MBLK][NKS][64] resident 1x1 weights over a3
  uint4* s_a3 = s_w3 + (W3RES ? MBLK * NKS * 64 : 0);               // [C / 8][TH][64] the a3 tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
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
    s_epi[tid] = a.biasA ? a.biasA[tid] : 0.0f;
    s_epi[C + tid] = a.bnA_a ? a.bnA_a[tid] : 1.0f;
    s_epi[2 * C + tid] = a.bnA_a ? a.bnA_b[tid] : 0.0f;
    s_epi[3 * C + tid] = a.biasB ? a.biasB[tid] : 0.0f;
    s_epi[4 * C + tid] = a.bnB_a ? a.bnB_a[tid] : 1.0f;
    s_epi[5 * C + tid] = a.bnB_a ? a.bnB_b[tid] : 0.0f;
  }
  if constexpr (W3RES) {
    for (int blk = wave; blk < MBLK * NKS; blk += NWAVE) {             // resident 1x1 weights of the a3 third: K-steps 2 NKS .. 3 NKS - 1
      const int m = blk / NKS, k = blk - m * NKS;
      SLU_GLDS16_T(a.w1 + ((size_t)m * 3 * NKS + 2 * NKS + k) * 64 + lane, s_w3 + blk * 64);
    }
  }

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
    pc_rc[i] = r | (c << 8) | ((g2 & 1) << 16) | ((e < NREC_B ? 1 : 0) << 17);
    pc_off[i] = r * a.W + c;
  }
  // chunk c of a tile: c < NKS: K-step c of a2 (+ its 2x2 and 1x1 weight fragments); c < 2 NKS: K-step c - NKS of a1 (+ its 1x1
  // fragments); else (streamed w3): only the 1x1 fragments of K-step c - 2 NKS of a3
  auto stage = [&](const TilePos& tp, int c, int buf) {
    const bool second = c >= NKS, third = c >= 2 * NKS;
    const int q = third ? c - 2 * NKS : (second ? c - NKS : c);
    const uint4* src = second ? a.a1 : a.a2;
    const uintptr_t base0 = reinterpret_cast<uintptr_t>(src) +
                            16 * ((long long)(((size_t)tp.n * a.G + 2 * q) * HW) + (long long)(tp.y0 - PAD) * a.W + (tp.x0 - PAD));
    const uintptr_t base1 = base0 + 16 * (long long)HW;
    uint4* db = s_b + buf * NB_ALLOC;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int blk = i * NWAVE + wave;
      if (!third && (NBLK_B % NWAVE == 0 || blk < NBLK_B)) {
        const int rc = pc_rc[i];
        const int gy = tp.y0 - PAD + (rc & 255), gx = tp.x0 - PAD + ((rc >> 8) & 255);
        const bool ok = (rc >> 17) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const uintptr_t p = ok ? (((rc >> 16) & 1) ? base1 : base0) + 16 * (long long)pc_off[i] : reinterpret_cast<uintptr_t>(&g_zero_rec_t);
        SLU_GLDS16_T(reinterpret_cast<const uint4*>(p), db + blk * 64);
      }
    }
    uint4* da = s_a + buf * NREC_A;
#pragma unroll
    for (int i = 0; i < NIA; ++i) {
      const int blk = i * NWAVE + wave;                                // = m * (T + 1) + f ; f < T: tap fragment, f == T: centre (1x1) fragment
      if (NBLK_A % NWAVE == 0 || blk < NBLK_A) {
        const int m = blk / (T + 1), f = blk - m * (T + 1);
        if (f < T) {
          if (!second) SLU_GLDS16_T(a.w2 + (((size_t)m * NKS + q) * T + f) * 64 + lane, da + blk * 64);      // a1 chunks have no 2x2 taps
        } else {
          SLU_GLDS16_T(a.w1 + ((size_t)m * 3 * NKS + (third ? 2 * NKS + q : (second ? q : NKS + q))) * 64 + lane, da + blk * 64);   // cat order (a1, a2, a3)
        }
      }
    }
  };

  const int bbase = hh * REC + (wn * RPW) * LW + jj;      // wn: the wave's row group; wm: its channel-block group
  TilePos cur = decode(t_beg), nxt = cur;
  stage(cur, 0, 0);
  int buf = 0;
  const float2v slA = {a.slopeA, a.slopeA}, slB = {a.slopeB, a.slopeB};
  const float4* se4 = reinterpret_cast<const float4*>(s_epi);

  for (int tile = t_beg; tile < t_end; tile += t_step) {
    f32x16 acc3[MB][NB], acco[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc3[i][b][r] = 0.0f; acco[i][b][r] = 0.0f; }

    // epilogue A: a3 tile -> LDS (fp16, the rounding the unfused path applies when it stores a3)
    auto a3_to_lds = [&]() {
      uint2* s3 = reinterpret_cast<uint2*>(s_a3);
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int row = wn * RPW + (b >> 1), px = (b & 1) * 32 + jj, ml = wm * MB + i;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c4 = (ml * 32 + 8 * q) / 4 + hh;
            const float4 bi = se4[c4], ba = se4[C / 4 + c4], bb = se4[2 * C / 4 + c4];
            float2v t0 = {acc3[i][b][4 * q], acc3[i][b][4 * q + 1]}, t1 = {acc3[i][b][4 * q + 2], acc3[i][b][4 * q + 3]};
            t0 += float2v{bi.x, bi.y};
            t1 += float2v{bi.z, bi.w};
            t0 = __builtin_elementwise_max(t0, t0 * slA);
            t1 = __builtin_elementwise_max(t1, t1 * slA);
            t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
            t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
            s3[((((ml * 4 + q) * TH + row) * 64 + px) << 1) + hh] =
                make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v)), __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v)));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    };
    // one K-step of the 1x1 over a3: B operands from the LDS tile (channel blocks 2k, 2k+1), weight fragments from `wf`
    auto a3_step = [&](int k, const uint4* wf, int wstride) {
      half8 af[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) af[i] = __builtin_bit_cast(half8, wf[i * wstride]);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const half8 bf = __builtin_bit_cast(half8, s_a3[((2 * k + hh) * TH + wn * RPW + (b >> 1)) * 64 + (b & 1) * 32 + jj]);
#pragma unroll
        for (int i = 0; i < MB; ++i) acco[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf, acco[i][b], 0, 0, 0);
      }
    };

    for (int c = 0; c < NCH; ++c) {
      // the chunk has landed; at a tile's first chunk only the NST stores of the previous tile's epilogue are younger than its DMA
      if (c == 0 && tile != t_beg) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST < 63 ? NST : 63) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (c + 1 < NCH) {
        stage(cur, c + 1, buf ^ 1);
      } else if (tile + t_step < t_end) {
        nxt = decode(tile + t_step);
        stage(nxt, 0, buf ^ 1);
      }
      const uint4* sb = s_b + buf * NB_ALLOC + bbase;
      const uint4* sa = s_a + buf * NREC_A + (wm * MB) * (T + 1) * 64 + lane;
      if (c < NKS) {
#pragma unroll
        for (int tap = 0; tap < T; ++tap) {
          const int dy = (tap >> 1) * DIL, dx = (tap & 1) * DIL;
          half8 af[MB];
#pragma unroll
          for (int i = 0; i < MB; ++i) af[i] = __builtin_bit_cast(half8, sa[(i * (T + 1) + tap) * 64]);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const half8 bf = __builtin_bit_cast(half8, sb[((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
#pragma unroll
            for (int i = 0; i < MB; ++i) acc3[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf, acc3[i][b], 0, 0, 0);
          }
        }
      }
      if (W3RES || c < 2 * NKS) {      // centre tap: the 1x1 conv over this K-step of a2 / a1
        half8 af[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) af[i] = __builtin_bit_cast(half8, sa[(i * (T + 1) + T) * 64]);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const half8 bf = __builtin_bit_cast(half8, sb[((b >> 1) + PAD) * LW + (b & 1) * 32 + PAD]);
#pragma unroll
          for (int i = 0; i < MB; ++i) acco[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf, acco[i][b], 0, 0, 0);
        }
      } else {
        a3_step(c - 2 * NKS, sa + T * 64, (T + 1) * 64);               // streamed w3: this chunk carried only the centre fragments
      }
      if (!W3RES && c == 2 * NKS - 1) a3_to_lds();                     // visible to every wave after the next chunk's barrier
      buf ^= 1;
    }

    if constexpr (W3RES) {
      a3_to_lds();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < NKS; ++k) a3_step(k, s_w3 + ((wm * MB) * NKS + k) * 64 + lane, NKS * 64);
    }

    // ---- epilogue B: every lane issues its 4 stores per accumulator tile (counted vmcnt above) ----
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj, ml = wm * MB + i;
        const bool ok = gy < a.H && gx < a.W;
        const size_t idx0 = ok ? ((((size_t)cur.n * a.G + ml * 4) * HW + (size_t)gy * a.W + gx) << 1) + hh : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c4 = (ml * 32 + 8 * q) / 4 + hh;
          const float4 bi = se4[3 * C / 4 + c4], ba = se4[4 * C / 4 + c4], bb = se4[5 * C / 4 + c4];
          float2v t0 = {acco[i][b][4 * q], acco[i][b][4 * q + 1]}, t1 = {acco[i][b][4 * q + 2], acco[i][b][4 * q + 3]};
          t0 += float2v{bi.x, bi.y};
          t1 += float2v{bi.z, bi.w};
          t0 = __builtin_elementwise_max(t0, t0 * slB);
          t1 = __builtin_elementwise_max(t1, t1 * slB);
          t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
          t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
          const size_t idx = idx0 + (size_t)q * HW * 2;
          if (a.resid) {
            const uint2 r = *(ok ? a.resid + idx : reinterpret_cast<const uint2*>(&g_zero_rec_t));
            t0 += __builtin_convertvector(__builtin_bit_cast(half2v, r.x), float2v);
            t1 += __builtin_convertvector(__builtin_bit_cast(half2v, r.y), float2v);
          }
          *(ok ? a.out + idx : reinterpret_cast<uint2*>(&g_trash_rec_t)) =
              make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v)), __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Version 2 (C = 32, 64: one wave owns ALL channels of its pixels).  The kernel above keeps ~21 KB of DMA in flight per CU -- against
// the ~43 KB that 5.5 TB/s x ~2 us of loaded HBM latency asks of each of the 256 CUs -- because 64 KB of LDS hold the a3 tile and
// every chunk re-fetches its weight fragments; its epilogue loads the residual one record at a time.  Here
//   * a3 never leaves the registers: the 32x32 accumulator layout gives lane (pixel jj, half hh) the channels 32 i + 8 q + 4 hh + e;
//     the 1x1 over a3 runs its K-steps in THAT channel order (k = 2 i + p: channels 32 i + 16 p + {4 hh + e, 8 + 4 hh + e}), so the
//     converted accumulators ARE the B operands, and the matching weight fragments are permuted once per workgroup when they are
//     made resident -- no LDS image, no barrier, no ds_read for a3;
//   * a1 and the residual never touch LDS either: the 1x1 conv is pixel-local, so lane (jj, hh) loads exactly the h8 records that are
//     its B operands (a1) / its epilogue addends (resid) straight into registers, one half-tile ahead of their use;
//   * every weight (2x2, 1x1 over a1 | a2 | a3) is resident in LDS for the life of the persistent workgroup;
//   * the freed LDS is a ring of D a2 chunks (with halo), D - 1 of them in flight (C = 64: 3 x 21 KB; C = 32: 2 x 38 KB);
//   * every wait is a COUNTED vmcnt: each wave issues the same VM operations at every position of every tile (surplus DMA slots copy
//     the zero record to a trash block; beyond the last tile whole chunks do), the register loads are inline asm the compiler does not
//     track (it would wait for them with vmcnt(0), draining the ring), so "chunk c has landed" is vmcnt(younger(c)), a constant.
// One tile = NKS positions; position s, in program order:
//   wait(chunk s) | barrier | DMA(chunk s + P) | s == 0: a1 loads | s == NKS / 2: wait + consume a1, then residual loads |
//   chunk s: 4 dilated taps -> acc3, centre tap -> acc_out | s == NKS - 1: a3 in registers -> acc_out ;  after the last: epilogue, stores
template <int NKS, int NIB, int P, int NA1, int NRES, int NST>
constexpr int tail2_ops_after_dma(int s) {       // VM operations position s issues after its DMA
  return (s == 0 ? NA1 : 0) + (s == NKS / 2 ? NRES : 0) + (s == NKS - 1 ? NST : 0);
}
template <int NKS, int NIB, int P, int NA1, int NRES, int NST>
constexpr int tail2_younger(int c) {             // ... issued after the DMA of chunk c and before the top of position c (cyclic over tiles)
  const int s0 = ((c - P) % NKS + NKS) % NKS;
  int n = tail2_ops_after_dma<NKS, NIB, P, NA1, NRES, NST>(s0);
  for (int d = 1; d < P; ++d) n += NIB + tail2_ops_after_dma<NKS, NIB, P, NA1, NRES, NST>((s0 + d) % NKS);
  return n;
}
template <class F, int... Cs>
__device__ __forceinline__ void tail2_static_for(F&& f, std::integer_sequence<int, Cs...>) {
  (f(std::integral_constant<int, Cs>{}), ...);
}

// RES: 0 no residual; 1 residual tensor; 2 residual = LeakyReLU(conv1x1(sc_x) + sc_bias), the shortcut of a ResBlock whose input has 32
// channels, computed in the epilogue from 2 K-steps of sc_x (4 record loads per lane instead of 8, and no shortcut tensor in HBM at all)
template <int MB, int RPW, int D, int RES>
__global__ __launch_bounds__(512, 2) void tail2_h8_kernel(const TailArgs a) {
  constexpr int NWAVE = 8, T = 4, PAD = 1, DIL = 2, P = D - 1;
  constexpr int C = 32 * MB, NKS = 2 * MB;
  constexpr int TW = 64, TH = NWAVE * RPW, NB = 2 * RPW;
  constexpr int LW = TW + 2 * PAD, LH = TH + 2 * PAD, REC = LH * LW;
  constexpr int NBLK_B = (2 * REC + 63) / 64, NIB = (NBLK_B + NWAVE - 1) / NWAVE;     // a2 chunk: 64-record blocks, DMA slots per wave
  constexpr int BUFREC = NBLK_B * 64;
  constexpr bool HASRES = RES == 1;
  constexpr int NKX = 2;                                                               // K-steps of the shortcut conv (32 input channels)
  constexpr int NA1 = NKS * NB, NRES = RES == 1 ? MB * NB * 2 : (RES == 2 ? NKX * NB : 0), NST = MB * NB * 2;      // whole 16-byte records per lane, see epilogue B
  constexpr int CA = NKS / 2;                                                          // position that consumes a1 and loads the residual
  constexpr int A1WAIT = CA * NIB;                                                     // DMAs issued after the a1 loads, before their use
  constexpr int RESWAIT = (NKS - 1 - CA) * NIB;                                        // ... after the residual loads, before the epilogue
  static_assert(P >= 1 && P <= NKS, "ring depth");
  static_assert(NA1 == 8 && MB * NB == 4, "the counted waits below name their registers");

  // NATIVE vector loads of the epilogue constants: the waitcnt pass guards an LDS read that carries no TBAA tag (a HIP float4 / uint4
  // struct copied by value) with s_waitcnt vmcnt(0) while any LDS-DMA is in flight -- which would drain the ring; tagged reads are left
  // to the counted waits of this kernel
  __shared__ __attribute__((aligned(16))) float s_epi[7 * C];
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_w2 = reinterpret_cast<uint4*>(smem);                     // [MB][NKS][4][64]
  uint4* s_w1 = s_w2 + MB * NKS * T * 64;                           // [MB][3 NKS][64], the a3 third K-permuted
  uint4* s_ws = s_w1 + MB * 3 * NKS * 64;                           // RES == 2: [MB][NKX][64] shortcut weights
  uint4* s_ring = s_ws + (RES == 2 ? MB * NKX * 64 : 0);            // [D][BUFREC]
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
    s_epi[tid] = a.biasA ? a.biasA[tid] : 0.0f;
    s_epi[C + tid] = a.bnA_a ? a.bnA_a[tid] : 1.0f;
    s_epi[2 * C + tid] = a.bnA_a ? a.bnA_b[tid] : 0.0f;
    s_epi[3 * C + tid] = a.biasB ? a.biasB[tid] : 0.0f;
    s_epi[4 * C + tid] = a.bnB_a ? a.bnB_a[tid] : 1.0f;
    s_epi[5 * C + tid] = a.bnB_a ? a.bnB_b[tid] : 0.0f;
    s_epi[6 * C + tid] = (RES == 2 && a.sc_bias) ? a.sc_bias[tid] : 0.0f;
  }
  if constexpr (RES == 2)
    for (int blk = wave; blk < MB * NKX; blk += NWAVE) SLU_GLDS16_T(a.sc_w + (size_t)blk * 64 + lane, s_ws + blk * 64);
  for (int blk = wave; blk < MB * NKS * T; blk += NWAVE) SLU_GLDS16_T(a.w2 + (size_t)blk * 64 + lane, s_w2 + blk * 64);
  for (int blk = wave; blk < MB * 3 * NKS; blk += NWAVE) {
    const int k = blk % (3 * NKS);
    if (k < 2 * NKS) {
      SLU_GLDS16_T(a.w1 + (size_t)blk * 64 + lane, s_w1 + blk * 64);
    } else {   // K-step over a3 in accumulator order: elements 4 hh .. 4 hh + 3 of the standard fragments of lanes (row, 0) and (row, 1)
      const uint2* w = reinterpret_cast<const uint2*>(a.w1 + (size_t)blk * 64);
      const uint2 lo = w[jj * 2 + hh], hi = w[(32 + jj) * 2 + hh];
      s_w1[blk * 64 + lane] = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
  }

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
  // K-step c of a2 with its halo into ring slot `slot`.  `valid` false (no such tile): the same DMA instructions, every lane copying the
  // zero record
  auto stage = [&](const TilePos& tp, int c, int slot, bool valid) {
    uint4* db = s_ring + slot * BUFREC;
    const uintptr_t zero = reinterpret_cast<uintptr_t>(&g_zero_rec_t);
    const uintptr_t base0 = reinterpret_cast<uintptr_t>(a.a2) +
                            16 * ((long long)(((size_t)tp.n * a.G + 2 * c) * HW) + (long long)(tp.y0 - PAD) * a.W + (tp.x0 - PAD));
    const uintptr_t base1 = base0 + 16 * (long long)HW;
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
      const int blk = i * NWAVE + wave;
      const int rc = pc_rc[i];
      const int gy = tp.y0 - PAD + (rc & 255), gx = tp.x0 - PAD + ((rc >> 8) & 255);
      const bool ok = valid && (rc >> 17) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      const uintptr_t p = ok ? (((rc >> 16) & 1) ? base1 : base0) + (long long)pc_off[i] : zero;
      SLU_GLDS16_T(reinterpret_cast<const uint4*>(p), (NBLK_B % NWAVE == 0 || blk < NBLK_B) ? db + blk * 64 : s_trash);
    }
    asm volatile("" ::: "memory");
  };

  const int bbase2 = hh * REC + (wn * RPW) * LW + jj;
  TilePos cur = decode(t_beg), nxt = cur;
  bool has_next = t_beg + t_step < t_end;
  if (has_next) nxt = decode(t_beg + t_step);
#pragma unroll
  for (int c = 0; c < P; ++c) stage(cur, c, c, !(a.dbg & 1));
  int rslot = 0, wslot = P % D;
  bool first = true;
  const float2v slA = {a.slopeA, a.slopeA}, slB = {a.slopeB, a.slopeB}, slS = {a.slopeS, a.slopeS};
  const f32x4v* se4p = reinterpret_cast<const f32x4v*>(s_epi) + hh;
  auto se4 = [&](int k) { return se4p[k]; };

  for (int tile = t_beg; tile < t_end; tile += t_step) {
    f32x16 acc3[MB][NB], acco[MB][NB];
    u32x4v a1r[NKS][NB];
    u32x4v res[HASRES ? MB : 1][HASRES ? NB : 1][2];
    u32x4v xs[RES == 2 ? NKX : 1][RES == 2 ? NB : 1];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc3[i][b][r] = 0.0f; acco[i][b][r] = 0.0f; }
    // this lane's records inside image cur.n: channel block 0 (+ hh), pixel of accumulator block b; lanes outside the image read record 0
    unsigned voff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj;
      voff[b] = (gy < a.H && gx < a.W) ? (unsigned)(((size_t)gy * a.W + gx) << 4) : 0u;
    }
    const unsigned long long img16 = 16ull * ((size_t)cur.n * a.G * HW);

    auto position = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr int YOUNG = tail2_younger<NKS, NIB, P, NA1, NRES, NST>(c);
      static_assert(YOUNG <= 63, "vmcnt is a 6-bit counter");
      if (first && c < P) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the first tile's prologue (and the resident weights)
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNG) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        constexpr int cn = (c + P) % NKS;
        if (c + P < NKS) stage(cur, cn, wslot, !(a.dbg & 1));
        else stage(nxt, cn, wslot, has_next && !(a.dbg & 1));
        wslot = wslot + 1 == D ? 0 : wslot + 1;
      }
      if constexpr (c == 0) {       // a1: K-step k of pixel block b = the record of channel block 2 k + hh
        const unsigned long long base = reinterpret_cast<unsigned long long>(a.a1) + img16;
#pragma unroll
        for (int k = 0; k < NKS; ++k)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const unsigned vo = voff[b] + (unsigned)((2 * k + hh) * HW * 16);
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(a1r[k][b]) : "v"(vo), "s"(base) : "memory");
          }
      }
      if constexpr (c == CA) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(a1r[0][0]), "+v"(a1r[0][1]), "+v"(a1r[(2 / NB) % NKS][2 % NB]), "+v"(a1r[(3 / NB) % NKS][3 % NB]),
                       "+v"(a1r[(4 / NB) % NKS][4 % NB]), "+v"(a1r[(5 / NB) % NKS][5 % NB]), "+v"(a1r[(6 / NB) % NKS][6 % NB]), "+v"(a1r[(7 / NB) % NKS][7 % NB])
                     : "n"(A1WAIT));
        if (!(a.dbg & 2)) {
#pragma unroll
          for (int k = 0; k < NKS; ++k) {
            half8 af[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) af[i] = __builtin_bit_cast(half8, s_w1[(i * 3 * NKS + k) * 64 + lane]);      // cat order (a1, a2, a3)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int i = 0; i < MB; ++i) acco[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], __builtin_bit_cast(half8, a1r[k][b]), acco[i][b], 0, 0, 0);
          }
        }
        if constexpr (RES == 2) {     // the block's input: K-step k of pixel block b = the record of channel block 2 k + hh (as for a1)
          const unsigned long long base = reinterpret_cast<unsigned long long>(a.sc_x) + 16ull * ((size_t)cur.n * (2 * NKX) * HW);
#pragma unroll
          for (int k = 0; k < NKX; ++k)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const unsigned vo = voff[b] + (unsigned)((2 * k + hh) * HW * 16);
              asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(xs[k][b]) : "v"(vo), "s"(base) : "memory");
            }
        }
        if constexpr (HASRES) {       // whole records: lane (jj, hh) loads record 2 pr + hh of channel block i (un-swapped in epilogue B)
          const unsigned long long base = reinterpret_cast<unsigned long long>(a.resid) + img16;
#pragma unroll
          for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
              for (int pr = 0; pr < 2; ++pr) {
                const unsigned vo = voff[b] + (unsigned)((i * 4 + 2 * pr + hh) * HW * 16);
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(res[i][b][pr]) : "v"(vo), "s"(base) : "memory");
              }
        }
      }
      const uint4* sb = s_ring + rslot * BUFREC + bbase2;
      rslot = rslot + 1 == D ? 0 : rslot + 1;
      if (!(a.dbg & 2)) {
        // 5 steps (4 dilated taps -> acc3, the centre tap = the 1x1 over a2 -> acc_out), fragments double-buffered: the reads of step
        // s + 1 are issued before the MFMAs of step s, in THIS order (the compiler's own schedule is read, wait, MFMA, read, wait, ...)
        half8 fa[2][MB], fb[2][NB];
        auto rd = [&](int st, int buf) {
          const int dy = st < T ? (st >> 1) * DIL : PAD, dx = st < T ? (st & 1) * DIL : PAD;
#pragma unroll
          for (int i = 0; i < MB; ++i)
            fa[buf][i] = __builtin_bit_cast(half8, st < T ? s_w2[((i * NKS + c) * T + st) * 64 + lane] : s_w1[(i * 3 * NKS + NKS + c) * 64 + lane]);
#pragma unroll
          for (int b = 0; b < NB; ++b) fb[buf][b] = __builtin_bit_cast(half8, sb[((b >> 1) + dy) * LW + (b & 1) * 32 + dx]);
        };
        rd(0, 0);
#pragma unroll
        for (int st = 0; st <= T; ++st) {
          if (st < T) rd(st + 1, (st + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < MB; ++i) {
              if (st < T) acc3[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st & 1][i], fb[st & 1][b], acc3[i][b], 0, 0, 0);
              else acco[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[st & 1][i], fb[st & 1][b], acco[i][b], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (c == NKS - 1) {
        // epilogue A in registers: a3 = bnA(act(acc3 + biasA)) rounded to fp16 (the rounding the unfused path applies when it stores a3)
        // is the B operand of K-step k = 2 i + p of the 1x1 over a3 (channel order of the accumulator, see the head of this kernel)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            half8 wf[MB];
#pragma unroll
            for (int io = 0; io < MB; ++io) wf[io] = __builtin_bit_cast(half8, s_w1[(io * 3 * NKS + 2 * NKS + 2 * i + p) * 64 + lane]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              unsigned u[4];
#pragma unroll
              for (int q2 = 0; q2 < 2; ++q2) {
                const int q = 2 * p + q2;
                const int c4 = (i * 32 + 8 * q) / 4;
                const f32x4v bi = se4(c4), ba = se4(C / 4 + c4), bb = se4(2 * C / 4 + c4);
                float2v t0 = {acc3[i][b][4 * q], acc3[i][b][4 * q + 1]}, t1 = {acc3[i][b][4 * q + 2], acc3[i][b][4 * q + 3]};
                t0 += float2v{bi.x, bi.y};
                t1 += float2v{bi.z, bi.w};
                t0 = __builtin_elementwise_max(t0, t0 * slA);
                t1 = __builtin_elementwise_max(t1, t1 * slA);
                t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
                t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
                u[2 * q2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v));
                u[2 * q2 + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v));
              }
              const half8 bf = __builtin_bit_cast(half8, make_uint4(u[0], u[1], u[2], u[3]));
#pragma unroll
              for (int io = 0; io < MB; ++io) acco[io][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[io], bf, acco[io][b], 0, 0, 0);
            }
          }
      }
    };
    tail2_static_for(position, std::make_integer_sequence<int, NKS>{});
    first = false;

    // ---- epilogue B: NST stores per lane (counted in tail2_younger).  The accumulator gives lane (jj, hh) HALF of each 16-byte record
    // (channels 8 q + 4 hh + 0..3 of pixel jj); v_permlane32_swap trades halves between lanes jj and jj + 32 so that lane (jj, hh) owns
    // the WHOLE record 2 pr + hh: one dwordx4 per lane, each half-wave a contiguous 512 bytes -- half the memory instructions and no
    // half-written 64-byte lines on the way to L2.  The residual comes in the same way and is un-swapped first. ----
    if constexpr (HASRES) {
      asm volatile("s_waitcnt vmcnt(%8)"
                   : "+v"(res[0][0][0]), "+v"(res[0][0][1]), "+v"(res[(1 / NB) % MB][1 % NB][0]), "+v"(res[(1 / NB) % MB][1 % NB][1]),
                     "+v"(res[(2 / NB) % MB][2 % NB][0]), "+v"(res[(2 / NB) % MB][2 % NB][1]), "+v"(res[(3 / NB) % MB][3 % NB][0]), "+v"(res[(3 / NB) % MB][3 % NB][1])
                   : "n"(RESWAIT));
    }
    if constexpr (RES == 2) {
      static_assert(NKX * NB == 4, "the wait below names its registers");
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(xs[0][0]), "+v"(xs[0][NB - 1]), "+v"(xs[NKX - 1][0]), "+v"(xs[NKX - 1][NB - 1]) : "n"(RESWAIT));
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int gy = cur.y0 + wn * RPW + (b >> 1), gx = cur.x0 + (b & 1) * 32 + jj;
        const bool ok = gy < a.H && gx < a.W;
        const size_t idx0 = ok ? ((size_t)cur.n * a.G + i * 4 + hh) * HW + (size_t)gy * a.W + gx : 0;
        f32x16 sacc;
        if constexpr (RES == 2) {      // the shortcut's 32 x 32 tile: 2 MFMAs, in the accumulator layout of acc_out
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
          for (int k = 0; k < NKX; ++k)
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, s_ws[(i * NKX + k) * 64 + lane]), __builtin_bit_cast(half8, xs[k][b]), sacc, 0, 0, 0);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          unsigned rw[4] = {0u, 0u, 0u, 0u};       // residual words: [0..1] for q = 2 pr, [2..3] for q = 2 pr + 1
          if constexpr (HASRES) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(res[i][b][pr].x, res[i][b][pr].z, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(res[i][b][pr].y, res[i][b][pr].w, false, false);
            rw[0] = s0[0]; rw[1] = s1[0]; rw[2] = s0[1]; rw[3] = s1[1];
          }
          unsigned hw[4];
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2) {
            const int q = 2 * pr + q2;
            const int c4 = (i * 32 + 8 * q) / 4;
            const f32x4v bi = se4(3 * C / 4 + c4), ba = se4(4 * C / 4 + c4), bb = se4(5 * C / 4 + c4);
            float2v t0 = {acco[i][b][4 * q], acco[i][b][4 * q + 1]}, t1 = {acco[i][b][4 * q + 2], acco[i][b][4 * q + 3]};
            t0 += float2v{bi.x, bi.y};
            t1 += float2v{bi.z, bi.w};
            t0 = __builtin_elementwise_max(t0, t0 * slB);
            t1 = __builtin_elementwise_max(t1, t1 * slB);
            t0 = t0 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
            t1 = t1 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w};
            if constexpr (HASRES) {
              t0 += __builtin_convertvector(__builtin_bit_cast(half2v, rw[2 * q2]), float2v);
              t1 += __builtin_convertvector(__builtin_bit_cast(half2v, rw[2 * q2 + 1]), float2v);
            }
            if constexpr (RES == 2) {      // rounded to fp16 where the separate launch stores the shortcut
              const f32x4v bs = se4(6 * C / 4 + c4);
              float2v u0 = {sacc[4 * q], sacc[4 * q + 1]}, u1 = {sacc[4 * q + 2], sacc[4 * q + 3]};
              u0 += float2v{bs.x, bs.y};
              u1 += float2v{bs.z, bs.w};
              u0 = __builtin_elementwise_max(u0, u0 * slS);
              u1 = __builtin_elementwise_max(u1, u1 * slS);
              t0 += __builtin_convertvector(__builtin_convertvector(u0, half2v), float2v);
              t1 += __builtin_convertvector(__builtin_convertvector(u1, half2v), float2v);
            }
            hw[2 * q2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t0, half2v));
            hw[2 * q2 + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(t1, half2v));
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(hw[0], hw[2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(hw[1], hw[3], false, false);
          uint4* dst = ok ? reinterpret_cast<uint4*>(a.out) + idx0 + (size_t)(2 * pr) * HW : &g_trash_rec_t;
          *dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    asm volatile("" ::: "memory");
    cur = nxt;
    has_next = tile + 2 * t_step < t_end;
    if (has_next) nxt = decode(tile + 2 * t_step);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the zero-record DMAs issued for the tile after the last: LDS must not be released under them
}

template <int MB, int RPW, int D, int RES>
int launch_tail2(TailArgs& a, hipStream_t st) {
  constexpr int TH = 8 * RPW, C = 32 * MB, NKS = 2 * MB;
  constexpr size_t nblk_b = (size_t)(2 * (TH + 2) * 66 + 63) / 64;
  constexpr size_t lds = ((size_t)MB * NKS * 4 * 64 + (size_t)MB * 3 * NKS * 64 + (RES == 2 ? (size_t)MB * 2 * 64 : 0) + (size_t)D * nblk_b * 64 + 64) * 16;      // + 7 C floats static
  static_assert(lds + 7 * C * 4 <= 160 * 1024, "ring does not fit in LDS");
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
  if (nt <= 0 || nt > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  long long gx = 256;
  if (gx > nt) gx = nt;
  auto kern = tail2_h8_kernel<MB, RPW, D, RES>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(512), lds, st, a);
  SLU_CHECK_LAUNCH();
}

template <int MB, int WM, int WN, int RPW, bool W3RES>
int launch_tail(TailArgs& a, hipStream_t st) {
  constexpr int TH = WN * RPW, MBLK = MB * WM, C = 32 * MBLK, NKS = 2 * MBLK;
  constexpr size_t nb_alloc = (size_t)((2 * (TH + 2) * 66 + 63) / 64) * 64;
  constexpr size_t lds = (size_t)6 * C * 4 + 2 * nb_alloc * 16 + (size_t)2 * MBLK * 5 * 64 * 16 + (W3RES ? (size_t)MBLK * NKS * 64 * 16 : 0) +
                         (size_t)(C / 8) * TH * 64 * 16;
  static_assert(lds <= 160 * 1024, "tile does not fit in LDS");
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
  if (nt <= 0 || nt > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  long long gx = 256;                                                 // one 8-wave workgroup per CU (LDS)
  if (gx > nt) gx = nt;
  auto kern = tail_h8_kernel<MB, WM, WN, RPW, W3RES>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(64 * WM * WN), lds, st, a);
  SLU_CHECK_LAUNCH();
}

}  // namespace

extern "C" int slu_conv_tail_h8_shortcut_supported(int C, int sc_cin) { return C == 64 && sc_cin == 32 ? 1 : 0; }

extern "C" int slu_conv_tail_h8_supported(int C, int H, int W) { return (C == 32 || C == 64 || C == 128) && H > 0 && W > 0 ? 1 : 0; }

extern "C" int slu_conv_tail_h8_fwd(const slu_conv_tail_h8_desc* d, slu_stream_t stream) {
  if (!d || !d->a1 || !d->a2 || !d->w2x2 || !d->w1x1 || !d->out || d->N <= 0 || d->H <= 0 || d->W <= 0) return SLU_EINVAL;
  if (((uintptr_t)d->a1 | (uintptr_t)d->a2 | (uintptr_t)d->out | (uintptr_t)d->resid | (uintptr_t)d->w2x2 | (uintptr_t)d->w1x1) & 15) return SLU_EINVAL;
  if ((d->bnA_a == nullptr) != (d->bnA_b == nullptr) || (d->bnB_a == nullptr) != (d->bnB_b == nullptr)) return SLU_EINVAL;
  if (!slu_conv_tail_h8_supported(d->C, d->H, d->W)) return SLU_EUNSUPPORTED;
  if (d->hasactA && !(d->slopeA >= 0.0f && d->slopeA <= 1.0f)) return SLU_EINVAL;      // LeakyReLU as max(t, slope t)
  if (d->hasactB && !(d->slopeB >= 0.0f && d->slopeB <= 1.0f)) return SLU_EINVAL;
  TailArgs a{};
  a.a1 = reinterpret_cast<const uint4*>(d->a1);
  a.a2 = reinterpret_cast<const uint4*>(d->a2);
  a.w2 = reinterpret_cast<const uint4*>(d->w2x2);
  a.w1 = reinterpret_cast<const uint4*>(d->w1x1);
  a.biasA = d->biasA; a.bnA_a = d->bnA_a; a.bnA_b = d->bnA_b;
  a.biasB = d->biasB; a.bnB_a = d->bnB_a; a.bnB_b = d->bnB_b;
  a.slopeA = d->hasactA ? d->slopeA : 1.0f;
  a.slopeB = d->hasactB ? d->slopeB : 1.0f;
  a.resid = reinterpret_cast<const uint2*>(d->resid);
  a.out = reinterpret_cast<uint2*>(d->out);
  a.N = d->N; a.H = d->H; a.W = d->W; a.G = d->C / 8;
  static const int dbg = [] { const char* e = getenv("SLU_TAIL_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  hipStream_t st = slu_stream(stream);
  static const bool v1 = [] { const char* e = getenv("SLU_TAIL_V1"); return e && e[0] == '1'; }();     // A/B switch: the round-1 kernel
  if (d->sc_x) {      // shortcut mode
    if (d->resid || !d->sc_w || !slu_conv_tail_h8_shortcut_supported(d->C, d->sc_cin) || (((uintptr_t)d->sc_x | (uintptr_t)d->sc_w) & 15)) return SLU_EINVAL;
    if (d->sc_hasact && !(d->sc_slope >= 0.0f && d->sc_slope <= 1.0f)) return SLU_EINVAL;
    a.sc_x = reinterpret_cast<const uint4*>(d->sc_x);
    a.sc_w = reinterpret_cast<const uint4*>(d->sc_w);
    a.sc_bias = d->sc_bias;
    a.slopeS = d->sc_hasact ? d->sc_slope : 1.0f;
    return launch_tail2<2, 1, 4, 2>(a, st);
  }
  if (!v1 && d->C == 32) return a.resid ? launch_tail2<1, 2, 3, 1>(a, st) : launch_tail2<1, 2, 3, 0>(a, st);
  if (!v1 && d->C == 64) return a.resid ? launch_tail2<2, 1, 4, 1>(a, st) : launch_tail2<2, 1, 4, 0>(a, st);
  if (d->C == 32) return launch_tail<1, 1, 8, 2, true>(a, st);
  if (d->C == 64) return launch_tail<2, 1, 8, 1, true>(a, st);
  return launch_tail<2, 2, 4, 1, false>(a, st);
}
