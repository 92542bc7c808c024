// Fused ResContextBlock of SalsaNext on the half-precision ("h8") path (reference SalsaNext.py:10-39):
//
//     s   = leaky(conv1x1(x) + b1)                                   (conv1 + act1; the block's shortcut)
//     a1  = bn1(leaky(conv3x3_pad1(s) + b2))                          (conv2, act2, bn1)
//     out = s + bn2(leaky(conv3x3_dil2_pad2(a1) + b3))                (conv3, act3, bn2, + shortcut)
//
// Unfused that is three launches moving 7 full-resolution tensor passes (x -> s; s -> a1; a1, s -> out); here s and a1 never leave
// the CU: the block reads x and writes out -- 2 passes.  All three layers have 32 output channels (the three context blocks of the
// network run at full resolution, where they are HBM-bound).
//
// Persistent workgroup of 8 waves, output tile = 8 rows x 64 columns; tiles dealt round-robin, XCD-aware (see conv2d_h8.hip).  Per
// tile three phases, each an implicit GEMM on v_mfma_f32_32x32x16_f16 over a FLATTENED pixel range (32 consecutive pixels of the
// region's row-major order = one MFMA N-block; ds_read_b128 addresses are per lane, so a block may straddle rows):
//   P1  s  on the tile + 3 halo pixels  (14 x 70 = 980 px, 31 N-blocks): B operands straight from global memory -- loaded into
//       registers one tile AHEAD (issued at the start of the previous tile's P3, so their latency hides under the MFMAs);
//       epilogue -> fp16 -> LDS image S [4 blocks][14][70] in B-operand layout.  Outside the image s = 0 (conv2's zero padding).
//   P2  a1 on the tile + 2 halo pixels  (12 x 68 = 816 px, 26 N-blocks) from S; epilogue -> fp16 -> LDS image A1 [4][12][68],
//       zero outside the image (conv3's zero padding).
//   P3  out on the tile (8 x 64, 16 N-blocks: one row per wave) from A1; epilogue adds the centre of S and stores h8.
// Rounding points (fp16 for s and a1, fp32 accumulation in the same k-step / tap order) are those of the unfused kernels, so the
// two paths agree to the last bit of fp16 except where an FMA contraction differs.  Halo recompute: 1.33 x the MFMA work.
// LDS: S 61.3 KB + A1 51 KB + all weights resident 38 KB + epilogue constants = 151 KB (one workgroup per CU).
#include <type_traits>
#include "slu_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

namespace {

struct CtxArgs {
  const uint4* x;              // h8 [N][Gin][H][W]
  const uint4 *w1, *w2, *w3;   // packed weights: [nks1][1][64], [2][9][64], [2][9][64] records
  const float *b1, *b2, *bn1_a, *bn1_b, *b3, *bn2_a, *bn2_b;      // [32] fp32 or nullptr
  float slope;                 // LeakyReLU slope of the three activations
  uint2* out;                  // h8 [N][4][H][W] as 8-byte half records
  int N, H, W, Gin;
  int tiles_x, tiles_y;
};

constexpr int TH = 8, TW = 64;
constexpr int SH = TH + 6, SW = TW + 6, SREC = SH * SW;        // s region (halo 3)
constexpr int AH = TH + 4, AW = TW + 4, AREC = AH * AW;        // a1 region (halo 2)
constexpr int NB1 = (SREC + 31) / 32, NB2 = (AREC + 31) / 32;  // N-blocks of P1 / P2
constexpr int SRECP = NB1 * 32, ARECP = NB2 * 32;              // records per channel block of the LDS images: whole N-blocks, so every lane of
                                                               // every block owns a record (the few past the region are never read)
constexpr int NWAVE = 8;
#ifndef SLU_CTX_RING
#define SLU_CTX_RING 4
#endif
constexpr int RING = SLU_CTX_RING;                             // B-operand ring: an LDS read is issued RING - 1 MFMAs before its use
constexpr int PW1 = (NB1 + NWAVE - 1) / NWAVE, PW2 = (NB2 + NWAVE - 1) / NWAVE, PW3 = 2;
// LDS map in 16-byte records from offset 0 (the kernel has no static LDS): the two images first, so that every access is
// "opaque per-lane record index + a constant below 64 KB" and needs no address register of its own
constexpr int OFF_S = 0, OFF_A = OFF_S + 4 * SRECP, OFF_W1 = OFF_A + 4 * ARECP;
template <int NKS1> constexpr int off_w2() { return OFF_W1 + NKS1 * 64; }
template <int NKS1> constexpr int off_w3() { return off_w2<NKS1>() + 18 * 64; }
template <int NKS1> constexpr int off_epi() { return off_w3<NKS1>() + 18 * 64; }      // 7 x 32 floats = 56 records
template <int NKS1> constexpr size_t lds_bytes() { return (size_t)(off_epi<NKS1>() + 56) * 16; }

#ifdef SLU_CTX_PROF      // development aid: shader clocks of wave 0 of every workgroup per phase (P1 | barrier | P2 | barrier | P3 | barrier)
__device__ unsigned long long g_ctx_prof[8];
#define CTX_PROF_MARK(i)                                             \
  {                                                                  \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();  \
    asm volatile("" ::: "memory");                                   \
    prof_acc[i] += t_now - prof_t;                                   \
    prof_t = t_now;                                                  \
  }
#else
#define CTX_PROF_MARK(i)
#endif

__device__ uint4 g_trash_rec_c;     // where lanes outside the image store (every lane of every tile issues its stores: no branch)

// a value the optimiser must treat as unknown: keeps "index + constant" LDS addresses in base-register + immediate form
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// LeakyReLU(t) = max(t, slope t) for 0 <= slope <= 1: a multiply + one v_max_f32 per element (the builtin max first canonicalises
// both operands with a v_max v, v, v each; accumulator values need no quieting)
__device__ __forceinline__ float2v leaky2(float2v t, float2v sl) {
  const float2v m = t * sl;
  float2v r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r[0]) : "v"(t[0]), "v"(m[0]));
  asm("v_max_f32 %0, %1, %2" : "=v"(r[1]) : "v"(t[1]), "v"(m[1]));
  return r;
}

__device__ __forceinline__ unsigned pack2h(float2v t) { return __builtin_bit_cast(unsigned, __builtin_convertvector(t, half2v)); }

template <int NKS1>
__global__ __launch_bounds__(64 * NWAVE, 2) void ctx_h8_kernel(const CtxArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* lds = reinterpret_cast<uint4*>(smem);
  uint2* lds2 = reinterpret_cast<uint2*>(smem);
  float4* lds4 = reinterpret_cast<float4*>(smem);
  constexpr int OFF_W2 = off_w2<NKS1>(), OFF_W3 = off_w3<NKS1>(), OFF_EPI = off_epi<NKS1>();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

  {
    float* s_epi = reinterpret_cast<float*>(lds + OFF_EPI);      // b1 | b2 | bn1_a | bn1_b | b3 | bn2_a | bn2_b   (7 x 32 floats)
    if (tid < 32) {
      s_epi[tid] = a.b1 ? a.b1[tid] : 0.0f;
      s_epi[32 + tid] = a.b2 ? a.b2[tid] : 0.0f;
      s_epi[64 + tid] = a.bn1_a ? a.bn1_a[tid] : 1.0f;
      s_epi[96 + tid] = a.bn1_a ? a.bn1_b[tid] : 0.0f;
      s_epi[128 + tid] = a.b3 ? a.b3[tid] : 0.0f;
      s_epi[160 + tid] = a.bn2_a ? a.bn2_a[tid] : 1.0f;
      s_epi[192 + tid] = a.bn2_a ? a.bn2_b[tid] : 0.0f;
    }
    for (int e = tid; e < NKS1 * 64; e += 64 * NWAVE) lds[OFF_W1 + e] = a.w1[e];
    for (int e = tid; e < 18 * 64; e += 64 * NWAVE) {
      lds[OFF_W2 + e] = a.w2[e];
      lds[OFF_W3 + e] = a.w3[e];
    }
  }

  // Per-lane geometry of the N-blocks this wave owns (tile independent).  P1 / P2: blocks wave, wave + 8, ... of the flattened
  // region; lanes past the region's last pixel compute on a clamped pixel and write into the padding records.  P3: row `wave`.
  int p1_rc[PW1], p1_goff[PW1], p1_w[PW1];     // r | c << 8 | inside-region << 16 ; r * W + c ; uint2 index of the S record half this lane writes
  int p2_rc[PW2], p2_r[PW2], p2_w[PW2];        // the same for a1 ; record index of tap (0, 0) in channel block hh of S ; uint2 index written in A1
#pragma unroll
  for (int i = 0; i < PW1; ++i) {
    const int e = 32 * (wave + NWAVE * i) + jj;
    const int ec = e < SREC ? e : SREC - 1;
    const int r = ec / SW, c = ec - r * SW;
    p1_rc[i] = r | (c << 8) | ((e < SREC ? 1 : 0) << 16);
    p1_goff[i] = r * a.W + c;
    p1_w[i] = opaque(((OFF_S + (e < SRECP ? e : 0)) << 1) + hh);
  }
#pragma unroll
  for (int i = 0; i < PW2; ++i) {
    const int e = 32 * (wave + NWAVE * i) + jj;
    const int ec = e < AREC ? e : AREC - 1;
    const int r = ec / AW, c = ec - r * AW;
    p2_rc[i] = r | (c << 8);
    p2_r[i] = opaque(OFF_S + hh * SRECP + r * SW + c);
    p2_w[i] = opaque(((OFF_A + (e < ARECP ? e : 0)) << 1) + hh);
  }
  const int p3_r = opaque(OFF_A + hh * ARECP + wave * AW + jj);          // tap (0, 0) of row `wave`, column jj of the tile in channel block hh of A1
  const int p3_s = opaque(((OFF_S + (wave + 3) * SW + jj + 3) << 1) + hh);  // the same pixel's half record in S (the shortcut)
  const int i_w1 = opaque(OFF_W1 + lane), i_w2 = opaque(OFF_W2 + lane), i_w3 = opaque(OFF_W3 + lane), i_epi = opaque(OFF_EPI + hh);

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
  // B operands of P1 for tile tp: lane (pixel jj, half hh) of block i, K-step k holds channel block 2 k + hh of its pixel (zero
  // outside the image / past the last channel block)
  uint4 xr[PW1][NKS1];
  auto load_x = [&](const TilePos& tp) {
    const long long org = (long long)(tp.y0 - 3) * a.W + (tp.x0 - 3);
#pragma unroll
    for (int i = 0; i < PW1; ++i) {
      const int rc = p1_rc[i];
      const int gy = tp.y0 - 3 + (rc & 255), gx = tp.x0 - 3 + ((rc >> 8) & 255);
      const bool in = (rc >> 16) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W && wave + NWAVE * i < NB1;
#pragma unroll
      for (int k = 0; k < NKS1; ++k) {
        const int g = 2 * k + hh;
        const bool ok = in && g < a.Gin;
        const uint4* p = a.x + (ok ? (long long)(((size_t)tp.n * a.Gin + g) * HW) + org + p1_goff[i] : 0);
        const uint4 v = *p;
        xr[i][k] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };

  const float2v sl = {a.slope, a.slope};
  // this lane's 16 biases of a layer (channel 8 q + 4 hh + k at [4 q + k]): the C operand of a block's first MFMA
  auto lane_bias = [&](int first_f4) {
    f32x16 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = lds4[i_epi + first_f4 + 2 * q];
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    return v;
  };

  TilePos cur = decode(t_beg);
  load_x(cur);
  __syncthreads();                                               // weights + constants visible
#ifdef SLU_CTX_PROF
  unsigned long long prof_acc[6] = {0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
#endif

  for (int tile = t_beg; tile < t_end; tile += t_step) {
    // ---------------- P1: s = leaky(conv1x1(x) + b1) on the 14 x 70 region -> S (zero outside the image: conv2's padding) ----------------
    auto phase1 = [&](auto nl_c) {
      constexpr int NL = decltype(nl_c)::value;
      const f32x16 biasv = lane_bias(0);
      half8 af[NKS1];
#pragma unroll
      for (int k = 0; k < NKS1; ++k) af[k] = __builtin_bit_cast(half8, lds[i_w1 + k * 64]);
      f32x16 acc[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int k = 0; k < NKS1; ++k)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[k], __builtin_bit_cast(half8, xr[i][k]), k == 0 ? biasv : acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int rc = p1_rc[i];
        const int gy = cur.y0 - 3 + (rc & 255), gx = cur.x0 - 3 + ((rc >> 8) & 255);
        const bool in = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float2v t0 = leaky2(float2v{acc[i][4 * q], acc[i][4 * q + 1]}, sl), t1 = leaky2(float2v{acc[i][4 * q + 2], acc[i][4 * q + 3]}, sl);
          lds2[p1_w[i] + q * SRECP * 2] = in ? make_uint2(pack2h(t0), pack2h(t1)) : make_uint2(0u, 0u);
        }
      }
    };
    static_assert(NB1 > (PW1 - 1) * NWAVE && NB1 <= PW1 * NWAVE, "every wave owns PW1 - 1 or PW1 blocks");
    if (wave < NB1 - (PW1 - 1) * NWAVE) phase1(std::integral_constant<int, PW1>{});
    else phase1(std::integral_constant<int, PW1 - 1>{});
    CTX_PROF_MARK(0)
    __syncthreads();
    CTX_PROF_MARK(1)

    // ---------------- P2: a1 = bn1(leaky(conv3x3(s) + b2)) on the 12 x 68 region -> A1 (zero outside the image) ----------------
    // The 18 weight fragments live in registers for the phase (one LDS read per MFMA: the B operand, issued 3 MFMAs ahead into a ring
    // of 4).  Blocks run one after the other on two alternating accumulator tiles; the epilogue of block i - 1 is cut into 16 pieces of
    // 2 - 5 vector instructions that are issued behind the first 16 MFMAs of block i, in program order pinned by sched_barrier: the
    // matrix pipe never waits for an epilogue and the epilogue never waits for the matrix pipe (hipcc does not find this interleave).
    auto phase2 = [&](auto nl_c) {
      constexpr int NL = decltype(nl_c)::value, G = 18 * NL;
      const f32x16 biasv = lane_bias(8);
      half8 af[18];                                                  // read behind the MFMAs of block 0 (no 18 KB burst per wave at the phase start)
      auto read_a = [&](int t) { af[t] = __builtin_bit_cast(half8, lds[i_w2 + t * 64]); };
      f32x16 acc[2];
      half8 bq[RING];
      unsigned hp[8];                                                // the block's 16 results as 8 packed fp16 pairs
      float4 ba4[2], bb4[2];                                         // folded BatchNorm of the q being finished and of the next one (LDS table)
      float2v tp2;
      auto read_b = [&](int g) {
        const int i = g / 18, m = g % 18, k = m / 9, tap = m % 9;
        bq[g % RING] = __builtin_bit_cast(half8, lds[p2_r[i] + 2 * k * SRECP + (tap / 3) * SW + (tap % 3)]);
      };
      // the per-channel constants a q needs are read four pieces (= MFMA slots) before their first use: an LDS round trip under load
      // is longer than one slot, and a wait in the epilogue stream stalls the MFMAs behind it
      auto prefetch_q = [&](int q) {
        ba4[q & 1] = lds4[i_epi + 16 + 2 * q];
        bb4[q & 1] = lds4[i_epi + 24 + 2 * q];
      };
      // piece m (0..15) of block ib's epilogue: pair p = m / 2; even m: LeakyReLU, odd m: BatchNorm + fp16 pair (+ the LDS write of a finished q)
      auto epi_piece = [&](int ib, int m) {
        const int pr = m >> 1, q = pr >> 1;
        const f32x16& ac = acc[ib & 1];
        if ((m & 1) == 0) {
          if ((pr & 1) == 0 && q < 3) prefetch_q(q + 1);
          tp2 = leaky2(float2v{ac[2 * pr], ac[2 * pr + 1]}, sl);
        } else {
          const float4 ba = ba4[q & 1], bb = bb4[q & 1];
          tp2 = (pr & 1) ? tp2 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w} : tp2 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
          hp[pr] = pack2h(tp2);
          if (pr & 1) {
            const int rc = p2_rc[ib];
            const int gy = cur.y0 - 2 + (rc & 255), gx = cur.x0 - 2 + (rc >> 8);
            const bool in = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            lds2[p2_w[ib] + q * ARECP * 2] = in ? make_uint2(hp[pr - 1], hp[pr]) : make_uint2(0u, 0u);
          }
        }
      };
#pragma unroll
      for (int g = 0; g < RING - 1; ++g) { read_a(g); read_b(g); }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int i = g / 18, m = g % 18;
        if (g + RING - 1 < 18) read_a(g + RING - 1);
        if (g + RING - 1 < G) read_b(g + RING - 1);
        acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bq[g % RING], m == 0 ? biasv : acc[i & 1], 0, 0, 0);
        if (i > 0 && m < 16) epi_piece(i - 1, m);
        if (m == 16) prefetch_q(0);                                  // for the epilogue of block i, which starts two slots from here
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) epi_piece(NL - 1, m);
    };
    static_assert(NB2 > (PW2 - 1) * NWAVE && NB2 <= PW2 * NWAVE, "every wave owns PW2 - 1 or PW2 blocks");
    if (wave < NB2 - (PW2 - 1) * NWAVE) phase2(std::integral_constant<int, PW2>{});
    else phase2(std::integral_constant<int, PW2 - 1>{});
    CTX_PROF_MARK(2)
    __syncthreads();
    CTX_PROF_MARK(3)

    // the next tile's x goes into registers now: its latency hides under P3
    const bool more = tile + t_step < t_end;
    const TilePos nxt = more ? decode(tile + t_step) : cur;
    if (more) load_x(nxt);

    // ---------------- P3: out = s + bn2(leaky(conv3x3_dil2(a1) + b3)) on the 8 x 64 tile (same pipeline, two blocks) ----------------
    {
      constexpr int G = 18 * PW3;
      const f32x16 biasv = lane_bias(32);
      half8 af[18];
      auto read_a = [&](int t) { af[t] = __builtin_bit_cast(half8, lds[i_w3 + t * 64]); };
      f32x16 acc[2];
      half8 bq[RING];
      unsigned hp[8];
      float4 ba4[2], bb4[2];
      uint2 sv[2];
      float2v tp2;
      const int gy = cur.y0 + wave;
      auto read_b = [&](int g) {
        const int i = g / 18, m = g % 18, k = m / 9, tap = m % 9;
        bq[g % RING] = __builtin_bit_cast(half8, lds[p3_r + 2 * k * ARECP + 32 * i + (tap / 3) * 2 * AW + (tap % 3) * 2]);
      };
      auto prefetch_q = [&](int ib, int q) {
        sv[q & 1] = lds2[p3_s + (q * SRECP + 32 * ib) * 2];         // shortcut values of this q (two pairs)
        ba4[q & 1] = lds4[i_epi + 40 + 2 * q];
        bb4[q & 1] = lds4[i_epi + 48 + 2 * q];
      };
      auto epi_piece = [&](int ib, int m) {
        const int pr = m >> 1, q = pr >> 1;
        const f32x16& ac = acc[ib & 1];
        if ((m & 1) == 0) {
          if ((pr & 1) == 0 && q < 3) prefetch_q(ib, q + 1);
          tp2 = leaky2(float2v{ac[2 * pr], ac[2 * pr + 1]}, sl);
        } else {
          const float4 ba = ba4[q & 1], bb = bb4[q & 1];
          tp2 = (pr & 1) ? tp2 * float2v{ba.z, ba.w} + float2v{bb.z, bb.w} : tp2 * float2v{ba.x, ba.y} + float2v{bb.x, bb.y};
          tp2 += __builtin_convertvector(__builtin_bit_cast(half2v, (pr & 1) ? sv[q & 1].y : sv[q & 1].x), float2v);
          hp[pr] = pack2h(tp2);
          if (pr & 1) {
            const int gx = cur.x0 + 32 * ib + jj;
            const bool ok = gy < a.H && gx < a.W;
            uint2* dst = ok ? a.out + (((((size_t)cur.n * 4 + q) * HW + (size_t)gy * a.W + gx) << 1) + hh) : reinterpret_cast<uint2*>(&g_trash_rec_c);
            *dst = make_uint2(hp[pr - 1], hp[pr]);
          }
        }
      };
#pragma unroll
      for (int g = 0; g < RING - 1; ++g) { read_a(g); read_b(g); }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int i = g / 18, m = g % 18;
        if (g + RING - 1 < 18) read_a(g + RING - 1);
        if (g + RING - 1 < G) read_b(g + RING - 1);
        acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[m], bq[g % RING], m == 0 ? biasv : acc[i & 1], 0, 0, 0);
        if (i > 0 && m < 16) epi_piece(i - 1, m);
        if (m == 16) prefetch_q(i, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int m = 0; m < 16; ++m) epi_piece(PW3 - 1, m);
    }
    CTX_PROF_MARK(4)
    __syncthreads();                                             // S and A1 are free for the next tile
    CTX_PROF_MARK(5)
    cur = nxt;
  }
#ifdef SLU_CTX_PROF
  if (tid == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&g_ctx_prof[i], prof_acc[i]);
    atomicAdd(&g_ctx_prof[6], 1ull);
  }
#endif
}

template <int NKS1>
int launch_ctx(CtxArgs& a, hipStream_t st) {
  constexpr size_t lds = lds_bytes<NKS1>();
  static_assert(lds <= 160 * 1024, "tile does not fit in LDS");
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long nt = (long long)a.tiles_x * a.tiles_y * a.N;
  if (nt <= 0 || nt > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  long long gx = 256;                                            // one 8-wave workgroup per CU (LDS)
  if (gx > nt) gx = nt;
  auto kern = ctx_h8_kernel<NKS1>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(64 * NWAVE), lds, st, a);
  SLU_CHECK_LAUNCH();
}

}  // namespace

#ifdef SLU_CTX_PROF
extern "C" int slu_ctx_prof_read(unsigned long long* out8) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ctx_prof), sizeof(unsigned long long) * 8) != hipSuccess) return SLU_ELAUNCH;
  unsigned long long z[8] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_ctx_prof), z, sizeof(z)) == hipSuccess ? SLU_OK : SLU_ELAUNCH;
}
#endif

extern "C" int slu_ctx_block_h8_supported(int Cin, int C, int H, int W) { return (Cin >= 1 && Cin <= 32 && C == 32 && H > 0 && W > 0 && W < (1 << 20)) ? 1 : 0; }

extern "C" int slu_ctx_block_h8_fwd(const slu_ctx_block_h8_desc* d, slu_stream_t stream) {
  if (!d || !d->x || !d->w1 || !d->w2 || !d->w3 || !d->out || d->N <= 0 || d->H <= 0 || d->W <= 0) return SLU_EINVAL;
  if (((uintptr_t)d->x | (uintptr_t)d->out | (uintptr_t)d->w1 | (uintptr_t)d->w2 | (uintptr_t)d->w3) & 15) return SLU_EINVAL;
  if ((d->bn1_a == nullptr) != (d->bn1_b == nullptr) || (d->bn2_a == nullptr) != (d->bn2_b == nullptr)) return SLU_EINVAL;
  if (!slu_ctx_block_h8_supported(d->Cin, d->C, d->H, d->W)) return SLU_EUNSUPPORTED;
  if (!(d->slope >= 0.0f && d->slope <= 1.0f)) return SLU_EINVAL;      // LeakyReLU as max(t, slope t)
  if (d->x == d->out) return SLU_EINVAL;                                  // tiles read their neighbours' halo
  CtxArgs a{};
  a.x = reinterpret_cast<const uint4*>(d->x);
  a.w1 = reinterpret_cast<const uint4*>(d->w1);
  a.w2 = reinterpret_cast<const uint4*>(d->w2);
  a.w3 = reinterpret_cast<const uint4*>(d->w3);
  a.b1 = d->bias1; a.b2 = d->bias2; a.bn1_a = d->bn1_a; a.bn1_b = d->bn1_b; a.b3 = d->bias3; a.bn2_a = d->bn2_a; a.bn2_b = d->bn2_b;
  a.slope = d->slope;
  a.out = reinterpret_cast<uint2*>(d->out);
  a.N = d->N; a.H = d->H; a.W = d->W; a.Gin = (d->Cin + 7) / 8;
  hipStream_t st = slu_stream(stream);
  return d->Cin <= 16 ? launch_ctx<1>(a, st) : launch_ctx<2>(a, st);
}
