// Split-fp16 ("f16x3") variant of the fused conv (same fusion, same launch arguments, same epilogue as
// conv2d.hip): every fp32 operand is split x = hi + lo with hi = fp16(x), lo = fp16(x - hi), and each K=16
// step issues three v_mfma_f32_32x32x16_f16 (hi*hi + hi*lo + lo*hi, fp32 accumulate).  The dropped lo*lo
// term is ~2^-22 relative, so results agree with the exact-fp32 kernel to fp32-level accuracy while the
// matrix-core time drops 5.3x (3 x 32 cycles per 16 channels x 32 x 32 instead of 8 x 64), which moves the
// 3x3 / 2x2 convs of the range-image stack from MFMA-bound towards HBM-bound.
//
// LDS images (channel-innermost so one ds_read_b128 is one MFMA operand):
//   B: [2 groups of 8 channels][rows][cols][8 x fp16]  hi and lo  -- lane (pixel r, half h) reads record
//      (group h, pixel r + tap offset): consecutive lanes = consecutive 16-byte records, conflict-free
//   A: [channel block][tap][64 lanes][8 x fp16] hi and lo, pre-packed by slu_pack_conv_weight_f16x3
// Staging converts fp32 -> (hi, lo) on the fly: a thread loads 8 channels x 4 adjacent pixels (eight 16-byte
// loads, each coalesced along azimuth), and writes four 16-byte records per image in a rotated order so the
// eight lanes of a ds_write_b128 group hit distinct banks.
#include "conv_common.h"

using namespace slu_conv;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int CK16 = 16;

// x = hi + lo with hi = fp16(x) rounded toward zero (v_cvt_pkrtz_f16_f32 converts two values per instruction)
// and lo = fp16(x - hi): 3 VALU per element; hi carries 11 bits, lo at least 10 more.
__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const auto a = __builtin_amdgcn_cvt_pkrtz(x[2 * k], x[2 * k + 1]);
    const auto b = __builtin_amdgcn_cvt_pkrtz(x[2 * k] - (float)a[0], x[2 * k + 1] - (float)a[1]);
    h[k] = __builtin_bit_cast(unsigned, a);
    l[k] = __builtin_bit_cast(unsigned, b);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int KS, int DIL, int PAD, int MB, int WM, int WN, int RPW, bool GEN>
__global__ __launch_bounds__(64 * WM * WN, (MB * RPW >= 4) ? 1 : 2) void conv_f16x3_kernel(const ConvArgs a, const float* __restrict__ resid,
                                                                                           float* __restrict__ out) {
  constexpr int NT = 64 * WM * WN;
  constexpr int T = KS * KS;
  constexpr int TW = 64, TH = WN * RPW, NB = 2 * RPW;
  constexpr int XO = PAD ? 4 : 0;
  constexpr int LW = TW + 2 * XO, LH = TH + 2 * PAD, LW4 = LW / 4;
  constexpr int LWP = LW + LW / 8;                  // one pad record after every 8: quads of neighbouring lanes start 64,80,64,80..
                                                    // bytes apart, so the 8 lanes of a ds_write_b128 group hit distinct banks
  constexpr int REC = LH * LWP;                     // 16-byte record slots per 8-channel group
  constexpr int MBLK = WM * MB;
  constexpr int NITEM = 2 * LH * LW4, NI = (NITEM + NT - 1) / NT;       // (group, row, 4-pixel quad) items
  constexpr int NWV = MBLK * T * 128, NW = (NWV + NT - 1) / NT;          // uint4 items of the weight tile

  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_bh = reinterpret_cast<uint4*>(smem);     // [2][REC]
  uint4* s_bl = s_bh + 2 * REC;
  uint4* s_ah = s_bl + 2 * REC;                     // [MBLK][T][64]
  uint4* s_al = s_ah + MBLK * T * 64;
  float* s_epi = reinterpret_cast<float*>(s_al + MBLK * T * 64);        // bias | bn_a | bn_b

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
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

  if (tid < MBLK * 32) {
    const int co = mblk0 * 32 + tid;
    const bool ok = co < a.Cout;
    s_epi[tid] = (ok && a.bias) ? a.bias[co] : 0.0f;
    s_epi[MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_a[co] : 1.0f;
    s_epi[2 * MBLK * 32 + tid] = (ok && a.bn_a) ? a.bn_b[co] : 0.0f;
  }

  const int hh = lane >> 5, jj = lane & 31;
  int bcol[KS];                                     // padded slot of (first column + lane + tap shift), per horizontal tap
#pragma unroll
  for (int tj = 0; tj < KS; ++tj) {
    const int c = (XO - PAD) + jj + tj * DIL;
    bcol[tj] = hh * REC + (wn * RPW) * LWP + c + (c >> 3);
  }
  const int abase = (wm * MB) * T * 64 + lane;
  const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpack);

  // The raw fp32 tile of chunk q+1 is fetched into registers (all loads of a thread back to back) before the
  // MFMA phase of chunk q and converted / written to LDS after the barrier that retires chunk q's LDS reads,
  // so HBM latency overlaps the matrix-core phase.
  Item<GEN> st[NI][8];
  unsigned okm[NI], psm[NI];
  auto fetch_tile = [&](int q, int tq) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tq + i * NT;
      const int g2 = e / (LH * LW4);
      const int rem = e - g2 * (LH * LW4);
      const int r = rem / LW4;
      const int c4 = rem - r * LW4;
      const int gy = y0 + r - PAD, gx4 = x0 - XO + 4 * c4;
      const bool pix_ok = (NITEM % NT == 0 || e < NITEM) && gy >= 0 && gy < a.H && gx4 >= 0 && gx4 < a.W;
      okm[i] = 0; psm[i] = 0;
      bool fast = false;
      if constexpr (!GEN) {
        // common case: the 8 channels of this item are 8 consecutive channels of ONE plain source and the quad is
        // inside the image: one base address, constant channel stride, no per-channel selects
        const int cg0 = q * CK16 + g2 * 8;
        const SrcPick p0 = pick_src(a, im, (pix_ok && cg0 < a.Cin) ? cg0 : 0);
        fast = pix_ok && cg0 + 8 <= a.Cin && p0.cl + 8 <= p0.C;
        if (fast) {
          const size_t cs = (size_t)a.H * a.W;
          const float* bp = p0.ptr + (((size_t)p0.ns * p0.C + p0.cl) * a.H + gy) * a.W + gx4;
#pragma unroll
          for (int k = 0; k < 8; ++k) st[i][k].v = *reinterpret_cast<const float4*>(bp + k * cs);
          okm[i] = 0xFFu;
        }
      }
      if (!fast) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int cg = q * CK16 + g2 * 8 + k;
          const bool ok = pix_ok && cg < a.Cin;
          bool ps;
          fetch_item<GEN>(a, im, n, ok ? cg : 0, gy, gx4, ok, st[i][k], ps);
          okm[i] |= (ok ? 1u : 0u) << k;
          psm[i] |= (ps ? 1u : 0u) << k;
        }
      }
    }
  };
  if (a.vec) fetch_tile(0, tid);

  for (int q = 0; q < a.nchunks; ++q) {
    __syncthreads();
    int tq = tid;
    asm volatile("" : "+v"(tq));
    // ---- weight fragments of this chunk: MBLK contiguous runs of T*128 uint4 ([tap][hi|lo][lane]) ----
    uint4 sw[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tq + i * NT;
      const int m = e / (T * 128);
      const int r = e - m * (T * 128);
      const bool ok = (NWV % NT == 0 || e < NWV) && mblk0 + m < a.nmblk;
      const size_t off = ok ? ((size_t)(mblk0 + m) * a.nchunks + q) * (T * 128) + r : 0;
      sw[i] = wsrc[off];
      if (!ok) sw[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    // ---- input tile: 16 channels, converted to (hi, lo) fp16 and transposed to channel-innermost ----
    if (a.vec) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int e = tq + i * NT;
        if (NITEM % NT == 0 || e < NITEM) {
          const int g2 = e / (LH * LW4);
          const int rem = e - g2 * (LH * LW4);
          const int r = rem / LW4;
          const int c4 = rem - r * LW4;
          float4 v[8];
          if constexpr (!GEN) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = st[i][k].v;
            if (okm[i] != 0xFFu) {          // border / channel tail: zero what lies outside
#pragma unroll
              for (int k = 0; k < 8; ++k)
                if (!((okm[i] >> k) & 1u)) v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = item_value<GEN>(st[i][k], (okm[i] >> k) & 1u, (psm[i] >> k) & 1u);
          }
          const int rec0 = g2 * REC + r * LWP + 4 * c4 + (c4 >> 1);
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            float x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = p == 0 ? v[k].x : (p == 1 ? v[k].y : (p == 2 ? v[k].z : v[k].w));
            uint4 hi, lo;
            split8(x, hi, lo);
            s_bh[rec0 + p] = hi;
            s_bl[rec0 + p] = lo;
          }
        }
      }
    } else {   // any W: element-wise (slow path, odd test shapes only)
      for (int e = tq; e < 2 * LH * LW; e += NT) {
        const int g2 = e / (LH * LW);
        const int rem = e - g2 * (LH * LW);
        const int r = rem / LW;
        const int c = rem - r * LW;
        const int gy = y0 + r - PAD, gx = x0 + c - XO;
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int cg = q * CK16 + g2 * 8 + k;
          x[k] = (cg < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? load_input(a, im, n, cg, gy, gx) : 0.0f;
        }
        uint4 hi, lo;
        split8(x, hi, lo);
        s_bh[g2 * REC + r * LWP + c + (c >> 3)] = hi;
        s_bl[g2 * REC + r * LWP + c + (c >> 3)] = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tq + i * NT;
      if (NWV % NT == 0 || e < NWV) {
        const int m = e / (T * 128);
        const int r = e - m * (T * 128);
        const int tap = r >> 7, hl = (r >> 6) & 1, ln = r & 63;
        (hl ? s_al : s_ah)[(m * T + tap) * 64 + ln] = sw[i];
      }
    }
    __syncthreads();
    if (a.vec && q + 1 < a.nchunks) fetch_tile(q + 1, tq);     // in flight during the MFMA phase below
    // ---- one K=16 step per tap: 3 MFMAs per (channel block, pixel block) ----
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
      const int dy = (tap / KS) * DIL;
      half8 ah[MB], al[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        ah[i] = __builtin_bit_cast(half8, s_ah[abase + (i * T + tap) * 64]);
        al[i] = __builtin_bit_cast(half8, s_al[abase + (i * T + tap) * 64]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int rr = b >> 1, cb = b & 1;
        const int idx = bcol[tap % KS] + (rr + dy) * LWP + cb * 36;
        const half8 bh = __builtin_bit_cast(half8, s_bh[idx]);
        const half8 bl = __builtin_bit_cast(half8, s_bl[idx]);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, acc[i][b], 0, 0, 0);
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, acc[i][b], 0, 0, 0);
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, acc[i][b], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (identical to the fp32 kernel: the D fragment layout does not depend on the input dtype) ----
  // activation as two leaky slopes (1.0 = identity): before BatchNorm/residual, or (has_act & 4) after them
  const int act_kind = a.has_act & 3;
  const bool act_late = (a.has_act & 4) != 0, act_tanh = act_kind == 2, act_silu = act_kind == 3;
  const float slope_pre = (act_kind == 1 && !act_late) ? a.slope : 1.0f;
  const float slope_post = (act_kind == 1 && act_late) ? a.slope : 1.0f;
  const size_t plane = (size_t)a.H * a.W;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int ml = wm * MB + i;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int gy = y0 + wn * RPW + (b >> 1), gx = x0 + (b & 1) * 32 + jj;
      const bool pix_ok = gy < a.H && gx < a.W;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = ml * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const int co = mblk0 * 32 + cl;
        const bool ok = pix_ok && co < a.Cout;
        const size_t o = ok ? ((size_t)n * a.Cout + co) * plane + (size_t)gy * a.W + gx : 0;
        float v = acc[i][b][r] + s_epi[cl];
        v = v > 0.0f ? v : v * slope_pre;
        if (act_tanh) v = tanhf(v);
        if (act_silu) v = v / (1.0f + expf(-v));      // nn.SiLU (EfficientNetV2 blocks)
        v = v * s_epi[MBLK * 32 + cl] + s_epi[2 * MBLK * 32 + cl];
        if (resid) v += resid[o];
        v = v > 0.0f ? v : v * slope_post;
        if (ok) out[o] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// 1x1 convolution, streaming form.  No halo => no LDS input tile: a wave owns NBW blocks of 32 consecutive pixels
// and ALL output channels (MB blocks of 32), reads its B operand straight from global memory (lane (r, h): 8 channels
// of pixel r, eight 4-byte loads that are each two 128-byte segments per wave), converts to (hi, lo) in registers
// and runs 3 MFMAs per channel block.  The input is read from HBM exactly once, the output written once; only
// the weight fragments (shared by the 4 waves) go through LDS, 64 input channels per barrier pair.
// Requirements (else the tiled kernel is used): plain sources whose channel counts are multiples of 16 (the last
// one may be ragged), H*W % 32 == 0.
// ---------------------------------------------------------------------------------------------------------
template <int MB, int NBW>
__global__ __launch_bounds__(256, (MB * NBW >= 8) ? 2 : ((MB * NBW >= 4) ? 3 : 4)) void conv1x1_f16x3_kernel(const ConvArgs a, const float* __restrict__ resid,
                                                                                      float* __restrict__ out) {
  constexpr int KSPC = 4;                               // K-steps (16 channels each) per weight chunk
  constexpr int NWV = MB * KSPC * 128, NW = (NWV + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* s_ah = reinterpret_cast<uint4*>(smem);         // [MB][KSPC][64]
  uint4* s_al = s_ah + MB * KSPC * 64;
  float* s_epi = reinterpret_cast<float*>(s_al + MB * KSPC * 64);

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

  // per pixel block: image index and offset inside the H*W plane of this lane's pixel
  long long img[NBW];
  long long hw[NBW];
  bool live[NBW];
#pragma unroll
  for (int b = 0; b < NBW; ++b) {
    const long long pb = pb0 + b;
    live[b] = pb < nblocks;
    const long long pix = (live[b] ? pb : 0) * 32 + jj;
    img[b] = pix / HW;
    hw[b] = pix - img[b] * HW;
  }

  // image of every source tensor read by each pixel block (batch-broadcast sources): resolved once, not per K-step
  int simg[NBW][SLU_MAX_SRC];
#pragma unroll
  for (int b = 0; b < NBW; ++b)
#pragma unroll
    for (int s = 0; s < SLU_MAX_SRC; ++s) simg[b][s] = (s < a.nsrc && a.src[s].nb) ? (int)img[b] % a.src[s].nb : (int)img[b];

  const uint4* wsrc = reinterpret_cast<const uint4*>(a.wpack);
  const int nq = (a.nchunks + KSPC - 1) / KSPC;         // a.nchunks counts 16-channel groups
  int src = 0;                                          // source that holds channel group `g16` (sources are 16-aligned)
  for (int q = 0; q < nq; ++q) {
    // ---- B operands of the whole chunk: issued before the barriers so HBM latency overlaps the weight staging ----
    float x[KSPC][NBW][8];
    {
      int s_ = src;
#pragma unroll
      for (int ks = 0; ks < KSPC; ++ks) {
        const int c0 = (q * KSPC + ks) * CK16;          // first channel of this K-step (wave-uniform)
        while (s_ + 1 < a.nsrc && c0 >= a.src[s_ + 1].cbeg) ++s_;
        const SrcDev& S = a.src[s_];
        const int cl0 = c0 - S.cbeg + 8 * hh;           // this lane half's first channel inside the source
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          const int is_ = s_ == 0 ? simg[b][0] : (s_ == 1 ? simg[b][1] : simg[b][2]);
          const size_t base = ((size_t)is_ * S.C + cl0) * HW + hw[b];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            // channels past the end of the source (ragged tail / K padding): read element 0 instead, use 0
            const bool ok = live[b] && c0 < a.Cin && cl0 + k < S.ccount;
            const float v = S.ptr[ok ? base + (size_t)k * HW : 0];
            x[ks][b][k] = ok ? v : 0.0f;
          }
        }
      }
      src = s_;
    }
    __syncthreads();
    uint4 sw[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 256;
      const int m = e / (KSPC * 128);
      const int r = e - m * (KSPC * 128);
      const int ks = r >> 7;
      const bool ok = (NWV % 256 == 0 || e < NWV) && m < a.nmblk && q * KSPC + ks < a.nchunks;
      const size_t off = ok ? ((size_t)m * a.nchunks + q * KSPC + ks) * 128 + (r & 127) : 0;
      sw[i] = wsrc[off];
      if (!ok) sw[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int e = tid + i * 256;
      if (NWV % 256 == 0 || e < NWV) {
        const int m = e / (KSPC * 128);
        const int r = e - m * (KSPC * 128);
        const int ks = r >> 7, hl = (r >> 6) & 1, ln = r & 63;
        (hl ? s_al : s_ah)[(m * KSPC + ks) * 64 + ln] = sw[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KSPC; ++ks) {
      half8 bh[NBW], bl[NBW];
#pragma unroll
      for (int b = 0; b < NBW; ++b) {
        uint4 hi, lo;
        split8(x[ks][b], hi, lo);
        bh[b] = __builtin_bit_cast(half8, hi);
        bl[b] = __builtin_bit_cast(half8, lo);
      }
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const half8 ah = __builtin_bit_cast(half8, s_ah[(i * KSPC + ks) * 64 + lane]);
        const half8 al = __builtin_bit_cast(half8, s_al[(i * KSPC + ks) * 64 + lane]);
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[b], acc[i][b], 0, 0, 0);
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[b], acc[i][b], 0, 0, 0);
          acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[b], acc[i][b], 0, 0, 0);
        }
      }
    }
  }

  // activation as two leaky slopes (1.0 = identity): before BatchNorm/residual, or (has_act & 4) after them
  const int act_kind = a.has_act & 3;
  const bool act_late = (a.has_act & 4) != 0, act_tanh = act_kind == 2, act_silu = act_kind == 3;
  const float slope_pre = (act_kind == 1 && !act_late) ? a.slope : 1.0f;
  const float slope_post = (act_kind == 1 && act_late) ? a.slope : 1.0f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const bool ok = live[b] && co < a.Cout;
        const size_t o = ok ? ((size_t)img[b] * a.Cout + co) * HW + hw[b] : 0;
        float v = acc[i][b][r] + s_epi[co];
        v = v > 0.0f ? v : v * slope_pre;
        if (act_tanh) v = tanhf(v);
        if (act_silu) v = v / (1.0f + expf(-v));      // nn.SiLU (EfficientNetV2 blocks)
        v = v * s_epi[MB * 32 + co] + s_epi[2 * MB * 32 + co];
        if (resid) v += resid[o];
        v = v > 0.0f ? v : v * slope_post;
        if (ok) out[o] = v;
      }
}

template <int MB, int NBW>
int launch_1x1(ConvArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * MB * 4 * 64 * 16 + (size_t)3 * MB * 32 * 4;
  const long long nblocks = (long long)a.N * a.H * a.W / 32;
  const long long gx = (nblocks + 4 * NBW - 1) / (4 * NBW);
  if (gx <= 0 || gx > 0x7fffffffLL) return SLU_EUNSUPPORTED;
  auto kern = conv1x1_f16x3_kernel<MB, NBW>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), lds, st, a, a.resid, a.out);
  SLU_CHECK_LAUNCH();
}

// the streaming kernel applies when every source is plain and 16-channel aligned and pixel blocks do not straddle images
bool stream_1x1_ok(const slu_conv_desc* d, const ConvArgs& a) {
  if (d->ksize != 1 || a.gen || a.nmblk > 8) return false;
  if (((long long)a.H * a.W) % 32) return false;
  for (int s = 0; s + 1 < a.nsrc; ++s)
    if (a.src[s].ccount % 16) return false;
  return true;
}

int launch_1x1_any(ConvArgs& a, hipStream_t st) {
  if (a.nmblk == 1) return launch_1x1<1, 1>(a, st);
  if (a.nmblk == 2) return launch_1x1<2, 1>(a, st);
  if (a.nmblk <= 4) return launch_1x1<4, 1>(a, st);
  return launch_1x1<8, 1>(a, st);
}

// wpack16[mblk][chunk][tap][hi|lo][lane][8]: lane (r, h) holds W[co = 32 mblk + r][ci = 16 chunk + 8 h + j][tap], j = 0..7
__global__ void pack_f16x3_kernel(const float* __restrict__ w, int cout, int cin, int ks, int nchunks, size_t total_frag,
                                  uint4* __restrict__ out) {
  const int T = ks * ks;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_frag; e += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    size_t r = e >> 6;
    const int tap = (int)(r % T);
    r /= T;
    const int q = (int)(r % nchunks);
    const int m = (int)(r / nchunks);
    const int co = m * 32 + (lane & 31);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = q * CK16 + 8 * (lane >> 5) + j;
      x[j] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * T + tap] : 0.0f;
    }
    uint4 hi, lo;
    split8(x, hi, lo);
    const size_t base = (((size_t)m * nchunks + q) * T + tap) * 128;
    out[base + lane] = hi;
    out[base + 64 + lane] = lo;
  }
}

template <int KS, int DIL, int PAD, int MB, int WM, int WN, int RPW, bool GEN>
int launch_cfg16(ConvArgs& a, hipStream_t st) {
  constexpr int TH = WN * RPW, MBLK = WM * MB, T = KS * KS, XO = PAD ? 4 : 0;
  constexpr int REC = (TH + 2 * PAD) * ((64 + 2 * XO) + (64 + 2 * XO) / 8);
  constexpr size_t lds = (size_t)4 * REC * 16 + (size_t)2 * MBLK * T * 64 * 16 + (size_t)3 * MBLK * 32 * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  a.tiles_x = (a.W + 63) / 64;
  a.tiles_y = (a.H + TH - 1) / TH;
  const long long gx = (long long)a.tiles_x * a.tiles_y * a.N;
  const int gy = (a.nmblk + MBLK - 1) / MBLK;
  if (gx <= 0 || gx > 0x7fffffffLL || gy > 65535) return SLU_EUNSUPPORTED;
  auto kern = conv_f16x3_kernel<KS, DIL, PAD, MB, WM, WN, RPW, GEN>;
  static SluLdsGrant grant;
  if (slu_grant_dynamic_lds(reinterpret_cast<const void*>(kern), lds, grant) != SLU_OK) return SLU_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)gy), dim3(64 * WM * WN), lds, st, a, a.resid, a.out);
  SLU_CHECK_LAUNCH();
}

template <int KS, int DIL, int PAD, bool GEN>
int launch_tiles16(ConvArgs& a, int cfg, hipStream_t st) {
  switch (cfg) {
    case M32_TH8:  return launch_cfg16<KS, DIL, PAD, 1, 1, 4, 2, GEN>(a, st);
    case M64_TH8:  return launch_cfg16<KS, DIL, PAD, 2, 1, 4, 2, GEN>(a, st);
    case M128_TH4: return launch_cfg16<KS, DIL, PAD, 2, 2, 2, 2, GEN>(a, st);
    case M32_TH4:  return launch_cfg16<KS, DIL, PAD, 1, 1, 4, 1, GEN>(a, st);
    case M64_TH4:  return launch_cfg16<KS, DIL, PAD, 2, 1, 4, 1, GEN>(a, st);
  }
  return SLU_EUNSUPPORTED;
}

template <int KS, int DIL, int PAD>
int launch_family16(ConvArgs& a, int cfg, hipStream_t st) {
  return a.gen ? launch_tiles16<KS, DIL, PAD, true>(a, cfg, st) : launch_tiles16<KS, DIL, PAD, false>(a, cfg, st);
}

}  // namespace

extern "C" size_t slu_packed_weight_bytes_f16x3(int cout, int cin, int ksize) {
  if (cout <= 0 || cin <= 0 || ksize <= 0) return 0;
  const size_t nmblk = (cout + 31) / 32, nchunks = (cin + CK16 - 1) / CK16;
  return nmblk * nchunks * (size_t)(ksize * ksize) * 128 * 16;
}

extern "C" int slu_pack_conv_weight_f16x3(const float* w, int cout, int cin, int ksize, void* out, slu_stream_t stream) {
  if (!w || !out) return SLU_EINVAL;
  const size_t bytes = slu_packed_weight_bytes_f16x3(cout, cin, ksize);
  if (bytes == 0) return SLU_EINVAL;
  const size_t frags = bytes / 32;      // one (hi, lo) pair of uint4 per lane
  const int nchunks = (cin + CK16 - 1) / CK16;
  const unsigned blocks = (unsigned)((frags + 255) / 256 > 4096 ? 4096 : (frags + 255) / 256);
  hipLaunchKernelGGL(pack_f16x3_kernel, dim3(blocks), dim3(256), 0, slu_stream(stream), w, cout, cin, ksize, nchunks, frags,
                     reinterpret_cast<uint4*>(out));
  SLU_CHECK_LAUNCH();
}

// called by slu_conv2d_fwd (conv2d.hip) when desc->precision == SLU_CONV_F16X3
int slu_conv2d_fwd_f16x3_impl(const slu_conv_desc* d, hipStream_t st) {
  ConvArgs a{};
  const int rc = fill_args(d, a);
  if (rc != SLU_OK) return rc;
  if (stream_1x1_ok(d, a)) return launch_1x1_any(a, st);
  int cfg = choose_cfg(a);
  // 4-row tiles only: the (hi, lo) input tile of an 8-row tile leaves room for a single workgroup per CU
  if (cfg == M32_TH8) cfg = M32_TH4;
  if (cfg == M64_TH8) cfg = M64_TH4;
  if (d->ksize == 1 && d->dil == 1 && d->pad == 0) return launch_family16<1, 1, 0>(a, cfg, st);
  if (d->ksize == 3 && d->dil == 1 && d->pad == 1) return launch_family16<3, 1, 1>(a, cfg, st);
  if (d->ksize == 3 && d->dil == 2 && d->pad == 2) return launch_family16<3, 2, 2>(a, cfg, st);
  if (d->ksize == 2 && d->dil == 2 && d->pad == 1) return launch_family16<2, 2, 1>(a, cfg, st);
  if (d->ksize == 2 && d->dil == 1 && d->pad == 1) return launch_family16<2, 1, 1>(a, cfg, st);
  return SLU_EUNSUPPORTED;
}
