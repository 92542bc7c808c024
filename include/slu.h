/*
 * slu.h -- C ABI of libslu_hip.so: the MI355X (gfx950) kernels behind the SemanticLiDARUnc
 * range-image hot path.
 *
 * The reference (kav-institute/SemanticLiDARUnc) is pure Python on torch: it has no FFI layer, so
 * there is no reference-side binding to replace one-for-one.  Each entry point below names the
 * reference Python code whose arithmetic it replaces (path:line under the reference's src/), and
 * INTEGRATION.md shows the ctypes stub a maintainer adds to call it.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says HOST; tensors are dense NCHW fp32
 *    (W = azimuth fastest), labels / predictions int64, exactly the layouts the reference uses;
 *  - the library never allocates, frees or synchronises: outputs and scratch are caller-owned,
 *    every launch is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default);
 *  - return value 0 = launched, negative = SLU_E* (nothing was launched); never throws/exits;
 *  - re-entrant, no mutable global state.
 */
#ifndef SLU_H_
#define SLU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLU_ABI_VERSION 31

#define SLU_OK            0
#define SLU_EINVAL       -1   /* null pointer / non-positive size / inconsistent descriptor   */
#define SLU_EUNSUPPORTED -2   /* shape or kernel family outside what the library instantiates */
#define SLU_ELAUNCH      -3   /* hipGetLastError() != hipSuccess right after the launch       */

typedef void* slu_stream_t;

int slu_abi_version(void);
const char* slu_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Convolution (replaces nn.Conv2d + nn.LeakyReLU + eval nn.BatchNorm2d + residual add +
 * torch.cat + nn.PixelShuffle(2) + nn.Dropout2d of  baselines/SalsaNext/SalsaNext.py:25-39,
 * :73-109, :142-170, :213)
 *
 *   out[n,co,y,x] = resid[n,co,y,x]
 *                 + bn_a[co] * leaky( bias[co] + sum_{ci,i,j} w[co,ci,i,j] * in[n,ci,y-pad+i*dil,x-pad+j*dil] ) + bn_b[co]
 *
 * `in` is the channel concatenation of up to 3 sources; a source may be read through
 * PixelShuffle(2) and multiplied by a per-(n, source-channel) factor (folded Dropout2d).
 * `out` must not overlap any input (sources, resid).  Stride 1, "same" output size; kernel families (ksize,dil,pad) = (1,1,0) (3,1,1) (3,2,2) (2,2,1).
 * ------------------------------------------------------------------------------------------ */
#define SLU_MAX_SRC 3

typedef struct slu_conv_src {
  const float* ptr;    /* [N, C, H, W]  or, when pixel_shuffle, [N, C, H/2, W/2]                */
  const float* scale;  /* [N, C] multiplier per (sample, source channel), or NULL               */
  int32_t C;           /* channels of the stored tensor                                          */
  int32_t pixel_shuffle; /* 1: contributes C/4 channels, in[c,y,x] = ptr[4c+2(y&1)+(x&1), y/2, x/2] */
  int32_t nbatch;      /* 0: the tensor holds N images; k > 0: it holds k images and output image n reads image n % k
                          (a deterministic skip tensor shared by the stacked MC passes); `scale` stays [N, C]      */
  int32_t cuse;        /* 0: every channel contributes; k > 0: only the first k channels do (the reference overwrites the
                          last meta_channel_dim channels of a stage output: x[:, :-m], semanticFCN.py:309-313).  Only
                          allowed on the LAST source, without pixel_shuffle                                          */
} slu_conv_src;

typedef struct slu_conv_desc {   /* HOST struct */
  slu_conv_src src[SLU_MAX_SRC];
  int32_t nsrc;
  int32_t N, H, W;       /* output (= input) spatial size                                        */
  int32_t Cin, Cout;     /* Cin = sum of contributed channels                                    */
  int32_t ksize, dil, pad;
  int32_t ck;            /* K-chunk the weights were packed with (slu_conv_ck)                   */
  const float* wpack;    /* slu_pack_conv_weight output                                          */
  const float* bias;     /* [Cout] or NULL                                                       */
  int32_t has_act;       /* 0 none; 1: leaky(v) = v > 0 ? v : slope * v (slope 0 = ReLU); 2: tanh(v); 3: SiLU v * sigmoid(v);
                            +4 (with 1 only): apply it after bn_a/bn_b and the residual add (ResNet BasicBlock) instead of before */
  float slope;
  const float* bn_a;     /* [Cout] or NULL (then bn_b ignored): folded eval BatchNorm            */
  const float* bn_b;
  const float* resid;    /* [N, Cout, H, W] or NULL                                              */
  float* out;            /* [N, Cout, H, W]                                                      */
  int32_t precision;     /* SLU_CONV_FP32 (exact fp32 MFMA) or SLU_CONV_F16X3 (split-fp16, see below)             */
  double* stats;         /* NULL, or f64 [2][Cout] (zero it first): the kernel adds the per-channel sum and sum of squares of the values it
                            stores -- the batch statistics of a train-mode BatchNorm that follows (SalsaNext.py:30-36) without a second pass
                            over the output.  SLU_CONV_FP32 only.                                                                        */
} slu_conv_desc;

/* precision of the multiply-accumulate inside slu_conv2d_fwd (inputs, outputs and accumulation are fp32 either way):
 *   SLU_CONV_FP32  : v_mfma_f32_32x32x2_f32, bit-for-bit a k-ordered fmaf chain; wpack from slu_pack_conv_weight, ck = slu_conv_ck
 *   SLU_CONV_F16X3 : every fp32 operand x is split x = hi + lo (two fp16), products hi*hi + hi*lo + lo*hi on
 *                    v_mfma_f32_32x32x16_f16 (fp32 accumulate): ~2^-22 relative error per product, 5.3x fewer MFMA
 *                    cycles; |x| must stay below 65504; wpack from slu_pack_conv_weight_f16x3, ck = 16 */
#define SLU_CONV_FP32  0
#define SLU_CONV_F16X3 1

/* K-chunk (input channels staged per LDS round) used for a kernel family. */
int slu_conv_ck(int ksize);
/* number of floats of the packed (MFMA A-fragment ordered, zero padded) weight image */
size_t slu_packed_weight_floats(int cout, int cin, int ksize, int ck);
/* w: [cout, cin, ksize, ksize] (torch OIHW) -> out: packed image.  Re-run after every weight update. */
int slu_pack_conv_weight(const float* w, int cout, int cin, int ksize, int ck, float* out, slu_stream_t stream);
/* many weights in one launch (a training step repacks all of them after the optimizer step).  jobs_dev: DEVICE array; job i packs w
 * [cout][cin][k][k] into `out` (slu_packed_weight_floats(cout, cin, k, ck) floats) as slu_pack_conv_weight does, or -- dgrad = 1 -- the weights
 * of the data-gradient conv (slu_dgrad_weight + slu_pack_conv_weight: slu_packed_weight_floats(cin, cout, k, ck) floats); begin = sum of the
 * output sizes of the jobs before it (ascending), total = sum over all jobs. */
typedef struct slu_pack_job {
  const float* w;
  float* out;
  int32_t cout, cin, ksize, ck;
  int32_t dgrad, reserved;
  uint64_t begin;
} slu_pack_job;
int slu_pack_conv_weights_multi(const slu_pack_job* jobs_dev, int njobs, size_t total, slu_stream_t stream);
int slu_conv2d_fwd(const slu_conv_desc* desc, slu_stream_t stream);
/* split-fp16 weight image (bytes) and its packer (w: [cout,cin,k,k] fp32 OIHW) */
size_t slu_packed_weight_bytes_f16x3(int cout, int cin, int ksize);
int slu_pack_conv_weight_f16x3(const float* w, int cout, int cin, int ksize, void* out, slu_stream_t stream);
/* name of the kernel instantiation slu_conv2d_fwd launches for `desc` (as rocprofv3 prints it); HOST buf >= 64 bytes */
int slu_conv2d_kernel_name(const slu_conv_desc* desc, char* buf, size_t buflen);

/* a = gamma / sqrt(var + eps), b = beta - mean * a   (eval-mode nn.BatchNorm2d, SalsaNext.py:32,36,...) */
int slu_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                int C, float* a, float* b, slu_stream_t stream);

/* nn.AvgPool2d(3, stride=2, padding=1) (count_include_pad) of x * scale[n,c]  (SalsaNext.py:69,98-101).
 * x [N,C,H,W] -> y [N,C,(H+1)/2,(W+1)/2]; scale [N,C] or NULL. */
int slu_avgpool3s2_fwd(const float* x, const float* scale, float* y, int N, int C, int H, int W,
                       slu_stream_t stream);
/* same, but x holds only `in_batch` images and output image n pools x[n % in_batch] (scale is [N,C]) */
int slu_avgpool3s2_bcast_fwd(const float* x, const float* scale, float* y, int N, int in_batch, int C, int H, int W,
                             slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * MC-dropout reduction (replaces models/trainer.py:1105-1136,1143-1154 = tester.py:412-451)
 *   probs_t = exp(log_softmax(logits_t)); p_bar = mean_t probs_t;
 *   H_norm = -sum_c clamp(p_bar,eps) ln clamp(p_bar,eps) / ln C
 *   MI_norm = max(0, (H - mean_t H[probs_t]) / ln C);  preds = argmax_c p_bar (first max wins)
 * logits [T,B,C,HW]; p_bar [B,C,HW]; h_norm, mi_norm [B,HW]; preds int64 [B,HW].  C <= 32.
 * ------------------------------------------------------------------------------------------ */
int slu_mc_reduce(const float* logits, int T, int B, int C, int HW, float eps,
                  float* p_bar, float* h_norm, float* mi_norm, int64_t* preds, slu_stream_t stream);

/* Single-pass eval (replaces models/trainer.py:1180-1184,1211-1214):
 *   probs = exp(log_softmax(logits)); preds = argmax; H_norm = -sum p ln max(p, eps) / ln C */
int slu_softmax_entropy(const float* logits, int B, int C, int HW, float eps,
                        float* probs, float* h_norm, int64_t* preds, slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Metrics
 * ------------------------------------------------------------------------------------------ */
/* confmat[t*C+p] += 1 for every pixel with 0 <= t,p < C   (models/evaluator.py:45-53; rows = GT) */
int slu_confusion_update(const int64_t* preds, const int64_t* targets, int64_t n, int C,
                         int64_t* confmat, slu_stream_t stream);

/* Top-label calibration bins (metrics/ece.py:55-84,136-140), mode 'probs':
 *   p = max(probs,0) / max(sum_c, 1e-12); conf = clamp(max_c p, 0, 1); correct = argmax == label;
 *   pixels with label == ignore_index are skipped (pass INT64_MIN for "no ignore");
 *   bin b = [b/n_bins, (b+1)/n_bins) on float32 edges linspace(0,1,n_bins+1), last bin closed.
 * Accumulates: count int64[n_bins], sum_correct f64[n_bins], sum_conf f64[n_bins]. */
int slu_ece_update(const float* probs, const int64_t* labels, int B, int C, int HW, int64_t ignore_index,
                   int n_bins, int64_t* count, double* sum_correct, double* sum_conf, slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Loss (models/trainer.py:511-514): probs = softmax(logits);
 *   nll_sum += sum_pixels -ln max(probs[label], clamp)     (caller divides by the pixel count)
 * logits [B,C,HW], labels int64 [B,HW] in [0,C); probs out [B,C,HW] (may be NULL); nll_sum f64[1].
 * ------------------------------------------------------------------------------------------ */
int slu_softmax_nll_fwd(const float* logits, const int64_t* labels, int B, int C, int HW, float clamp,
                        float* probs, double* nll_sum, slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Per-pixel NLL / cross-entropy (models/losses.py:55-73, trainer.py:514).  x [B,C,HW]; pixels whose label is
 * == ignore_index or outside [0,C) are skipped.  kind: 0 logits (nll = logsumexp - x_y, i.e. CrossEntropyLoss),
 * 1 probs, -ln max(p_y, param);  2 probs, -ln(p_y + param);  3 log-probs, -x_y.
 * fwd: nll_sum f64[1] += sum, count int64[1] += #pixels (caller zeroes both, divides).
 * bwd: grad_x[B,C,HW] = gscale[0] * d(sum nll)/dx   (gscale: DEVICE scalar = upstream grad / count).
 * ------------------------------------------------------------------------------------------ */
int slu_nll_fwd(const float* x, const int64_t* labels, int B, int C, int HW, int kind, float param, int64_t ignore_index,
                double* nll_sum, int64_t* count, slu_stream_t stream);
int slu_nll_bwd(const float* x, const int64_t* labels, int B, int C, int HW, int kind, float param, int64_t ignore_index,
                const float* gscale, float* grad_x, slu_stream_t stream);

/* Softmax backward of the fused SalsaNext loss (trainer.py:511-516):
 *   g_c = gout[0] * ( w_dense * dense[c]  -  [c == y and p_y >= clamp] * w_nll / p_y ),  grad_logits_c = p_c (g_c - sum_k g_k p_k)
 * probs, dense (nullable), grad_logits [B,C,HW]; labels nullable; gout DEVICE scalar or NULL (= 1). */
int slu_softmax_loss_bwd(const float* probs, const int64_t* labels, const float* dense, float w_dense, float w_nll, float clamp,
                         const float* gout, int B, int C, int HW, float* grad_logits, slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Lovasz-Softmax (losses/lovasz.py:12-88).  probs [B,C,HW] (rows sum to 1), labels int64 [B,HW];
 * pixels with label == ignore_index are dropped (pass INT64_MIN for none).
 * class_mask: 0 = classes='present' (the classes that occur among the valid labels); else bit c set = class c is summed whether it occurs
 *   or not (classes='all': all C bits; classes=[...]: those) -- an absent class contributes its largest probability (lovasz.py:62-75).
 *   loss[1]      : mean over the summed classes of  sum_k err_(k) * (J_k - J_{k-1})   (errors sorted descending)
 *   n_present[1] : number of summed classes (float)
 *   grad_probs   : [B,C,HW] d loss / d probs, or NULL
 * workspace: slu_lovasz_workspace_bytes(B,C,HW) bytes, 256-byte aligned, caller-owned scratch.
 * ------------------------------------------------------------------------------------------ */
size_t slu_lovasz_workspace_bytes(int B, int C, int HW);
int slu_lovasz_fwd(const float* probs, const int64_t* labels, int B, int C, int HW, int64_t ignore_index, unsigned class_mask,
                   void* workspace, size_t workspace_bytes, float* loss, float* n_present, float* grad_probs,
                   slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Training side of the conv stack (autograd of SalsaNext.py:25-39,73-109,142-170; nn.BatchNorm2d train mode).
 * All tensors dense fp32; [N,C,HW] = NCHW with H*W flattened.  f64 accumulators are caller-zeroed, += semantics.
 * ------------------------------------------------------------------------------------------ */
/* sum[c] += sum y, sumsq[c] += sum y^2 over (N,HW) */
int slu_bn_stats(const float* y, int N, int C, int HW, double* sum, double* sumsq, slu_stream_t stream);
/* s1[c] += sum dz, s2[c] += sum dz * (y - mean[c]) * invstd[c] */
int slu_bn_bwd_reduce(const float* dz, const float* y, const float* mean, const float* invstd, int N, int C, int HW,
                      double* s1, double* s2, slu_stream_t stream);
/* per-channel BatchNorm coefficients in one launch (fp64 inside).  train != 0: statistics from sum/sumsq over `count` elements
 * (biased variance), running_mean/var updated in place with `momentum` and the unbiased variance (nullable: no update);
 * train == 0: statistics = running_mean/var.  Outputs mean, invstd, a = gamma*invstd, b = beta - mean*a (all [C]). */
int slu_bn_coeffs_fwd(const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float eps,
                      float momentum, int train, float* running_mean, float* running_var, int C, float* mean, float* invstd,
                      float* a, float* b, slu_stream_t stream);
/* backward coefficients for slu_act_affine_bwd from slu_bn_bwd_reduce sums: k1 = gamma*invstd; train: k3 = -gamma*invstd^2*s2/count,
 * k2 = -gamma*invstd*s1/count - k3*mean; eval: k2 = k3 = 0; dgamma = s2, dbeta = s1 */
int slu_bn_coeffs_bwd(const double* s1, const double* s2, double count, const float* gamma, const float* mean, const float* invstd,
                      int train, int C, float* k1, float* k2, float* k3, float* dgamma, float* dbeta, slu_stream_t stream);
/* z = a[c]*y + b[c] + resid   (a, b, resid nullable: 1, 0, 0) */
int slu_affine_fwd(const float* y, const float* a, const float* b, const float* resid, float* z, int N, int C, int HW,
                   slu_stream_t stream);
/* da = (k1[c]*dz + k2[c] + k3[c]*y) * (has_act && y <= 0 ? slope : 1);  dbias[c] += sum da  (k*, y, dbias nullable) */
int slu_act_affine_bwd(const float* dz, const float* y, const float* k1, const float* k2, const float* k3, float slope,
                       int has_act, int N, int C, int HW, float* da, double* dbias, slu_stream_t stream);
/* The two BatchNorm passes of a layer with their coefficient launches folded in (a training step has 42 such layers):
 * slu_bn_apply_fwd = slu_bn_coeffs_fwd + slu_affine_fwd: z = gamma (y - mean) invstd + beta [+ resid]; mean / invstd [C] are written for the
 *   backward, the running statistics updated as slu_bn_coeffs_fwd does.
 * slu_bn_act_bwd = slu_bn_coeffs_bwd + slu_act_affine_bwd + the fp64 -> fp32 rounding of the bias gradient: has_bn = 0: k1 = 1, k2 = k3 = 0.
 *   acc64 [C] fp64 and ticket [C] uint32 are caller-ZEROED scratch (the last workgroup of a channel to add its partial sum writes dbias[c]);
 *   dbias nullable (then acc64 / ticket are unused); dgamma / dbeta [C] fp32 (has_bn only). */
int slu_bn_apply_fwd(const float* y, const double* sum, const double* sumsq, double count, const float* gamma, const float* beta, float eps,
                     float momentum, int train, float* running_mean, float* running_var, const float* resid, float* z, int N, int C, int HW,
                     float* mean, float* invstd, slu_stream_t stream);
int slu_bn_act_bwd(const float* dz, const float* y, const double* s1, const double* s2, double count, const float* gamma, const float* mean,
                   const float* invstd, int has_bn, int train, float slope, int has_act, int N, int C, int HW, float* da, double* acc64,
                   unsigned* ticket, float* dbias, float* dgamma, float* dbeta, slu_stream_t stream);
/* src [N,C,HW] -> dst [N,HW,Cp], Cp = C rounded up to 32, padding channels zero */
int slu_nchw_to_nhwc(const float* src, int N, int C, int HW, float* dst, slu_stream_t stream);
/* the conv input described by `src` (concat / PixelShuffle / multipliers applied) as [N,H*W,Cp] */
int slu_gather_nhwc(const slu_conv_src* src, int nsrc, int N, int H, int W, float* dst, slu_stream_t stream);
/* gradient of one source: dsrc = (channels [cbeg, cbeg+contributed) of dcat [N,Ccat,H,W], un-shuffled) * scale[n,cs] */
int slu_split_grad(const float* dcat, int N, int Ccat, int cbeg, int H, int W, int Csrc, int pixel_shuffle, const float* scale,
                   float* dsrc, slu_stream_t stream);
/* backward of slu_avgpool3s2_fwd: dy [N,C,(H+1)/2,(W+1)/2] -> dx [N,C,H,W] */
int slu_avgpool3s2_bwd(const float* dy, const float* scale, float* dx, int N, int C, int H, int W, slu_stream_t stream);
/* weights of the data-gradient conv: wd[ci][co][k-1-i][k-1-j] = w[co][ci][i][j]; feed to slu_pack_conv_weight and run
 * slu_conv2d_fwd on da with Cin/Cout swapped (the four kernel families are symmetric: 2*pad == (k-1)*dil) */
int slu_dgrad_weight(const float* w, int cout, int cin, int ksize, float* wd, slu_stream_t stream);
/* dW [cout,cin,k,k] = sum_pixels da * shifted input; da_t [N,HW,Cop], in_t [N,HW,Cip] channel-last (padded to 32);
 * dWp: scratch of slu_wgrad_packed_floats floats.  W must be even.  fp32 atomics: not bit-reproducible. */
size_t slu_wgrad_packed_floats(int cout, int cin, int ksize);
int slu_conv2d_wgrad(const float* da_t, const float* in_t, int N, int H, int W, int Cout, int Cin, int ksize, int dil, int pad,
                     float* dWp, float* dW, slu_stream_t stream);
/* weight gradient of a 1x1 conv straight from the NCHW tensors (no channel-last copies): da [N,Cout,HW], the conv's sources as in
 * slu_conv2d_fwd (plain tensors only: no PixelShuffle / multiplier / broadcast; every source but the last a multiple of 32 channels; HW a
 * multiple of 32; 16-byte aligned bases) -> dW [Cout, sum of source channels].  SLU_EUNSUPPORTED otherwise (use slu_conv2d_wgrad).
 * prezeroed != 0: the caller hands over an accumulator that is already zero (dW here, dWp below) and the launcher skips its fill -- a
 * training step issues ~50 of these calls, one fill each was 0.2 ms of it. */
int slu_conv1x1_wgrad_nchw(const float* da, const slu_conv_src* src, int nsrc, int N, int HW, int Cout, float* dW, int prezeroed, slu_stream_t stream);
/* the 3x3 (dil 1 / 2, pad = dil) and 2x2 (dil 2, pad 1) families: da [N,Cout,H,W]; sources as in slu_conv2d_fwd including PixelShuffle and
 * multipliers (no broadcast / channel cut), any channel counts; W a multiple of 16 (PixelShuffle sources: H even);
 * dWp: scratch of slu_wgrad_packed_floats floats -> dW [Cout, Cin, k, k].  SLU_EUNSUPPORTED otherwise (use slu_conv2d_wgrad). */
int slu_conv2d_wgrad_nchw(const float* da, const slu_conv_src* src, int nsrc, int N, int H, int W, int Cout, int ksize, int dil, int pad,
                          float* dWp, float* dW, int prezeroed, slu_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ResNet-FPN pieces (models/semanticFCN.py:145-153, 230-245, 266-354): data movement around the conv kernel.
 * ------------------------------------------------------------------------------------------ */
/* nn.MaxPool2d(3, 2, 1): x [N,C,H,W] -> y [N,C,(H+1)/2,(W+1)/2] */
int slu_maxpool3s2_fwd(const float* x, float* y, int N, int C, int H, int W, slu_stream_t stream);
/* F.interpolate(mode='nearest', scale_factor=1/factor): y[..,oy,ox] = x[..,oy*factor,ox*factor] */
int slu_nearest_down(const float* x, float* y, int N, int C, int H, int W, int factor, slu_stream_t stream);
/* y[n,(2p+q)*C+c,oy,ox] = x[n,c,2oy+p,2ox+q]: a stride-2 3x3/p1 conv on x == a (2,1,1) conv on y with re-indexed weights */
int slu_space_to_depth2(const float* x, float* y, int N, int C, int H, int W, slu_stream_t stream);
/* the same of cat(a[:, :ca], b) (a [N,Ca,H,W], b [N,Cb,H,W]) -> y [N, 4*(ca+Cb), H/2, W/2] */
int slu_space_to_depth2_cat(const float* a, int Ca, int ca, const float* b, int Cb, float* y, int N, int H, int W, slu_stream_t stream);
/* nn.PixelShuffle(r) of x [N,Cout*r*r,H,W], optionally followed by ELU(v)+1, written into channels [c_off, c_off+Cout) of
 * y [N,Ctot,H*r,W*r] (c_off = 0, Ctot = Cout: a plain PixelShuffle; otherwise several maps share one concatenated buffer) */
int slu_depth_to_space(const float* x, float* y, int N, int Cout, int H, int W, int r, int elu_plus_one, int c_off, int Ctot,
                       slu_stream_t stream);
/* out = value * softmax(score, dim=-1): score [N,1,H,W], value/out [N,C,H,W]; W <= 4096 (AttentionModule :32-38) */
int slu_row_softmax_mul(const float* score, const float* value, float* out, int N, int C, int H, int W, slu_stream_t stream);

/* ---- pieces of the `semanticFCN_opt` variant (baselines/Reichert/semanticFCN_opt.py) -----------------------------
 * slu_bilinear_upsample: F.interpolate(x, scale_factor = scale, mode = 'bilinear', align_corners = False) (UpsampleBlock :24-27):
 *   x [N,C,H,W] -> y [N,C,H*scale,W*scale].
 * slu_groupnorm_fwd: nn.GroupNorm(groups, C, eps) [+ ReLU] (:20,66-70): statistics per (sample, group) over (C/groups)*HW (biased
 *   variance, accumulated in fp64), y = (x - mean) * rstd * gamma[c] + beta[c]; mean / rstd fp32 [N*groups] are outputs; y may be x.
 * slu_spatial_softmax_gate: SpatialAttention's gate (:80-85): w = softmax(score[n] over H*W); out = x * w + x.  x, out [N,C,HW];
 *   score [N,1,HW]; stats fp32 [2 N] scratch (max, 1 / sum). */
int slu_bilinear_upsample(const float* x, float* y, int N, int C, int H, int W, int scale, slu_stream_t stream);
int slu_groupnorm_fwd(const float* x, const float* gamma, const float* beta, int N, int C, int HW, int groups, float eps, int relu, float* mean,
                      float* rstd, float* y, slu_stream_t stream);
int slu_spatial_softmax_gate(const float* x, const float* score, float* stats, float* out, int N, int C, int HW, slu_stream_t stream);

/* ---- fp16 channel-blocked ("h8") inference path: BASELINE.json configs[2],[4] (half-precision storage, fp32 accumulate) -------
 * Activation layout: x[N][G = ceil(C/8)][H][W][8] fp16, pad channels = 0, base pointers 16-byte aligned.
 * Replaces the same reference arithmetic as slu_conv2d_fwd (SalsaNext.py:25-39,73-109,142-170,197-215) with fp16 operands. */
typedef struct slu_h8_src {
  const void* ptr;     /* h8 tensor [nimg][G][H][W][8]                                                              */
  const float* scale;  /* [N][8 G] fp32 multiplier per (output image, channel) (Dropout2d), or NULL; 16-byte aligned  */
  int32_t G;           /* channel blocks                                                                             */
  int32_t nbatch;      /* 0: nimg = N; k > 0: the tensor holds k images, output image n reads image n % k            */
} slu_h8_src;

typedef struct slu_conv_h8_desc {   /* HOST struct */
  slu_h8_src src[SLU_MAX_SRC];  /* concatenated along channels, block-wise (torch.cat(dim=1) of 8-aligned tensors)   */
  int32_t nsrc;
  int32_t N, H, W, Cout, ksize, dil, pad;
  const void* wpack;     /* slu_pack_conv_weight_h8 output                                                           */
  const float* bias;     /* [Cout] fp32 or NULL                                                                      */
  int32_t has_act;       /* 0 none; 1: leaky(v) = v > 0 ? v : slope * v                                              */
  float slope;
  const float* bn_a;     /* [Cout] fp32 or NULL: folded eval BatchNorm                                               */
  const float* bn_b;
  const void* resid;     /* h8 [N][ceil(Cout/8)][H][W][8] or NULL                                                    */
  void* out;             /* h8 [N][ceil(Cout/8)][H][W][8], or fp32 [N][Cout][H][W] when out_f32_nchw                  */
  int32_t out_f32_nchw;  /* 1: the logits head (SalsaNext.py:213) keeps the reference's fp32 NCHW output; 1x1 convs only */
} slu_conv_h8_desc;

size_t slu_packed_weight_bytes_h8(int cout, int cin, int ksize);
/* w: OIHW fp32 [cout][cin][k][k] (cin = real channels of the concatenated input; only the LAST source may be padded) */
int slu_pack_conv_weight_h8(const float* w, int cout, int cin, int ksize, void* out, slu_stream_t stream);
/* out = [resid +] bn_a * act(conv(cat(src * scale)) + bias) + bn_b, rounded to fp16 once.
 * Families: (k,dil,pad) = (1,1,0), (3,1,1), (3,2,2), (2,2,1) as in slu_conv2d_fwd. */
int slu_conv2d_h8_fwd(const slu_conv_h8_desc* desc, slu_stream_t stream);
/* name of the kernel instantiation the call above launches (as rocprofv3 prints it); host only, no launch */
int slu_conv2d_h8_kernel_name(const slu_conv_h8_desc* desc, char* buf, size_t n);
/* fp32 NCHW <-> h8 (set_model_inputs' tensor on the way in, utils/inputs.py:4-34; scale [N][C] optional) */
int slu_nchw_to_h8(const float* x, const float* scale, void* y, int N, int C, int H, int W, slu_stream_t stream);
int slu_h8_to_nchw(const void* x, float* y, int N, int C, int H, int W, slu_stream_t stream);
/* AvgPool2d(3, 2, 1) of x * scale[n][c] (SalsaNext.py:69,98-101); in_batch > 0: x holds in_batch images shared by all n */
int slu_avgpool3s2_h8(const void* x, const float* scale, void* y, int N, int in_batch, int G, int H, int W, slu_stream_t stream);
/* nn.PixelShuffle(2) (SalsaNext.py:143): y[n][c][2h+i][2w+j] = x[n][4c+2i+j][h][w] * scale_in[n][4c+2i+j] * scale_out[n][c];
 * x: h8 with Gin blocks at HxW; y: h8 with ceil(2 Gin / 8) blocks at 2Hx2W; scales fp32 [N][8 Gin] / [N][2 Gin] or NULL */
int slu_pixel_shuffle_h8(const void* x, const float* scale_in, const float* scale_out, void* y, int N, int Gin, int H, int W,
                         slu_stream_t stream);

/* Fused tail of a SalsaNext block on the h8 path (ResBlock.conv4 + conv5, SalsaNext.py:59-68; UpBlock.conv3 + conv4, :157-167):
 *   a3  = bnA_a * actA(conv2x2_dil2_pad1(a2) + biasA) + bnA_b          (kept on chip, rounded to fp16 like the stored tensor would be)
 *   out = [resid +] bnB_a * actB(conv1x1(cat(a1, a2, a3)) + biasB) + bnB_b
 * a1, a2, resid, out: h8 [N][C/8][H][W][8]; w2x2 = slu_pack_conv_weight_h8 of [C][C][2][2]; w1x1 = the same of [C][3C][1][1].
 * C in {32, 64, 128} (slu_conv_tail_h8_supported); anything else: run the two layers through slu_conv2d_h8_fwd. */
typedef struct slu_conv_tail_h8_desc {   /* HOST struct */
  const void *a1, *a2;
  int32_t N, H, W, C;
  const void *w2x2, *w1x1;
  const float *biasA, *bnA_a, *bnA_b;   /* [C] fp32 or NULL */
  int32_t hasactA;
  float slopeA;
  const float *biasB, *bnB_a, *bnB_b;
  int32_t hasactB;
  float slopeB;
  const void* resid;
  void* out;
  /* shortcut mode (exclusive with resid): the residual is LeakyReLU(conv1x1(sc_x) + sc_bias), rounded to fp16 -- the shortcut branch of a
   * ResBlock (SalsaNext.py:52-53) computed inside the tail from the block's input instead of being written and read back.
   * sc_x: h8 [N][sc_cin/8][H][W][8]; sc_w = slu_pack_conv_weight_h8 of [C][sc_cin][1][1]; slu_conv_tail_h8_shortcut_supported(C, sc_cin). */
  const void *sc_x, *sc_w;
  const float* sc_bias;
  int32_t sc_cin, sc_hasact;
  float sc_slope;
} slu_conv_tail_h8_desc;
int slu_conv_tail_h8_supported(int C, int H, int W);
int slu_conv_tail_h8_shortcut_supported(int C, int sc_cin);
int slu_conv_tail_h8_fwd(const slu_conv_tail_h8_desc* desc, slu_stream_t stream);

/* Fused ResContextBlock on the h8 path (SalsaNext.py:10-39: conv1 1x1 + act -> shortcut; conv2 3x3 + act + bn1; conv3 3x3 dil 2 +
 * act + bn2; + shortcut), the three full-resolution blocks at the head of the network:
 *   s = act(conv1x1(x) + bias1);  a1 = bn1(act(conv3x3_pad1(s) + bias2));  out = s + bn2(act(conv3x3_dil2_pad2(a1) + bias3))
 * s and a1 stay on chip (rounded to fp16 exactly where the three separate slu_conv2d_h8_fwd launches would store them).
 * x: h8 [N][ceil(Cin/8)][H][W][8]; out: h8 [N][4][H][W][8]; w1 / w2 / w3 = slu_pack_conv_weight_h8 of [32][Cin][1][1] / [32][32][3][3] /
 * [32][32][3][3]; act = LeakyReLU(slope), 0 <= slope <= 1.  Cin <= 32, C == 32 (slu_ctx_block_h8_supported). */
typedef struct slu_ctx_block_h8_desc {   /* HOST struct */
  const void* x;
  int32_t N, H, W, Cin, C;
  const void *w1, *w2, *w3;
  const float* bias1;                    /* [32] fp32 or NULL */
  const float *bias2, *bn1_a, *bn1_b;
  const float *bias3, *bn2_a, *bn2_b;
  float slope;
  void* out;
} slu_ctx_block_h8_desc;
int slu_ctx_block_h8_supported(int Cin, int C, int H, int W);
int slu_ctx_block_h8_fwd(const slu_ctx_block_h8_desc* desc, slu_stream_t stream);

/* Segmentation head + MC-dropout reduction in one pass on the h8 path (SalsaNext.py:213 + trainer.py:1143-1154):
 * x: h8 [T*B][G][HW][8], pass-major (image t*B + b = pass t of scan b), 8 G = input channels of the head (G in {2, 4, 8});
 * wpack = slu_pack_conv_weight_h8 of the head's [C][8 G][1][1] weight (C <= 32); bias [C] or NULL.  Outputs as slu_mc_reduce:
 * p_bar fp32 [B][C][HW], h_norm / mi_norm fp32 [B][HW], preds int64 [B][HW].  HW % 32 == 0. */
int slu_head_mc_h8(const void* x, int T, int B, int G, int HW, const void* wpack, const float* bias, int C, float eps, float* p_bar,
                   float* h_norm, float* mi_norm, int64_t* preds, slu_stream_t stream);

/* ---- Dirichlet head (the reference's default loss path, configs/SemanticKitti_default.yaml:10) ---------------
 * alpha = 1 + softplus(scale / T) * softmax(shape) + eps   (probability_helper.py:89-105; trainer.py:533-535 splits the C+1
 *   output channels into shape = outputs[:, :C] and scale = outputs[:, C:C+1], hence the batch strides in elements)
 * alpha0 = sum_c alpha + eps, p_hat = alpha / alpha0        (trainer.py:537-538)
 * entropy   = -sum_c p_hat log(p_hat + eps)                 (get_predictive_entropy :116-121; divide by ln C for _norm :148-153)
 * aleatoric = -sum_c p_hat (digamma(alpha+1) - digamma(alpha0+1))   (get_aleatoric_uncertainty :124-130); epistemic = entropy - aleatoric
 * preds = argmax_c alpha.  Every output pointer may be NULL.  C <= 32. */
int slu_dirichlet_head(const float* shape_logits, long long shape_batch_stride, const float* scale_logits, long long scale_batch_stride,
                       int B, int C, int HW, float temperature, float eps, float* alpha, float* p_hat, float* entropy, float* aleatoric,
                       int64_t* preds, slu_stream_t stream);
/* the same uncertainty measures from a given alpha [B, C, H, W] */
int slu_dirichlet_uncertainty(const float* alpha, int B, int C, int HW, float eps, float* p_hat, float* entropy, float* aleatoric,
                              int64_t* preds, slu_stream_t stream);

/* ---- AUROC of error detection (metrics/auroc.py:36-78) -----------------------------------------------------------
 * slu_auroc_scores: per pixel, probabilities by mode (0 alpha: a / (sum a + eps); 1 logits: softmax; 2 probs: clamp >= 0, renormalise,
 *   auroc.py:36-45), prediction = argmax, uncertainty score by score_kind (0 entropy, 1 entropy_norm, 2 mi, 3 mi_norm, 4 1-maxprob;
 *   mi / mi_norm are the Dirichlet mutual information for mode alpha and fall through to the plain entropy otherwise, :47-63) or
 *   score_override [B,H,W] verbatim; flags[pix] = 0 correct / 1 error / 2 label == ignore_index (has_ignore).  preds [B,C,H,W], C <= 32.
 * slu_auroc_compute: AUROC of `n` samples (score, is_error in {0,1}) exactly as :65-78 (sort by score descending, trapezoid over the
 *   ROC) = sum over negatives of #positives ranked before it / (P N); out3 = {auroc (NaN if P or N is 0), P, N}; optionally the
 *   samples in sorted order (for ROC curves).  The order among equal scores is arbitrary (as numpy's argsort in the reference). */
int slu_auroc_scores(const float* preds, const int64_t* labels, const float* score_override, int B, int C, int HW, int mode, int score_kind,
                     int has_ignore, int64_t ignore_index, float eps, float* scores, uint8_t* flags, slu_stream_t stream);
size_t slu_auroc_workspace_bytes(long long n);
int slu_auroc_compute(const float* scores, const uint8_t* is_error, long long n, void* workspace, size_t workspace_bytes, double* out3,
                      float* sorted_scores, uint8_t* sorted_is_error, slu_stream_t stream);

/* ---- accuracy vs uncertainty bins (models/evaluator.py:640-749 UncertaintyAccuracyAggregator) ---------------------
 * slu_ua_samples: u_out = clamp(uncertainty, 0, 1); flags = 1 label == pred / 0 otherwise / 2 label in ignore_ids (:659-673).
 * slu_binned_counts: count[b] += #samples in bin b, n_correct[b] += #correct ones, bins as np.histogram(u, bins=edges) (:733-739):
 *   [e_b, e_b+1), the last one closed, values outside [e_0, e_K] dropped; edges float32 [n_bins + 1] on the device, n_bins <= 256;
 *   count / n_correct int64 [n_bins], accumulated (zero them first). */
int slu_ua_samples(const int64_t* labels, const int64_t* preds, const float* uncertainty, long long n, const int64_t* ignore_ids, int n_ignore,
                   float* u_out, uint8_t* flags, slu_stream_t stream);
int slu_binned_counts(const float* u, const uint8_t* correct, long long n, const float* edges, int n_bins, int64_t* count, int64_t* n_correct,
                      slu_stream_t stream);

/* ---- ECE sample buffers (metrics/ece.py:55-111,115-140 ECEAggregator with max_samples / binning='adaptive') ----------
 * slu_ece_samples: per pixel conf = clamp(max_c p_c, 0, 1) with p by mode (0 alpha: a / (sum a + eps); 1 logits: softmax; 2 probs:
 *   clamp >= 0, / max(sum, eps); ece.py:55-64) and flags = 1 argmax == label / 0 otherwise / 2 label == ignore_index (has_ignore)
 *   (:75-84).  preds [B,C,HW], C <= 32; conf f32 [B,HW], flags u8 [B,HW].
 * slu_binned_stats: slu_binned_counts plus sum_u[b] += sum of u over bin b (f64): the three np.histogram calls of :136-140. */
int slu_ece_samples(const float* preds, const int64_t* labels, int B, int C, int HW, int mode, int has_ignore, int64_t ignore_index, float eps,
                    float* conf, uint8_t* flags, slu_stream_t stream);
int slu_binned_stats(const float* u, const uint8_t* correct, long long n, const float* edges, int n_bins, int64_t* count, int64_t* n_correct,
                     double* sum_u, slu_stream_t stream);

/* ---- per-class sample lists (models/evaluator.py:211-232 UncertaintyPerClassAggregator.update) --------------------
 * out_values = [values[labels == 0] ..., values[labels == 1] ..., ...] in scan order inside each class (what the reference's boolean
 * masks on host copies return), counts int64 [C] = samples per class; labels outside [0, C) are dropped.  out_values holds n floats;
 * C <= 32, n < 2^32. */
size_t slu_group_by_class_workspace_bytes(long long n);
int slu_group_by_class(const int64_t* labels, const float* values, long long n, int C, float* out_values, int64_t* counts, void* workspace,
                       size_t workspace_bytes, slu_stream_t stream);

/* ---- Tversky loss (models/losses.py:74-128, the 'Tversky' loss branch trainer.py:497-503) -----------------------
 * valid = 0 <= y < C and (no ignore or y != ignore_index); p by model_act (0 logits: softmax, 1 probs, 2 log_probs: exp);
 * per class over valid pixels S = sum p, TP = sum p [y = c], N = #[y = c];  tversky = (TP + s) / (TP + alpha (S - TP) + beta (N - TP) + s);
 * loss = mean (reduction 0) / sum (1) / per class (2) of 1 - tversky; 0 when no pixel is valid.
 * fwd: sums double [3][C] (S | TP | N, zeroed inside), coef float [2][C] (backward coefficients), loss float [1] or [C], any_valid [1].
 * bwd: grad_x [B,C,H,W] = d loss / d x through the activation; grad_out = upstream gradient (DEVICE, [1] or [C], NULL = 1). */
int slu_tversky_fwd(const float* x, const int64_t* labels, int B, int C, int HW, int model_act, int has_ignore, int64_t ignore_index, float alpha,
                    float beta, float smooth, int reduction, double* sums, float* coef, float* loss, float* any_valid, slu_stream_t stream);
int slu_tversky_bwd(const float* x, const int64_t* labels, int B, int C, int HW, int model_act, int has_ignore, int64_t ignore_index, float alpha,
                    float beta, const float* coef, const float* grad_out, int grad_out_per_class, float* grad_x, slu_stream_t stream);

/* ---- per-pixel Dirichlet losses (losses/dirichlet_losses.py:73-221,317-385, losses/regularizers.py:291-389) -----
 * loss = sum over valid pixels (label != ignore_index, 0 <= label < C) of v(alpha[:, pixel], label) / #valid, with
 *   kind 0 NLLDirichletCategorical: -(log(a_y + eps) - log(a0 + eps))          kind 1 DigammaDirichletCE: psi(a0) - psi(a_y)
 *   kind 2 BrierDirichlet: sum_i E[p_i^2] - 2 p_y + 1 (param = s_ref, < 0: alpha0)   kind 3 DirichletMSELoss (Sensoy eq. 5)
 *   kind 4 KL_offClasses_to_uniform (with_conf_weighting = False): KL(Dir(alpha~) || Dir(1)), true class replaced by 1.
 * fwd: sum (double[1]) and count (int64[1]) are zeroed and accumulated; the caller divides (0 valid pixels: loss 0).
 * bwd: grad_alpha [B,C,H,W] = gscale[0] * d v / d alpha at valid pixels, 0 elsewhere; gscale = upstream gradient / count (DEVICE). */
int slu_dirichlet_loss_fwd(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, float param, float eps, int has_ignore,
                           int64_t ignore_index, double* sum, int64_t* count, slu_stream_t stream);
int slu_dirichlet_loss_bwd(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, float param, float eps, int has_ignore,
                           int64_t ignore_index, const float* gscale, float* grad_alpha, slu_stream_t stream);

/* The same pass with a parameter vector (HOST floats) and the two gated regularisers of the Dirichlet path:
 *   kind 5 ComplementKLUniform (losses/dirichlet_losses.py:228-314): w(p_y) * KL(p_off / (1 - p_y) || U); params = {gamma, tau, sigma,
 *          s_target (< 0: no evidence gate), normalize (0/1), detach_uncert (0/1)}; mean over the valid pixels; 0 when C <= 2.
 *   kind 6 WrongLowEvidence (losses/regularizers.py:218-289): gate * relu(ln a0 - ln(C + s_low + eps))^2 with gate = [argmax != y] *
 *          sigmoid((p_max - p_y - margin) / k) (k = 0: hard margin; margin <= 0: no margin gate); params = {s_low, margin, k};
 *          the mean runs over sum(gate): fwd writes sums2 = {sum of values, sum of gates} and the caller divides by max(sums2[1], 1).
 * kinds 0-4 take params = {param} (or none).  bwd: gscale = upstream gradient / the forward's denominator (DEVICE). */
int slu_dirichlet_loss_fwd_ex(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, const float* params, int nparams, float eps,
                              int has_ignore, int64_t ignore_index, double* sums2, int64_t* count, slu_stream_t stream);
int slu_dirichlet_loss_bwd_ex(const float* alpha, const int64_t* labels, int B, int C, int HW, int kind, const float* params, int nparams, float eps,
                              int has_ignore, int64_t ignore_index, const float* gscale, float* grad_alpha, slu_stream_t stream);

/* ---- spherical projection of a point cloud into the range image (dataset/utils.py:61-67,288-349) ------------------
 * pc: float64 [N][C] (x, y, z, then any channels: intensity, label ...; the reference's dataloader concatenates to float64,
 * dataloader_semantic_KITTI.py:49).  Rows / columns by numpy.digitize on the reversed linspace bins (theta range = data min / max
 * when use_data_theta_range, else [theta_min, theta_max]; phi in [-pi, pi]); per pixel the NEAREST point survives (the reference
 * writes in descending range order); img float32 [H][W][C], zeros where no point fell; theta_range_out double[2] or NULL. */
size_t slu_spherical_projection_workspace_bytes(int N, int H, int W);
int slu_spherical_projection(const double* pc, int N, int C, int H, int W, int use_data_theta_range, double theta_min, double theta_max,
                             void* workspace, size_t workspace_bytes, float* img, double* theta_range_out, slu_stream_t stream);
/* The same with the remaining options of the reference function and the dataloaders' flip augmentation:
 *   bins_h: explicit row bins, float64 [H] on the device, strictly monotone (bins_increasing 0 / 1), rows = numpy.digitize(theta, bins_h) - 1
 *           (dataset/utils.py:326-327,334; NULL: the reversed linspace of the theta range);
 *   keep_farthest: sort_largest_first=True of the reference -- points are then written in ASCENDING range order, so the farthest
 *           point of a pixel survives (:301-304);
 *   flip: write the image with reversed columns and negated y (dataloader_semantic_KITTI.py:72-74). */
int slu_spherical_projection_ex(const double* pc, int N, int C, int H, int W, int use_data_theta_range, double theta_min, double theta_max,
                                const double* bins_h, int bins_increasing, int keep_farthest, int flip, void* workspace, size_t workspace_bytes,
                                float* img, double* theta_range_out, slu_stream_t stream);

/* ---- the two ends of the dataloader's __getitem__ around the projection (dataloader_semantic_KITTI.py:31-99) --------
 * slu_kitti_decode: xyzi float32 [N][4] (the .bin file), label uint32 [N] (the .label file; low 16 bits = semantic id), lut int32
 *   [lut_size] = dataset/definitions.py id_map (-1 where the dict has no key) -> pc float64 [N][5] = (x, y, z, intensity, class);
 *   rotate != 0: xyz @ Rz with the given cosine / sine (rotate_z, dataset/utils.py:4-18).  bad_count int32[1] (zero it first) counts
 *   labels without a LUT entry (the reference raises KeyError).
 * slu_range_image_split: projected image float32 [H][W][C >= 5] (x, y, z, intensity, class) and optionally its normals [H][W][3] ->
 *   range [H][W] = |xyz|, reflectivity [H][W], xyz [3][H][W], normals_chw [3][H][W], labels int64 [H][W] (:83-99). */
int slu_kitti_decode(const float* xyzi, const uint32_t* label, int N, const int32_t* lut, int lut_size, int rotate, double cos_a, double sin_a,
                     double* pc, int32_t* bad_count, slu_stream_t stream);
/* SemanticKitti(resize=True) (dataloader_semantic_KITTI.py:61-62): cv2.resize(img, (OW, OH), interpolation=cv2.INTER_NEAREST) of the projected
 * [H][W][C] image -- source index min(floor(dst * src_size / dst_size), src_size - 1) per axis, OpenCV's definition (cv2 is absent from the
 * build image: unpinned like the normals); flip != 0: the dataloader's flip augmentation applied after it (:71-73). */
int slu_resize_nearest_hwc(const float* img, int H, int W, int C, float* out, int OH, int OW, int flip, slu_stream_t stream);
int slu_range_image_split(const float* img, const float* normals, int H, int W, int C, float* range, float* refl, float* xyz, float* normals_chw,
                          int64_t* labels, slu_stream_t stream);

/* ---- surface normals of the projected image (dataset/utils.py:30-58 build_normal_xyz; inference_ouster.py:70) ------
 * xyz: fp32 [H][W][channels >= 3] (x, y, z first); normals: fp32 [H][W][3] = -(d xyz/d col x d xyz/d row) / (|.| + 1e-10) with the 3x3
 * Scharr derivatives of OpenCV (cv2.Scharr(..., scale = 1 / norm_factor), BORDER_REFLECT_101). */
int slu_build_normals(const float* xyz, int H, int W, int channels, float norm_factor, float* normals, slu_stream_t stream);

/* ---- training path of the ResNet-FPN models (SURVEY 8(a) row a3 backward, 8(b) Autograd row; models/semanticFCN.py:266-354,
 * baselines/Reichert/semanticFCN_opt.py:366-455; trainer.py:783-786 calls loss.backward() on them): per-pixel / data-movement kernels of
 * the autograd nodes in semanticlidarunc_amd/fpn_autograd.py.  Dense fp32 NCHW tensors, device pointers.
 * slu_pointwise_{fwd,bwd}: op 0 = LeakyReLU(slope >= 0; 0 = nn.ReLU), 1 = tanh (AttentionModule, semanticFCN.py:32), 2 = ELU(alpha 1) + 1
 *   (decoder_semantic + 1, :352).  The backward takes the forward's OUTPUT y: dx = dy * f'(x(y)).
 * slu_maxpool3s2_bwd: nn.MaxPool2d(3, 2, 1) (stem, :149); the gradient goes to the FIRST maximum of a window in row-major order, as ATen does.
 * slu_nearest_down_bwd: backward of F.interpolate(mode='nearest', scale_factor=1/factor) (:283-285): dx [N][C][H][W] from dy [N][C][H/f][W/f].
 * slu_replace_tail_{fwd,bwd}: out = cat(x[:, :C-m], meta) (:309-313); dx (zero in the replaced channels) and / or dmeta may be NULL.
 * slu_row_softmax_mul_bwd: out = value * softmax_W(score) (:35-38); dscore [N][1][H][W] and / or dvalue may be NULL.  W <= 4096.
 * slu_depth_to_space_bwd: dx [N][Cout r r][H][W] from the channel slice [c_off, c_off + Cout) of dy [N][Ctot][H r][W r] (slu_depth_to_space).
 * slu_bilinear_upsample_bwd: backward of slu_bilinear_upsample (align_corners = False); dx [N][C][H][W] from dy [N][C][H s][W s].
 * slu_groupnorm_bwd: x, dy [N][C][HW]; mean / rstd [N groups] from slu_groupnorm_fwd; relu != 0: the forward applied ReLU and y is its output;
 *   dgamma / dbeta: float64 [C], ADDED to (zero them first), may be NULL.
 * slu_spatial_softmax_gate_bwd: backward of slu_spatial_softmax_gate with its stats [N][2]; workspace: float [N][HW]. */
int slu_pointwise_fwd(const float* x, float* y, size_t n, int op, float slope, slu_stream_t stream);
int slu_pointwise_bwd(const float* dy, const float* y, float* dx, size_t n, int op, float slope, slu_stream_t stream);
int slu_maxpool3s2_bwd(const float* x, const float* dy, float* dx, int N, int C, int H, int W, slu_stream_t stream);
int slu_nearest_down_bwd(const float* dy, float* dx, int N, int C, int H, int W, int factor, slu_stream_t stream);
int slu_replace_tail_fwd(const float* x, const float* meta, float* out, int N, int C, int m, int H, int W, slu_stream_t stream);
int slu_replace_tail_bwd(const float* dout, float* dx, float* dmeta, int N, int C, int m, int H, int W, slu_stream_t stream);
int slu_row_softmax_mul_bwd(const float* score, const float* value, const float* dout, float* dscore, float* dvalue, int N, int C, int H, int W,
                            slu_stream_t stream);
int slu_depth_to_space_bwd(const float* dy, float* dx, int N, int Cout, int H, int W, int r, int c_off, int Ctot, slu_stream_t stream);
int slu_bilinear_upsample_bwd(const float* dy, float* dx, int N, int C, int H, int W, int scale, slu_stream_t stream);
int slu_groupnorm_bwd(const float* x, const float* y, const float* dy, const float* gamma, const float* mean, const float* rstd, float* dx,
                      double* dgamma, double* dbeta, int N, int C, int HW, int groups, int relu, slu_stream_t stream);
int slu_spatial_softmax_gate_bwd(const float* x, const float* score, const float* stats, const float* dout, float* dx, float* dscore,
                                 float* workspace, int N, int C, int HW, slu_stream_t stream);

/* ---- all Dropout2d multipliers of one MC-dropout evaluation in one launch (SalsaNext.py:98,106,145,149,168; utils/mc_dropout.py:13-34) ----
 * A site is one nn.Dropout2d application: an [N][C] table of multipliers, 0 with probability p, else 1 / (1 - p); inactive sites give 1.
 * The draws are Philox4x32-10(key = seed, counter = offset + g / 4)[g % 4] for the g-th element of the call (sites in table order, `begin` =
 * first g of the site), i.e. a pure function of torch's generator state: the host reads (seed, offset) from the CUDA generator and advances
 * the offset by ceil(total draws / 4).  An output is an [N][C] table buf[begin ...] = product over its <= 3 sites:
 *   site_a[n][off_a + c] * site_b[n][off_b + c'] * site_c[n][off_c + c'],  c' = shuffled ? c / 4 : c  (stored channel c of a tensor read
 *   through PixelShuffle(2) feeds shuffled channel c / 4: UpBlock's dropout1 / dropout2 act on the shuffled tensor, SalsaNext.py:141-149);
 * a site index of -1 drops the factor.  `sites`, `outs`: DEVICE arrays; outs sorted by `begin`; total = sum of N * C over the outputs. */
typedef struct slu_dropout_site {
  long long begin;
  int C, active;
  float p;
  int reserved;
} slu_dropout_site;
typedef struct slu_dropout_out {
  long long begin;
  int C, site_a, off_a, site_b, off_b, site_c, off_c, shuffled;
} slu_dropout_out;
int slu_dropout_draw(const slu_dropout_site* sites, int nsites, const slu_dropout_out* outs, int nout, int N, unsigned long long seed,
                     unsigned long long offset, float* buf, long long total, slu_stream_t stream);

/* ---- EfficientNetV2 blocks of the semanticFCN_opt encoder (baselines/Reichert/semanticFCN_opt.py:170-180,238-247,396-404;
 * the block structure is torchvision's efficientnet_v2_{s,m,l}: FusedMBConv / MBConv with depthwise 3x3, squeeze-excitation, SiLU) -----------
 * slu_dwconv3x3_fwd: depthwise 3x3, padding 1, stride 1 or 2, with the eval BatchNorm folded into w [C][9] / bias [C]; act 0 none, 3 SiLU.
 *   x [N][C][H][W] -> y [N][C][ceil(H/stride)][ceil(W/stride)].
 * slu_global_avgpool: mean over H W -> out [N][C] (SqueezeExcitation.avgpool).
 * slu_se_gate: scale [N][C] = sigmoid(w2 [C][S] . SiLU(w1 [S][C] . avg + b1) + b2)  (fc1 / fc2 of SqueezeExcitation are 1x1 convs on a 1x1 map);
 *   the scale reaches the block's projection conv as a per-(sample, channel) input multiplier (slu_conv_src.scale).
 * slu_dwconv3x3_wgrad (training, trainer.py:783-786 backward through the MBConv blocks): dw [C][9] = sum over (n, y, x) of dy[n][c][y][x] *
 *   x[n][c][y + i - 1][x + j - 1] for the stride-1 depthwise conv (a stride-2 one is run as stride 1 + sub-sampling in training); the data
 *   gradient is slu_dwconv3x3_fwd on dy with the nine taps reversed. */
int slu_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, int N, int C, int H, int W, slu_stream_t stream);
int slu_dwconv3x3_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int stride, int act, slu_stream_t stream);
int slu_global_avgpool(const float* x, float* out, int N, int C, int HW, slu_stream_t stream);
int slu_se_gate(const float* avg, const float* w1, const float* b1, const float* w2, const float* b2, float* scale, int N, int C, int S,
                slu_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SLU_H_ */
