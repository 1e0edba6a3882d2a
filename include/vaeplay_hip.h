/* vaeplay_hip.h -- C ABI of libvaeplay_hip.so, the MI355X (gfx950) back end of the
 * convolutional-VAE training step of kungyao/vae-play.
 *
 * The reference has no FFI of its own (pure PyTorch): the boundary it exposes is
 * nn.Module.forward / loss / optimizer.step.  Every entry point below replaces the ATen/cuDNN
 * call that a reference line dispatches; the citation after each prototype names that line
 * (paths relative to the reference checkout).  The host binds them with ctypes
 * (vae_play_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller
 *     (PyTorch's allocator); the library never allocates or frees device memory and keeps no
 *     pointer after return;
 *   - activations are NHWC fp32 (= torch channels_last), weights are passed in the reference's
 *     own layouts unless a prototype says "packed";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); no internal sync;
 *   - return 0 on success, a negative vp_status otherwise; vp_last_error() gives the message
 *     for the calling thread; nothing throws across the ABI;
 *   - `ws`/`ws_bytes`: caller-provided scratch, size from the matching *_workspace_bytes().
 */
#ifndef VAEPLAY_HIP_H
#define VAEPLAY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vp_stream; /* hipStream_t */

enum vp_status { VP_OK = 0, VP_ERR_ARG = -1, VP_ERR_LAUNCH = -2, VP_ERR_WORKSPACE = -3, VP_ERR_STATE = -4 };
enum vp_act { VP_ACT_NONE = 0, VP_ACT_RELU = 1, VP_ACT_LRELU = 2, VP_ACT_TANH = 3, VP_ACT_SIGMOID = 4 };

int vp_abi_version(void);
const char* vp_last_error(void);

/* ---- layout ------------------------------------------------------------------------------ */
/* NCHW <-> NHWC copies (module I/O is NCHW like the reference's tensors, datasets/dataset.py:68) */
int vp_nchw_to_nhwc_f32(const float* in, float* out, int B, int C, int H, int W, vp_stream stream);
int vp_nhwc_to_nchw_f32(const float* in, float* out, int B, int C, int H, int W, vp_stream stream);

/* 5x5 weight repack.  w_ref is the reference tensor W[Csmall][Cbig][5][5]
 * (nn.Conv2d weight (Cout,Cin,5,5) models/networks.py:14; nn.ConvTranspose2d weight
 * (Cin,Cout,5,5) models/networks.py:38).  p0 = [Csmall][25][Cbig], p1 = [Cbig][25][Csmall];
 * either output may be NULL. */
int vp_pack_w5_f32(const float* w_ref, float* p0, float* p1, int Csmall, int Cbig, vp_stream stream);

/* ---- 5x5 pad-2 convolution families (stride 1 or 2; big = stride * small) ------------------ */
/* small[B,Hs,Ws,Cs] = act(bias + conv5(big[B,s*Hs,s*Ws,Cb]))
 *   replaces nn.Conv2d fwd models/networks.py:27 (EncoderBlock), :100-103 (final conv + Sigmoid,
 *   act = VP_ACT_SIGMOID) and nn.ConvTranspose2d's input gradient (autograd of :43). */
int vp_conv5_gather_f32(const float* big, const float* w_p0, const float* bias, float* small_out,
                        int B, int Hs, int Ws, int Cbig, int Csmall, int stride, int act, vp_stream stream);
/* big[B,s*Hs,s*Ws,Cb] = convT5(small)  (sub-pixel phases, no zero insertion)
 *   replaces nn.ConvTranspose2d fwd models/networks.py:43 (DecoderBlock) and nn.Conv2d's input
 *   gradient (autograd of :27, :100). */
int vp_conv5_scatter_f32(const float* small, const float* w_p1, float* big_out,
                         int B, int Hs, int Ws, int Csmall, int Cbig, int stride, vp_stream stream);
/* dW[Cs][Cb][5][5] (reference layout) = sum_pixels small (x) shifted big
 *   replaces the weight gradient of both layer kinds (autograd of models/networks.py:27,:43,:100). */
size_t vp_conv5_wgrad_workspace_bytes(int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int vp_conv5_wgrad_f32(const float* big, const float* small, float* dw_ref,
                       int B, int Hs, int Ws, int Cbig, int Csmall, int stride,
                       void* ws, size_t ws_bytes, vp_stream stream);
/* As vp_conv5_wgrad_f32 with a bound on the CUs the launch occupies (see vp_conv5_wgrad_bf16x3_cus; max_cus <= 0: the whole chip). */
int vp_conv5_wgrad_f32_cus(const float* big, const float* small, float* dw_ref, int B, int Hs, int Ws, int Cbig, int Csmall,
                           int stride, int max_cus, void* ws, size_t ws_bytes, vp_stream stream);
/* Exact-fp32 convolution + the statistics pass of the BatchNorm that follows it (models/networks.py:14-16,38-40) in one call, as
 * vp_conv5_*_stats_bf16x3: {pivot, sum(x - pivot), sum((x - pivot)^2)} per (workgroup, channel) from the fp32 accumulators, one
 * finaliser launch for mean / rstd / running statistics (momentum semantics of torch).  vp_conv5_stats_f32_workspace_bytes() == 0:
 * the shape cannot emit them (it splits K, or does not take the fast fp32 path) -- use vp_conv5_*_f32 + vp_bn_stats_f32. */
size_t vp_conv5_stats_f32_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int vp_conv5_gather_stats_f32(const float* big, const float* w_p0, float* small_out, int B, int Hs, int Ws, int Cbig, int Csmall, int stride,
                              float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var, void* ws,
                              size_t ws_bytes, vp_stream stream);
int vp_conv5_scatter_stats_f32(const float* small, const float* w_p1, float* big_out, int B, int Hs, int Ws, int Csmall, int Cbig, int stride,
                               float eps, float momentum, float* mean, float* rstd, float* running_mean, float* running_var, void* ws,
                               size_t ws_bytes, vp_stream stream);

/* ---- k x k generalisation (ks = 1, 3 or 5, padding (ks-1)/2, stride 1 or 2) -------------------------------------
 * The conv/norm/act vocabulary of models/blocks.py:5-34 (nn.Conv2d(k, stride, padding=(k-1)//2)); same three families.
 * The big side is passed explicitly (Hb, Wb) so odd sizes work: Hs = floor((Hb + 2*pad - ks)/stride) + 1.
 * Weights: reference tensor W[Csmall][Cbig][ks][ks]; packed p0 = [Csmall][ks*ks][Cbig], p1 = [Cbig][ks*ks][Csmall]. */
int vp_pack_w_f32(const float* w_ref, float* p0, float* p1, int Csmall, int Cbig, int ks, vp_stream stream);
int vp_conv_gather_f32(const float* big, const float* w_p0, const float* bias, float* small_out,
                       int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream);
int vp_conv_scatter_f32(const float* small, const float* w_p1, float* big_out,
                        int B, int Hs, int Ws, int Hb, int Wb, int Csmall, int Cbig, int ks, int stride, vp_stream stream);
size_t vp_conv_wgrad_workspace_bytes(int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride);
int vp_conv_wgrad_f32(const float* big, const float* small, float* dw_ref,
                      int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride,
                      void* ws, size_t ws_bytes, vp_stream stream);

/* k x k (k = 1, 3, 5; padding (k-1)/2; explicit big size Hb x Wb as for vp_conv_gather_f32) forms of the three split-bf16
 * families, for the models/blocks.py vocabulary; weights packed by vp_pack_w_split: p0 = [Csmall][k*k][Cbig],
 * p1 = [Cbig][k*k][Csmall], both as bf16 hi/lo planes. */
int vp_pack_w_split(const float* w_ref, void* p0_split, void* p1_split, int Csmall, int Cbig, int ks, vp_stream stream);
int vp_conv_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                          int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, vp_stream stream);
int vp_conv_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                           int Csmall, int Cbig, int ks, int stride, vp_stream stream);
size_t vp_conv_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Hb, int Wb, int Cbig, int Csmall, int ks, int stride);
int vp_conv_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                         int Csmall, int ks, int stride, void* ws, size_t ws_bytes, vp_stream stream);
/* ---- split-bf16 ("bf16x3") variants of the three families ------------------------------------------
 * A "split" tensor stores an fp32 tensor of n elements as two bf16 planes in one buffer of 2*n
 * uint16: hi = bf16(x) at [0,n), lo = bf16(x - hi) at [n,2n).  The contraction issues three
 * v_mfma_f32_32x32x16_bf16 per fragment pair (lo*hi + hi*lo + hi*hi) with fp32 accumulation:
 * ~5e-6 relative error per contraction.  Channel counts on the contracted/vector side must be
 * multiples of 8; outputs are plain fp32.  Same reference lines as the f32 entry points above. */
int vp_split_f32(const float* x, void* out_split, size_t n, vp_stream stream);
/* the same for an NHWC tensor [npix][C] whose channel count is not a multiple of 8: planes [npix][Cpad], channels C..Cpad-1
 * zero (models/blocks.py:97-146 AddCoords makes 32 + 2 channels; the heads of models/networks_BE.py:39-89 end in 1 channel) */
int vp_split_pad_f32(const float* x, void* out_split, size_t npix, int C, int Cpad, vp_stream stream);
int vp_pack_w5_split(const float* w_ref, void* p0_split, void* p1_split, int Csmall, int Cbig, vp_stream stream);
int vp_conv5_gather_bf16x3(const void* big_split, const void* w_p0_split, const float* bias, float* small_out,
                           int B, int Hs, int Ws, int Cbig, int Csmall, int stride, int act, vp_stream stream);
int vp_conv5_scatter_bf16x3(const void* small_split, const void* w_p1_split, float* big_out,
                            int B, int Hs, int Ws, int Csmall, int Cbig, int stride, vp_stream stream);
/* Final conv + bias + sigmoid (nn.Conv2d(64, C, k5, s1, p2) + nn.Sigmoid, models/networks.py:100-103, C = 1 or 3) in split-bf16
 * arithmetic on the matrix cores: the 25 taps x C outputs are the MFMA column dimension ("tap-in-N", csrc/narrow.hip); fp32 NHWC
 * input (split into bf16 hi/lo in registers) and the fp32 packed weights P0 [C][25][64] of vp_pack_w5_f32, fp32 output. */
int vp_conv5_smallout_bf16x3(const float* big, const float* w_p0, const float* bias, float* small_out, int B, int H, int W,
                             int Cbig, int Csmall, int act, vp_stream stream);
/* ... and its weight gradient dW[C][64][5][5] = sum_pixels dlogit x shifted activation, with the 25 taps x C gradient channels as
 * the MFMA row dimension ("taps in M", csrc/edge.hip): fp32 activation [B,H,W,64] and fp32 dlogit [B,H,W,C] in, reference weight
 * layout out; deterministic slab reduction.  The workspace query returns 0 for shapes it does not take (width not a multiple of
 * 64, height not of 16): use vp_conv5_wgrad_f32 there. */
size_t vp_conv5_smallout_wgrad_bf16x3_workspace_bytes(int B, int H, int W, int Cbig, int Csmall);
int vp_conv5_smallout_wgrad_bf16x3(const float* big, const float* small, float* dw_ref, int B, int H, int W, int Cbig, int Csmall,
                                   void* ws, size_t ws_bytes, vp_stream stream);
/* The same weight gradient in exact fp32 (v_mfma_f32_32x32x2_f32, the dlogit patch and u as fp32 in LDS, the pixels of a super-step
 * split over the four waves and their partial sums added in a fixed order): same workspace size, same slab layout and reduction. */
size_t vp_conv5_smallout_wgrad_f32_workspace_bytes(int B, int H, int W, int Cbig, int Csmall);
int vp_conv5_smallout_wgrad_f32(const float* big, const float* small, float* dw_ref, int B, int H, int W, int Cbig, int Csmall,
                                void* ws, size_t ws_bytes, vp_stream stream);
/* First encoder conv (nn.Conv2d(C, 64, k5, s2, p2, bias=False) with C = 1 or 3 image channels, models/networks.py:14 via :55):
 * its im2col is materialised once per step as split planes [B*Hs*Ws][KC] (KC = vp_im2col5s2_cols(C): 96 / 64), after which the
 * forward convolution is vp_conv_gather_bf16x3(ks = 1, Cbig = KC) and the weight gradient vp_conv_wgrad_bf16x3(ks = 1) on the
 * MFMA kernels; the packed weight / the gradient use the same column order (r*GW + q*C + cin) and are converted from / to the
 * reference layout [64][C][5][5] by the two small kernels below.  x is NCHW (nchw = 1) or NHWC fp32. */
int vp_im2col5s2_cols(int C);
int vp_im2col5s2_split_f32(const float* x, void* out_split, int B, int C, int Hb, int Wb, int nchw, vp_stream stream);
int vp_pack_w_im2col5_split(const float* w_ref, void* out_split, int Cout, int C, vp_stream stream);
int vp_unpack_dw_im2col5_f32(const float* dw_cols, float* dw_ref, int Cout, int C, vp_stream stream);
/* The same im2col and weight re-ordering as plain fp32 ([B*Hs*Ws][KC] and [Cout][KC]) for the exact-f32 plan: the first conv and its
 * weight gradient then run as 1x1 layers through vp_conv_gather_f32 / vp_conv_wgrad_f32 (ks = 1) on the fp32 matrix cores. */
int vp_im2col5s2_f32(const float* x, float* out, int B, int C, int Hb, int Wb, int nchw, vp_stream stream);
int vp_pack_w_im2col5_f32(const float* w_ref, float* out, int Cout, int C, vp_stream stream);
/* Convolution + BatchNorm batch statistics in one call (replaces nn.Conv2d / nn.ConvTranspose2d followed by the statistics
 * pass of nn.BatchNorm2d(momentum=0.9), models/networks.py:14-16,27-28 and :38-40,43-44): the convolution's epilogue emits
 * per-workgroup {pivot, sum(x - pivot), sum((x - pivot)^2)} per output channel from its accumulators and one finaliser
 * launch produces mean / rstd and updates the running buffers exactly as vp_bn_stats_f32 does -- the activation is not read
 * again for its statistics.  family: 0 = gather, 1 = scatter.  vp_conv5_stats_workspace_bytes() == 0 means that this launch
 * shape cannot emit statistics (split-K layers, the narrow-channel halo kernels): use the plain entry point + vp_bn_stats_f32. */
size_t vp_conv5_stats_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int vp_conv5_gather_stats_bf16x3(const void* big_split, const void* w_p0_split, float* small_out,
                                 int B, int Hs, int Ws, int Cbig, int Csmall, int stride, float eps, float momentum,
                                 float* mean, float* rstd, float* running_mean, float* running_var,
                                 void* ws, size_t ws_bytes, vp_stream stream);
int vp_conv5_scatter_stats_bf16x3(const void* small_split, const void* w_p1_split, float* big_out,
                                  int B, int Hs, int Ws, int Csmall, int Cbig, int stride, float eps, float momentum,
                                  float* mean, float* rstd, float* running_mean, float* running_var,
                                  void* ws, size_t ws_bytes, vp_stream stream);
/* Input gradient of the final conv (nn.Conv2d(64 -> C, k5, s1, p2), models/networks.py:100-103, autograd backward) for C = 1 | 3
 * image channels: big_out[b,h,w,cf] = sum_{r,q,n} small[b, h-r+2, w-q+2, n] * w_ref[n][cf][r][q], fp32 NHWC operands, reference weight
 * layout, split-bf16 arithmetic on the matrix cores with one kernel row of taps (5*C contiguous floats of `small`) per MFMA k-step. */
int vp_conv5_smallin_dgrad_bf16x3(const float* small, const float* w_ref, float* big_out, int B, int H, int W, int Csmall, int Cbig,
                                  vp_stream stream);
/* The same input gradient (models/networks.py:100-103 backward) in exact fp32: v_mfma_f32_32x32x2_f32 on the same rows-in-K tiling. */
int vp_conv5_smallin_dgrad_f32(const float* small, const float* w_ref, float* big_out, int B, int H, int W, int Csmall, int Cbig,
                               vp_stream stream);
/* Forward pass of a stride-1 first conv on a 1- or 3-channel image (nn.Conv2d(C, 32 | 64, k5, s1, p2) + ReLU, the VAE-GAN
 * discriminator's first layer, models/networks.py:160-163): big_out[b,h,w,cf] = act(bias[cf] + sum small[b,h+r-2,w+q-2,n] * w_ref[cf][n][r][q]),
 * fp32 NHWC operands, reference weight layout, act none | relu; the same rows-in-K kernel as vp_conv5_smallin_dgrad_bf16x3. */
int vp_conv5_smallin_fwd_bf16x3(const float* small, const float* w_ref, const float* bias, float* big_out, int B, int H, int W, int Csmall,
                                int Cbig, int act, vp_stream stream);
size_t vp_conv5_wgrad_bf16x3_workspace_bytes(int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int vp_conv5_wgrad_bf16x3(const void* big_split, const void* small_split, float* dw_ref,
                          int B, int Hs, int Ws, int Cbig, int Csmall, int stride,
                          void* ws, size_t ws_bytes, vp_stream stream);
/* The same with a bound on the CUs the launch occupies.  The row-of-taps kernel (csrc/wgrad5.h: one kernel row of 5 taps per
 * workgroup, replaces cuDNN's weight gradient behind models/networks.py:14,38) owns a whole CU per work item; a caller that runs the
 * weight gradient BESIDE other kernels -- the fused step's side stream -- leaves the rest of the chip to them (160 of 256 CUs is the
 * measured optimum there).  max_cus <= 0: the whole chip (= vp_conv5_wgrad_bf16x3).  Same workspace query, same result up to the
 * summation order of the pixel ranges. */
int vp_conv5_wgrad_bf16x3_cus(const void* big_split, const void* small_split, float* dw_ref,
                              int B, int Hs, int Ws, int Cbig, int Csmall, int stride, int max_cus,
                              void* ws, size_t ws_bytes, vp_stream stream);
/* Weight gradient of a 3x3 / stride 1 / padding 1 nn.Conv2d with a handful of channels on both sides (Cin <= 40, Cout <= 8: the mask /
 * edge heads of models/networks_BE.py:39-66 via models/blocks.py:9-17), exact fp32 on the vector ALUs: x (B, H, W, Cin) and dy (B, H, W,
 * Cout) NHWC, dw_ref (Cout, Cin, 3, 3); bit-reproducible.  vp_conv3_small_wgrad_workspace_bytes() == 0: shape not taken. */
size_t vp_conv3_small_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int vp_conv3_small_wgrad_f32(const float* x, const float* dy, float* dw_ref, int B, int H, int W, int Cin, int Cout,
                             void* ws, size_t ws_bytes, vp_stream stream);
/* ... and their forward pass / input gradient: y = bias + conv3x3(x, w) and dx = conv3x3^T(dy, w), one output pixel per thread, exact
 * fp32, reading the reference weight layout (Cout, Cin, 3, 3) directly (no packing, no channel padding). */
int vp_conv3_small_fwd_f32(const float* x, const float* w_ref, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                           vp_stream stream);
int vp_conv3_small_dgrad_f32(const float* dy, const float* w_ref, float* dx, int B, int H, int W, int Cin, int Cout, vp_stream stream);
/* Second half of every weight-gradient entry point above (the gradient of nn.Conv2d / nn.ConvTranspose2d weights,
 * models/networks.py:14,38): the K-split launch leaves slab[split][tap][Csmall][Cbig]; this sums the splits in a fixed order
 * (even splits, odd splits, their sum: bit-reproducible, no atomics) and writes the reference layout dw[Csmall][Cbig][tap].
 * variant: -1 = what the library dispatches, 0 = 4-B loads, 1 = 16-B loads (needs Csmall * Cbig % 64 == 0 and 16-B aligned
 * pointers); every variant returns the same bits.  Exposed for callers that split K themselves and for the parity tests. */
int vp_wgrad_slab_reduce_f32(const float* slab, float* dw_ref, int Csmall, int Cbig, int nsplit, int ntaps, int variant,
                             vp_stream stream);
/* producers of split tensors fused into the elementwise passes (y / dx / out may be NULL when only
 * the split copy is wanted) */
/* nn.BatchNorm1d(momentum=0.9) + activation behind nn.Linear (models/networks.py:66-67,89-90), forward and backward, for at most
 * 64 rows (the batch): statistics, finalisation and the normalised output / the two channel sums and the input gradient in ONE
 * launch each, bit-identical to vp_bn_stats_f32 + vp_bn_act_fwd_f32 and to vp_bn_act_bwd_f32 (same arithmetic, step for step). */
int vp_bn_small_fwd_f32(const float* x, int R, int C, float eps, float momentum, const float* gamma, const float* beta, float* mean,
                        float* rstd, float* running_mean, float* running_var, float* y, int act, float slope, vp_stream stream);
int vp_bn_small_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        float* dx, float* dgamma, float* dbeta, int R, int C, int act, float slope, int batch_stats, vp_stream stream);
int vp_bn_act_fwd_split_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                            float* y, void* y_split, int R, int C, int act, float slope, vp_stream stream);
int vp_bn_act_bwd_split_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                            const float* gamma, const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta,
                            int R, int C, int act, float slope, int batch_stats,
                            void* ws, size_t ws_bytes, vp_stream stream);
int vp_nchw_to_nhwc_split_f32(const float* in, float* out, void* out_split, int B, int C, int H, int W, vp_stream stream);
/* Every conv weight of a step in ONE launch: job i packs w (reference layout [Csmall][Cbig][5][5]) into p0
 * ([Csmall][25][Cbig]) and/or p1 ([Cbig][25][Csmall_pad]); split = 1 writes bf16 hi/lo planes, 2 fp16 hi/lo planes, 0 fp32.
 * jobs is a HOST array (read during the call only); at most 32 output layouts per batch. */
typedef struct vp_pack_job {
  const float* w;
  void* p0;
  void* p1;
  int Csmall, Cbig, Csmall_pad, split;
} vp_pack_job;
int vp_pack_w5_batch(const vp_pack_job* jobs, int njobs, vp_stream stream);
/* edge layers (1/3 image channels) on the bf16x3 path: the small-channel dimension zero-padded to a multiple of 8.
 * p1_split = [Cbig][25][Csmall_pad] planes; dlogit_split = [npix][Cpad] planes next to the fp32 dlogit [npix][C]. */
int vp_pack_w5_p1_split_padded(const float* w_ref, void* p1_split, int Csmall, int Cbig, int Csmall_pad, vp_stream stream);
int vp_bce_sigmoid_bwd_pad_split_f32(const float* p, const float* t, float gscale, float* dlogit, void* dlogit_split,
                                     size_t npix, int C, int Cpad, vp_stream stream);

/* ---- dense layers ------------------------------------------------------------------------ */
/* C[m][n] = bias[n] + sum_k A(m,k) B(n,k) with element strides (sam,sak) / (sbn,sbk).
 * mode 0: both operands k-contiguous (Linear fwd, models/networks.py:75-77,109)
 * mode 1: A k-contiguous, B n-contiguous (Linear input gradient)
 * mode 2: A m-contiguous, B n-contiguous (Linear weight gradient) */
size_t vp_gemm_workspace_bytes(int M, int N, int K);
int vp_gemm_f32(const float* A, long sam, long sak, const float* B, long sbn, long sbk,
                float* C, int ldc, const float* bias, int M, int N, int K, int mode,
                void* ws, size_t ws_bytes, vp_stream stream);
/* out[c] = sum_r x[r][c]   (bias gradients) */
size_t vp_colsum_workspace_bytes(int R, int C);
int vp_colsum_f32(const float* x, float* out, int R, int C, void* ws, size_t ws_bytes, vp_stream stream);

/* ---- BatchNorm (+ activation) over an [R][C] NHWC view (R = B*H*W, or B for BatchNorm1d) ---- */
/* replaces nn.BatchNorm2d/1d(momentum=0.9) + F.relu  models/networks.py:16,28-29,40,44-45,66-67,89-90;
 * also the norm/act vocabulary of models/blocks.py:19-30. */
size_t vp_bn_workspace_bytes(int R, int C);
/* batch statistics: mean[c], rstd[c] = 1/sqrt(biased_var+eps); if running_* != NULL they are
 * updated in place: run = (1-momentum)*run + momentum*batch (unbiased var), torch semantics. */
int vp_bn_stats_f32(const float* x, int R, int C, float eps, float momentum,
                    float* mean, float* rstd, float* running_mean, float* running_var,
                    void* ws, size_t ws_bytes, vp_stream stream);
/* y = act(gamma*(x-mean)*rstd + beta); gamma/beta may be NULL (affine=False). */
int vp_bn_act_fwd_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      float* y, int R, int C, int act, float slope, vp_stream stream);
/* backward through act + BN.  batch_stats=1: training-mode formula (statistics depend on x);
 * batch_stats=0: eval mode (mean/rstd constants).  dgamma/dbeta may be NULL.  dx may alias dy. */
int vp_bn_act_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                      const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                      int R, int C, int act, float slope, int batch_stats,
                      void* ws, size_t ws_bytes, vp_stream stream);
/* nn.InstanceNorm2d(affine=False, eps) + activation over B images of [R = H*W][C] NHWC (models/blocks.py:22): per-(image,
 * channel) statistics; mean / rstd [B][C] are outputs of the forward and inputs of the backward. */
size_t vp_instnorm_workspace_bytes(int B, int R, int C);
int vp_instnorm_act_fwd_f32(const float* x, float* y, float* mean, float* rstd, int B, int R, int C, float eps, int act,
                            float slope, void* ws, size_t ws_bytes, vp_stream stream);
int vp_instnorm_act_bwd_f32(const float* x, const float* dy, const float* mean, const float* rstd, float* dx, int B, int R, int C,
                            int act, float slope, void* ws, size_t ws_bytes, vp_stream stream);
/* the same with the bf16 hi/lo planes of y / dx written by the apply pass (C % 4 == 0; plane = B * R * C elements): the operand
 * of the split-bf16 convolution that consumes the normalised activation / its input gradient (models/blocks.py:22-30) */
int vp_instnorm_act_fwd_split_f32(const float* x, float* y, void* y_split, float* mean, float* rstd, int B, int R, int C, float eps,
                                  int act, float slope, void* ws, size_t ws_bytes, vp_stream stream);
int vp_instnorm_act_bwd_split_f32(const float* x, const float* dy, const float* mean, const float* rstd, float* dx, void* dx_split, int B,
                                  int R, int C, int act, float slope, void* ws, size_t ws_bytes, vp_stream stream);
/* plain activation (conv + bias + act blocks of models/blocks.py:24-30 with bn=None) */
int vp_act_fwd_f32(const float* x, float* y, size_t n, int act, float slope, vp_stream stream);
/* dx = dy * act'(.) evaluated from the OUTPUT y (relu/lrelu/tanh/sigmoid); dx may alias dy */
int vp_act_bwd_from_y_f32(const float* y, const float* dy, float* dx, size_t n, int act, float slope, vp_stream stream);

/* ---- models/blocks.py helpers (NHWC) ------------------------------------------------------------------------- */
/* F.interpolate(scale_factor=2, mode='bilinear') of blocks.Up (models/blocks.py:145): y[B,2H,2W,C]; and its adjoint */
int vp_upsample2x_bilinear_fwd_f32(const float* x, float* y, int B, int H, int W, int C, vp_stream stream);
int vp_upsample2x_bilinear_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, vp_stream stream);
/* AddCoords (models/blocks.py:97-112): out[B,H,W,C+2] = cat(x, column index, row index); normalize as if_normalize */
int vp_add_coords_f32(const float* x, float* out, int B, int H, int W, int C, int normalize, vp_stream stream);
/* out[p][0..Cout) = in[p][0..Cout): gradient of AddCoords / channel slice */
int vp_slice_channels_f32(const float* in, float* out, size_t npix, int Cin, int Cout, vp_stream stream);

/* ---- latent: reparameterisation + KL -------------------------------------------------------- */
/* z = eps*exp(0.5*logvar) + mu   (models/networks.py:228-231);
 * kl[b] = -0.5*sum_j(1 + logvar - mu^2 - exp(logvar))  (models/networks.py:270); kl may be NULL */
int vp_latent_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, float* kl,
                      int B, int Z, vp_stream stream);
/* dmu = dz + gkl[b]*mu ; dlogvar = dz*eps*0.5*exp(0.5*logvar) + gkl[b]*0.5*(exp(logvar)-1)
 * dz may be NULL (KL term only); gkl is per-sample dL/dkl[b] (NULL = 0) */
int vp_latent_bwd_f32(const float* mu, const float* logvar, const float* eps, const float* dz, const float* gkl,
                      float gkl_scalar, float* dmu, float* dlogvar, int B, int Z, vp_stream stream);

/* ---- per-pixel BCE ------------------------------------------------------------------------ */
/* out[0] = sum_i -[t*max(log p,-100) + (1-t)*max(log(1-p),-100)]  (torch F.binary_cross_entropy,
 * reduction='sum'; call form train_BE_font.py:107).  p and t may have different memory order only
 * if the caller made them match. */
size_t vp_reduce_workspace_bytes(size_t n);
int vp_bce_sum_f32(const float* p, const float* t, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream);
/* dp = g * (p - t) / max(p*(1-p), 1e-12)   (torch's BCE backward) ; g read from device scalar gptr * gscale */
int vp_bce_bwd_f32(const float* p, const float* t, const float* gptr, float gscale, float* dp, size_t n, vp_stream stream);
/* fused sigmoid+BCE backward on the logits: dlogit = gscale * (p - t) */
int vp_bce_sigmoid_bwd_f32(const float* p, const float* t, float gscale, float* dlogit, size_t n, vp_stream stream);
/* ---- font network pieces (models/networks_BE_font.py, models/blocks.py:66-96, train_BE_font.py:158) ------------- */
/* nn.AdaptiveAvgPool2d((1,1)) on NHWC: out[b][c] = mean_p x[b][p][c]; bwd: dx[b][p][c] = dy[b][c] / HW */
int vp_global_avgpool_fwd_f32(const float* x_nhwc, float* out, int B, int HW, int C, vp_stream stream);
int vp_global_avgpool_bwd_f32(const float* dy, float* dx_nhwc, int B, int HW, int C, vp_stream stream);
/* nn.Softmax(dim=-1) over R rows of n; bwd: dx = y * (dy - sum_j dy_j y_j) */
int vp_softmax_rows_fwd_f32(const float* x, float* y, int R, int n, vp_stream stream);
int vp_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, int R, int n, vp_stream stream);
/* F.cross_entropy(logits, labels) with torch's defaults (mean over the R rows; labels int64 class indices): train_BE_GAN.py:135,159
 * (d_type_loss / g_type_loss), train_BE_font.py:109.  loss[0] = mean_r (logsumexp(x_r) - x_r[label_r]); prob (R x n) = softmax rows,
 * kept for the backward pass: dlogits = g[0] / R * (prob - onehot(labels)).  Bit-reproducible (fixed-order sum in fp64). */
int vp_cross_entropy_fwd_f32(const float* logits, const long long* labels, float* loss, float* prob, int R, int n, vp_stream stream);
int vp_cross_entropy_bwd_f32(const float* prob, const long long* labels, const float* gptr, float* dlogits, int R, int n, vp_stream stream);
/* F.l1_loss(a, b) (mean): out[0]; ws >= 2 * vp_reduce_workspace_bytes(n).  bwd: da = g * sign(a-b) / n, db = -da */
int vp_l1_mean_f32(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream);
int vp_l1_mean_bwd_f32(const float* a, const float* b, const float* gptr, float* da, float* db, size_t n, vp_stream stream);

/* ---- segmentation loss of train_BE.py:58-59: bce_weight * F.binary_cross_entropy_with_logits(x, t) (mean over B*n)
 * + compute_dice_loss(sigmoid(x), t, smooth) (tools/ops.py:12-19) for B samples of n logits each.
 * fwd: loss[0] and sums[B][4] = {sum bce, sum p*t, sum p, sum t} per sample (kept for the backward);
 * bwd: dlogits = g * d loss / d logits with g read from the device scalar gptr (NULL = 1). */
size_t vp_be_loss_workspace_bytes(int B, int n);
int vp_be_loss_fwd_f32(const float* logits, const float* targets, float* loss, float* sums, int B, int n, float bce_weight,
                       float smooth, void* ws, size_t ws_bytes, vp_stream stream);
int vp_be_loss_bwd_f32(const float* logits, const float* targets, const float* sums, const float* gptr, float* dlogits, int B,
                       int n, float bce_weight, float smooth, vp_stream stream);

/* plain dice loss on probabilities (tools/ops.py:178-185, used by edge_loss :187-215): 1 - mean_b (2 I_b + s)/(P_b + T_b + s);
 * sums[B][4] as above (entry 0 unused); workspace = vp_be_loss_workspace_bytes(B, n). */
int vp_dice_loss_fwd_f32(const float* probs, const float* targets, float* loss, float* sums, int B, int n, float smooth, void* ws,
                         size_t ws_bytes, vp_stream stream);
int vp_dice_loss_bwd_f32(const float* probs, const float* targets, const float* sums, const float* gptr, float* dprobs, int B, int n,
                         float smooth, vp_stream stream);

/* ---- 0.5*(a-b)^2 (VaeGan.loss, models/networks.py:267 "nle" per element, :273 "mse" summed per row) ---- */
int vp_half_sqdiff_f32(const float* a, const float* b, float* out, size_t n, vp_stream stream);
/* out[r] = sum_j 0.5*(a[r][j]-b[r][j])^2 for R contiguous rows of n_per_row elements */
int vp_half_sqdiff_rowsum_f32(const float* a, const float* b, float* out, int R, int n_per_row, vp_stream stream);
/* da = g*(a-b), db = -da (either may be NULL); g has one value per row (g_per_row=1) or per element (0) */
int vp_half_sqdiff_bwd_f32(const float* a, const float* b, const float* g, float* da, float* db, int R, int n_per_row,
                           int g_per_row, vp_stream stream);
/* ---- VAE-GAN loss heads for the fused VAE-GAN step (VaeGan.loss, models/networks.py:275-279; weights of train.py:63-66) ---- */
/* logit [3B] = discriminator scores before F.sigmoid (models/networks.py:190) of (original | reconstructed | sampled);
 * p_out = sigmoid(logit); sums[0..2] = sum_b -log(p + 1e-3) over the originals and sum_b -log(1 - p + 1e-3) over the reconstructed /
 * sampled rows (:275-277); dlogit = coef * d(sums[0] + sums[1] + sums[2]) / dlogit.  Any output may be NULL. */
int vp_gan_head_f32(const float* logit, int B, float coef, float* p_out, float* sums, float* dlogit, vp_stream stream);
/* loss[0] = scale * F.smooth_l1_loss(targets, cat(a, b, dim=1), reduction="sum") (models/networks.py:279, beta = 1) with targets
 * [B][n1 + n2], a [B][n1], b [B][n2] (DirectDecoder's two heads, models/networks.py:144-147); da, db = d loss / d a, b. */
int vp_smooth_l1_cat_f32(const float* targets, const float* a, const float* b, int B, int n1, int n2, float scale, float* loss,
                         float* da, float* db, vp_stream stream);
/* out[0] = sum x */
int vp_sum_f32(const float* x, size_t n, float* out, void* ws, size_t ws_bytes, vp_stream stream);
/* loss tail of the composed VAE step (F.binary_cross_entropy(x_tilde, x, reduction='sum') + the KL of
 * models/networks.py:270 summed over the batch, as train.py:63 does) in two launches: recon = sum of per-pixel BCE,
 * kl_sum = sum_b kl[b], loss = (recon + kl_sum) * loss_scale (device scalars; loss_scale = 1/B gives the per-image loss). */
int vp_vae_loss_f32(const float* x_tilde, const float* x, size_t n, const float* kl, int B, float* recon, float* kl_sum,
                    float* loss, float loss_scale, void* ws, size_t ws_bytes, vp_stream stream);
/* out = a + b (the two latent heads' input gradients, models/networks.py:76-77 backward) */
int vp_add_f32(const float* a, const float* b, float* out, size_t n, vp_stream stream);

/* ---- fp16-pair operands: a cheaper contraction for the same convolutions (nn.Conv2d / nn.ConvTranspose2d of
 * models/networks.py:14,38 and their autograd backward); engine precision "f16x2" --------------------------------------------------
 * Operands are fp16 pairs (x = hi + lo, format 1 below).  `products` = 3: al*bh + ah*bl + ah*bh with v_mfma_f32_32x32x16_f16
 * (up to 22 significant bits per operand, ~1e-6 relative: the forward layers, so OUTPUTS keep the bf16x3 tolerance);
 * `products` = 2: (ah + al)*bh, two MFMAs instead of three -- the operand that keeps only its hi plane (11 significant bits,
 * 2^-12 rms rounding) is the weight in the gather / scatter families and `big` in the weight gradient: DECLARED TOLERANCE ~2e-4
 * relative per layer, used for the backward layers (gradients).  fp16's range is narrow (|x| <= 65504, 2^-24 absolute step below
 * 2^-14), so a producer of GRADIENT planes multiplies by a power of two (`scale`) and the consuming launch multiplies its
 * accumulators by out_scale = 1/scale; activations and weights use scale 1 (values saturate at +-65504 instead of overflowing).
 * Split formats: 0 = bf16 pair (the *_bf16x3 entry points), 1 = fp16 pair (the *_f16* entry points); same buffer sizes.
 * Every *_fmt producer with format 0 and scale 1 is bit-identical to the entry point without the suffix. */
#define VP_SPLIT_BF16 0
#define VP_SPLIT_F16 1
int vp_split_fmt_f32(const float* x, void* out_split, size_t n, int fmt, float scale, vp_stream stream);
int vp_nchw_to_nhwc_split_fmt_f32(const float* in, float* out, void* out_split, int B, int C, int H, int W, int fmt, vp_stream stream);
int vp_bn_act_fwd_split_fmt_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                float* y, void* y_split, int R, int C, int act, float slope, int fmt, vp_stream stream);
/* dx (fp32, unscaled) and/or dx_split = split(scale * dx) */
int vp_bn_act_bwd_split_fmt_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta,
                                int R, int C, int act, float slope, int batch_stats, int fmt, float scale,
                                void* ws, size_t ws_bytes, vp_stream stream);
/* the same with a sticky saturation flag: *saturated (device int, owned and zeroed by the caller) is set to 1 when fmt = fp16 pairs and
 * |scale * dx| exceeds fp16's range (65504) somewhere -- the planes then hold clamped values; engine.FusedVAEStep(precision="f16x2")
 * reads the flag in sync_counters() and asks for a smaller grad_scale16 */
int vp_bn_act_bwd_split_fmt_sat_f32(const float* x, const float* dy, const float* mean, const float* rstd,
                                    const float* gamma, const float* beta, float* dx, void* dx_split, float* dgamma, float* dbeta,
                                    int R, int C, int act, float slope, int batch_stats, int fmt, float scale, int* saturated,
                                    void* ws, size_t ws_bytes, vp_stream stream);
int vp_im2col5s2_split_fmt_f32(const float* x, void* out_split, int B, int C, int Hb, int Wb, int nchw, int fmt, vp_stream stream);
int vp_pack_w_im2col5_split_fmt(const float* w_ref, void* out_split, int Cout, int C, int fmt, vp_stream stream);
/* the three families; arguments as the *_bf16x3 entry points plus products (2 | 3) and out_scale (workspace queries of the weight
 * gradient: the *_bf16x3 ones).  The weight gradient exists in the two-product form only. */
int vp_conv5_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out,
                        int B, int Hs, int Ws, int Cbig, int Csmall, int stride, int act, int products, float out_scale, vp_stream stream);
int vp_conv_gather_f16(const void* big_split, const void* w_p0_split, const float* bias, float* small_out, int B, int Hs, int Ws,
                       int Hb, int Wb, int Cbig, int Csmall, int ks, int stride, int act, int products, float out_scale, vp_stream stream);
int vp_conv5_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out,
                         int B, int Hs, int Ws, int Csmall, int Cbig, int stride, int products, float out_scale, vp_stream stream);
int vp_conv_scatter_f16(const void* small_split, const void* w_p1_split, float* big_out, int B, int Hs, int Ws, int Hb, int Wb,
                        int Csmall, int Cbig, int ks, int stride, int products, float out_scale, vp_stream stream);
int vp_conv5_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                         int Csmall, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream);
int vp_conv5_wgrad_f16x2_cus(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Cbig,
                             int Csmall, int stride, float out_scale, int max_cus, void* ws, size_t ws_bytes, vp_stream stream);
int vp_conv_wgrad_f16x2(const void* big_split, const void* small_split, float* dw_ref, int B, int Hs, int Ws, int Hb, int Wb, int Cbig,
                        int Csmall, int ks, int stride, float out_scale, void* ws, size_t ws_bytes, vp_stream stream);
/* forward layers with the BatchNorm statistics in the epilogue (as vp_conv5_*_stats_bf16x3; own workspace query because the
 * launch shapes differ where bf16x3 uses its halo / pipelined kernels) */
size_t vp_conv5_stats_f16_workspace_bytes(int family, int B, int Hs, int Ws, int Cbig, int Csmall, int stride);
int vp_conv5_gather_stats_f16(const void* big_split, const void* w_p0_split, float* small_out,
                              int B, int Hs, int Ws, int Cbig, int Csmall, int stride, int products, float eps, float momentum,
                              float* mean, float* rstd, float* running_mean, float* running_var,
                              void* ws, size_t ws_bytes, vp_stream stream);
int vp_conv5_scatter_stats_f16(const void* small_split, const void* w_p1_split, float* big_out,
                               int B, int Hs, int Ws, int Csmall, int Cbig, int stride, int products, float eps, float momentum,
                               float* mean, float* rstd, float* running_mean, float* running_var,
                               void* ws, size_t ws_bytes, vp_stream stream);


/* ---- optimiser step on a flat arena (train_BE.py:62-64,131; train.py:136-140) --------------- */
/* torch.optim.Adam semantics (no amsgrad, no weight decay); g is multiplied by grad_scale first
 * (1/world_size after a sum all-reduce). step is the 1-based step count. */
int vp_adam_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                float eps, int step, float grad_scale, vp_stream stream);
/* The same update for a weight matrix p[R][Cn] whose gradient is a sum of K outer products, g = grad_scale * A^T B (A [K][R] =
 * the output gradients, Bm [K][Cn] = the inputs of a Linear layer, models/networks.py:22,79): the gradient is contracted inside the
 * update and never written.  Replaces the weight-gradient GEMM + vp_adam_f32 on that slice (encoder.fc.0: 134 MB not written
 * and not re-read per step).  R, Cn multiples of 4. */
int vp_adam_outer_f32(float* p, float* m, float* v, const float* A, const float* Bm, int K, int R, int Cn, float lr, float beta1,
                      float beta2, float eps, int step, float grad_scale, vp_stream stream);
/* torch.optim.RMSprop semantics (alpha, eps; no momentum, not centered) */
int vp_rmsprop_f32(float* p, const float* g, float* sq, size_t n, float lr, float alpha, float eps,
                   float grad_scale, vp_stream stream);

#ifdef __cplusplus
}
#endif
#endif
