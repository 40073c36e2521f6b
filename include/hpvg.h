/* hpvg.h - C ABI of libhpvg.so, the MI355X (gfx950) kernels behind the HP-VAE-GAN train step.
 *
 * Boundary contract: the reference (lior1990/hp-vae-gan, /root/reference) has no
 * FFI of its own - the hot path sits behind a Python module surface (modules/networks_3d.py,
 * networks_2d.py, losses.py, utils.py, utils/images.py) whose arithmetic is torch ATen ops.  Each
 * entry point below replaces one such ATen op (or a fused group) at the call site cited.  The Python
 * mirror of that module surface (package hp-vae-gan_amd/) binds these with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Rules for every entry point:
 *   - plain pointers and sizes only; all tensors fp32, contiguous, NCDHW (2-D convs: T = 1, KT = 1);
 *   - pointers are DEVICE pointers unless stated; the caller owns every buffer, including workspace;
 *   - never allocates, never synchronises, never throws; work is enqueued on `stream` (a hipStream_t);
 *   - returns 0 (HPVG_OK) or a negative error code.
 */
#ifndef HPVG_H
#define HPVG_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define HPVG_OK 0
#define HPVG_ERR_ARG (-1)
#define HPVG_ERR_WORKSPACE (-2)
#define HPVG_ERR_UNSUPPORTED (-3)
#define HPVG_ERR_LAUNCH (-4)

/* ---- convolution: nn.Conv3d/nn.Conv2d k=3 s=1 p=1 (modules/networks_3d.py:51,63,175,341,362; networks_2d.py:56,68,181,202,223)
 * KT = 3: 3x3x3 taps, KT = 1: 3x3 taps.  Weights are first packed into MFMA fragment order. */
size_t hpvg_conv_wpack_floats(int Cin, int Cout, int KT);
/* w: layer weight [Cout_layer][Cin_layer][KT][3][3]; inv_scale: device scalar 1/sigma (spectral norm) or NULL.
 * transpose_flip=0 -> pack for forward; 1 -> pack for backward-data (a conv with Cin=Cout_layer, Cout=Cin_layer). */
int hpvg_conv_pack_weight_f32(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                              int transpose_flip, void* stream);
/* y = conv(f(x), wp) + bias; f = identity or LeakyReLU_opt(in_scale[c]*x + in_shift[c]) applied before zero padding
 * (fused BatchNorm-apply of the producing ConvBlock3D, networks_3d.py:54-55); out_lrelu: LeakyReLU(0.2) epilogue
 * (ConvBlock3DSN, networks_3d.py:59-70). bias may be NULL. */
/* n <= HPVG_PACK_BATCH_MAX weights of one square layer shape (C -> C, C > 4) packed in one launch; flip[i] != 0: the
 * backward-data pack of item i.  w / wp / flip: host arrays. */
#define HPVG_PACK_BATCH_MAX 16
int hpvg_conv_pack_weight_batch_f32(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, void* stream);
/* Geometry-aware packs.  A pack made by the functions above serves every launch of the layer; the _for forms leave out the
 * two-axis Winograd fragments (the largest section) unless a launch of the given geometry - kernel view: B, T, H, W of the
 * conv's input - will read them (hpvg_conv_wants_wino2d), and are valid for launches of that geometry only. */
size_t hpvg_conv_wpack_floats_for(int Cin, int Cout, int KT, int B, int T, int H, int W);
int hpvg_conv_wants_wino2d(int B, int Cin, int Cout, int T, int H, int W, int KT); /* host only: 1 / 0 */
int hpvg_conv_pack_weight_for_f32(const float* w, const float* inv_scale, float* wp, int Cin_layer, int Cout_layer, int KT,
                                  int transpose_flip, int B, int T, int H, int W, void* stream);
int hpvg_conv_pack_weight_batch_for_f32(int n, const float* const* w, float* const* wp, const int* flip, int C, int KT, int B, int T,
                                        int H, int W, void* stream);
/* ws (optional, may be NULL): scratch of hpvg_conv_fwd_ws_bytes() bytes for the stream-K schedule (512 persistent
 * workgroups share the (tile, channel-chunk) items evenly; tiles cut across workgroups pass through ws as partial sums
 * and are finished in fixed order).  Without it the same kernel runs one workgroup per tile: slower on grids of 1-5
 * tiles per CU slot, results equal up to fp32 summation order. */
size_t hpvg_conv_fwd_ws_bytes(int B, int Cin, int Cout, int T, int H, int W, int KT);
/* out_mask (optional, shaped like y): y *= (out_mask > 0 ? 1 : 0.2) in the epilogue - leaky_relu_backward
 * (networks_3d.py:21) of the activation BELOW, fused into the backward-data conv of the layer that consumed it (the
 * conv's output is the gradient w.r.t. that activation; out_mask = the activated tensor, i.e. this layer's saved input). */
int hpvg_conv_fwd_f32(const float* x, const float* wp, const float* bias, const float* in_scale, const float* in_shift,
                      int in_lrelu, float* y, int out_lrelu, const float* out_mask, void* ws, size_t ws_bytes, int B, int Cin,
                      int Cout, int T, int H, int W, int KT, void* stream);
/* The same conv with the LeakyReLU sign mask in 1-BIT form ([B][ceil(C/32)][T*H*W] words, bit c%32 of word c/32 = activation
 * of channel c > 0; hpvg_conv_mask_words() of them): `bits_out` (nullable) is written by an out_lrelu epilogue for the
 * backward-data conv of the layer that consumes the activation, which passes it as `mask_bits` (nullable; instead of the
 * fp32 `out_mask` of hpvg_conv_fwd_f32: 8 mask loads per tile and lane instead of 128).  Cout > 4 only. */
size_t hpvg_conv_mask_words(int B, int C, int T, int H, int W);
int hpvg_conv_fwd_bits_f32(const float* x, const float* wp, const float* bias, float* y, int out_lrelu, const unsigned* mask_bits,
                           unsigned* bits_out, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT,
                           void* stream);
int hpvg_conv_fwd_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out10); /* host only: tile plan */
/* host only: the kernel family a plain launch of this shape runs on: 0 direct implicit GEMM, 1 Winograd F(2,3) along W (2/3 of
 * the direct matrix-core work), 2 Winograd F(2x2,3x3) (4/9), 3 narrow-output kernel (Cout <= 4).  bench.py prices the
 * roofline's executed flops with it. */
int hpvg_conv_fwd_kernel_kind(int B, int Cin, int Cout, int T, int H, int W, int KT);
/* Wide layers (kernel view Cin >= 8, Cout > 32) have a second kernel behind the same entry points: Winograd F(2,3) along W
 * (four products per output pair and (dt, dh) instead of six, summed over the channels before the output transform: 2/3 of
 * the matrix-core work; fp32, ~1e-6 of the output scale away from the direct kernel).  The weight pack carries both forms;
 * which kernel a launch runs is decided per shape.  hpvg_conv_wino_config: mode 0 = direct kernel only, 1 = Winograd from
 * `min_positions` output positions (B*T*H*W) up, 2 = every eligible launch (3 / 4: the same with the kernel's staging form
 * forced - rows as they lie in memory with 16-byte LDS-DMA pieces / halo'd bands with dword pieces; otherwise by size;
 * 5: every eligible launch, the TWO-axis kernel F(2x2,3x3) - 4/9 of the direct matrix-core work, 3x3x3 only - wherever it
 * can run, 6: the one-axis kernel only; by default the two-axis kernel takes the largest launches); a
 * negative argument leaves that setting as it is; returns the mode in force (HPVG_ERR_UNSUPPORTED when the process was
 * started with HPVG_WINO=0).  Host only. */
int hpvg_conv_wino_config(int mode, long min_positions);
/* host only: the Winograd kernel's tile plan: out[0..9] = L, Tw, nrange, ntw, RS, pair blocks per wave, m-tiles per
 * workgroup, gridy, lds_bytes, ntiles */
int hpvg_conv_wino_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out10);
/* host only: tile plan of the narrow-output kernel (Cout <= 4): out[0..6] = RS, Th, nth, nb, npos, G, pitch, then nb triples
 * (window start, first output column, output columns); out needs 7 + 3*16 ints */
int hpvg_conv_narrow_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out55);
/* dW (natural layout) of the conv above: aten::convolution_backward weight half, reached from
 * total_loss.backward() / errD_total.backward() (train_video.py:182,200). accumulate!=0: dw += result. */
size_t hpvg_conv_bwd_weight_ws_bytes(int B, int Cin, int Cout, int T, int H, int W, int KT);
int hpvg_conv_bwd_weight_f32(const float* dy, const float* x, const float* in_scale, const float* in_shift, int in_lrelu,
                             float* dw, int accumulate, void* ws, size_t ws_bytes, int B, int Cin, int Cout, int T, int H,
                             int W, int KT, void* stream);
int hpvg_conv_bwd_weight_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out10); /* host only */
/* host only: the kernel family of the weight gradient at this shape: 0 / 1 direct (a workgroup per time tap / all taps per
 * workgroup), 2 Winograd along W (2/3 of the direct matrix-core work), 3 Winograd over H and W (4/9), 4 narrow kernels */
int hpvg_conv_bwd_weight_kernel_kind(int B, int Cin, int Cout, int T, int H, int W, int KT);
/* Wide layers (Cin > 4 and Cout > 4) have a Winograd weight-gradient kernel behind the same entry point (the transpose of
 * the forward F(2,3) along W: four products per pair of output columns and (dt, dh) instead of six, summed over all
 * positions before the output transform: 2/3 of the matrix-core work, fp32).  mode 0 = never (the direct kernels), 1 = the
 * default: every wide layer on a Winograd kernel - the one-axis kernel wins at every size - and the two-axis kernel (mode 5)
 * where its size rule picks it, 2 = every wide layer on the one-axis kernel,
 * 3 = every wide layer without the 16-byte staging form (widths that are multiples of 4 otherwise get it), 4 = every
 * wide layer with the 16-byte form on four waves instead of eight; 5 = every wide layer, the TWO-axis kernel (the transpose
 * of F(2x2,3x3) over H and W: 16 products per 2 x 2 output positions and dt instead of 36, 4/9 of the direct matrix-core work)
 * (any width; by default it takes the launches whose workgroups walk enough tiles), 6 = every wide layer,
 * the one-axis kernel only (2, 3, 4 also keep to the one-axis kernel); negative = query.  Returns the mode in force.  Host only. */
int hpvg_conv_bwd_weight_wino_config(int mode);
/* The weight gradient and the conv's bias gradient db[o] (+)= sum_{b,positions} dy[b][o] from ONE launch, for the layers
 * hpvg_conv_bwd_weight_fuses_bias() reports 1 for (the Winograd weight-gradient kernel runs them: its centre-tap workgroups
 * hold every dy pair in registers anyway); HPVG_ERR_UNSUPPORTED elsewhere (use hpvg_channel_sum_f32).  Same workspace as
 * hpvg_conv_bwd_weight_f32.  Reference: the bias half of aten::convolution_backward, train_video.py:182,200. */
int hpvg_conv_bwd_weight_fuses_bias(int B, int Cin, int Cout, int T, int H, int W, int KT);
int hpvg_conv_bwd_weight_bias_f32(const float* dy, const float* x, float* dw, int accumulate, float* db, int accumulate_db, void* ws,
                                  size_t ws_bytes, int B, int Cin, int Cout, int T, int H, int W, int KT, void* stream);
int hpvg_conv_bwd_weight_wino_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out10); /* host only */
/* host only: tile plan of the two-axis Winograd weight-gradient kernel: out[0..9] as above, out[10] = 1 when the shape runs it by
 * default (size rule) */
int hpvg_conv_bwd_weight_wino2_plan(int B, int Cin, int Cout, int T, int H, int W, int KT, int* out11);
/* out[c] = sum_{b,s} x[b][c][s]: conv bias gradient.  accumulate != 0: out[c] += (the caller passes the parameter's
 * gradient buffer, which removes autograd's AccumulateGrad add kernel) */
size_t hpvg_channel_sum_ws_bytes(int C);
int hpvg_channel_sum_f32(const float* x, float* out, int accumulate, void* ws, size_t ws_bytes, int B, int C, long S,
                         void* stream);

/* ---- BatchNorm3d/2d, train mode on every forward (networks_3d.py:54), eps 1e-5, momentum 0.1 */
size_t hpvg_bn_ws_bytes(int C);
int hpvg_bn_train_stats_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, void* ws,
                            size_t ws_bytes, int B, int C, long S, void* stream);
/* the two above in two launches instead of three (the apply kernel finalizes its own channel's statistics) */
int hpvg_bn_train_fwd_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale, float* shift, float* y, int lrelu,
                          int groups, void* ws, size_t ws_bytes, int B, int C, long S, void* stream);
/* groups > 1: the batch holds `groups` independent passes (B / groups samples each, back to back), each normalised with its
 * own statistics; running statistics updated once per group, in order; statistics arrays laid out [groups][4][C] (pass the
 * addresses of group 0's mean / invstd / scale / shift); workspace hpvg_bn_ws_bytes(C * groups) */
/* y = LeakyReLU_opt(scale[c]*x + shift[c]) (BN apply + nn.LeakyReLU(0.2), networks_3d.py:21,54-56) */
int hpvg_affine_act_f32(const float* x, const float* scale, const float* shift, float* y, int lrelu, int B, int C, long S,
                        void* stream);
/* backward of h = LeakyReLU_opt(BN_train(r)): dr, dgamma, dbeta (native_batch_norm_backward + leaky_relu_backward);
 * accumulate != 0: dgamma / dbeta += */
int hpvg_bn_act_bwd_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                        const float* shift, int lrelu, int groups, float* dr, float* dgamma, float* dbeta, int accumulate, void* ws,
                        size_t ws_bytes, int B, int C, long S, void* stream);
/* SECOND-order backward of the same block: the WGAN-GP of a critic that contains BatchNorm (WDiscriminatorBaselines,
 * networks_3d.py:184-210; modules/utils.py:14-18) differentiates the first-order backward once more.  g = dL/d(dr);
 * outputs (each nullable): g_dh = dL/d(dh), g_r = dL/d(r), g_gamma = dL/d(gamma) (+= when accumulate).  torch:
 * batchnorm_double_backward + leaky_relu_backward (LeakyReLU'' = 0). */
size_t hpvg_bn_bwd2_ws_bytes(int C);
int hpvg_bn_act_bwd2_f32(const float* dh, const float* g, const float* r, const float* mean, const float* invstd, const float* scale,
                         const float* shift, int lrelu, float* g_dh, float* g_r, float* g_gamma, int accumulate, void* ws,
                         size_t ws_bytes, int B, int C, long S, void* stream);
/* BatchNorm with the batch split over ranks (one process per GPU): the rank-local per-channel sums are double pairs the
 * caller all-reduces (RCCL) between the two halves; statistics, running buffers and dr then follow from the global sums.
 * Same arithmetic as the single-GPU entry points above (which are sums + finalize / sums + apply in one call). */
int hpvg_bn_sums_f32(const float* x, double* sums, void* ws, size_t ws_bytes, int B, int C, long S, void* stream);
int hpvg_bn_finalize_f32(const double* sums, double count, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                         int C, void* stream);
int hpvg_bn_act_bwd_sums_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                             const float* shift, int lrelu, double* sums, void* ws, size_t ws_bytes, int B, int C, long S,
                             void* stream);
int hpvg_bn_act_bwd_apply_f32(const float* dh, const float* r, const float* mean, const float* invstd, const float* scale,
                              const float* shift, int lrelu, const float* sums, float inv_count, float* dr, int B, int C, long S,
                              void* stream);
/* out = dy * (h > 0 ? 1 : 0.2): leaky_relu_backward on the in-place activated tensor (networks_3d.py:21) */
int hpvg_lrelu_mask_mul_f32(const float* dy, const float* h, float* out, long n, void* stream);

/* ---- generator glue: tanh / residual (networks_3d.py:377,404), reparameterize (networks_3d.py:29-33) */
/* out = a + b: residual add of the SinGAN baseline generator (networks_3d.py:319) */
int hpvg_add_f32(const float* a, const float* b, float* out, long n, void* stream);
/* dst = src, or dst = 0 when src is NULL - as a kernel (concatenations / zero fills of an iteration that may be captured
 * into a hipGraph must not become memcpy / memset nodes) */
int hpvg_copy_f32(const float* src, float* dst, long n, void* stream);
int hpvg_tanh_fwd_f32(const float* x, const float* res /*nullable*/, float* y, long n, void* stream);
int hpvg_tanh_bwd_f32(const float* dy, const float* y, float* dx, long n, void* stream);
int hpvg_reparam_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, long n, void* stream);
int hpvg_reparam_bwd_f32(const float* dz, const float* logvar, const float* eps, float* dlogvar, long n, void* stream);

/* ---- losses: kl_criterion (modules/losses.py:7-9), nn.MSELoss (train_video.py:355), WGAN means (train_video.py:170,178,194) */
size_t hpvg_reduce_ws_bytes(void);
int hpvg_kl_fwd_f32(const float* mu, const float* logvar, float* out, void* ws, size_t ws_bytes, long n, void* stream);
int hpvg_kl_bwd_f32(const float* gout, const float* mu, const float* logvar, float* dmu, float* dlogvar, long n, void* stream);
int hpvg_mse_fwd_f32(const float* a, const float* b, float* out, void* ws, size_t ws_bytes, long n, void* stream);
int hpvg_mse_bwd_f32(const float* gout, const float* a, const float* b, float* da, long n, void* stream);
int hpvg_sum_scaled_f32(const float* x, float* out, double scale, void* ws, size_t ws_bytes, long n, void* stream);
int hpvg_sqsum_f32(const float* x, float* out, void* ws, size_t ws_bytes, long n, void* stream);
int hpvg_fill_scaled_f32(const float* gout, float coef, float* out, long n, void* stream);

/* ---- WGAN-GP (modules/utils.py:4-19): interpolate, per-voxel channel norm penalty and its gradient */
int hpvg_lerp_f32(const float* a, const float* b, const float* alpha, float* out, long n, void* stream);
int hpvg_gp_fwd_f32(const float* g, float* out, float lambda, void* ws, size_t ws_bytes, int B, int C, long S, void* stream);
int hpvg_gp_bwd_f32(const float* gout, const float* g, float* dg, float lambda, int B, int C, long S, void* stream);

/* ---- pyramid resize: F.interpolate(mode=tri/bilinear, align_corners=True) (utils/images.py:13,17,24,83-105),
 * optional fused noise injection yn = y + amp*noise (networks_3d.py:399-400).  BC = batch*channels. */
int hpvg_upsample_linear_ac_f32(const float* x, float* y, const float* noise, float amp, float* yn, long BC, int Ti, int Hi,
                                int Wi, int To, int Ho, int Wo, void* stream);
/* ---- N(0,1) noise (utils/images.py:49 zeros(..).normal_(0,1); networks_3d.py:32 reparameterisation eps): counter-based
 * Philox4x32-10 + Box-Muller, stream = (seed, call_id, iter[0]); iter: device int bumped once per train iteration
 * (hpvg_counter_inc_i32), so a replayed hipGraph draws fresh noise with unchanged launch arguments; NULL = 0. */
int hpvg_normal_f32(float* out, long n, unsigned long long seed, unsigned call_id, const int* iter, void* stream);
int hpvg_uniform_f32(float* out, long n, unsigned long long seed, unsigned call_id, const int* iter, void* stream); /* U[0,1) */
/* resize + level noise generated in the kernel (networks_3d.py:395-400): y = resize(x), yn = y + amp * N(0,1) where the noise
 * is exactly what hpvg_normal_f32 would write for the same stream; samples b < first_noisy get yn = y (C channels per sample). */
int hpvg_upsample_linear_ac_noise_f32(const float* x, float* y, float* yn, float amp, long BC, int C, int first_noisy, int Ti, int Hi,
                                      int Wi, int To, int Ho, int Wo, unsigned long long seed, unsigned call_id, const int* iter,
                                      void* stream);
/* backward: dx = resize^T(dy + dy2); dy2 (nullable) is the gradient of the noisy output yn.  A gather over the input
 * voxels in a fixed order: no float atomics, bitwise reproducible, dx need not be zeroed. */
int hpvg_upsample_linear_ac_bwd_f32(const float* dy, const float* dy2, float* dx, long BC, int Ti, int Hi, int Wi, int To, int Ho,
                                    int Wo, void* stream);

/* ---- data front-end (the step before the path): frames [N][H][W][3] uint8 RGB -> the stage's clip tensor
 * [3][count][h][w] fp32: cv2.resize(INTER_LINEAR) geometry per frame (datasets/generate_frames.py:44-46), temporal window
 * frame[first + k*step] (datasets/video.py:56-57), /255, K.hflip, K.normalize(0.5, 0.5), permute C,T,H,W (video.py:58-82).
 * quantize != 0 rounds the resized value to a uint8 level first, as cv2.resize on uint8 does. */
int hpvg_frames_resize_norm_u8_f32(const unsigned char* src, float* dst, int N, int H, int W, int first, int step, int count, int h,
                                   int w, int hflip, int quantize, void* stream);

/* ---- spectral norm (nn.utils.spectral_norm, networks_3d.py:63): one power iteration, sigma, 1/sigma; backward through sigma */
int hpvg_sn_power_iter_f32(const float* w, float* u, float* v, float* sigma, float* inv_sigma, float* uv_copy, int Co, int K,
                           int do_iter, float eps, void* ws, size_t ws_bytes, void* stream);
/* the same for n <= HPVG_SN_BATCH_MAX independent layers in one launch (one workgroup per layer, e.g. the six SN convs of
 * WDiscriminator3D at the start of a forward), also writing w_eff[i] = w[i] / sigma_i.  The pointer / size arrays live on
 * the host; sig[i] -> 2 floats (sigma, 1/sigma); uv_copy may be NULL or hold NULLs; ws: sum(Co) floats. */
#define HPVG_SN_BATCH_MAX 8
int hpvg_sn_power_iter_batch_f32(int n, const float* const* w, float* const* u, float* const* v, float* const* sig,
                                 float* const* uv_copy, float* const* w_eff, const int* Co, const int* K, int do_iter, float eps,
                                 void* ws, size_t ws_bytes, void* stream);
/* out = x / s[0]: weight = weight_orig / sigma (torch SpectralNorm.compute_weight) */
int hpvg_div_scalar_f32(const float* x, const float* s, float* out, long n, void* stream);
/* dworig (+)= dweff/sigma - (sum(dweff .* worig)/sigma^2) u v^T  (backward of weight_orig -> weight) */
size_t hpvg_sn_bwd_ws_bytes(int Co, int K);
/* n layers in two launches (host arrays as above; uv[i] = the forward's (u, v) copy; ws: sum of hpvg_sn_bwd_ws_bytes) */
int hpvg_sn_bwd_batch_f32(int n, const float* const* dweff, const float* const* worig, const float* const* uv,
                          const float* const* sigma, float* const* dworig, const int* accumulate, const int* Co, const int* K,
                          void* ws, size_t ws_bytes, void* stream);
int hpvg_sn_bwd_f32(const float* dweff, const float* worig, const float* u, const float* v, const float* sigma, float* dworig,
                    int accumulate, void* ws, size_t ws_bytes, int Co, int K, void* stream);

/* ---- optimizer: clip_grad_norm_ (train_video.py:201) and optim.Adam (train_video.py:55,88) over flat arenas */
int hpvg_clip_scale_f32(float* g, long n, const float* sqsum, float max_norm, float* coef_out /*nullable, 2 floats*/,
                        void* stream);
/* step: 1-based count from the host, or step_dev (device int, non-NULL) for hipGraph replays */
int hpvg_adam_step_f32(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                       int step, const int* step_dev, void* stream);
int hpvg_counter_inc_i32(int* counter, void* stream);

/* ---- variant models (no trainer of the reference drives them; module-surface completeness): Encode3DVAE_nb / Encode2DVAE_nb
 * (networks_3d.py:110-138, networks_2d.py:115-143), GeneratorVAE_nb (networks_3d.py:409-485), reparameterize_bern
 * (networks_3d.py:38-45), kl_bern_criterion (losses.py:12-14).  Tensors [B][C][S], S = T*H*W. */
/* bern[b][s] = sigmoid(logit[b][s]); out = bern * f (broadcast over channels); backward: df, dlogit from dout and the gradient
 * arriving at bern itself (either may be NULL) */
int hpvg_gate_fwd_f32(const float* f, const float* logit, float* out, float* bern, int B, int C, long S, void* stream);
int hpvg_gate_bwd_f32(const float* dout, const float* f, const float* bern, const float* dbern_ext, float* df, float* dlogit, int B,
                      int C, long S, void* stream);
/* out[b][c] = scale * sum_s x[b][c][s] * (w ? w[b][s] : 1): nn.AdaptiveAvgPool3d(1) with scale = 1/S, and d(outer)/d(code) */
int hpvg_rowsum_f32(const float* x, const float* w, float* out, float scale, int B, int C, long S, void* stream);
/* out[b][c][s] = g[b][c] * (w ? w[b][s] : scale): code x map (z_vae_norm * z_vae_bern) and the average pool's backward */
int hpvg_outer_f32(const float* g, const float* w, float* out, float scale, int B, int C, long S, void* stream);
/* out[b][s] = sum_c a[b][c][s] * v[b][c]: d(outer)/d(map) */
int hpvg_colsum_f32(const float* a, const float* v, float* out, int B, int C, long S, void* stream);
int hpvg_reparam_bern_fwd_f32(const float* x, const float* eps, float* z, long n, void* stream);
int hpvg_reparam_bern_bwd_f32(const float* dz, const float* x, float* dx, long n, void* stream);
int hpvg_kl_bern_fwd_f32(const float* x, float* out, void* ws, size_t ws_bytes, long n, void* stream);
int hpvg_kl_bern_bwd_f32(const float* gout, const float* x, float* dx, long n, void* stream);

/* ---- hipGraph replay of the iteration (host only): node census of a captured graph, counts[t] per hipGraphNodeType t
 * (0 kernel, 1 memcpy, 2 memset, ...; child graphs included).  The trainers refuse a captured iteration that holds
 * memcpy / memset nodes: those are not reliably ordered against kernel nodes on this runtime. */
int hpvg_graph_node_census(void* graph, int* counts, int ntypes);

#ifdef __cplusplus
}
#endif
#endif /* HPVG_H */
