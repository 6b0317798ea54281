/*
 * hesic_hip.h -- C ABI of libhesic_hip.so: the MI355X (gfx950) kernels behind the HESIC / HESIC+
 * forward + backward hot path.
 *
 * The reference has no FFI for this path: its "plugin interface" is the Python operator surface
 * (compressai.layers / compressai.entropy_models / compressai.models.utils).  This
 * header is the boundary a maintainer binds with ctypes (INTEGRATION.md shows the stub); every entry
 * point names the reference operator it replaces.  Citations are path:line inside the reference repo.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; sizes are element counts
 *  - feature maps are NHWC ("channels_last"): element (b,y,x,c) at ((b*H+y)*W+x)*pix_stride + c_off + c
 *  - 3-channel images use explicit element strides (sb, sc, sy, sx) so NCHW planar inputs need no copy
 *  - dtype: HESIC_F32 or HESIC_H16 storage; all arithmetic accumulates in fp32.  HESIC_H16 is the 16-bit format the library
 *    was BUILT for (hesic_h16_format()): libhesic_hip.so = bfloat16 (training + inference), libhesic_hip_f16.so = IEEE binary16
 *    (inference: same matrix-core rate on gfx950, 11-bit significand; activations saturate at +-65504).  Same entry points, same
 *    sources; a caller binds the library whose format its tensors have.  "bf16" in the comments below reads "the 16-bit format".
 *  - `stream` is a hipStream_t (0 = the null stream); calls are asynchronous on it
 *  - return value: 0 on success, otherwise a hipError_t (> 0) or HESIC_EINVAL (-1) for bad arguments;
 *    hesic_last_error() returns a static description of the last failure on the calling thread
 */
#ifndef HESIC_HIP_H
#define HESIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HESIC_ABI_VERSION 2   /* round 5: bumped (entry points added since v1: the tape calls, *_hilo_out, msssim, joint decode) */
#define HESIC_EINVAL (-1)

enum { HESIC_F32 = 0, HESIC_H16 = 1, HESIC_BF16 = 1 /* historical name of HESIC_H16 */ };
enum { HESIC_H16_BFLOAT16 = 0, HESIC_H16_FLOAT16 = 1 };
enum { HESIC_ACT_NONE = 0, HESIC_ACT_RELU = 1, HESIC_ACT_LEAKY = 2 /* slope 0.01 */ };

int hesic_abi_version(void);
/* HESIC_H16_BFLOAT16 or HESIC_H16_FLOAT16: what HESIC_H16 storage means in this build of the library */
int hesic_h16_format(void);
const char* hesic_last_error(void);

/* ------------------------------------------------------------------ convolution (rows A1,A2,A4-A7,A11)
 * Replaces nn.Conv2d / nn.ConvTranspose2d as built by conv()/deconv()
 * (compressai/models/utils.py:104-118) and MaskedConv2d (compressai/layers/layers.py:21-45).       */
typedef struct {
    int32_t B, H, W, Cin;           /* input  (NHWC)                                               */
    int32_t Ho, Wo, Cout;           /* output (NHWC); transposed: Ho = H*stride, Wo = W*stride       */
    int32_t KH, KW, stride, pad;    /* pad = k/2; transposed => output_padding = stride-1            */
    int32_t transposed;             /* 0: Conv2d   1: ConvTranspose2d                                */
    int32_t dtype;                  /* storage type of x, packed weights and y                       */
    int32_t act;                    /* fused activation on the output (HESIC_ACT_*)                  */
    int32_t in_abs;                 /* 1: |x| on load (encode_hyper, newnet1.py:434)                 */
    int32_t x_pix_stride, x_c_off;  /* channels per pixel of the x buffer, first channel used        */
    int32_t y_pix_stride, y_c_off;  /* same for y: lets a conv write straight into a concat buffer   */
    int32_t tap_mask_lo;            /* bit t set => tap t (ky*KW+kx) is live; 0 = all (MaskedConv2d) */
} hesic_conv_desc;

/* Repack an fp32 PyTorch weight into the kernels' layout [KH*KW][Cout][Cin] of `dtype`.
 * transposed=0: w is (Cout,Cin,KH,KW)  (nn.Conv2d);  transposed=1: w is (Cin,Cout,KH,KW)
 * (nn.ConvTranspose2d).  flip=1 mirrors the taps (used for stride-1 transposed conv == conv with the
 * flipped kernel, and by the data-gradient kernels).  mask (may be NULL): (Cout,Cin,KH,KW) multiplier. */
int hesic_pack_conv_weight(const float* w, const float* mask, void* w_packed, int Cout, int Cin, int KH, int KW,
                           int transposed, int flip, int dtype, void* stream);

/* The same layout in HESIC_H16 for a Conv2d weight (Cout,Cin,KH,KW) of an INFERENCE layer that multiplies single operands, rounded with
 * error feedback over the taps of each (cout, cin) pair (serpentine walk over the window): the 16-bit weights of a pair sum to the fp32
 * weights' sum within half an ulp, so on spatially smooth feature maps the weight-rounding error of the layer's output cancels
 * ("x3c2").  A drop-in for hesic_pack_conv_weight(w, NULL, wp, ..., 0, 0, HESIC_H16): same buffer, same consumers.        */
int hesic_pack_conv_weight_shaped(const float* w, void* w_packed, int Cout, int Cin, int KH, int KW, void* stream);
/* Round 5: the same for a ConvTranspose2d weight (Cin, Cout, KH, KW) of the given stride: the error is fed back inside each output
 * phase's tap class ((ky % stride, kx % stride)), the taps one output pixel actually sums (the synthesis stacks, deconv() of
 * compressai/models/utils.py:112-118, at 16-bit inference).  Same [tap][Cout][Cin] layout as hesic_pack_conv_weight(transposed = 1). */
int hesic_pack_conv_weight_shaped_tr(const float* w, void* wp, int Cout, int Cin, int KH, int KW, int stride, void* stream);

/* Many repacks in one launch (a training step repacks every conv weight after the optimiser update): `jobs_device` is a
 * DEVICE array of n_jobs descriptors, job i owns blocks [block0_i, block0_{i+1}), one per tile of 8 couts x 32 cins
 * (block0 ascending, block0_0 = 0, total_blocks = sum of ceil(Cout/8) * ceil(Cin/32)); fields as hesic_pack_conv_weight,
 * KH*KW <= 25.                                                                                           */
typedef struct {
    const float* w; const float* mask; void* w_packed;
    int32_t Cout, Cin, KH, KW, transposed, flip, dtype, block0;
} hesic_pack_job;
int hesic_pack_conv_weights_batched(const hesic_pack_job* jobs_device, int n_jobs, int total_blocks, void* stream);

/* y = act(conv(x, w) + bias).  bias may be NULL.  w_packed from hesic_pack_conv_weight. */
int hesic_conv2d_forward(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                         void* y, void* stream);

/* Fused y = (I)GDN(conv(x)) (bf16 storage, Cout == 128, no activation): the conv output tile is normalised in the
 * epilogue of the implicit-GEMM kernel, saving the activation's HBM round trip between conv()/deconv() and GDN.forward
 * (newnet1.py:594-600, :617-623).  gamma_packed (2*128*128 bf16: an LDS-image copy for the image-side kernel, then an
 * MFMA-fragment-order copy for the implicit-GEMM kernel) / beta_packed (128 fp32) come from hesic_gdn_pack_params
 * (NonNegativeParametrizer applied).                                                                                   */
int hesic_gdn_pack_params(const float* beta, const float* gamma, float beta_min, void* gamma_packed, float* beta_packed,
                          int C, void* stream);
/* The same for every 128-channel GDN of a model in ONE launch: a table of jobs in device memory (the training step refreshes all its
 * packed GDN parameters behind the optimiser update, next to hesic_pack_conv_weights_batched).                                      */
typedef struct {
    const float* beta; const float* gamma;     /* reparametrised-domain parameters (128, 128 x 128) */
    void* gamma_packed; float* beta_packed;    /* as hesic_gdn_pack_params */
    float beta_min; int reserved;
} hesic_gdn_pack_job;
int hesic_gdn_pack_params_batched(const hesic_gdn_pack_job* jobs_device, int n_jobs, void* stream);
int hesic_conv2d_gdn_forward(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                             const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* stream);
/* Training form: additionally stores the conv output (conv + bias, bf16, same geometry as y) that GDN's backward needs
 * (hesic_gdn_backward takes it as x), so the forward of a training step keeps the fusion too.                          */
int hesic_conv2d_gdn_forward_train(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                   const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                   void* stream);

/* Same op with a caller-provided scratch buffer.  Low-resolution layers (the hyper path's 8x8 .. 32x32 maps,
 * newnet1.py:420-577) have too few pixels to fill the GPU with one block per output tile; given `ws` of at least
 * hesic_conv2d_ws_bytes(d) bytes the launcher cuts their K loop (taps x channels) into slices that run as separate
 * blocks, keeps the fp32 partial tiles in `ws` and finishes with a reduce + bias + activation pass.  ws_bytes(d) == 0
 * means the plain launch is used (ws may be NULL).  Results are deterministic (fixed slice order).                   */
size_t hesic_conv2d_ws_bytes(const hesic_conv_desc* d);
int hesic_conv2d_forward_ws(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                            void* y, void* ws, size_t ws_bytes, void* stream);

/* bf16 storage only: the same op that ALSO (y != NULL) or ONLY (y == NULL) stores act(conv + bias) as fp32, straight from
 * the fp32 accumulators, into channels [y32_c_off, y32_c_off + Cout) of an NHWC fp32 buffer with y32_pix_stride channels.
 * The reference computes the whole path in fp32 (newnet1.py:726-731): what feeds round() and the likelihoods -- the latents
 * y = g_a_conv4(.), z = encode_hyper(.) and the sigma / mu maps of gmm_hyper_y1/y2 or entropy_parameters -- must not pass
 * through 8-bit-mantissa storage (at |y| ~ 10 one bf16 ulp is 1/16: latents near .5 flip).  ws / ws_bytes as
 * hesic_conv2d_forward_ws (may be NULL / 0).                                                                        */
int hesic_conv2d_forward_f32out(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                void* y, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes,
                                void* stream);
/* Scratch bytes of hesic_conv2d_forward_f32out for `d` (0: no split; may differ from hesic_conv2d_ws_bytes: launches that feed round() /
 * a likelihood decide their K split per image so that a pair's latents never depend on the batch it is in). */
size_t hesic_conv2d_f32out_ws_bytes(const hesic_conv_desc* d);

/* Several convolutions of the same geometry in ONE launch (bf16 storage): the packed weight holds `groups` weights side by side along
 * Cout ([KH*KW][Cout][Cin], d->Cout = total, each slice written by hesic_pack_conv_weight_slice), group g's couts read input
 * channels [x_c_off + g * x_group_step, + d->Cin) of x (x_group_step = 0: all groups share one input).  Couts >= act_split take
 * activation act2 instead of d->act (0 = no split).  y and / or y_f32 as hesic_conv2d_forward_f32out.  This is how the three
 * branches of gmm_hyper_y1 / y2 (sigma: ReLU, means / weights: LeakyReLU; newnet1.py:456-577) run: first layers = one launch with
 * Cout = 3 * 128 on the shared input, the sigma / mean second layers = one launch of two groups, the two 128 -> 960 output layers
 * = one launch of two groups whose fp32 output holds sigma | means.  Couts per group and act_split must be multiples of 128.   */
int hesic_conv2d_forward_grouped(const hesic_conv_desc* d, int groups, int x_group_step, int act2, int act_split, const void* x,
                                 const void* w_packed, const float* bias, void* y, float* y_f32, int y32_pix_stride,
                                 int y32_c_off, void* stream);
int hesic_pack_conv_weight_slice(const float* w, void* w_packed_bf16, int Cout, int Cin, int KH, int KW, int transposed,
                                 int Cout_total, int co_off, void* stream);

/* ---- bf16x3 ("hi/lo") analysis path.  The reference runs g_a in fp32 (newnet1.py:580-601, :626-655) and round()s its output
 * (entropy_models.py:661-702): with single-bf16 operands ~1 % of the latents land on the other side of .5.  Here a value v is the
 * pair (hi = bf16(v), lo = bf16(v - hi)), stored [hi(C) | lo(C)] per pixel; the weights are packed as [w_hi | w_lo] along
 * Cin (twice the logical Cin: pack the concatenated fp32 weight with hesic_pack_conv_weight), a stage of the K loop brings a
 * channel chunk of all four halves into LDS and the matrix cores form x_hi w_hi + x_lo w_hi + x_hi w_lo in the fp32
 * accumulators (2^-17 relative per operand instead of 2^-9; Cin % 32 == 0).  d->Cin / d->Cout are
 * the LOGICAL channel counts, x_pix_stride >= 2 Cin.  With gamma_packed (hesic_gdn_pack_params) + gamma_lo_packed
 * (hesic_gdn_pack_params_lo) + beta_packed: fused hi/lo (I)GDN (Cout == 128), the output y_hilo is [hi(128) | lo(128)] at
 * y_c_off of a buffer with y_pix_stride >= 2 * 128 channels, y_f32 must be NULL.  Without: v = act(conv + bias) goes to y_f32 as
 * in hesic_conv2d_forward_f32out (the latent y = g_a_conv4(.), z = encode_hyper(.)) and / or to y_hilo as [hi(Cout) | lo(Cout)]
 * pairs -- of |v| when y_abs != 0: encode_hyper reads |y| (newnet1.py:434), and the magnitude is taken where v is still fp32.   */
int hesic_conv2d_forward_hilo(const hesic_conv_desc* d, const void* x_hilo, const void* w_packed_hilo, const float* bias,
                              const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                              void* y_hilo, int y_abs, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes,
                              void* stream);
/* Pair weights packed times 2^s (see above; any of the hi/lo conv entry points): the NEXT hi/lo launch of the calling thread multiplies its
 * accumulators by `scale` (= 2^-s) once, behind the K loop -- in front of bias, activation, the fused (I)GDN and the K-slice partials alike.  */
int hesic_conv2d_hilo_set_acc_scale(float scale);
/* Pairs x SINGLE weights (analysis mode "x3c2": g_a_conv3 / g_a_conv4): same arguments and layouts as hesic_conv2d_forward_hilo, but only
 * the first Cin values of every packed weight row are read -- the weights rounded to 16 bits with error feedback over the taps
 * (hesic_pack_conv_weight_shaped) -- and a pair costs TWO products (x_hi w + x_lo w); the fused (I)GDN stops at gamma'_hi (sq_hi + sq_lo).
 * CPU study / GPU: +5e-5 flipped latents per layer against the three-product form, 109 -> ~88 us on g_a_conv3 + GDN at B=8 512^2.       */
int hesic_conv2d_forward_hilo_w1(const hesic_conv_desc* d, const void* x_hilo, const void* w_packed_hilo, const float* bias,
                                 const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                 void* y_hilo, int y_abs, float* y_f32, int y32_pix_stride, int y32_c_off, void* ws, size_t ws_bytes,
                                 void* stream);
/* The bridge between a one-product layer and the hi/lo layers behind it (analysis mode "x3c2": g_a_conv2 -- 128 -> 128 5x5 s2 on the
 * 256^2 map of a 512^2 image, 70 % of g_a's MACs, newnet1.py:585-586 -- multiplies SINGLE 16-bit operands, everything else pairs):
 * x and w_packed are plain 16-bit (hesic_pack_conv_weight), v = conv + bias stays in the fp32 accumulators, the (I)GDN contraction
 * runs on pairs as above, y_hilo leaves as [hi(128) | lo(128)].  Conv2d only, Cout == 128.                                       */
int hesic_conv2d_gdn_forward_hilo_out(const hesic_conv_desc* d, const void* x, const void* w_packed, const float* bias,
                                      const void* gamma_packed, const void* gamma_lo_packed, const float* beta_packed, int inverse,
                                      void* y_hilo, void* stream);
/* split-K scratch for the plain (no GDN) hi/lo launch, as hesic_conv2d_ws_bytes (0: none needed; ws may be NULL) */
size_t hesic_conv2d_hilo_ws_bytes(const hesic_conv_desc* d);
/* lo half of gamma' (128*128 bf16, MFMA fragment order) for the hi/lo GDN epilogue; the hi half is hesic_gdn_pack_params'. */
int hesic_gdn_pack_params_lo(const float* gamma, void* gamma_lo_packed, int C, void* stream);
/* Column matrix of the image side as hi/lo bf16: cols[pixel][k], k = (ci*KH + ky)*KW + kx (zero for k >= C*KH*KW up to KP, KP % 8
 * == 0), written [hi(KP) | lo(KP)] per output pixel; x is fp32 with element strides (sb, sc, sy, sx).  g_a_conv1 (conv(3, N),
 * newnet1.py:583) then is a 1x1 hesic_conv2d_forward_hilo with Cin = KP on it.                                               */
int hesic_im2col_hilo(const float* x, const int64_t x_strides[4], int B, int C, int H, int W, int KH, int KW, int stride, int pad,
                      int Ho, int Wo, int KP, void* cols, void* stream);

/* Which kernel instantiation hesic_conv2d_forward picks for `d` (for profiling / roofline accounting):
 * out[0..3] = {pixel tile BM, cout tile BN, K step BK, 1 if the LDS-DMA (bf16) kernel else 0}.        */
int hesic_conv2d_variant(const hesic_conv_desc* d, int* bm_bn_bk_glds);

/* deconv() (compressai/models/utils.py:112-118; g_s of ywz/mywork/newnet1.py:603-624, and the data gradient of the stride-2 analysis
 * convs): a transposed stride-2 layer is four stride-1 output phases.  By default (mode 1) hesic_conv2d_forward / _gdn_forward[_train]
 * run the four phases of a 128-pixel x 128-cout tile in ONE block (one LDS-DMA pipeline across the phases, igemm_tr4_kernel) when
 * the launch is large enough to fill the chip that way; results are bit-identical to the one-block-per-phase form.  mode 0: never
 * (A/B measurements), 2: whenever the shape is eligible (tests).  Process-wide; returns the previous mode, -1 on a bad argument.
 * hesic_conv2d_variant reports the fused form as out[3] == 2.  Environment default: HESIC_IGEMM_PHASE4.                          */
int hesic_conv2d_set_phase_fusion(int mode);

/* Weight gradient of the same op: dw_packed[KH*KW][Cout][Cin] (fp32, packed layout) = sum over pixels of
 * dy (x) x; dbias (Cout, fp32) may be NULL.  Pixels are split over blocks; the partial tiles live in `ws`
 * (hesic_conv2d_wgrad_ws_bytes(d) bytes) and are summed in a fixed order, so the result is deterministic.
 * Unpack to the PyTorch layout with hesic_unpack_conv_wgrad.                                           */
int64_t hesic_conv2d_wgrad_ws_bytes(const hesic_conv_desc* d);
int hesic_conv2d_wgrad(const hesic_conv_desc* d, const void* x, const void* dy, float* dw_packed, float* dbias,
                       void* ws, int64_t ws_bytes, void* stream);
int hesic_unpack_conv_wgrad(const float* dw_packed, const float* mask, float* dw, int Cout, int Cin, int KH, int KW,
                            int transposed, void* stream);
/* The same gradient written straight in the PyTorch layout ((Cout,Cin,KH,KW), transposed: (Cin,Cout,KH,KW)) with the
 * K-slice reduce, the layout change and the bias column sums in ONE finishing launch; accumulate != 0: dw += ..., dbias += ...
 * (the gradients of a training step live in one flat buffer that is cleared once per step; a weight used twice per step --
 * encoder1, newnet1.py:726,754 -- simply adds twice).  Dead taps of a masked conv are left untouched when accumulating.  */
/* bf16 storage: the bias column sums are formed inside the split-K launch (one more MFMA per fragment against ones in the blocks of one tap)
 * and added up in a fixed order by the finishing launch -- deterministic, no second pass over dy.                        */
int hesic_conv2d_wgrad_direct(const hesic_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias, int accumulate,
                              void* ws, int64_t ws_bytes, void* stream);
/* Only the first of its two launches -- the split-K MFMA kernel that leaves the fp32 partial tiles in ws -- for profiling
 * (bench.py brackets it with HIP events to price the weight-gradient kernel against the MFMA roofline).               */
int hesic_conv2d_wgrad_partial(const hesic_conv_desc* d, const void* x, const void* dy, void* ws, int64_t ws_bytes, void* stream);
/* hesic_conv2d_wgrad_partial for n layers in one call: the split-K launches of the 128-channel-tile MFMA layers share grids of up to 14 jobs
 * (block ranges back to back, longest K slices first), so a layer with few K stages per block no longer pays its own launch ramp and tail;
 * the other layers are launched one by one.  Each job's partials land in its own ws[j] (hesic_conv2d_wgrad_ws_bytes(descs + j) bytes),
 * bit-identical to the one-by-one calls; hesic_conv2d_wgrad_finish_batched follows.  x[j] and dy[j] must stay valid until then -- the
 * backward pass of a training step (newtrain1.py:85-96) queues its layers and issues both calls every few layers.            */
int hesic_conv2d_wgrad_partial_batched(int n, const hesic_conv_desc* descs, const void* const* x, const void* const* dy, void* const* ws,
                                       const int64_t* ws_bytes, const int32_t* nsplit, void* stream);
/* K-slice counts named by the caller (nsplit[j]; NULL or 0 = the library's choice for a launch of its own).  A shared grid is kept full by
 * the other jobs' blocks, so a layer wants fewer and longer slices there -- fewer fp32 partial tiles to write and reduce:
 * hesic_conv2d_wgrad_nsplit(d, 1) is the count for that route (0: the one-launch-per-layer count; the row kernel's layers always keep
 * theirs), hesic_conv2d_wgrad_ws_bytes_n(d, nsplit) the workspace for it, and the finishing call must be given the same counts.          */
int hesic_conv2d_wgrad_nsplit(const hesic_conv_desc* d, int batched);
int64_t hesic_conv2d_wgrad_ws_bytes_n(const hesic_conv_desc* d, int nsplit);
int hesic_conv2d_wgrad_finish_batched_n(int n, const hesic_conv_desc* descs, const void* const* ws, const void* const* dy, float* const* dw,
                                        float* const* dbias, int accumulate, const int32_t* nsplit, void* stream);
/* The second launch of hesic_conv2d_wgrad_direct for n layers at once: job j reduces the K slices hesic_conv2d_wgrad_partial(descs + j,
 * ..) left in ws[j] into dw[j] (PyTorch layout) and sums dy[j]'s columns into dbias[j] (NULL: no bias), `accumulate` as above.  The
 * backward pass of a training step (newtrain1.py:85-96) has one such pass per conv layer -- 37 launches of ~17 us on small grids;
 * batches of 8 jobs share a launch.  Two jobs of one launch must not name the same dw (a weight used twice per step goes into
 * two calls); one storage type per call.                                                                                        */
int hesic_conv2d_wgrad_finish_batched(int n, const hesic_conv_desc* descs, const void* const* ws, const void* const* dy, float* const* dw,
                                      float* const* dbias, int accumulate, void* stream);

/* 3-channel image convs with arbitrary element strides on the image side (conv1 3->N, pre_conv 6->3,
 * g_s_conv4 N->3, after_conv 6->3: newnet1.py:583,629,612,670).  `img_*` strides describe the tensor
 * with the small channel count (input for Conv2d here, output for ConvTranspose2d).                   */
typedef struct {
    int32_t B, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, transposed;
    int32_t x_dtype, y_dtype, act;
    int64_t xs_b, xs_c, xs_y, xs_x;   /* element strides of x */
    int64_t ys_b, ys_c, ys_y, ys_x;   /* element strides of y */
} hesic_sconv_desc;

/* g_a_conv1 + g_a_gdn1 of the hi/lo analysis path in ONE kernel (conv(3, N) 5x5 stride 2 -> GDN(N), N == 128; newnet1.py:583-584,
 * :633-634): fp32 planar image in (unit pixel stride, even width), [hi(128) | lo(128)] bf16 out (ys_x >= 256, ys_c == 1).
 * image_hilo (128 KB) = hesic_sconv_pack_weight_image_hilo(w (128,3,5,5) fp32, raw GDN gamma (128,128) fp32): the LDS images
 * [w_hi | w_lo | gamma'_hi | gamma'_lo]; beta_packed as hesic_gdn_pack_params.  Other layouts: hesic_im2col_hilo + a 1x1
 * hesic_conv2d_forward_hilo.                                                                                                    */
int hesic_sconv_pack_weight_image_hilo(const float* w, const float* gamma, void* image_hilo, void* stream);
int hesic_sconv2d_gdn_forward_hilo(const hesic_sconv_desc* d, const float* x, const void* image_hilo, const float* bias,
                                   const float* beta_packed, int inverse, void* y_hilo, void* stream);
/* The same kernel for a one-product consumer (hesic_conv2d_gdn_forward_hilo_out): y leaves as ONE 16-bit value per channel (128
 * channels, ys_x >= 128).  Its rounding (2^-12) bounds what the arithmetic in front of it must deliver, so this form multiplies two
 * products per operand pair -- x and the squares as pairs, w and gamma' single -- with w rounded by error feedback over the taps:
 * image_hilo must come from hesic_sconv_pack_weight_image_hilo_out1 (same 128 KB layout; its lo halves are not read).               */
/* Pair form with the conv weights packed times `wscale` (a power of two): in the binary16 build the lo half of a weight of 0.02 is a subnormal
 * half (the pair then carries 2^-20 instead of 2^-22); with wscale = 2^s the caller hands hesic_sconv2d_gdn_forward_hilo bias * 2^s and
 * beta_packed * 4^s -- GDN's output is unchanged (v / sqrt(beta' + sum gamma' v^2) is invariant under v -> 2^s v, beta' -> 4^s beta').        */
int hesic_sconv_pack_weight_image_hilo_scaled(const float* w, const float* gamma, float wscale, void* image, void* stream);
int hesic_sconv_pack_weight_image_hilo_out1(const float* w, const float* gamma, void* image_hilo, void* stream);
int hesic_sconv2d_gdn_forward_hilo_out1(const hesic_sconv_desc* d, const float* x, const void* image_hilo, const float* bias,
                                        const float* beta_packed, int inverse, void* y, void* stream);

/* w is the raw fp32 PyTorch weight ((Cout,Cin,KH,KW) or, transposed, (Cin,Cout,KH,KW)). */
int hesic_sconv2d_forward(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias, void* y,
                          void* stream);
/* conv(torch.cat((xa, xb), 1)) without the cat: channels [0, ca) come from xa (d's x strides / dtype), [ca, Cin) from xb
 * (xb_strides in elements, b/c/y/x).  The 6 -> 3 5x5 stride-1 stages only: pre_conv on cat(x1_warp, x2) and after_conv on
 * cat(IGDN(.), x1_hat_warp) (newnet1.py:643,686).                                                                      */
int hesic_sconv2d_forward_cat(const hesic_sconv_desc* d, const void* xa, const void* xb, const int64_t xb_strides[4],
                              int xb_dtype, int ca, const float* w, const float* bias, void* y, void* stream);

/* The same launch with the 3-channel (I)GDN next to it fused in (compressai/layers/gdn.py:55-70 on three channels):
 * gdn_on_input = 0: y = (I)GDN(conv(cat(xa, xb)))        -- pre_conv -> GDN(3) of the second encoder (ywz/mywork/newnet1.py:643-644);
 * gdn_on_input = 1: y = conv(cat((I)GDN(xa), xb)), ca == 3 -- IGDN(3) -> cat -> after_conv of the second decoder (newnet1.py:684-686).
 * gdn_beta [3] / gdn_gamma [3][3] are the RAW parameters (reparametrised in the kernel like hesic_gdn_forward_planar). */
int hesic_sconv2d_forward_cat_gdn(const hesic_sconv_desc* d, const void* xa, const void* xb, const int64_t xb_strides[4],
                                  int xb_dtype, int ca, const float* w, const float* bias, const float* gdn_beta,
                                  const float* gdn_gamma, float beta_min, int inverse, int gdn_on_input, void* y, void* stream);
/* g_a_gdn1(g_a_conv1(image)) in one kernel (inference; 3 -> 128, 5x5 stride 2, bf16 NHWC output): newnet1.py:594-595.
 * gamma_packed / beta_packed from hesic_gdn_pack_params.                                                              */
int hesic_sconv2d_gdn_forward(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias,
                              const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* stream);
/* training form: also stores the conv output (bf16 NHWC, y's geometry) for GDN's backward */
int hesic_sconv2d_gdn_forward_train(const hesic_sconv_desc* d, const void* x, const float* w, const float* bias,
                                    const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                    void* stream);
/* The two image-side MFMA kernels keep their whole weight panel in LDS.  Building that LDS image inside the kernel (scalar
 * gathers from the PyTorch layout, several dependent round trips, by every block of every launch) cost ~14 us of a ~49 us launch;
 * hesic_sconv_pack_weight_image builds it ONCE per weight update and the *_prepacked forms start with a straight copy.
 *   kind 0: g_a_conv1 + GDN (3 -> 128, 5x5 s2): image = 65536 bytes, needs gamma_packed (hesic_gdn_pack_params)
 *   kind 1: g_s_conv4 (128 -> 3 transposed, 5x5 s2): image = 24576 bytes, gamma_packed ignored
 *   kind 2: as kind 1 with error-feedback rounding over the taps of each output phase (16-bit inference; round 5)
 * w is still passed (other geometries fall back to the ordinary kernels, which read it).                              */
int hesic_sconv_pack_weight_image(int kind, const float* w, const void* gamma_packed, void* image, void* stream);
int hesic_sconv2d_forward_prepacked(const hesic_sconv_desc* d, const void* x, const float* w, const void* w_image, const float* bias,
                                    void* y, void* stream);
int hesic_sconv2d_gdn_forward_prepacked(const hesic_sconv_desc* d, const void* x, const float* w, const void* w_image, const float* bias,
                                        const void* gamma_packed, const float* beta_packed, int inverse, void* y, void* y_pre,
                                        void* stream);
/* dx of the same op (dy has y's strides, dx has x's strides). */
int hesic_sconv2d_dgrad(const hesic_sconv_desc* d, const void* dy, const float* w, void* dx, void* stream);
/* dw (raw PyTorch layout, fp32) and dbias (may be NULL).  ws (hesic_sconv2d_wgrad_ws_bytes(d) bytes, may be NULL/0) enables the
 * matrix-core route for the 3 <-> 128 stages (im2col of the image side + the 1x1 weight-gradient kernel).                     */
int64_t hesic_sconv2d_wgrad_ws_bytes(const hesic_sconv_desc* d);
int hesic_sconv2d_wgrad(const hesic_sconv_desc* d, const void* x, const void* dy, float* dw, float* dbias, void* ws,
                        int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------ GDN / IGDN (row A3)
 * Replaces GDN.forward (compressai/layers/gdn.py:55-70).  beta/gamma are the RAW parameters (reparam
 * domain); the NonNegativeParametrizer (compressai/ops/parametrizers.py:41-44) is applied inside.
 * x, y: NHWC with C channels, P = B*H*W pixels.                                                       */
int hesic_gdn_forward(const void* x, const float* beta, const float* gamma, void* y, int64_t P, int C,
                      int inverse, float beta_min, int dtype, void* stream);
/* Same op on a planar (B,C,H,W) image, C = 3: pre_gdn / after_gdn (newnet1.py:630,669) without a layout copy.          */
int hesic_gdn_forward_planar(const void* x, const float* beta, const float* gamma, void* y, int B, int C, int64_t HW,
                             int inverse, float beta_min, int dtype, void* stream);
/* dx (same dtype as x), dbeta (C) and dgamma (C*C) fp32, gradients w.r.t. the RAW parameters
 * (LowerBound rule of compressai/ops/bound_ops.py:28-31 included).  ws: fp32 workspace of
 * hesic_gdn_backward_ws_bytes(P,C) bytes.                                                             */
int64_t hesic_gdn_backward_ws_bytes(int64_t P, int C);
int hesic_gdn_backward(const void* x, const void* dy, const float* beta, const float* gamma, void* dx,
                       float* dbeta, float* dgamma, void* ws, int64_t P, int C, int inverse, float beta_min,
                       int dtype, void* stream);
/* accumulate != 0: dbeta += ..., dgamma += ... (flat gradient buffer of a training step) */
int hesic_gdn_backward_acc(const void* x, const void* dy, const float* beta, const float* gamma, void* dx,
                           float* dbeta, float* dgamma, int accumulate, void* ws, int64_t P, int C, int inverse,
                           float beta_min, int dtype, void* stream);
/* Round 5: the parameter-gradient finishing passes of several fused GDN backwards (compressai/layers/gdn.py:55-70 under autograd, the
 * backward of newtrain1.py:85-96) in ONE launch.  hesic_gdn_backward_partial = hesic_gdn_backward stopped behind its main kernel: dx is
 * complete, the per-block (dgamma' | dbeta') partials stay in ws (which the caller keeps alive); hesic_gdn_backward_partial_ok says
 * whether (P, C, dtype) has that form (C == 128, 16-bit storage).  hesic_gdn_param_finish_batched sums the partials of n such calls in
 * the order hesic_gdn_backward would (bit-identical), applies the reparametrisation chain and writes / adds (accumulate) dgamma, dbeta;
 * two jobs of one call must not share a gradient.                                                                               */
int hesic_gdn_backward_partial_ok(int64_t P, int C, int dtype);
int hesic_gdn_backward_partial(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, void* ws, int64_t P,
                               int C, int inverse, float beta_min, int dtype, void* stream);
int hesic_gdn_param_finish_batched(int n, const void* const* ws, const int64_t* P, const float* const* beta, const float* const* gamma,
                                   float* const* dgamma, float* const* dbeta, const float* beta_min, int accumulate, void* stream);
/* The 3-channel image-side GDNs (pre_gdn / after_gdn, newnet1.py:630,669) under autograd on PLANAR (B, 3, HW) tensors, as
 * hesic_gdn_forward_planar: dx and the parameter gradients from one pass over x and dy (no NHWC copies on either side, so the conv
 * behind pre_gdn keeps its planar-input kernel).  ws: at least 64 bytes.  accumulate as hesic_gdn_backward_acc.                    */
int hesic_gdn_backward_planar_acc(const void* x, const void* dy, const float* beta, const float* gamma, void* dx, float* dbeta,
                                  float* dgamma, int accumulate, void* ws, int B, int64_t HW, int C, int inverse, float beta_min, int dtype,
                                  void* stream);

/* ------------------------------------------------------------------- warp_perspective (row A10)
 * Replaces kornia.warp_perspective(src, M, dsize) (third party; call sites newnet1.py:746,753,767).
 * M_src_to_dst: (B,3,3) fp32 row-major on the DEVICE, maps source pixel -> destination pixel; the kernel
 * inverts it.  align_corners=1: exact inverse-map bilinear, zeros outside (kornia >= 0.5);
 * align_corners=0: kornia <= 0.4 default (samples at xs*W/(W-1) - 1/2).                              */
typedef struct {
    int32_t B, C, H, W, Ho, Wo, align_corners, src_dtype, dst_dtype;
    int32_t m_is_dst_to_src;          /* 1: M maps destination -> source pixels already (no inversion): warping by H^-1
                                         given H, Independent_EN.forward (newnet1.py:1290-1291)                      */
    int64_t ss_b, ss_c, ss_y, ss_x;   /* src element strides */
    int64_t ds_b, ds_c, ds_y, ds_x;   /* dst element strides */
} hesic_warp_desc;
int hesic_warp_perspective_forward(const hesic_warp_desc* d, const void* src, const float* M_src_to_dst, void* dst,
                                   void* stream);
/* d_src (fp32, same strides as src, must be zero-filled by the caller) += transpose of the gather. */
int hesic_warp_perspective_backward(const hesic_warp_desc* d, const void* d_dst, const float* M_src_to_dst,
                                    float* d_src, void* stream);

/* ------------------------------------------------------------ EntropyBottleneck (row A8)
 * Replaces EntropyBottleneck.forward (compressai/entropy_models/entropy_models.py:384-411) for the
 * default filters=(3,3,3,3).  params: fp32 [C][64] packed by hesic_eb_pack_params: 58 MLP values
 * (softplus / tanh NOT yet applied) + median (slot 58) + the likelihood lower bound (slot 60; 1e-9 in the reference,
 * 0 = none: EntropyModel(likelihood_bound=...), entropy_models.py:60-66).  z: NHWC (P pixels, C channels).  noise: NULL => eval
 * (round(z-med)+med), else training (z+noise).  z_hat in z's dtype, lik fp32 NHWC.                   */
#define HESIC_EB_PARAM_STRIDE 64
int hesic_eb_forward(const void* z, const float* params, const void* noise, void* z_hat, float* lik, int32_t* symbols,
                     int64_t P, int C, int dtype, void* stream);
/* Inference form for the bf16 mode: z is fp32 (hesic_conv2d_forward_f32out), z_hat is stored as out_dtype (it feeds the
 * hyper-synthesis convs), eval quantisation only.                                                          */
int hesic_eb_forward_f32in(const float* z, const float* params, void* z_hat, int out_dtype, float* lik, int32_t* symbols,
                           int64_t P, int C, void* stream);
/* Inference cache: apply softplus / tanh to a packed table once ([C][64] -> [C][64], slot 59 marks it); hesic_eb_forward
 * accepts either form, hesic_eb_backward needs the raw one.                                            */
int hesic_eb_prepare_params(const float* params, float* prepared, int C, void* stream);
/* dz (z dtype) and dparams [C][64] fp32 (zero-filled by caller; atomically accumulated).
 * g_lik: fp32 gradient of lik, g_zhat: gradient of z_hat (may be NULL).                              */
int hesic_eb_backward(const void* z, const float* params, const void* noise, const float* g_lik, const void* g_zhat,
                      void* dz, float* dparams, int64_t P, int C, int dtype, void* stream);

/* Training plumbing of the bottleneck parameters (EntropyBottleneck.__init__, entropy_models.py:262-300): the 13 MLP
 * tensors + quantiles <-> the [C][64] table, one launch each way.  Entry j: a (C, ...) fp32 tensor with `stride[j]` floats
 * per channel whose `width[j]` values starting at `first[j]` occupy table columns [col[j], col[j] + width[j]).  The struct
 * is read on the HOST and travels in the kernel arguments; ptr[] are device pointers.                              */
#define HESIC_EB_MAX_TENSORS 16
typedef struct {
    float* ptr[HESIC_EB_MAX_TENSORS];
    int32_t width[HESIC_EB_MAX_TENSORS], stride[HESIC_EB_MAX_TENSORS], first[HESIC_EB_MAX_TENSORS], col[HESIC_EB_MAX_TENSORS];
    int32_t n;
    float lik_bound;                 /* written to slot 60 by hesic_eb_pack_table */
} hesic_eb_layout;
int hesic_eb_pack_table(const hesic_eb_layout* layout_host, float* table, int C, void* stream);
/* d(tensor j)[c][first + k] (+)= dtable[c][col_j + k]: hesic_eb_backward's dparams back into the parameters' gradients */
int hesic_eb_scatter_grads(const hesic_eb_layout* layout_host, const float* dtable, int C, int accumulate, void* stream);
/* EntropyBottleneck.loss (entropy_models.py:345-348; newtrain1.py:94) forward + backward: loss[0] += sum |c(quantiles) -
 * (-t, 0, t)|, t = ln(2 / tail_mass - 1), cumulative parameters detached; dquantiles (C,3) (+)= the gradient (may be NULL).
 * params: the raw [C][64] table.                                                                                   */
int hesic_eb_aux_loss(const float* params, const float* quantiles, float tail_mass, float* loss, float* dquantiles, int C,
                      int accumulate, void* stream);

/* ------------------------------------------------- GaussianMixtureConditional (row A9) / Gaussian (A11)
 * Replaces GaussianMixtureConditional.forward (entropy_models.py:661-702) and
 * GaussianConditional.forward (:546-554).  y: NHWC M channels; scales, means: NHWC K*M channels with
 * channel k*M+m; weights: (B, K*M) fp32 (K==1 && weights==NULL => plain Gaussian).
 * use_means_in_quant: 0 => y_hat=round(y) (GMM), 1 => round(y-mu)+mu (GaussianConditional, K==1).
 * noise NULL => eval.  Outputs: y_hat (y dtype), lik fp32, symbols int32 (may be NULL).              */
typedef struct {
    int32_t B, HW, M, K, dtype, use_means_in_quant;
    int32_t sm_pix_stride, s_c_off, m_c_off;   /* scales/means may live in one buffer (chunk(2,1))   */
    float scale_bound, lik_bound;
} hesic_gmm_desc;
int hesic_gmm_forward(const hesic_gmm_desc* d, const void* y, const void* scales, const void* means,
                      const float* weights, const void* noise, void* y_hat, float* lik, int32_t* symbols,
                      void* stream);
/* Inference form for the bf16 mode: y, scales, means fp32 (d->dtype is ignored), y_hat stored as out_dtype; K in {1, 5},
 * eval quantisation only.  Same arithmetic as the bf16 kernel (branch-free erfc, 1.8e-7 relative).        */
int hesic_gmm_forward_f32in(const hesic_gmm_desc* d, const float* y, const float* scales, const float* means,
                            const float* weights, void* y_hat, int out_dtype, float* lik, int32_t* symbols, void* stream);
/* Per-element cumulative-frequency tables for the real bit-stream of HSIC.compress / decompress (newnet1.py:925-978,
 * :1137-1175): for image b, every listed channel and pixel, cdf[(j*HW + hw)*(A+1) + 0..A], A = 2*minmax+1,
 * = [0, cumsum(round(clip(pmf, 2^-16, 1) / sum * 65536))] over the shifted alphabet 0..2*minmax with the mixture pmf of
 * GaussianMixtureConditional._likelihood -- the reference's Python double loop as one launch.  channels: device int32.  */
int hesic_gmm_cdf(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                  const int32_t* channels, int n_channels, int minmax, uint32_t* cdf, void* stream);
/* The same with a choice of row order: pixel_major != 0 puts row (pixel hw, listed channel j) at hw * n_channels + j (the order a
 * pixel-by-pixel decoder consumes them) instead of j * HW + hw.                                                                       */
int hesic_gmm_cdf_rows(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                       const int32_t* channels, int n_channels, int minmax, int pixel_major, uint32_t* cdf, void* stream);
/* The same tables with n_channels and minmax read on the device (state = {unused, n_channels, minmax}, int32; channels sized for
 * max_channels): a launch that can be captured into a HIP graph and replayed for other images (the HESIC+ wavefront step).  Alphabets
 * of more than 1024 entries are not written (the caller then uses hesic_gmm_cdf).                                                     */
int hesic_gmm_cdf_dyn(const hesic_gmm_desc* d, int b, const void* scales, const void* means, const float* weights,
                      const int32_t* channels, int max_channels, const int32_t* state, int pixel_major, uint32_t* cdf, void* stream);
/* Gradients: dy (y dtype; only meaningful in noise mode), dscales/dmeans (scales dtype, same layout),
 * dweights (B,K*M) fp32 zero-filled by caller (atomic accumulate).                                   */
int hesic_gmm_backward(const hesic_gmm_desc* d, const void* y, const void* scales, const void* means,
                       const float* weights, const void* noise, const float* g_lik, const void* g_yhat, void* dy,
                       void* dscales, void* dmeans, float* dweights, void* stream);

/* -------------------------------------------------------------------------------- glue (rows A6,A7,A12)
 * Bilinear x4 upsample, align_corners=True (nn.UpsamplingBilinear2d, newnet1.py:524), written into
 * channels [y_c_off, y_c_off+C) of an NHWC buffer with y_pix_stride channels (fused torch.cat :557). */
int hesic_upsample4_forward(const void* x, void* y, int B, int H, int W, int C, int y_pix_stride, int y_c_off,
                            int dtype, void* stream);
int hesic_upsample4_backward(const void* dy, void* dx, int B, int H, int W, int C, int y_pix_stride, int y_c_off,
                             int dtype, void* stream);
/* Copy an NHWC tensor into a channel slice of a wider NHWC buffer (the other half of torch.cat). */
int hesic_copy_channels(const void* x, void* y, int64_t P, int C, int x_pix_stride, int x_c_off, int y_pix_stride,
                        int y_c_off, int dtype, void* stream);
/* spatial_pool2d + LeakyReLU (newnet1.py:441-453,497): out[b,c] = leaky(max_{hw} x[b,hw,c]); argmax kept
 * for the backward.  out fp32 (B,C).                                                                  */
int hesic_spatial_max(const void* x, float* out, int32_t* argmax, int B, int HW, int C, int dtype, int leaky,
                      void* stream);
/* Its backward: dx (B,HW,C) of `dtype`, written whole -- g (B,C) fp32 at the arg-max pixel (x 0.01 where leaky and out <= 0), 0 elsewhere. */
int hesic_spatial_max_backward(const float* g, const float* out, const int32_t* argmax, void* dx, int B, int HW, int C, int dtype, int leaky,
                               void* stream);
/* The 1x1 conv + softmax-over-K head (newnet1.py:500,510-512): logits (B,K*M) = W (KM,KM) @ pooled + b;
 * weights[b, k*M+m] = softmax_k.  All fp32.                                                          */
int hesic_mix_weights_forward(const float* pooled, const float* w, const float* bias, float* logits, float* weights,
                              int B, int K, int M, void* stream);

/* The same 1x1 conv on the pooled vector as a differentiable pair (training form of newnet1.py:500; N = K*M):
 * logits (B,N) = pooled (B,N) @ W^T (N,N) + bias; backward: dpooled = g @ W, dw = g^T @ pooled, dbias = sum_b g.
 * All fp32, W in the Conv2d layout (N, N, 1, 1).  dpooled / dw / dbias may be NULL (not needed).       */
int hesic_pooled_linear_forward(const float* pooled, const float* w, const float* bias, float* logits, int B, int N, void* stream);
int hesic_pooled_linear_backward(const float* pooled, const float* w, const float* g, float* dpooled, float* dw, float* dbias,
                                 int B, int N, void* stream);

/* 3x3 stride-1 pad-1 convolution over a 32-channel NHWC bf16 map at full resolution -- the layer shape of the stage-2
 * enhancement net (ResidualBlock / Enhancement / Independent_EN: compressai/layers/layers.py:125-147,
 * ywz/mywork/newnet1.py:272-311, 1278-1300), inference form:  y = act(conv(x, w) + bias) + res1 + res2.
 *   Cout == 32: y, res1, res2 bf16 NHWC (B,H,W,32) (res may be NULL): conv2 of a ResidualBlock with its identity and, for the
 *               last block of an Enhancement_Block, the block's outer skip, in one pass;
 *   Cout <= 4 : y and res1 fp32 planar (B,Cout,H,W): the 32 -> 3 output conv plus the image it refines (res2 must be NULL).
 * w: fp32 (Cout,32,3,3) as stored by nn.Conv2d (packed in registers by the kernel), act: HESIC_ACT_*.            */
int hesic_conv3x3_c32_forward(const void* x, const float* w, const float* bias, int Cout, int act, const void* res1,
                              const void* res2, void* y, int B, int H, int W, void* stream);

/* A whole ResidualBlock of that net in one launch (compressai/layers/layers.py:125-147 with in_ch == out_ch == 32, inference):
 *   y = act(conv(act(conv(x, w1) + b1), w2) + b2) + x + res2
 * x, y, res2 bf16 NHWC (B,H,W,32) (res2 may be NULL: the Enhancement_Block's outer skip, newnet1.py:286), w1 / w2 fp32 (32,32,3,3),
 * b1 / b2 fp32 (32) or NULL.  The intermediate map stays in LDS.  Round 6 (row-rolling kernel on 16x16x32 MFMAs, a tap's 32 input
 * channels per instruction): the 288 products of an output value are summed in a different order than by two hesic_conv3x3_c32_forward calls and the
 * intermediate is rounded once to 16 bits as there -- agreement to <= 2^-8 of the output scale on < 2e-3 of the values (tests/test_gpu_ops.py), the
 * golden bars of the stage (en_64.npz) unchanged.
 * y must not alias x.                                                                                          */
int hesic_resblock_c32_forward(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, int act,
                               const void* res2, void* y, int B, int H, int W, void* stream);

/* Weight / bias gradient of the same convs (stage 2 trains the enhancement net with HSIC frozen: newnet1.py:272-311,
 * ywz/mywork/newtrain6_real.py): x and g are 32-channel NHWC bf16 maps (B,H,W,32) -- the layer input and the gradient w.r.t.
 * conv + bias (after the activation's derivative); dw is fp32 (Cout,Cin,3,3) with Cout, Cin <= 32 (a narrower conv -- the 6 -> 32
 * input layer on the zero-padded image map, the 32 -> 3 output layer on a zero-padded gradient map -- takes the first channels),
 * dbias fp32 (Cout) or NULL; accumulate != 0: dw += ..., dbias += ....  ws: hesic_conv3x3_c32_wgrad_ws_bytes() bytes of scratch.
 * (The data gradient is hesic_conv3x3_c32_forward itself on the mirrored, transposed weight.)                              */
int64_t hesic_conv3x3_c32_wgrad_ws_bytes(void);
int hesic_conv3x3_c32_wgrad(const void* x, const void* g, float* dw, float* dbias, int Cout, int Cin, int accumulate, void* ws,
                            int64_t ws_bytes, int B, int H, int W, void* stream);

/* torch.cat((xa, xb), 1) of two fp32 planar (B,3,H,W) images (Enhancement.forward, newnet1.py:300) written as channels 0..5
 * of a zero-padded (B,H,W,32) bf16 NHWC map: the 6 -> 32 input conv then runs on hesic_conv3x3_c32_forward with its weight
 * zero-padded along Cin.                                                                                  */
int hesic_pack_images_c32(const float* xa, const float* xb, void* out_nhwc32_bf16, int B, int H, int W, void* stream);

/* conv3x3(torch.cat((xa, xb), 1)) of Enhancement.forward (ywz/mywork/newnet1.py:300-301; compressai/layers/layers.py conv3x3) in ONE launch (round 6):
 * xa, xb fp32 planar (B,3,H,W) contiguous, w fp32 (32,6,3,3), bias fp32 (32) or NULL, y (B,H,W,32) 16-bit NHWC.  Bit-identical to
 * hesic_pack_images_c32 + hesic_conv3x3_c32_forward with the weight zero-padded along Cin, without the 32-channel round trip through HBM. */
int hesic_conv3x3_c32_forward_img6(const float* xa, const float* xb, const float* w, const float* bias, int act, void* y, int B, int H, int W,
                                   void* stream);

/* torch.optim.Adam(params, lr) update (ywz/mywork/newtrain1.py:294-295; no amsgrad, no weight decay) for up to
 * HESIC_ADAM_MAX_TENSORS fp32 tensors per call: p, g, m (exp_avg), v (exp_avg_sq) dense arrays of numel elements in the
 * same element order, step = the tensor's own fp32 step counter (device scalar, incremented by the call).  The struct is
 * read on the HOST (pointers inside are device pointers) and travels in the kernel arguments; block0 is filled in.   */
#define HESIC_ADAM_MAX_TENSORS 24
typedef struct {
    float* p[HESIC_ADAM_MAX_TENSORS]; const float* g[HESIC_ADAM_MAX_TENSORS]; float* m[HESIC_ADAM_MAX_TENSORS];
    float* v[HESIC_ADAM_MAX_TENSORS]; float* step[HESIC_ADAM_MAX_TENSORS];
    int64_t numel[HESIC_ADAM_MAX_TENSORS];
    int32_t block0[HESIC_ADAM_MAX_TENSORS + 1];
    int32_t n;
    float lr, beta1, beta2, eps;
} hesic_adam_chunk;
int hesic_adam_step(const hesic_adam_chunk* chunk_host, void* stream);

/* softmax over K of logits laid out (B, K*M) with channel k*M+m, and its backward
 * dlogits = w * (g - sum_k g*w).                                                                       */
int hesic_softmax_k_forward(const float* logits, float* weights, int B, int K, int M, void* stream);
int hesic_softmax_k_backward(const float* weights, const float* g, float* dlogits, int B, int K, int M, void* stream);

/* -------------------------------------------------------------------------------- reductions (rows T,M)
 * out[0] += sum(log2(lik)) over n fp32 values (bits = -out[0]);  fp64 accumulator on device.          */
int hesic_sum_log2(const float* lik, int64_t n, double* out, void* stream);
/* The criterion's three numbers from the sums above (newtrain1.py:37-56): acc = {sum log2 lik, sum sq diff view 1, view 2} (fp64, device);
 * out3 = {loss, bpp, mse} fp32 with bpp = -acc[0]/npix, mse = (acc[1]+acc[2])/numel, loss = lambda_255sq * mse + bpp.                 */
int hesic_rd_loss_combine(const double* acc, double lambda_255sq, int64_t npix, int64_t numel, float* out3, void* stream);
/* out[0] += sum((a-b)^2) with a, b given by element strides over a (B,C,H,W) index space.             */
int hesic_sum_sq_diff(const void* a, int a_dtype, const int64_t a_strides[4], const void* b, int b_dtype,
                      const int64_t b_strides[4], int B, int C, int H, int W, double* out, void* stream);
/* Round 5: the reductions behind one forward's bpp / PSNR (newtrain1.py:44-56, test3real.py:69-72) in ONE launch: n_lik <= 8 likelihood
 * maps (sum log2 into lik_out[i]) and n_sq <= 2 image pairs (sum of squared differences into sq_out[i]; strides / dims as four values per
 * pair: a_strides[4 i ..], dims = B, C, H, W).  The accumulators are ADDED to (the caller zero-fills them), as by hesic_sum_log2 /
 * hesic_sum_sq_diff, whose per-job sums this reproduces.                                                                          */
int hesic_rd_sums(int n_lik, const float* const* lik, const int64_t* numel, double* const* lik_out, int n_sq, const void* const* a,
                  const int* a_dtype, const int64_t* a_strides, const void* const* b, const int* b_dtype, const int64_t* b_strides,
                  const int* dims, double* const* sq_out, void* stream);

/* Backward of the two reductions inside the R-D loss (newtrain1.py:44-56):
 *   g_lik[i] = scale / lik[i]                      (d/dlik of scale * sum(ln lik))
 *   g_a[i]   = scale * (a[i] - b[i])  (fp32, a's (B,C,H,W) index space, contiguous NCHW output)         */
int hesic_log_backward(const float* lik, float scale, float* g_lik, int64_t n, void* stream);
int hesic_sq_diff_backward(const void* a, int a_dtype, const int64_t a_strides[4], const void* b, int b_dtype,
                           const int64_t b_strides[4], int B, int C, int H, int W, float scale, float* g_a,
                           void* stream);

/* elementwise helpers used by the autograd wrappers */
int hesic_act_backward(const void* y, const void* dy, void* dx, int64_t n, int act, int dtype, void* stream);
int hesic_cast(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream);
/* y = round_half_even(x): EntropyModel._quantize(x, "dequantize") without means (entropy_models.py:98-125; newnet1.py:755) */
int hesic_round(const void* x, int x_dtype, void* y, int y_dtype, int64_t n, void* stream);

/* ------------------------------------------------- in front of the path: HomographyNet -> h_matrix
 * The convolutions / linear layers of `Net` (ywz/mywork/model.py:73-96) go through hesic_(s)conv2d_forward (a Linear
 * is a 1x1 conv over the NHWC-flattened map); these are the remaining pieces.
 * MaxPool2d(2,2) of Block (model.py:62-63): NHWC (B,H,W,C) -> (B,H/2,W/2,C); C % 8 == 0 (bf16) / % 4 (fp32).        */
int hesic_maxpool2_forward(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* kornia.get_perspective_transform(src, dst) (model.py:26,108): H (B,3,3) fp32 with dst ~ H src, src/dst (B,4,2) fp32.
 * 4-point DLT with h33 = 1, solved per pair in fp64 (partial pivoting); NaNs for a singular configuration.          */
int hesic_perspective_transform(const float* src, const float* dst, float* H, int B, void* stream);
/* corner deltas -> the h_matrix HSIC.forward takes (newtrain1_real.py:113-123): corners0 = corners - corners[:,0]
 * (subtract_origin != 0); h = get_perspective_transform(corners0, corners0 + delta); h_matrix = h_adjust(inverse(h))
 * with the reference's scaling (:47-57: row 0 *= a, col 0 /= a, row 1 *= b, col 1 /= b; a = H_img/pic, b = W_img/pic).
 * subtract_origin = 0 and a = b = 1 is Net.get_h (model.py:99-111).                                                  */
int hesic_h_from_delta(const float* corners, const float* delta, float ratio_a, float ratio_b, int subtract_origin,
                       float* H, int B, void* stream);

/* ------------------------------------------------------------------ HESIC+ wavefront decode: the first node of a group's graph (row f3)
 * The reference's decoder walks the latent map pixel by pixel (ywz/mywork/newnet1_joint.py:1190-1260); here the pixels of one
 * wavefront group t = w + 3h are decoded together and the device work of a group is ONE HIP graph replayed per step.  This launch opens
 * that graph and keeps every per-step quantity in device memory (one block, phases separated by block barriers):
 *  1. scatter the previous group's symbols (sym: [nprev][C] int32, pixel-major; value = sym - minmax) into the padded latent map y_rows
 *     ((Hp*Wp) rows of M values of `dtype`) at rows prev_centre[], channels[];  state = {nprev, C, minmax} (device int32[3]);
 *  2. gather this group's P pixels (offset *pos in all_centre / all_rows: padded-map row of the pixel, raster row of the pixel): their
 *     5 x 5 crops -> crops[P][25][M], par[row][0..c_par) -> feat[p][0..c_par), ext[row][0..M) -> feat[p][e_off..e_off+M) (ext may be NULL);
 *  3. *pos += P; state[0] = P; prev_centre[0..P) = the group's padded rows.
 * P == 0: step 1 only (after the last group).  Rows must be whole 16-byte chunks.                                                    */
int hesic_joint_step(void* y_rows, int dtype, int M, int Wp, const int32_t* sym, int64_t* prev_centre, int32_t* state,
                     const int32_t* channels, const int64_t* all_centre, const int64_t* all_rows, int64_t* pos, int P, void* crops,
                     const void* par, int c_par, const void* ext, int e_off, void* feat, int c_feat, void* stream);

/* Host-loop helpers of the decode walk: hipMemcpyAsync (kind 1 = host -> device, 2 = device -> host; pinned host memory) and
 * hipStreamSynchronize behind the same error plumbing.                                                                               */
int hesic_memcpy_async(void* dst, const void* src, size_t bytes, int kind, void* stream);
/* The decode walk of one view of HESIC+ (newnet1_joint.py:1190-1260 regrouped into wavefronts) as ONE call: for each of the n_groups groups
 * (group_size[g] pixels) -- copy the previous group's symbols (sym_host, pinned) to sym_dev, launch graph_exec[g] (a hipGraphExec_t: the
 * group's captured device step, whose output scale_mean[g] is [P][2M] fp32 rows), hesic_gmm_cdf(descs[g]) into tab_dev, copy the tables to
 * (rows pixel-major: hesic_gmm_cdf_rows) tab_host (pinned), wait (spin != 0: poll hipStreamQuery), and call decode(decoder, tab_host, P,
 * n_channels, n_channels, 1, 2*minmax+2, sym_host)
 * -- the signature of hesic_rc_decoder_decode_grid (libhesic_host.so).  descs == NULL: the table launch is part of graph_exec[g]
 * (hesic_gmm_cdf_dyn) and is not issued again.  sym_dev == sym_host / tab_dev == tab_host (pinned, device-addressable
 * memory used by the kernels directly) skips the respective copies.  The last group's symbols are copied up before returning; the
 * caller scatters them (hesic_joint_step with P = 0).  A decoder error or a HIP error ends the walk with a non-zero return.              */
typedef int (*hesic_decode_grid_fn)(void* decoder, const uint32_t* cdf, int64_t n_outer, int64_t n_inner, int64_t row_step_outer,
                                    int64_t row_step_inner, int32_t stride, int32_t* symbols_out);
int hesic_joint_decode_groups(int n_groups, const int32_t* group_size, void* const* graph_exec, const hesic_gmm_desc* descs,
                              void* const* scale_mean, const int32_t* channels, int n_channels, int minmax, uint32_t* tab_dev,
                              uint32_t* tab_host, int32_t* sym_dev, int32_t* sym_host, hesic_decode_grid_fn decode, void* decoder, int spin,
                              void* stream);
/* The same walk with each group's device step given as a TAPE of recorded launches instead of a graph: the entry point (HESIC_TAPE_*) and
 * its arguments as 64-bit words, the trailing stream argument left out (supplied at replay).  Back-to-back launches of a dependent chain
 * of small kernels start closer together than the nodes of a captured graph (ROCm 7.2).                                                */
enum { HESIC_TAPE_JOINT_STEP = 0, HESIC_TAPE_CONV2D_FORWARD = 1, HESIC_TAPE_CONV2D_FORWARD_F32OUT = 2 };
typedef struct { int32_t fn, nargs; uint64_t a[20]; } hesic_tape_call;
int hesic_joint_decode_groups_tape(int n_groups, const int32_t* group_size, const hesic_tape_call* const* tapes, const int32_t* tape_len,
                                   const hesic_gmm_desc* descs, void* const* scale_mean, const int32_t* channels, int n_channels, int minmax,
                                   uint32_t* tab_dev, uint32_t* tab_host, int32_t* sym_dev, int32_t* sym_host, hesic_decode_grid_fn decode,
                                   void* decoder, int spin, void* stream);
int hesic_stream_synchronize(void* stream);

/* Measurement aid, not part of the path (bench.py `roofline.power_state`): a register-only loop of independent 32x32x16 MFMAs on the library's
 * 16-bit format over the whole chip -- the rate the matrix pipe sustains under the board's power management for operands of the kind `src_64k`
 * holds (>= 64 KB of 16-bit values: random, or zeros).  `sink_4k`: 4 KB the kernel never really writes.  *flops_out = flops of the launch. */
int hesic_probe_mfma_loop(const void* src_64k, float* sink_4k, int iters, double* flops_out, void* stream);

/* ------------------------------------------------------------------ MS-SSIM (row M: the second published quality metric)
 * The reference's evaluation reports pytorch_msssim.ms_ssim(x_hat, x, data_range=1, size_average=False) next to PSNR
 * (ywz/mywork/test3real.py:107-109; third party, absent: algorithm restated, see oracle/hesic_oracle.py::ms_ssim).
 * hesic_ssim_scale: ONE scale -- for every (image, channel) the SUMS over the valid positions ((H-10) x (W-10)) of the ssim and cs
 * maps are ADDED to sums[(b*C + c)*2 + {0,1}] (fp64, zeroed by the caller); x / y are fp32 with element strides (b, c, row, col).
 * hesic_avgpool2_pad: the 2 x 2 average pool between scales (zero padding of odd sides, divisor 4), contiguous fp32 planar output of
 * ((H + 2 (H%2) - 2)/2 + 1) x ((W + 2 (W%2) - 2)/2 + 1).  The five-scale combination is host arithmetic on B*C*5*2 numbers.      */
int hesic_ssim_scale(const float* x, const int64_t x_strides[4], const float* y, const int64_t y_strides[4], int B, int C, int H, int W,
                     float data_range, double* sums, void* stream);
int hesic_avgpool2_pad(const float* x, const int64_t x_strides[4], float* y, int B, int C, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HESIC_HIP_H */
