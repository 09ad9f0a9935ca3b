/* propainter_mi355.h -- C ABI of libpropainter_mi355.so
 *
 * The drop-in boundary of the MI355X-native ProPainter hot path.  The reference
 * (daniabib/ComfyUI_ProPainter_Nodes) is pure Python: its "FFI" for this path is
 * the set of torch / torchvision / scipy operator calls made by
 * propainter_inference.py and model/ (kernel rows K1..K20 of
 * section 2.2).  Every entry point below names the reference call site(s) it
 * replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller
 *     (PyTorch tensor storage); the library never allocates persistent memory;
 *   - activations are channels-last: [N][H][W][C] with an explicit channel pitch
 *     `ldc` (elements per pixel) so that a "torch.cat along C" is a view;
 *   - dtype codes PP_F32 / PP_F16 / PP_U8 / PP_I32;
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = null stream);
 *   - return 0 on success, negative pp_status on error; pp_last_error() gives text;
 *   - no global mutable state besides the per-thread error string.
 *
 * Struct layout rule (parsed by comfyui_propainter_nodes_amd/lib.py to build the
 * ctypes mirrors): only `const void*`, `void*`, `int64_t`, `int32_t`, `float`
 * members, optionally fixed-size arrays of those.
 */
#ifndef PROPAINTER_MI355_H_
#define PROPAINTER_MI355_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PP_ABI_VERSION 12

enum pp_dtype {
  PP_F32 = 0,
  PP_F16 = 1,
  PP_U8 = 2,
  PP_I32 = 3,
  /* pp_conv2d only: f32 tensors multiplied on the f16 matrix pipe with two-term operand splits
   * (v = h + l, h = f16_rtz(v), l = f16_rtz(v - h) -- r05 / ABI v9: the low term is UNSCALED, the matrix pipe honours f16
   * subnormals --, three MFMAs per product into one fp32 accumulator: |v - h - l| <= max(2^-20 |v|, 2^-24); beyond the f16
   * range both terms saturate: |v| <= 131008 is representable to <= 32 absolute, larger values saturate, never Inf/NaN).
   * Weights must be in the split packing described at pp_conv2d. */
  PP_F32X2 = 4
};

enum pp_status {
  PP_OK = 0,
  PP_ERR_BAD_ARG = -1,
  PP_ERR_UNSUPPORTED = -2,
  PP_ERR_LAUNCH = -3
};

enum pp_act { PP_ACT_NONE = 0, PP_ACT_RELU = 1, PP_ACT_LEAKY = 2, PP_ACT_SIGMOID = 3, PP_ACT_TANH = 4, PP_ACT_GELU = 5 };

enum pp_epilogue {
  PP_EPI_NONE = 0,
  PP_EPI_MUL_AUX1 = 1,       /* y = v * aux1                      (GRU r*h, update.py:44-46)   */
  PP_EPI_ADD_AUX1 = 2,       /* y = v + aux1                      (residual adds)              */
  PP_EPI_ADD_AUX1_RELU = 3,  /* y = relu(v + aux1)                (extractor.py:57)            */
  PP_EPI_GRU = 4             /* y = (1-aux1)*aux2 + aux1*v        (update.py:47, z=aux1,h=aux2) */
};

enum pp_pad_mode { PP_PAD_ZEROS = 0, PP_PAD_REPLICATE = 1 };

#define PP_MAX_SEG 4

int32_t pp_version(void);
const char* pp_last_error(void);
/* size in bytes of a parameter struct by name, for ABI self-checks */
int64_t pp_struct_size(const char* name);
/* The tuning / test knobs (PP_CONV_HALO, PP_CONV_HALO_CT, PP_CONV_KSPLIT, PP_CONV_TILE, PP_CONV_DIRECT, PP_CONV_TRACE:
 * csrc/pp_options.h) are read from the environment once, at the first call that needs them; pp_reload_options() reads
 * them again (tests that switch a kernel family inside one process).  No knob changes what an entry point computes. */
void pp_reload_options(void);

/* ------------------------------------------------------------------------------------
 * pp_conv2d -- implicit-GEMM convolution on MFMA (f16 inputs: 16x16x32; f32 inputs: 32x32x2 /
 * 16x16x4 exact-f32 products, or PP_F32X2: three f16 products per multiply-add), fp32
 * accumulate, fused bias/activation/epilogue.
 * Replaces every torch.nn.Conv2d / Conv3d(1,k,k) / Conv3d(3,1,1) / Linear / matmul on
 * the hot path: RAFT extractor.py:170-193, update.py:6-154, corr.py:52-60 (volume as a
 * batched 1x1), recurrent_flow_completion.py:17-26,69-75,162-300, propainter.py:48-57,
 * 100-110,238-275,304-312, sparse_transformer.py:16,33,47,53,65,83-84,162-171.
 *
 * Input = channel-concatenation of up to PP_MAX_SEG channels-last tensors of the
 * same [N][H][W] (segment s contributes in_C[s] channels, read with pitch in_ldc[s]).
 * Weights are pre-packed by the host (weights.py: pack_conv_weight):
 *   w[z][cout][tap = ky*kw+kx][seg][c padded to a multiple of 32], dtype = `dtype`.
 * PP_F32X2: inputs / bias / outputs are f32; every 32-channel chunk of the f32 packing (128
 * bytes) is replaced by 32 f16 values h = f16(S w) followed by 32 f16 values l = f16(S w - h), S a power of two per layer
 * chosen by the packer (ops.py: split_pack_weight; 1 / S travels as `acc_scale`, ABI v9); same byte size and chunk order as the
 * f32 packing.
 * gridDim.z = Z selects a group (grouped conv) or a batch item (batched GEMM):
 * every pointer advances by its *_zoff (in elements) per z.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;      /* PP_F32, PP_F16 or PP_F32X2: inputs and weights */
  int32_t out_dtype;  /* PP_F32 or PP_F16: out, aux1, aux2 */
  int32_t nseg;
  int32_t pad_mode;
  const void* in_ptr[PP_MAX_SEG];
  int64_t in_C[PP_MAX_SEG];
  int64_t in_ldc[PP_MAX_SEG];
  int64_t in_zoff[PP_MAX_SEG];
  int64_t N, H, W, Ho, Wo;
  int32_t kh, kw, sh, sw, ph, pw, dh, dw;
  const void* weight;
  int64_t w_zoff;
  const void* bias; /* fp32 [Cout] or NULL */
  int64_t bias_zoff;
  int64_t Cout; /* per z */
  int64_t Z;
  void* out;
  int64_t out_ldc;
  int64_t out_zoff;
  int32_t act;       /* pp_act for channels < act_split (or all when act_split<=0) */
  int32_t act2;      /* pp_act for channels >= act_split */
  int32_t act_split; /* 0 = single activation */
  int32_t epi;       /* pp_epilogue */
  float act_param;   /* leaky slope */
  float out_scale;   /* multiplies act() output of channels < act_split (all if 0) */
  const void* aux1;
  int64_t aux1_ldc;
  int64_t aux1_zoff;
  const void* aux2;
  int64_t aux2_ldc;
  int64_t aux2_zoff;
  const void* pre_add; /* optional [N][Ho][Wo][Cout] (out dtype) added BEFORE the activation: a partial
                          convolution over constant input channels computed once (RAFT GRU context term) */
  int64_t pre_add_ldc;
  int64_t epi_from;    /* (ABI v6) the epilogue op applies to output channels >= epi_from only, and reads aux1 / aux2 at
                          channel (c - epi_from); multiple of 4; 0 = all channels.  RAFT's GRU computes z and r in ONE
                          256-channel convolution: sigmoid on all, r * h (PP_EPI_MUL_AUX1) on channels 128..255 */
  const void* weight_f32; /* optional (ABI v5), only read when Cout <= 4: the same weights as fp32
                             [tap][32-channel chunk][Cout padded to 2 or 4][32] for the vector-ALU kernel of the 2-3
                             channel layers (conv_direct.hip reads them through the scalar cache instead of decoding
                             `weight` into LDS per work-group); NULL = decode from `weight` */
  int32_t flat_taps; /* (ABI v8, PP_F16, nseg 1, in_C % 8 == 0) != 0: the weights are ONE row of kh*kw*in_C channels per
                        output channel in (ky, kx, c) order, zero-padded to a multiple of 32 at the END only -- the Linear
                        that consumes F.unfold()'s tap-major patch vectors -- and the kernel gathers the patches itself:
                        conv(x) == linear(unfold(x)) without the unfolded matrix (sparse_transformer.py:413-433: fc2 of the
                        fusion feed-forward reads the folded 40-channel map, 7x7 / stride 3, instead of a 49x copy).
                        (ABI v10) also with PP_F32X2 on an f32 input of 1..4 channels (any pitch, 4-byte aligned), Cout 64 or
                        128, kh*kw*in_C <= 160, zero padding, any stride: RAFT's 7x7 convolutions on the 2-channel flow
                        (update.py:100-106, every GRU iteration) and on the 3-channel frames (extractor.py:130-136) without
                        pp_im2col's patch tensor -- conv_patch.hip builds the patches as MFMA fragments in LDS */
  float acc_scale;   /* (ABI v9, PP_F32X2) the accumulators are multiplied by this before the bias is added; 0 = 1.  The packed
                        weights of a layer carry a power-of-two scale S (ops.split_pack_weight: max|w| S in [8192, 16384), so that
                        the LOW f16 term of every weight is a normal number) and acc_scale = 1 / S undoes it exactly */
  int32_t many_images; /* (ABI v12) != 0: the caller states that this LAYER always runs on a batch of many images (flow completion's
                          encoder / decoder layers over a whole sub-video: 2 (T - 1) images), so the kernel for small images with a
                          long reduction (in-work-group split K, built for the ONE-time-step launches of the recurrences) is not
                          considered.  Kernel selection otherwise never looks at the batch -- it decides on the pixels of one image,
                          Cout and K, so that a rank of a sharded run and the single-GPU run sum every layer in the same order --
                          and this flag keeps that property: it belongs to the layer, not to the launch.  0 = the r02-r06 rules */
} pp_conv2d_params;

int32_t pp_conv2d(void* stream, const pp_conv2d_params* p);

/* ------------------------------------------------------------------------------------
 * pp_split_pack -- PP_F32X2 packing of an fp32 matrix [rows][K] (K % 32 == 0) that will be the WEIGHT operand of
 * pp_conv2d(PP_F32X2): the all-pairs correlation volume (corr.py:52-60) multiplies two activation tensors, so the
 * packing the host does once for constant weights (ops.split_pack_weight) runs on the device per clip.
 * out: same byte size, every 32-float chunk -> 32 f16 h = f16_rtz(v) | 32 f16 l = f16_rtz(v - h)  (r05: the low term is
 * UNSCALED -- the matrix pipe honours f16 subnormals, tools/probes/mfma_denorm.hip -- so all three products of a
 * multiply-add, wh xh + wh xl + wl xh, go to ONE fp32 accumulator).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* in;
  void* out;
  int64_t rows, K;
} pp_split_pack_params;
int32_t pp_split_pack(void* stream, const pp_split_pack_params* p);

/* ------------------------------------------------------------------------------------
 * pp_im2col -- explicit patch matrix for the few convolutions whose Cin is too small for
 * 16-byte channel pieces (RAFT extractor.py:135 conv1 3->64 k7 s2, update.py:100 convf1
 * 2->128 k7, propainter.py:240 5->64 k3 s2, recurrent_flow_completion.py:240 3->32 k5 s2
 * replicate).  out[m][k], m = (n,ho,wo), k = (ky*kw+kx)*C + c, zero-filled up to Kpad;
 * the result feeds pp_conv2d as a 1x1 convolution.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;     /* input dtype */
  int32_t out_dtype;
  int32_t pad_mode;
  int32_t kh, kw, sh, sw, ph, pw;
  const void* in;
  int64_t in_ldc;
  int64_t N, H, W, C, Ho, Wo;
  void* out;
  int64_t Kpad;
} pp_im2col_params;
int32_t pp_im2col(void* stream, const pp_im2col_params* p);

/* ------------------------------------------------------------------------------------
 * pp_instnorm -- torch.nn.InstanceNorm2d(affine=False, eps=1e-5) of RAFT's fnet
 * (extractor.py:32-36,134-135, forward :44-57,181-183) on channels-last fp32, fused with
 * the surrounding ReLU / residual: y = post( skip + pre( (x-mean)*rstd ) ).
 * `partials` is caller-owned scratch of N*nchunks*C*2 doubles.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  int64_t x_ldc;
  void* y;
  int64_t y_ldc;
  const void* skip; /* optional residual input (same shape), NULL = none */
  int64_t skip_ldc;
  int64_t N, HW, C;
  int32_t relu_pre;
  int32_t relu_post;
  void* partials;
  int64_t nchunks;
  float eps;
} pp_instnorm_params;
int32_t pp_instnorm(void* stream, const pp_instnorm_params* p);

/* ------------------------------------------------------------------------------------
 * pp_avgpool2x2 -- F.avg_pool2d(corr, 2, stride=2) over a batch of fp32 planes
 * (corr.py:24-27): in [B][H][W] -> out [B][H/2][W/2] (floor).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* in;
  void* out;
  int64_t B, H, W;
  int32_t in_tiled;  /* (ABI v6) planes stored as [ceil(H/4)][ceil(W/8)][4][8] tiles of 128 bytes (see pp_corr_lookup) */
  int32_t out_tiled;
} pp_avgpool2x2_params;
int32_t pp_avgpool2x2(void* stream, const pp_avgpool2x2_params* p);

/* ------------------------------------------------------------------------------------
 * pp_corr_lookup -- CorrBlock.__call__ (corr.py:29-50) + bilinear_sampler
 * (RAFT/utils/utils.py:66-80): for every pixel of every pair, sample a 9x9 window around
 * (grid + flow)/2^l in each of the 4 pyramid levels (bilinear, zeros outside,
 * align_corners=True).  out[n][y][x][l*81 + i*9 + j], i offsets X, j offsets Y.
 * pyramid level l: fp32 [N][h*w][h_l][w_l]; flow: channels-last (dx,dy) with pitch flow_ldc.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* pyr[4];
  int64_t ph[4];
  int64_t pw[4];
  const void* flow;
  int64_t flow_ldc;
  void* out;
  int64_t out_ldc;
  int64_t N, h, w;
  int32_t tiled[4]; /* (ABI v6) level l is stored in 4 x 8 tiles: element (y, x) of a plane at
                       ((y/4) * ceil(w_l/8) + x/8) * 32 + (y%4) * 8 + x%8, plane pitch ceil(h_l/4) * ceil(w_l/8) * 32 floats.
                       A 12 x 12 lookup window then touches ~9 128-byte lines instead of ~17 on 320-byte rows (r02 measured
                       2.0x the algorithmic HBM traffic on row-major planes). */
} pp_corr_lookup_params;
int32_t pp_corr_lookup(void* stream, const pp_corr_lookup_params* p);

/* ------------------------------------------------------------------------------------
 * pp_corr_lookup_conv (ABI v10) -- the lookup FUSED with the motion encoder's first convolution:
 *   cor = act(convc1(CorrBlock(coords)))   corr.py:29-50 feeding update.py:94-112 (BasicMotionEncoder.convc1, relu),
 * evaluated every GRU iteration.  `lookup` describes the pyramid and the flow exactly as for pp_corr_lookup (its `out` /
 * `out_ldc` are ignored: the 324 sampled correlations of a pixel never reach memory -- they are built as the MFMA operand
 * of the projection in LDS); `conv` is the pp_conv2d block of the 1x1 PP_F32X2 convolution 324 -> 256 over those pixels
 * (N, H, W = the lookup's N, h, w; in_C[0] = 324, its in_ptr is not read; weight / bias / act / out / epilogue fields as for
 * pp_conv2d).  Same lookup arithmetic (operation by operation) and the same three-product PP_F32X2 arithmetic as the two
 * launches it replaces; 737 MB less written and read per iteration at 158 x 45 x 80 pixels.
 * ---------------------------------------------------------------------------------- */
int32_t pp_corr_lookup_conv(void* stream, const pp_corr_lookup_params* lookup, const pp_conv2d_params* conv);

/* ------------------------------------------------------------------------------------
 * pp_convex_upsample -- RAFT.upsample_flow (raft.py:81-92): softmax over the 9 taps of
 * mask[n][y][x][k*64 + a*8 + b], convex combination of 8*flow over the 3x3 neighbourhood
 * (zero padded).  out: channels-last [N][8h][8w][2] fp32.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* mask;
  int64_t mask_ldc;
  const void* flow;
  int64_t flow_ldc;
  void* out;
  int64_t N, h, w;
} pp_convex_upsample_params;
int32_t pp_convex_upsample(void* stream, const pp_convex_upsample_params* p);

/* ------------------------------------------------------------------------------------
 * pp_deform_cols -- sampling half of torchvision.ops.deform_conv2d (3x3, stride 1, pad 1,
 * dilation 1, weight groups 1; call sites recurrent_flow_completion.py:44-53 and
 * propainter.py:73-82): cols[n][y][x][k*Cin + c] = mask[g*9+k] * bilinear(x[.., c],
 * (y-1+ky+dy, x-1+kx+dx)) with c in deformable group g = c / (Cin/dg), each out-of-range
 * corner contributing 0.  The GEMM half is a 1x1 pp_conv2d over 9*Cin channels.
 * `om` holds the conv_offset output already activated: channels [0, 2*dg*9) offsets
 * (g*18 + 2k = dy, +1 = dx), channels [2*dg*9, 3*dg*9) modulation masks.
 * `flow` (optional, fp32 (dx,dy)) is added to every tap offset (propainter.py:67-68).
 * x may be the concatenation of two channels-last tensors (x0: C0 ch, x1: C1 ch).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype; /* x, om, cols dtype (PP_F16 or PP_F32) */
  int32_t dg;
  const void* x0;
  int64_t x0_C, x0_ldc;
  const void* x1;
  int64_t x1_C, x1_ldc;
  const void* om;
  int64_t om_ldc;
  const void* flow; /* fp32 [N][H][W][>=2] or NULL */
  int64_t flow_ldc;
  void* cols;
  int64_t N, H, W;
} pp_deform_cols_params;
int32_t pp_deform_cols(void* stream, const pp_deform_cols_params* p);

/* ------------------------------------------------------------------------------------
 * pp_deform_conv -- the whole modulated deformable convolution of the same two call sites
 * (recurrent_flow_completion.py:44-53, propainter.py:73-82: torchvision.ops.deform_conv2d with
 * 3x3 taps, stride 1, pad 1, weight groups 1) in ONE launch: pp_deform_cols followed by the
 * 1x1 pp_conv2d over the 9*Cin columns, without the column tensor.  It takes the two parameter
 * blocks of that pair: `sample` as for pp_deform_cols (`cols` is ignored; f16 tensors only),
 * `gemm` as for the 1x1 f16 pp_conv2d over [N][H][W][9*Cin] (one segment whose pointer is
 * ignored, kh = kw = 1, Z = 1; weight packed as for that convolution, K order tap-major; bias,
 * activations and the fused epilogue as in pp_conv2d).  The sampled values are rounded to f16
 * exactly as pp_deform_cols stores them, so the result differs from the two-launch form only by
 * the fp32 summation order of the kernel that would have run the 1x1 convolution.
 * Needs Cin % 32 == 0, x0_C % 32 == 0 and (Cin/dg) % 8 == 0 (128 / 256 channels, dg 16 here).
 * ---------------------------------------------------------------------------------- */
int32_t pp_deform_conv(void* stream, const pp_deform_cols_params* sample, const pp_conv2d_params* gemm);

/* ------------------------------------------------------------------------------------
 * pp_upsample2x -- F.interpolate(scale_factor=2, mode="bilinear", align_corners=True)
 * of the `deconv` blocks (recurrent_flow_completion.py:146-159, propainter.py:278-291),
 * channels-last.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  const void* in;
  int64_t in_ldc;
  void* out;
  int64_t out_ldc;
  int64_t N, H, W, C;
} pp_upsample2x_params;
int32_t pp_upsample2x(void* stream, const pp_upsample2x_params* p);

/* ------------------------------------------------------------------------------------
 * pp_rfc_prep -- input of RecurrentFlowCompleteNet (recurrent_flow_completion.py:366-377,
 * 322-325): out[t'][d][y][x] = (flow*(1-m), m, 0) as 4 f16 channels, for direction d = 0
 * (forward flows, masks[:-1]) and d = 1 (backward flows, masks[1:], TIME-FLIPPED: t' = T-1-t).
 * flows: fp32 [2][T][H][W][2]; masks: u8 [T+1][H][W]; out: f16 / f32 [T][2][H][W][4].
 * pp_flow_combine -- combine_flow (:389-400) incl. the flip back:
 * out[d][t] = pred*m + gt*(1-m), pred f16 / f32 [T][2][H][W][2] (time-flipped for d = 1).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* flows;
  const void* masks;
  void* out;
  int64_t T, H, W;
  int32_t out_dtype; /* PP_F16 (fp16 "enable") or PP_F32 (fp16 "disable"): storage type of the network's activations */
} pp_rfc_prep_params;
int32_t pp_rfc_prep(void* stream, const pp_rfc_prep_params* p);

typedef struct {
  const void* pred;
  int64_t pred_ldc;
  const void* flows;
  const void* masks;
  void* out; /* fp32 [2][T][H][W][2] */
  int64_t T, H, W;
  int32_t pred_dtype; /* PP_F16 or PP_F32 */
} pp_flow_combine_params;
int32_t pp_flow_combine(void* stream, const pp_flow_combine_params* p);

/* ------------------------------------------------------------------------------------
 * pp_img_prop_step -- one step of BidirectionalPropagation(learnable=False)
 * (propainter.py:149-205 incl. fbConsistencyCheck :27-36 and flow_warp
 * flow_loss_utils.py:6-51), fused per pixel, fp32 coordinates:
 *   valid = fbCheck(flow_prop, flow_check);  Fw = warp_nearest(F_prev, flow_prop)
 *   Mv = bin(warp_bilinear(M_prev, flow_prop));  U = m_cur & valid & !Mv
 *   F_new = U ? Fw : x_cur;   M_new = m_cur & !(valid & !Mv)
 * F_*: fp32 [H][W][3] (pitch 3), masks: u8 [H][W], flows: fp32 [H][W][2].
 * first != 0: F_new = x_cur, M_new = m_cur (the first frame of a pass, :158-160).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* f_prev;
  const void* m_prev;
  const void* x_cur;
  const void* m_cur;
  const void* flow_prop;
  const void* flow_check;
  void* f_new;
  void* m_new;
  int64_t H, W;
  int32_t first;
  int32_t mask_input; /* != 0: x_cur is used as x_cur*(1-m_cur)  (masked_frames, propainter_inference.py:171) */
} pp_img_prop_step_params;
int32_t pp_img_prop_step(void* stream, const pp_img_prop_step_params* p);

/* ------------------------------------------------------------------------------------
 * pp_pack_encoder_input -- image_propagation's blend (propainter_inference.py:194-198,
 * 218-221) fused with the generator's input concat (propainter.py:380-388):
 *   out[t][y][x] = (frames*(1-m) + prop*m (3 ch), m_in, m_updated, 0, 0, 0) as 8 f16
 * channels; also writes updated_frames fp32 [T][H][W][3] when `updated` != NULL.
 * frames/prop: fp32 [T][H][W][3]; m_in/m_upd: u8 [T][H][W].
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* frames;
  const void* prop;
  const void* m_in;
  const void* m_upd;
  void* out;
  void* updated;
  int64_t total_pixels;
  int32_t out_dtype; /* PP_F16 or PP_F32: storage type of the generator's activations */
} pp_pack_encoder_input_params;
int32_t pp_pack_encoder_input(void* stream, const pp_pack_encoder_input_params* p);

/* ------------------------------------------------------------------------------------
 * pp_flow_down4 -- F.interpolate(flow, scale_factor=1/4, bilinear, align_corners=False)/4
 * (propainter.py:391-408): out[n][i][j] = mean of the 2x2 block at (4i+1..2, 4j+1..2) / 4.
 * in fp32 [N][H][W][2] -> out fp32 [N][H/4][W/4][2].
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* in;
  void* out;
  int64_t N, H, W;
} pp_flow_down4_params;
int32_t pp_flow_down4(void* stream, const pp_flow_down4_params* p);

/* ------------------------------------------------------------------------------------
 * pp_featprop_aux -- per-step conditioning planes of the learnable propagation
 * (propainter.py:166-188): valid = fbConsistencyCheck(flow_prop, flow_check);
 * out[n][y][x] = (flow_prop.x, flow_prop.y, valid, mask[..0], mask[..1], 0,0,0) as 8 f16
 * channels.  flows fp32 [N][h][w][2]; maskpair f16 [N][h][w][8] (channels 0,1 used).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* flow_prop;
  const void* flow_check;
  const void* maskpair;
  void* out;
  int64_t N, H, W;
  int32_t dtype; /* maskpair and out: PP_F16 or PP_F32 */
} pp_featprop_aux_params;
int32_t pp_featprop_aux(void* stream, const pp_featprop_aux_params* p);

/* ------------------------------------------------------------------------------------
 * pp_flow_warp -- flow_warp(x, flow, "bilinear", zeros, align_corners=True)
 * (flow_loss_utils.py:6-51) on channels-last features; coordinates in fp32.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  const void* x;
  int64_t x_ldc;
  const void* flow; /* fp32 [N][H][W][2] */
  void* out;
  int64_t out_ldc;
  int64_t N, H, W, C;
} pp_flow_warp_params;
int32_t pp_flow_warp(void* stream, const pp_flow_warp_params* p);

/* ------------------------------------------------------------------------------------
 * pp_layernorm -- nn.LayerNorm(512) of the transformer blocks (sparse_transformer.py:413,
 * 429; eps 1e-5) on f16 tokens [t][fh][fw][C], fp32 statistics.  The output may live in a
 * zero-padded token grid [t][Hp][Wp][C] (the window padding of :212-216 happens AFTER the
 * norm, so padded tokens stay exactly zero: the caller zero-fills `out` once).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  void* out;
  const void* gamma; /* fp32 [C] */
  const void* beta;  /* fp32 [C] */
  int64_t T, fh, fw, Hp, Wp, C;
  float eps;
  int32_t dtype; /* x and out: PP_F16 or PP_F32 */
} pp_layernorm_params;
int32_t pp_layernorm(void* stream, const pp_layernorm_params* p);

/* ------------------------------------------------------------------------------------
 * pp_pool_tokens -- SparseWindowAttention.pool_layer (sparse_transformer.py:170-180,289):
 * depth-wise Conv2d(C, C, k4, s4, groups=C) over the padded token grid.
 * x f16 [T][Hp][Wp][C] -> out f16 [T][Hp/4][Wp/4][C]; weight fp32 [16][C] (tap-major), bias fp32 [C].
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* x;
  void* out;
  const void* weight;
  const void* bias;
  int64_t T, Hp, Wp, C;
  int32_t dtype; /* x and out: PP_F16 or PP_F32 */
} pp_pool_tokens_params;
int32_t pp_pool_tokens(void* stream, const pp_pool_tokens_params* p);

/* ------------------------------------------------------------------------------------
 * pp_window_attention -- SparseWindowAttention.forward core (sparse_transformer.py:218-385):
 * per (window, head), flash-style on MFMA, never materialising the rolled/pooled key copies.
 *   masked window  : queries = the window's 45 tokens of all t frames; keys/values = for each
 *                    frame in t_ind: 45 own + 148 rolled-neighbour (circular) + all pooled tokens;
 *   unmasked window: each frame attends to its own 45 tokens.
 * qkv f16 [t][Hp][Wp][3*512] (q|k|v of the padded grid), pkv f16 [t][npool][2*512] (k|v of the
 * pooled tokens), win_masked i32 [nwh*nww], t_ind i32 [nt]; out f16 [t][fh][fw][512] (unpadded).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* qkv;
  const void* pkv;
  const void* win_masked;
  const void* t_ind;
  void* out;
  int64_t t, nt, Hp, Wp, fh, fw, npool;
  float scale;
  int32_t dtype; /* qkv, pkv, out: PP_F16 or PP_F32 (fp32 storage is rounded to f16 for the MFMA operands unless `exact`) */
  int32_t exact; /* ABI v11, PP_F32 storage only: 1 = q, k, v and the probabilities stay fp32 and both products run on
                    v_mfma_f32_16x16x4_f32 with libm's expf -- the reference's fp32 attention (sparse_transformer.py:366-393)
                    at fp32 rounding level, ~5x the time of the default; 0 = f16 MFMA operands.  Must be 0 for PP_F16. */
} pp_window_attention_params;
int32_t pp_window_attention(void* stream, const pp_window_attention_params* p);

/* ------------------------------------------------------------------------------------
 * pp_fold / pp_unfold_gelu -- F.fold / F.unfold with kernel 7, stride 3, padding 3
 * (SoftComp sparse_transformer.py:56-62, FusionFeedForward :95-121).  Token vectors are laid
 * out TAP-MAJOR: channel (ky*7+kx)*C + c (the host permutes the producing / consuming Linear
 * weights accordingly).  pp_fold: out[t][y][x][c] = sum of the overlapping taps, divided by the
 * overlap count when `normalize` (the constant "fold of ones" map of :88-101).
 * pp_unfold_gelu: re-extract the patches and apply the exact (erf) GELU of fc2 (:83).
 * ABI v8: the GELU may be applied by pp_fold instead (`gelu` != 0: once per folded value, after the value has been rounded
 * to the storage type -- the number pp_unfold_gelu would read back) and pp_unfold_gelu told so (`pre_activated` != 0: it
 * then only copies).  Bit-identical to fold + unfold_gelu; 5x fewer erf evaluations.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* in; /* f16 [T][fh*fw][49*C] */
  void* out;      /* f16 [T][H][W][C] */
  int64_t T, H, W, C, fh, fw;
  int32_t normalize;
  int32_t dtype; /* in and out: PP_F16 or PP_F32 */
  int32_t gelu;  /* (v8) != 0: exact GELU on the (normalised, storage-rounded) result */
} pp_fold_params;
int32_t pp_fold(void* stream, const pp_fold_params* p);

typedef struct {
  const void* in; /* f16 [T][H][W][C] */
  void* out;      /* f16 [T][fh*fw][49*C] */
  int64_t T, H, W, C, fh, fw;
  int32_t dtype; /* in and out: PP_F16 or PP_F32 */
  int32_t pre_activated; /* (v8) != 0: `in` already holds GELU(x) (pp_fold with `gelu`): copy only */
} pp_unfold_gelu_params;
int32_t pp_unfold_gelu(void* stream, const pp_unfold_gelu_params* p);

/* ------------------------------------------------------------------------------------
 * pp_compose_u8 -- the uint8 compose of feature_propagation (propainter_inference.py:283-307):
 *   p = trunc_u8(((pred+1)/2)*255); img = m ? p : orig;
 *   first visit: comp = img; later: comp = trunc_u8(0.5*comp + 0.5*img)   (window order)
 * pred f16 [L][H][W][pred_ldc>=3] (tanh already applied), frame_ids i32 [L] (global frame index
 * of each local frame), first i32 [L]; masks u8 [T][H][W]; orig / comp u8 [T][H][W][3].
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* pred;
  int64_t pred_ldc;
  const void* frame_ids;
  const void* first;
  const void* masks;
  const void* orig;
  void* comp;
  int64_t L, H, W;
  int32_t pred_dtype; /* PP_F16 or PP_F32 */
} pp_compose_u8_params;
int32_t pp_compose_u8(void* stream, const pp_compose_u8_params* p);

/* ------------------------------------------------------------------------------------
 * Device-side pre / post-processing (the reference does this on the host with
 * numpy / PIL / scipy and marks it TODO at propainter_nodes.py:110,248).  All bit-exact.
 *
 * pp_frames_from_image -- convert_image_to_frames (utils/image_utils.py:106-116) + to_tensors
 * and "*2-1" (:178-191) + the outpaint canvas of extrapolation (:200-252):
 *   out_u8 [T][Ho][Wo][3] = trunc(clip(image*255, 0, 255)) placed at (oy, ox), 0 elsewhere;
 *   out_f32 (optional)    = (out_u8 / 255) * 2 - 1.
 * image: fp32 [T][H][W][3] (ComfyUI IMAGE).  With image == NULL the uint8 frames in_u8
 * [T][Ho][Wo][3] (resized on the host by PIL) are only converted to out_f32.
 * pp_image_from_u8 -- handle_output (:276-290): out fp32 = float(k) / 255; total % 4 == 0.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* image;
  const void* in_u8;
  void* out_u8;
  void* out_f32;
  int64_t T, H, W, Ho, Wo, oy, ox;
} pp_frames_from_image_params;
int32_t pp_frames_from_image(void* stream, const pp_frames_from_image_params* p);

typedef struct {
  const void* in;
  void* out;
  int64_t total;
} pp_image_from_u8_params;
int32_t pp_image_from_u8(void* stream, const pp_image_from_u8_params* p);

/* ------------------------------------------------------------------------------------
 * pp_mask_dilate -- read_masks (utils/image_utils.py:142-175): convert_mask_to_frames'
 * u8 = trunc(clamp(m*255, 0, 255)) for a float MASK (dtype PP_F32; PP_U8 input is used as is),
 * then scipy.ndimage.binary_dilation(arr, iterations) = "a non-zero pixel within L1 distance
 * <= iterations" (border 0); iterations == 0 is binary_mask (arr > 0.1).  in [N][H][W],
 * out u8 {0,1} [N][H][W], scratch u8 [N][H][W] (caller owned).
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;
  int32_t iterations;
  const void* in;
  void* out;
  void* scratch;
  int64_t N, H, W;
} pp_mask_dilate_params;
int32_t pp_mask_dilate(void* stream, const pp_mask_dilate_params* p);

/* ------------------------------------------------------------------------------------
 * pp_clip_masks -- the binary planes InpaintGenerator.forward derives from the masks
 * (propainter.py:409-428): maskpair f16 [T][H/4][W/4][8] = (nearest x1/4 of m_in, of m_upd, 0..)
 * and tokmask u8 [T][fh][fw] = MaxPool2d(7, 3, 3) of the 1/4-res m_in plane.
 * pp_window_flags -- sparse_transformer.py:321-326: flags i32 [ceil(fh/wh)*ceil(fw/ww)], 1 iff any
 * token mask of frames [g0, g0+lt) falls inside the window.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  const void* m_in;  /* u8 [T][H][W] */
  const void* m_upd; /* u8 [T][H][W] */
  void* maskpair;
  void* tokmask;
  int64_t T, H, W, fh, fw;
  int32_t dtype; /* maskpair: PP_F16 or PP_F32 */
} pp_clip_masks_params;
int32_t pp_clip_masks(void* stream, const pp_clip_masks_params* p);

typedef struct {
  const void* tokmask; /* u8 [T][fh][fw] */
  void* flags;
  int64_t T, fh, fw, g0, lt, wh, ww;
} pp_window_flags_params;
int32_t pp_window_flags(void* stream, const pp_window_flags_params* p);

#ifdef __cplusplus
}
#endif
#endif /* PROPAINTER_MI355_H_ */
