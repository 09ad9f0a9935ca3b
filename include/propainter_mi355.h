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

#define PP_ABI_VERSION 1

enum pp_dtype { PP_F32 = 0, PP_F16 = 1, PP_U8 = 2, PP_I32 = 3 };

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

/* ------------------------------------------------------------------------------------
 * pp_conv2d -- implicit-GEMM convolution on MFMA (f16 inputs: 16x16x32, f32 inputs:
 * 16x16x4 exact-f32), fp32 accumulate, fused bias/activation/epilogue.
 * Replaces every torch.nn.Conv2d / Conv3d(1,k,k) / Conv3d(3,1,1) / Linear / matmul on
 * the hot path: RAFT extractor.py:170-193, update.py:6-154, corr.py:52-60 (volume as a
 * batched 1x1), recurrent_flow_completion.py:17-26,69-75,162-300, propainter.py:48-57,
 * 100-110,238-275,304-312, sparse_transformer.py:16,33,47,53,65,83-84,162-171.
 *
 * Input = channel-concatenation of up to PP_MAX_SEG channels-last tensors of the
 * same [N][H][W] (segment s contributes in_C[s] channels, read with pitch in_ldc[s]).
 * Weights are pre-packed by the host (weights.py: pack_conv_weight):
 *   w[z][cout][tap = ky*kw+kx][seg][c padded to a multiple of 32], dtype = `dtype`.
 * gridDim.z = Z selects a group (grouped conv) or a batch item (batched GEMM):
 * every pointer advances by its *_zoff (in elements) per z.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype;      /* PP_F32 or PP_F16: inputs and weights */
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
} pp_conv2d_params;

int32_t pp_conv2d(void* stream, const pp_conv2d_params* p);

#ifdef __cplusplus
}
#endif
#endif /* PROPAINTER_MI355_H_ */
