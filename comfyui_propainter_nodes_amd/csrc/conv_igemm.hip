// conv_igemm.hip -- implicit-GEMM convolution for gfx950 MFMA (see include/propainter_mi355.h).
// pp_conv2d: parameter validation and dispatch over the kernel families.  The kernels live in conv_igemm_kernel.h (flat
// tiles; instantiated per dtype pair in conv_igemm_hh / _hf / _ff / _fh.hip), conv_halo*.hip, conv_split.hip, conv_ksplit.hip and
// conv_direct.hip.
#include "conv_common.h"

#include <stdio.h>

namespace pp {

static int fail2(int code, const char* who, const char* what) {
  char msg[192];
  snprintf(msg, sizeof(msg), "%s: %s", who, what);
  return pp_fail(code, msg);
}

// pp_conv2d_params -> kernel-argument block, with the argument checks every entry point that takes the block shares
// (`who` prefixes the error text; `virtual_input`: the input segments describe a tensor that is never materialised -- the
// sampled columns of pp_deform_conv -- so their pointers are not looked at)
int convk_from_params(const pp_conv2d_params* p, ConvK* kp, const char* who, bool virtual_input) {
  ConvK& k = *kp;
  if (!p) return fail2(PP_ERR_BAD_ARG, who, "null params");
  if (p->nseg < 1 || p->nseg > PP_MAX_SEG) return fail2(PP_ERR_BAD_ARG, who, "nseg out of range");
  if (p->dtype != PP_F32 && p->dtype != PP_F16 && p->dtype != PP_F32X2)
    return fail2(PP_ERR_UNSUPPORTED, who, "dtype");
  if (p->out_dtype != PP_F32 && p->out_dtype != PP_F16) return fail2(PP_ERR_UNSUPPORTED, who, "out_dtype");
  if (!p->weight || !p->out) return fail2(PP_ERR_BAD_ARG, who, "null weight/out");
  if (p->Z < 1 || p->Z > 65535) return fail2(PP_ERR_BAD_ARG, who, "Z out of range");
  if (p->epi != PP_EPI_NONE && !p->aux1) return fail2(PP_ERR_BAD_ARG, who, "epilogue needs aux1");
  if (p->epi == PP_EPI_GRU && !p->aux2) return fail2(PP_ERR_BAD_ARG, who, "GRU epilogue needs aux2");
  // (r06: the PP_F32X2 patch form -- flat_taps on an f32 input of 1..4 channels, conv_patch.hip -- reads its input with 4-byte
  //  loads: any channel count, pitch and 4-byte-aligned base)
  const bool patch32 = p->flat_taps != 0 && p->dtype == PP_F32X2;
  const int epp = patch32 ? 1 : p->dtype == PP_F16 ? 8 : 4;
  memset(&k, 0, sizeof(k));
  int cpt = 0;
  for (int s = 0; s < p->nseg; ++s) {
    if (!virtual_input && !p->in_ptr[s]) return fail2(PP_ERR_BAD_ARG, who, "null input segment");
    if (p->in_C[s] <= 0 || (p->in_C[s] % epp) != 0)
      return fail2(PP_ERR_BAD_ARG, who, "segment channels must be a positive multiple of 16 bytes");
    if ((p->in_ldc[s] % epp) != 0 || (p->in_zoff[s] % epp) != 0)
      return fail2(PP_ERR_BAD_ARG, who, "segment pitch / z offset must be 16-byte multiples");
    if (!virtual_input && (reinterpret_cast<uintptr_t>(p->in_ptr[s]) & (patch32 ? 3 : 15)) != 0)
      return fail2(PP_ERR_BAD_ARG, who, "segment base must be 16-byte aligned");
    k.in_ptr[s] = p->in_ptr[s];
    k.in_C[s] = (int)p->in_C[s];
    k.in_ldc[s] = (int)p->in_ldc[s];
    k.in_zoff[s] = p->in_zoff[s];
    k.seg_chunks[s] = (int)((p->in_C[s] + 31) / 32);
    cpt += k.seg_chunks[s];
  }
  if ((reinterpret_cast<uintptr_t>(p->weight) & 15) != 0)
    return fail2(PP_ERR_BAD_ARG, who, "weight must be 16-byte aligned");
  k.nseg = p->nseg;
  k.N = (int)p->N; k.H = (int)p->H; k.W = (int)p->W; k.Ho = (int)p->Ho; k.Wo = (int)p->Wo;
  k.kh = p->kh; k.kw = p->kw; k.sh = p->sh; k.sw = p->sw; k.ph = p->ph; k.pw = p->pw; k.dh = p->dh; k.dw = p->dw;
  if (k.kh < 1 || k.kw < 1 || k.sh < 1 || k.sw < 1 || k.dh < 1 || k.dw < 1)
    return fail2(PP_ERR_BAD_ARG, who, "bad kernel geometry");
  k.pad_mode = p->pad_mode;
  k.weight = p->weight; k.w_zoff = p->w_zoff;
  k.chunks_per_tap = cpt;
  k.nchunks = cpt * k.kh * k.kw;
  k.flat_taps = p->flat_taps != 0;
  if (k.flat_taps) {  // one (ky, kx, c)-ordered weight row per output channel, padded to 32 at the end only
    if ((p->dtype != PP_F16 && p->dtype != PP_F32X2) || p->nseg != 1 || p->Z != 1 || p->pad_mode == PP_PAD_REPLICATE)
      return fail2(PP_ERR_UNSUPPORTED, who, "flat_taps: f16 or PP_F32X2, one segment, Z 1, zero padding only");
    k.nchunks = (int)(((int64_t)k.kh * k.kw * p->in_C[0] + 31) / 32);
  }
  k.Kp = k.nchunks * 32;
  k.bias = reinterpret_cast<const float*>(p->bias); k.bias_zoff = p->bias_zoff;
  k.Cout = (int)p->Cout;
  k.M = p->N * p->Ho * p->Wo;
  if (k.M <= 0 || k.Cout <= 0) return fail2(PP_ERR_BAD_ARG, who, "empty problem");
  k.out = p->out; k.out_ldc = (int)p->out_ldc; k.out_zoff = p->out_zoff;
  k.act = p->act; k.act2 = p->act2; k.act_split = p->act_split;
  k.act_param = p->act_param; k.out_scale = p->out_scale;
  k.epi = p->epi;
  if (p->epi_from < 0 || (p->epi_from & 3) != 0 || (p->epi_from != 0 && p->epi_from >= p->Cout))
    return fail2(PP_ERR_BAD_ARG, who, "epi_from must be a multiple of 4 below Cout");
  k.epi_from = p->epi == PP_EPI_NONE ? 0 : (int)p->epi_from;
  k.aux1 = p->aux1; k.aux1_ldc = (int)p->aux1_ldc; k.aux1_zoff = p->aux1_zoff;
  k.aux2 = p->aux2; k.aux2_ldc = (int)p->aux2_ldc; k.aux2_zoff = p->aux2_zoff;
  k.pre_add = p->pre_add; k.pre_add_ldc = (int)p->pre_add_ldc;
  k.weight_f32 = reinterpret_cast<const float*>(p->weight_f32);
  k.tile_order = options().conv_order;
  k.epi_lds = options().epi_lds;
  k.epi_oct = options().epi_oct;
  k.many_images = p->many_images != 0;
  k.acc_scale = (p->dtype == PP_F32X2 && p->acc_scale != 0.f) ? p->acc_scale : 1.f;
  if (p->pre_add && p->Z != 1) return fail2(PP_ERR_UNSUPPORTED, who, "pre_add with Z > 1");
  return PP_OK;
}

}  // namespace pp

extern "C" int32_t pp_conv2d(void* stream, const pp_conv2d_params* p) {
  using namespace pp;
  ConvK k;
  const int bad = convk_from_params(p, &k, "pp_conv2d", false);
  if (bad != PP_OK) return bad;
  const int Z = (int)p->Z;
  if (k.flat_taps && p->dtype == PP_F32X2) {
    if (p->out_dtype != PP_F32) return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: PP_F32X2 writes f32 only");
    return launch_patch_split(stream, k, Z);
  }
  if (k.flat_taps && p->dtype != PP_F16) return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: flat_taps needs PP_F16 or PP_F32X2");
  if (k.Cout <= 4 && !k.flat_taps && p->dtype == PP_F16) {
    const int rs = launch_halo_f16_small_cout(stream, k, Z, p->out_dtype == PP_F16);
    if (rs != 1) return rs;
  }
  if (k.Cout <= 4 && !k.flat_taps) {  // 2-3 output channels on a 32-channel MFMA tile are wasted matrix work: streaming vector-ALU kernel
    const int rd = launch_direct_small_cout(stream, k, Z, p->dtype, p->out_dtype == PP_F16);
    if (rd != 1) return rd;
  }
  if (p->dtype == PP_F16) {
    if (k.flat_taps) return launch_gemm_f16_patch(stream, k, Z, p->out_dtype == PP_F16);
    const int rc = launch_ksplit_f16(stream, k, Z, p->out_dtype == PP_F16);  // small M, long K: in-work-group split K
    if (rc != 1) return rc;
    const int rh = launch_halo_f16(stream, k, Z, p->out_dtype == PP_F16);  // stride-1 multi-tap: pixel tile + halo staged once per chunk
    if (rh != 1) return rh;
    const int rg = launch_gemm_f16(stream, k, Z, p->out_dtype == PP_F16);  // 1x1 / stride 1 / no padding: plain GEMM
    if (rg != 1) return rg;
    return p->out_dtype == PP_F16 ? launch_igemm_hh(stream, k, Z) : launch_igemm_hf(stream, k, Z);
  }
  if (p->dtype == PP_F32X2) {
    if (p->out_dtype != PP_F32) return pp_fail(PP_ERR_UNSUPPORTED, "pp_conv2d: PP_F32X2 writes f32 only");
    return launch_split(stream, k, Z);
  }
  return p->out_dtype == PP_F16 ? launch_igemm_fh(stream, k, Z) : launch_igemm_ff(stream, k, Z);
}
