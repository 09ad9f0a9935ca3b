// pp_api.hip -- library-level entry points of libpropainter_mi355 (version, errors, ABI self-check).
#include "pp_device.h"
#include "pp_host.h"
#include "pp_options.h"

#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

namespace pp {

static thread_local char g_err[512] = {0};

int pp_fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

static thread_local bool g_grid_overflow = false;

unsigned pp_blocks_1d(int64_t total) {
  const int64_t blocks = (total + 255) / 256;
  if (blocks * 256 >= ((int64_t)1 << 32)) g_grid_overflow = true;
  return (unsigned)blocks;
}

int pp_check_launch(const char* what) {
  if (g_grid_overflow) {
    g_grid_overflow = false;
    snprintf(g_err, sizeof(g_err), "%s: the launch needs 2^32 threads or more (HIP truncates such a grid): tensor too large for this kernel", what);
#ifndef PP_EMU
    (void)hipGetLastError();
#endif
    return PP_ERR_UNSUPPORTED;
  }
#ifndef PP_EMU
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return PP_ERR_LAUNCH;
  }
#else
  (void)what;
#endif
  return PP_OK;
}

void pp_allow_big_lds(const void* func, size_t bytes, std::atomic<unsigned long long>* mask) {
#ifndef PP_EMU
  if (bytes <= 48 * 1024) return;
  int dev = 0;
  hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask->load(std::memory_order_acquire) & bit) return;
  hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  mask->fetch_or(bit, std::memory_order_release);
#else
  (void)func;
  (void)bytes;
  (void)mask;
#endif
}

// The knobs are read ONCE (first use, under std::call_once: two host threads may issue their first launch concurrently) and
// again only by pp_reload_options().  A reload publishes a fresh immutable snapshot; readers keep whichever snapshot they saw.
static std::atomic<const Options*> g_options{nullptr};
static std::once_flag g_options_once;
static std::mutex g_options_mutex;

static int tri(const char* name) {  // unset -> 1 (auto), "0..." -> 0, "f..." -> 2
  const char* e = getenv(name);
  return !e ? 1 : (e[0] == '0' ? 0 : (e[0] == 'f' ? 2 : 1));
}

static void load_options() {
  Options o;
  o.halo = tri("PP_CONV_HALO");
  o.halo_ct = tri("PP_CONV_HALO_CT") == 0 ? 0 : 1;
  o.ksplit = tri("PP_CONV_KSPLIT");
  o.ksplit_nst = 3;
  if (const char* e = getenv("PP_CONV_KSPLIT_NST")) o.ksplit_nst = e[0] == '4' ? 4 : 3;
  o.direct = tri("PP_CONV_DIRECT");
  o.trace = getenv("PP_CONV_TRACE") != nullptr;
  o.conv_order = 1;
  if (const char* e = getenv("PP_CONV_ORDER")) o.conv_order = e[0] != 'l';
  o.tile = 0;
  if (const char* e = getenv("PP_CONV_TILE")) {
    if (e[0] == 'x') o.tile = strcmp(e, "xlforce") == 0 ? 4 : 0;
    else if (e[0] == 't') o.tile = 5;
    else if (e[0] == 'c') o.tile = 6;
    else o.tile = e[0] == 'l' ? 1 : (e[0] == 's' ? 2 : 0);
  }
  o.halo_c64 = 0;
  if (const char* e = getenv("PP_CONV_HALO_C64")) o.halo_c64 = e[0] == '1';
  o.small_halo = 1;
  if (const char* e = getenv("PP_CONV_SMALL_HALO")) o.small_halo = e[0] != '0';
  o.gemm = tri("PP_CONV_GEMM");
  o.gemm_cfg = 0;
  if (const char* e = getenv("PP_CONV_GEMM_CFG")) o.gemm_cfg = atoi(e);
  o.epi_lds = 1;
  if (const char* e = getenv("PP_CONV_EPI")) o.epi_lds = e[0] != 'd';
  o.epi_oct = 1;
  if (const char* e = getenv("PP_CONV_EPI_OCT")) o.epi_oct = e[0] != '0';
  o.halo_min_cout = 33;
  if (const char* e = getenv("PP_CONV_HALO_MINCOUT")) o.halo_min_cout = atoi(e) > 0 ? atoi(e) : 33;
  o.upsample_b4 = 1;
  if (const char* e = getenv("PP_UPSAMPLE_B4")) o.upsample_b4 = e[0] != '0';
  o.deform_xcd = 1;
  if (const char* e = getenv("PP_DEFORM_XCD")) o.deform_xcd = e[0] != '0';
  std::lock_guard<std::mutex> lock(g_options_mutex);
  g_options.store(new Options(o), std::memory_order_release);  // old snapshots stay valid for readers (a few bytes per reload)
}

const Options& options() {
  std::call_once(g_options_once, load_options);
  return *g_options.load(std::memory_order_acquire);
}

}  // namespace pp

extern "C" void pp_reload_options(void) { pp::load_options(); }

extern "C" int32_t pp_version(void) { return PP_ABI_VERSION; }

extern "C" const char* pp_last_error(void) { return pp::g_err; }

#define PP_SIZEOF_CASE(T) \
  if (strcmp(name, #T) == 0) return (int64_t)sizeof(T);

extern "C" int64_t pp_struct_size(const char* name) {
  if (!name) return -1;
  PP_SIZEOF_CASE(pp_conv2d_params)
  PP_SIZEOF_CASE(pp_im2col_params)
  PP_SIZEOF_CASE(pp_split_pack_params)
  PP_SIZEOF_CASE(pp_instnorm_params)
  PP_SIZEOF_CASE(pp_avgpool2x2_params)
  PP_SIZEOF_CASE(pp_corr_lookup_params)
  PP_SIZEOF_CASE(pp_convex_upsample_params)
  PP_SIZEOF_CASE(pp_deform_cols_params)
  PP_SIZEOF_CASE(pp_upsample2x_params)
  PP_SIZEOF_CASE(pp_rfc_prep_params)
  PP_SIZEOF_CASE(pp_flow_combine_params)
  PP_SIZEOF_CASE(pp_img_prop_step_params)
  PP_SIZEOF_CASE(pp_pack_encoder_input_params)
  PP_SIZEOF_CASE(pp_flow_down4_params)
  PP_SIZEOF_CASE(pp_featprop_aux_params)
  PP_SIZEOF_CASE(pp_flow_warp_params)
  PP_SIZEOF_CASE(pp_layernorm_params)
  PP_SIZEOF_CASE(pp_pool_tokens_params)
  PP_SIZEOF_CASE(pp_window_attention_params)
  PP_SIZEOF_CASE(pp_fold_params)
  PP_SIZEOF_CASE(pp_unfold_gelu_params)
  PP_SIZEOF_CASE(pp_compose_u8_params)
  PP_SIZEOF_CASE(pp_frames_from_image_params)
  PP_SIZEOF_CASE(pp_image_from_u8_params)
  PP_SIZEOF_CASE(pp_mask_dilate_params)
  PP_SIZEOF_CASE(pp_clip_masks_params)
  PP_SIZEOF_CASE(pp_window_flags_params)
  return -1;
}
