// pp_options.h -- the library's tuning / test knobs, read from the environment ONCE (first use, thread-safe) and FROZEN from then
// on: a later os.environ change is ignored until pp_reload_options() is called (lib.reload_options(); tests and tools that switch a
// knob inside one process).  Every knob pins a kernel family for
// tests or A/B runs; none changes results beyond the summation order of the selected kernel.
//   PP_CONV_HALO     0 | force   halo-tile kernels off / for every eligible geometry regardless of the problem size
//   PP_CONV_HALO_CT  0           f16: the runtime-tap halo kernel everywhere; PP_F32X2: the flat kernel instead of the
//                                compile-time-tap halo kernels (the runtime-tap PP_F32X2 halo kernel lives in tools/experiments)
//   PP_CONV_KSPLIT   0 | force   in-work-group split-K kernel off / for every f16 problem with >= 4 chunks
//   PP_CONV_KSPLIT_NST 3 | 4     ring stages per K group of the split-K kernel (default 3; 4 = three chunk copies per group in flight, 160 KB:
//                                measured r04: 8.1 vs 7.9 ms on the 470 128 -> 128 step convolutions, no gain -- the chain is not copy-bound)
//   PP_CONV_TILE     large | small | xlforce | tiny | classic   pin one flat-tile family
//   PP_CONV_DIRECT   0 | force   <= 4-output-channel streaming kernel off / regardless of the image size
//   PP_CONV_ORDER    launch      flat-tile kernels: work-groups in launch order (pixel tiles first) instead of XCD-contiguous,
//                                channel-tile-adjacent order (conv_common.h: flat_tile_of)
//   PP_CONV_SMALL_HALO 0         <= 4-output-channel 3x3 f16 layers on the vector-ALU kernel of conv_direct.hip instead of 16-channel halo
//                                MFMA tiles (r04 default: the generator's 64 -> 3 output layer 300 -> 186 us per launch)
//   PP_CONV_HALO_C64 1           f16 compile-time-tap halo layers on 64-channel tiles whatever Cout: 48 KB and 100 registers per work-group,
//                                THREE work-groups per CU instead of two (72 KB, 152-175 registers) -- the occupancy A/B the r04 SQ counters
//                                ask for (profiles/r04_conv_counters.md); default 0 until measured
//   PP_CONV_GEMM     0 | force   the GEMM kernel for 1x1 f16 layers (conv_gemm_f16.hip) off / for every eligible layer whatever its size
//   PP_CONV_GEMM_CFG 1..5        pin one tile configuration of that kernel (tuning; conv_gemm_f16.hip: launch_gemm_t)
//   PP_CONV_TRACE    (set)       print which convolution kernel family ran (debugging aid)
//   PP_CONV_EPI      direct      convolution epilogues store their quads as the MFMA leaves them (r01-r04) instead of transposing them
//                                through LDS into whole-cache-line rows (r05 default; bit-identical)
//   PP_CONV_EPI_OCT  0           the GEMM kernel stores its f16 outputs as 8-byte quads (r01-r05) instead of pairing channel-adjacent quads
//                                into 16-byte stores with v_permlane16_swap (r06 default; bit-identical)
//   PP_CONV_HALO_MINCOUT n       f16 halo-tile kernels for layers with at least n output channels (default 33; the 17..32-channel layers
//                                of flow completion's full-resolution decoder otherwise run on the flat 32-channel tiles)
//   PP_UPSAMPLE_B4   0           pp_upsample2x with one thread per output pixel (r01-r05) instead of per 2 x 2 output block (bit-identical)
//   PP_DEFORM_XCD    0           pp_deform_cols / pp_deform_conv walk their pixel blocks in launch order instead of XCD-contiguous order
#pragma once

namespace pp {
struct Options {
  int halo;     // 0 off, 1 auto, 2 force
  int halo_ct;  // 0 runtime taps, 1 auto
  int ksplit;   // 0 off, 1 auto, 2 force
  int ksplit_nst;  // 3 or 4
  int tile;     // 0 auto, 1 large, 2 small, 4 xlforce, 5 tiny, 6 classic
  int direct;   // 0 off, 1 auto, 2 force
  int trace;
  int conv_order;  // 1 (default): XCD-contiguous, channel tiles adjacent; 0: launch order
  int small_halo;  // 1: 3x3 f16 layers with <= 4 output channels on 16-channel halo MFMA tiles instead of the vector-ALU kernel
  int halo_c64;    // 1: f16 compile-time-tap halo layers on 64-channel tiles whatever Cout (default 0)
  int gemm;        // 0 off, 1 auto, 2 force
  int gemm_cfg;    // 0 auto, 1..5 pinned
  int epi_lds;     // 1 (default): LDS-transposed epilogue; 0: direct quads
  int epi_oct;     // 1 (default): GEMM kernel, f16 outputs: paired quads, 16-byte stores (r06); 0: 8-byte quads
  int halo_min_cout;  // f16 halo-tile kernels from this many output channels on (default 33; r06 A/B: 17 = the 32-channel layers too)
  int upsample_b4; // 1 (default): pp_upsample2x computes 2 x 2 output blocks per thread (r06); 0: one output pixel per thread
  int deform_xcd;  // 1 (default): the deformable-sampling kernels walk their pixel blocks in XCD-contiguous order
};
const Options& options();
}  // namespace pp
