// pp_host.h -- host-side helpers shared by the C-ABI entry points.
#pragma once
#include <stdint.h>
#include <string.h>

#include <atomic>

#include "../../include/propainter_mi355.h"

namespace pp {
// records the message for pp_last_error() (thread-local) and returns `code`
int pp_fail(int code, const char* msg);
// returns PP_OK or PP_ERR_LAUNCH after a kernel launch (PP_ERR_UNSUPPORTED when pp_blocks_1d() saw a launch of 2^32 threads
// or more since the last check)
int pp_check_launch(const char* what);
// 256-thread blocks of a 1-D elementwise launch over `total` work items.  HIP launches a grid as a 32-bit global size: a
// launch of 2^32 threads or more is silently TRUNCATED to total mod 2^32 (r04: the per-element fp32 upsample of flow completion's
// decoder at 160 images of 720x1280x32 -- 4.7e9 threads -- wrote only its first 14.6 images).  Such a launch is recorded here
// and reported by the pp_check_launch() that follows it: the entry point fails instead of returning garbage.
unsigned pp_blocks_1d(int64_t total);
// raises the dynamic-LDS limit of `func` when a launch needs more than the default.  The attribute belongs to the (function,
// device) pair: `mask` is the call site's record of the devices already served (one static word per kernel instantiation), so a
// process that drives several GPUs (distributed.run_multi_device: one thread per device) sets it on each of them, once.
void pp_allow_big_lds(const void* func, size_t bytes, std::atomic<unsigned long long>* mask);
}  // namespace pp
// once per (kernel instantiation, device); thread-safe
#define PP_ALLOW_BIG_LDS(func, bytes)                                                         \
  do {                                                                                        \
    static std::atomic<unsigned long long> pp_lds_mask_{0ull};                                \
    ::pp::pp_allow_big_lds(reinterpret_cast<const void*>(func), (bytes), &pp_lds_mask_);      \
  } while (0)
