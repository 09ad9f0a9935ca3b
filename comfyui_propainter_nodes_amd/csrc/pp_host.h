// pp_host.h -- host-side helpers shared by the C-ABI entry points.
#pragma once
#include <stdint.h>
#include <string.h>

#include <atomic>

#include "../../include/propainter_mi355.h"

namespace pp {
// records the message for pp_last_error() (thread-local) and returns `code`
int pp_fail(int code, const char* msg);
// returns PP_OK or PP_ERR_LAUNCH after a kernel launch
int pp_check_launch(const char* what);
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
