// pp_host.h -- host-side helpers shared by the C-ABI entry points.
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/propainter_mi355.h"

namespace pp {
// records the message for pp_last_error() (thread-local) and returns `code`
int pp_fail(int code, const char* msg);
// returns PP_OK or PP_ERR_LAUNCH after a kernel launch
int pp_check_launch(const char* what);
// raises the dynamic-LDS limit of `func` when a launch needs more than the default
void pp_allow_big_lds(const void* func, size_t bytes);
}  // namespace pp
