// Error plumbing shared by the translation units behind include/qinco_hip.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/qinco_hip.h"

namespace qinco {
// records the message for qinco_last_error() (thread-local) and returns `code`
int abi_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace qinco

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess)                                                                               \
      return qinco::abi_fail(QINCO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
