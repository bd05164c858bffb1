// Shared host-side helpers for libdeephisto_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/deephisto_hip.h"
#include "../../include/deephisto_hip_debug.h"   // test hooks: exported, unversioned

namespace dh {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Run-time switches: the table of env_knobs.h is the only way the library reads the environment (besides DH_RCCL_LIB, a path).
// env_int: the variable's value, or the knob's default when unset or unparsable (then recorded: env_check fails from then on).
// env_check: validates EVERY knob of the table now; DH_EINVAL + dh_last_error() naming the first bad variable.  Called by the
// create entry points, so a mistyped value stops a run before its first kernel instead of being atoi'd into something else.
int env_int(const char* name);
int env_check();

}  // namespace dh

#define DH_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      dh::set_error(__VA_ARGS__);        \
      return DH_EINVAL;                  \
    }                                    \
  } while (0)

#define DH_HIP(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      dh::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return e_ == hipErrorOutOfMemory ? DH_ENOMEM : DH_EHIP;                     \
    }                                                                             \
  } while (0)

#define DH_LAUNCH_CHECK()                                                         \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      dh::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return DH_EHIP;                                                             \
    }                                                                             \
  } while (0)
