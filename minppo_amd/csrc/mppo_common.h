// Shared host-side helpers: error reporting and launch checking.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include <cstdlib>

#include "../../include/minppo_hip.h"

// Measurement switches (A/B experiments that skip work, add launches or change the launch structure) exist only in variant libraries built
// with -DMPPO_EXPERIMENTS (tools/build_variant.sh); the product library does not read them - not even their names are in it
// (tests/test_abi.py lists the environment variables the shipped library may read).
#ifdef MPPO_EXPERIMENTS
#define MPPO_EXPERIMENT_ENV(name) getenv(name)
#else
#define MPPO_EXPERIMENT_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace mppo {

char* last_error_buf();  // thread-local, 512 bytes

inline int32_t fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MPPO_CHECK_HIP(expr)                                                                       \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return ::mppo::fail(MPPO_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define MPPO_CHECK_LAUNCH(name)                                                                    \
  do {                                                                                             \
    hipError_t _e = hipGetLastError();                                                             \
    if (_e != hipSuccess) return ::mppo::fail(MPPO_EHIP, "launch of %s failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

#define MPPO_REQUIRE(cond, ...)                                   \
  do {                                                            \
    if (!(cond)) return ::mppo::fail(MPPO_EINVAL, __VA_ARGS__);   \
  } while (0)

#define MPPO_TRY(expr)            \
  do {                            \
    int32_t _r = (expr);          \
    if (_r != MPPO_OK) return _r; \
  } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace mppo
