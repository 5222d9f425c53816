// Shared host-side helpers of librtxn.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "rtxn.h"

namespace rtxn {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail_hip(hipError_t e, const char* what);

inline hipStream_t as_stream(rtxn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// The library has no CPU path: every compute entry point goes through this.
int require_device();

// Zeroes n_words 32-bit words with a KERNEL.  Not hipMemsetAsync: as a memset node of a captured hipGraph it filled the 4 bytes
// of a loss sum with a stray byte (0xE8E8E8E8, 0x78787878, ...) on the graph's FIRST launch and with zeros from the second on
// (ROCm 7.2, round 3: tools/probe/captured_loss_dbg.py, profiles/r03/graph_memset_first_launch.txt), so nothing on a capturable
// path uses it any more.
hipError_t zero_words(void* p, size_t n_words, hipStream_t stream);

}  // namespace rtxn

#define RTXN_HIP(expr)                                           \
  do {                                                           \
    hipError_t e_ = (expr);                                      \
    if (e_ != hipSuccess) return ::rtxn::fail_hip(e_, #expr);    \
  } while (0)

#define RTXN_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      ::rtxn::set_error(__VA_ARGS__);      \
      return RTXN_ERR_INVALID;             \
    }                                      \
  } while (0)

#define RTXN_LAUNCH_CHECK(name)                                   \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess) return ::rtxn::fail_hip(e_, name);      \
  } while (0)

#define RTXN_DEVICE_OR_FAIL()                        \
  do {                                               \
    int rc_ = ::rtxn::require_device();              \
    if (rc_ != RTXN_OK) return rc_;                  \
  } while (0)
