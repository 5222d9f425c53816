// Error plumbing and version of librtxn.so.
#include "common.h"

#include <cstdarg>
#include <cstdio>

namespace rtxn {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail_hip(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return RTXN_ERR_HIP;
}

int require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (librtxn has no CPU fallback)");
    return RTXN_ERR_HIP;
  }
  return RTXN_OK;
}

const char* last_error() { return g_err; }

static __global__ void zero_words_kernel(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}

hipError_t zero_words(void* p, size_t n_words, hipStream_t stream) {
  if (n_words == 0) return hipSuccess;
  const size_t blocks = (n_words + 255) / 256;
  zero_words_kernel<<<(unsigned)(blocks > 1024 ? 1024 : blocks), 256, 0, stream>>>(static_cast<unsigned*>(p), n_words);
  return hipGetLastError();
}

}  // namespace rtxn

extern "C" int rtxn_version(void) { return RTXN_VERSION; }
extern "C" const char* rtxn_last_error(void) { return rtxn::last_error(); }
