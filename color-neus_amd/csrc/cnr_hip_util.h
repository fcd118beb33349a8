// shared error bookkeeping of the HIP translation units
#pragma once
#include <hip/hip_runtime.h>
namespace cnr {
extern hipError_t g_first_error;
extern const char* g_first_error_where;
// optional per-launch timing with HIP events on the launch stream (bench / profiling only; off by default)
void timing_begin(const char* name, int kind, int nt, long P, int N, int K, int pairs, hipStream_t s, double bytes);
void timing_end(hipStream_t s);
struct TimingScope {
  hipStream_t s;
  TimingScope(const char* name, int kind, int nt, long P, int N, int K, int pairs, hipStream_t st, double bytes = 0.0) : s(st) {
    timing_begin(name, kind, nt, P, N, K, pairs, s, bytes);
  }
  ~TimingScope() { timing_end(s); }
};
}
#define CNR_LAUNCH_CHECK(where)                                   \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess && cnr::g_first_error == hipSuccess) {   \
      cnr::g_first_error = e_;                                    \
      cnr::g_first_error_where = where;                           \
    }                                                             \
  } while (0)
