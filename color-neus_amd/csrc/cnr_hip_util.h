// shared error bookkeeping of the HIP translation units
#pragma once
#include <hip/hip_runtime.h>
namespace cnr {
extern hipError_t g_first_error;
extern const char* g_first_error_where;
}
#define CNR_LAUNCH_CHECK(where)                                   \
  do {                                                            \
    hipError_t e_ = hipGetLastError();                            \
    if (e_ != hipSuccess && cnr::g_first_error == hipSuccess) {   \
      cnr::g_first_error = e_;                                    \
      cnr::g_first_error_where = where;                           \
    }                                                             \
  } while (0)
