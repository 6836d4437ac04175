// runtime.h -- host-side runtime state shared by the translation units of libp2mt_hip.so.
// One process drives one GPU (one process per GPU); state is process-global.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/p2mt.h"

namespace p2mt {

struct Runtime {
  bool initialised = false;
  int device = 0;
  hipStream_t stream = nullptr;
  int mds = 1;      // v_dot2_u32_u16 MDS
  int partial = 0;  // spec-form partial rounds
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  char err[512] = {0};
};

Runtime& rt();
int fail_hip(hipError_t e, const char* what, const char* file, int line);
int fail(int code, const char* msg);
int ensure_init();

// RAII device scratch buffer
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t bytes) {
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
      p = nullptr;
      snprintf(rt().err, sizeof(rt().err), "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
      return P2MT_ENOMEM;
    }
    return P2MT_OK;
  }
  template <typename T>
  T* as() {
    return static_cast<T*>(p);
  }
};

}  // namespace p2mt

#define P2MT_HIP(x)                                                           \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) return p2mt::fail_hip(e_, #x, __FILE__, __LINE__);  \
  } while (0)
#define P2MT_TRY(x)          \
  do {                       \
    int rc_ = (x);           \
    if (rc_ != P2MT_OK) return rc_; \
  } while (0)
#define P2MT_LAUNCH_CHECK() P2MT_HIP(hipGetLastError())
