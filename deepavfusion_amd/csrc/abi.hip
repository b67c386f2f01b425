#include "common.h"
#include "dav_kernels.h"
extern "C" int dav_abi_version(void) { return DAV_ABI_VERSION; }
extern "C" const char* dav_last_error_string(void) { return hipGetErrorString((hipError_t)dav_last_hip_error); }
// bit 0: built with -DDAV_EXPERIMENTAL (the measured-and-rejected GEMM tile configurations are present; make EXPERIMENTAL=1)
extern "C" int dav_build_flags(void) {
#ifdef DAV_EXPERIMENTAL
  return 1;
#else
  return 0;
#endif
}
