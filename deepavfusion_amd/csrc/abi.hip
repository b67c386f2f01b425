#include "common.h"
#include "dav_kernels.h"
extern "C" int dav_abi_version(void) { return DAV_ABI_VERSION; }
extern "C" const char* dav_last_error_string(void) { return hipGetErrorString((hipError_t)dav_last_hip_error); }
