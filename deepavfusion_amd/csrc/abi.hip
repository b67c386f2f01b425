#include "common.h"
#include "dav_kernels.h"
extern "C" int dav_abi_version(void) { return DAV_ABI_VERSION; }
