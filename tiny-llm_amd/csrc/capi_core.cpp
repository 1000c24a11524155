// Error plumbing and library-level entry points of the C ABI (include/tinyllm_hip.h).
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/tinyllm_hip.h"

namespace tl {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

}  // namespace tl

extern "C" const char *tl_last_error(void) { return tl::g_last_error.c_str(); }

extern "C" int tl_abi_version(void) { return 1; }

// reference: load_library registers the .metallib (src/extensions_ref/src/utils.cpp:9-14).
// Here the gfx950 code objects are linked into this .so; we only verify a device exists.
extern "C" int tl_load_library(const char *path) {
    (void)path;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        return tl::fail(TL_ERR_HIP, "tl_load_library: no HIP device visible (the extension is GPU-only)");
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, 0);
    if (e != hipSuccess) return tl::fail(TL_ERR_HIP, std::string("tl_load_library: ") + hipGetErrorString(e));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        return tl::fail(TL_ERR_UNSUPPORTED,
                        std::string("tl_load_library: kernels are built for gfx950, found ") + prop.gcnArchName);
    }
    return TL_OK;
}
