// rrl_host.hpp -- host-side helpers shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rrl_hip.h"

namespace rrl_host {

extern thread_local int last_hip_error;

constexpr int kBlock = 256;
constexpr int kMaxGrid = 2048;  // 256 CUs x 8 workgroups; the rest is grid-strided

inline int grid_for(int64_t n, int per_block = kBlock) {
    int64_t g = (n + per_block - 1) / per_block;
    return int(g < 1 ? 1 : (g > kMaxGrid ? kMaxGrid : g));
}

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        last_hip_error = int(e);
        return RRL_ELAUNCH;
    }
    return RRL_OK;
}

}  // namespace rrl_host
