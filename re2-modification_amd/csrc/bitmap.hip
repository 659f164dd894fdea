// Result bitmaps: what a rank sends when the results of a sharded batch are gathered (one process per GPU, strings are independent: the only
// exchange of the path is this gather -- mfa_amd/sharding.py, bench.py).  bit k % 8 of byte k / 8 = string k was accepted (result code 1;
// 0 = rejected and 2 = not matched, longer than the device limit, both leave the bit clear).  One kernel instead of the five a tensor
// library makes of "compare, pad, reshape, weigh, sum": behind every step of a batch, the launches are what it costs.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "mfa_internal.h"

namespace mfa {

// bit j set iff byte j of x is 1
__device__ __forceinline__ uint32_t ones8(uint64_t x) {
    const uint64_t d = x ^ 0x0101010101010101ull;                                                  // zero where the byte is 1
    const uint64_t nz = (((d & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | d) & 0x8080808080808080ull;      // bit 7 of every non-zero byte
    return (uint32_t)((((nz ^ 0x8080808080808080ull) >> 7) * 0x0102040810204080ull) >> 56);
}

__global__ void __launch_bounds__(256) pack_bitmap_kernel(const uint8_t* __restrict__ results, uint64_t n, uint8_t* __restrict__ bitmap, uint32_t aligned) {
    const uint64_t nbytes = (n + 7u) >> 3;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t at = b << 3;
        uint64_t x = 0;
        if (aligned && at + 8u <= n) x = *reinterpret_cast<const uint64_t*>(results + at);
        else
            for (uint32_t j = 0; j < 8u && at + j < n; j++) x |= (uint64_t)results[at + j] << (8u * j);
        bitmap[b] = (uint8_t)ones8(x);
    }
}

}  // namespace mfa

extern "C" int mfa_pack_result_bitmap(const uint8_t* d_results, uint64_t n, uint8_t* d_bitmap, void* stream) {
    if ((!d_results || !d_bitmap) && n) return MFA_ERR_INVALID_ARG;
    if (n == 0) return MFA_OK;
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0) return MFA_ERR_NO_DEVICE;      // (no CPU path, like every match entry point)
    const uint64_t nbytes = (n + 7u) >> 3;
    const unsigned grid = (unsigned)(nbytes + 255u) / 256u > 4096u ? 4096u : (unsigned)((nbytes + 255u) / 256u);
    hipLaunchKernelGGL(mfa::pack_bitmap_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_results, n, d_bitmap,
                       ((uintptr_t)d_results & 7u) == 0u ? 1u : 0u);
    if (hipGetLastError() != hipSuccess) return MFA_ERR_HIP;
    return MFA_OK;
}
