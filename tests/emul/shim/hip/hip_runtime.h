// TEST INFRASTRUCTURE ONLY: the few HIP names csrc/device_common.h and csrc/walk_core.h use, for a host build in which a
// "wave" is one lane (tests/emul/walk_emul.cpp).  Nothing in the product includes this.
#ifndef MFA_EMUL_HIP_SHIM_H
#define MFA_EMUL_HIP_SHIM_H
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __restrict__
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline int __any(int p) { return p != 0; }
static inline int __all(int p) { return p != 0; }
static inline unsigned long long __ballot(int p) { return p ? 1ull : 0ull; }
template <class T> static inline T __shfl(T v, int) { return v; }
#endif
