// Probe (development): does hipStreamWaitValue32 hold back a kernel on stream B until a kernel on stream A has raised a counter in signal
// memory with device atomics?  And what does the release cost?   hipcc --offload-arch=gfx950 -o wait_value wait_value.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(uint32_t* sig, unsigned long long* t, uint32_t spin_us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(10);      // wall_clock64: 100 MHz
    if (threadIdx.x == 0) {
        t[blockIdx.x] = wall_clock64();
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void consumer(unsigned long long* t, const uint32_t* sig, uint32_t* seen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = wall_clock64(); *seen = __hip_atomic_load(sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    uint32_t* sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long *tp, *tc; uint32_t* seen;
    CK(hipMalloc(&tp, 8 * 64)); CK(hipMalloc(&tc, 8)); CK(hipMalloc(&seen, 4));
    for (int round = 0; round < 4; round++) {
        const uint32_t blocks = 8, need = round == 3 ? 4 : blocks;      // last round: released when half of the producers are done
        CK(hipMemsetAsync(sig, 0, 8, a));
        CK(hipStreamSynchronize(a));
        hipLaunchKernelGGL(producer, dim3(blocks), dim3(64), 0, a, sig, tp, 500u * (round + 1));
        CK(hipStreamWaitValue32(b, sig, need, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, tc, sig, seen);
        CK(hipStreamSynchronize(b)); CK(hipStreamSynchronize(a));
        unsigned long long hp[8], hc; uint32_t hs;
        CK(hipMemcpy(hp, tp, sizeof hp, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, tc, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost));
        unsigned long long last = 0, first = ~0ull;
        for (int k = 0; k < 8; k++) { if (hp[k] > last) last = hp[k]; if (hp[k] < first) first = hp[k]; }
        printf("round %d: producers spin %u us, wait for >= %u: consumer saw %u, started %.1f us after the first producer's add, %.1f us after the last\n", round, 500u * (round + 1), need, hs,
               ((double)hc - (double)first) / 100.0, ((double)hc - (double)last) / 100.0);
    }
    return 0;
}
