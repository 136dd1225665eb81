// Exhaustive check of gcfr::sqrt_rn_normal against __builtin_sqrtf (tests/test_gpu_sqrt_rn.py).
#include "../../geomconsistentfr_amd/csrc/gcfr_device.hpp"

#include <cstdio>

__global__ void check_kernel(unsigned lo, unsigned long long count, unsigned long long *bad, unsigned *first_bad)
{
    const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long mine = 0;
    for (unsigned long long i = i0; i < count; i += stride) {
        const unsigned bits = lo + (unsigned)i;
        const float x = __builtin_bit_cast(float, bits);
        const float a = gcfr::sqrt_rn_normal(x), b = __builtin_sqrtf(x);
        const unsigned ab = __builtin_bit_cast(unsigned, a), bb = __builtin_bit_cast(unsigned, b);
        const bool both_nan = (a != a) && (b != b);
        if (ab != bb && !both_nan) {
            ++mine;
            atomicMin(first_bad, bits);
        }
    }
    if (mine)
        atomicAdd(bad, mine);
}

int main()
{
    // positive floats from 2^-96 (0x0f800000) up to and including +inf (0x7f800000), then the positive NaNs up to 0x7fffffff
    const unsigned lo = 0x0f800000u;
    const unsigned long long count = 0x80000000ull - lo;
    unsigned long long *bad;
    unsigned *first_bad;
    if (hipMalloc(&bad, 8) != hipSuccess || hipMalloc(&first_bad, 4) != hipSuccess)
        return 2;
    (void)hipMemset(bad, 0, 8);
    (void)hipMemset(first_bad, 0xff, 4);
    hipLaunchKernelGGL(check_kernel, dim3(4096), dim3(256), 0, nullptr, lo, count, bad, first_bad);
    unsigned long long h_bad = 0;
    unsigned h_first = 0;
    if (hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(&h_first, first_bad, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return 2;
    std::printf("checked %.3f G bit patterns, mismatches %llu, first 0x%08x\n", (double)count / 1e9, h_bad, h_first);
    return h_bad == 0 ? 0 : 1;
}
