// micro-benchmark: cost of v_permlane16_swap_b32 (inline asm form used by cgconv_cb.inc) and of the paired gate block
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../matdeeplearn_amd/csrc/mdl_common.h"
using namespace mdl;
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16_nonop(float& a, float& b) {
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <int MODE>
__global__ __launch_bounds__(256) void k(long long* out, float* sink, int iters) {
    typedef Gate<true> GT;
    float a[16];
    for (int r = 0; r < 16; ++r) a[r] = threadIdx.x * 0.01f - 1.0f + r * 0.1f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) swap16(a[r], a[r + 8]);
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) swap16_nonop(a[r], a[r + 8]);
        } else if (MODE == 2) {          // the kernel's block: 8 swaps, 8 gates, 8 swaps
#pragma unroll
            for (int r = 0; r < 8; ++r) swap16(a[r], a[r + 8]);
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r] = GT::sigmoid(a[r]) * GT::softplus_u(a[r + 8]); a[r + 8] = 1.0f; }
#pragma unroll
            for (int r = 0; r < 8; ++r) swap16(a[r], a[r + 8]);
        } else {                          // 8 gates only
#pragma unroll
            for (int r = 0; r < 8; ++r) { a[r] = GT::sigmoid(a[r]) * GT::softplus_u(a[r + 8]); a[r + 8] = a[r] + 0.5f; }
        }
    }
    long long t1 = clock64();
    float z = 0; for (int r = 0; r < 16; ++r) z += a[r];
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = z;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <typename F> static void run(const char* name, F launch, long long* d, int iters) {
    long long h; launch(); (void)hipDeviceSynchronize(); launch(); (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f cycles/iter\n", name, (double)h / iters);
}
int main() {
    long long* d; float* sink; (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 1 << 20);
    const int iters = 2000;
    run("8 swaps (with s_nops), 1 wave", [&] { k<0><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("8 swaps (no nops), 1 wave", [&] { k<1><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("8 swaps + 8 gates + 8 swaps, 1 wave", [&] { k<2><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("8 gates, 1 wave", [&] { k<3><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("8 swaps + 8 gates + 8 swaps, 256 WG x 4 waves", [&] { k<2><<<256, 256>>>(d, sink, iters); }, d, iters);
    run("8 swaps + 8 gates + 8 swaps, 512 WG x 4 waves", [&] { k<2><<<512, 256>>>(d, sink, iters); }, d, iters);
    run("8 gates, 512 WG x 4 waves", [&] { k<3><<<512, 256>>>(d, sink, iters); }, d, iters);
    return 0;
}
