// micro-benchmark: do the matrix core and the vector ALU overlap across two waves of one SIMD?
// 8 waves per workgroup = 2 per SIMD.  Role per wave: 0 = gate math (VALU + transcendentals), 1 = MFMA chain, 2 = idle.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../matdeeplearn_amd/csrc/mdl_common.h"
using namespace mdl;
typedef __attribute__((ext_vector_type(8))) short bf16x8_;
__global__ __launch_bounds__(512) void k(long long* out, float* sink, int iters, int roleA, int roleB) {
    typedef Gate<true> GT;
    const int wave = threadIdx.x >> 6;                 // waves 0..3 -> SIMD 0..3 (first wave of each SIMD), 4..7 second
    const int role = (wave < 4) ? roleA : roleB;
    float f[16], s[16];
    for (int r = 0; r < 16; ++r) { f[r] = threadIdx.x * 0.01f - 1.0f + r * 0.1f; s[r] = 0.5f - r * 0.07f; }
    f32x16 acc = {0}, acc2 = {0};
    bf16x8_ a = {1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    long long t0 = clock64();
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = GT::sigmoid(f[r]) * GT::softplus_u(s[r]) + s[r];
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        }
    } else if (role == 3) {                      // two independent accumulators, alternating
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc2, 0, 0, 0);
            }
        }
    } else if (role == 4) {                      // dependent chain with the wave parked between issues
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
                __builtin_amdgcn_s_sleep(1);
            }
        }
    } else if (role == 6) {                      // ONE wave: gate of 16 elements and 12 dependent MFMAs, left to the compiler
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = GT::sigmoid(f[r]) * GT::softplus_u(s[r]) + s[r];
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        }
    } else if (role == 7) {                      // same work, schedule pinned: one MFMA, then a slice of the gate's VALU/trans ops
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = GT::sigmoid(f[r]) * GT::softplus_u(s[r]) + s[r];
#pragma unroll
            for (int q = 0; q < 12; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);     // 9 VALU
                __builtin_amdgcn_sched_group_barrier(0x400, 6, 0);     // 6 transcendental
            }
        }
    } else if (role == 5) {                      // 16x16x32 MFMAs (4 passes), dependent chain, same FLOPs: 24 of them
        typedef __attribute__((ext_vector_type(4))) float f32x4_;
        f32x4_ c4 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 24; ++q) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, c4, 0, 0, 0);
        }
        acc[0] += c4[0];
    }
    long long t1 = clock64();
    float z = acc[0] + acc2[1];
    for (int r = 0; r < 16; ++r) z += f[r];
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = z;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}
int main() {
    long long* d; float* sink; (void)hipMalloc(&d, 64); (void)hipMalloc(&sink, 1 << 20);
    const int iters = 2000;
    const char* names[] = {"gate x16", "12 mfma", "idle", "mfma 2acc", "mfma+sleep", "24 mfma16", "gate;mfma", "gate|mfma"};
    const int pairs[][2] = {{0, 2}, {1, 2}, {0, 1}, {6, 2}, {7, 2}, {6, 6}, {7, 7}};
    for (auto& pr : pairs) {
        const int ra = pr[0], rb = pr[1];
        {
            k<<<1, 512>>>(d, sink, iters, ra, rb); (void)hipDeviceSynchronize();
            k<<<1, 512>>>(d, sink, iters, ra, rb); (void)hipDeviceSynchronize();
            long long h[8]; (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("wave A: %-9s wave B: %-9s -> A %7.1f cycles/iter   B %7.1f cycles/iter\n", names[ra], names[rb],
                   (double)h[0] / iters, (double)h[4] / iters);
        }
    }
    return 0;
}
