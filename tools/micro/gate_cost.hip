// micro-benchmark: cycles of the fused gate math per 16 elements/lane (forward gate, backward derivative),
// alone and on a fully occupied chip (also reports the shader clock under that load from wall_clock64).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../matdeeplearn_amd/csrc/mdl_common.h"
using namespace mdl;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8_;

template <int MODE>
__global__ __launch_bounds__(256) void k_gate(long long* out, float* sink, int iters) {
    typedef Gate<true> GT;
    float f[16], s[16], d[16];
    for (int r = 0; r < 16; ++r) { f[r] = threadIdx.x * 0.01f - 1.0f + r * 0.1f; s[r] = 0.5f - r * 0.07f; d[r] = 0.3f + r; }
    f32x16 acc = {0};
    bf16x8_ a = {1, 2, 3, 4, 5, 6, 7, 8};
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = GT::sigmoid(f[r]) * GT::softplus_u(s[r]) + s[r];
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sf, sp, ss;
                GT::deriv(f[r], s[r], sf, sp, ss);
                const float t = d[r] * sf;
                f[r] = (t * GT::M_SCALE) * (1.0f - sf) * sp;
                s[r] = t * ss + 0.1f;
            }
        } else if (MODE == 2) {   // forward gate + 24 MFMAs (independent): do they overlap inside one wave?
#pragma unroll
            for (int r = 0; r < 16; ++r) f[r] = GT::sigmoid(f[r]) * GT::softplus_u(s[r]) + s[r];
#pragma unroll
            for (int k = 0; k < 24; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        } else {                  // 24 MFMAs only
#pragma unroll
            for (int k = 0; k < 24; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float z = acc[0];
    for (int r = 0; r < 16; ++r) z += f[r] + s[r];
    sink[(blockIdx.x * blockDim.x + threadIdx.x) & 0xffff] = z;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
}
template <typename F>
static void run(const char* name, F launch, long long* d, int iters) {
    long long h[2];
    launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-56s %8.1f cycles/iter   shader clock %.2f GHz\n", name, (double)h[0] / iters, (double)h[0] / (double)h[1] * 0.1);
}
int main() {
    long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 1 << 20);
    const int iters = 2000;
    run("fwd gate x16, 1 wave", [&] { k_gate<0><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("bwd deriv x16, 1 wave", [&] { k_gate<1><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("fwd gate x16 + 24 mfma, 1 wave", [&] { k_gate<2><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("24 mfma, 1 wave", [&] { k_gate<3><<<1, 64>>>(d, sink, iters); }, d, iters);
    run("fwd gate x16, 256 WG x 4 waves (1/SIMD)", [&] { k_gate<0><<<256, 256>>>(d, sink, iters); }, d, iters);
    run("fwd gate x16, 512 WG x 4 waves (2/SIMD)", [&] { k_gate<0><<<512, 256>>>(d, sink, iters); }, d, iters);
    run("bwd deriv x16, 256 WG x 4 waves", [&] { k_gate<1><<<256, 256>>>(d, sink, iters); }, d, iters);
    run("gate+24 mfma, 256 WG x 4 waves (1/SIMD)", [&] { k_gate<2><<<256, 256>>>(d, sink, iters); }, d, iters);
    run("gate+24 mfma, 512 WG x 4 waves (2/SIMD)", [&] { k_gate<2><<<512, 256>>>(d, sink, iters); }, d, iters);
    run("24 mfma, 256 WG x 4 waves", [&] { k_gate<3><<<256, 256>>>(d, sink, iters); }, d, iters);
    run("24 mfma, 512 WG x 4 waves", [&] { k_gate<3><<<512, 256>>>(d, sink, iters); }, d, iters);
    return 0;
}
