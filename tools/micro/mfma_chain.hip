// micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for dependent chains vs independent accumulators,
// and cost of wave64 transcendental / plain VALU instructions (one wave per SIMD and two waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void k_mfma(long long* out, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24 / NACC; ++k)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0; for (int j = 0; j < NACC; ++j) s += acc[j][0];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
template <int MODE>
__global__ void k_valu(long long* out, float* sink, int iters) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 0.001f + r;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) v[r] = __builtin_amdgcn_exp2f(v[r]);
            else if (MODE == 1) v[r] = v[r] * 1.0001f + 0.5f;
            else if (MODE == 2) v[r] = __builtin_amdgcn_rcpf(v[r]);
            else v[r] = __builtin_amdgcn_logf(v[r]);
        }
    }
    long long t1 = clock64();
    float s = 0; for (int r = 0; r < 16; ++r) s += v[r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <typename F>
static void run(const char* name, F launch, long long* d, int iters, int per) {
    long long h[2];
    launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-48s %7.1f cycles/instr\n", name, (double)h[0] / iters / per);
}
int main() {
    long long* d; float* sink; hipMalloc(&d, 64); hipMalloc(&sink, 1 << 20);
    const int iters = 1000;
    run("mfma 32x32x16 bf16, 1 acc (dependent), 1 wave", [&] { k_mfma<1><<<1, 64>>>(d, iters); }, d, iters, 24);
    run("mfma, 2 accs interleaved, 1 wave", [&] { k_mfma<2><<<1, 64>>>(d, iters); }, d, iters, 24);
    run("mfma, 4 accs interleaved, 1 wave", [&] { k_mfma<4><<<1, 64>>>(d, iters); }, d, iters, 24);
    run("mfma, 1 acc, 8 waves/CU (2 per SIMD)", [&] { k_mfma<1><<<1, 512>>>(d, iters); }, d, iters, 24);
    run("mfma, 2 accs, 8 waves/CU", [&] { k_mfma<2><<<1, 512>>>(d, iters); }, d, iters, 24);
    run("v_exp_f32, 1 wave", [&] { k_valu<0><<<1, 64>>>(d, sink, iters); }, d, iters, 16);
    run("v_fma_f32, 1 wave", [&] { k_valu<1><<<1, 64>>>(d, sink, iters); }, d, iters, 16);
    run("v_rcp_f32, 1 wave", [&] { k_valu<2><<<1, 64>>>(d, sink, iters); }, d, iters, 16);
    run("v_log_f32, 1 wave", [&] { k_valu<3><<<1, 64>>>(d, sink, iters); }, d, iters, 16);
    run("v_exp_f32, 8 waves/CU", [&] { k_valu<0><<<1, 512>>>(d, sink, iters); }, d, iters, 16);
    run("v_fma_f32, 8 waves/CU", [&] { k_valu<1><<<1, 512>>>(d, sink, iters); }, d, iters, 16);
    return 0;
}
