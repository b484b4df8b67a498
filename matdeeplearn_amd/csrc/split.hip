// split.hip — fp32 -> (hi, lo) bf16 operand pairs for the split-product ("bf16x3", MDL_SPLIT_BF16) parity mode.
//
// v = hi + lo to 16 significant bits (both parts rounded to nearest even), so that a fp32 product a.b can run on the bf16 matrix
// core as a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation.  The CGConv kernels split their operands in registers
// (cgconv.hip, cgconv_node.hip); this kernel prepares the operands of a node-level Linear's WEIGHT GRADIENT dW = g^T x
// (contraction over the N ~ 2e5 nodes: the pre-FC layer, matdeeplearn/models/cgcnn.py:64-74,124-130 through autograd), which the
// library runs at 0.6 ms in fp32 and the TN GEMM of gemm_tn.hip at 0.03 ms per bf16 product.  HBM-bound: 4 B read + 4 B written
// per element.
#include "mdl_common.h"

namespace mdl {

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi,
                                                         bf16_t* __restrict__ lo, int64_t n2) {
    typedef __attribute__((ext_vector_type(2))) float f32x2s;
    const f32x2s* s2 = reinterpret_cast<const f32x2s*>(src);
    unsigned* h1 = reinterpret_cast<unsigned*>(hi);
    unsigned* l1 = reinterpret_cast<unsigned*>(lo);
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n2; q += (int64_t)gridDim.x * blockDim.x) {
        const f32x2s v = s2[q];
        const unsigned h = pk_bf16(v[0], v[1]);
        h1[q] = h;
        l1[q] = pk_bf16(v[0] - __builtin_bit_cast(float, h << 16), v[1] - __builtin_bit_cast(float, h & 0xffff0000u));
    }
}

}  // namespace mdl

extern "C" int mdl_split_bf16(const float* src, void* hi, void* lo, int64_t n, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(n >= 0 && (n == 0 || (src && hi && lo)), MDL_E_ARG, "mdl_split_bf16: null pointer");
    MDL_REQUIRE(n % 2 == 0 && reinterpret_cast<uintptr_t>(src) % 8 == 0 && reinterpret_cast<uintptr_t>(hi) % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(lo) % 4 == 0, MDL_E_ARG, "mdl_split_bf16: n must be even, src 8-byte, hi / lo 4-byte aligned");
    if (n == 0) return MDL_OK;
    const int64_t n2 = n / 2;
    const unsigned grid = (unsigned)std::min<int64_t>((n2 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, static_cast<bf16_t*>(hi),
                       static_cast<bf16_t*>(lo), n2);
    return check_launch("mdl_split_bf16");
}
