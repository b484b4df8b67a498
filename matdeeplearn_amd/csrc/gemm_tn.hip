// gemm_tn.hip — tall-skinny "TN" GEMM:  C[M, K] += A[N, M]^T . B[N, K]   (bf16 in, fp32 out), N >> M, K.
// This is the weight-gradient shape of every node-level Linear in the reference models
// (dW = grad_out^T . input, e.g. pre_lin_list of /root/reference/matdeeplearn/models/cgcnn.py:64-74,124-130
// with N ~ 2e5 nodes, M = 64, K = 114): the contraction runs over the node dimension, which library
// GEMMs handle poorly (measured 0.7 ms for 64 x 114 x 2e5).  HBM bound: N*(M+K)*2 bytes read once.
//
// Workgroup = 4 waves = 128 consecutive rows per step of a grid-stride loop.  MFMA 32x32x16 with the
// ROW index as the contraction dimension: both operands are read "down the columns" (8 rows per lane,
// consecutive lanes = consecutive columns -> every load is a contiguous 64-byte run).  Wave w owns the
// 32-column tiles w, w+4, ... of C for all M; partial sums stay in registers and are flushed once per
// wave with fp32 atomics (C must be zero-filled by the caller).
#include "mdl_common.h"

namespace mdl {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_g;

template <int MT, int NTW>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda, int M,
                                                         const bf16_t* __restrict__ B, int ldb, int K,
                                                         float* __restrict__ C, int64_t N) {
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int64_t n_super = (N + 127) / 128;
    for (int64_t sc = blockIdx.x; sc < n_super; sc += gridDim.x) {
        const int64_t nb = sc * 128;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int64_t r0 = nb + 16 * ks + 8 * h;
            bf16x8 bfr[NTW];
#pragma unroll
            for (int b = 0; b < NTW; ++b) {
                const int col = (wv + 4 * b) * 32 + i;
                const int cc = min(col, K - 1);
                bf16x8 t;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int64_t n = min(r0 + q, N - 1);                       // clamp, never guard
                    const short v = (short)B[n * ldb + cc];
                    t[q] = (r0 + q < N && col < K) ? v : (short)0;
                }
                bfr[b] = t;
            }
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const int m = a * 32 + i;
                const int mc = min(m, M - 1);
                bf16x8 af;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int64_t n = min(r0 + q, N - 1);
                    const short v = (short)A[n * lda + mc];
                    af[q] = (r0 + q < N && m < M) ? v : (short)0;
                }
#pragma unroll
                for (int b = 0; b < NTW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[b], acc[a][b], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b) {
            const int col = (wv + 4 * b) * 32 + i;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = a * 32 + d_row(r, h);
                    if (m < M) unsafeAtomicAdd(C + (int64_t)m * K + col, acc[a][b][r]);
                }
            }
        }
}

}  // namespace mdl

extern "C" int mdl_gemm_tn(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c, int64_t N,
                           int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_gemm_tn: bf16 only");
    MDL_REQUIRE(M >= 1 && M <= 128 && K >= 1 && K <= 256, MDL_E_UNSUPP, "mdl_gemm_tn: need 1<=M<=128, 1<=K<=256 (got %d, %d)", M, K);
    MDL_REQUIRE(N >= 0 && lda >= M && ldb >= K && (N == 0 || (a && b && c)), MDL_E_ARG, "mdl_gemm_tn: bad arguments");
    if (N == 0) return MDL_OK;
    int64_t grid = cdiv(N, 128);
    if (grid > 512) grid = 512;
    const int mt = (M + 31) / 32, ntw = ((K + 31) / 32 + 3) / 4;
    hipStream_t st = (hipStream_t)stream;
#define MDL_TN(MT_, NTW_) hipLaunchKernelGGL((gemm_tn_kernel<MT_, NTW_>), dim3((unsigned)grid), dim3(256), 0, st, \
        (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, N)
    if (ntw == 1) { if (mt == 1) MDL_TN(1, 1); else if (mt == 2) MDL_TN(2, 1); else if (mt == 3) MDL_TN(3, 1); else MDL_TN(4, 1); }
    else { if (mt == 1) MDL_TN(1, 2); else if (mt == 2) MDL_TN(2, 2); else if (mt == 3) MDL_TN(3, 2); else MDL_TN(4, 2); }
#undef MDL_TN
    return check_launch("mdl_gemm_tn");
}
