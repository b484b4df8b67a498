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
#include <cstdlib>

// waves per workgroup of the streaming kernel's weight-gradient forms: 8 for the shapes with a 5-tile side (register-allocated
// for one workgroup per CU: 150 x 150 on 1.5e6 rows 258 -> 233 us, 150 x 50 128 -> 125), 4 otherwise (100 x 100: 136 vs 144 us)
#define MDL_TN_NW(MT_, NT_) (((MT_) > 4 || (NT_) > 4) ? 8 : 4)

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

#include "gemm_tn_stream.inc"

}  // namespace mdl

extern "C" int mdl_gemm_tn(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c, int64_t N,
                           int dtype, mdlStream_t stream) {
    return mdl_gemm_tn_colsum(a, lda, M, b, ldb, K, c, nullptr, N, dtype, stream);
}

extern "C" int mdl_gemm_tn_colsum(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c,
                                  float* colsum, int64_t N, int dtype, mdlStream_t stream) {
    return mdl_gemm_tn_act(a, lda, M, nullptr, 0, 0, b, ldb, K, c, colsum, N, dtype, stream);
}

extern "C" size_t mdl_tn_scratch_bytes(void) { return (size_t)512 * 25 * 1024 * sizeof(float); }

extern "C" int mdl_gemm_tn_act(const void* a, int64_t lda, int M, const void* y, int64_t ldy, int act, const void* b,
                               int64_t ldb, int K, float* c, float* colsum, int64_t N, int dtype, mdlStream_t stream) {
    return mdl_gemm_tn_ex(a, lda, M, y, ldy, act, b, ldb, K, c, colsum, nullptr, N, dtype, stream);
}

extern "C" int mdl_gemm_tn_ex(const void* a, int64_t lda, int M, const void* y, int64_t ldy, int act, const void* b,
                              int64_t ldb, int K, float* c, float* colsum, void* scratch, int64_t N, int dtype, mdlStream_t stream) {
    using namespace mdl;
    // MDL_DETERMINISTIC: one workgroup — every element of c / colsum then gets ONE add from one wave (no cross-block order)
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;
    dtype &= MDL_DTYPE_MASK;
    MDL_REQUIRE(act >= 0 && act <= 2 && (act == 0 || (y && ldy >= M && ldy % 2 == 0 && reinterpret_cast<uintptr_t>(y) % 4 == 0)),
                MDL_E_ARG, "mdl_gemm_tn_act: act must be 0, 1 (relu) or 2 (shifted softplus), with the saved output y for 1 / 2");
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_gemm_tn: bf16 only");
    MDL_REQUIRE(M >= 1 && M <= 160 && K >= 1 && K <= 256 && (M <= 128 || (K <= 160 && M % 2 == 0 && K % 2 == 0)), MDL_E_UNSUPP,
                "mdl_gemm_tn: need 1<=M<=160, 1<=K<=256 (even M, K <= 160 when M > 128) (got %d, %d)", M, K);
    MDL_REQUIRE(N >= 0 && lda >= M && ldb >= K && (N == 0 || (a && b && c)), MDL_E_ARG, "mdl_gemm_tn: bad arguments");
    if (N == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    {   // streaming kernel: needs dword-addressable rows and K <= 128 (LDS budget / instantiations)
        int mt = (M + 31) / 32, nt = (K + (colsum ? 1 : 0) + 31) / 32;   // colsum rides in padding column K of the B tile
        if (mt == 3) mt = 4;                      // (instantiated for 1, 2, 4, 5 tiles; the padding columns are zero)
        if (nt == 3) nt = 4;
        const bool ok = (M % 2 == 0) && (K % 2 == 0) && (lda % 2 == 0) && (ldb % 2 == 0) && nt <= 5 &&
                        reinterpret_cast<uintptr_t>(a) % 4 == 0 && reinterpret_cast<uintptr_t>(b) % 4 == 0;
        if (ok) {
            // one block per CU: every block ends with M*K atomics on the same addresses — on 2e5 rows (64 x 114) the flush
            // costs ~10 us per 256 blocks (256 / 512 / 768 / 1024 blocks: 27 / 37 / 47 / 57 us); two per CU only where the
            // stream is long enough to pay for it (1.5e6 rows: 100 x 100 156 -> 134 us, 150 x 50 141 -> 128 us; the 5 x 5-tile
            // shapes are register-allocated for one block per CU).
            const int grid_cap = det ? 1 : ((N >= (1 << 20) && mt * nt < 25) ? 512 : 256);
            int64_t sgrid = cdiv(N, 64);
            if (sgrid > grid_cap) sgrid = grid_cap;
            // scratch: the blocks leave as plain stores and a second launch adds them (the streaming kernel's `part`); not worth a
            // launch for a handful of workgroups, not reproducible enough for the deterministic shape
            float* part = (scratch && !det && sgrid >= 32 && reinterpret_cast<uintptr_t>(scratch) % 16 == 0) ? static_cast<float*>(scratch) : nullptr;
#define MDL_TNS(MT_, NT_)                                                                                                   \
    do {                                                                                                                    \
        if (act == 0) hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 0, false, MDL_TN_NW(MT_, NT_)>), dim3((unsigned)sgrid), dim3(64 * MDL_TN_NW(MT_, NT_)), 0, st,      \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)nullptr, 0, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0, 0, (bf16_t*)nullptr, part); \
        else if (act == 1) hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 1, false, MDL_TN_NW(MT_, NT_)>), dim3((unsigned)sgrid), dim3(64 * MDL_TN_NW(MT_, NT_)), 0, st, \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)y, (int)ldy, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0, 0, (bf16_t*)nullptr, part); \
        else hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 2, false, MDL_TN_NW(MT_, NT_)>), dim3((unsigned)sgrid), dim3(64 * MDL_TN_NW(MT_, NT_)), 0, st,               \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)y, (int)ldy, (const bf16_t*)nullptr, (bf16_t*)nullptr, 0, 0, (bf16_t*)nullptr, part); \
    } while (0)
            if (mt == 1) { if (nt == 1) MDL_TNS(1, 1); else if (nt == 2) MDL_TNS(1, 2); else if (nt == 4) MDL_TNS(1, 4); else MDL_TNS(1, 5); }
            else if (mt == 2) { if (nt == 1) MDL_TNS(2, 1); else if (nt == 2) MDL_TNS(2, 2); else if (nt == 4) MDL_TNS(2, 4); else MDL_TNS(2, 5); }
            else if (mt == 4) { if (nt == 1) MDL_TNS(4, 1); else if (nt == 2) MDL_TNS(4, 2); else if (nt == 4) MDL_TNS(4, 4); else MDL_TNS(4, 5); }
            else { if (nt == 1) MDL_TNS(5, 1); else if (nt == 2) MDL_TNS(5, 2); else if (nt == 4) MDL_TNS(5, 4); else MDL_TNS(5, 5); }
#undef MDL_TNS
            if (part)
                hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(mt * nt * 4), 16), dim3(256), 0, st, part, (int)sgrid, nt, mt * nt, M, K, c, colsum);
            return check_launch("mdl_gemm_tn");
        }
    }
    MDL_REQUIRE(act == 0, MDL_E_UNSUPP, "mdl_gemm_tn_act: the activation staging needs even M, K, lda, ldb, 4-byte aligned rows and K <= 158");
    MDL_REQUIRE(!colsum, MDL_E_UNSUPP, "mdl_gemm_tn_colsum: the column sums need even M, K, lda, ldb, 4-byte aligned rows and K <= 158");
    int64_t grid = cdiv(N, 128);
    if (grid > 512) grid = 512;
    if (det) grid = 1;
    const int mt = (M + 31) / 32, ntw = ((K + 31) / 32 + 3) / 4;
#define MDL_TN(MT_, NTW_) hipLaunchKernelGGL((gemm_tn_kernel<MT_, NTW_>), dim3((unsigned)grid), dim3(256), 0, st, \
        (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, N)
    if (ntw == 1) { if (mt == 1) MDL_TN(1, 1); else if (mt == 2) MDL_TN(2, 1); else if (mt == 3) MDL_TN(3, 1); else MDL_TN(4, 1); }
    else { if (mt == 1) MDL_TN(1, 2); else if (mt == 2) MDL_TN(2, 2); else if (mt == 3) MDL_TN(3, 2); else MDL_TN(4, 2); }
#undef MDL_TN
    return check_launch("mdl_gemm_tn");
}
