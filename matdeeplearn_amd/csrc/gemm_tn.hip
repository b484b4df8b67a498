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

// ------------------------------------------------------------------------------------------
// Streaming version (default when rows are dword-addressable): 64-row tiles of A and B are read once with coalesced
// dword loads (registers, one tile ahead), written to row-major LDS tiles, and both MFMA operands — which must be
// k-major over the ROWS — come out of those tiles with the LDS transpose read ds_read_b64_tr_b16.
// Wave w owns the (32 x 32) blocks w, w+4, ... of C (row-major over (m-tile, n-tile)).
// (the 5-tile instantiations — SchNet's 150-wide filters — carry up to 7 accumulator tiles and 40 staging dwords per thread:
// they are register-allocated for ONE workgroup per CU, which is how the kernel is launched anyway; at two they spilled
// 664 bytes and ran 12x slower)
template <int MT, int NT>
__global__ __launch_bounds__(256, (MT > 4 || NT > 4) ? 1 : 2) void gemm_tn_stream_kernel(const bf16_t* __restrict__ A, int lda, int M,
                                                                const bf16_t* __restrict__ B, int ldb, int K,
                                                                float* __restrict__ C, float* __restrict__ colsum,
                                                                int64_t N) {
    // colsum (optional): column sums of A, i.e. the bias gradient of the Linear whose dW this is.  Column K of the B tile
    // (padding; the launcher picks NT so that it exists) is set to 1.0 for the valid rows, so the sums fall out of the
    // same MFMAs as column K of the product and are flushed to colsum instead of C.
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    constexpr int TN = 64;
    constexpr int LA = 32 * MT + 8, LB = 32 * NT + 8;           // LDS row strides (bf16): 8-byte aligned rows
    constexpr int NLA = 4 * MT, NLB = 4 * NT;                   // dwords per thread: 64 rows x 16*MT dwords / 256
    constexpr int NBLK = (MT * NT + 3) / 4;                     // C blocks per wave
    __shared__ __attribute__((aligned(16))) bf16_t al[TN * LA];
    __shared__ __attribute__((aligned(16))) bf16_t bl[TN * LB];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = tid >> 6;
    f32x16 acc[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    // chunk c = l*256 + tid of a tile: row = c / (16*MT), dword d = c % (16*MT)  (16*MT divides 256 for MT in {1,2,4}; for
    // MT = 3 the mapping is still a bijection onto [0, 64*48) but rows advance irregularly -> computed per l)
    unsigned areg[NLA], breg[NLB];
    auto load_tile = [&](int64_t tile) {
        const int64_t nb = tile * TN;
#pragma unroll
        for (int l = 0; l < NLA; ++l) {
            const int c = l * 256 + tid, row = c / (16 * MT), d = c - row * (16 * MT);
            const int64_t n = min(nb + row, N - 1);
            areg[l] = *reinterpret_cast<const unsigned*>(A + n * lda + 2 * min(d, (M - 1) / 2));      // clamp, never guard
        }
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
            const int c = l * 256 + tid, row = c / (16 * NT), d = c - row * (16 * NT);
            const int64_t n = min(nb + row, N - 1);
            breg[l] = *reinterpret_cast<const unsigned*>(B + n * ldb + 2 * min(d, (K - 1) / 2));
        }
    };
    const int64_t n_tiles = (N + TN - 1) / TN;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t nb = tile * TN;
        __syncthreads();
#pragma unroll
        for (int l = 0; l < NLA; ++l) {
            const int c = l * 256 + tid, row = c / (16 * MT), d = c - row * (16 * MT);
            unsigned v = areg[l];
            if (nb + row >= N || 2 * d >= M) v = 0u;                      // (M, K even: a dword never straddles the edge)
            *reinterpret_cast<unsigned*>(al + row * LA + 2 * d) = v;
        }
#pragma unroll
        for (int l = 0; l < NLB; ++l) {
            const int c = l * 256 + tid, row = c / (16 * NT), d = c - row * (16 * NT);
            unsigned v = breg[l];
            if (nb + row >= N || 2 * d >= K) v = (colsum && 2 * d == K && nb + row < N) ? 0x3F80u : 0u;
            *reinterpret_cast<unsigned*>(bl + row * LB + 2 * d) = v;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x);
        const int t = i & 15;
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int blk = wv + 4 * j;
            if (blk < MT * NT) {
                const int mt = blk / NT, nt = blk - mt * NT;
#pragma unroll
                for (int ks = 0; ks < TN / 16; ++ks) {
                    const bf16_t* pa = al + (16 * ks + 8 * h + (t >> 2)) * LA + mt * 32 + (i & 16) + 4 * (t & 3);
                    const bf16_t* pb = bl + (16 * ks + 8 * h + (t >> 2)) * LB + nt * 32 + (i & 16) + 4 * (t & 3);
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pa + 4 * LA));
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pb), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pb + 4 * LB));
                    const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const bf16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[j], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        const int blk = wv + 4 * j;
        if (blk < MT * NT) {
            const int mt = blk / NT, nt = blk - mt * NT;
            const int col = nt * 32 + i;
            if (col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mt * 32 + d_row(r, h);
                    if (m < M) unsafeAtomicAdd(C + (int64_t)m * K + col, acc[j][r]);
                }
            }
        }
    }
    // column sums: column K of the product lives in ONE lane per 32-row block (two with the h halves).  Flushed row by
    // row that is one two-lane atomic instruction per row — thousands of separate transactions on the same two cache
    // lines per launch (measured: 40 -> 83 us).  Gather the block's rows through LDS and add them with one instruction.
    if (colsum) {                                          // (kernel argument: uniform, the barriers are safe)
        __syncthreads();                                   // every wave is done with the tiles
        float* sc = reinterpret_cast<float*>(al);          // [32*MT] floats, al is 64*(32*MT+8) bf16
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int blk = wv + 4 * j;
            if (blk < MT * NT) {
                const int mt = blk / NT, nt = blk - mt * NT;
                if (K >= nt * 32 && K < nt * 32 + 32 && i == K - nt * 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[mt * 32 + d_row(r, h)] = acc[j][r];
                }
            }
        }
        __syncthreads();
        if (tid < 32 * MT && tid < M) unsafeAtomicAdd(colsum + tid, sc[tid]);
    }
}

}  // namespace mdl

extern "C" int mdl_gemm_tn(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c, int64_t N,
                           int dtype, mdlStream_t stream) {
    return mdl_gemm_tn_colsum(a, lda, M, b, ldb, K, c, nullptr, N, dtype, stream);
}

extern "C" int mdl_gemm_tn_colsum(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c,
                                  float* colsum, int64_t N, int dtype, mdlStream_t stream) {
    using namespace mdl;
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
            int64_t sgrid = cdiv(N, 64);
            if (sgrid > 256) sgrid = 256;     // one block per CU: every block ends with M*K atomics on the same addresses
                                              // (measured 128 / 256 / 512 / 1024 blocks: 49 / 40 / 43 / 63 us on 2e5 rows)
#define MDL_TNS(MT_, NT_) hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_>), dim3((unsigned)sgrid), dim3(256), 0, st, \
        (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N)
            if (mt == 1) { if (nt == 1) MDL_TNS(1, 1); else if (nt == 2) MDL_TNS(1, 2); else if (nt == 4) MDL_TNS(1, 4); else MDL_TNS(1, 5); }
            else if (mt == 2) { if (nt == 1) MDL_TNS(2, 1); else if (nt == 2) MDL_TNS(2, 2); else if (nt == 4) MDL_TNS(2, 4); else MDL_TNS(2, 5); }
            else if (mt == 4) { if (nt == 1) MDL_TNS(4, 1); else if (nt == 2) MDL_TNS(4, 2); else if (nt == 4) MDL_TNS(4, 4); else MDL_TNS(4, 5); }
            else { if (nt == 1) MDL_TNS(5, 1); else if (nt == 2) MDL_TNS(5, 2); else if (nt == 4) MDL_TNS(5, 4); else MDL_TNS(5, 5); }
#undef MDL_TNS
            return check_launch("mdl_gemm_tn");
        }
    }
    MDL_REQUIRE(!colsum, MDL_E_UNSUPP, "mdl_gemm_tn_colsum: the column sums need even M, K, lda, ldb, 4-byte aligned rows and K <= 126");
    int64_t grid = cdiv(N, 128);
    if (grid > 512) grid = 512;
    const int mt = (M + 31) / 32, ntw = ((K + 31) / 32 + 3) / 4;
#define MDL_TN(MT_, NTW_) hipLaunchKernelGGL((gemm_tn_kernel<MT_, NTW_>), dim3((unsigned)grid), dim3(256), 0, st, \
        (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, N)
    if (ntw == 1) { if (mt == 1) MDL_TN(1, 1); else if (mt == 2) MDL_TN(2, 1); else if (mt == 3) MDL_TN(3, 1); else MDL_TN(4, 1); }
    else { if (mt == 1) MDL_TN(1, 2); else if (mt == 2) MDL_TN(2, 2); else if (mt == 3) MDL_TN(3, 2); else MDL_TN(4, 2); }
#undef MDL_TN
    return check_launch("mdl_gemm_tn");
}
