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
// The four waves split the MT x NT grid of (32 x 32) C blocks into rectangles (2 x 2 waves; 1 x 4 / 4 x 1 when one side
// is a single tile): per k-step a wave reads the A fragments of its block rows and the B fragments of its block columns
// ONCE and issues rows x columns MFMAs.  (With blocks dealt round-robin every MFMA read both of its operands: 2 KB of
// LDS transpose reads per 32-cycle MFMA and wave — the 5 x 5 instantiation, SchNet's 150 x 150 filter layer, ran at a
// quarter of the HBM rate, bound by LDS.)
// (the 5-tile instantiations — SchNet's 150-wide filters — carry up to 7 accumulator tiles and 40 staging dwords per thread:
// they are register-allocated for ONE workgroup per CU, which is how the kernel is launched anyway; at two they spilled
// 664 bytes and ran 12x slower)
// ACT (mdl_gemm_tn_act): A is the gradient w.r.t. the OUTPUT y of an activated Linear and the product wanted is the one
// with the pre-activation gradient  A .* act'(y)  — relu: y > 0; shifted softplus: 1 - exp(-(y + ln 2)), both functions of
// the saved output.  The factor is applied while the A tile is staged (y is fetched with the same coalesced dwords), so
// the masked gradient never exists in HBM: for a Linear whose input needs no gradient (the first layer of SchNet's filter
// network acts on the edge features) that removes one pass over [E, M] entirely (read g, read y, write dpre).
template <int MT, int NT, int ACT = 0>
__global__ __launch_bounds__(256, (MT > 4 || NT > 4) ? 1 : 2) void gemm_tn_stream_kernel(const bf16_t* __restrict__ A, int lda, int M,
                                                                const bf16_t* __restrict__ B, int ldb, int K,
                                                                float* __restrict__ C, float* __restrict__ colsum,
                                                                int64_t N, const bf16_t* __restrict__ Y, int ldy) {
    // colsum (optional): column sums of A, i.e. the bias gradient of the Linear whose dW this is.  Column K of the B tile
    // (padding; the launcher picks NT so that it exists) is set to 1.0 for the valid rows, so the sums fall out of the
    // same MFMAs as column K of the product and are flushed to colsum instead of C.
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    constexpr int TN = 64;
    constexpr int LA = 32 * MT + 8, LB = 32 * NT + 8;           // LDS row strides (bf16): 8-byte aligned rows
    constexpr int WR = (MT == 1) ? 1 : (NT == 1 ? 4 : 2), WC = 4 / WR;       // waves along m / along n
    constexpr int RB = (MT + WR - 1) / WR, CB = (NT + WC - 1) / WC;           // block rows / columns per wave
    __shared__ __attribute__((aligned(16))) bf16_t al[TN * LA];
    __shared__ __attribute__((aligned(16))) bf16_t bl[TN * LB];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = tid >> 6;
    const int mt0 = __builtin_amdgcn_readfirstlane(wv / WC) * RB, nt0 = __builtin_amdgcn_readfirstlane(wv % WC) * CB;
    f32x16 acc[RB][CB];
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
    // Staging map (per wave: rows 16*wv .. 16*wv+15 of the 64-row tile; a padded row is DW = 16*MT dwords): dwords
    // 0 .. 64*Q-1 of a row are one load per 64 dwords (lane = dword), the remaining R = DW % 64 (16 or 32) dwords are
    // loaded for 64/R rows at once.  Everything about a thread's element is a compile-time constant except two per-thread
    // offsets, so a tile costs ~2 VALU per dword (an earlier flat chunk map — row = c / (16*MT) — spent ~15 VALU per dword
    // on index arithmetic, 1000+ instructions per tile and wave, and every shape ran at 2.2-2.9 TB/s regardless of its
    // MFMA or LDS load).  Rows past N and columns past M / K read as zeros through the buffer range check: the row offset
    // is part of the VGPR offset (the check does not cover the SGPR offset), invalid columns aim past the range.
    constexpr int DWA = 16 * MT, QA = DWA / 64, RA = DWA % 64, NRA = RA ? 16 * RA / 64 : 0, NLA = 16 * QA + NRA;
    constexpr int DWB = 16 * NT, QB = DWB / 64, RB_ = DWB % 64, NRB = RB_ ? 16 * RB_ / 64 : 0, NLB = 16 * QB + NRB;
    constexpr unsigned FAR = 0x40000000u;                         // beyond any tile's byte range
    const int w16 = 16 * wv;
    const unsigned arow = (unsigned)lda * 2u, brow = (unsigned)ldb * 2u, yrow = (unsigned)ldy * 2u;   // row strides in bytes
    unsigned areg[NLA], breg[NLB], yreg[ACT ? NLA : 1];
    auto load_tile = [&](int64_t tile) {
        const int64_t nb = tile * TN;
        const int64_t rows_left = N - nb;                                  // >= 1
        auto rsrc = [&](const bf16_t* base, int ld, int width) {
            const int64_t bytes = ((rows_left - 1) * (int64_t)ld + width) * 2;
            const int64_t cap = (int64_t)TN * ld * 2;
            return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + nb * (int64_t)ld), 0, (int)(bytes < cap ? bytes : cap), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t ra = rsrc(A, lda, M), rb = rsrc(B, ldb, K);
        const __amdgpu_buffer_rsrc_t ry = rsrc(ACT ? Y : A, ACT ? ldy : lda, M);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < QA; ++j) {
                const int d = lane + 64 * j;
                areg[r * QA + j] = __builtin_amdgcn_raw_buffer_load_b32(ra, (2 * d < M) ? (unsigned)(w16 + r) * arow + 4u * d : FAR, 0, 0);
                if constexpr (ACT != 0)
                    yreg[r * QA + j] = __builtin_amdgcn_raw_buffer_load_b32(ry, (2 * d < M) ? (unsigned)(w16 + r) * yrow + 4u * d : FAR, 0, 0);
            }
        if constexpr (RA != 0) {
            const int d = 64 * QA + lane % RA, rr = lane / RA;
#pragma unroll
            for (int k = 0; k < NRA; ++k) {
                const unsigned row = (unsigned)(w16 + k * (64 / RA) + rr);
                areg[16 * QA + k] = __builtin_amdgcn_raw_buffer_load_b32(ra, (2 * d < M) ? row * arow + 4u * d : FAR, 0, 0);
                if constexpr (ACT != 0)
                    yreg[16 * QA + k] = __builtin_amdgcn_raw_buffer_load_b32(ry, (2 * d < M) ? row * yrow + 4u * d : FAR, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const int d = lane + 64 * j;
                breg[r * QB + j] = __builtin_amdgcn_raw_buffer_load_b32(rb, (2 * d < K) ? (unsigned)(w16 + r) * brow + 4u * d : FAR, 0, 0);
            }
        if constexpr (RB_ != 0) {
            const int d = 64 * QB + lane % RB_, rr = lane / RB_;
#pragma unroll
            for (int k = 0; k < NRB; ++k)
                breg[16 * QB + k] = __builtin_amdgcn_raw_buffer_load_b32(rb, (2 * d < K) ? (unsigned)(w16 + k * (64 / RB_) + rr) * brow + 4u * d : FAR, 0, 0);
        }
    };
    auto act_fix = [&](unsigned v, unsigned yv) -> unsigned {
        if constexpr (ACT == 1) {
            return (__uint_as_float(yv << 16) > 0.0f ? (v & 0xffffu) : 0u) | (__uint_as_float(yv & 0xffff0000u) > 0.0f ? (v & 0xffff0000u) : 0u);
        } else if constexpr (ACT == 2) {                                  // same arithmetic as ssp_bwd_kernel (gather.hip)
            const float s0 = 1.0f - __expf(-(__uint_as_float(yv << 16) + 0.6931471805599453f));
            const float s1 = 1.0f - __expf(-(__uint_as_float(yv & 0xffff0000u) + 0.6931471805599453f));
            return pk_bf16(__uint_as_float(v << 16) * s0, __uint_as_float(v & 0xffff0000u) * s1);
        } else {
            return v;
        }
    };
    const int64_t n_tiles = (N + TN - 1) / TN;
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t nb = tile * TN;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < QA; ++j)
                *reinterpret_cast<unsigned*>(al + (w16 + r) * LA + 2 * (lane + 64 * j)) = act_fix(areg[r * QA + j], yreg[ACT ? r * QA + j : 0]);
        if constexpr (RA != 0) {
#pragma unroll
            for (int k = 0; k < NRA; ++k)
                *reinterpret_cast<unsigned*>(al + (w16 + k * (64 / RA) + lane / RA) * LA + 2 * (64 * QA + lane % RA)) =
                    act_fix(areg[16 * QA + k], yreg[ACT ? 16 * QA + k : 0]);
        }
        // B tile; with colsum, column K (a padding column: zero from the range check) becomes 1.0 for the rows that exist
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const int d = lane + 64 * j;
                unsigned v = breg[r * QB + j];
                if (colsum && 2 * d == K) v = (nb + w16 + r < N) ? 0x3F80u : 0u;
                *reinterpret_cast<unsigned*>(bl + (w16 + r) * LB + 2 * d) = v;
            }
        if constexpr (RB_ != 0) {
            const int d = 64 * QB + lane % RB_;
#pragma unroll
            for (int k = 0; k < NRB; ++k) {
                const int row = w16 + k * (64 / RB_) + lane / RB_;
                unsigned v = breg[16 * QB + k];
                if (colsum && 2 * d == K) v = (nb + row < N) ? 0x3F80u : 0u;
                *reinterpret_cast<unsigned*>(bl + row * LB + 2 * d) = v;
            }
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x);
        const int t = i & 15;
#pragma unroll
        for (int ks = 0; ks < TN / 16; ++ks) {
            const int roff = 16 * ks + 8 * h + (t >> 2), coff = (i & 16) + 4 * (t & 3);
            bf16x8 af[RB], bfr[CB];
#pragma unroll
            for (int a = 0; a < RB; ++a) {
                const bf16_t* pa = al + roff * LA + min(mt0 + a, MT - 1) * 32 + coff;      // (a row past MT re-reads the last one)
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pa + 4 * LA));
                af[a] = bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            }
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                const bf16_t* pb = bl + roff * LB + min(nt0 + b, NT - 1) * 32 + coff;
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pb), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pb + 4 * LB));
                bfr[b] = bf16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
#pragma unroll
            for (int a = 0; a < RB; ++a)
#pragma unroll
                for (int b = 0; b < CB; ++b)
                    if (mt0 + a < MT && nt0 + b < NT)                                          // wave-uniform
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < RB; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            const int mt = mt0 + a, nt = nt0 + b, col = nt * 32 + i;
            if (mt < MT && nt < NT && col < K) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mt * 32 + d_row(r, h);
                    if (m < M) unsafeAtomicAdd(C + (int64_t)m * K + col, acc[a][b][r]);
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
        for (int a = 0; a < RB; ++a)
#pragma unroll
            for (int b = 0; b < CB; ++b) {
                const int mt = mt0 + a, nt = nt0 + b;
                if (mt < MT && nt < NT && K >= nt * 32 && K < nt * 32 + 32 && i == K - nt * 32) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[mt * 32 + d_row(r, h)] = acc[a][b][r];
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
    return mdl_gemm_tn_act(a, lda, M, nullptr, 0, 0, b, ldb, K, c, colsum, N, dtype, stream);
}

extern "C" int mdl_gemm_tn_act(const void* a, int64_t lda, int M, const void* y, int64_t ldy, int act, const void* b,
                               int64_t ldb, int K, float* c, float* colsum, int64_t N, int dtype, mdlStream_t stream) {
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
#define MDL_TNS(MT_, NT_)                                                                                                   \
    do {                                                                                                                    \
        if (act == 0) hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 0>), dim3((unsigned)sgrid), dim3(256), 0, st,      \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)nullptr, 0);          \
        else if (act == 1) hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 1>), dim3((unsigned)sgrid), dim3(256), 0, st, \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)y, (int)ldy);         \
        else hipLaunchKernelGGL((gemm_tn_stream_kernel<MT_, NT_, 2>), dim3((unsigned)sgrid), dim3(256), 0, st,               \
            (const bf16_t*)a, (int)lda, M, (const bf16_t*)b, (int)ldb, K, c, colsum, N, (const bf16_t*)y, (int)ldy);         \
    } while (0)
            if (mt == 1) { if (nt == 1) MDL_TNS(1, 1); else if (nt == 2) MDL_TNS(1, 2); else if (nt == 4) MDL_TNS(1, 4); else MDL_TNS(1, 5); }
            else if (mt == 2) { if (nt == 1) MDL_TNS(2, 1); else if (nt == 2) MDL_TNS(2, 2); else if (nt == 4) MDL_TNS(2, 4); else MDL_TNS(2, 5); }
            else if (mt == 4) { if (nt == 1) MDL_TNS(4, 1); else if (nt == 2) MDL_TNS(4, 2); else if (nt == 4) MDL_TNS(4, 4); else MDL_TNS(4, 5); }
            else { if (nt == 1) MDL_TNS(5, 1); else if (nt == 2) MDL_TNS(5, 2); else if (nt == 4) MDL_TNS(5, 4); else MDL_TNS(5, 5); }
#undef MDL_TNS
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
