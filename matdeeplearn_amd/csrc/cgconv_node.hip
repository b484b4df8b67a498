// cgconv_node.hip — K3c: node-level dense half of the CGConv backward (bf16 mode).
//
// After the edge pass (cgconv.hip) has reduced dpre by target (r_tgt) and by source (r_src),
//     dx  = grad_out + [r_tgt | r_src] (N x 4Cp)  @  Wn (4Cp x C)
//     dWn = [r_tgt | r_src]^T (4Cp x N)           @  x  (N x C)
// with Wn rows ordered (f_tgt, s_tgt, f_src, s_src) like the columns of [r_tgt | r_src].
// The reference gets these from autograd through eager cat/addmm ops
// (/root/reference/matdeeplearn/models/cgcnn.py:136-145 via PyG CGConv); a library GEMM handles the
// (4Cp x N)(N x C) product badly (K = N ~ 2e5, 256 x 64 output), so both products are done here in one
// pass over r_tgt/r_src: HBM-bound, algorithmic bytes N*(2*2Cp*4 + 3*C*2).
//
// Workgroup = 4 waves = 128 consecutive nodes.  Wave w: dx rows of its own 32 nodes (MFMA, K = 4Cp),
// and the dWn row block [w*Cp, (w+1)*Cp) over all 128 nodes (MFMA, K = nodes).  dWn partials stay
// in registers across the grid-stride loop and are flushed once per wave with fp32 atomics.
#include "mdl_common.h"

#ifndef MDL_NODE_STREAM
#define MDL_NODE_STREAM 1   // LDS-staged streaming kernel (0: the first, strided-load kernel below)
#endif

namespace mdl {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4_t r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, r);
}

template <int CP>   // CP = padded channels = C (32 or 64)
__global__ __launch_bounds__(256, 2) void cgconv_node_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gout,
                                                             const float* __restrict__ r_tgt,
                                                             const float* __restrict__ r_src,
                                                             const bf16_t* __restrict__ wn_t, bf16_t* __restrict__ dx,
                                                             float* __restrict__ dwn, int64_t N) {
    constexpr int K4 = 4 * CP;          // columns of [r_tgt | r_src]
    constexpr int LD = K4 + 8;          // LDS row stride of Wn^T (odd number of 16-byte slots)
    constexpr int NT = CP / 32;         // 32-wide feature tiles
    constexpr int MT = CP / 32;         // 32-row tiles of this wave's dWn row block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);
    for (int q = threadIdx.x; q < CP * (K4 / 8); q += blockDim.x) {
        const int row = q / (K4 / 8), c8 = q - row * (K4 / 8);
        *reinterpret_cast<bf16x8*>(wl + row * LD + c8 * 8) = *reinterpret_cast<const bf16x8*>(wn_t + row * K4 + c8 * 8);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    f32x16 dw[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[a][b][r] = 0.0f;

    const int64_t n_super = (N + 127) / 128;
    for (int64_t sc = blockIdx.x; sc < n_super; sc += gridDim.x) {
        const int64_t nb = sc * 128;
        // ---- dx for this wave's 32 nodes -------------------------------------------------------
        {
            const int64_t node = nb + wv * 32 + i;
            const bool ok = node < N;
            const float* rt = r_tgt + node * (2 * CP);
            const float* rs = r_src + node * (2 * CP);
            f32x16 acc[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K4 / 16; ++kk) {
                const int c0 = 16 * kk + 8 * h;
                const float* src = (c0 < 2 * CP) ? rt + c0 : rs + (c0 - 2 * CP);
                float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                }
                const bf16x8 a = pack8(v);
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const bf16x8 bb = *reinterpret_cast<const bf16x8*>(wl + (b * 32 + i) * LD + 16 * kk + 8 * h);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t n = nb + wv * 32 + d_row(r, h);
                    if (n < N) {
                        const int64_t o = n * CP + b * 32 + i;
                        dx[o] = f2bf(bf2f(gout[o]) + acc[b][r]);
                    }
                }
        }
        // ---- dWn rows [wv*CP, (wv+1)*CP) over the 128 nodes ---------------------------------------
        {
            const float* rbase = (wv < 2) ? r_tgt : r_src;
            const int coff = (wv & 1) * CP;                 // f-half / s-half inside the 2CP row
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 bfr[NT];
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    bf16x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int64_t n = nb + 16 * ks + 8 * h + q;
                        t[q] = (n < N) ? (short)x[n * CP + b * 32 + i] : (short)0;
                    }
                    bfr[b] = t;
                }
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int64_t n = nb + 16 * ks + 8 * h + q;
                        v[q] = (n < N) ? rbase[n * (2 * CP) + coff + a * 32 + i] : 0.0f;
                    }
                    const bf16x8 af = pack8(v);
#pragma unroll
                    for (int b = 0; b < NT; ++b) dw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[b], dw[a][b], 0, 0, 0);
                }
            }
        }
    }
    // flush dWn partials: D rows = channel slot, cols = feature
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wv * CP + a * 32 + d_row(r, h);
                unsafeAtomicAdd(dwn + (int64_t)row * CP + b * 32 + i, dw[a][b][r]);
            }
}

// ------------------------------------------------------------------------------------------
// Streaming version (default): 64-node tiles staged through LDS.
//   * [r_tgt | r_src] rows of the tile are read ONCE from HBM with 16-byte coalesced loads (registers, one tile
//     ahead), converted to bf16 and written to an LDS tile; x rows likewise;
//   * dx uses the tile row-wise (A fragments = ds_read_b128), dWn needs both operands k-major over the NODES
//     (R^T and x as [k = node] fragments): those come out of the same row-major tiles with the LDS transpose
//     read ds_read_b64_tr_b16 — no strided global loads, no second pass over r_tgt/r_src.
// Wave w: one (32-node, 32-feature) block of dx and MT*NT/4... of the (4Cp x C) dWn blocks (kept in registers over
// the grid-stride loop, flushed once with fp32 atomics).
template <int CP>
__global__ __launch_bounds__(256, 2) void cgconv_node_stream_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gout,
                                                                    const float* __restrict__ r_tgt,
                                                                    const float* __restrict__ r_src,
                                                                    const bf16_t* __restrict__ wn_t, bf16_t* __restrict__ dx,
                                                                    float* __restrict__ dwn, int64_t N) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    constexpr int TN = 64;               // nodes per tile
    constexpr int K4 = 4 * CP;           // columns of [r_tgt | r_src]
    constexpr int LD = K4 + 8;           // LDS row stride of Wn^T and of the R tile (odd number of 16-byte slots)
    constexpr int LX = CP + 8;           // LDS row stride of the x tile
    constexpr int NT = CP / 32;          // 32-wide feature tiles
    constexpr int RCH = K4 / 4;          // 16-byte chunks (4 floats) per R row
    constexpr int NRL = TN * RCH / 256;  // R chunks per thread: 16 (CP 64) / 8 (CP 32)
    constexpr int XCH = CP / 8;          // 16-byte chunks per x row
    constexpr int NXL = TN * XCH / 256;  // x chunks per thread: 2 / 1
    constexpr int MJ = (K4 / 32) * NT / 4;   // (32-row, 32-col) dWn blocks per wave: 4 / 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                    // Wn^T  [CP][LD]
    bf16_t* rl = wl + CP * LD;                                       // R tile [TN][LD]
    bf16_t* xl = rl + TN * LD;                                       // x tile [TN][LX]
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = tid >> 6;
    for (int q = tid; q < CP * (K4 / 8); q += 256) {
        const int row = q / (K4 / 8), c8 = q - row * (K4 / 8);
        *reinterpret_cast<bf16x8*>(wl + row * LD + c8 * 8) = *reinterpret_cast<const bf16x8*>(wn_t + row * K4 + c8 * 8);
    }
    f32x16 dw[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[j][r] = 0.0f;

    const int64_t n_tiles = (N + TN - 1) / TN;
    f32x4 rreg[NRL];
    u32x4_t xreg[NXL];
    // thread -> chunk mapping: chunk c = l*256 + tid; with RCH | 256 or 256 | RCH*k the row advances by a constant per l,
    // so full tiles use one per-thread pointer + compile-time offsets
    static_assert(256 % RCH == 0 && 256 % XCH == 0, "chunk mapping");
    constexpr int RROWS = 256 / RCH, XROWS = 256 / XCH;      // rows covered by one load of the whole workgroup
    const int rrow0 = tid / RCH, rcc = tid % RCH, xrow0 = tid / XCH, xcc = tid % XCH;
    const float* rsel = (rcc < RCH / 2) ? r_tgt + 4 * rcc : r_src + 4 * (rcc - RCH / 2);
    auto load_tile = [&](int64_t tile) {
        const int64_t nb = tile * TN;
        if (nb + TN <= N) {
            const float* rp = rsel + (nb + rrow0) * (2 * CP);
            const bf16_t* xp = x + (nb + xrow0) * CP + 8 * xcc;
#pragma unroll
            for (int l = 0; l < NRL; ++l) rreg[l] = *reinterpret_cast<const f32x4*>(rp + l * (RROWS * 2 * CP));
#pragma unroll
            for (int l = 0; l < NXL; ++l) xreg[l] = *reinterpret_cast<const u32x4_t*>(xp + l * (XROWS * CP));
        } else {                                                                          // last tile: clamp the rows
#pragma unroll
            for (int l = 0; l < NRL; ++l)
                rreg[l] = *reinterpret_cast<const volatile f32x4*>(rsel + min(nb + rrow0 + l * RROWS, N - 1) * (2 * CP));
#pragma unroll
            for (int l = 0; l < NXL; ++l)
                xreg[l] = *reinterpret_cast<const volatile u32x4_t*>(x + min(nb + xrow0 + l * XROWS, N - 1) * CP + 8 * xcc);
        }
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t nb = tile * TN;
        __syncthreads();                                   // everyone is done reading the previous tile (and Wn is in)
#pragma unroll
        for (int l = 0; l < NRL; ++l) {
            const int row = rrow0 + l * RROWS, cc = rcc;
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
            const u32x2_t v = {pk_bf16(rreg[l][0], rreg[l][1]), pk_bf16(rreg[l][2], rreg[l][3])};
            *reinterpret_cast<u32x2_t*>(rl + row * LD + 4 * cc) = v;
        }
#pragma unroll
        for (int l = 0; l < NXL; ++l) {
            const int row = xrow0 + l * XROWS, cc = xcc;
            u32x4_t v = xreg[l];
            if (nb + row >= N) v = u32x4_t{0u, 0u, 0u, 0u};         // rows past the end drop out of dWn
            *reinterpret_cast<u32x4_t*>(xl + row * LX + 8 * cc) = v;
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) load_tile(tile + gridDim.x);   // next tile's loads fly during this tile's MFMAs

        // ---- dx block (mt, nt) of this wave: rows = 32 nodes, K = 4Cp, cols = 32 features
        if (wv < 2 * NT) {
            const int mt = wv / NT, nt = wv - mt * NT;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K4 / 16; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rl + (mt * 32 + i) * LD + 16 * kk + 8 * h);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(wl + (nt * 32 + i) * LD + 16 * kk + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
            bf16_t gv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[r] = gout[min(nb + mt * 32 + d_row(r, h), N - 1) * CP + nt * 32 + i];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t n = nb + mt * 32 + d_row(r, h);
                if (n < N) dx[n * CP + nt * 32 + i] = f2bf(bf2f(gv[r]) + acc[r]);
            }
        }
        // ---- dWn blocks of this wave: rows = 32 columns of R, cols = 32 features, K = the tile's 64 nodes
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int blk = wv * MJ + j, mt = blk / NT, nt = blk - mt * NT;
            const int t = i & 15;
#pragma unroll
            for (int ks = 0; ks < TN / 16; ++ks) {
                // k-major fragments over the nodes 16ks + 8h + {0..7}: two transpose reads of [4 nodes][16 cols] blocks
                const bf16_t* pa = rl + (16 * ks + 8 * h + (t >> 2)) * LD + mt * 32 + (i & 16) + 4 * (t & 3);
                const bf16_t* pb = xl + (16 * ks + 8 * h + (t >> 2)) * LX + nt * 32 + (i & 16) + 4 * (t & 3);
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pa + 4 * LD));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pb), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pb + 4 * LX));
                const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const bf16x8 bfr = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                dw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, dw[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int blk = wv * MJ + j, mt = blk / NT, nt = blk - mt * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) unsafeAtomicAdd(dwn + (int64_t)(mt * 32 + d_row(r, h)) * CP + nt * 32 + i, dw[j][r]);
    }
}

}  // namespace mdl

namespace mdl {
// wn_t [C][4Cp] (dtype bf16) = transpose of Wn = rows (f_tgt, s_tgt, f_src, s_src) of the two Linears' node columns
__global__ __launch_bounds__(256) void cgconv_pack_node_kernel(const float* __restrict__ wf, const float* __restrict__ ws, int C,
                                                               int Cp, int ldw, bf16_t* __restrict__ wn_t) {
    const int total = C * 4 * Cp;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const int k = q / (4 * Cp), r = q - k * (4 * Cp);          // wn_t[k][r] = Wn[r][k]
        const int blk = r / Cp, c = r - blk * Cp;                  // blk: 0 f_tgt, 1 s_tgt, 2 f_src, 3 s_src
        float v = 0.0f;
        if (c < C) v = ((blk & 1) ? ws : wf)[c * ldw + (blk >> 1) * C + k];
        wn_t[q] = f2bf(v);
    }
}
// dW_f / dW_s [C][2C+G] and db_f / db_s [C] from the kernels' partial layouts (dwn [4Cp][C], dwe [2Cp][GP], db [2Cp])
__global__ __launch_bounds__(256) void cgconv_grads_kernel(const float* __restrict__ dwn, const float* __restrict__ dwe,
                                                           const float* __restrict__ db, int C, int Cp, int G, int GP,
                                                           float* __restrict__ dwf, float* __restrict__ dws,
                                                           float* __restrict__ dbf, float* __restrict__ dbs) {
    const int ldw = 2 * C + G, total = 2 * C * ldw;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const int part = q / (C * ldw), rem = q - part * (C * ldw);
        const int c = rem / ldw, k = rem - c * ldw;
        float v;
        if (k < C) v = dwn[(part * Cp + c) * C + k];                              // target columns
        else if (k < 2 * C) v = dwn[((2 + part) * Cp + c) * C + (k - C)];         // source columns
        else v = dwe[(part * Cp + c) * GP + (k - 2 * C)];                         // edge-feature columns
        (part ? dws : dwf)[rem] = v;
    }
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < 2 * C; q += gridDim.x * blockDim.x) {
        const int part = q / C, c = q - part * C;
        float* dst = part ? dbs : dbf;
        if (dst) dst[c] = db[part * Cp + c];
    }
}
}  // namespace mdl

extern "C" int mdl_cgconv_pack_node_weights(const float* w_f, const float* w_s, int C, int G, void* wn_t, int dtype,
                                            mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_pack_node_weights: bf16 only");
    MDL_REQUIRE(w_f && w_s && wn_t && C >= 1 && G >= 0, MDL_E_ARG, "mdl_cgconv_pack_node_weights: bad arguments");
    const int Cp = (C + 31) / 32 * 32;
    hipLaunchKernelGGL(cgconv_pack_node_kernel, dim3((unsigned)cdiv((int64_t)C * 4 * Cp, 256)), dim3(256), 0, (hipStream_t)stream, w_f,
                       w_s, C, Cp, 2 * C + G, (bf16_t*)wn_t);
    return check_launch("mdl_cgconv_pack_node_weights");
}

extern "C" int mdl_cgconv_assemble_grads(const float* dwn, const float* dwe, const float* db, int C, int G, float* dw_f,
                                         float* dw_s, float* db_f, float* db_s, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dwn && dwe && db && dw_f && dw_s && C >= 1 && G >= 0, MDL_E_ARG, "mdl_cgconv_assemble_grads: bad arguments");
    const int Cp = (C + 31) / 32 * 32, GP = (G + 63) / 64 * 64;
    hipLaunchKernelGGL(cgconv_grads_kernel, dim3((unsigned)cdiv((int64_t)2 * C * (2 * C + G), 256)), dim3(256), 0, (hipStream_t)stream,
                       dwn, dwe, db, C, Cp, G, GP, dw_f, dw_s, db_f, db_s);
    return check_launch("mdl_cgconv_assemble_grads");
}

extern "C" int mdl_cgconv_bwd_node(const void* x, const void* grad_out, const float* r_tgt, const float* r_src,
                                   const void* wn_t, void* dx, float* dwn, int64_t N, int C, int dtype,
                                   mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: bf16 only (fp32 parity mode uses library GEMMs)");
    MDL_REQUIRE(C == 32 || C == 64, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: C must be 32 or 64 (got %d)", C);
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && grad_out && r_tgt && r_src && wn_t && dx && dwn)), MDL_E_ARG,
                "mdl_cgconv_bwd_node: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(wn_t) % 16 == 0 && reinterpret_cast<uintptr_t>(r_tgt) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(r_src) % 16 == 0, MDL_E_ARG, "mdl_cgconv_bwd_node: 16-byte alignment required");
    if (N == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
#if MDL_NODE_STREAM
    {
        int64_t sgrid = cdiv(N, 64);
        if (sgrid > 512) sgrid = 512;
        if (C == 64) {
            const int lds = (64 * (256 + 8) + 64 * (256 + 8) + 64 * (64 + 8)) * 2;
            auto kf = cgconv_node_stream_kernel<64>;
            hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL(kf, dim3((unsigned)sgrid), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)grad_out,
                               r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
        } else {
            const int lds = (32 * (128 + 8) + 64 * (128 + 8) + 64 * (32 + 8)) * 2;
            auto kf = cgconv_node_stream_kernel<32>;
            hipLaunchKernelGGL(kf, dim3((unsigned)sgrid), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)grad_out,
                               r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
        }
        return check_launch("mdl_cgconv_bwd_node");
    }
#endif
    int64_t grid = cdiv(N, 128);
    if (grid > 512) grid = 512;
    if (C == 64) {
        const int lds = 64 * (256 + 8) * 2;
        hipLaunchKernelGGL((cgconv_node_kernel<64>), dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)x,
                           (const bf16_t*)grad_out, r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
    } else {
        const int lds = 32 * (128 + 8) * 2;
        hipLaunchKernelGGL((cgconv_node_kernel<32>), dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)x,
                           (const bf16_t*)grad_out, r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
    }
    return check_launch("mdl_cgconv_bwd_node");
}
