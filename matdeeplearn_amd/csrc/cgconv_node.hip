// cgconv_node.hip — K3c: node-level dense half of the CGConv backward (bf16 mode).
//
// After the edge pass (cgconv.hip) has reduced dpre by target (r_tgt) and by source (r_src),
//     dx  = grad_out + [r_tgt | r_src] (N x 4Cp)  @  Wn (4Cp x C)
//     dWn = [r_tgt | r_src]^T (4Cp x N)           @  x  (N x C)
// with Wn rows ordered (f_tgt, s_tgt, f_src, s_src) like the columns of [r_tgt | r_src].
// The reference gets these from autograd through eager cat/addmm ops
// (/root/reference/matdeeplearn/models/cgcnn.py:136-145 via PyG CGConv); a library GEMM handles the
// (4Cp x N)(N x C) product badly (K = N ~ 2e5, 256 x 64 output), so both products are done here in one
// pass over r_tgt/r_src: HBM-bound, algorithmic bytes N*(2Cp*2 + 2Cp*4 + 3*C*2).
//
// r_tgt arrives in bf16 (the edge kernel writes it once per node in the compute dtype), r_src in fp32 (it is
// accumulated with atomics).  dWn partials stay in registers across the grid-stride loop and are flushed once per
// wave with fp32 atomics.
#include "mdl_common.h"


#ifndef MDL_K3C_NW
#define MDL_K3C_NW 8        // waves per workgroup of the C = 64 node kernel (see the kernel's comment)
#endif

namespace mdl {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4_t r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, r);
}

// ------------------------------------------------------------------------------------------
// 64-node tiles staged through LDS.
//   * [r_tgt | r_src] rows of the tile are read ONCE from HBM with 16-byte coalesced loads (registers, one tile
//     ahead), converted to bf16 and written to an LDS tile; x rows likewise;
//   * dx uses the tile row-wise (A fragments = ds_read_b128), dWn needs both operands k-major over the NODES
//     (R^T and x as [k = node] fragments): those come out of the same row-major tiles with the LDS transpose
//     read ds_read_b64_tr_b16 — no strided global loads, no second pass over r_tgt/r_src.
// Wave w: one (32-node, 32-feature) block of dx and MT*NT/4... of the (4Cp x C) dWn blocks (kept in registers over
// the grid-stride loop, flushed once with fp32 atomics).
// RS16: r_src arrives in bf16 as well (the edge pass accumulated it with packed bf16 atomics, mdl_cgconv_bwd_h): half the
// bytes to read and to zero, no conversion.
// NW: waves per workgroup.  The kernel runs one workgroup per CU (its tiles and Wn^T take 77 KB of LDS at C = 64, and every
// workgroup ends with 4Cp*C atomics on the same addresses), so 4 waves are ONE wave per SIMD; 8 waves split the staging and the
// dWn blocks in half per wave and put two waves on every SIMD.
template <int CP, bool RS16 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void cgconv_node_stream_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gout,
                                                                    const bf16_t* __restrict__ r_tgt,
                                                                    const float* __restrict__ r_src,
                                                                    const bf16_t* __restrict__ wn_t, bf16_t* __restrict__ dx,
                                                                    float* __restrict__ dwn, int64_t N, int zero_src, int ld_out) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    constexpr int TN = 64;               // nodes per tile
    constexpr int K4 = 4 * CP;           // columns of [r_tgt | r_src]
    constexpr int LD = K4 + 8;           // LDS row stride of Wn^T and of the R tile (odd number of 16-byte slots)
    constexpr int LX = CP + 8;           // LDS row stride of the x tile
    constexpr int NT = CP / 32;          // 32-wide feature tiles
    constexpr int TCH = 2 * CP / 8;      // 16-byte chunks per r_tgt row (8 bf16)
    constexpr int SCH = 2 * CP / 4;      // 16-byte chunks per r_src row (4 floats)
    constexpr int NTH = 64 * NW;         // threads
    constexpr int NTL = TN * TCH / NTH;  // r_tgt chunks per thread: 4 (CP 64) / 2 (CP 32) with 4 waves
    constexpr int NSL = RS16 ? NTL : TN * SCH / NTH;  // r_src chunks per thread: 8 / 4 (fp32), like r_tgt when it is bf16
    constexpr int XCH = CP / 8;          // 16-byte chunks per x row
    constexpr int NXL = TN * XCH / NTH;  // x chunks per thread: 2 / 1
    constexpr int MJ = (K4 / 32) * NT / NW;  // (32-row, 32-col) dWn blocks per wave: 4 / 1
    static_assert(NTL >= 1 && NSL >= 1 && NXL >= 1 && MJ >= 1, "too many waves for this width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                    // Wn^T  [CP][LD]
    bf16_t* rl = wl + CP * LD;                                       // R tile [TN][LD]  (columns: r_tgt | r_src)
    bf16_t* xl = rl + TN * LD;                                       // x tile [TN][LX]
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int q = tid; q < CP * (K4 / 8); q += NTH) {
        const int row = q / (K4 / 8), c8 = q - row * (K4 / 8);
        *reinterpret_cast<bf16x8*>(wl + row * LD + c8 * 8) = *reinterpret_cast<const bf16x8*>(wn_t + row * K4 + c8 * 8);
    }
    f32x16 dw[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[j][r] = 0.0f;

    const int64_t n_tiles = (N + TN - 1) / TN;
    u32x4_t treg[NTL];
    f32x4 sreg[NSL];
    u32x4_t xreg[NXL];
    // thread -> chunk mapping: chunk c = l*256 + tid; the chunk counts per row divide 256, so the row advances by a
    // constant per l
    static_assert(NTH % TCH == 0 && NTH % SCH == 0 && NTH % XCH == 0, "chunk mapping");
    constexpr int TROWS = NTH / TCH, SROWS = NTH / SCH, XROWS = NTH / XCH;     // rows covered by one load of the workgroup
    const int trow0 = tid / TCH, tcc = tid % TCH, srow0 = tid / SCH, scc = tid % SCH, xrow0 = tid / XCH, xcc = tid % XCH;
    // Buffer loads: the tile base is uniform (a fresh resource per tile, so any N works), each thread keeps ONE 32-bit
    // byte offset per array plus a constant per-l row stride in the same VOFFSET expression (the range check covers
    // voffset + immediate, NOT an SGPR soffset), and rows past N read as zeros / are dropped on stores (range check
    // against the bytes that remain) — no clamps, no separate last-tile path.
    const int to = (trow0 * (2 * CP) + 8 * tcc) * 2, so = (srow0 * (2 * CP) + 4 * scc) * 4, xo = (xrow0 * CP + 8 * xcc) * 2;
    auto rsrc = [](const void* base, int64_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0x7fffffffLL ? bytes : 0x7fffffffLL), 0x00020000);
    };
    // the per-l strides (multiples of 4096: too large for the 12-bit immediate) are added to the per-thread offset right
    // at the load from an SGPR the optimiser cannot see through — hoisted out of the tile loop they would be 11 more
    // loop-invariant VGPRs in a kernel that sits at the 256-register limit (they spilled, and a reload drains the queue)
    auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto load_tile = [&](int64_t tile) {
        const int64_t nb = tile * TN, rem = N - nb;
        const __amdgpu_buffer_rsrc_t tr = rsrc(r_tgt + nb * (2 * CP), rem * (2 * CP * 2));
        const __amdgpu_buffer_rsrc_t sr = RS16 ? rsrc(reinterpret_cast<const bf16_t*>(r_src) + nb * (2 * CP), rem * (2 * CP * 2))
                                               : rsrc(r_src + nb * (2 * CP), rem * (2 * CP * 4));
        const __amdgpu_buffer_rsrc_t xr = rsrc(x + nb * CP, rem * (CP * 2));
#pragma unroll
        for (int l = 0; l < NTL; ++l) treg[l] = __builtin_amdgcn_raw_buffer_load_b128(tr, to + opaque(l * (TROWS * 2 * CP * 2)), 0, 0);
#pragma unroll
        for (int l = 0; l < NSL; ++l)
            sreg[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                sr, RS16 ? to + opaque(l * (TROWS * 2 * CP * 2)) : so + opaque(l * (SROWS * 2 * CP * 4)), 0, 0));
#pragma unroll
        for (int l = 0; l < NXL; ++l) xreg[l] = __builtin_amdgcn_raw_buffer_load_b128(xr, xo + opaque(l * (XROWS * CP * 2)), 0, 0);
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t nb = tile * TN;
        __syncthreads();                                   // everyone is done reading the previous tile (and Wn is in)
#pragma unroll
        for (int l = 0; l < NTL; ++l) *reinterpret_cast<u32x4_t*>(rl + (trow0 + l * TROWS) * LD + 8 * tcc) = treg[l];
#pragma unroll
        for (int l = 0; l < NSL; ++l) {
            if constexpr (RS16) {
                *reinterpret_cast<u32x4_t*>(rl + (trow0 + l * TROWS) * LD + 2 * CP + 8 * tcc) = __builtin_bit_cast(u32x4_t, sreg[l]);
            } else {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
                const u32x2_t v = {pk_bf16(sreg[l][0], sreg[l][1]), pk_bf16(sreg[l][2], sreg[l][3])};
                *reinterpret_cast<u32x2_t*>(rl + (srow0 + l * SROWS) * LD + 2 * CP + 4 * scc) = v;
            }
        }
        // zero_src: this kernel is the only reader of r_src, so it can hand the buffer back ZEROED for the next edge
        // pass (which accumulates into it with atomics) — the rows are in registers at this point, the stores ride behind
        // the loads, and the caller's 100-MB fill per layer disappears (mdl_cgconv_bwd_node_z)
        if (zero_src) {
            const int64_t rem = N - nb;
            const __amdgpu_buffer_rsrc_t sz = RS16 ? rsrc(reinterpret_cast<const bf16_t*>(r_src) + nb * (2 * CP), rem * (2 * CP * 2))
                                                   : rsrc(r_src + nb * (2 * CP), rem * (2 * CP * 4));
#pragma unroll
            for (int l = 0; l < NSL; ++l)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, sz,
                                                       RS16 ? to + opaque(l * (TROWS * 2 * CP * 2)) : so + opaque(l * (SROWS * 2 * CP * 4)), 0, 0);
        }
#pragma unroll
        for (int l = 0; l < NXL; ++l) {
            *reinterpret_cast<u32x4_t*>(xl + (xrow0 + l * XROWS) * LX + 8 * xcc) = xreg[l];   // rows past N are zeros
        }
        __syncthreads();
        // ---- dx block (mt, nt) of this wave: rows = 32 nodes, K = 4Cp, cols = 32 features.  The grad_out loads go
        // out BEFORE the next tile's prefetch (loads complete in order: behind it they would wait for the whole tile)
        const bool dxw = wv < 2 * NT;
        const int mt0 = wv / NT, nt0 = wv - mt0 * NT;
        const int64_t remg = N - nb - mt0 * 32;
        const int go = (4 * h * CP + nt0 * 32 + i) * 2;
        const bool more = tile + gridDim.x < n_tiles;
        if (!dxw) {
            if (more) load_tile(tile + gridDim.x);                     // next tile's loads fly during this tile's MFMAs
        } else {
            short gv[16];
            const __amdgpu_buffer_rsrc_t gr = rsrc(gout + (nb + mt0 * 32) * CP, (remg > 0 ? remg : 0) * (CP * 2));
#pragma unroll
            for (int r = 0; r < 16; ++r) gv[r] = __builtin_amdgcn_raw_buffer_load_b16(gr, go + ((r & 3) + 8 * (r >> 2)) * CP * 2, 0, 0);
            if (more) load_tile(tile + gridDim.x);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K4 / 16; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(rl + (mt0 * 32 + i) * LD + 16 * kk + 8 * h);
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(wl + (nt0 * 32 + i) * LD + 16 * kk + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t dr = rsrc(dx + (nb + mt0 * 32) * CP, (remg > 0 ? remg : 0) * (CP * 2));
#pragma unroll
            for (int r = 0; r < 16; ++r)                                 // rows past N fall outside the resource: dropped
                __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(bf2f((bf16_t)gv[r]) + acc[r]), dr, go + ((r & 3) + 8 * (r >> 2)) * CP * 2, 0, 0);
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
        for (int r = 0; r < 16; ++r) {
            const int R = mt * 32 + d_row(r, h), K = nt * 32 + i;           // row of [r_tgt | r_src]^T x (blocks f_tgt, s_tgt, f_src, s_src), column of x
            // ld_out > 0 (MdlCgNode.ld_dwn): straight into the two Linears' stacked weight gradient [2C][ld_out] (rows f | s, columns
            // target | source | edge): block b of rows goes to rows (b & 1) C + c, columns (b >> 1) C + K — no assembly pass
            const int64_t at = ld_out > 0 ? (int64_t)(((R / CP) & 1) * CP + R % CP) * ld_out + ((R / CP) >> 1) * CP + K : (int64_t)R * CP + K;
#ifdef MDL_NODE_NOFLUSH      // experiment builds only: what the atomics of the flush cost
            if (N < 0)
#endif
            unsafeAtomicAdd(dwn + at, dw[j][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// x3 (MDL_SPLIT_BF16): the same pass on fp32 STORAGE with split-bf16 products (round 6).  r_tgt, r_src, x, grad_out and dx
// are fp32; every tile row is split into a (hi, lo) bf16 pair while it is staged (v = hi + lo to 16 significant bits), the
// LDS holds a hi and a lo copy of the R tile, the x tile and Wn^T (154 KB at C = 64: one 8-wave workgroup per CU), and every
// product of the bf16 kernel runs three times: hi hi + lo hi + hi lo, fp32 accumulation.  Replaces, for the parity mode, the
// three exact-fp32 library GEMMs + the concatenation of [r_tgt | r_src] (0.95 ms per layer at N = 2.1e5; fp32 matrix rate =
// 1/16 of bf16) by one HBM-bound pass: N (2 Cp 4 + 2 Cp 4 + 3 C 4) bytes.
// ------------------------------------------------------------------------------------------
template <int CP, int NW>
__global__ __launch_bounds__(64 * NW, 1) void cgconv_node_x3_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                                    const float* __restrict__ r_tgt, float* __restrict__ r_src,
                                                                    const float* __restrict__ wn_t, float* __restrict__ dx,
                                                                    float* __restrict__ dwn, int64_t N, int zero_src, int ld_out) {
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((address_space(3))) s16x4* lds4_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    constexpr int TN = 64, K4 = 4 * CP, LD = K4 + 8, LX = CP + 8, NT = CP / 32, NTH = 64 * NW;
    constexpr int RCH = 2 * CP / 4;          // 16-byte chunks (4 floats) per r_tgt / r_src row
    constexpr int XCH = CP / 4;              // ... per x row
    constexpr int NRL = TN * RCH / NTH;      // chunks per thread and array: 4 (CP 64, 8 waves)
    constexpr int NXL = TN * XCH / NTH;      // 2
    constexpr int MJ = (K4 / 32) * NT / NW;  // dWn blocks per wave: 2
    static_assert(NRL >= 1 && NXL >= 1 && MJ >= 1 && NTH % RCH == 0 && NTH % XCH == 0 && 2 * NT <= NW, "shape / wave count");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl_h = reinterpret_cast<bf16_t*>(smem);      // Wn^T hi / lo [CP][LD]
    bf16_t* wl_l = wl_h + CP * LD;
    bf16_t* rl_h = wl_l + CP * LD;                       // R tile hi / lo [TN][LD]  (columns: r_tgt | r_src)
    bf16_t* rl_l = rl_h + TN * LD;
    bf16_t* xl_h = rl_l + TN * LD;                       // x tile hi / lo [TN][LX]
    bf16_t* xl_l = xl_h + TN * LX;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    auto split4 = [](const f32x4& v, u32x2_t& hi, u32x2_t& lo) {
        const unsigned ha = pk_bf16(v[0], v[1]), hb = pk_bf16(v[2], v[3]);
        hi = u32x2_t{ha, hb};
        lo = u32x2_t{pk_bf16(v[0] - __builtin_bit_cast(float, ha << 16), v[1] - __builtin_bit_cast(float, ha & 0xffff0000u)),
                     pk_bf16(v[2] - __builtin_bit_cast(float, hb << 16), v[3] - __builtin_bit_cast(float, hb & 0xffff0000u))};
    };
    for (int q = tid; q < CP * (K4 / 4); q += NTH) {          // Wn^T (fp32 [CP][K4]) -> hi / lo LDS copies
        const int row = q / (K4 / 4), c4 = q - row * (K4 / 4);
        u32x2_t hi, lo;
        split4(*reinterpret_cast<const f32x4*>(wn_t + row * K4 + c4 * 4), hi, lo);
        *reinterpret_cast<u32x2_t*>(wl_h + row * LD + c4 * 4) = hi;
        *reinterpret_cast<u32x2_t*>(wl_l + row * LD + c4 * 4) = lo;
    }
    f32x16 dw[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dw[j][r] = 0.0f;

    const int64_t n_tiles = (N + TN - 1) / TN;
    f32x4 treg[NRL], sreg[NRL], xreg[NXL];
    constexpr int RROWS = NTH / RCH, XROWS = NTH / XCH;
    const int rrow0 = tid / RCH, rcc = tid % RCH, xrow0 = tid / XCH, xcc = tid % XCH;
    const int ro = (rrow0 * (2 * CP) + 4 * rcc) * 4, xo = (xrow0 * CP + 4 * xcc) * 4;
    auto rsrc = [](const void* base, int64_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0 ? 0 : (bytes < 0x7fffffffLL ? bytes : 0x7fffffffLL)), 0x00020000);
    };
    auto opaque = [](int v) { asm volatile("" : "+s"(v)); return v; };
    auto load_tile = [&](int64_t tile) {
        const int64_t nb = tile * TN, rem = N - nb;
        const __amdgpu_buffer_rsrc_t tr = rsrc(r_tgt + nb * (2 * CP), rem * (2 * CP * 4));
        const __amdgpu_buffer_rsrc_t sr = rsrc(r_src + nb * (2 * CP), rem * (2 * CP * 4));
        const __amdgpu_buffer_rsrc_t xr = rsrc(x + nb * CP, rem * (CP * 4));
#pragma unroll
        for (int l = 0; l < NRL; ++l) treg[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tr, ro + opaque(l * (RROWS * 2 * CP * 4)), 0, 0));
#pragma unroll
        for (int l = 0; l < NRL; ++l) sreg[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(sr, ro + opaque(l * (RROWS * 2 * CP * 4)), 0, 0));
#pragma unroll
        for (int l = 0; l < NXL; ++l) xreg[l] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, xo + opaque(l * (XROWS * CP * 4)), 0, 0));
    };
    int64_t tile = blockIdx.x;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        const int64_t nb = tile * TN;
        __syncthreads();                                   // everyone is done reading the previous tile (and Wn is in)
#pragma unroll
        for (int l = 0; l < NRL; ++l) {
            u32x2_t hi, lo;
            split4(treg[l], hi, lo);
            *reinterpret_cast<u32x2_t*>(rl_h + (rrow0 + l * RROWS) * LD + 4 * rcc) = hi;
            *reinterpret_cast<u32x2_t*>(rl_l + (rrow0 + l * RROWS) * LD + 4 * rcc) = lo;
            split4(sreg[l], hi, lo);
            *reinterpret_cast<u32x2_t*>(rl_h + (rrow0 + l * RROWS) * LD + 2 * CP + 4 * rcc) = hi;
            *reinterpret_cast<u32x2_t*>(rl_l + (rrow0 + l * RROWS) * LD + 2 * CP + 4 * rcc) = lo;
        }
        if (zero_src) {                                    // hand r_src back zeroed for the next edge pass (its only reader is here)
            const __amdgpu_buffer_rsrc_t sz = rsrc(r_src + nb * (2 * CP), (N - nb) * (2 * CP * 4));
#pragma unroll
            for (int l = 0; l < NRL; ++l)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{0u, 0u, 0u, 0u}, sz, ro + opaque(l * (RROWS * 2 * CP * 4)), 0, 0);
        }
#pragma unroll
        for (int l = 0; l < NXL; ++l) {
            u32x2_t hi, lo;
            split4(xreg[l], hi, lo);                       // rows past N are zeros
            *reinterpret_cast<u32x2_t*>(xl_h + (xrow0 + l * XROWS) * LX + 4 * xcc) = hi;
            *reinterpret_cast<u32x2_t*>(xl_l + (xrow0 + l * XROWS) * LX + 4 * xcc) = lo;
        }
        __syncthreads();
        // ---- dx block (mt0, nt0) of waves 0 .. 2 NT - 1: rows = 32 nodes, K = 4Cp, cols = 32 features
        const bool dxw = wv < 2 * NT;
        const int mt0 = wv / NT, nt0 = wv - mt0 * NT;
        const int64_t remg = N - nb - mt0 * 32;
        const int go = (4 * h * CP + nt0 * 32 + i) * 4;
        const bool more = tile + gridDim.x < n_tiles;
        if (!dxw) {
            if (more) load_tile(tile + gridDim.x);
        } else {
            float gv[16];
            const __amdgpu_buffer_rsrc_t gr = rsrc(gout + (nb + mt0 * 32) * CP, (remg > 0 ? remg : 0) * (CP * 4));
#pragma unroll
            for (int r = 0; r < 16; ++r)
                gv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, go + ((r & 3) + 8 * (r >> 2)) * CP * 4, 0, 0));
            if (more) load_tile(tile + gridDim.x);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K4 / 16; ++kk) {
                const int ao = (mt0 * 32 + i) * LD + 16 * kk + 8 * h, bo = (nt0 * 32 + i) * LD + 16 * kk + 8 * h;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(rl_h + ao), al = *reinterpret_cast<const bf16x8*>(rl_l + ao);
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wl_h + bo), bl = *reinterpret_cast<const bf16x8*>(wl_l + bo);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t dr = rsrc(dx + (nb + mt0 * 32) * CP, (remg > 0 ? remg : 0) * (CP * 4));
#pragma unroll
            for (int r = 0; r < 16; ++r)                                 // rows past N fall outside the resource: dropped
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, gv[r] + acc[r]), dr, go + ((r & 3) + 8 * (r >> 2)) * CP * 4, 0, 0);
        }
        // ---- dWn blocks of this wave: rows = 32 columns of R, cols = 32 features, K = the tile's 64 nodes
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int blk = wv * MJ + j, mt = blk / NT, nt = blk - mt * NT;
            const int t = i & 15;
#pragma unroll
            for (int ks = 0; ks < TN / 16; ++ks) {
                const int pa = (16 * ks + 8 * h + (t >> 2)) * LD + mt * 32 + (i & 16) + 4 * (t & 3);
                const int pb = (16 * ks + 8 * h + (t >> 2)) * LX + nt * 32 + (i & 16) + 4 * (t & 3);
                auto frag = [&](const bf16_t* base, int off, int ld) {
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base + off));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base + off + 4 * ld));
                    return bf16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                };
                const bf16x8 ah = frag(rl_h, pa, LD), al = frag(rl_l, pa, LD), bh = frag(xl_h, pb, LX), bl = frag(xl_l, pb, LX);
                dw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, dw[j], 0, 0, 0);
                dw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, dw[j], 0, 0, 0);
                dw[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, dw[j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int blk = wv * MJ + j, mt = blk / NT, nt = blk - mt * NT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int R = mt * 32 + d_row(r, h), K = nt * 32 + i;
            const int64_t at = ld_out > 0 ? (int64_t)(((R / CP) & 1) * CP + R % CP) * ld_out + ((R / CP) >> 1) * CP + K : (int64_t)R * CP + K;
#ifdef MDL_NODE_NOFLUSH      // experiment builds only: what the atomics of the flush cost
            if (N < 0)
#endif
            unsafeAtomicAdd(dwn + at, dw[j][r]);
        }
    }
}

}  // namespace mdl

namespace mdl {
// wn_t [C][4Cp] (dtype bf16) = transpose of Wn = rows (f_tgt, s_tgt, f_src, s_src) of the two Linears' node columns
template <typename TO>
__global__ __launch_bounds__(256) void cgconv_pack_node_kernel(const float* __restrict__ wf, const float* __restrict__ ws, int C,
                                                               int Cp, int ldw, TO* __restrict__ wn_t) {
    const int total = C * 4 * Cp;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const int k = q / (4 * Cp), r = q - k * (4 * Cp);          // wn_t[k][r] = Wn[r][k]
        const int blk = r / Cp, c = r - blk * Cp;                  // blk: 0 f_tgt, 1 s_tgt, 2 f_src, 3 s_src
        float v = 0.0f;
        if (c < C) v = ((blk & 1) ? ws : wf)[c * ldw + (blk >> 1) * C + k];
        if constexpr (std::is_same<TO, float>::value) wn_t[q] = v; else wn_t[q] = f2bf(v);
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
    // MDL_F32: the fp32 transpose the split-product node kernel (MDL_SPLIT_BF16) splits itself
    MDL_REQUIRE(dtype == MDL_BF16 || dtype == MDL_F32, MDL_E_UNSUPP, "mdl_cgconv_pack_node_weights: dtype must be MDL_BF16 or MDL_F32");
    MDL_REQUIRE(w_f && w_s && wn_t && C >= 1 && G >= 0, MDL_E_ARG, "mdl_cgconv_pack_node_weights: bad arguments");
    const int Cp = (C + 31) / 32 * 32;
    if (dtype == MDL_F32)
        hipLaunchKernelGGL(cgconv_pack_node_kernel<float>, dim3((unsigned)cdiv((int64_t)C * 4 * Cp, 256)), dim3(256), 0, (hipStream_t)stream,
                           w_f, w_s, C, Cp, 2 * C + G, (float*)wn_t);
    else
    hipLaunchKernelGGL(cgconv_pack_node_kernel<bf16_t>, dim3((unsigned)cdiv((int64_t)C * 4 * Cp, 256)), dim3(256), 0, (hipStream_t)stream, w_f,
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

static int bwd_node_launch(const void* x, const void* grad_out, const void* r_tgt, float* r_src, const void* wn_t, void* dx,
                           float* dwn, int64_t N, int C, int dtype, int zero_src, bool rs16, int ld_out, mdlStream_t stream);

extern "C" int mdl_cgconv_bwd_node_ex(const MdlCgNode* a, mdlStream_t stream) {
    MDL_REQUIRE(a && a->size == sizeof(MdlCgNode), MDL_E_ARG, "mdl_cgconv_bwd_node_ex: argument struct of another layout (size %u, expected %u)",
                a ? a->size : 0u, (unsigned)sizeof(MdlCgNode));
    MDL_REQUIRE((a->flags & ~(uint32_t)(MDL_DETERMINISTIC | MDL_SPLIT_BF16)) == 0, MDL_E_ARG, "mdl_cgconv_bwd_node_ex: unknown flag bits %#x", a->flags);
    MDL_REQUIRE(a->r_src_dtype == MDL_F32 || a->r_src_dtype == MDL_BF16, MDL_E_ARG, "mdl_cgconv_bwd_node_ex: r_src_dtype must be MDL_F32 or MDL_BF16");
    if (a->flags & MDL_SPLIT_BF16) {
        // split-product node kernel: fp32 x / grad_out / r_tgt / r_src / dx, wn_t = the fp32 transpose (mdl_cgconv_pack_node_weights, MDL_F32)
        using namespace mdl;
        MDL_REQUIRE(a->dtype == MDL_F32 && a->r_src_dtype == MDL_F32 && a->C == 64 && !(a->flags & MDL_DETERMINISTIC), MDL_E_UNSUPP,
                    "mdl_cgconv_bwd_node_ex: MDL_SPLIT_BF16 needs MDL_F32 storage, fp32 by-source sums, C = 64 (C=%d dtype=%d)", a->C, a->dtype);
        MDL_REQUIRE(a->N >= 0 && (a->N == 0 || (a->x && a->grad_out && a->r_tgt && a->r_src && a->wn_t && a->dx && a->dwn)), MDL_E_ARG,
                    "mdl_cgconv_bwd_node_ex: bad arguments");
        MDL_REQUIRE(reinterpret_cast<uintptr_t>(a->wn_t) % 16 == 0 && reinterpret_cast<uintptr_t>(a->r_tgt) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(a->r_src) % 16 == 0 && reinterpret_cast<uintptr_t>(a->x) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(a->grad_out) % 4 == 0 && reinterpret_cast<uintptr_t>(a->dx) % 4 == 0, MDL_E_ARG,
                    "mdl_cgconv_bwd_node_ex: 16-byte alignment required");
        MDL_REQUIRE(a->ld_dwn == 0 || a->ld_dwn >= 2 * a->C, MDL_E_ARG, "mdl_cgconv_bwd_node_ex: ld_dwn (%d) below 2 C", a->ld_dwn);
        if (a->N == 0) return MDL_OK;
        constexpr int NWX = 8;
        const int lds = (2 * 64 * (256 + 8) + 2 * 64 * (256 + 8) + 2 * 64 * (64 + 8)) * 2;
        int64_t sgrid = cdiv(a->N, 64);
        if (sgrid > 256) sgrid = 256;
        auto kf = cgconv_node_x3_kernel<64, NWX>;
        hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        if (e != hipSuccess) { set_error("mdl_cgconv_bwd_node_ex: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
        hipLaunchKernelGGL(kf, dim3((unsigned)sgrid), dim3(64 * NWX), lds, (hipStream_t)stream, (const float*)a->x, (const float*)a->grad_out,
                           (const float*)a->r_tgt, static_cast<float*>(a->r_src), (const float*)a->wn_t, (float*)a->dx, a->dwn, a->N, a->zero_src, a->ld_dwn);
        return check_launch("mdl_cgconv_bwd_node_ex");
    }
    return bwd_node_launch(a->x, a->grad_out, a->r_tgt, static_cast<float*>(a->r_src), a->wn_t, a->dx, a->dwn, a->N, a->C,
                           a->dtype | (int)a->flags, a->zero_src, a->r_src_dtype == MDL_BF16, a->ld_dwn, stream);
}

static int bwd_node_launch(const void* x, const void* grad_out, const void* r_tgt, float* r_src, const void* wn_t, void* dx,
                           float* dwn, int64_t N, int C, int dtype, int zero_src, bool rs16, int ld_out, mdlStream_t stream) {
    using namespace mdl;
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;      // one workgroup: every dwn element gets one add from one wave
    dtype &= MDL_DTYPE_MASK;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: bf16 only (fp32 parity mode uses library GEMMs)");
    MDL_REQUIRE(C == 32 || C == 64, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: C must be 32 or 64 (got %d)", C);
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && grad_out && r_tgt && r_src && wn_t && dx && dwn)), MDL_E_ARG,
                "mdl_cgconv_bwd_node: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(wn_t) % 16 == 0 && reinterpret_cast<uintptr_t>(r_tgt) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(r_src) % 16 == 0, MDL_E_ARG, "mdl_cgconv_bwd_node: 16-byte alignment required");
    MDL_REQUIRE(ld_out == 0 || ld_out >= 2 * C, MDL_E_ARG, "mdl_cgconv_bwd_node_ex: ld_dwn (%d) below 2 C", ld_out);
    if (N == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    int64_t sgrid = cdiv(N, 64);
    if (det) sgrid = 1;
    if (sgrid > 256) sgrid = 256;         // one block per CU: each flushes 4Cp*C atomics on the same addresses at the end
                                          // (measured 128 / 256 / 512 / 1024 blocks: 70 / 56 / 67 / 97 us)
    if (C == 64) {
        const int lds = (64 * (256 + 8) + 64 * (256 + 8) + 64 * (64 + 8)) * 2;
        auto kf = rs16 ? cgconv_node_stream_kernel<64, true, MDL_K3C_NW> : cgconv_node_stream_kernel<64, false, MDL_K3C_NW>;
        set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)sgrid), dim3(64 * MDL_K3C_NW), lds, st, (const bf16_t*)x, (const bf16_t*)grad_out,
                           (const bf16_t*)r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N, zero_src, ld_out);
    } else {
        const int lds = (32 * (128 + 8) + 64 * (128 + 8) + 64 * (32 + 8)) * 2;
        auto kf = rs16 ? cgconv_node_stream_kernel<32, true> : cgconv_node_stream_kernel<32, false>;
        hipLaunchKernelGGL(kf, dim3((unsigned)sgrid), dim3(256), lds, st, (const bf16_t*)x, (const bf16_t*)grad_out,
                           (const bf16_t*)r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N, zero_src, ld_out);
    }
    return check_launch("mdl_cgconv_bwd_node");
}
