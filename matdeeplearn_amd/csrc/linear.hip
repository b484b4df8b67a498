// linear.hip — node-level dense layer forward, fused:  out[N, M] = act(x[N, K] . W[M, K]^T + bias)   (bf16 in / out).
// Replaces `getattr(F, act)(lin(out))` of the reference's pre-FC / post-FC loops
// (/root/reference/matdeeplearn/models/cgcnn.py:124-130,155-166) for the tall-skinny shapes of this path (N = 2e5 nodes
// or 8e3 graphs, K <= 256, M <= 128): the library picks a 64x64x32 macro tile for them (69 us for 2e5 x 114 x 64, plus
// 12 us for the separate activation), while the layer is a stream: N*(K + M)*2 bytes, read / written once.
//
// Workgroup = 4 waves = 64 rows per step of a grid-stride loop.  W (all of it) sits in LDS for the whole kernel, rows
// padded to an odd number of 16-byte slots.  The x tile is fetched one tile ahead with coalesced dword buffer loads
// (wave = 16 rows, lane = dword of the row; range = the bytes left in the array, so the last tile needs no clamps and
// reads zeros past the end) and written to a padded LDS tile.  MFMA 32x32x16: A = x rows, B = W rows (both row-wise
// ds_read_b128 fragments), K zero-padded to a multiple of 32 in LDS; a wave owns one 32-row block row and every second
// 32-column block, so an A fragment is read once per k-step for all of them; bias + activation on the accumulators.
#include "mdl_common.h"

#ifndef MDL_EXPERIMENTS
#define MDL_EXPERIMENTS 0
#endif
#include <type_traits>

#ifndef MDL_LIN_GRID
#define MDL_LIN_GRID 512        // workgroups of a launch (2 per CU)
#endif
// waves per workgroup: 8 (each wave stages 8 rows of the 64-row tile and owns half as many output blocks; two workgroups
// still share a CU) except for KP = 160, whose 75 KB of LDS and wider rows leave the 8-wave form one workgroup per CU
// (1.5e6 rows, tools/bench_dense.py: 100 -> 100 137 -> 124 us, 114 -> 64 109 -> 98, 50 -> 150 209 -> 190, the wide layer of
// MPNN -8 %; 150 -> 150 217 -> 263 us)
// The gathering forms (K6) keep 4 as well: with the table rows in flight they need 137-148 registers, so an 8-wave workgroup
// would be alone on its CU (228 -> ~275 us on the MEGNet bench batch).
#ifndef MDL_WIDE_NT
#define MDL_WIDE_NT 6      // mdl_linear_wide: 32-column blocks per workgroup — 192 columns = 384 B of a bf16 output row: whole 128-byte
                           // lines, so no line is shared by workgroups of two XCDs (5 -> 6: 513 -> 456 us for NNConv's Y; 4: 506, 8: 494)
#endif
#define MDL_LIN_NW(KP_, G_) (((KP_) == 160 || (G_) != 0) ? 4 : 8)

namespace mdl {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_l;

// K6 (MEGNet edge / node blocks, megnet.py:41-56,84-101): the first layer of those MLPs acts on a concatenation of
// GATHERED rows — [x[row] | x[col] | e | u[batch]] — which the reference materialises as an [E, 4d] tensor.  Split by
// column blocks of the weight, W [x[row] | x[col] | e | u[b]]^T = e Wc^T + (x Wa^T)[row] + (x Wb^T)[col] + (u Wd^T + b)[b]:
// three small per-node / per-graph projections (done by the caller) and ONE streaming GEMM over the edge state, whose
// epilogue adds the three gathered projection rows before the activation.  The concatenation never exists and the dense
// product shrinks from K = 4d to K = d.
struct GatherAdd {
    const bf16_t* p[3];        // projection tables [rows_i, M] (NULL = unused)
    const int32_t* idx[3];     // row of table i for every row of x
};

// XACT (mdl_linear_act_in): the input rows are a gradient w.r.t. the OUTPUT y of an activated layer and the product wanted
// is the one with x .* act'(y) (relu: y > 0, shifted softplus: 1 - exp(-(y + ln 2))) — the dX product of a fused
// Linear + activation, with the activation derivative applied while the tile is staged, like gemm_tn.hip does for dW:
// `threshold_backward` / the softplus backward never run as a pass over [N, K] of their own.
// STATS (mdl_linear_act_stats): the layer feeds a training-mode BatchNorm1d; the per-column sum and sum of squares of the
// ROUNDED outputs are kept per thread over the grid-stride loop and added to one of the MDL_BN_REPLICAS copies of the
// BatchNorm sums at the end, so the statistics pass over [N, M] (mdl_bn_stats) does not run.
template <int KP, int NT, int GATHER = 0, int XACT = 0, bool STATS = false>     // KP: K padded to {64, 128, 160, 256}; NT: 32-column tiles of the output (M <= 32*NT); GATHER: tables
__global__ __launch_bounds__(64 * MDL_LIN_NW(KP, GATHER), 2) void linear_act_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ bias, bf16_t* __restrict__ out,
                                                            int64_t N, int K, int M, int act, GatherAdd ga,
                                                            const bf16_t* __restrict__ xy, int ldo = 0,
                                                            float* __restrict__ stats = nullptr,
                                                            const int64_t* __restrict__ n_dev = nullptr) {
    // mdl_linear_wide: blockIdx.x = block of 32*NT output columns of a wider layer (ldo = its full width, the leading dimension
    // of `out`), blockIdx.y = row chunk; every other caller has gridDim.y = 1, ldo = 0 (= M) and blockIdx.x = row chunk
    // (wide layers are launched with the COLUMN block as the fast grid dimension: the workgroups that run together then cover
    // whole rows of `out` — 20 KB contiguous per row for NNConv's Y — instead of a 320-byte segment of many rows)
    const bool wide = ldo > 0;                       // (only mdl_linear_wide passes ldo)
    unsigned bx = wide ? blockIdx.y : blockIdx.x, cbx = blockIdx.x;
    const unsigned gdx = wide ? gridDim.y : gridDim.x;
    if (wide && (gridDim.y & 7u) == 0u) {
        // XCD-aware tile map: workgroups are dealt to the 8 XCDs round-robin in dispatch order, and the column blocks of one row
        // chunk write NEIGHBOURING segments of the same output rows — with rows that are not a multiple of 128 bytes (NNConv's
        // Y: 20,000 B) most segment boundaries fall inside a cache line, which two XCDs' L2s would each hold half of.  All column
        // blocks of a row chunk therefore go to ONE XCD (consecutive slots there): its L2 merges the shared lines and serves the
        // chunk's x tile to all of them.
        const unsigned f = blockIdx.y * gridDim.x + blockIdx.x, xcd = f & 7u, k = f >> 3;
        cbx = k % gridDim.x;
        bx = xcd + 8u * (k / gridDim.x);
    }
    if (wide) {
        const int c0 = (int)cbx * 32 * NT;
        w += (int64_t)c0 * K;
        if (bias) bias += c0;
        out += c0;
        M = min(M - c0, 32 * NT);
    }
    if (ldo == 0) ldo = M;
    constexpr int TN = 64;
    constexpr int LD = KP + 8;                       // LDS row stride (bf16): odd number of 16-byte slots
    constexpr int NW = MDL_LIN_NW(KP, GATHER), XS = NW / 2, RPW = 64 / NW, NTH = 64 * NW;
    constexpr int NB = (NT + XS - 1) / XS;           // output blocks per wave: block row wv & 1, block columns (wv >> 1) + XS j
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);    // [32*NT][LD]
    bf16_t* xl = wl + 32 * NT * LD;                  // [TN][LD]
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k2 = K >> 1;                           // dwords per row

    // W -> LDS, zero padded to [32*NT][KP]
    for (int q = tid; q < 32 * NT * (KP / 2); q += NTH) {
        const int row = q / (KP / 2), d = q - row * (KP / 2);
        unsigned v = 0u;
        if (row < M && d < k2) v = *reinterpret_cast<const unsigned*>(w + (int64_t)row * K + 2 * d);
        *reinterpret_cast<unsigned*>(wl + row * LD + 2 * d) = v;
    }
    const int mt = wv & 1, ntb = wv >> 1;            // this wave's block row and first block column
    float bv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int col = (ntb + XS * j) * 32 + i;
        bv[j] = (bias && col < M) ? bf2f(bias[col]) : 0.0f;
    }

    // x tile staging (same map as gemm_tn.hip): wave wv stages rows 16*wv .. 16*wv+15; a padded row is DW = KP/2 dwords,
    // dwords 0..64*Q-1 one load per 64 (lane = dword), the remaining R dwords for 64/R rows per load.  Row offsets are in
    // the VGPR offset (range checked): rows past N and columns K..KP-1 read as zeros, so the padding columns of the LDS
    // tile are rewritten with zeros by every tile and nothing needs a clamp.  (The previous version fetched the tile as
    // flat 16-byte chunks and split every dword into (row, column) with a multiply-high: ~8 VALU per dword.)
    constexpr int DW = KP / 2, Q = DW / 64, R = DW % 64, NR = R ? RPW * R / 64 : 0, NLX = RPW * Q + NR;
    constexpr unsigned FAR = 0x40000000u;
    const int w16 = RPW * wv;
    const unsigned xrow = (unsigned)K * 2u;          // dense rows
    const int64_t n_tiles = (N + TN - 1) / TN;
    unsigned xr[NLX], yr[XACT ? NLX : 1];
    auto load_tile = [&](int64_t tile) {
        // (per call: the dword offsets below depend on the lane only — hoisted out of the tile loop they are registers the wider
        // forms of this kernel do not have, and a spilled offset comes back through a scratch load whose wait drains the prefetch)
        // Only the forms with gathered addends / BatchNorm statistics need it (they are the ones that spilled: 3-42 registers,
        // every reload inside the tile loop); on the plain forms the recomputation costs up to 2 % (150 -> 150: 218 -> 222 us).
        int lane = threadIdx.x & 63;
        if constexpr (GATHER != 0 || STATS) asm volatile("" : "+v"(lane));
        const int64_t nb = tile * TN;
        const int64_t bytes = (N - nb) * (int64_t)K * 2, cap = (int64_t)TN * K * 2;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x + nb * (int64_t)K), 0,
                                                                            (int)(bytes < cap ? bytes : cap), 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>((XACT ? xy : x) + nb * (int64_t)K), 0,
                                                                            (int)(bytes < cap ? bytes : cap), 0x00020000);
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < Q; ++j) {
                const int d = lane + 64 * j;
                const unsigned off = (d < k2) ? (unsigned)(w16 + r) * xrow + 4u * d : FAR;
                xr[r * Q + j] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
                if constexpr (XACT != 0) yr[r * Q + j] = __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0);
            }
        if constexpr (R != 0) {
            const int d = 64 * Q + lane % R, rr = lane / R;
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const unsigned off = (d < k2) ? (unsigned)(w16 + k * (64 / R) + rr) * xrow + 4u * d : FAR;
                xr[RPW * Q + k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
                if constexpr (XACT != 0) yr[RPW * Q + k] = __builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0);
            }
        }
    };
    auto xfix = [&](unsigned v, unsigned yv) -> unsigned {
        if constexpr (XACT == 1) {
            return (__uint_as_float(yv << 16) > 0.0f ? (v & 0xffffu) : 0u) | (__uint_as_float(yv & 0xffff0000u) > 0.0f ? (v & 0xffff0000u) : 0u);
        } else if constexpr (XACT == 2) {                                 // same arithmetic as ssp_bwd_kernel (gather.hip)
            const float s0 = 1.0f - __expf(-(__uint_as_float(yv << 16) + 0.6931471805599453f));
            const float s1 = 1.0f - __expf(-(__uint_as_float(yv & 0xffff0000u) + 0.6931471805599453f));
            return pk_bf16(__uint_as_float(v << 16) * s0, __uint_as_float(v & 0xffff0000u) * s1);
        } else {
            return v;
        }
    };
    float st0[STATS ? NB : 1], st1[STATS ? NB : 1], shv[STATS ? NB : 1];
    int64_t n_true = N;
    if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < NB; ++j) { st0[j] = 0.0f; st1[j] = 0.0f; }
        if (n_dev) n_true = max((int64_t)1, min(*n_dev, N));
        // The sums are formed about a per-column SHIFT near the column mean — E[v^2] - E[v]^2 in fp32 cancels for a post-ReLU
        // column whose mean is far above its spread (mdl_bn_stats shifts by the first row for the same reason).  Every
        // workgroup evaluates output row 0 for itself (one K-long dot product per column from the global weights: same
        // instructions, same bits everywhere), workgroup 0 publishes it behind the totals rows of the sums for
        // mdl_bn_apply_n(... | MDL_BN_SHIFT_ROW).  The x tile's LDS area is free until the first tile is staged.
        float* shl = reinterpret_cast<float*>(xl);
        if (tid < M) {
            float a = bias ? bf2f(bias[tid]) : 0.0f;
            const unsigned* wr0 = reinterpret_cast<const unsigned*>(w + (int64_t)tid * K);
            const unsigned* xr0 = reinterpret_cast<const unsigned*>(x);
            for (int d = 0; d < k2; ++d) {
                const unsigned wv2 = wr0[d], xv2 = xr0[d];
                a = fmaf(__uint_as_float(wv2 << 16), __uint_as_float(xv2 << 16), a);
                a = fmaf(__uint_as_float(wv2 & 0xffff0000u), __uint_as_float(xv2 & 0xffff0000u), a);
            }
            if constexpr (GATHER != 0) {
#pragma unroll
                for (int t = 0; t < GATHER; ++t) a += bf2f(ga.p[t][(int64_t)ga.idx[t][0] * M + tid]);
            }
            if (act == 1) a = a > 0.0f ? a : 0.0f;
            else if (act == 2) a = fmaf(0.5f, a + fabsf(a), fmaf(LN2_F, __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-LOG2E_F * fabsf(a))), -LN2_F));
            a = bf2f(f2bf(a));
            shl[tid] = a;
            if (bx == 0) stats[(size_t)(2 * MDL_BN_REPLICAS + 2) * M + tid] = a;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = (ntb + XS * j) * 32 + i;
            shv[j] = col < M ? shl[col] : 0.0f;
        }
        // (the first barrier of the tile loop stands between these reads and the staging writes)
    }
    int64_t tile = bx;
    if (tile < n_tiles) load_tile(tile);
    for (; tile < n_tiles; tile += gdx) {
        const int64_t nb = tile * TN;
        // K6: the row indices of this tile's gathered table rows are requested here, a whole staging + MFMA phase ahead of the
        // epilogue that uses them (they depend on the row only: one coalesced load per table for all of the wave's block
        // columns), so a block column pays ONE dependent round trip (its table rows) instead of two
        int gidl[GATHER ? GATHER : 1];                      // lane l: the index of row (l & 31) of this wave's block row
        if constexpr (GATHER != 0) {
#pragma unroll
            for (int t = 0; t < GATHER; ++t) gidl[t] = ga.idx[t][min(nb + (wv & 1) * 32 + i, N - 1)];
        }
        __syncthreads();                                    // previous tile's fragments read; W in place
#pragma unroll
        for (int r = 0; r < RPW; ++r)
#pragma unroll
            for (int j = 0; j < Q; ++j)
                *reinterpret_cast<unsigned*>(xl + (w16 + r) * LD + 2 * (lane + 64 * j)) = xfix(xr[r * Q + j], yr[XACT ? r * Q + j : 0]);
        if constexpr (R != 0) {
#pragma unroll
            for (int k = 0; k < NR; ++k)
                *reinterpret_cast<unsigned*>(xl + (w16 + k * (64 / R) + lane / R) * LD + 2 * (64 * Q + lane % R)) =
                    xfix(xr[RPW * Q + k], yr[XACT ? RPW * Q + k : 0]);
        }
        __syncthreads();
        if (tile + gdx < n_tiles) load_tile(tile + gdx);      // next tile's loads fly during the MFMAs
        if (ntb < NT) {                                                    // (NT == 1: waves 2, 3 have no block)
            // blocks (mt, ntb + 2j): one A fragment (x rows) per k-step feeds all of the wave's block columns
            f32x16 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = bv[j];
#pragma unroll
            for (int kk = 0; kk < KP / 16; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(xl + (mt * 32 + i) * LD + 16 * kk + 8 * h);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int nt = min(ntb + XS * j, NT - 1);              // (a block column past NT repeats the last one — no
                    const bf16x8 b = *reinterpret_cast<const bf16x8*>(wl + (nt * 32 + i) * LD + 16 * kk + 8 * h);   // branch in the
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);                       // MFMA chain — and is not stored)
                }
            }
            const int64_t remr = N - nb - mt * 32;                        // rows of this block row that exist
            if constexpr (GATHER != 0) {
                // + the gathered projection rows (clamped row index: rows past N are computed on a valid row and then dropped by
                // the store's range check): all table loads of a block column go out before the first add
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int colc = min(min(ntb + XS * j, NT - 1) * 32 + i, M - 1);
                    float gv[GATHER][16];
#pragma unroll
                    for (int t = 0; t < GATHER; ++t) {
                        // (32-bit byte offsets into the table through a buffer resource: one address register per load — as 64-bit
                        // pointers the 16 x GATHER x NB addresses in flight spilled; tables are < 2 GB by the launcher's check)
                        const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(ga.p[t]), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            // (the row's index sits in lane `row` of gidl: a cross-lane read instead of 16 registers per table)
                            const int id = __builtin_amdgcn_ds_bpermute(4 * ((r & 3) + 8 * (r >> 2) + 4 * h), gidl[t]);
                            gv[t][r] = bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(tr, (id * M + colc) * 2, 0, 0));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < GATHER; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] += gv[t][r];
                    __builtin_amdgcn_sched_barrier(0);      // (the next column's loads stay behind these adds: both columns' rows in flight spill)
                }
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nt = ntb + XS * j, col = nt * 32 + i;
                if (nt < NT) {
                    if (col < M && remr > 0) {
                        // (range = up to the end of this block row's last existing row; with column blocks the rows are ldo wide)
                        const int64_t rbytes = ((remr - 1) * (int64_t)ldo + M) * 2;
                        const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
                            out + (nb + mt * 32) * (int64_t)ldo, 0,
                            (int)(rbytes < 0x7fffffffLL ? rbytes : 0x7fffffffLL), 0x00020000);
                        const int vo = (4 * h * ldo + col) * 2;
                        // (2-byte stores, 16 per block: pairing neighbouring lanes' columns into dword stores — half the store
                        // instructions — measured no faster; these streams run at 4.2-4.9 TB/s of the ~6.3 TB/s a copy reaches)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[j][r];
                            if (act == 1) v = v > 0.0f ? v : 0.0f;
                            else if (act == 2) {    // shifted softplus = max(v,0) + ln2 * (log2(1 + 2^(-|v| log2 e)) - 1): two transcendentals, 5 VALU
                                const float l = __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-LOG2E_F * fabsf(v)));
                                v = fmaf(0.5f, v + fabsf(v), fmaf(LN2_F, l, -LN2_F));
                            }
                            const bf16_t vb = f2bf(v);
                            if constexpr (STATS) {
                                const float vr = (nb + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h < n_true) ? bf2f(vb) - shv[j] : 0.0f;
                                st0[j] += vr;
                                st1[j] = fmaf(vr, vr, st1[j]);
                            }
                            __builtin_amdgcn_raw_buffer_store_b16((short)vb, os, vo + ((r & 3) + 8 * (r >> 2)) * ldo * 2, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        if (ntb < NT) {
            float* dst = stats + (size_t)(bx % MDL_BN_REPLICAS) * 2 * M;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int nt = ntb + XS * j, col = nt * 32 + i;
                const float t0 = st0[j] + __shfl_xor(st0[j], 32), t1 = st1[j] + __shfl_xor(st1[j], 32);
                if (nt < NT && col < M && h == 0) {
                    unsafeAtomicAdd(dst + col, t0);
                    unsafeAtomicAdd(dst + M + col, t1);
                }
            }
        }
    }
}


#if MDL_EXPERIMENTS
#include "../../experiments/csrc/mlp2.inc"   // two chained dense layers in one launch: measured slower than two launches
#endif

}  // namespace mdl

extern "C" int mdl_linear_act(const void* x, const void* w, const void* bias, void* out, int64_t N, int K, int M, int act,
                              int dtype, mdlStream_t stream) {
    return mdl_linear_gather_act(x, w, bias, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out, N, K, M, act, dtype, stream);
}

static int linear_launch(const void* x, const void* xy, int xact, const void* w, const void* bias, const mdl::GatherAdd& ga,
                         int gather, void* out, int64_t N, int K, int M, int act, mdlStream_t stream, float* stats = nullptr,
                         const int64_t* n_dev = nullptr);

extern "C" int mdl_linear_act_in(const void* x, const void* y, int xact, const void* w, const void* bias, void* out, int64_t N,
                                 int K, int M, int act, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_linear_act_in: bf16 only");
    MDL_REQUIRE(xact >= 0 && xact <= 2 && (xact == 0 || (y && reinterpret_cast<uintptr_t>(y) % 4 == 0)), MDL_E_ARG,
                "mdl_linear_act_in: xact must be 0, 1 (relu) or 2 (shifted softplus), with the activated output y for 1 / 2");
    MDL_REQUIRE(K >= 4 && K <= 256 && K % 2 == 0 && M >= 1 && M <= 160 && (M <= 128 || K <= 160), MDL_E_UNSUPP,
                "mdl_linear_act: need even 4<=K<=256 and 1<=M<=160 (K<=160 when M>128) (got K=%d M=%d)", K, M);
    MDL_REQUIRE(act >= 0 && act <= 2, MDL_E_ARG, "mdl_linear_act: act must be 0 (none), 1 (relu) or 2 (shifted softplus)");
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && w && out)), MDL_E_ARG, "mdl_linear_act: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 4 == 0 && reinterpret_cast<uintptr_t>(w) % 4 == 0, MDL_E_ARG, "mdl_linear_act_in: misaligned pointer");
    if (N == 0) return MDL_OK;
    return linear_launch(x, y, xact, w, bias, GatherAdd{}, 0, out, N, K, M, act, stream);
}

extern "C" int mdl_linear_gather_act(const void* x, const void* w, const void* bias, const void* p1, const int32_t* idx1,
                                     const void* p2, const int32_t* idx2, const void* p3, const int32_t* idx3, void* out,
                                     int64_t N, int K, int M, int act, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_linear_act: bf16 only");
    MDL_REQUIRE((!p1 || idx1) && (!p2 || idx2) && (!p3 || idx3), MDL_E_ARG, "mdl_linear_gather_act: table without index");
    MDL_REQUIRE(!(p1 || p2 || p3) || M <= 128, MDL_E_UNSUPP, "mdl_linear_gather_act: gathered tables need M <= 128 (got %d)", M);
    // (the tables present are packed to the front: the kernel is instantiated per table COUNT)
    GatherAdd ga = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int gather = 0;
    if (p1) { ga.p[gather] = (const bf16_t*)p1; ga.idx[gather++] = idx1; }
    if (p2) { ga.p[gather] = (const bf16_t*)p2; ga.idx[gather++] = idx2; }
    if (p3) { ga.p[gather] = (const bf16_t*)p3; ga.idx[gather++] = idx3; }
    MDL_REQUIRE(K >= 4 && K <= 256 && K % 2 == 0 && M >= 1 && M <= 160 && (M <= 128 || K <= 160), MDL_E_UNSUPP,
                "mdl_linear_act: need even 4<=K<=256 and 1<=M<=160 (K<=160 when M>128) (got K=%d M=%d)", K, M);
    MDL_REQUIRE(act >= 0 && act <= 2, MDL_E_ARG, "mdl_linear_act: act must be 0 (none), 1 (relu) or 2 (shifted softplus)");
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && w && out)), MDL_E_ARG, "mdl_linear_act: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 4 == 0 &&
                reinterpret_cast<uintptr_t>(out) % 2 == 0, MDL_E_ARG, "mdl_linear_act: misaligned pointer");
    if (N == 0) return MDL_OK;
    return linear_launch(x, nullptr, 0, w, bias, ga, gather, out, N, K, M, act, stream);
}

extern "C" int mdl_linear_act_stats(const void* x, const void* w, const void* bias, const void* p1, const int32_t* idx1,
                                    const void* p2, const int32_t* idx2, const void* p3, const int32_t* idx3, void* out, int64_t N,
                                    int K, int M, int act, float* bn_sums, const int64_t* n_dev, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_linear_act_stats: bf16 only");
    MDL_REQUIRE(bn_sums, MDL_E_ARG, "mdl_linear_act_stats: needs the BatchNorm sums buffer");
    MDL_REQUIRE((!p1 || idx1) && (!p2 || idx2) && (!p3 || idx3), MDL_E_ARG, "mdl_linear_act_stats: table without index");
    GatherAdd ga = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    int gather = 0;
    if (p1) { ga.p[gather] = (const bf16_t*)p1; ga.idx[gather++] = idx1; }
    if (p2) { ga.p[gather] = (const bf16_t*)p2; ga.idx[gather++] = idx2; }
    if (p3) { ga.p[gather] = (const bf16_t*)p3; ga.idx[gather++] = idx3; }
    MDL_REQUIRE(K >= 4 && K <= 160 && K % 2 == 0 && M >= 34 && M <= 160 && M % 2 == 0 && (!gather || M <= 128), MDL_E_UNSUPP,
                "mdl_linear_act_stats: need even 4<=K<=160 and even 34<=M<=160 (<=128 with tables) (got K=%d M=%d)", K, M);
    MDL_REQUIRE(act >= 0 && act <= 2, MDL_E_ARG, "mdl_linear_act_stats: act must be 0 (none), 1 (relu) or 2 (shifted softplus)");
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && w && out)), MDL_E_ARG, "mdl_linear_act_stats: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 4 == 0 &&
                reinterpret_cast<uintptr_t>(out) % 2 == 0, MDL_E_ARG, "mdl_linear_act_stats: misaligned pointer");
    if (N == 0) return MDL_OK;
    return linear_launch(x, nullptr, 0, w, bias, ga, gather, out, N, K, M, act, stream, bn_sums, n_dev);
}

static int linear_launch(const void* x, const void* xy, int xact, const void* w, const void* bias, const mdl::GatherAdd& ga,
                         int gather, void* out, int64_t N, int K, int M, int act, mdlStream_t stream, float* stats,
                         const int64_t* n_dev) {
    using namespace mdl;
    hipStream_t st = (hipStream_t)stream;
    const int kp = K <= 64 ? 64 : (K <= 128 ? 128 : (K <= 160 ? 160 : 256));
    const int nt = M <= 32 ? 1 : (M <= 64 ? 2 : (M <= 128 ? 4 : 5));
    int64_t grid = cdiv(N, 64);
    if (grid > MDL_LIN_GRID) grid = MDL_LIN_GRID;
    const int lds = (32 * nt + 64) * (kp + 8) * 2;
#define MDL_LIN_K(KP_, NT_, G_, X_)                                                                                  \
    do {                                                                                                             \
        auto kf = linear_act_kernel<KP_, NT_, G_, X_>;                                                               \
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                           \
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(64 * MDL_LIN_NW(KP_, G_)), lds, st, (const bf16_t*)x, (const bf16_t*)w,     \
                           (const bf16_t*)bias, (bf16_t*)out, N, K, M, act, ga, (const bf16_t*)xy, 0, (float*)nullptr, \
                           (const int64_t*)nullptr);                                                                 \
    } while (0)
#define MDL_LIN_S(KP_, NT_, G_)                                                                                      \
    do {                                                                                                             \
        auto kf = linear_act_kernel<KP_, NT_, G_, 0, true>;                                                          \
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                           \
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(64 * MDL_LIN_NW(KP_, G_)), lds, st, (const bf16_t*)x, (const bf16_t*)w,     \
                           (const bf16_t*)bias, (bf16_t*)out, N, K, M, act, ga, (const bf16_t*)xy, 0, stats, n_dev); \
    } while (0)
    if (stats) {
        // (the statistics forms exist for the shapes of the one-pass dense backward: 34 <= M <= 160, K <= 160; 0 or 2 tables)
        if (kp > 160 || nt < 2 || (gather != 0 && gather != 2) || (gather && nt > 4) || xact != 0) {
            set_error("mdl_linear_act_stats: unsupported shape (K=%d M=%d tables=%d)", K, M, gather);
            return MDL_E_UNSUPP;
        }
#define MDL_LIN_SK(KP_)                                                                                              \
        do {                                                                                                         \
            if (gather) { if (nt == 2) MDL_LIN_S(KP_, 2, 2); else MDL_LIN_S(KP_, 4, 2); }                             \
            else { if (nt == 2) MDL_LIN_S(KP_, 2, 0); else if (nt == 4) MDL_LIN_S(KP_, 4, 0); else MDL_LIN_S(KP_, 5, 0); } \
        } while (0)
        if (kp == 64) MDL_LIN_SK(64); else if (kp == 128) MDL_LIN_SK(128); else MDL_LIN_SK(160);
#undef MDL_LIN_SK
        return check_launch("mdl_linear_act_stats");
    }
#define MDL_LIN(KP_, NT_)                                                                                            \
    do {                                                                                                             \
        constexpr int NG_ = (NT_ <= 4) ? 1 : 0;       /* (the gathering forms exist for M <= 128 only) */           \
        if (gather == 1) MDL_LIN_K(KP_, NT_, 1 * NG_, 0);                                                            \
        else if (gather == 2) MDL_LIN_K(KP_, NT_, 2 * NG_, 0);                                                       \
        else if (gather == 3) MDL_LIN_K(KP_, NT_, 3 * NG_, 0);                                                       \
        else if (xact == 1) MDL_LIN_K(KP_, NT_, 0, 1);                                                           \
        else if (xact == 2) MDL_LIN_K(KP_, NT_, 0, 2);                                                           \
        else MDL_LIN_K(KP_, NT_, 0, 0);                                                                          \
    } while (0)
    if (kp == 64) { if (nt == 1) MDL_LIN(64, 1); else if (nt == 2) MDL_LIN(64, 2); else if (nt == 4) MDL_LIN(64, 4); else MDL_LIN(64, 5); }
    else if (kp == 128) { if (nt == 1) MDL_LIN(128, 1); else if (nt == 2) MDL_LIN(128, 2); else if (nt == 4) MDL_LIN(128, 4); else MDL_LIN(128, 5); }
    else if (kp == 160) { if (nt == 1) MDL_LIN(160, 1); else if (nt == 2) MDL_LIN(160, 2); else if (nt == 4) MDL_LIN(160, 4); else MDL_LIN(160, 5); }
    else { if (nt == 1) MDL_LIN(256, 1); else if (nt == 2) MDL_LIN(256, 2); else MDL_LIN(256, 4); }
#undef MDL_LIN
#undef MDL_LIN_K
#undef MDL_LIN_S
    return check_launch("mdl_linear_act");
}

#if MDL_EXPERIMENTS
extern "C" int mdl_mlp2(const void* x, const void* w1, const void* b1, int act1, const void* w2, const void* b2, int act2,
                        void* h, void* y, int64_t N, int K, int M1, int M2, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_mlp2: bf16 only");
    MDL_REQUIRE(K >= 4 && K <= 64 && K % 2 == 0 && M1 >= 2 && M1 <= 160 && M1 % 2 == 0 && M2 >= 1 && M2 <= 160, MDL_E_UNSUPP,
                "mdl_mlp2: need even 4<=K<=64, even 2<=M1<=160, 1<=M2<=160 (got K=%d M1=%d M2=%d)", K, M1, M2);
    MDL_REQUIRE(act1 >= 0 && act1 <= 2 && act2 >= 0 && act2 <= 2, MDL_E_ARG, "mdl_mlp2: act must be 0 (none), 1 (relu) or 2 (shifted softplus)");
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && w1 && w2 && h && y)), MDL_E_ARG, "mdl_mlp2: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w1) % 4 == 0 &&
                reinterpret_cast<uintptr_t>(w2) % 4 == 0, MDL_E_ARG, "mdl_mlp2: misaligned pointer");
    if (N == 0) return MDL_OK;
    int64_t grid = cdiv(N, 128);
    if (grid > 256) grid = 256;
    auto kf = mlp2_kernel<64, 160, 5, 5>;
    const int lds = (32 * 5 * (64 + 8) + 32 * 5 * (160 + 8) + 128 * (64 + 8) + 128 * (160 + 8)) * 2;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
    if (e != hipSuccess) { set_error("mdl_mlp2: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)w1,
                       (const bf16_t*)b1, (const bf16_t*)w2, (const bf16_t*)b2, (bf16_t*)h, (bf16_t*)y, N, K, M1, M2, act1, act2);
    return check_launch("mdl_mlp2");
}
#endif

// out[N, M] = x[N, K] w[M, K]^T for a WIDE output (M in the thousands: NNConv's Y = x W2r, mpnn.py:83-88 — C_out * d3 = 10^4
// columns per node): the layer is a write stream of N * M * 2 bytes; column blocks of 32 * MDL_WIDE_NT = 192 run as the second grid dimension of
// the streaming kernel, each workgroup keeping its block of w in LDS for ~1/64 of the rows.
extern "C" int mdl_linear_wide(const void* x, const void* w, void* out, int64_t N, int K, int64_t M, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_linear_wide: bf16 only");
    MDL_REQUIRE(K >= 4 && K <= 160 && K % 2 == 0 && M >= 1 && M <= 32 * MDL_WIDE_NT * 65535LL && M * 2 * 64 < 0x7fffffffLL, MDL_E_UNSUPP,
                "mdl_linear_wide: need even 4<=K<=160 (got K=%d M=%lld)", K, (long long)M);
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && w && out)), MDL_E_ARG, "mdl_linear_wide: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 4 == 0 &&
                reinterpret_cast<uintptr_t>(out) % 2 == 0, MDL_E_ARG, "mdl_linear_wide: misaligned pointer");
    if (N == 0) return MDL_OK;
    const int kp = K <= 64 ? 64 : (K <= 128 ? 128 : 160);
    constexpr int WNT = MDL_WIDE_NT;                 // 32-column blocks per workgroup
    const unsigned gy = (unsigned)((M + 32 * WNT - 1) / (32 * WNT));
    int64_t gx = cdiv(N, 64);
#ifdef MDL_WIDE_ROUNDS
    {   // experiment: row chunks so that the launch is MDL_WIDE_ROUNDS rounds of 512 resident workgroups
        int64_t cap = std::max<int64_t>(1, (512LL * MDL_WIDE_ROUNDS) / gy);
        if (gx > cap) gx = cap;
    }
#else
    if (gx > 64) gx = 64;
#endif
    const int lds = (32 * WNT + 64) * (kp + 8) * 2;
    const GatherAdd ga{};
#define MDL_WIDE(KP_)                                                                                                 \
    do {                                                                                                              \
        auto kf = linear_act_kernel<KP_, WNT, 0, 0>;                                                              \
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                            \
        hipLaunchKernelGGL(kf, dim3(gy, (unsigned)gx), dim3(64 * MDL_LIN_NW(KP_, 0)), lds, (hipStream_t)stream, (const bf16_t*)x,         \
                           (const bf16_t*)w, (const bf16_t*)nullptr, (bf16_t*)out, N, K, (int)M, 0, ga,               \
                           (const bf16_t*)nullptr, (int)M, (float*)nullptr, (const int64_t*)nullptr);                 \
    } while (0)
    if (kp == 64) MDL_WIDE(64); else if (kp == 128) MDL_WIDE(128); else MDL_WIDE(160);
#undef MDL_WIDE
    return check_launch("mdl_linear_wide");
}
