// cfconv_bwd.hip — K4b: the parameter gradients of SchNet's filter network with the filter RECOMPUTED (SURVEY 7, K4 "fwd + bwd").
//
// Reference path: matdeeplearn/models/schnet.py:131-145 → torch_geometric InteractionBlock / CFConv (2.0.1); autograd runs the
// backward of  W_e = Linear2(ssp(Linear1(rbf_e))),  out_i = sum_{e: j -> i} h_j * W_e * C(d_e)  as five passes over [E, F] tensors.
// With the fused forward (cfconv.hip) storing nothing per edge, the backward of the block is two launches:
//   dh  = the SAME fused forward kernel on the by-source CSR with the output gradient in the place of h (the transposed
//         convolution: dh_j = sum_{e: j -> i} g_i * W_e * C_e), mdl_cfconv_fwd on (rbf, cut) in by-source order;
//   dW1, db1, dW2, db2 = THIS kernel: one pass over the edges in any order, per 64-edge tile
//         dw   = g[tgt] * h[src] * C                      two gathered rows and a scalar per edge, bf16 into LDS
//         a1   = ssp(W1p . rbf^T)                         GEMM1 recomputed (40 MFMAs), bf16 into LDS, unit FP-1 = 1 (carries db2)
//         dW2 += dw^T . a1                                100 MFMAs, both operands by transposed LDS reads (edges on the K axis)
//         da   = (W2^T . dw^T) * ssp'(a1)                 100 MFMAs, lane = edge, bf16 into LDS
//         dW1 += da^T . [rbf | 1]                         40 MFMAs (the constant-1 column of the rbf tile carries db1)
// Nothing per edge is written: the edge pass reads 100 B/edge of rbf + 12 B of indices and gathers 2 x 2F bytes from L2 / MALL,
// against 2F (dw) + 2F (a1) + 2F (w) written by the forward / the gather kernel and 6F read back by the two dense backward
// kernels it replaces (F = 150: 1.8 KB/edge).  The filter W itself is not needed for the parameter gradients.
//
// One 512-thread workgroup per CU, 8 waves; three workgroup barriers per tile (dw / rbf staged | a1 | da); the accumulators of
// dW2 (25 blocks of 32 x 32) and dW1 (10 blocks) are dealt to the waves so that every phase between two barriers carries about
// the same number of MFMAs per wave; flushed once per workgroup with fp32 atomics (MDL_DETERMINISTIC: one workgroup).
#include <algorithm>

#include "mdl_common.h"

// experiment builds only (tools/build_variant.sh): MDL_CFB_SKIP = bit mask of phases left out, for timing ablations (results are then wrong)
#ifndef MDL_CFB_SKIP
#define MDL_CFB_SKIP 0
#endif

namespace mdl {
namespace cfb {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds4_t;

constexpr int G_ = 50, GH = G_ / 2;            // Gaussians per edge; dwords per rbf row
constexpr int KE = 64, ES = KE + 8;            // K of GEMM1 (G + bias column, padded); row stride of the rbf tile / W1p (halfwords)
constexpr int TE = 64;                         // edges per tile
constexpr int NW = 8, NT = NW * WAVE, RPW = TE / NW;
constexpr int NRB = (TE * GH + NT - 1) / NT;   // rbf dwords of a tile per thread
// static at the padded filter width FP = 32 NB of the forward kernel (cfconv.hip: NB = 3 for F in [64, 94], 4 for [96, 126], 5 for [128, 158])
template <int NB_> struct Shape {
    static constexpr int NB = NB_, FP = 32 * NB_;
    static constexpr int W2S = FP + 8;                    // row stride of W2p in the packed weights (cfconv.hip)
    static constexpr int LA = FP + 8;                     // row stride of the [edge][unit] tiles and of W2^T (halfwords)
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_WT = OFF_W1 + FP * ES * 2;   // W2^T [k][m]
    static constexpr int OFF_AL = OFF_WT + FP * LA * 2;   // dw  [edge][unit]
    static constexpr int OFF_BL = OFF_AL + TE * LA * 2;   // a1  [edge][unit]
    static constexpr int OFF_DL = OFF_BL + TE * LA * 2;   // da  [edge][unit]
    static constexpr int OFF_ET = OFF_DL + TE * LA * 2;   // two rbf tiles [edge][ES]
    static constexpr int LDS = OFF_ET + 2 * TE * ES * 2;
    static constexpr int NBLK = NB * NB + 2 * NB;         // accumulator blocks per workgroup: NB^2 of dW2 (id = NB row + col), 2 NB of dW1 (NB^2 + b)
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(OFF_WT % 16 == 0 && OFF_AL % 16 == 0 && OFF_BL % 16 == 0 && OFF_DL % 16 == 0 && OFF_ET % 16 == 0 && (LA * 2) % 16 == 0 && (ES * 2) % 16 == 0, "alignment");
};

struct Params {
    const bf16_t* rbf;       // [E, G] edge features, CSR order
    const float* cut;        // [E]
    const bf16_t* h;         // [N, F] lin1(x)
    const bf16_t* g;         // [N, F] gradient w.r.t. the aggregated messages
    const int32_t* rowptr;   // [N + 1] (rowptr[N] = number of edges that exist)
    const int32_t* src;      // [E]
    const int32_t* tgt;      // [E]
    const bf16_t* wpack;     // W1p [FP][ES] | W2p [FP][W2S] (mdl_cfconv_pack_weights)
    float* dw1;              // [F, G]   +=
    float* db1;              // [F] or nullptr
    float* dw2;              // [F, F]
    float* db2;              // [F] or nullptr
    float* part;             // per-workgroup partial sums [grid][NBLK][1024] or nullptr (then: atomics into the outputs)
    int N, F;
};
constexpr int MAXGRID = 256, RED_SPLIT = 32;

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* base, int row, int ld, int k0, int h) {
    return *reinterpret_cast<const bf16x8*>(base + row * ld + k0 + 8 * h);
}

// k-major fragment (rows = the 16 edges of k-step ks, 32 columns from col0) out of a row-major tile: gemm_tn_stream.inc's read
__device__ __forceinline__ bf16x8 ld_frag_t(const bf16_t* base, int ld, int ks, int col0, int i, int h) {
    const int t = i & 15;
    const bf16_t* pa = base + (16 * ks + 8 * h + (t >> 2)) * ld + col0 + (i & 16) + 4 * (t & 3);
    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(pa + 4 * ld));
    return bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}

__device__ __forceinline__ int pi32(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// one tile's contribution to an NR x NC rectangle of 32 x 32 blocks of  X^T . Y  (X, Y: row-major [edge][..] tiles in LDS): per
// 16-edge k-step NR + NC transposed fragments feed NR * NC MFMAs
template <int NR, int NC>
__device__ __forceinline__ void tn_step(const bf16_t* x, int ldx, int r0, const bf16_t* y, int ldy, int c0, int i, int h, f32x16* acc) {
#pragma unroll
    for (int ks = 0; ks < TE / 16; ++ks) {
        bf16x8 a[NR], b[NC];
#pragma unroll
        for (int u = 0; u < NR; ++u) a[u] = ld_frag_t(x, ldx, ks, 32 * (r0 + u), i, h);
#pragma unroll
        for (int v = 0; v < NC; ++v) b[v] = ld_frag_t(y, ldy, ks, 32 * (c0 + v), i, h);
#pragma unroll
        for (int u = 0; u < NR; ++u)
#pragma unroll
            for (int v = 0; v < NC; ++v) acc[u * NC + v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[v], acc[u * NC + v], 0, 0, 0);
    }
}

template <int NB>
__global__ __launch_bounds__(NT, 2) void cfconv_bwd_w_kernel(Params p) {
    typedef Gate<true> GT;
    typedef Shape<NB> S;
    constexpr int FP = S::FP, LA = S::LA, W2S = S::W2S, LDS = S::LDS, NBLK = S::NBLK;
    constexpr int OFF_W1 = S::OFF_W1, OFF_WT = S::OFF_WT, OFF_AL = S::OFF_AL, OFF_BL = S::OFF_BL, OFF_DL = S::OFF_DL, OFF_ET = S::OFF_ET;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = p.F, FH = F >> 1, N = p.N;
    bf16_t* const w1l = reinterpret_cast<bf16_t*>(smem + OFF_W1);
    bf16_t* const wt = reinterpret_cast<bf16_t*>(smem + OFF_WT);
    bf16_t* const al = reinterpret_cast<bf16_t*>(smem + OFF_AL);
    bf16_t* const bl = reinterpret_cast<bf16_t*>(smem + OFF_BL);
    bf16_t* const dl = reinterpret_cast<bf16_t*>(smem + OFF_DL);

    const int Et = __builtin_amdgcn_readfirstlane(p.rowptr[N]);
    const int64_t n_tiles = ((int64_t)Et + TE - 1) / TE;
    if ((int64_t)blockIdx.x >= n_tiles) return;

    // ---- one-time setup: everything zeroed; W1p verbatim; W2^T (row rho of a 32-row block of k = the a1 unit pi(rho), columns m
    // natural; the bias slot FP - 1 of W2p's K axis is not a unit: zero); the constant-1 column of both rbf tiles
    {
        unsigned* z = reinterpret_cast<unsigned*>(smem);
        for (int q = tid; q < LDS / 4; q += NT) z[q] = 0u;
    }
    __syncthreads();
    {
        constexpr int FPw = FP, W2Sw = W2S;
        const u32x4* gsrc = reinterpret_cast<const u32x4*>(p.wpack);
        u32x4* l = reinterpret_cast<u32x4*>(smem + OFF_W1);
        for (int q = tid; q < FPw * ES * 2 / 16; q += NT) l[q] = gsrc[q];
        const bf16_t* w2p = p.wpack + FPw * ES;
        for (int q = tid; q < FPw * FPw; q += NT) {
            const int prow = q / FPw, pos = q - prow * FPw;               // W2p[prow][pos]: output unit m = pi(prow), input unit k = pos
            const int m = (prow & ~31) | pi32(prow & 31);
            const int krow = (pos & ~31) | pi32(pos & 31);
            wt[krow * LA + m] = pos == FPw - 1 ? (bf16_t)0 : w2p[prow * W2Sw + pos];
        }
        for (int q = tid; q < 2 * TE; q += NT) reinterpret_cast<bf16_t*>(smem + OFF_ET)[q * ES + G_] = 0x3F80;
    }
    // (the first barrier of the tile loop orders these writes against every read)

    // ---- which accumulator blocks this wave owns.  dW2 (5 x 5 blocks, rows m / columns k): waves 2..5 a 2 x 2 square, wave 6 row 4
    // x columns 0..3, wave 7 rows 0..3 x column 4, wave 0 the corner (4, 4), wave 1 none — waves 0 and 1 carry the ninth and tenth
    // block of GEMM1 / da instead.  dW1 (5 x 2 blocks): waves 2..7 one block, waves 0 and 1 two.
    // (class: 0 = 2 x 2, 1 = 1 x 4, 2 = 4 x 1, 3 = one block, 4 = none, 5 = 1 x 2; block j of the rectangle is (r0 + (j >> lnc), c0 + (j & (nc - 1))))
    // NB = 4: 16 + 8 blocks — every wave a 1 x 2 piece of dW2 and one dW1 block.  NB = 3: 9 + 6 blocks — waves 0..6 one dW2 block,
    // wave 7 the 1 x 2 piece (2, 1..2); waves 0..5 one dW1 block.
    const int cls5 = wv >= 2 && wv <= 5 ? 0 : (wv == 6 ? 1 : (wv == 7 ? 2 : (wv == 0 ? 3 : 4)));
    const int cls2 = NB == 5 ? cls5 : (NB == 4 ? 5 : (wv == 7 ? 5 : 3));
    const int r0 = NB == 5 ? (cls5 == 0 ? 2 * ((wv - 2) >> 1) : (cls5 == 1 || cls5 == 3 ? 4 : 0)) : (NB == 4 ? wv >> 1 : (wv == 7 ? 2 : wv / 3));
    const int c0 = NB == 5 ? (cls5 == 0 ? 2 * ((wv - 2) & 1) : (cls5 == 2 || cls5 == 3 ? 4 : 0)) : (NB == 4 ? 2 * (wv & 1) : (wv == 7 ? 1 : wv % 3));
    const int lnc = NB == 5 ? (cls5 == 0 ? 1 : (cls5 == 1 ? 2 : 0)) : (NB == 4 ? 1 : (wv == 7 ? 1 : 0));
    const int nblk2 = NB == 5 ? (cls5 <= 2 ? 4 : (cls5 == 3 ? 1 : 0)) : (NB == 4 ? 2 : (wv == 7 ? 2 : 1));
    // dW1 blocks: (unit block = b >> 1, Gaussian block = b & 1)
    const int b1a = NB == 5 ? (wv >= 2 ? wv - 2 : 6 + 2 * wv) : (NB == 4 ? wv : (wv < 6 ? wv : -1));
    const int b1b = NB == 5 ? (wv >= 2 ? -1 : 7 + 2 * wv) : -1;
    // five accumulator tiles per wave: dW2 blocks in acc[0 .. nblk2 - 1], the first dW1 block in acc[4], the second one (waves 0
    // and 1, which own at most one dW2 block) in acc[3]
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    // ---- staging registers: this wave's 8 rows of the next tile (g and h rows as dwords: lane, and lane + 64 for the tail of the
    // row), the cutoff factors (wave-uniform), this thread's dwords of the rbf tile
    // (the tail of a row — dwords 64 .. F/2 - 1, at most 16 — is fetched for four rows per load: lane >> 4 picks the row)
    // The indices of a tile's rows travel one tile further ahead than its rows (lane l: row l & 7 of this wave's eight): a row
    // load needs its index in a scalar register, and scalar loads are not available behind a barrier (the compiler treats the
    // barrier as a clobber), so the wave reads them with one vector load and broadcasts with v_readlane when the rows are requested.
    unsigned g0[RPW], h0[RPW], g1[RPW / 4], h1[RPW / 4], rb[NRB];
    int vs = 0, vt = 0;
    float vc = 0.0f, cuc = 0.0f;                 // cutoff factors: of the rows indexed by (vs, vt); of the rows in flight / staged
    const int tsel = lane >> 4, tdw = 64 + (lane & 15);
    auto request_idx = [&](int64_t tile) {
        const int e = (int)(tile * TE) + RPW * wv + (lane & 7), ec = min(e, Et - 1);
        vs = p.src[ec];
        vt = p.tgt[ec];
        const float c = p.cut[ec];
        vc = e < Et ? c : 0.0f;
    };
    auto request_rows = [&](int64_t tile) {
        int sn[RPW], tn[RPW];
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            sn[r] = __builtin_amdgcn_readlane(vs, r);
            tn[r] = __builtin_amdgcn_readlane(vt, r);
        }
        cuc = vc;
        const int d0 = min(lane, FH - 1), d1 = min(tdw, FH - 1);
        if (MDL_CFB_SKIP & 16) {            // (ablation: every row is node 0's)
#pragma unroll
            for (int r = 0; r < RPW; ++r) sn[r] = tn[r] = 0;
        }
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            h0[r] = reinterpret_cast<const unsigned*>(p.h + (int64_t)sn[r] * F)[d0];
            g0[r] = reinterpret_cast<const unsigned*>(p.g + (int64_t)tn[r] * F)[d0];
        }
#pragma unroll
        for (int k = 0; k < RPW / 4; ++k) {
            const int s_ = tsel == 0 ? sn[4 * k] : (tsel == 1 ? sn[4 * k + 1] : (tsel == 2 ? sn[4 * k + 2] : sn[4 * k + 3]));
            const int t_ = tsel == 0 ? tn[4 * k] : (tsel == 1 ? tn[4 * k + 1] : (tsel == 2 ? tn[4 * k + 2] : tn[4 * k + 3]));
            h1[k] = reinterpret_cast<const unsigned*>(p.h + (int64_t)s_ * F)[d1];
            g1[k] = reinterpret_cast<const unsigned*>(p.g + (int64_t)t_ * F)[d1];
        }
        const int64_t left = ((int64_t)Et - tile * TE) * (G_ * 2), cap = (int64_t)TE * G_ * 2;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.rbf + tile * (int64_t)(TE * G_)), 0,
                                                                             (int)(left < cap ? left : cap), 0x00020000);
#pragma unroll
        for (int j = 0; j < NRB; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b32(rr, 4u * (unsigned)(tid + NT * j), 0, 0);
    };

    // ---- the phases of a tile as lambdas (the loop below interleaves the phases of consecutive tiles)
    auto commit_al = [&]() {                    // dw rows of the staged tile (fp32 products, rounded once) -> al
        unsigned* al32 = reinterpret_cast<unsigned*>(al);
        auto prod = [](unsigned gv, unsigned hv, float c) {
            return pk_bf16(__uint_as_float(gv << 16) * __uint_as_float(hv << 16) * c,
                           __uint_as_float(gv & 0xffff0000u) * __uint_as_float(hv & 0xffff0000u) * c);
        };
#pragma unroll
        for (int r = 0; r < RPW; ++r)
            if (lane < FH) al32[(RPW * wv + r) * (LA / 2) + lane] = prod(g0[r], h0[r], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cuc), r)));
#pragma unroll
        for (int k = 0; k < RPW / 4; ++k) {
            const float c = __shfl(cuc, 4 * k + tsel);
            if (tdw < FH) al32[(RPW * wv + 4 * k + tsel) * (LA / 2) + tdw] = prod(g1[k], h1[k], c);
        }
    };
    auto commit_et = [&](bf16_t* et) {          // the staged rbf dwords -> the rbf tile
        unsigned* et32 = reinterpret_cast<unsigned*>(et);
#pragma unroll
        for (int j = 0; j < NRB; ++j) {
            const int d = tid + NT * j;
            const int row = d / GH, col = d - row * GH;
            if (d < TE * GH) et32[row * (ES / 2) + col] = rb[j];
        }
    };
    // S1: a1 = ssp(W1p . rbf^T); lane = edge.  Rows of W1p are permuted (pi) so that registers 8 t .. 8 t + 7 of lane half h are
    // units 16 t + 8 h .. + 7 of the block.  Ten blocks (edge block eb, unit block ub) for eight waves: wave w takes (w & 1, w >> 1);
    // the fifth unit block of edge block eb is computed by all four waves of that parity (4 MFMAs: cheap) and each applies the
    // activation to a quarter of it
    auto s1 = [&](const bf16_t* et) {
        if (MDL_CFB_SKIP & 1) return;
        auto ssp2 = [](float t0, float t1) { return pk_bf16(LN2_F * (GT::softplus_u(t0) - 1.0f), LN2_F * (GT::softplus_u(t1) - 1.0f)); };
        const int eb = wv & 1, ub = wv >> 1;
        if constexpr (NB == 5) {
            f32x16 d, d4;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = d4[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < KE / 16; ++k) {
                const bf16x8 bfrag = ld_frag(et, 32 * eb + i, ES, 16 * k, h);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w1l, 32 * ub + i, ES, 16 * k, h), bfrag, d, 0, 0, 0);
                d4 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w1l, 32 * (NB - 1) + i, ES, 16 * k, h), bfrag, d4, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ssp2(d[8 * t + 2 * q], d[8 * t + 2 * q + 1]);
                *reinterpret_cast<u32x4*>(bl + (32 * eb + i) * LA + 32 * ub + 16 * t + 8 * h) = v;
            }
            // quarter ub of the fifth block: registers 4 ub .. 4 ub + 3 = units 128 + 16 (ub >> 1) + 8 h + 4 (ub & 1) .. + 3
            float q0, q1, q2, q3;
            if (ub == 0) { q0 = d4[0]; q1 = d4[1]; q2 = d4[2]; q3 = d4[3]; }
            else if (ub == 1) { q0 = d4[4]; q1 = d4[5]; q2 = d4[6]; q3 = d4[7]; }
            else if (ub == 2) { q0 = d4[8]; q1 = d4[9]; q2 = d4[10]; q3 = d4[11]; }
            else { q0 = d4[12]; q1 = d4[13]; q2 = d4[14]; q3 = d4[15]; }
            u32x2 v = u32x2{ssp2(q0, q1), ssp2(q2, q3)};
            if (ub == 3 && h == 1) v[1] = (v[1] & 0x0000ffffu) | 0x3F800000u;                           // unit FP - 1: the constant 1 (db2)
            *reinterpret_cast<u32x2*>(bl + (32 * eb + i) * LA + 32 * (NB - 1) + 16 * (ub >> 1) + 8 * h + 4 * (ub & 1)) = v;
        } else if (ub < NB) {                                    // 2 NB <= 8 blocks: one per wave (NB = 3: waves 6, 7 have none)
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < KE / 16; ++k)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w1l, 32 * ub + i, ES, 16 * k, h), ld_frag(et, 32 * eb + i, ES, 16 * k, h), d, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                u32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ssp2(d[8 * t + 2 * q], d[8 * t + 2 * q + 1]);
                if (ub == NB - 1 && t == 1 && h == 1) v[3] = (v[3] & 0x0000ffffu) | 0x3F800000u;        // unit FP - 1: the constant 1 (db2)
                *reinterpret_cast<u32x4*>(bl + (32 * eb + i) * LA + 32 * ub + 16 * t + 8 * h) = v;
            }
        }
    };
    auto s2 = [&]() {
        // S2a: dW2 += dw^T . a1 (k = the tile's 64 edges)
        if (MDL_CFB_SKIP & 2) { }
        else if (cls2 == 0) tn_step<2, 2>(al, LA, r0, bl, LA, c0, i, h, acc);
        else if (cls2 == 1) tn_step<1, 4>(al, LA, r0, bl, LA, c0, i, h, acc);
        else if (cls2 == 2) tn_step<4, 1>(al, LA, r0, bl, LA, c0, i, h, acc);
        else if (cls2 == 3) tn_step<1, 1>(al, LA, r0, bl, LA, c0, i, h, acc);
        else if (cls2 == 5) tn_step<1, 2>(al, LA, r0, bl, LA, c0, i, h, acc);
        // S2b: da = (W2^T . dw^T) .* ssp'(a1); lane = edge, rows of W2^T permuted like W1p's
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
            const int bidx = wv + NW * bb;
            if (bidx < 2 * NB && !(MDL_CFB_SKIP & 4)) {
                const int eb = bidx & 1, kb = bidx >> 1;
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.0f;
#pragma unroll
                for (int kk = 0; kk < FP / 16; ++kk)
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(wt, 32 * kb + i, LA, 16 * kk, h), ld_frag(al, 32 * eb + i, LA, 16 * kk, h), d, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int off = (32 * eb + i) * LA + 32 * kb + 16 * t + 8 * h;
                    const u32x4 y = *reinterpret_cast<const u32x4*>(bl + off);
                    u32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // ssp'(x) = sigmoid(x) = 1 - exp(-(y + ln 2)) from the rounded output y (what the unfused backward uses)
                        const float s0 = 1.0f - 0.5f * __builtin_amdgcn_exp2f(-LOG2E_F * __uint_as_float(y[q] << 16));
                        const float s1v = 1.0f - 0.5f * __builtin_amdgcn_exp2f(-LOG2E_F * __uint_as_float(y[q] & 0xffff0000u));
                        v[q] = pk_bf16(d[8 * t + 2 * q] * s0, d[8 * t + 2 * q + 1] * s1v);
                    }
                    *reinterpret_cast<u32x4*>(dl + off) = v;
                }
            }
        }
    };
    auto s3 = [&](const bf16_t* et) {           // S3: dW1 += da^T . [rbf | 1]
        if (MDL_CFB_SKIP & 8) return;
        if (NB >= 4 || b1a >= 0) tn_step<1, 1>(dl, LA, b1a >> 1, et, ES, b1a & 1, i, h, acc + 4);
        if (b1b >= 0) tn_step<1, 1>(dl, LA, b1b >> 1, et, ES, b1b & 1, i, h, acc + 3);
    };
    bf16_t* const et0 = reinterpret_cast<bf16_t*>(smem + OFF_ET);
    bf16_t* const et1 = et0 + TE * ES;

    // ---- the tile loop.  Two workgroup barriers per tile; between them the phases of consecutive tiles are interleaved so that an
    // interval mixes matrix work with vector work and with the staging traffic:
    //   interval A:  dW2(t), da(t) -> dl                       | commit rbf(t+1) -> the other rbf tile
    //   interval B:  dW1(t)  (dl, rbf(t))                      | commit dw(t+1) -> al, request rows / rbf (t+2), a1(t+1) -> bl
    // (al and bl were last read in interval A, dl and the rbf tile of t are read in B and rewritten in the next A: every buffer has
    // one barrier between its last reader and its next writer.)  Rows are requested a whole tile before they are committed.
    const int64_t G = gridDim.x;
    int64_t tile = blockIdx.x;
    request_idx(tile);
    request_rows(tile);
    request_idx(min(tile + G, n_tiles - 1));
    commit_et(et0);
    commit_al();
    __syncthreads();
    if (tile + G < n_tiles) {
        request_rows(tile + G);
        request_idx(min(tile + 2 * G, n_tiles - 1));
    }
    s1(et0);
    __syncthreads();
    int cur = 0;
    for (; tile < n_tiles; tile += G, cur ^= 1) {
        bf16_t* const et = cur ? et1 : et0;
        bf16_t* const etn = cur ? et0 : et1;
        const bool more = tile + G < n_tiles;
        // ---- interval A
        s2();
        if (more) commit_et(etn);
        __syncthreads();
        // ---- interval B
        s3(et);
        if (more) {
            commit_al();
            if (tile + 2 * G < n_tiles) {
                request_rows(tile + 2 * G);
                request_idx(min(tile + 3 * G, n_tiles - 1));
            }
            s1(etn);
        }
        __syncthreads();
    }

    if (MDL_CFB_SKIP & 32) return;
    // ---- flush.  256 workgroups adding 35 blocks each into the same 33 k addresses with atomics cost 160 us of a 530-us launch
    // (all workgroups finish together): with a scratch buffer the blocks leave as plain coalesced stores in accumulator layout
    // (register r of a block: 64 consecutive floats) and cfconv_bwd_w_reduce_kernel sums them
    if (p.part) {
        float* const mine = p.part + (int64_t)blockIdx.x * (NBLK * 1024) + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nblk2) {
                float* dst = mine + (NB * (r0 + (j >> lnc)) + c0 + (j & ((1 << lnc) - 1))) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[64 * r] = acc[j][r];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int b = j == 0 ? b1a : b1b;
            if (b >= 0) {
                float* dst = mine + (NB * NB + b) * 1024;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[64 * r] = acc[4 - j][r];
            }
        }
        return;
    }
    // ---- flush with atomics: lane = column (k of dW2 / Gaussian of dW1), registers = rows (output unit m / unit)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < nblk2) {
            const int col = 32 * (c0 + (j & ((1 << lnc) - 1))) + i, rb0 = 32 * (r0 + (j >> lnc));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rb0 + d_row(r, h);
                if (m < F) {
                    if (col < F) unsafeAtomicAdd(p.dw2 + (int64_t)m * F + col, acc[j][r]);
                    else if (col == FP - 1 && p.db2) unsafeAtomicAdd(p.db2 + m, acc[j][r]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int b = j == 0 ? b1a : b1b;
        if (b >= 0) {
            const int col = 32 * (b & 1) + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = 32 * (b >> 1) + d_row(r, h);
                if (u < F) {
                    if (col < G_) unsafeAtomicAdd(p.dw1 + (int64_t)u * G_ + col, acc[4 - j][r]);
                    else if (col == G_ && p.db1) unsafeAtomicAdd(p.db1 + u, acc[4 - j][r]);
                }
            }
        }
    }
}

// sums the workgroups' partial blocks: one thread per accumulator element (block id, register, lane), the workgroup range cut in
// RED_SPLIT pieces (gridDim.y) whose sums meet in the outputs with RED_SPLIT-way atomics
__global__ __launch_bounds__(256) void cfconv_bwd_w_reduce_kernel(const float* __restrict__ part, int nwg, const int32_t* __restrict__ rowptr,
                                                                   int N, int F, int NB, float* __restrict__ dw1, float* __restrict__ db1,
                                                                   float* __restrict__ dw2, float* __restrict__ db2) {
    const int NBLK = NB * NB + 2 * NB, FP = 32 * NB;
    nwg = min(nwg, (rowptr[N] + TE - 1) / TE);                      // workgroups without a tile (padded batch) wrote nothing
    const int el = blockIdx.x * 256 + threadIdx.x;                  // < NBLK * 1024
    const int id = el >> 10, r = (el >> 6) & 15, lane = el & 63, i = lane & 31, h = lane >> 5;
    const int per = (nwg + RED_SPLIT - 1) / RED_SPLIT, w0 = blockIdx.y * per, w1 = min(nwg, w0 + per);
    if (w0 >= w1) return;
    // (eight workgroups' values per thread, all loads in flight at once: the pass is latency-bound otherwise — 4 pieces of 64
    // measured 100 us for 37 MB)
    const float* q = part + (int64_t)w0 * (NBLK * 1024) + el;
    float v = 0.0f;
    for (int w = w0; w < w1; w += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = w + k < w1 ? q[(int64_t)k * (NBLK * 1024)] : 0.0f;
        v += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        q += 8 * (NBLK * 1024);
    }
    if (id < NB * NB) {
        const int m = 32 * (id / NB) + d_row(r, h), col = 32 * (id % NB) + i;
        if (m < F) {
            if (col < F) unsafeAtomicAdd(dw2 + (int64_t)m * F + col, v);
            else if (col == FP - 1 && db2) unsafeAtomicAdd(db2 + m, v);
        }
    } else {
        const int b = id - NB * NB, u = 32 * (b >> 1) + d_row(r, h), col = 32 * (b & 1) + i;
        if (u < F) {
            if (col < G_) unsafeAtomicAdd(dw1 + (int64_t)u * G_ + col, v);
            else if (col == G_ && db1) unsafeAtomicAdd(db1 + u, v);
        }
    }
}

}  // namespace cfb
}  // namespace mdl

using namespace mdl;

extern "C" size_t mdl_cfconv_bwd_w_scratch_bytes(void) { return (size_t)cfb::MAXGRID * cfb::Shape<5>::NBLK * 1024 * sizeof(float); }

template <int NB>
static int cfconv_bwd_w_launch(const cfb::Params& p, int64_t grid, int64_t N, const int32_t* rowptr, hipStream_t st) {
    auto kf = cfb::cfconv_bwd_w_kernel<NB>;
    constexpr int lds = cfb::Shape<NB>::LDS;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
    if (e != hipSuccess) { set_error("mdl_cfconv_bwd_w: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(cfb::NT), lds, st, p);
    if (p.part)
        hipLaunchKernelGGL(cfb::cfconv_bwd_w_reduce_kernel, dim3(cfb::Shape<NB>::NBLK * 1024 / 256, cfb::RED_SPLIT), dim3(256), 0, st, p.part,
                           (int)grid, rowptr, (int)N, p.F, NB, p.dw1, p.db1, p.dw2, p.db2);
    return check_launch("mdl_cfconv_bwd_w");
}

extern "C" int mdl_cfconv_bwd_w(const void* rbf, const float* cut, const void* h, const void* g, const int32_t* rowptr,
                                const int32_t* src, const int32_t* tgt, const void* wpack, float* dw1, float* db1, float* dw2,
                                float* db2, void* scratch, int64_t N, int64_t E, int F, int G, int dtype, mdlStream_t stream) {
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;      // one workgroup: every element gets its adds from one wave in tile order
    dtype &= MDL_DTYPE_MASK;
    MDL_REQUIRE(mdl_cfconv_supported(F, G, dtype), MDL_E_UNSUPP, "mdl_cfconv_bwd_w: bf16, G = 50 and even F in [64, 158] only (F = %d, G = %d, dtype %d)", F, G, dtype);
    MDL_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) && E < (1ll << 31) - 64, MDL_E_ARG, "mdl_cfconv_bwd_w: sizes out of range");
    if (N == 0 || E == 0) return MDL_OK;
    MDL_REQUIRE(rbf && cut && h && g && rowptr && src && tgt && wpack && dw1 && dw2, MDL_E_ARG, "mdl_cfconv_bwd_w: null argument");
    MDL_REQUIRE(((uintptr_t)rbf % 4) == 0 && ((uintptr_t)h % 4) == 0 && ((uintptr_t)g % 4) == 0 && ((uintptr_t)wpack % 16) == 0 &&
                    ((uintptr_t)scratch % 16) == 0, MDL_E_ARG, "mdl_cfconv_bwd_w: misaligned tensor");
    cfb::Params p{static_cast<const bf16_t*>(rbf), cut, static_cast<const bf16_t*>(h), static_cast<const bf16_t*>(g), rowptr, src, tgt,
                  static_cast<const bf16_t*>(wpack), dw1, db1, dw2, db2, det ? nullptr : static_cast<float*>(scratch), (int)N, F};
    const int64_t grid = det ? 1 : std::min<int64_t>(cfb::MAXGRID, std::max<int64_t>(1, cdiv(E, cfb::TE)));
    switch ((F + 2 + 31) / 32) {                              // the padded width of the packed weights (cfconv.hip: nbk_for)
        case 3: return cfconv_bwd_w_launch<3>(p, grid, N, rowptr, (hipStream_t)stream);
        case 4: return cfconv_bwd_w_launch<4>(p, grid, N, rowptr, (hipStream_t)stream);
        default: return cfconv_bwd_w_launch<5>(p, grid, N, rowptr, (hipStream_t)stream);
    }
}
