// cfconv.hip — K4: the fused forward of SchNet's continuous-filter convolution (kernel list K4).
//
// Reference path: matdeeplearn/models/schnet.py:131-145 calls torch_geometric's InteractionBlock / CFConv (2.0.1):
//     W_e  = Linear(F, F)( ssp( Linear(G, F)(rbf_e) ) )                 the filter network on the Gaussian expansion of d_e
//     out_i = sum_{e: j -> i}  h_j * W_e * C(d_e)                       h = lin1(x), C = cosine cutoff
// Unfused (nn.CFConv before this kernel) that is three passes over the edges: two dense layers (100 + 300 B read, 300 + 300 B
// written per edge at F = 150) and the gather-multiply-reduce (300 B read + the h rows).  Here ONE kernel walks the edges in
// CSR order: a wave stages the rbf rows of a 32-edge tile in LDS and chains, in registers,
//     GEMM1  D1[unit][edge] = W1p . rbf^T          (4 k-steps, weights from LDS, bias in the constant-1 column of the tile)
//     ssp, bf16 -> these registers ARE the B operand of
//     GEMM2  D2[unit][edge] = W2p . a1             (10 k-steps; W2p's K columns packed in the order GEMM1 leaves the units in)
//     msg = bf16(D2) * h[src] * C                  (lane = edge: the cutoff is one scalar per lane, h[src] arrives as 8-byte chunks)
//     out[tgt] += one-hot(tgt) . msg               (32-unit block by block through a 2.3-KB LDS transpose, like kernel 2 of K3)
// and writes the two activations the existing backward consumes (a1 = layer-1 output, W = filter) straight from the
// accumulator layout (8-byte chunks: a 32-edge tile is one contiguous 9.6-KB run of either tensor).  Without them (inference:
// a1 == w == nullptr) the edge pass reads 100 B and gathers 300 B per edge and writes nothing per edge.
//
// Lane = EDGE in both products (the MFMA computes the transposed layer, A = weights, B = activations), so GEMM1's D registers
// feed GEMM2 without leaving the wave, and the reduction over the edges — the one step that needs the edges on the K axis — is
// the only trip through LDS.  One-hot operands: two packed instructions per pair (value 2^-126, the messages travel scaled by
// 2^64 folded into the cutoff factor; see cgconv_ep2.inc).
#include <algorithm>

#include "mdl_common.h"

namespace mdl {
namespace cf {

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds4_t;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int G_ = 50, GH = G_ / 2;           // Gaussians per edge (the reference's edge features); dwords per rbf row
constexpr int KE = 64, EKS = KE + 8;          // K of GEMM1 (G + bias column, padded); row stride of the rbf tile (halfwords)
constexpr int W1S = KE + 8;                   // row stride of the packed layer-1 weights in LDS (halfwords)
constexpr int DCS = 32 * 8 + 32;              // one chunk row of the message block: 32 edges x 8 bytes + pad
#ifndef MDL_CF_NWAVE
#define MDL_CF_NWAVE 8
#endif
constexpr int NWAVE = MDL_CF_NWAVE, NT = NWAVE * WAVE;
constexpr int NJ = (32 * GH + WAVE - 1) / WAVE;           // dwords of an rbf tile per lane
constexpr int OFF_ET = 0, OFF_DP = 32 * EKS * 2, OFF_TSL = OFF_DP + 8 * DCS, OFF_STASH = OFF_TSL + 64;
constexpr int WAVE_BYTES = OFF_STASH + NJ * WAVE * 4;     // (the stash: the NEXT tile's rbf dwords, parked in LDS once they have arrived)
// Filter width: the kernel is static at a padded width FP = 32 NBK with F <= FP - 2 (unit FP - 1 carries the bias of layer 2) and
// F >= 32 (NBK - 1) (every block but the last lies inside the h rows): NBK = 3 for F in [64, 94], 4 for [96, 126], 5 for [128, 158]
// (SchNet_demo: 150).  The packed weights' layout follows FP.
template <int NBK_> struct Shape {
    static constexpr int NBK = NBK_, FP = 32 * NBK_;
    static constexpr int W2S = FP + 8;                    // row stride of the packed layer-2 weights in LDS (halfwords)
    static constexpr int W1_BYTES = FP * W1S * 2, W2_BYTES = FP * W2S * 2;
    static constexpr int LDS = W1_BYTES + W2_BYTES + NWAVE * WAVE_BYTES;
    static_assert(W1_BYTES % 16 == 0 && W2_BYTES % 16 == 0, "alignment");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};
constexpr float UP = 18446744073709551616.0f;             // 2^64: the messages travel scaled (folded into the cutoff factor)
constexpr float RW = 4611686018427387904.0f;              // 2^62 = 2^126 / 2^64: scale of the one-hot sums
static_assert(WAVE_BYTES % 16 == 0 && OFF_DP % 16 == 0 && OFF_TSL % 16 == 0, "alignment");
__host__ __device__ inline int nbk_for(int F) { return (F + 2 + 31) / 32; }

#ifdef MDL_CF_TIMING      // experiment builds only: per-phase cycle counters of wave 0 of workgroup 0 (tools/bench_cfconv.py prints them)
__device__ long long g_cf_dbg[16];
#define CF_TDECL long long tprev = clock64(), tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tcount = 0
#define CF_TMARK(k) do { __builtin_amdgcn_sched_barrier(0); const long long _t = clock64(); tacc[k] += _t - tprev; tprev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CF_TTILE() (tcount += 1)
#define CF_TFLUSH() do { if (gw == 0 && lane == 0) { for (int _k = 0; _k < 8; ++_k) g_cf_dbg[_k] += tacc[_k]; g_cf_dbg[8] += tcount; } } while (0)
#else
#define CF_TDECL do { } while (0)
#define CF_TMARK(k) do { } while (0)
#define CF_TTILE() do { } while (0)
#define CF_TFLUSH() do { } while (0)
#endif

struct Params {
    const bf16_t* rbf;       // [E, G] edge features, CSR order
    const float* cut;        // [E] cutoff factor C(d_e)
    const bf16_t* h;         // [N, F] lin1(x)
    const int32_t* rowptr;   // [N + 1]
    const int32_t* src;      // [E] source node per CSR position
    const int32_t* tgt;      // [E] target node per CSR position
    const bf16_t* wpack;     // W1p [FP][W1S] | W2p [FP][W2S] (mdl_cfconv_pack_weights)
    bf16_t* out;             // [N, F]
    bf16_t* a1;              // [E, F] layer-1 output (post ssp) or nullptr
    bf16_t* w;               // [E, F] filter or nullptr
    int N, E, F;
};

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* base, int row, int ld, int k0, int h) {
    return *reinterpret_cast<const bf16x8*>(base + row * ld + k0 + 8 * h);
}

// two bf16 one-hot values (2^-126 where the halfword of x is zero): see cgconv_ep2.inc
__device__ __forceinline__ unsigned oh2x(unsigned x) {
    const u16x2 k = {0x0080, 0x0080};
    return __builtin_bit_cast(unsigned, (u16x2)__builtin_elementwise_sub_sat(k, __builtin_bit_cast(u16x2, x)));
}

// smallest n in [0, N] with rowptr[n] + n >= b (the cost in front of node n: its edges and itself); 64 probes per round
__device__ __forceinline__ int lower_bound_cost(const int32_t* __restrict__ rowptr, int N, int64_t b, int lane) {
    int lo = 0, hi = N;
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;
        const int n = min(lo + lane * step, hi);
        const bool ge = n >= hi ? true : ((int64_t)rowptr[n] + n >= b);
        const unsigned long long m = __ballot(ge);
        if (m == 0ull) { lo = min(lo + 63 * step, hi - 1) + 1; continue; }
        const int fl = __builtin_ctzll(m);
        if (fl == 0) { hi = lo; break; }
        hi = min(lo + fl * step, hi);
        lo = lo + (fl - 1) * step + 1;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

template <int NBK>
__global__ __launch_bounds__(NT, 2) void cfconv_fwd_kernel(Params p) {
    typedef Gate<true> GT;
    typedef Shape<NBK> SH;
    constexpr int FP = SH::FP, W2S = SH::W2S, W1_BYTES = SH::W1_BYTES, W2_BYTES = SH::W2_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, t16 = lane & 15;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = p.F, N = p.N;

    // ---- one-time setup: packed weights -> LDS, the waves' regions zeroed, bias column of the rbf tiles = 1
    {
        const u32x4* g = reinterpret_cast<const u32x4*>(p.wpack);
        u32x4* l = reinterpret_cast<u32x4*>(smem);
        for (int q = tid; q < (W1_BYTES + W2_BYTES) / 16; q += NT) l[q] = g[q];
        unsigned* z = reinterpret_cast<unsigned*>(smem + W1_BYTES + W2_BYTES);
        for (int q = tid; q < NWAVE * WAVE_BYTES / 4; q += NT) z[q] = 0u;
    }
    __syncthreads();
    char* const mybuf = smem + W1_BYTES + W2_BYTES + wv * WAVE_BYTES;
    bf16_t* const et = reinterpret_cast<bf16_t*>(mybuf + OFF_ET);
    char* const dp = mybuf + OFF_DP;
    if (lane < 32) et[lane * EKS + G_] = 0x3F80;
    const bf16_t* const w1l = reinterpret_cast<const bf16_t*>(smem);
    const bf16_t* const w2l = reinterpret_cast<const bf16_t*>(smem + W1_BYTES);
    wave_lds_fence();

    const int Et = __builtin_amdgcn_readfirstlane(p.rowptr[N]);
    // rows past the last edge (a padded static batch): the activations the backward reads must be finite there
    if (p.a1 || p.w) {
        const int64_t d0 = (int64_t)Et * F / 2, d1 = (int64_t)p.E * F / 2;
        unsigned* a = reinterpret_cast<unsigned*>(p.a1);
        unsigned* b = reinterpret_cast<unsigned*>(p.w);
        for (int64_t q = d0 + (int64_t)blockIdx.x * NT + tid; q < d1; q += (int64_t)gridDim.x * NT) {
            if (a) a[q] = 0u;
            if (b) b[q] = 0u;
        }
    }

    // ---- this wave's node range: equal cost (edges + nodes) per wave
    const int gw = blockIdx.x * NWAVE + wv, Wn = gridDim.x * NWAVE;
    const int64_t total = (int64_t)Et + N;
    const int na = gw == 0 ? 0 : lower_bound_cost(p.rowptr, N, total * gw / Wn, lane);
    const int nb = gw == Wn - 1 ? N : lower_bound_cost(p.rowptr, N, total * (gw + 1) / Wn, lane);

    const unsigned rsplat = (unsigned)(i << 7) * 0x00010001u;
    if (na >= nb) return;

    // ---- software pipeline: the operands a tile needs from memory (its indices, cutoff factors and rbf dwords) are requested ONE
    // TILE AHEAD, and its h[src] rows at its top; everything is waited for ONCE per tile, behind GEMM1 (20 MFMAs + 80 softplus
    // values: ~2 us of cover) and in front of the tile's first global store — from there to the end of the tile the wave only
    // stores.  (Loads and stores share one counter that the compiler can only drain completely once both kinds are in flight:
    // a load wait anywhere else would also wait for the stores issued just before it.)  The edges of consecutive tiles are
    // contiguous in CSR order whatever the group boundaries, so "the next tile" starts at eb + nv before its length is known.
    // The rbf dwords of the next tile are 13 registers that would stay live through the whole tile (with the 80 persistent
    // accumulators, 40 a1 and 40 h registers that is past 256: the compiler spilled, and a spill RELOAD is a scratch load —
    // it shares the vector memory counter and its wait drains every load just issued): once arrived they are parked in LDS.
    struct Pre { int src, tg; float cu; unsigned v[NJ]; };
    unsigned* const stash = reinterpret_cast<unsigned*>(mybuf + OFF_STASH);
    auto park = [&](const Pre& q) {
        int lane_p = lane;                                           // (per call: the address is not kept across the tile)
        asm volatile("" : "+v"(lane_p));
#pragma unroll
        for (int j = 0; j < NJ; ++j) stash[j * WAVE + lane_p] = q.v[j];
    };
    auto request = [&](Pre& q, int eb) {
        const int ec = min(eb + i, Et - 1);
        q.src = p.src[ec];
        q.tg = p.tgt[ec];
        q.cu = p.cut[ec];
        const unsigned* g = reinterpret_cast<const unsigned*>(p.rbf + (int64_t)eb * G_);
        const int64_t lim = ((int64_t)Et - eb) * GH;                  // dwords left in the array
#pragma unroll
        for (int j = 0; j < NJ; ++j) q.v[j] = g[min((int64_t)(lane + WAVE * j), lim - 1)];
    };
    int e0 = __builtin_amdgcn_readfirstlane(p.rowptr[na]);
    if (Et <= 0) {                                                   // no edges at all: every row of out is zero
        for (int n = na + (lane >> 5); n < nb; n += 2)
            for (int u = i; u < F; u += 32) p.out[(int64_t)n * F + u] = 0;
        return;
    }
    int c_src, c_tg;
    float c_cu;
    {
        Pre first;
        request(first, min(e0, Et - 1));
        park(first);
        c_src = first.src; c_tg = first.tg; c_cu = first.cu;
    }

    CF_TDECL;
    for (int n0 = na; n0 < nb;) {
        const int n1 = min(n0 + 32, nb);
        const int e1 = __builtin_amdgcn_readfirstlane(p.rowptr[n1]);
        CF_TMARK(6);
        f32x16 oacc[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[b][r] = 0.0f;

        for (int eb = e0; eb < e1; eb += 32) {
            const int nv = min(32, e1 - eb);
            const bool valid_i = i < nv;
            const int srcn = c_src;
            const float cu = valid_i ? c_cu * UP : 0.0f;
            // ---- the tile's rbf rows (nv x 100 contiguous bytes, in registers since the last tile) -> LDS rows of EKS halfwords
            {
                const int nd = nv * GH;
                unsigned* etd = reinterpret_cast<unsigned*>(et);
                // (the 13 row / column pairs depend on the lane only: hoisted out of the tile loop they were 13 more registers
                // for a kernel that has none — spilled, and reloaded here behind a vmcnt(0) each; the empty asm makes them per-tile)
                int lane_t = lane;
                asm volatile("" : "+v"(lane_t));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int d = lane_t + WAVE * j;
                    const int row = (d * 1311) >> 15;            // d / 25, exact below 800
                    const int col = d - row * GH;
                    const unsigned v = stash[j * WAVE + lane_t];
                    if (d < 32 * GH) etd[row * (EKS / 2) + col] = d < nd ? v : 0u;    // (rows past the tile: zeros, never stale staging bytes)
                }
                // columns G (the constant 1 of the bias), G + 1 .. 63 (zeros): the e-tile region is the activations' staging area
                // in the second half of the tile, so they are rewritten with every tile — dwords 25 .. 31 of the 32 rows
#pragma unroll
                for (int j = 0; j < (32 * 7 + WAVE - 1) / WAVE; ++j) {
                    const int d = lane_t + WAVE * j;
                    const int row = (d * 9363) >> 16;            // d / 7, exact below 224
                    const int c = d - row * 7;
                    if (d < 32 * 7) etd[row * (EKS / 2) + GH + c] = c == 0 ? 0x00003F80u : 0u;
                }
                if (lane_t < 32) reinterpret_cast<unsigned short*>(mybuf + OFF_TSL)[lane_t] = valid_i ? (unsigned short)((c_tg - n0) << 7) : (unsigned short)0xffff;
            }
            // ---- this tile's h[src] chunks, then the next tile's operands: all in flight under GEMM1
            const bf16_t* const hrow = p.h + (int64_t)srcn * F;
            // (lane half h holds units 8 h .. 8 h + 7 and 16 + 8 h .. 16 + 8 h + 7 of every 32-unit block — see the row order of the
            // packed weights — so a lane's share of its h row is TWO 16-byte pieces per block: half the L2 requests of 8-byte chunks)
            // (last block: the 16 bytes are read where they END no later than the row; the shift that puts the piece's dwords in
            // place is applied at the use, behind the wait — anything that touches a loaded register here would wait for it)
            u32x4 hq[NBK][2];
            int h_t = h;                                            // (per-tile copy: lane-only values hoisted out of the tile loop get
            asm volatile("" : "+v"(h_t));                           // spilled, and a spill reload waits for every load in flight)
#pragma unroll
            for (int b = 0; b < NBK; ++b)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int u0 = 32 * b + 16 * t + 8 * h_t;
                    hq[b][t] = *reinterpret_cast<const u32x4*>(hrow + (b < NBK - 1 ? u0 : min(u0, F - 8)));
                }
            Pre nxt;
            request(nxt, min(eb + nv, Et - 1));                          // (unconditional: past the last edge it re-reads the last row)
            wave_lds_fence();
            CF_TMARK(0);

            // ---- GEMM1 + ssp.  Row rho of a weight block holds unit pi(rho) = rho with bits 2 and 3 exchanged, so accumulator
            // register r of lane half h (row (r & 3) + 8 (r >> 2) + 4 h) is unit 16 (r >> 3) + 8 h + (r & 7) of the block: the packed
            // registers are k-slots 8 h .. 8 h + 7 of fragments 2 b and 2 b + 1 in NATURAL unit order (W2p's K columns need no
            // permutation) and, as 8-byte chunks (registers 4 q .. 4 q + 3), rows 4 (q >> 1) + 2 h + (q & 1) of the chunk buffer
            unsigned ad[NBK][8];
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int k = 0; k < KE / 16; ++k)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w1l, 32 * b + i, W1S, 16 * k, h), ld_frag(et, i, EKS, 16 * k, h), acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    ad[b][q] = pk_bf16(LN2_F * (GT::softplus_u(acc[2 * q]) - 1.0f), LN2_F * (GT::softplus_u(acc[2 * q + 1]) - 1.0f));
            }
            // every load of this tile (and the next tile's operands) has had GEMM1 to arrive; from here on the tile only stores
            CF_TMARK(1);
#ifndef MDL_CF_NOWAIT
            __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0)
#endif
            CF_TMARK(2);
            park(nxt);
            // one-hot operands of the by-target reduction (k-slot q of k-step ks = edge slot 16 ks + 8 (q >> 2) + 4 h + (q & 3))
            bf16x8 tf[2];
            {
                const u32x2* tw = reinterpret_cast<const u32x2*>(mybuf + OFF_TSL) + h;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x2 t0 = tw[4 * ks], t1 = tw[4 * ks + 2];
                    tf[ks] = __builtin_bit_cast(bf16x8, u32x4{oh2x(t0[0] ^ rsplat), oh2x(t0[1] ^ rsplat), oh2x(t1[0] ^ rsplat), oh2x(t1[1] ^ rsplat)});
                }
            }


            // The activations leave through the 2.3-KB chunk buffer: a lane owns an EDGE, so stores from the accumulator layout
            // are 8-byte pieces of 32 different rows per instruction (37 L2 transactions per edge and tensor: measured +245 us
            // for the two tensors).  Staged as chunks and read back row-wise, a lane writes 16 bytes and four neighbouring lanes
            // one 64-byte run of a row: a quarter of the transactions.
            auto store_block = [&](const char* buf, bf16_t* dst, int b) {
                const int pc = lane & 3;
                const int u0 = 32 * b + 8 * pc;                        // first unit of my 16-byte piece
                const int ndw = min(4, max(0, (F - u0) >> 1));         // dwords of it that exist
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int r = (lane >> 2) + 16 * half;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(buf + (2 * pc) * DCS + r * 8);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(buf + (2 * pc + 1) * DCS + r * 8);
                    if (r < nv) {
#ifdef MDL_CF_SMALLDST
                        unsigned* g = reinterpret_cast<unsigned*>(dst + (int64_t)((eb + r) & 4095) * F + u0);   // (A/B: stores that stay in L2)
#else
                        unsigned* g = reinterpret_cast<unsigned*>(dst + (int64_t)(eb + r) * F + u0);
#endif
                        if (b < NBK - 1 || ndw == 4) {
                            // (non-temporal stores measured: 522 -> 950 us — the 64-byte runs need the L2 to merge them into lines)
                            *reinterpret_cast<u32x4*>(g) = u32x4{lo[0], lo[1], hi[0], hi[1]};
                        } else {
                            if (ndw >= 1) g[0] = lo[0];
                            if (ndw >= 2) g[1] = lo[1];
                            if (ndw >= 3) g[2] = hi[0];
                        }
                    }
                }
            };
            // Staging buffers: the e-tile region (GEMM1 has read it; the next tile's commit rewrites it) holds TWO chunk buffers, so
            // a block's chunks are written while the block before it is still being read back and stored, and in the GEMM2 loop the
            // filter's staging does not share a buffer with the message transpose.  One compiler fence between a buffer's writes and
            // its reads; the wave's LDS queue is in order, so a later write never passes an earlier read.
            char* const sb0 = mybuf + OFF_ET;
            char* const sb1 = mybuf + OFF_ET + 8 * DCS;
            static_assert(2 * 8 * DCS <= 32 * EKS * 2, "two staging buffers fit the e-tile region");
            if (p.a1) {
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    char* const sb = (b & 1) ? sb1 : sb0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<u32x2*>(sb + (4 * (q >> 1) + 2 * h + (q & 1)) * DCS + i * 8) = u32x2{ad[b][2 * q], ad[b][2 * q + 1]};
                    wave_lds_fence();
                    store_block(sb, p.a1, b);
                }
                wave_lds_fence();
            }
            CF_TMARK(3);
            // unit FP - 1 (= 16 + 8 + 7 of the last block: lane half 1, register 15) is the constant 1 that carries the bias of layer 2
            if (h == 1) ad[NBK - 1][7] = (ad[NBK - 1][7] & 0x0000ffffu) | 0x3F800000u;
            bf16x8 af[2 * NBK];
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                af[2 * b] = __builtin_bit_cast(bf16x8, u32x4{ad[b][0], ad[b][1], ad[b][2], ad[b][3]});
                af[2 * b + 1] = __builtin_bit_cast(bf16x8, u32x4{ad[b][4], ad[b][5], ad[b][6], ad[b][7]});
            }

            // ---- GEMM2, messages, by-target reduction: one 32-unit block at a time
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < 2 * NBK; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(w2l, 32 * b + i, W2S, 16 * ks, h), af[ks], acc, 0, 0, 0);
                unsigned wd[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) wd[q] = pk_bf16(acc[2 * q], acc[2 * q + 1]);
                char* const sbw = (b & 1) ? sb1 : sb0;
                if (p.w) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<u32x2*>(sbw + (4 * (q >> 1) + 2 * h + (q & 1)) * DCS + i * 8) = u32x2{wd[2 * q], wd[2 * q + 1]};
                }
                u32x2 hv[4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    u32x4 v = hq[b][t];
                    if (b == NBK - 1) {
                        // the piece's dword k is dword k + sh of what was read; sh is one scalar per lane half (0 inside the row, 4: none)
                        const int U0 = 32 * b + 16 * t;
                        const int sh0 = min(4, (U0 - min(U0, F - 8)) >> 1), sh1 = min(4, (U0 + 8 - min(U0 + 8, F - 8)) >> 1);
                        auto pick = [&](int idx) -> unsigned { return idx == 0 ? v[0] : idx == 1 ? v[1] : idx == 2 ? v[2] : idx == 3 ? v[3] : 0u; };
                        u32x4 a, c;
#pragma unroll
                        for (int k = 0; k < 4; ++k) { a[k] = pick(k + sh0); c[k] = pick(k + sh1); }
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = h ? c[k] : a[k];
                    }
                    hv[2 * t] = u32x2{v[0], v[1]};
                    hv[2 * t + 1] = u32x2{v[2], v[3]};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned w0 = wd[2 * q], w1 = wd[2 * q + 1];
                    // message = the ROUNDED filter (what the unfused path multiplies) * h[src] * cutoff, scaled by 2^64
                    const float m0 = (__uint_as_float(w0 << 16) * __uint_as_float(hv[q][0] << 16)) * cu;
                    const float m1 = (__uint_as_float(w0 & 0xffff0000u) * __uint_as_float(hv[q][0] & 0xffff0000u)) * cu;
                    const float m2 = (__uint_as_float(w1 << 16) * __uint_as_float(hv[q][1] << 16)) * cu;
                    const float m3 = (__uint_as_float(w1 & 0xffff0000u) * __uint_as_float(hv[q][1] & 0xffff0000u)) * cu;
                    *reinterpret_cast<u32x2*>(dp + (4 * (q >> 1) + 2 * h + (q & 1)) * DCS + i * 8) = u32x2{pk_bf16(m0, m1), pk_bf16(m2, m3)};
                }
                wave_lds_fence();
                {
                    const char* a = dp + (4 * (i >> 4) + (t16 & 3)) * DCS + (4 * h + (t16 >> 2)) * 8;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 128 * ks));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(a + 128 * ks + 64));
                        const bf16x8 mf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        oacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf[ks], mf, oacc[b], 0, 0, 0);
                    }
                }
                if (p.w) store_block(sbw, p.w, b);                 // (behind the same fence as the transposed reads: they travel together)
                wave_lds_fence();
            }
            c_src = nxt.src; c_tg = nxt.tg; c_cu = nxt.cu;
            CF_TMARK(4);
            CF_TTILE();
        }
        // ---- the group's rows: lane = unit, registers = node slots (D layout)
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            const int unit = 32 * b + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + d_row(r, h);
                if (n < n1 && (b < NBK - 1 || unit < F)) p.out[(int64_t)n * F + unit] = f2bf(oacc[b][r] * RW);
            }
        }
        n0 = n1;
        e0 = e1;
        CF_TMARK(5);
    }
    CF_TFLUSH();
}

// W1 [F, G], b1 [F], W2 [F, F], b2 [F] (fp32 masters) -> W1p [FP][W1S] (scaled by log2 e for the base-2 softplus, bias in column
// G) | W2p [FP][W2S], the bias in the K slot of unit FP - 1, which the kernel sets to 1.  ROWS of both: row rho of a 32-row block
// holds unit pi(rho) (bits 2 and 3 exchanged) — a lane half of the accumulator then owns 8 + 8 consecutive units per block
__global__ __launch_bounds__(256) void cfconv_pack_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ w2, const float* __restrict__ b2, int F,
                                                          bf16_t* __restrict__ wpack) {
    const int FP = 32 * nbk_for(F), W2S = FP + 8;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    auto pi = [](int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); };     // row of a 32-row block -> unit of the block
    if (idx < FP * W1S) {
        const int prow = idx / W1S, col = idx - prow * W1S;
        const int row = pi(prow);
        float v = 0.0f;
        if (row < F) {
            if (col < G_) v = w1[row * G_ + col];
            else if (col == G_ && b1) v = b1[row];
        }
        wpack[idx] = f2bf(v * LOG2E_F);
    } else if (idx < FP * W1S + FP * W2S) {
        const int j = idx - FP * W1S;
        const int prow = j / W2S, pos = j - prow * W2S;
        const int n = pi(prow);
        float v = 0.0f;
        if (n < F && pos < FP) {
            const int unit = pos;                                   // GEMM1 leaves the units in natural order (row permutation pi)
            if (unit < F) v = w2[n * F + unit];
            else if (unit == FP - 1 && b2) v = b2[n];
        }
        wpack[idx] = f2bf(v);
    }
}

}  // namespace cf
}  // namespace mdl

using namespace mdl;

#ifdef MDL_CF_TIMING
extern "C" int mdl_debug_read_cf(long long* host16) {
    const hipError_t e = hipMemcpyFromSymbol(host16, HIP_SYMBOL(cf::g_cf_dbg), 16 * sizeof(long long));
    static const long long zero[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(cf::g_cf_dbg), zero, sizeof(zero));
    return e == hipSuccess ? 0 : 1;
}
#endif
extern "C" size_t mdl_cfconv_wpack_bytes(void) { return (size_t)cf::Shape<5>::W1_BYTES + cf::Shape<5>::W2_BYTES; }     // (the widest layout)

extern "C" int mdl_cfconv_supported(int F, int G, int dtype) {
    return dtype == MDL_BF16 && G == cf::G_ && F >= 64 && F <= 158 && F % 2 == 0;
}

extern "C" int mdl_cfconv_pack_weights(const float* w1, const float* b1, const float* w2, const float* b2, int F, int G, void* wpack,
                                       mdlStream_t stream) {
    MDL_REQUIRE(mdl_cfconv_supported(F, G, MDL_BF16), MDL_E_UNSUPP, "mdl_cfconv_pack_weights: F = %d, G = %d (supported: G = 50, even F in [64, 158])", F, G);
    MDL_REQUIRE(w1 && w2 && wpack, MDL_E_ARG, "mdl_cfconv_pack_weights: null weights");
    const int FP = 32 * cf::nbk_for(F);
    const int total = FP * cf::W1S + FP * (FP + 8);
    hipLaunchKernelGGL(cf::cfconv_pack_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w1, b1, w2, b2, F,
                       static_cast<bf16_t*>(wpack));
    return check_launch("mdl_cfconv_pack_weights");
}

template <int NBK>
static int cfconv_launch(const cf::Params& p, int64_t grid, hipStream_t st) {
    auto kf = cf::cfconv_fwd_kernel<NBK>;
    constexpr int lds = cf::Shape<NBK>::LDS;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
    if (e != hipSuccess) { set_error("mdl_cfconv_fwd: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(cf::NT), lds, st, p);
    return check_launch("mdl_cfconv_fwd");
}

extern "C" int mdl_cfconv_fwd(const void* rbf, const float* cut, const void* h, const int32_t* rowptr, const int32_t* src,
                              const int32_t* tgt, const void* wpack, void* out, void* a1, void* w, int64_t N, int64_t E, int F, int G,
                              int dtype, mdlStream_t stream) {
    MDL_REQUIRE(mdl_cfconv_supported(F, G, dtype), MDL_E_UNSUPP, "mdl_cfconv_fwd: bf16, G = 50 and even F in [64, 158] only (F = %d, G = %d, dtype %d)", F, G, dtype);
    MDL_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) && E * (int64_t)F < (1ll << 40) && E < (1ll << 31), MDL_E_ARG, "mdl_cfconv_fwd: sizes out of range");
    if (N == 0) return MDL_OK;
    MDL_REQUIRE(rowptr && out && wpack && h, MDL_E_ARG, "mdl_cfconv_fwd: null argument");
    MDL_REQUIRE(E == 0 || (rbf && cut && src && tgt), MDL_E_ARG, "mdl_cfconv_fwd: null edge arrays");
    MDL_REQUIRE(((uintptr_t)rbf % 4) == 0 && ((uintptr_t)h % 4) == 0 && ((uintptr_t)a1 % 4) == 0 && ((uintptr_t)w % 4) == 0 && ((uintptr_t)wpack % 16) == 0,
                MDL_E_ARG, "mdl_cfconv_fwd: misaligned tensor");
    cf::Params p{static_cast<const bf16_t*>(rbf), cut, static_cast<const bf16_t*>(h), rowptr, src, tgt, static_cast<const bf16_t*>(wpack),
                 static_cast<bf16_t*>(out), static_cast<bf16_t*>(a1), static_cast<bf16_t*>(w), (int)N, (int)E, F};
    // one 8-wave workgroup per CU; small problems: about two tiles per wave at least
    const int64_t grid = std::min<int64_t>(256, std::max<int64_t>(1, cdiv(E + N, 32 * cf::NWAVE * 2)));
    switch (cf::nbk_for(F)) {
        case 3: return cfconv_launch<3>(p, grid, (hipStream_t)stream);
        case 4: return cfconv_launch<4>(p, grid, (hipStream_t)stream);
        default: return cfconv_launch<5>(p, grid, (hipStream_t)stream);
    }
}
