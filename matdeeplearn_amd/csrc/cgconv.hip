// cgconv.hip — K2/K3: fused CGConv forward and backward edge pass for gfx950 (CDNA4, wave64).
//
// Replaces torch_geometric.nn.CGConv (2.0.1) as used by /root/reference/matdeeplearn/models/cgcnn.py:80-83
// (construction) and :136-145 (call):  z = [x_i | x_j | e_ij];  m = sigmoid(W_f z + b_f) * softplus(W_s z + b_s);
// out_i = x_i + mean_{j->i} m.   The reference path materialises z (E x (2C+G)), both gate
// pre-activations (E x C each) and the message (E x C) in HBM and aggregates with atomics.  Here
// nothing per-edge is ever written:
//
//   partition  = one contiguous, edge-balanced NODE range per wave (NodeRange; the backward can instead take 32-node
//                groups from a counter in the caller's workspace), walked in groups of <= 32 target nodes
//   work item  = (group, 32-channel slice) -> one wave, no barriers; the static-shape forward handles all slices
//                of a group in one wave ("all-slices": the tile staging is shared)
//   edge tile  = 32 consecutive CSR slots of the group (edges are sorted by target)
//   pre        = z_tile (32 x KT) * Wpack^T (KT x 64)      MFMA 32x32, K order [e | x_tgt | x_src]
//                  e_tile: streamed coalesced HBM -> per-wave LDS (the dominant HBM stream)
//                  x rows: gathered straight into A fragments (L1/L2 hits: neighbours are in-graph)
//                  Wpack : staged once per workgroup in LDS, read as B fragments
//   gate       = VALU on the accumulator registers (lane = channel, 16 edge slots per lane)
//   aggregate  = second MFMA with a one-hot "slot == node" A matrix: the segmented reduction over
//                the tile's edges lands in a 32-node x 32-channel register accumulator (no atomics,
//                deterministic), epilogue adds the residual and divides by the in-degree.
//
// Backward recomputes pre (no E x 2C activations are stored), expands grad_out to edges with a
// one-hot MFMA, forms dpre = d/d(pre) on registers and reduces it three ways:
//   r_tgt (by target: MFMA, registers), r_src (by source: one-hot MFMA into a 64-node register window, fp32 atomics
//   outside it), dwe = dpre^T e (MFMA).  The node-level dense products (dx, dW_tgt, dW_src) are cgconv_node.hip.
//
// Load discipline of the tile loops (it is worth 10-15 % of either kernel): every prefetch is issued on EVERY path —
// uniform selects of the tile base, clamped indices, buffer loads whose range check replaces the end-of-array path —
// because hipcc derives each s_waitcnt from the worst path into it: one path that skips the prefetches makes the waits
// in front of the x-fragment MFMAs drain the loads issued a few instructions earlier.
// cgconv_cb.inc (included below) holds a second, cooperative weight-stationary design of both kernels (opt-in).
// DESIGN.md section 4 has the measured phase breakdown and the list of variants behind the MDL_* switches below.
//
// dtype MDL_BF16: v_mfma_f32_32x32x16_bf16, fast gate math.  MDL_F32 (parity mode):
// v_mfma_f32_32x32x2_f32 (bit-exact fp32 fma chain), precise gate math.
//
// Algorithmic bytes: fwd E*(G*s + C*s + 4) + N*(2*C*s + 4);
//                                   bwd E*(G*s + 2*C*s + 4) + N*(3*C*s + 4).
#include <stdlib.h>
#include <type_traits>

#include "mdl_common.h"

#ifndef MDL_EXPERIMENTS
#define MDL_EXPERIMENTS 0   // 1: the experiments build (experiments/build.py): measured-negative kernel variants and their
                            // environment switches are compiled in; libmdl_hip.so itself never reads the environment
#endif
#ifndef MDL_CG_WM
#define MDL_CG_WM 1       // where the static bf16 kernels keep W: 1 LDS, 2 registers, 3 x-part registers + e-part LDS
#endif
#ifndef MDL_FWD_XDB
#define MDL_FWD_XDB 0     // 1: x-row gathers of tile t+1 in flight during tile t (costs 32 VGPRs)
#endif
#ifndef MDL_BWD_XDB
#define MDL_BWD_XDB 1
#endif
#ifndef MDL_FWD_ALLSLICES
#ifndef MDL_FWD_AGE_SKEW
#define MDL_FWD_AGE_SKEW 60     // all-slices forward at two workgroups per CU: per mille of extra work for the older half
#endif
#define MDL_FWD_ALLSLICES 1   // static shapes: one forward wave handles all channel slices of its group
#endif
#ifndef MDL_FWD_XEARLY
#define MDL_FWD_XEARLY 1   // all-slices forward: gather the next tile's x rows right after the last slice's MFMAs
#endif
#ifndef MDL_CG_PHASE_BARRIERS
#define MDL_CG_PHASE_BARRIERS 0
#endif
#ifndef MDL_CG_CB_DEFAULT
#define MDL_CG_CB_DEFAULT 0   // 1: cooperative column-block kernels for the static bf16 shapes
#endif
#ifndef MDL_CG_CB_BWD_DEFAULT
#define MDL_CG_CB_BWD_DEFAULT 0   // 1: cooperative column-block backward edge pass for the static bf16 shapes
#endif
#ifndef MDL_CB_FWD_WG_PER_CU
#define MDL_CB_FWD_WG_PER_CU 2
#endif
#ifndef MDL_FWD_PRE_DEPTH
#define MDL_FWD_PRE_DEPTH 3   // all-slices forward: pinned LDS-read / MFMA interleave in pre_tile, reads issued ahead (-6 %)
#endif
#ifndef MDL_BWD_DERIV2
#define MDL_BWD_DERIV2 0  // 1: bf16 backward with the select-free gate derivative (Gate<true>::deriv2, 3 VALU fewer per element): measured +-0
#endif
#ifndef MDL_FWD_RANGE_EDGES
#define MDL_FWD_RANGE_EDGES 64   // edges per node range (= per wave) below which the launch shrinks instead: two 32-edge tiles
#endif
#ifndef MDL_BWD_RANGE_EDGES
#define MDL_BWD_RANGE_EDGES 128  // (64 -> 128: -7 of 45 us at the reference's batch size — half as many waves flush their weight-gradient sums;
                                 // from 6.5e4 edges on the grid is capped at one workgroup per CU either way)
#endif
#ifndef MDL_BWD_WAVES
#define MDL_BWD_WAVES 1   // waves per SIMD the backward kernel is register-allocated for
#endif

#ifdef MDL_CG_EP_TU      // compiled a second time as cgconv_ep.hip (see the include of cgconv_ep.inc below): own debug symbols
#define mdl_debug_life mdl_debug_life_ep
#define mdl_debug_read mdl_debug_read_ep
#define mdl_debug_reset mdl_debug_reset_ep
#endif
#ifdef MDL_CG_TIMING   // experiment builds only: per-phase cycle counters of wave 0 (kept in SGPRs, flushed at the end)
__device__ long long g_cg_dbg[48];
__device__ long long g_cg_life[2][4096][3];     // [fwd|bwd][wave] = wall start, wall end, tiles (last launch)
extern "C" int mdl_debug_life(long long* host, int which) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cg_life), sizeof(long long) * 4096 * 3, sizeof(long long) * 4096 * 3 * which);
}
extern "C" int mdl_debug_read(long long* host48) {
    return (int)hipMemcpyFromSymbol(host48, HIP_SYMBOL(g_cg_dbg), 48 * sizeof(long long));
}
extern "C" int mdl_debug_reset() {
    long long z[48] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cg_dbg), z, sizeof(z));
}
#define TDECL const long long tstart = clock64(), wstart = wall_clock64(); long long tprev = tstart; long long tacc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; long long tcount = 0
#define TMARK(k) do { __builtin_amdgcn_sched_barrier(0); const long long _t = clock64(); tacc[k] += _t - tprev; tprev = _t; __builtin_amdgcn_sched_barrier(0); } while (0)
#define TPIN16(v) do { _Pragma("unroll") for (int _r = 0; _r < 16; ++_r) asm volatile("" : "+v"(v[_r])); } while (0)
#define TTILE() (tcount += 1)
#define TRESET() TMARK(10)
#define TFLUSH(base) do { if (gw == 0 && lane == 0) { for (int _k = 0; _k < 13; ++_k) g_cg_dbg[(base) + _k] += tacc[_k]; g_cg_dbg[(base) + 15] += tcount; g_cg_dbg[(base) + 14] += clock64() - tstart; g_cg_dbg[(base) + 13] += wall_clock64() - wstart; } if (lane == 0 && gw < 4096) { g_cg_life[(base) / 16][gw][0] = wstart; g_cg_life[(base) / 16][gw][1] = wall_clock64(); g_cg_life[(base) / 16][gw][2] = tcount; } } while (0)
#else
#define TDECL do { } while (0)
#if MDL_CG_PHASE_BARRIERS   // keep the phases of a tile apart in the instruction schedule (no timing)
#define TMARK(k) __builtin_amdgcn_sched_barrier(0)
#define TPIN16(v) do { _Pragma("unroll") for (int _r = 0; _r < 16; ++_r) asm volatile("" : "+v"(v[_r])); } while (0)
#else
#define TMARK(k) do { } while (0)
#define TPIN16(v) do { } while (0)
#endif
#define TTILE() do { } while (0)
#define TRESET() do { } while (0)
#define TFLUSH(base) do { } while (0)
#endif


namespace mdl {

struct CgParams {
    const void* x;
    const void* ea;
    const int32_t* rowptr;
    const int32_t* src;
    const int32_t* tgt;
    const int32_t* eperm;
    const void* wpack;
    const float* bpack;
    void* out;          // fwd
    const void* gout;   // bwd
    void* r_tgt;        // bwd [N, 2Cp] in the compute dtype (written once per node)
    float* r_src;       // bwd [N, 2Cp]
    float* dwe;         // bwd [2Cp, GP]
    float* db;          // bwd [2Cp] bias gradient = column sums of r_tgt (may be null)
    unsigned* ctr;      // bwd, optional: NS zeroed work counters (caller workspace) -> dynamic group scheduling
    void* ab;           // saved gate factors [E][Cp][2] bf16 (A | B per channel): written by the training forward, read by
                        // the saved-gate backward (cgconv_bwd_ab_kernel)
    const int32_t* balance;   // bwd, optional: [N + 1] non-decreasing cost prefix the workgroups' node ranges are balanced on
                              // (mdl_cgconv_balance); null: edges + nodes in front of a node
    int ldwe;           // bwd: leading dimension of dwe in floats (0: GP) — MdlCgConv.ld_dwe
    int dwe_combine;    // bwd, per-wave kernel: the two waves of a workgroup that share a channel slice combine their dwe sums in LDS
    int rs16;           // bwd, bf16: r_src is a bf16 array accumulated with packed bf16 atomics (mdl_cgconv_bwd_h)
    int flags;          // host side: MDL_DETERMINISTIC / MDL_K3_* bits the caller OR-ed into `dtype`
    // fwd, optional (mdl_cgconv_fwd_ex): statistics of the layer's OUTPUT for the training-mode BatchNorm1d behind it (cgcnn.py:143)
    // in the epilogue — per column sum (v - shift) and sum (v - shift)^2 of the ROUNDED outputs over the rows that exist, into
    // one of the MDL_BN_REPLICAS copies of the sums (layout of mdl_bn_stats; caller zero-fills); the shift row the sums are
    // about is published behind the sums' totals rows for mdl_bn_apply_n(... | MDL_BN_SHIFT_ROW)
    float* bn_sums;
    const float* bn_shift;     // [C] fp32 or null (= 0): any per-column value near the column mean (the previous BatchNorm's beta)
    const int64_t* bn_nrows;   // device row count of a padded static batch, or null (= N)
    const void* pt;     // W-split kernels: per-node projections P_t = x [W_f,tgt ; W_s,tgt]^T and P_s = x [W_f,src ; W_s,src]^T,
    const void* ps;     // [N, 2Cp] each in the compute dtype (columns f | s), scaled like the packed weights
    int64_t N, E;
    int C, G, Cp, KE, KT, WS, EKS, NS, GP, aggr;
    int GW;             // staging words per e row
    unsigned gw_inv;    // ceil(2^32 / GW)
    int n_groups;
    int g_full;         // bwd, dynamic scheduling: group ids < g_full are 32-node groups, the rest 16-node half groups (tail)
    int w_elems;        // 2*Cp*WS (w_slice: 64*WS)
    int w_slice;        // 1: the workgroup's LDS copy of W holds only the 64 rows of its channel slice (all its waves share the
                        // slice) — wide layers whose packed weights do not fit 160 KB (C = 100: 168 KB) still read them from LDS
    int wave_lds_bytes; // per-wave LDS region
    int bias_col;       // 1: bias lives in K column G of wpack (e tile column G holds 1.0)
};

struct CgDims {
    int Cp, KE, KT, WS, EKS, NS, GP;
};

static inline int rup(int a, int b) { return (a + b - 1) / b * b; }

// x3 (MDL_SPLIT_BF16, fp32 storage): the K = 2C + G product runs as three bf16 MFMAs on split operands, whose fragments are
// 16-byte LDS reads — rows padded by FOUR dwords (an odd number of 16-byte slots) instead of one
static CgDims cg_dims(int C, int G, int dtype, bool x3 = false) {
    CgDims d;
    d.Cp = rup(C, 32);
    d.KE = rup(G, 16);
    d.KT = d.KE + 2 * d.Cp;
    const int pad = dtype == MDL_BF16 ? 8 : (x3 ? 4 : 1);   // bf16: odd number of 16-B slots; f32: odd dword stride
    d.WS = d.KT + pad;
    d.EKS = d.KE + pad;
    d.NS = d.Cp / 32;
    d.GP = rup(G, 64);
    return d;
}

// ------------------------------------------------------------------------------------------
// MFMA traits
// ------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KSTEP = 16;
    static constexpr bool FAST = true;
    typedef bf16x8 frag_t;
    __device__ static __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ frag_t zero() { return frag_t{0, 0, 0, 0, 0, 0, 0, 0}; }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 2;
    static constexpr bool FAST = false;
    typedef float frag_t;
    __device__ static __forceinline__ f32x16 mma(frag_t a, frag_t b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ frag_t zero() { return 0.0f; }
};

// A/B fragment of a row-major [rows][ld] matrix held in LDS or global memory:
// lane (i = lane&31, h = lane>>5) takes row i, columns k0 + KSTEP/2*h .. (8 bf16 / 1 float).
__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* base, int row, int ld, int k0, int h) {
    return *reinterpret_cast<const bf16x8*>(base + row * ld + k0 + 8 * h);
}
__device__ __forceinline__ float ld_frag(const float* base, int row, int ld, int k0, int h) {
    return base[row * ld + k0 + h];
}

// x-row fragment gathered from global memory with column bound C (columns >= C read as 0).
template <int VEC>
__device__ __forceinline__ bf16x8 ld_xfrag(const bf16_t* rowp, int c0, int h, int C) {
    const int c = c0 + 8 * h;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (VEC == 9) {            // 16-byte loads, channel count == padded count: no column mask
        v = *reinterpret_cast<const bf16x8*>(rowp + c);
    } else if (VEC == 8) {
        if (c < C) v = *reinterpret_cast<const bf16x8*>(rowp + c);
    } else if (VEC == 4) {
        bf16x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (c < C) lo = *reinterpret_cast<const bf16x4*>(rowp + c);
        if (c + 4 < C) hi = *reinterpret_cast<const bf16x4*>(rowp + c + 4);
        v = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c + j < C) v[j] = (short)rowp[c + j];
    }
    return v;
}
template <int VEC>
__device__ __forceinline__ float ld_xfrag(const float* rowp, int c0, int h, int C) {
    const int c = c0 + h;
    if (VEC == 9) return rowp[c];
    return c < C ? rowp[c] : 0.0f;
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ bf16x8 pack_bf16x8(const float* v) {
    u32x4 r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, r);
}

// ------------------------------------------------------------------------------------------
// x3: fp32 values as (hi, lo) bf16 pairs — v = hi + lo to 16 significant bits (both parts rounded to nearest), so that
// a * b ~ a_hi b_hi + a_lo b_hi + a_hi b_lo on the bf16 matrix core at 1/5 of the cost of the exact-fp32 MFMA
// (3 x 32 cycles per 16 k-values against 8 x 64): relative error 2^-16 per product, fp32 accumulation.
// ------------------------------------------------------------------------------------------
struct SplitFrag { bf16x8 hi, lo; };
__device__ __forceinline__ void split_pair(float v0, float v1, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(v0, v1);
    const float r0 = v0 - __builtin_bit_cast(float, hi << 16);
    const float r1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pk_bf16(r0, r1);
}
__device__ __forceinline__ SplitFrag split8(const f32x4& a, const f32x4& b) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split_pair(a[0], a[1], h0, l0);
    split_pair(a[2], a[3], h1, l1);
    split_pair(b[0], b[1], h2, l2);
    split_pair(b[2], b[3], h3, l3);
    return SplitFrag{__builtin_bit_cast(bf16x8, u32x4{h0, h1, h2, h3}), __builtin_bit_cast(bf16x8, u32x4{l0, l1, l2, l3})};
}
// eight consecutive fp32 values of a row (16-byte aligned: LDS tile rows of EKS = KE + 4 dwords, or a global x row)
__device__ __forceinline__ SplitFrag split8_at(const float* p) {
    return split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4));
}
__device__ __forceinline__ f32x16 mma_x3(const SplitFrag& a, const SplitFrag& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
}
// B fragment of the x3 weight pack: a row holds, per group of eight k-values, [8 x hi bf16 | 8 x lo bf16] (32 bytes, the
// footprint of the eight floats it replaces); lane half h of k-step k0 / 16 takes group k0 / 8 + h
__device__ __forceinline__ SplitFrag ld_wfrag_x3(const float* wbase, int row, int ld, int k0, int h) {
    const char* g = reinterpret_cast<const char*>(wbase + row * ld) + (k0 / 8 + h) * 32;
    return SplitFrag{*reinterpret_cast<const bf16x8*>(g), *reinterpret_cast<const bf16x8*>(g + 16)};
}

// ------------------------------------------------------------------------------------------
// Shared per-tile machinery
// ------------------------------------------------------------------------------------------
// Problem dimensions: compile-time when the kernel is instantiated for a fixed (CP_, G_) — loops
// unroll fully, so all loads of a tile are issued before the first MFMA that needs them — or
// run-time (CP_ = 0 / G_ = 0) for the generic fallback.
template <typename T, int CP_, int G_, int EW, int WSP = 0, bool X3 = false>
struct Dims {
    static constexpr bool STATIC = (CP_ != 0) && (G_ != 0);
    static constexpr int PADW = std::is_same<T, bf16_t>::value ? 8 : (X3 ? 4 : 1);
    static_assert(!X3 || (std::is_same<T, float>::value && STATIC && WSP == 0), "x3: fp32 storage, static shapes");
    int C, Cp, G, KE, WS, EKS, GW;
    __device__ __forceinline__ Dims(const CgParams& p) {
        C = p.C;
        Cp = CP_ ? CP_ : p.Cp;
        G = G_ ? G_ : p.G;
        KE = G_ ? ((G_ + 15) / 16 * 16) : p.KE;
        WS = KE + (WSP ? 0 : 2 * Cp) + PADW;      // W-split: the packed weights hold the edge-feature part only
        EKS = KE + PADW;
        GW = G / EW;
    }
};

template <typename T>
struct WaveCtx {
    T* et;              // per-wave e tile  [32][EKS]
    unsigned* tsl;      // per-wave target-slot bytes (32 B) viewed as 8 dwords
    int* srcl;          // per-wave source ids (32 ints)   (backward only)
    unsigned* ssl;      // per-wave source-window slot bytes (32 B) (backward only)
    bf16_t* oh_t;       // one-hot tables (bf16 kernels), see oh_update()
    bf16_t* oh_e;
    bf16_t* oh_w;
    unsigned long long* touched;  // per-wave bitmap of window slots that received an edge (saved-gate kernel)
    unsigned char* touched_b;     // the same as 64 bytes (one per window slot: plain idempotent byte writes, read back with a ballot)
    int* dummy;                   // 64 dwords: where a lane's table writes go when it has nothing to write (branch-free updates)
    const T* wbase;     // packed weights (LDS or global)
};

template <typename T, int EW> struct StageWord {
    typedef typename std::conditional<EW * sizeof(T) == 4, unsigned, unsigned short>::type type;
};

// Generic (run-time G) staging of the 32 x G edge-feature tile: load -> LDS, word by word.
template <typename T, int EW, typename D>
__device__ __forceinline__ void stage_e_tile(const CgParams& p, const D& dm, const WaveCtx<T>& w, int lane, int eb,
                                             int nv, int my_ep) {
    typedef typename StageWord<T, EW>::type word_t;
    const int total = 32 * dm.GW;
    const T* ea = static_cast<const T*>(p.ea);
    for (int q0 = 0; q0 < total; q0 += WAVE) {
        const int q = q0 + lane;
        const bool act = q < total;
        const int row = act ? (int)__umulhi((unsigned)q, p.gw_inv) : 0;
        const int cw = q - row * dm.GW;
        const int ep = p.eperm ? __shfl(my_ep, row) : eb + row;
        if (act && row < nv) {
            const word_t v = *reinterpret_cast<const word_t*>(ea + (int64_t)ep * dm.G + cw * EW);
            *reinterpret_cast<word_t*>(w.et + row * dm.EKS + cw * EW) = v;
        }
    }
}

// Static-G staging, split in two halves so the HBM latency of tile t+1 hides under the compute of
// tile t: prefetch() issues the loads into registers, commit() writes them to the wave's LDS tile
// at the top of the next iteration.
template <typename T, int G_, int EW>
struct EWords {
    typedef typename StageWord<T, EW>::type word_t;
    static constexpr int GW = G_ ? G_ / EW : 1;
    static constexpr int NW = G_ ? (32 * GW + WAVE - 1) / WAVE : 1;
    word_t w[NW];

    // The static kernels are only launched for target-sorted edge features (no eperm; the host permutes
    // once).  Buffer loads: a fresh resource per tile (uniform base in SGPRs, range = the bytes that remain in
    // the array), ONE per-lane 32-bit offset, the word index j in the instruction's 12-bit immediate.  The constant is
    // written as part of the VOFFSET expression (the compiler splits it into register + immediate): the hardware range
    // check covers voffset + immediate only — an SGPR soffset is added AFTER the check, so rows addressed through it
    // would read (or, for stores, write) past the end of the array instead of being dropped.  No predication and no second
    // code path: rows past the end of the group belong to later edges (finite data, multiplied by exact zeros
    // downstream), words past the end of the array fail the range check and read as zeros.  (A clamped second
    // path for the last tile costs more than its instructions: every control-flow join in the tile loop makes
    // hipcc's wait-count bookkeeping assume the worse of the two paths.)
    __device__ __forceinline__ void prefetch(const CgParams& p, int lane, int eb, int /*nv*/, int /*my_ep*/) {
        constexpr int WB = EW * (int)sizeof(T);
        static_assert(WB == 4 || WB == 2, "staging word");
        const char* tb = reinterpret_cast<const char*>(p.ea) + (int64_t)eb * (G_ * (int)sizeof(T));
        const int64_t rem = (p.E - (int64_t)eb) * (G_ * (int)sizeof(T));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(tb), 0, (int)(rem < 0 ? 0 : (rem < 0x7fffffffLL ? rem : 0x7fffffffLL)), 0x00020000);
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if constexpr (WB == 4) w[j] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * WB + j * (WAVE * WB), 0, 0);
            else w[j] = __builtin_amdgcn_raw_buffer_load_b16(rs, lane * WB + j * (WAVE * WB), 0, 0);
        }
    }
    __device__ __forceinline__ void commit(T* et, int EKS, int lane) const {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int q = j * WAVE + lane;
            const int row = q / GW, cw = q - row * GW;
            if (row < 32) *reinterpret_cast<word_t*>(et + row * EKS + cw * EW) = w[j];
        }
    }
};

// x-row A fragments of one tile (target rows and source rows), all issued up front when the
// channel count is static.
template <typename T, int CP_, int VEC, bool X3 = false>
struct XFrags {
    typedef Mma<T> M;
    static constexpr int NF = CP_ ? CP_ / M::KSTEP : 1;
    typename M::frag_t t[NF], s[NF];
    __device__ __forceinline__ void load(const T* x, int C, int my_tgt, int my_src, int h) {
        const T* xt = x + (int64_t)my_tgt * C;
        const T* xs = x + (int64_t)my_src * C;
#pragma unroll
        for (int f = 0; f < NF; ++f) t[f] = ld_xfrag<VEC>(xt, f * M::KSTEP, h, C);
#pragma unroll
        for (int f = 0; f < NF; ++f) s[f] = ld_xfrag<VEC>(xs, f * M::KSTEP, h, C);
    }
};
// x3: the raw fp32 chunks of the two rows (16-byte loads; same 2 x CP_ / 2 registers as the k-pair fragments of the exact
// form): chunk f of a row = columns 16 f + 8 h .. + 7, split into (hi, lo) where it is used
template <int CP_, int VEC>
struct XFrags<float, CP_, VEC, true> {
    static constexpr int NF = CP_ / 16;
    f32x4 t[NF][2], s[NF][2];
    __device__ __forceinline__ void load(const float* x, int C, int my_tgt, int my_src, int h) {
        const float* xt = x + (int64_t)my_tgt * C + 8 * h;
        const float* xs = x + (int64_t)my_src * C + 8 * h;
#pragma unroll
        for (int f = 0; f < NF; ++f) { t[f][0] = *reinterpret_cast<const f32x4*>(xt + 16 * f); t[f][1] = *reinterpret_cast<const f32x4*>(xt + 16 * f + 4); }
#pragma unroll
        for (int f = 0; f < NF; ++f) { s[f][0] = *reinterpret_cast<const f32x4*>(xs + 16 * f); s[f][1] = *reinterpret_cast<const f32x4*>(xs + 16 * f + 4); }
    }
};

// W-split: z W^T = e W_e^T + P_t[tgt] + P_s[src] with per-node projections
// P = x [W_tgt | W_src]^T from ONE dense launch per layer, so that per edge only the K = 64 edge-feature product remains.
// The gathered projection rows enter the accumulators through the matrix core: they ARE A fragments (lane = edge, eight
// consecutive columns = 16 bytes, exactly like the x rows they replace) of a product with an IDENTITY B operand —
// D[edge][ch] += sum_k P[edge][k] (k == ch) — two k-steps per 32-channel block instead of the K = 64 (four k-steps) of the
// x part, no unpack / add VALU, no weight-fragment LDS reads.
// Fragment f of a slice: column (f >> 1) * Cp + 32 * slice + 16 * (f & 1) of the [f | s] row, i.e. (part f / s, k-step).
template <int CP_, int NSLF>
struct PFrags {
    bf16x8 t[4 * NSLF], s[4 * NSLF];
    __device__ __forceinline__ void load(const void* pt, const void* ps, int my_tgt, int my_src, int h, int sl0) {
        const bf16_t* a = static_cast<const bf16_t*>(pt) + (int64_t)my_tgt * (2 * CP_) + 8 * h + 32 * sl0;
        const bf16_t* b = static_cast<const bf16_t*>(ps) + (int64_t)my_src * (2 * CP_) + 8 * h + 32 * sl0;
#pragma unroll
        for (int q = 0; q < 4 * NSLF; ++q) t[q] = *reinterpret_cast<const bf16x8*>(a + ((q >> 1) & 1) * CP_ + 32 * (q >> 2) + 16 * (q & 1));
#pragma unroll
        for (int q = 0; q < 4 * NSLF; ++q) s[q] = *reinterpret_cast<const bf16x8*>(b + ((q >> 1) & 1) * CP_ + 32 * (q >> 2) + 16 * (q & 1));
    }
};
// identity B fragments of the 32 x 32 product in two k-steps: B[k][n] = (k == n), lane n, k = 16 ks + 8 h + q
__device__ __forceinline__ void identity_frags(int i, int h, bf16x8 (&idf)[2]) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4_t;
    const bool mine = ((i >> 3) & 1) == h;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const unsigned one = (mine && (i >> 4) == ks) ? (0x3F80u << (16 * (i & 1))) : 0u;
        const int d = (i & 7) >> 1;
        const u4_t v = {d == 0 ? one : 0u, d == 1 ? one : 0u, d == 2 ? one : 0u, d == 3 ? one : 0u};
        idf[ks] = __builtin_bit_cast(bf16x8, v);
    }
}
// pre-activation tile of slice `sl` (global) whose projection fragments sit at local slice `sll` of pf
template <int CP_, int NSLF, int DEPTH, typename D>
__device__ __forceinline__ void pre_tile_wsp(const D& dm, const WaveCtx<bf16_t>& w, int lane, int sl, int sll,
                                             const PFrags<CP_, NSLF>& pf, const bf16x8 (&idf)[2], f32x16& accf, f32x16& accs) {
    const int i = lane & 31, h = lane >> 5;
    const int rowf = sl * 32 + i, rows = dm.Cp + sl * 32 + i;
#pragma unroll
    for (int k0 = 0; k0 < dm.KE; k0 += 16) {
        const bf16x8 a = ld_frag(w.et, i, dm.EKS, k0, h);
        accf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ld_frag(w.wbase, rowf, dm.WS, k0, h), accf, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ld_frag(w.wbase, rows, dm.WS, k0, h), accs, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        accf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.t[4 * sll + ks], idf[ks], accf, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.t[4 * sll + 2 + ks], idf[ks], accs, 0, 0, 0);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        accf = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.s[4 * sll + ks], idf[ks], accf, 0, 0, 0);
        accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf.s[4 * sll + 2 + ks], idf[ks], accs, 0, 0, 0);
    }
    if constexpr (DEPTH > 0) {      // weight-fragment reads of the edge-feature part a few deep ahead of the chain
        __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
}

// Packed weights of this wave's channel slice held in registers for the whole kernel (static
// shapes): the B fragments of all K steps, f rows and s rows.  Removes every per-tile LDS read of W.
template <typename T, int NK>
struct WRegs {
    typename Mma<T>::frag_t f[NK], s[NK];
    __device__ __forceinline__ void load(const T* wpack, int rowf, int rows, int WS, int h, int k_first) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            f[k] = ld_frag(wpack, rowf, WS, k_first + k * Mma<T>::KSTEP, h);
            s[k] = ld_frag(wpack, rows, WS, k_first + k * Mma<T>::KSTEP, h);
        }
    }
};

// pre-activation tile: accf/accs (32 edge slots x 32 channels of slice s), bias pre-loaded.
template <typename T, int CP_, int VEC, int WM, int NKW, int DEPTH = 0, bool X3 = false, typename D>
__device__ __forceinline__ void pre_tile(const CgParams& p, const D& dm, const WaveCtx<T>& w, int lane, int s,
                                         int my_tgt, int my_src, const XFrags<T, CP_, VEC, X3>& xf,
                                         const WRegs<T, NKW>& wr, f32x16& accf, f32x16& accs) {
    typedef Mma<T> M;
    const int i = lane & 31, h = lane >> 5;
    const bool wsl = (CP_ == 0 || CP_ > 64) && p.w_slice;    // (never for the static shapes whose W fits LDS: folds away there)
    const int rowf = (wsl ? 0 : s * 32) + i, rows = (wsl ? 32 : dm.Cp + s * 32) + i;
    if constexpr (X3) {
        // fp32 storage, split-bf16 products: z fragments split where they are read (e tile: 2 ds_read_b128 per fragment; x
        // chunks: registers), weight fragments pre-split by the pack kernel (no arithmetic here)
        static_assert(WM == 1 && CP_ != 0, "x3: static shapes, W in LDS");
#pragma unroll
        for (int k0 = 0; k0 < dm.KE; k0 += 16) {
            const SplitFrag a = split8_at(w.et + i * dm.EKS + k0 + 8 * h);
            accf = mma_x3(a, ld_wfrag_x3(w.wbase, rowf, dm.WS, k0, h), accf);
            accs = mma_x3(a, ld_wfrag_x3(w.wbase, rows, dm.WS, k0, h), accs);
        }
        constexpr int NF = CP_ / 16;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const SplitFrag a = split8(xf.t[f][0], xf.t[f][1]);   // (both channel slices of a tile split the same chunks: the compiler merges them)
            accf = mma_x3(a, ld_wfrag_x3(w.wbase, rowf, dm.WS, dm.KE + 16 * f, h), accf);
            accs = mma_x3(a, ld_wfrag_x3(w.wbase, rows, dm.WS, dm.KE + 16 * f, h), accs);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const SplitFrag a = split8(xf.s[f][0], xf.s[f][1]);
            accf = mma_x3(a, ld_wfrag_x3(w.wbase, rowf, dm.WS, dm.KE + dm.Cp + 16 * f, h), accf);
            accs = mma_x3(a, ld_wfrag_x3(w.wbase, rows, dm.WS, dm.KE + dm.Cp + 16 * f, h), accs);
        }
        return;
    } else {
    if constexpr (CP_ != 0 && (WM == 2 || WM == 3)) {
        // static shapes.  WM 2: all B fragments live in registers.  WM 3: the x-part of W lives in
        // registers, the e-part is read from the LDS copy (those reads depend on nothing and are
        // issued at the top of the tile).
        constexpr int NF = XFrags<T, CP_, VEC>::NF;
        constexpr int NE = (WM == 2) ? NKW - 2 * NF : 0;
        if constexpr (WM == 2) {
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                typename M::frag_t a = ld_frag(w.et, i, dm.EKS, k * M::KSTEP, h);
                accf = M::mma(a, wr.f[k], accf);
                accs = M::mma(a, wr.s[k], accs);
            }
        } else {
#pragma unroll
            for (int k0 = 0; k0 < dm.KE; k0 += M::KSTEP) {
                typename M::frag_t a = ld_frag(w.et, i, dm.EKS, k0, h);
                accf = M::mma(a, ld_frag(w.wbase, rowf, dm.WS, k0, h), accf);
                accs = M::mma(a, ld_frag(w.wbase, rows, dm.WS, k0, h), accs);
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            accf = M::mma(xf.t[f], wr.f[NE + f], accf);
            accs = M::mma(xf.t[f], wr.s[NE + f], accs);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            accf = M::mma(xf.s[f], wr.f[NE + NF + f], accf);
            accs = M::mma(xf.s[f], wr.s[NE + NF + f], accs);
        }
        return;
    }
#ifdef MDL_CG_IGLP
    if constexpr (CP_ != 0) __builtin_amdgcn_iglp_opt(MDL_CG_IGLP);
#endif
    // edge features (LDS tile)
#pragma unroll
    for (int k0 = 0; k0 < dm.KE; k0 += M::KSTEP) {
        typename M::frag_t a = ld_frag(w.et, i, dm.EKS, k0, h);
        accf = M::mma(a, ld_frag(w.wbase, rowf, dm.WS, k0, h), accf);
        accs = M::mma(a, ld_frag(w.wbase, rows, dm.WS, k0, h), accs);
    }
    if constexpr (CP_ != 0) {
        constexpr int NF = XFrags<T, CP_, VEC>::NF;
#ifdef MDL_ABL_WSPLIT_BOUND
        // Upper bound of what the W-split (per-node projections P = x [W_tgt | W_src]^T added to the accumulators) can buy:
        // the x part of the product as TWO MFMAs per gathered row and accumulator (what the identity-operand form of the
        // split issues: K = 32 instead of 64) and WITHOUT their weight-fragment LDS reads; the gathers stay.  Results are
        // wrong by construction — timing only.
#pragma unroll
        for (int f = 0; f < NF / 2; ++f) {
            accf = M::mma(xf.t[f], xf.t[f + NF / 2], accf);
            accs = M::mma(xf.t[f], xf.t[f + NF / 2], accs);
        }
#pragma unroll
        for (int f = 0; f < NF / 2; ++f) {
            accf = M::mma(xf.s[f], xf.s[f + NF / 2], accf);
            accs = M::mma(xf.s[f], xf.s[f + NF / 2], accs);
        }
#else
#pragma unroll
        for (int f = 0; f < NF; ++f) {      // target-node features (x_i)
            accf = M::mma(xf.t[f], ld_frag(w.wbase, rowf, dm.WS, dm.KE + f * M::KSTEP, h), accf);
            accs = M::mma(xf.t[f], ld_frag(w.wbase, rows, dm.WS, dm.KE + f * M::KSTEP, h), accs);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {      // source-node features (x_j)
            accf = M::mma(xf.s[f], ld_frag(w.wbase, rowf, dm.WS, dm.KE + dm.Cp + f * M::KSTEP, h), accf);
            accs = M::mma(xf.s[f], ld_frag(w.wbase, rows, dm.WS, dm.KE + dm.Cp + f * M::KSTEP, h), accs);
        }
#endif
        // Pin the schedule of the chain: DEPTH fragment reads up front, then one read behind every MFMA, so
        // that the LDS latency of a weight fragment hides under the MFMAs issued before it.  (Left alone, hipcc keeps
        // one or two reads in flight and every MFMA waits out a full LDS round trip.)
        if constexpr (DEPTH > 0 && std::is_same<T, bf16_t>::value) {
#ifdef MDL_ABL_WSPLIT_BOUND
            constexpr int NMMA = 2 * (D::STATIC ? ((50 + 15) / 16 + NF) : 0);
#else
            constexpr int NMMA = 2 * (D::STATIC ? ((50 + 15) / 16 + 2 * NF) : 0);
#endif
            __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
#pragma unroll
            for (int q = 0; q < NMMA; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    } else {
        const T* xt = static_cast<const T*>(p.x) + (int64_t)my_tgt * dm.C;
        const T* xs = static_cast<const T*>(p.x) + (int64_t)my_src * dm.C;
        for (int k0 = 0; k0 < dm.Cp; k0 += M::KSTEP) {
            typename M::frag_t a = ld_xfrag<VEC>(xt, k0, h, dm.C);
            accf = M::mma(a, ld_frag(w.wbase, rowf, dm.WS, dm.KE + k0, h), accf);
            accs = M::mma(a, ld_frag(w.wbase, rows, dm.WS, dm.KE + k0, h), accs);
        }
        for (int k0 = 0; k0 < dm.Cp; k0 += M::KSTEP) {
            typename M::frag_t a = ld_xfrag<VEC>(xs, k0, h, dm.C);
            accf = M::mma(a, ld_frag(w.wbase, rowf, dm.WS, dm.KE + dm.Cp + k0, h), accf);
            accs = M::mma(a, ld_frag(w.wbase, rows, dm.WS, dm.KE + dm.Cp + k0, h), accs);
        }
    }
    }   // (!X3)
}

// acc[node slot][ch] += sum over the tile's edge slots of onehot(slot -> node slot) * v[edge slot][ch]
// v is in D layout (lane = channel, register r = edge slot d_row(r, h)); t4 = the lane's 4 dwords of
// target-slot bytes (byte r&3 of t4[r>>2] is the node slot of edge slot d_row(r,h), 0xff = invalid).
template <typename T>
__device__ __forceinline__ void seg_reduce_mma(const f32x16& v, const unsigned t4[4], int ns, f32x16& acc) {
    if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a;
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * ks + q;
                const unsigned slot = (t4[r >> 2] >> (8 * (r & 3))) & 0xffu;
                a[q] = (slot == (unsigned)ns) ? (short)0x3F80 : (short)0;
                vv[q] = v[r];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pack_bf16x8(vv), acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned slot = (t4[r >> 2] >> (8 * (r & 3))) & 0xffu;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((slot == (unsigned)ns) ? 1.0f : 0.0f, v[r], acc, 0, 0, 0);
        }
    }
}

// x3: the same reduction on the bf16 matrix core with the values split into (hi, lo) — the one-hot operand is exact in bf16, so
// the sums carry the values' 16 bits: 4 MFMAs of 32 cycles instead of 16 exact-fp32 MFMAs of 64
__device__ __forceinline__ void seg_reduce_mma_x3(const f32x16& v, const unsigned t4[4], int ns, f32x16& acc) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * ks + q;
            const unsigned slot = (t4[r >> 2] >> (8 * (r & 3))) & 0xffu;
            a[q] = (slot == (unsigned)ns) ? (short)0x3F80 : (short)0;
        }
        const SplitFrag b = split8(f32x4{v[8 * ks], v[8 * ks + 1], v[8 * ks + 2], v[8 * ks + 3]},
                                   f32x4{v[8 * ks + 4], v[8 * ks + 5], v[8 * ks + 6], v[8 * ks + 7]});
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b.hi, acc, 0, 0, 0);
    }
}

// One-hot operand tables in LDS (bf16 kernels).  Building a one-hot MFMA operand in registers costs
// ~3 VALU per element (extract, compare, select, pack); instead every edge-slot lane keeps ONE 1.0 in
// a small per-wave LDS table up to date (clear the old position, set the new one: 2 ds_write_b16) and
// the consumers read their operand fragments with ds_read_b128.
//   oh_t [32 node slots ][OHS]  column pos(edge slot): 1 where the edge's target is the row's node slot
//   oh_e [32 edge slots ][OHS]  column node slot     : 1 at the edge's target slot   (grad_out expansion)
//   oh_w [64 window slot][OHS]  column pos(edge slot): 1 where the edge's source is the row's window slot
// pos() is the K order in which the D-layout registers of a lane map onto MFMA K slots.
constexpr int OHS = 40;                                   // row stride in bf16 (80 B: odd number of 16-B slots)
__device__ __forceinline__ int oh_pos(int slot) {         // inverse of slot = d_row(8*ks + q, h), pos = 16*ks + 8*h + q
    const int hh = (slot >> 2) & 1, r = (slot & 3) + 4 * (slot >> 3);
    return 16 * (r >> 3) + 8 * hh + (r & 7);
}
__device__ __forceinline__ void oh_update(bf16_t* tab, int old_row, int new_row, int colpos) {
    if (old_row != new_row) {
        if (old_row >= 0) tab[old_row * OHS + colpos] = 0;
        if (new_row >= 0) tab[new_row * OHS + colpos] = 0x3F80;
    }
}
__device__ __forceinline__ bf16x8 oh_frag(const bf16_t* tab, int row, int ks, int h) {
    return *reinterpret_cast<const bf16x8*>(tab + row * OHS + 16 * ks + 8 * h);
}

// dpre fragments of one tile, packed once and used three times (B operand of the two segmented
// reductions, A operand of the dwe product).  bf16: 2 K-steps of 8 values; f32: the 16 registers.
template <typename T> struct DFrags;
template <> struct DFrags<bf16_t> {
    bf16x8 f[2], s[2];
    __device__ __forceinline__ void pack(const f32x16& af, const f32x16& as) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float a[8], b[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { a[q] = af[8 * ks + q]; b[q] = as[8 * ks + q]; }
            f[ks] = pack_bf16x8(a);
            s[ks] = pack_bf16x8(b);
        }
    }
};
template <> struct DFrags<float> {
    f32x16 f, s;
    __device__ __forceinline__ void pack(const f32x16& af, const f32x16& as) { f = af; s = as; }
};

// x3: dpre of one tile as (hi, lo) bf16 fragments, split ONCE and used by the three segmented reductions and the dwe product
struct DFragsX3 {
    SplitFrag f[2], s[2];
    __device__ __forceinline__ void pack(const f32x16& af, const f32x16& as) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f[ks] = split8(f32x4{af[8 * ks], af[8 * ks + 1], af[8 * ks + 2], af[8 * ks + 3]},
                           f32x4{af[8 * ks + 4], af[8 * ks + 5], af[8 * ks + 6], af[8 * ks + 7]});
            s[ks] = split8(f32x4{as[8 * ks], as[8 * ks + 1], as[8 * ks + 2], as[8 * ks + 3]},
                           f32x4{as[8 * ks + 4], as[8 * ks + 5], as[8 * ks + 6], as[8 * ks + 7]});
        }
    }
};
__device__ __forceinline__ void seg_reduce2_x3(const DFragsX3& d, const unsigned b4[4], unsigned row_id, f32x16& accF, f32x16& accS) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * ks + q;
            const unsigned slot = (b4[r >> 2] >> (8 * (r & 3))) & 0xffu;
            a[q] = (slot == row_id) ? (short)0x3F80 : (short)0;
        }
        accF = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.f[ks].lo, accF, 0, 0, 0);
        accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.s[ks].lo, accS, 0, 0, 0);
        accF = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.f[ks].hi, accF, 0, 0, 0);
        accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.s[ks].hi, accS, 0, 0, 0);
    }
}

// accF/accS[slot row][ch] += onehot(byte(edge slot) == row_id) x dpre   (see seg_reduce_mma)
template <typename T>
__device__ __forceinline__ void seg_reduce2(const DFrags<T>& d, const unsigned b4[4], unsigned row_id, f32x16& accF,
                                            f32x16& accS) {
    if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * ks + q;
                const unsigned slot = (b4[r >> 2] >> (8 * (r & 3))) & 0xffu;
                a[q] = (slot == row_id) ? (short)0x3F80 : (short)0;
            }
            accF = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.f[ks], accF, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.s[ks], accS, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned slot = (b4[r >> 2] >> (8 * (r & 3))) & 0xffu;
            const float a = (slot == row_id) ? 1.0f : 0.0f;
            accF = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d.f[r], accF, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d.s[r], accS, 0, 0, 0);
        }
    }
}

// bf16: one-hot A fragments come from an LDS table (oh_t / oh_w) instead of being built in registers
__device__ __forceinline__ void seg_reduce2_tab(const DFrags<bf16_t>& d, const bf16_t* tab, int row, int h,
                                                f32x16& accF, f32x16& accS) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 a = oh_frag(tab, row, ks, h);
        accF = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.f[ks], accF, 0, 0, 0);
        accS = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, d.s[ks], accS, 0, 0, 0);
    }
}

template <typename T, typename D>
__device__ __forceinline__ void setup_wave(const CgParams& p, const D& dm, char* smem, bool w_lds, WaveCtx<T>& w) {
    const int wave = threadIdx.x >> 6;
    const int w_bytes = w_lds ? ((p.w_elems * (int)sizeof(T) + 15) & ~15) : 0;
    char* base = smem + w_bytes + wave * p.wave_lds_bytes;
    w.et = reinterpret_cast<T*>(base);
    const int et_bytes = (32 * dm.EKS * (int)sizeof(T) + 15) & ~15;
    w.tsl = reinterpret_cast<unsigned*>(base + et_bytes);
    w.srcl = reinterpret_cast<int*>(base + et_bytes + 32);
    w.ssl = reinterpret_cast<unsigned*>(base + et_bytes + 32 + 128);
    w.touched = reinterpret_cast<unsigned long long*>(base + et_bytes + 32 + 128 + 32);
    w.oh_t = reinterpret_cast<bf16_t*>(base + et_bytes + 32 + 128 + 32 + 16);
    w.oh_e = w.oh_t + 32 * OHS;
    w.oh_w = w.oh_e + 32 * OHS;
    w.touched_b = reinterpret_cast<unsigned char*>(w.oh_w + 64 * OHS);
    w.dummy = reinterpret_cast<int*>(w.touched_b + 64);
    w.wbase = w_lds ? reinterpret_cast<const T*>(smem) : static_cast<const T*>(p.wpack);
    if (w_lds && p.w_slice) {
        // rows [32 s, +32) of the f part and of the s part -> LDS rows [0, 32) and [32, 64)   (WS * sizeof(T) % 16 == 0)
        const int sl = (int)blockIdx.x % p.NS;
        const int row16 = dm.WS * (int)sizeof(T) / 16;
        const f32x4* g = reinterpret_cast<const f32x4*>(p.wpack);
        f32x4* l = reinterpret_cast<f32x4*>(smem);
        for (int q = threadIdx.x; q < 64 * row16; q += blockDim.x) {
            const int r = q / row16, c = q - r * row16;
            const int gr = (r < 32 ? 32 * sl + r : dm.Cp + 32 * sl + (r - 32));
            l[q] = g[(int64_t)gr * row16 + c];
        }
    } else if (w_lds) {
        const f32x4* g = reinterpret_cast<const f32x4*>(p.wpack);
        f32x4* l = reinterpret_cast<f32x4*>(smem);
        const int n16 = w_bytes / 16;
        for (int q = threadIdx.x; q < n16; q += blockDim.x) l[q] = g[q];
    }
    // zero the e tile once: padded columns [G, KE) stay 0 forever, rows never hold garbage bits
    unsigned* z = reinterpret_cast<unsigned*>(base);
    for (int q = threadIdx.x & 63; q < p.wave_lds_bytes / 4; q += WAVE) z[q] = 0u;
    wave_lds_fence();
    // bias column: e-tile column G is a constant 1 whose weight row holds the bias
    if (p.bias_col && (threadIdx.x & 63) < 32) Elem<T>::st(w.et + (threadIdx.x & 63) * dm.EKS + dm.G, 1.0f);
    __syncthreads();
}

// per-lane indices of one edge tile (lane i and lane i+32 hold the same edge slot i)
struct TileIdx {
    int src, tgt, ep;
    template <bool WITH_EP = true, bool GUARD_EMPTY = true>
    __device__ __forceinline__ void load(const CgParams& p, int eb, int e1, int i, int n0) {
        // RAW loads on a clamped index, nothing else.  Slots past the end of the group (eb + i >= e1) get
        // the indices of the group's last edge: valid rows whose contribution every consumer masks by
        // slot validity (i < nv).  No select / arithmetic on the loaded values here: this is called
        // inside `if (more tiles)` blocks, and any use of a loaded value inside the block makes hipcc
        // wait for it (and for every older load, i.e. the x gathers issued just before) at that point.
        // Likewise never `cond ? load : x`: hipcc branches around the load and waits at the join.
        if (GUARD_EMPTY && p.E == 0) { src = tgt = n0; ep = 0; return; }   // uniform: graph without edges (callers inside
                                                                           // a tile loop know e1 > 0 and drop the branch)
        const int ec = max(min(eb + i, e1 - 1), 0);
        src = p.src[ec];
        tgt = p.tgt[ec];
        ep = 0;
        if (WITH_EP && p.eperm) ep = p.eperm[ec];
    }
};

// ------------------------------------------------------------------------------------------
// Forward
// ------------------------------------------------------------------------------------------
// seg_reduce_mma + in-degree count: cnt[node slot][*] += number of edge slots of the tile that map to
// the node slot (one-hot x all-ones), so the epilogue needs no rowptr loads.
template <typename T>
__device__ __forceinline__ void seg_reduce_cnt(const f32x16& v, const unsigned t4[4], int ns, f32x16& acc,
                                               f32x16& cnt) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a;
            float vv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 8 * ks + q;
                const unsigned slot = (t4[r >> 2] >> (8 * (r & 3))) & 0xffu;
                a[q] = (slot == (unsigned)ns) ? (short)0x3F80 : (short)0;
                vv[q] = v[r];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pack_bf16x8(vv), acc, 0, 0, 0);
            cnt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, cnt, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned slot = (t4[r >> 2] >> (8 * (r & 3))) & 0xffu;
            const float a = (slot == (unsigned)ns) ? 1.0f : 0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, v[r], acc, 0, 0, 0);
            cnt = __builtin_amdgcn_mfma_f32_32x32x2f32(a, 1.0f, cnt, 0, 0, 0);
        }
    }
}

// Work partition: wave `wi` of `W` owns the contiguous NODE range [na, nb) whose incoming edges are the
// wi-th of W equal shares of the edge array (boundaries rounded to node boundaries), found with a wave-wide
// 64-ary search on rowptr.  Ranges partition [0, N); a node's edges are never split, so no cross-wave
// combination is needed, every wave carries the same number of edge tiles (+-1) whatever N is, and small
// batches still spread over the whole chip.  Inside its range a wave walks groups of up to 32 nodes.
// The balanced quantity is w(n) = rowptr[n] + n (edges + nodes in front of node n): strictly increasing, so nodes WITHOUT
// edges are spread over the waves too — the padding nodes of a static batch (thousands of empty rows behind the last real
// node) would otherwise all fall to the last wave, which then walks them group by group while the chip idles.
__device__ __forceinline__ int wave_lower_bound(const int32_t* __restrict__ rowptr, int N, int64_t b, int lane) {
    int lo = 0, hi = N;                        // answer = first n in [lo, hi] with w(n) >= b  (w(N) = E + N >= b)
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;  // >= 1
        const int n = min(lo + lane * step, hi);
        const unsigned long long ge = __ballot((int64_t)rowptr[n] + n >= b);   // monotone in lane
        if (ge == 0ull) { lo = min(lo + 63 * step, hi) + 1; continue; }   // all probes below b
        const int fl = __builtin_ctzll(ge);
        if (fl == 0) { hi = lo; break; }
        hi = min(lo + fl * step, hi);
        lo = lo + (fl - 1) * step + 1;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}
// the same search on an explicit non-decreasing key array key[0..N]
__device__ __forceinline__ int wave_lower_bound_key(const int32_t* __restrict__ key, int N, int64_t b, int lane) {
    int lo = 0, hi = N;
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;
        const int n = min(lo + lane * step, hi);
        const unsigned long long ge = __ballot((int64_t)key[n] >= b);
        if (ge == 0ull) { lo = min(lo + 63 * step, hi) + 1; continue; }
        const int fl = __builtin_ctzll(ge);
        if (fl == 0) { hi = lo; break; }
        hi = min(lo + fl * step, hi);
        lo = lo + (fl - 1) * step + 1;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}
struct NodeRange {
    int na, nb;
    __device__ __forceinline__ NodeRange(int a, int b) : na(a), nb(b) {}
    __device__ __forceinline__ NodeRange(const CgParams& p, int wi, int W, int lane) {
        // the edges that exist = rowptr[N], which may be fewer than the p.E slots of the edge arrays (padded static
        // batches of the HIP-graph path): balancing on p.E would search for edge counts rowptr never reaches
        const int64_t Et = (int64_t)p.rowptr[p.N] + p.N;
        const int64_t b0 = Et * (int64_t)wi / W, b1 = Et * (int64_t)(wi + 1) / W;
        na = (wi == 0) ? 0 : wave_lower_bound(p.rowptr, (int)p.N, b0, lane);
        nb = (wi == W - 1) ? (int)p.N : wave_lower_bound(p.rowptr, (int)p.N, b1, lane);
    }
};
// scalars of one group of <= 32 nodes [n0, n1) inside a wave's node range
struct GroupInfo {
    int n0, n1, e0, e1;
    __device__ __forceinline__ void load(const CgParams& p, int n0_, int nb) {
        n0 = n0_;                                 // wave-uniform: the two loads become s_load
        n1 = min(n0 + 32, nb);
        e0 = p.rowptr[n0];
        e1 = p.rowptr[n1];
    }
};

#ifndef MDL_FWD_THREADS
#define MDL_FWD_THREADS 256     // workgroup size of the forward kernel (waves share one LDS copy of W)
#define MDL_FWD_WAVES 2         // waves per SIMD it is register-allocated for
#endif
// BN_: the instantiation whose epilogue also forms the BatchNorm statistics of the output (p.bn_sums) — a variant of its own, so
// that the plain forward keeps its register allocation (it sits at 252 of 256 VGPRs)
template <typename T, int CP_, int G_, int VEC, int EW, int WM, bool AB_ = false, int WSP = 0, bool BN_ = false, bool X3 = false>   // WM: 0 global, 1 LDS, 2 registers; X3: split-bf16 products on fp32 storage
// (fp32: the packed weights alone are 99 KB of LDS, so one 4-wave workgroup fits a CU whatever the register count — the fp32
// STATIC instantiation is allocated for ONE wave per SIMD (512 registers) instead of spilling 102 registers at 256: round 6)
#ifndef MDL_FWD_WAVES_F32
#define MDL_FWD_WAVES_F32 1
#endif
__global__ __launch_bounds__(MDL_FWD_THREADS, ((sizeof(T) == 4 && CP_ != 0) ? MDL_FWD_WAVES_F32 : MDL_FWD_WAVES)) void cgconv_fwd_kernel(CgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Mma<T> M;
    typedef Gate<M::FAST || X3> GT;   // x3: hardware exp2 / log2 / rcp (1 ulp) on base-2 pre-activations, like the bf16 kernels
    typedef Dims<T, CP_, G_, EW, WSP, X3> D;
    constexpr bool ST = D::STATIC;
    const D dm(p);
    WaveCtx<T> w;
    setup_wave<T>(p, dm, smem, WM == 1 || WM == 3, w);
    TDECL;

    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int total_waves = gridDim.x * (blockDim.x >> 6);
    // channel slice of this wave and its index among the waves of that slice (w_slice: a workgroup's waves share the slice)
    const bool wsl = (CP_ == 0 || CP_ > 64) && p.w_slice;
    const int s = wsl ? (int)blockIdx.x % p.NS : gw % p.NS;
    const int gstride = total_waves / p.NS;
    const int gidx = wsl ? ((int)blockIdx.x / p.NS) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6) : gw / p.NS;
    // with a bias column the bias rides in the GEMM (K column G); otherwise it seeds the accumulators
    constexpr bool BC = G_ != 0 && (G_ % 16) != 0;   // bias column known at compile time
    const float bf = (BC || p.bias_col) ? 0.0f : p.bpack[s * 32 + i];
    const float bs = (BC || p.bias_col) ? 0.0f : p.bpack[dm.Cp + s * 32 + i];
    const T* x = static_cast<const T*>(p.x);
    T* out = static_cast<T*>(p.out);
    constexpr int NKW = (WM == 2) ? (((G_ + 15) / 16 * 16) + 2 * CP_) / M::KSTEP : (WM == 3 ? 2 * CP_ / M::KSTEP : 1);
    WRegs<T, NKW> wr;
    if constexpr (WM == 2) wr.load(static_cast<const T*>(p.wpack), s * 32 + i, dm.Cp + s * 32 + i, dm.WS, h, 0);
    if constexpr (WM == 3) wr.load(static_cast<const T*>(p.wpack), s * 32 + i, dm.Cp + s * 32 + i, dm.WS, h, dm.KE);
    const int ch = s * 32 + i;

#if MDL_FWD_ALLSLICES
    if constexpr (ST && WM == 1 && CP_ <= 64) {     // (wider static shapes: one slice per wave, the workgroup keeps that slice of W in LDS)
        // One wave handles ALL channel slices of its group.  Staging the tile (edge-feature stream, index
        // loads, x gathers, one-hot table) is slice independent: doing it once per tile instead of once per
        // (tile, slice) removes the duplicated per-tile overhead; the pre-GEMM / gate / aggregation then
        // run per 32-channel slice on the same staged operands.
        constexpr int NSL = CP_ / 32;
        const int nw_total = gridDim.x * (blockDim.x >> 6);
        // Two workgroups share a CU and its SIMDs arbitrate by age: the waves of the workgroup dispatched first (block index
        // below half the grid) run 14 % faster than their younger co-residents (measured wave lifetimes 161 vs 184 us), which
        // then finish alone at half occupancy.  The older half takes MDL_FWD_AGE_SKEW per mille more of the work.
        NodeRange R(0, 0);
        {
            const int half = nw_total >> 1, gwu = __builtin_amdgcn_readfirstlane(gw);
            if (MDL_FWD_AGE_SKEW == 0 || (nw_total & 1) || gridDim.x < 512) {
                R = NodeRange(p, gwu, nw_total, lane);
            } else {
                const int64_t Et = (int64_t)p.rowptr[p.N] + p.N;
                const int64_t wa = 1000 + MDL_FWD_AGE_SKEW, wb = 1000 - MDL_FWD_AGE_SKEW;        // weights of the two halves
                auto cut = [&](int k) -> int64_t {                                               // work in front of wave k
                    const int64_t units = k <= half ? wa * k : wa * half + wb * (k - half);
                    return Et * units / (1000 * (int64_t)nw_total);
                };
                R.na = gwu == 0 ? 0 : wave_lower_bound(p.rowptr, (int)p.N, cut(gwu), lane);
                R.nb = gwu == nw_total - 1 ? (int)p.N : wave_lower_bound(p.rowptr, (int)p.N, cut(gwu + 1), lane);
            }
        }
        // BatchNorm statistics of the output (p.bn_sums).  The forward has no register to spare across its tile loop (252 VGPRs
        // at two waves per SIMD), so the running sums live in the wave's LDS region ([2][CP_] floats behind the staging
        // buffers, zeroed by setup_wave): a group's epilogue adds its rows' contributions, the wave's exit turns them into two
        // atomics per channel on one of the MDL_BN_REPLICAS copies of the sums.
        constexpr bool bnst = BN_;
        // (address from scalars at every use: a pointer kept across the tile loop is two more VGPRs the kernel does not have)
        auto bnl_at = [&](int k) -> float* {
            const int wvs = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const int w_bytes = (p.w_elems * (int)sizeof(T) + 15) & ~15;
            return reinterpret_cast<float*>(smem + w_bytes + (wvs + 1) * p.wave_lds_bytes - 2 * CP_ * 4) + k;
        };
        int bn_rows = (int)p.N;
        if (bnst) {
            if (p.bn_nrows) bn_rows = (int)max((int64_t)1, min(*p.bn_nrows, p.N));
            if (gw == 0 && h == 0) {            // the shift row the sums are about, for the apply kernel (behind the totals rows)
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl)
                    p.bn_sums[(size_t)(2 * MDL_BN_REPLICAS + 2) * CP_ + sl * 32 + i] = p.bn_shift ? p.bn_shift[sl * 32 + i] : 0.0f;
            }
        }
        auto bn_add = [&](int sl, float t0, float t1) {      // this lane's sums over its rows -> the wave's LDS totals (h == 0 lanes)
            t0 += __shfl_xor(t0, 32);
            t1 += __shfl_xor(t1, 32);
            if (h == 0) {
                *bnl_at(sl * 32 + i) += t0;
                *bnl_at(CP_ + sl * 32 + i) += t1;
            }
        };
        auto bn_flush = [&]() {
            if (!bnst) return;
            wave_lds_fence();
            float* dst = p.bn_sums + (size_t)(gw % MDL_BN_REPLICAS) * 2 * CP_;
            if (h == 0) {
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    unsafeAtomicAdd(dst + sl * 32 + i, *bnl_at(sl * 32 + i));
                    unsafeAtomicAdd(dst + CP_ + sl * 32 + i, *bnl_at(CP_ + sl * 32 + i));
                }
            }
        };
        if (R.na >= R.nb) return;
        // The wave walks its groups as one continuous stream: the next group's row pointers are requested
        // at the top of the current group, and the indices / edge-feature words of the next group's first
        // tile during the current group's last tile, so a group boundary costs no dependent round trips.
        GroupInfo G, GN;
        G.load(p, R.na, R.nb);
        GN = G;
        TileIdx cur, nxt;
        EWords<T, G_, EW> ew;
        XFrags<T, WSP ? 0 : CP_, VEC, X3> xf;
        // W-split: projection rows instead of x rows, ONE slice's fragments at a time (32 registers, like the x rows): the
        // rows of slice sl + 1 — or of the next tile's slice 0 — are requested right after slice sl's MFMAs have consumed
        // the registers, and arrive under that slice's gate arithmetic and aggregation
        PFrags<CP_, 1> pf;
        bf16x8 idf[2];
        if constexpr (WSP != 0) identity_frags(i, h, idf);
        auto load_rows = [&](int tg, int sr) {
            if constexpr (WSP != 0) pf.load(p.pt, p.ps, tg, sr, h, 0); else xf.load(x, dm.C, tg, sr, h);
        };
        bool primed = false;                                  // cur / ew / xf already hold the group's first tile
        while (true) {
            const bool hasN = G.n1 < R.nb;
            if (hasN) GN.load(p, G.n1, R.nb);
            if (G.e0 == G.e1) {
                // Nodes without edges: out = x.  The whole RUN of edge-less nodes of this wave's range is copied at once
                // with 16-byte vectors (8 rows per trip) — the padding rows of a static batch come as thousands of such
                // nodes in a row, and walking them as 32-node groups (a scalar rowptr round trip + 32 two-byte copies
                // each) made the waves that own them 3x slower than the rest of the launch.
                int ne = G.n1;
                {
                    int lo = G.n1, hi = R.nb;                      // first node in [n1, nb] whose rowptr exceeds e0
                    while (lo < hi) {
                        const int step = (hi - lo + 63) >> 6;
                        const int n = min(lo + lane * step, hi);
                        const unsigned long long gt = __ballot(n < R.nb ? p.rowptr[n + 1] > G.e0 : true);
                        if (gt == 0ull) { lo = min(lo + 63 * step, hi) + 1; continue; }
                        const int fl = __builtin_ctzll(gt);
                        if (fl == 0) { hi = lo; break; }
                        hi = min(lo + fl * step, hi);
                        lo = lo + (fl - 1) * step + 1;
                    }
                    ne = min(__builtin_amdgcn_readfirstlane(lo), R.nb);
                }
                {
                    constexpr int VW = 16 / (int)sizeof(T);        // elements per 16-byte vector
                    constexpr int VPR = CP_ / VW;                    // vectors per row (CP_ == C for the static shapes)
                    typedef __attribute__((ext_vector_type(4))) unsigned u32x4c;
                    const int64_t v0 = (int64_t)G.n0 * VPR, v1 = (int64_t)ne * VPR;
                    const u32x4c* __restrict__ xs = reinterpret_cast<const u32x4c*>(x);
                    u32x4c* __restrict__ os = reinterpret_cast<u32x4c*>(out);
                    for (int64_t q = v0 + lane; q < v1; q += 4 * WAVE) {
                        u32x4c v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) v[u] = xs[min(q + u * WAVE, v1 - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (q + u * WAVE < v1) os[q + u * WAVE] = v[u];
                    }
                }
                if (bnst && G.n0 < bn_rows) {                      // (rare: isolated nodes below the row count; the padding is excluded)
#pragma unroll
                    for (int sl = 0; sl < NSL; ++sl) {
                        const float sh = p.bn_shift ? p.bn_shift[sl * 32 + i] : 0.0f;
                        float t0 = 0.0f, t1 = 0.0f;
                        for (int n = G.n0 + h; n < min(ne, bn_rows); n += 2) {
                            const float v = (float)Elem<T>::ld(x + (int64_t)n * dm.C + sl * 32 + i) - sh;
                            t0 += v;
                            t1 = fmaf(v, v, t1);
                        }
                        bn_add(sl, t0, t1);
                    }
                }
                if (ne >= R.nb) break;
                G.load(p, ne, R.nb);
                primed = false;
                continue;
            }
            f32x16 acc_out[NSL];
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_out[sl][r] = 0.0f;
            if (!primed) {
                cur.template load<false>(p, G.e0, G.e1, i, G.n0);
                ew.prefetch(p, lane, G.e0, min(32, G.e1 - G.e0), cur.ep);
                load_rows(cur.tgt, cur.src);
            }
            nxt = cur;
            bool nextHasEdges = false;                        // evaluated at the last tile (GN arrives by then)
            // saved-gate forward: the packed factors of a tile's LAST slice wait in registers and are stored during the
            // next tile (see below); abq_eb < 0 = nothing pending
            constexpr int ROWB = 4 * CP_;
            unsigned abq[16];
            int abq_eb = -1, abq_nv = 0;
            auto ab_store = [&](const unsigned (&v)[16], int eb_, int nv_, int sl_) {
                // rows past the group's end are dropped by the range check of the store resource (they belong to the
                // next group's owner); row offsets = two per-lane bases + an immediate < 4096, all inside the range check
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    static_cast<char*>(p.ab) + (int64_t)eb_ * ROWB, 0, nv_ * ROWB, 0x00020000);
                const int vo0 = 4 * h * ROWB + sl_ * 128 + i * 4, vo1 = vo0 + 16 * ROWB;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(v[r], rs, (r < 8 ? vo0 : vo1) + ((r & 3) + 8 * ((r >> 2) & 1)) * ROWB, 0, 0);
            };
            TRESET();
            for (int eb = G.e0; eb < G.e1; eb += 32) {
                const int nv = min(32, G.e1 - eb);
                const bool last = eb + 32 >= G.e1;
                TMARK(0);
                wave_lds_fence();
                ew.commit(w.et, dm.EKS, lane);
                if (h == 0) reinterpret_cast<unsigned char*>(w.tsl)[i] = (i < nv) ? (unsigned char)(cur.tgt - G.n0) : 0xff;
                wave_lds_fence();
                TMARK(1);
                // The next tile's loads are issued on EVERY path, from uniform selects of the tile base (next tile of
                // this group / first tile of the next group / this tile again when the stream ends): a path that skips
                // them joins the loop with "the x fragments are the newest loads in flight", and the waits hipcc then
                // places in front of the x MFMAs also drain the prefetches issued a few instructions earlier.
                {
                    const bool nh = hasN && GN.e0 < GN.e1;
                    nextHasEdges = last && nh;
                    const int pe = last ? (nh ? GN.e0 : eb) : eb + 32;
                    const int pe1 = nextHasEdges ? GN.e1 : G.e1;
                    const int pn0 = nextHasEdges ? GN.n0 : G.n0;
                    nxt.template load<false, false>(p, pe, pe1, i, pn0);
                    ew.prefetch(p, lane, pe, 32, 0);
                }
                TMARK(2);
                unsigned t4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t4[j] = w.tsl[2 * j + h];
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    f32x16 accf, accs;
                    const float b0 = (BC || p.bias_col) ? 0.0f : p.bpack[sl * 32 + i];
                    const float b1 = (BC || p.bias_col) ? 0.0f : p.bpack[dm.Cp + sl * 32 + i];
#pragma unroll
                    for (int r = 0; r < 16; ++r) { accf[r] = b0; accs[r] = b1; }
                    if constexpr (WSP != 0) {
                        pre_tile_wsp<CP_, 1, MDL_FWD_PRE_DEPTH>(dm, w, lane, sl, 0, pf, idf, accf, accs);
                        pf.load(p.pt, p.ps, sl == NSL - 1 ? nxt.tgt : cur.tgt, sl == NSL - 1 ? nxt.src : cur.src, h, (sl + 1) % NSL);
                    } else {
                        pre_tile<T, CP_, VEC, WM, NKW, MDL_FWD_PRE_DEPTH, X3>(p, dm, w, lane, sl, cur.tgt, cur.src, xf, wr, accf, accs);
                    }
                    TPIN16(accf); TPIN16(accs);
                    TMARK(3 + 3 * sl);
                    if constexpr (AB_) {
                        // Stores poison the wait counters: loads and stores complete out of order with respect to each
                        // other, so the next wait for ANY load (top of the next tile) also waits for every store in
                        // flight.  A slice's factors stored right after its gate leave the last slice ~10 % of a tile
                        // to complete before that wait (measured: +1500 cycles per tile in the first MFMA chain).  So the
                        // last slice's dwords stay in registers and go out HERE, behind the first slice's MFMA chain of
                        // the NEXT tile: more than half a tile ahead of the next wait.
                        if (sl == 0 && abq_eb >= 0) { ab_store(abq, abq_eb, abq_nv, NSL - 1); abq_eb = -1; }
                    }
#if MDL_FWD_XEARLY
                    // the x rows of the NEXT tile: requested as soon as the last slice's MFMAs have consumed this
                    // tile's fragments (same registers), so their latency hides under the gate / aggregation
                    if constexpr (WSP == 0) { if (sl == NSL - 1) load_rows(nxt.tgt, nxt.src); }   // unconditional, like the loads above
#endif
                    f32x16 m;
                    if constexpr (AB_) {
                        // m and, for the backward, A = dm/dpre_f = sigmoid'(f) softplus(s), B = dm/dpre_s = sigmoid(f) sigmoid(s)
                        // (same four transcendentals as the plain gate: the two reciprocals share one v_rcp), packed
                        // as one dword per (edge, channel): a row of a slice is 128 contiguous bytes
                        unsigned abv[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float sf, sp_u, ss;
                            GT::deriv(accf[r], accs[r], sf, sp_u, ss);
                            const float mm = sf * sp_u;
                            m[r] = mm;
                            abv[r] = pk_bf16((mm * GT::M_SCALE) * (1.0f - sf), sf * ss);
                        }
                        if (sl == NSL - 1) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) abq[r] = abv[r];
                            abq_eb = eb;
                            abq_nv = nv;
                        } else {
                            ab_store(abv, eb, nv, sl);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) m[r] = GT::sigmoid(accf[r]) * GT::softplus_u(accs[r]);
                    }
                    TPIN16(m);
                    TMARK(4 + 3 * sl);
                    if constexpr (X3) seg_reduce_mma_x3(m, t4, i, acc_out[sl]); else seg_reduce_mma<T>(m, t4, i, acc_out[sl]);
                    TPIN16(acc_out[sl]);
                    TMARK(5 + 3 * sl);
                    __builtin_amdgcn_sched_barrier(0);      // keep the slices' register footprints apart
                }
#if !MDL_FWD_XEARLY
                if (!last || nextHasEdges) load_rows(nxt.tgt, nxt.src);
#endif
                cur = nxt;
                TTILE();
            }
            if constexpr (AB_) {
                if (abq_eb >= 0) ab_store(abq, abq_eb, abq_nv, NSL - 1);        // the group's last tile
            }
            // epilogue: out = x + acc / deg.  All residual rows are requested first (clamped row index, no
            // guards), so the group pays one memory round trip instead of one per row.
            {
                // in-degrees straight from rowptr (lane i keeps the degree of node slot i; no count accumulator and
                // no count MFMAs in the tile loop), fetched together with the residual rows
                const int ndg = min(G.n0 + i, G.n1 - 1);
                const int dg0 = p.rowptr[ndg], dg1 = p.rowptr[ndg + 1];
                float xr[NSL][16];
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = min(G.n0 + d_row(r, h), G.n1 - 1);
                        xr[sl][r] = Elem<T>::ld(x + (int64_t)n * dm.C + sl * 32 + i);
                    }
                const float invd = (M::FAST && !X3) ? __builtin_amdgcn_rcpf((float)max(dg1 - dg0, 1)) : 1.0f / (float)max(dg1 - dg0, 1);
                float invr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) invr[r] = __shfl(invd, d_row(r, h));
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) {
                    const float sh = (bnst && p.bn_shift) ? p.bn_shift[sl * 32 + i] : 0.0f;
                    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = G.n0 + d_row(r, h);
                        float a = acc_out[sl][r] * GT::M_SCALE;
                        if (p.aggr == MDL_MEAN) a *= invr[r];
                        const float v = xr[sl][r] + a;
                        if (n < G.n1) Elem<T>::st(out + (int64_t)n * dm.C + sl * 32 + i, v);
                        if (bnst) {
                            // the statistics of what the BatchNorm will READ: the rounded value, rows that exist only
                            const float vr = (n < G.n1 && n < bn_rows) ? Elem<T>::rnd(v) - sh : 0.0f;
                            t0 += vr;
                            t1 = fmaf(vr, vr, t1);
                        }
                    }
                    if (bnst) bn_add(sl, t0, t1);
                }
            }
            TMARK(11);
            if (!hasN) break;
            G = GN;
            primed = nextHasEdges;
        }
        bn_flush();
        TFLUSH(0);
        return;
    }
#endif
    // The wave walks its groups as ONE continuous stream of edge tiles: while tile t computes, the
    // indices and edge-feature words of tile t+1 are in flight — across group boundaries too — so
    // the load pipeline never drains.  A group's epilogue (residual add, mean, store) needs no
    // dependent loads: the in-degree comes out of the one-hot MFMA (cnt) and the residual rows are
    // requested at the top of the group's last tile.
    const NodeRange R(p, __builtin_amdgcn_readfirstlane(gidx), gstride, lane);
    if (R.na >= R.nb) return;
    GroupInfo G, GN;
    G.load(p, R.na, R.nb);
    bool hasN = G.n1 < R.nb;
    if (hasN) GN.load(p, G.n1, R.nb);

    f32x16 acc_out, cnt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_out[r] = 0.0f; cnt[r] = 0.0f; }
    TileIdx cur, nxt;
    EWords<T, G_, EW> ew;
    XFrags<T, CP_, VEC, X3> xf;
    float xr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) xr[r] = 0.0f;

    int eb = G.e0;
    bool primed = false;          // cur / ew already hold the tile at eb (prefetched by the previous tile)
            TRESET();
    while (true) {
        if (G.e0 == G.e1) {
            // group without edges: out = x for its nodes (mean over nothing = 0)
            if (ch < dm.C) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = G.n0 + d_row(r, h);
                    if (n < G.n1) out[(int64_t)n * dm.C + ch] = x[(int64_t)n * dm.C + ch];
                }
            }
            if (!hasN) break;
            G = GN;
            hasN = G.n1 < R.nb;
            if (hasN) GN.load(p, G.n1, R.nb);
            eb = G.e0;
            primed = false;
            continue;
        }
        if (!primed) {
            cur.template load<!ST>(p, eb, G.e1, i, G.n0);
            if constexpr (ST) ew.prefetch(p, lane, eb, min(32, G.e1 - eb), cur.ep);
        }
        const int nv = min(32, G.e1 - eb);
        const bool valid_i = i < nv;
        const bool last = eb + 32 >= G.e1;
        TMARK(0);
        wave_lds_fence();
        if constexpr (ST) ew.commit(w.et, dm.EKS, lane); else stage_e_tile<T, EW>(p, dm, w, lane, eb, nv, cur.ep);
        if (h == 0) reinterpret_cast<unsigned char*>(w.tsl)[i] = valid_i ? (unsigned char)(cur.tgt - G.n0) : 0xff;
        wave_lds_fence();
        TMARK(1);

#ifndef MDL_ABL_NOX
        if constexpr (CP_ != 0) xf.load(x, dm.C, cur.tgt, cur.src, h);
#endif
        // prefetch the next tile of the stream (same group, or the first tile of the next group)
        primed = false;
        if (!last) {
            nxt.template load<!ST>(p, eb + 32, G.e1, i, G.n0);
#ifndef MDL_ABL_NOE
            if constexpr (ST) ew.prefetch(p, lane, eb + 32, min(32, G.e1 - eb - 32), nxt.ep);
#endif
            primed = true;
        } else {
            if (hasN && GN.e0 < GN.e1) {
                nxt.template load<!ST>(p, GN.e0, GN.e1, i, GN.n0);
#ifndef MDL_ABL_NOE
                if constexpr (ST) ew.prefetch(p, lane, GN.e0, min(32, GN.e1 - GN.e0), nxt.ep);
#endif
                primed = true;
            }
            // residual rows of this group, needed by the epilogue right after this tile
            if (ch < dm.C) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = min(G.n0 + d_row(r, h), G.n1 - 1);
                    xr[r] = Elem<T>::ld(x + (int64_t)n * dm.C + ch);
                }
            }
        }

        TMARK(2);
        f32x16 accf, accs;
#pragma unroll
        for (int r = 0; r < 16; ++r) { accf[r] = bf; accs[r] = bs; }
#ifndef MDL_ABL_NOPRE
        pre_tile<T, CP_, VEC, WM, NKW, 0, X3>(p, dm, w, lane, s, cur.tgt, cur.src, xf, wr, accf, accs);
#else
        if constexpr (std::is_same<T, bf16_t>::value) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { accf[r] += bf2f((bf16_t)xf.t[0][r & 7]); accs[r] += bf2f((bf16_t)xf.s[0][r & 7]) + bf2f(w.et[i * dm.EKS + r]); }
        }
#endif

        TMARK(3);
        unsigned t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] = w.tsl[2 * j + h];
        // gate.  Edge slots >= nv hold finite garbage; their one-hot column is all zero, so they
        // contribute exact zeros to the aggregation — no per-element masking needed.
        f32x16 m;
#pragma unroll
#ifndef MDL_ABL_NOGATE
        for (int r = 0; r < 16; ++r) m[r] = GT::sigmoid(accf[r]) * GT::softplus_u(accs[r]);
#else
        for (int r = 0; r < 16; ++r) m[r] = accf[r] * accs[r];
#endif
        TMARK(4);
        seg_reduce_cnt<T>(m, t4, i, acc_out, cnt);
        TMARK(5);

        if (last) {
            // epilogue: out = x + acc / deg   (rows = node slots in D layout, col = channel)
            if (ch < dm.C) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = G.n0 + d_row(r, h);
                    if (n < G.n1) {
                        float a = acc_out[r] * GT::M_SCALE;
                        if (p.aggr == MDL_MEAN) {
                            const float deg = fmaxf(cnt[r], 1.0f);
                            a = M::FAST ? a * __builtin_amdgcn_rcpf(deg) : a / deg;
                        }
                        Elem<T>::st(out + (int64_t)n * dm.C + ch, xr[r] + a);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc_out[r] = 0.0f; cnt[r] = 0.0f; }
            if (!hasN) break;
            G = GN;
            hasN = G.n1 < R.nb;
            if (hasN) GN.load(p, G.n1, R.nb);
            eb = G.e0;
        } else {
            eb += 32;
        }
        cur = nxt;
        TMARK(6);
            TTILE();
    }
}

// By-source sums in bf16 (p.rs16): two fp32 atomics per (row, lane) become ONE packed bf16 atomic per TWO rows.  A lane holds
// column c of rows R (value va) and R + 1 (vb); neighbouring lanes hold neighbouring columns.  Even lanes take the pair
// (c, c + 1) of row R, odd lanes the pair (c - 1, c) of row R + 1: each lane hands its partner (lane ^ 1) the value the partner
// needs with one DPP quad permute, packs with one v_cvt_pk_bf16_f32 and issues global_atomic_pk_add_bf16 on a 4-byte
// aligned column pair — half the atomic operations AND half the bytes of the fp32 form (atomic throughput at the L2 is what the
// window flush costs: ablated, 45 of 585 us).  ca / cb: whether row R / R + 1 is to be added at all.
__device__ __forceinline__ void rsrc16_add2(bf16_t* base_even_col, int64_t rowa_elems, int64_t rowb_elems, float va, float vb,
                                            bool ca, bool cb, int odd) {
    typedef __bf16 __attribute__((ext_vector_type(2))) bf2v;
    typedef __attribute__((address_space(1))) bf2v* gptr_t;
    const float send = odd ? va : vb;
    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));
    const float lo = odd ? recv : va, hi = odd ? vb : recv;
    if (odd ? cb : ca) {
        const unsigned pk = pk_bf16(lo, hi);
        __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr_t)(base_even_col + (odd ? rowb_elems : rowa_elems)), __builtin_bit_cast(bf2v, pk));
    }
}

// ------------------------------------------------------------------------------------------
// Backward edge pass
// ------------------------------------------------------------------------------------------
template <typename T, int CP_, int G_, int VEC, int EW, int WM, int WSP = 0, bool X3 = false>
__global__ __launch_bounds__(256, MDL_BWD_WAVES) void cgconv_bwd_kernel(CgParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef Mma<T> M;
    typedef Gate<M::FAST || X3> GT;   // x3: hardware exp2 / log2 / rcp (1 ulp) on base-2 pre-activations, like the bf16 kernels
    typedef Dims<T, CP_, G_, EW, WSP, X3> D;
    constexpr bool ST = D::STATIC;
    constexpr bool BF = std::is_same<T, bf16_t>::value;
    const D dm(p);
    WaveCtx<T> w;
    setup_wave<T>(p, dm, smem, WM == 1 || WM == 3, w);
    TDECL;

    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int total_waves = gridDim.x * (blockDim.x >> 6);
    // channel slice of this wave and its index among the waves of that slice (w_slice: a workgroup's waves share the slice)
    const bool wsl = (CP_ == 0 || CP_ > 64) && p.w_slice;
    const int s = wsl ? (int)blockIdx.x % p.NS : gw % p.NS;
    const int gstride = total_waves / p.NS;
    const int gidx = wsl ? ((int)blockIdx.x / p.NS) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6) : gw / p.NS;
    const int ch = s * 32 + i;
    constexpr bool BC = G_ != 0 && (G_ % 16) != 0;
    const float bf = (BC || p.bias_col) ? 0.0f : p.bpack[ch], bs = (BC || p.bias_col) ? 0.0f : p.bpack[dm.Cp + ch];
    const T* x = static_cast<const T*>(p.x);
    const T* go = static_cast<const T*>(p.gout);
    const int C2 = 2 * dm.Cp;
    constexpr int NKW = (WM == 2) ? (((G_ + 15) / 16 * 16) + 2 * CP_) / M::KSTEP : (WM == 3 ? 2 * CP_ / M::KSTEP : 1);
    WRegs<T, NKW> wr;
    if constexpr (WM == 2) wr.load(static_cast<const T*>(p.wpack), s * 32 + i, dm.Cp + s * 32 + i, dm.WS, h, 0);
    if constexpr (WM == 3) wr.load(static_cast<const T*>(p.wpack), s * 32 + i, dm.Cp + s * 32 + i, dm.WS, h, dm.KE);

    // dwe accumulators: [part f|s][n-tile of G]  (rows = channel slot, cols = edge feature)
    constexpr int GNT = G_ ? (G_ + 31) / 32 : 2;
    f32x16 dwe_acc[2][GNT];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < GNT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dwe_acc[a][b][r] = 0.0f;

    float dbf_acc = 0.0f, dbs_acc = 0.0f;
    int oh_a = -1;                   // where this lane currently has its 1.0 in the one-hot tables it keeps (bf16 path):
                                     // lanes h = 0 keep the TARGET tables (oh_t, oh_e), lanes h = 1 the WINDOW table (oh_w)
    // Work distribution.  Static: one edge-balanced node range per wave (NodeRange).  Dynamic (p.ctr, large
    // problems): the waves of a slice take 32-node groups from a shared counter — equal tile counts do not mean equal
    // time (graphs wider than the source window fall back to atomics, CUs differ in memory latency), and the
    // kernel ends with its slowest wave.  The next group id is requested at the top of the current group.
    const bool dyn = p.ctr != nullptr;
    NodeRange R{0, 0};
    if (!dyn) R = NodeRange(p, __builtin_amdgcn_readfirstlane(gidx), gstride, lane);
    const int nend = dyn ? (int)p.N : R.nb;
    int gpend = 0;                                   // lane 0: group id returned by the counter (in flight)
    if (dyn && lane == 0) gpend = (int)atomicAdd(p.ctr + s, 1u);
    // The last groups the counter hands out are HALF groups (16 nodes): a 32-node group is ~13 tiles = 45 us of a wave's
    // ~550, and the kernel ends with the wave that drew the last one (measured spread of the waves' end times: 10 %).
    auto group_start = [&](int g) { return g < p.g_full ? 32 * g : 32 * p.g_full + 16 * (g - p.g_full); };
    int n0 = dyn ? group_start(__builtin_amdgcn_readfirstlane(gpend)) : R.na;
    while (n0 < nend) {
        const int n1 = min(n0 + ((dyn && n0 >= 32 * p.g_full) ? 16 : 32), nend);
        if (dyn && lane == 0) gpend = (int)atomicAdd(p.ctr + s, 1u);   // next group: issued now, read at the end of this one
        const int e0 = p.rowptr[n0];          // n0 is wave-uniform: scalar loads
        const int e1 = p.rowptr[n1];

        // ---- group prologue.  Every load below is unconditional on a clamped index and all of them are
        // issued before the first use, so the group pays ~one memory round trip (a guarded load costs one each).
        // (1) in-degrees of the group's nodes: lane i keeps 1/deg of node slot i
        const int nd = min(n0 + i, n1 - 1);
        const int dg0 = p.rowptr[nd], dg1 = p.rowptr[nd + 1];
        // (2) grad_out columns of this lane's channel for the B fragments of the one-hot expansion to edges
        constexpr bool PK = BF || X3;                // k-slots of a fragment: eight node slots per lane half (bf16 MFMA shapes)
        constexpr int NFg = PK ? 2 : 16, PERg = PK ? 8 : 1;
        float graw[NFg][PERg];
        {
            const int chc = min(ch, dm.C - 1);
#pragma unroll
            for (int f = 0; f < NFg; ++f)
#pragma unroll
                for (int q = 0; q < PERg; ++q) {
                    const int ns = PK ? (16 * f + 8 * h + q) : (2 * f + h);
                    graw[f][q] = Elem<T>::ld(go + (int64_t)min(n0 + ns, n1 - 1) * dm.C + chc);
                }
        }
        // (3) source window base.  The sources of a group's edges are its in-graph neighbours, i.e. a short
        // node range: sums by SOURCE for the 64 nodes [wb, wb+64) are kept in registers (one-hot MFMA, like the
        // target reduction) and flushed once per group; only sources outside the window (very large graphs)
        // fall back to per-edge atomics.  wb = smallest source of the group.
        int wb = 0x7fffffff;
        for (int eb = e0; eb < e1; eb += 8 * WAVE) {
            int sv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sv[u] = p.src[min(eb + u * WAVE + lane, e1 - 1)];     // clamp, never guard
#pragma unroll
            for (int u = 0; u < 8; ++u) wb = min(wb, sv[u]);                                    // (clamped slots repeat a real edge)
        }
        // (4) first tiles of the edge stream
        TileIdx cur, nxt, nn;
        EWords<T, G_, EW> ew;
        XFrags<T, WSP ? 0 : CP_, VEC, X3> xf, xn;
        PFrags<CP_, 1> pf, pn;                      // W-split: the projection rows of this wave's slice instead of x rows
        bf16x8 idf[2];
        if constexpr (WSP != 0) identity_frags(i, h, idf);
        cur.template load<!ST>(p, e0, e1, i, n0);
        nxt = cur;
        if (e0 + 32 < e1) nxt.template load<!ST>(p, e0 + 32, e1, i, n0);
        nn = nxt;
        constexpr bool XDB = MDL_BWD_XDB != 0 && !(X3 && CP_ > 64);   // (x3 at 128 channels: two sets of fp32 x chunks are 256 registers)
        if constexpr (ST) ew.prefetch(p, lane, e0, min(32, e1 - e0), cur.ep);
        if constexpr (WSP != 0) pf.load(p.pt, p.ps, cur.tgt, cur.src, h, s);
        else if constexpr (CP_ != 0 && XDB) xf.load(x, dm.C, cur.tgt, cur.src, h);

        const float invd = (p.aggr == MDL_MEAN) ? 1.0f / (float)max(dg1 - dg0, 1) : 1.0f;
        typename M::frag_t gB[NFg];
        SplitFrag gBx[2];                            // x3: grad_out / deg as (hi, lo) fragments
#pragma unroll
        for (int f = 0; f < NFg; ++f) {
            float v[PERg];
#pragma unroll
            for (int q = 0; q < PERg; ++q) {
                const int ns = PK ? (16 * f + 8 * h + q) : (2 * f + h);
                const float sc = __shfl(invd, ns);
                v[q] = (n0 + ns < n1 && ch < dm.C) ? graw[f][q] * sc : 0.0f;
            }
            if constexpr (BF) gB[f] = pack_bf16x8(v);
            else if constexpr (X3) gBx[f] = split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
            else gB[f] = v[0];
        }

        TMARK(12);           // (timing builds: prologue up to here = loads + their wait + the gB shuffles)
        f32x16 Rf, Rs;
#pragma unroll
        for (int r = 0; r < 16; ++r) { Rf[r] = 0.0f; Rs[r] = 0.0f; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wb = min(wb, __shfl_xor(wb, o));
        wb = __builtin_amdgcn_readfirstlane(wb);
        f32x16 Wf[2], Ws[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { Wf[mt][r] = 0.0f; Ws[mt][r] = 0.0f; }
        if constexpr (BF) w.touched_b[lane] = 0; else if (lane == 0) *w.touched = 0ull;

            TRESET();
        for (int eb = e0; eb < e1; eb += 32) {
            TMARK(0);
            const int nv = min(32, e1 - eb);
            const bool valid_i = i < nv;
            const int my_ts = valid_i ? (cur.tgt - n0) : 0xff;
            const unsigned my_ss = (unsigned)(cur.src - wb);
            const bool in_win = valid_i && my_ss < 64u;
            const bool oob = valid_i && !in_win;
            wave_lds_fence();
            if constexpr (ST) ew.commit(w.et, dm.EKS, lane); else stage_e_tile<T, EW>(p, dm, w, lane, eb, nv, cur.ep);
            if constexpr (BF) {
                // Branch-free, and split over the two half-waves (both hold the indices of edge slot i): lanes h = 0 keep the
                // target-side tables, lanes h = 1 the source-window side.  Every store is unconditional; a lane with nothing to
                // write aims at its own dummy dword.  (As `if (h == 0) { if (changed) { if (old >= 0) ... } }` this was a chain
                // of ~20 divergent skip branches per tile, each a v_cmp -> saveexec -> branch round trip, and one basic block
                // boundary each for the scheduler.)
                unsigned char* const dmb = reinterpret_cast<unsigned char*>(w.dummy + lane);
                bf16_t* const dmh = reinterpret_cast<bf16_t*>(w.dummy + lane);
                reinterpret_cast<unsigned char*>(h ? w.ssl : w.tsl)[i] =
                    h ? (in_win ? (unsigned char)my_ss : (unsigned char)0xff) : (unsigned char)my_ts;
                *(h ? (w.srcl + i) : (w.dummy + lane)) = oob ? cur.src : -1;
                *((h && in_win) ? (w.touched_b + my_ss) : dmb) = 1;
                const int nA = h ? (in_win ? (int)my_ss : -1) : (valid_i ? my_ts : -1);
                bf16_t* const tabA = (h ? w.oh_w : w.oh_t) + oh_pos(i);
                const bool chg = oh_a != nA, clr = chg && oh_a >= 0, set = chg && nA >= 0;
                *(clr ? tabA + oh_a * OHS : dmh) = 0;
                *(set ? tabA + nA * OHS : dmh) = 0x3F80;
                *((clr && h == 0) ? w.oh_e + i * OHS + oh_a : dmh) = 0;
                *((set && h == 0) ? w.oh_e + i * OHS + nA : dmh) = 0x3F80;
                oh_a = nA;
            } else if (h == 0) {
                reinterpret_cast<unsigned char*>(w.tsl)[i] = (unsigned char)my_ts;
                reinterpret_cast<unsigned char*>(w.ssl)[i] = in_win ? (unsigned char)my_ss : (unsigned char)0xff;
                w.srcl[i] = oob ? cur.src : -1;
                if (in_win) atomicOr(w.touched, 1ull << my_ss);
            }
            wave_lds_fence();
            TMARK(1);

            if constexpr (CP_ != 0 && !XDB && WSP == 0) xf.load(x, dm.C, cur.tgt, cur.src, h);
            // Unconditional on purpose (indices are clamped, so the last tile of a group re-reads valid rows): a path
            // that skips these loads merges into the loop with "the x fragments are the newest loads in flight", and
            // the waits hipcc then puts in front of the MFMAs drain this tile's prefetches as well.
            if constexpr (WSP != 0) pn.load(p.pt, p.ps, nxt.tgt, nxt.src, h, s);
            else if constexpr (CP_ != 0 && XDB) xn.load(x, dm.C, nxt.tgt, nxt.src, h);
            nn.template load<!ST, false>(p, eb + 64, e1, i, n0);
            if constexpr (ST) ew.prefetch(p, lane, eb + 32, 32, nxt.ep);

            TMARK(2);
            f32x16 accf, accs;
#pragma unroll
            for (int r = 0; r < 16; ++r) { accf[r] = bf; accs[r] = bs; }
            if constexpr (WSP != 0) pre_tile_wsp<CP_, 1, 0>(dm, w, lane, s, 0, pf, idf, accf, accs);
            else pre_tile<T, CP_, VEC, WM, NKW, 0, X3>(p, dm, w, lane, s, cur.tgt, cur.src, xf, wr, accf, accs);
            TPIN16(accf); TPIN16(accs);
            TMARK(3);

            // dmv[edge slot][ch] = grad_out[tgt(edge)][ch] / deg : one-hot(edge -> node slot) x gB
            f32x16 dmv;
#pragma unroll
            for (int r = 0; r < 16; ++r) dmv[r] = 0.0f;
            if constexpr (BF) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    dmv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oh_frag(w.oh_e, i, ks, h), gB[ks], dmv, 0, 0, 0);
            } else if constexpr (X3) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 a;                        // one-hot(edge slot i -> node slot 16 ks + 8 h + q), exact in bf16
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[q] = (my_ts == 16 * ks + 8 * h + q) ? (short)0x3F80 : (short)0;
                    dmv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, gBx[ks].lo, dmv, 0, 0, 0);
                    dmv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, gBx[ks].hi, dmv, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int f = 0; f < 16; ++f)
                    dmv = __builtin_amdgcn_mfma_f32_32x32x2f32((my_ts == 2 * f + h) ? 1.0f : 0.0f, gB[f], dmv, 0, 0, 0);
            }

            TPIN16(dmv);
            TMARK(4);
            // gate derivative -> dpre (in place in accf/accs)
            // (dmv is an exact 0 for edge slots >= nv — their one-hot row is empty — so dpre is 0 there)
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float sf, sp_u, ss;
#if MDL_BWD_DERIV2
                    if constexpr (BF) GT::deriv2(accf[r], accs[r], sf, sp_u, ss); else GT::deriv(accf[r], accs[r], sf, sp_u, ss);
#else
                    GT::deriv(accf[r], accs[r], sf, sp_u, ss);
#endif
                    const float t = dmv[r] * sf;
                    accf[r] = (t * GT::M_SCALE) * (1.0f - sf) * sp_u;
                    accs[r] = t * ss;
                }
            }

            // sources outside the window: per-edge fp32 atomics (graphs wider than the window).  Issued as early as the
            // values exist: a pending atomic turns every later wait for a load into vmcnt(0) (loads and stores complete out of
            // order with respect to each other), so the more of this tile's MFMA work lies behind them the better
#ifdef MDL_ABL_NOOOB
            if (false) {
#else
            if (__any(oob) && ch < dm.C) {
#endif
                // the 16 source ids of this lane's rows (d_row(4q+k, h) = 8q + 4h + k) in four 16-byte LDS reads up front:
                // read one by one between the atomics, each costs an LDS round trip the compiler will not hoist
                typedef __attribute__((ext_vector_type(4))) int i32x4;
                int sj[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const i32x4 v = *reinterpret_cast<const i32x4*>(w.srcl + 8 * q + 4 * h);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sj[4 * q + k] = v[k];
                }
                if (BF && p.rs16) {
                    bf16_t* const b16 = reinterpret_cast<bf16_t*>(p.r_src) + (ch & ~1);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int64_t ra = (int64_t)max(sj[r], 0) * C2, rb = (int64_t)max(sj[r + 1], 0) * C2;
                        rsrc16_add2(b16, ra, rb, accf[r], accf[r + 1], sj[r] >= 0, sj[r + 1] >= 0, i & 1);
                        rsrc16_add2(b16 + dm.Cp, ra, rb, accs[r], accs[r + 1], sj[r] >= 0, sj[r + 1] >= 0, i & 1);
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (sj[r] >= 0) {
                        float* dst = p.r_src + (int64_t)sj[r] * C2 + ch;
                        unsafeAtomicAdd(dst, accf[r]);
                        unsafeAtomicAdd(dst + dm.Cp, accs[r]);
                    }
                }
                }
            }

            TPIN16(accf); TPIN16(accs);
            TMARK(5);
            DFrags<T> dp;
            DFragsX3 dpx;
            if constexpr (X3) dpx.pack(accf, accs); else dp.pack(accf, accs);
            TMARK(6);
            if constexpr (BF) {
                seg_reduce2_tab(dp, w.oh_t, i, h, Rf, Rs);                  // by target  -> r_tgt
                TPIN16(Rf); TPIN16(Rs);
                TMARK(7);
#ifndef MDL_ABL_NOWIN
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) seg_reduce2_tab(dp, w.oh_w, i + 32 * mt, h, Wf[mt], Ws[mt]);   // by source window
#endif
            } else {
                unsigned t4[4], s4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { t4[j] = w.tsl[2 * j + h]; s4[j] = w.ssl[2 * j + h]; }
                if constexpr (X3) {
                    seg_reduce2_x3(dpx, t4, (unsigned)i, Rf, Rs);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) seg_reduce2_x3(dpx, s4, (unsigned)(i + 32 * mt), Wf[mt], Ws[mt]);
                } else {
                    seg_reduce2<T>(dp, t4, (unsigned)i, Rf, Rs);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) seg_reduce2<T>(dp, s4, (unsigned)(i + 32 * mt), Wf[mt], Ws[mt]);
                }
            }

            TMARK(8);
            // dwe[ch][gcol] += sum_slot dpre[slot][ch] * e[slot][gcol]
            //   A = dpre^T (lane = channel, k = edge slots: own registers), B = e tile column (LDS)
#ifdef MDL_ABL_NODWE
            if (false)
#endif
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt) {
                const int gcol = nt * 32 + i;
                if (nt * 32 < dm.KE) {
                    if constexpr (BF) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            bf16x8 b;
                            if constexpr (ST) {
                                // column fragment of the row-major e tile with the LDS transpose read: each 16-lane group
                                // reads a [4 rows][16 cols] block (lane t supplies row t>>2, cols 4*(t&3)..) and lane t gets
                                // column t of the 4 rows.  K slots 8ks+q = rows 16ks+4h+q (q<4) and 16ks+8+4h+(q-4).
                                typedef __attribute__((ext_vector_type(4))) short s16x4;
                                typedef __attribute__((address_space(3))) s16x4* lds4_t;
                                const int t = i & 15;
                                const bf16_t* base = w.et + (16 * ks + 4 * h + (t >> 2)) * dm.EKS + nt * 32 + (i & 16) + 4 * (t & 3);
                                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base));
                                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(base + 8 * dm.EKS));
                                b = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q)
                                    b[q] = (gcol < dm.KE) ? (short)w.et[d_row(8 * ks + q, h) * dm.EKS + gcol] : (short)0;
                            }
                            dwe_acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp.f[ks], b, dwe_acc[0][nt], 0, 0, 0);
                            dwe_acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp.s[ks], b, dwe_acc[1][nt], 0, 0, 0);
                        }
                    } else if constexpr (X3) {
                        // both operands split: dpre_hi e_hi + dpre_lo e_hi + dpre_hi e_lo (k = the tile's 32 edge slots)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            float ev[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) ev[q] = w.et[d_row(8 * ks + q, h) * dm.EKS + gcol];
                            const SplitFrag b = split8(f32x4{ev[0], ev[1], ev[2], ev[3]}, f32x4{ev[4], ev[5], ev[6], ev[7]});
                            dwe_acc[0][nt] = mma_x3(dpx.f[ks], b, dwe_acc[0][nt]);
                            dwe_acc[1][nt] = mma_x3(dpx.s[ks], b, dwe_acc[1][nt]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float b = (gcol < dm.KE) ? w.et[d_row(r, h) * dm.EKS + gcol] : 0.0f;
                            dwe_acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(accf[r], b, dwe_acc[0][nt], 0, 0, 0);
                            dwe_acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(accs[r], b, dwe_acc[1][nt], 0, 0, 0);
                        }
                    }
                }
            }
            cur = nxt;
            nxt = nn;
            if constexpr (WSP != 0) pf = pn; else if constexpr (XDB) xf = xn;
            TMARK(9);
            TTILE();
        }

        // the id of the next group (requested at the top of this one) is read BEFORE the flush: behind ~100 atomics and
        // stores, waiting for any returning operation means waiting for all of them
        const int n0_dyn = dyn ? group_start(__builtin_amdgcn_readfirstlane(gpend)) : 0;
        // flush the source window: one atomic row update per touched window node (instead of per edge)
        wave_lds_fence();
        {
            // (the mask is the same in every lane: SGPRs; d_row(r, h) = d_row(r, 0) + 4h: one per-lane shift, then constant
            // bit tests — per-row masks 1 << sl hoisted out of the group loop are the first thing hipcc spills)
            unsigned long long tm;
            if constexpr (BF) {
                tm = __ballot(w.touched_b[lane] != 0);
            } else {
                const unsigned long long tmv = *w.touched;
                tm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(tmv >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tmv);
            }
            const unsigned long long tmh = tm >> (4 * h);
#ifdef MDL_ABL_NOWFLUSH
            if (p.N < 0)
#endif
            if (BF && p.rs16) {
                // (static bf16 shapes only: C == Cp, every lane owns a real column)
                bf16_t* const b16 = reinterpret_cast<bf16_t*>(p.r_src) + (int64_t)(wb + 4 * h) * C2 + (ch & ~1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int ra = 32 * mt + d_row(r, 0), rb = ra + 1;
                        const bool ca = (tmh >> ra) & 1ull, cb = (tmh >> rb) & 1ull;
                        rsrc16_add2(b16, (int64_t)ra * C2, (int64_t)rb * C2, Wf[mt][r], Wf[mt][r + 1], ca, cb, i & 1);
                        rsrc16_add2(b16 + dm.Cp, (int64_t)ra * C2, (int64_t)rb * C2, Ws[mt][r], Ws[mt][r + 1], ca, cb, i & 1);
                    }
            } else if (ch < dm.C) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int sl = 32 * mt + d_row(r, h);
                        if ((tmh >> (32 * mt + d_row(r, 0))) & 1ull) {          // bit sl of tm
                            float* dst = p.r_src + (int64_t)(wb + sl) * C2 + ch;
                            unsafeAtomicAdd(dst, Wf[mt][r]);
                            unsafeAtomicAdd(dst + dm.Cp, Ws[mt][r]);
                        }
                    }
            }
        }
        // r_tgt rows of this group (each written exactly once); bias gradient = their column sums
#pragma unroll
        for (int r = 0; r < 16; ++r) { dbf_acc += Rf[r]; dbs_acc += Rs[r]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + d_row(r, h);
            if (n < n1) {
                T* dst = static_cast<T*>(p.r_tgt) + (int64_t)n * C2 + ch;
                Elem<T>::st(dst, Rf[r]);
                Elem<T>::st(dst + dm.Cp, Rs[r]);
            }
        }
        TMARK(11);
        n0 = dyn ? n0_dyn : n1;
    }

    TFLUSH(16);
    if (p.db) {
        dbf_acc += __shfl_xor(dbf_acc, 32);
        dbs_acc += __shfl_xor(dbs_acc, 32);
        if (h == 0) {
            unsafeAtomicAdd(p.db + ch, dbf_acc);
            unsafeAtomicAdd(p.db + dm.Cp + ch, dbs_acc);
        }
    }
    // flush the wave's dwe partial sums: D rows = channel slot d_row(r,h) of slice s, cols = feature
#ifdef MDL_ABL_NODWEFLUSH      // (timing experiments: wrong results)
    if (p.N >= 0) return;
#endif
    if (p.dwe_combine) {
        // Two slices, four waves: waves w and w + 2 of a workgroup hold partial sums of the SAME slice.  The flush is 64 atomic
        // instructions per wave on addresses every wave of the slice hits (12.5 of 45 us at the reference's batch size, where a
        // wave has one or two tiles of work in front of it): wave w + 2 hands its sums over through the LDS area of the weights —
        // free once every wave is out of its loop — and only wave w flushes.
        const int wv4 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        float* const cb = reinterpret_cast<float*>(smem) + (wv4 & 1) * (2 * GNT * 16 * WAVE) + lane;
        __syncthreads();
        if (wv4 >= 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int nt = 0; nt < GNT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cb[((a * GNT + nt) * 16 + r) * WAVE] = dwe_acc[a][nt][r];
        }
        __syncthreads();
        if (wv4 >= 2) return;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int nt = 0; nt < GNT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) dwe_acc[a][nt][r] += cb[((a * GNT + nt) * 16 + r) * WAVE];
    }
#pragma unroll
    for (int nt = 0; nt < GNT; ++nt) {
        const int gcol = nt * 32 + i;
        if (gcol < dm.G) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = s * 32 + d_row(r, h);
                unsafeAtomicAdd(p.dwe + (int64_t)c * p.GP + gcol, dwe_acc[0][nt][r]);
                unsafeAtomicAdd(p.dwe + (int64_t)(dm.Cp + c) * p.GP + gcol, dwe_acc[1][nt][r]);
            }
        }
    }
}

#if MDL_EXPERIMENTS
#include "../../experiments/csrc/cgconv_ab.inc"   // saved-gate backward edge pass: net zero against the recomputing pass
#include "../../experiments/csrc/cgconv_cb.inc"   // namespace mdl::cb (cooperative weight-stationary kernels: measured slower)
#endif
// The edge-per-lane backward (cgconv_ep.inc, namespace mdl::ep; opt-in, MDL_CG_EP=1) lives in its own translation
// unit: cgconv_ep.hip defines MDL_CG_EP_TU and includes THIS file, so that it sees the shared tile machinery above, and is
// compiled with -mllvm -amdgpu-mfma-vgpr-form=1 — its phase-A MFMA results then land in the VGPRs the gate arithmetic reads
// (the default AGPR form costs one v_accvgpr_read per value, 96 per tile), while the per-wave kernels of this unit, which
// live on 256 + 179 registers, need the AGPR form.
#ifdef MDL_CG_EP_TU
#include "cgconv_ep_common.inc"
#if MDL_EXPERIMENTS
#include "../../experiments/csrc/cgconv_ep.inc"   // phases one after the other: measured slower than the per-wave kernel
#endif
#include "cgconv_ep2.inc"
namespace ep {
static bool cg_env_ep2_static() {          // experiments build, MDL_EP2_STATIC=1: kernel 2 without the dynamic tail (A/B)
#if MDL_EXPERIMENTS
    static const bool v = [] { const char* s = getenv("MDL_EP2_STATIC"); return s && atoi(s) != 0; }();
    return v;
#else
    return true;
#endif
}
int launch2(CgParams& p, hipStream_t st, int wgs, const char* name) {
    typedef Cfg2<64> F;
    const int64_t eg = std::min<int64_t>(wgs > 0 ? wgs : 256, std::max<int64_t>(1, cdiv(p.E, 32 * F::NA * 2)));
    auto kf = bwd2_kernel<64>;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), F::LDS);
    if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, F::LDS, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    // optional caller workspace: the chunk counter of the dynamic tail, zeroed on the stream
    if (MDL_EP2_TAIL == 0 || cg_env_ep2_static()) p.ctr = nullptr;      // (the dynamic tail: experiments build with -DMDL_EP2_TAIL=25)
    if (p.ctr && hipMemsetAsync(p.ctr, 0, 64, st) != hipSuccess) p.ctr = nullptr;
    hipLaunchKernelGGL(kf, dim3((unsigned)eg), dim3(F::NT), F::LDS, st, p);
    return check_launch(name);
}
#if MDL_EXPERIMENTS
int launch(CgParams& p, hipStream_t st, int wgs, const char* name) {
    typedef Cfg<64> F;
    // one workgroup per CU; small problems: at least two rounds of tiles per workgroup
    const int64_t eg = std::min<int64_t>(wgs > 0 ? wgs : 256, std::max<int64_t>(1, cdiv(p.E, 32 * F::NW * 2)));
    auto kf = bwd_kernel<64>;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), F::LDS);
    if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, F::LDS, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    hipLaunchKernelGGL(kf, dim3((unsigned)eg), dim3(F::NT), F::LDS, st, p);
    return check_launch(name);
}
#endif
}  // namespace ep
}  // namespace mdl
#else
namespace ep {
#ifndef MDL_EP_DEFAULT
#define MDL_EP_DEFAULT 2      // edge-per-lane backward edge pass for bf16, C = 64, G = 50: 2 = cgconv_ep2.inc where the by-source
                              // sums are bf16 (mdl_cgconv_bwd_h): 6-9 % faster than the per-wave kernel on the bench batch; 0 = the
                              // per-wave kernel always.  Callers override per launch with MDL_K3_PER_WAVE / MDL_K3_EDGE_LANE.
#endif
#ifndef MDL_EP2_MIN_EDGES
#define MDL_EP2_MIN_EDGES 400000   // default selection of kernel 2: at least ~12 rounds per workgroup (MDL_K3_EDGE_LANE forces it)
#endif
#if MDL_EXPERIMENTS
int launch(CgParams& p, hipStream_t st, int wgs, const char* name);      // experiments/csrc/cgconv_ep.inc: phases one after the other
#endif
int launch2(CgParams& p, hipStream_t st, int wgs, const char* name);     // cgconv_ep2.inc: producer / reducer waves side by side
}

// ------------------------------------------------------------------------------------------
// Weight packing: nn.Linear [C, 2C+G] (target|source|edge) -> [2Cp][WS] with K order [e|x_tgt|x_src]
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void cgconv_pack_body(const float* __restrict__ wf, const float* __restrict__ bfv,
                                                 const float* __restrict__ ws, const float* __restrict__ bsv,
                                                 int C, int G, const CgDims& d, T* __restrict__ wpack,
                                                 float* __restrict__ bpack, float scale, int bias_col,
                                                 bf16_t* __restrict__ wn_t, int x3 = 0) {
    const int total = 2 * d.Cp * d.WS;
    const int ldw = 2 * C + G;
    if (wn_t) {
        // (mdl_cgconv_pack_weights_node) the backward node kernel's operand in the same launch: wn_t [C][4Cp] = transpose of
        // Wn = rows (f_tgt, s_tgt, f_src, s_src) of the two Linears' node columns, unscaled
        const int tn = C * 4 * d.Cp;
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < tn; q += gridDim.x * blockDim.x) {
            const int k = q / (4 * d.Cp), r = q - k * (4 * d.Cp);
            const int blk = r / d.Cp, c = r - blk * d.Cp;
            float v = 0.0f;
            if (c < C) v = ((blk & 1) ? ws : wf)[c * ldw + (blk >> 1) * C + k];
            wn_t[q] = f2bf(v);
        }
    }
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
        const int row = q / d.WS, k = q - row * d.WS;
        const int part = row / d.Cp, c = row - part * d.Cp;
        const float* W = part ? ws : wf;
        float v = 0.0f;
        if (c < C) {
            if (k < d.KE) {
                if (k < G) v = W[c * ldw + 2 * C + k];
                else if (k == G && bias_col) { const float* b = part ? bsv : bfv; v = b ? b[c] : 0.0f; }
            } else if (k < d.KE + d.Cp) {
                const int kc = k - d.KE;
                if (kc < C) v = W[c * ldw + kc];
            } else if (k < d.KT) {
                const int kc = k - d.KE - d.Cp;
                if (kc < C) v = W[c * ldw + C + kc];
            }
        }
        if constexpr (std::is_same<T, float>::value) {
            if (x3) {
                // x3 layout (ld_wfrag_x3): per group of eight k-values [8 x hi bf16 | 8 x lo bf16] in the 32 bytes of the eight
                // floats; the row's padding dwords are never read
                if (k < d.KT) {
                    unsigned hi, lo;
                    split_pair(v * scale, 0.0f, hi, lo);
                    bf16_t* g = reinterpret_cast<bf16_t*>(wpack + row * d.WS) + (k >> 3) * 16 + (k & 7);
                    g[0] = (bf16_t)(hi & 0xffffu);
                    g[8] = (bf16_t)(lo & 0xffffu);
                } else {
                    wpack[q] = 0.0f;
                }
                continue;
            }
        }
        Elem<T>::st(wpack + q, v * scale);
    }
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < 2 * d.Cp; q += gridDim.x * blockDim.x) {
        const int part = q / d.Cp, c = q - part * d.Cp;
        const float* b = part ? bsv : bfv;
        bpack[q] = (c < C && b) ? b[c] * scale : 0.0f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cgconv_pack_kernel(const float* __restrict__ wf, const float* __restrict__ bfv,
                                                          const float* __restrict__ ws, const float* __restrict__ bsv,
                                                          int C, int G, CgDims d, T* __restrict__ wpack,
                                                          float* __restrict__ bpack, float scale, int bias_col,
                                                          bf16_t* __restrict__ wn_t = nullptr, int x3 = 0) {
    cgconv_pack_body<T>(wf, bfv, ws, bsv, C, G, d, wpack, bpack, scale, bias_col, wn_t, x3);
}

// every conv layer of a model in ONE launch (blockIdx.y = layer): the layers' weights are all known before the first one runs,
// and a pack launch per layer is 5 us of a step that is launch-bound at the reference's batch size (mdl_cgconv_pack_weights_multi)
constexpr int PACK_MAXL = 16;
struct PackMulti {
    const float* wf[PACK_MAXL]; const float* bf[PACK_MAXL]; const float* ws[PACK_MAXL]; const float* bs[PACK_MAXL];
    void* wpack[PACK_MAXL]; float* bpack[PACK_MAXL]; bf16_t* wn_t[PACK_MAXL];
};
template <typename T>
__global__ __launch_bounds__(256) void cgconv_pack_multi_kernel(PackMulti a, int C, int G, CgDims d, float scale, int bias_col, int x3 = 0) {
    const int l = blockIdx.y;
    cgconv_pack_body<T>(a.wf[l], a.bf[l], a.ws[l], a.bs[l], C, G, d, static_cast<T*>(a.wpack[l]), a.bpack[l], scale, bias_col, a.wn_t[l], x3);
}

#if MDL_EXPERIMENTS
// W-split packing: the edge-feature part [2Cp][KE + pad] (bias in column G) and the two projection weights
// wproj[side][2Cp][Cp] (side 0 target, 1 source; rows f then s; the `w` operand [M = 2Cp, K = C] of mdl_linear_act),
// everything scaled like the packed weights.
__global__ __launch_bounds__(256) void cgconv_pack_split_kernel(const float* __restrict__ wf, const float* __restrict__ bfv,
                                                                const float* __restrict__ ws, const float* __restrict__ bsv,
                                                                int C, int G, int Cp, int KE, int WSe, bf16_t* __restrict__ wpe,
                                                                bf16_t* __restrict__ wproj, float scale) {
    const int ldw = 2 * C + G;
    const int ne = 2 * Cp * WSe, np_ = 2 * 2 * Cp * Cp;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < ne + np_; q += gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (q < ne) {
            const int row = q / WSe, k = q - row * WSe;
            const int part = row / Cp, c = row - part * Cp;
            const float* W = part ? ws : wf;
            if (c < C) {
                if (k < G) v = W[c * ldw + 2 * C + k];
                else if (k == G) { const float* b = part ? bsv : bfv; v = b ? b[c] : 0.0f; }
            }
            wpe[q] = f2bf(v * scale);
        } else {
            const int r = q - ne;
            const int side = r / (2 * Cp * Cp), rr = r - side * (2 * Cp * Cp);
            const int row = rr / Cp, k = rr - row * Cp;
            const int part = row / Cp, c = row - part * Cp;
            const float* W = part ? ws : wf;
            if (c < C && k < C) v = W[c * ldw + side * C + k];
            wproj[r] = f2bf(v * scale);
        }
    }
}
#endif

// ------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------
static constexpr int LDS_CAP = 160 * 1024;

// Experiment switches.  libmdl_hip.so never reads the environment: variants a caller or a test wants are explicit flag bits
// in the `dtype` argument (MDL_DETERMINISTIC, MDL_K3_PER_WAVE, MDL_K3_EDGE_LANE).  The experiments build (experiments/build.py,
// -DMDL_EXPERIMENTS=1) reads them from the environment ONCE (first launch).
struct CgEnv {
    int64_t grid_cap;     // MDL_GRID_CAP: upper bound on the grid (0 = none)
    int cb_fwd, cb_bwd;   // MDL_CG_CB / MDL_CG_CB_BWD: cooperative column-block kernels (-1 = compile-time default)
    int cb_wgs;           // MDL_CB_WGS: their workgroups per CU (0 = default)
    int ab_wgs;           // MDL_AB_WGS: workgroups per CU of the saved-gate backward (0 = default 1)
    int no_half_groups;   // MDL_CG_NO_HALF=1: dynamic backward schedule without the half-group tail (A/B)
    int no_fast128;       // MDL_CG_NO_FAST128=1: 128-channel layers take the generic kernels (A/B)
    int no_w_slice;       // MDL_CG_NO_WSLICE=1: wide layers read the packed weights from global memory as before (A/B)
    int ep;               // MDL_CG_EP: edge-per-lane backward edge pass; -1 = compile-time default
    int ep_wgs;           // MDL_EP_WGS: its grid cap (0 = one workgroup per CU)
};
static const CgEnv& cg_env() {
#if MDL_EXPERIMENTS
    static const CgEnv e = [] {
        CgEnv v;
        const char* s;
        v.grid_cap = (s = getenv("MDL_GRID_CAP")) ? atoll(s) : 0;
        v.cb_fwd = (s = getenv("MDL_CG_CB")) ? (atoi(s) != 0) : -1;
        v.cb_bwd = (s = getenv("MDL_CG_CB_BWD")) ? (atoi(s) != 0) : -1;
        v.cb_wgs = (s = getenv("MDL_CB_WGS")) ? atoi(s) : 0;
        v.ab_wgs = (s = getenv("MDL_AB_WGS")) ? atoi(s) : 0;
        v.no_half_groups = (s = getenv("MDL_CG_NO_HALF")) ? (atoi(s) != 0) : 0;
        v.no_w_slice = (s = getenv("MDL_CG_NO_WSLICE")) ? (atoi(s) != 0) : 0;
        v.no_fast128 = (s = getenv("MDL_CG_NO_FAST128")) ? (atoi(s) != 0) : 0;
        v.ep = (s = getenv("MDL_CG_EP")) ? atoi(s) : -1;      // 0 off, 1 cgconv_ep.inc, 2 cgconv_ep2.inc
        v.ep_wgs = (s = getenv("MDL_EP_WGS")) ? atoi(s) : 0;
        return v;
    }();
#else
    static const CgEnv e = {0, -1, -1, 0, 0, 0, 0, 0, -1, 0};
#endif
    return e;
}

// which backward edge pass the last mdl_cgconv_bwd* call of this PROCESS launched (mdl_debug_last_k3: tests assert that the
// kernel they mean to check is the one that ran; not thread-local — autograd runs the backward on its own thread):
// 1 per-wave, 2 edge-per-lane kernel 2, 3 per-wave in deterministic shape.  A debug value, no part of the data path.
static volatile int g_last_k3 = 0;

template <typename T>
static int cg_launch(bool bwd, CgParams& p, int dtype, hipStream_t st, const char* name) {
    const bool x3 = (p.flags & MDL_SPLIT_BF16) != 0;
    const CgDims d = cg_dims(p.C, p.G, dtype, x3);
    p.Cp = d.Cp; p.KE = d.KE; p.KT = d.KT; p.WS = d.WS; p.EKS = d.EKS; p.NS = d.NS; p.GP = d.GP;
    if (p.ldwe > 0) p.GP = p.ldwe;      // dwe rows straight into the caller's [2C, 2C + G] weight-gradient matrix (MdlCgConv.ld_dwe)
    p.w_elems = 2 * d.Cp * d.WS;
    const bool wsp = p.pt != nullptr;  // W-split entry points: packed weights = edge-feature part only
    if (wsp) { p.WS = d.EKS; p.w_elems = 2 * d.Cp * d.EKS; }
    p.bias_col = (p.G % 16) != 0;      // a zero-padding K column is free to carry the bias
    p.n_groups = (int)cdiv(p.N, 32);
    if (p.n_groups == 0) return MDL_OK;

    // staging word: 4 bytes when the row length allows it
    const bool word4 = (p.G * sizeof(T)) % 4 == 0 && (reinterpret_cast<uintptr_t>(p.ea) % 4) == 0;
    const int EW = word4 ? (int)(4 / sizeof(T)) : 1;
    p.GW = p.G / EW;
    p.gw_inv = (unsigned)((0x100000000ull + p.GW - 1) / p.GW);

    int vec = 1;
    if (sizeof(T) == 2) {
        const bool a16 = reinterpret_cast<uintptr_t>(p.x) % 16 == 0, a8 = reinterpret_cast<uintptr_t>(p.x) % 8 == 0;
        if (p.C % 8 == 0 && a16) vec = 8; else if (p.C % 4 == 0 && a8) vec = 4;
    }

    const int et_bytes = (32 * d.EKS * (int)sizeof(T) + 15) & ~15;
    // e tile, tgt-slot bytes, source ids, src-slot bytes, bitmap, one-hot tables (32 + 32 + 64 rows)
    p.wave_lds_bytes = et_bytes + 32 + 128 + 32 + 16 + ((bwd && sizeof(T) == 2) ? 128 * OHS * 2 + 64 + 256 : 0) +
                       ((!bwd && p.bn_sums) ? 2 * d.Cp * 4 : 0);      // forward with BatchNorm statistics: the wave's [2][Cp] running sums
    const int w_bytes = (p.w_elems * (int)sizeof(T) + 15) & ~15;
    // MDL_DETERMINISTIC (backward): ONE wave per channel slice walks the whole batch — every sum this kernel forms with atomics
    // (r_src rows, dwe, db) then receives its terms from a single wave in program order, i.e. the same bits on every run.
    // For HIP-vs-HIP tests (graph replay vs eager, padded rows, data-parallel exchange); ~1/500 of the throughput.
    const bool det = bwd && (p.flags & MDL_DETERMINISTIC) != 0;
    const int waves = det ? 1 : (bwd ? 4 : MDL_FWD_THREADS / 64);
    // static fast shapes keep W in registers (no LDS copy); otherwise LDS if it fits, else global
    const bool fast = !p.eperm && p.G == 50 && p.C == d.Cp && (d.Cp == 32 || d.Cp == 64) &&
                      (sizeof(T) == 2 ? (vec == 8 && EW == 2) : d.Cp == 64) &&
                      (bwd || (reinterpret_cast<uintptr_t>(p.out) % 16 == 0 && reinterpret_cast<uintptr_t>(p.x) % 16 == 0));
    // 128 channels (the reference's default width 100, padded by the caller): the same static code, one channel slice per
    // wave, the workgroup's slice of W in LDS (w_slice below) — the whole W is 168 KB
    const bool fast128 = sizeof(T) == 2 && !wsp && !p.ab && !p.eperm && p.G == 50 && p.C == 128 && d.Cp == 128 && vec == 8 && EW == 2 &&
                         p.bias_col && reinterpret_cast<uintptr_t>(p.x) % 16 == 0 &&
                         (bwd ? reinterpret_cast<uintptr_t>(p.gout) % 16 == 0 : reinterpret_cast<uintptr_t>(p.out) % 16 == 0) &&
                         !cg_env().no_fast128;
    bool w_lds = (!fast || MDL_CG_WM != 2) && w_bytes + waves * p.wave_lds_bytes <= LDS_CAP;
    int lds = (w_lds ? w_bytes : 0) + waves * p.wave_lds_bytes;
    // packed weights that do not fit (C = 100 -> Cp = 128: 168 KB): every workgroup keeps the 64 rows of ONE channel slice
    p.w_slice = 0;
    if (!w_lds && !fast && !wsp && d.NS > 1 && (d.WS * (int)sizeof(T)) % 16 == 0 &&
        64 * d.WS * (int)sizeof(T) + waves * p.wave_lds_bytes <= LDS_CAP && !cg_env().no_w_slice) {
        p.w_slice = 1;
        p.w_elems = 64 * d.WS;
        w_lds = true;
        lds = 64 * d.WS * (int)sizeof(T) + waves * p.wave_lds_bytes;
    }
    const int wg_per_cu = lds * 2 <= LDS_CAP ? 2 : 1;

    const bool all_slices = !bwd && fast && MDL_FWD_ALLSLICES && MDL_CG_WM == 1 && (sizeof(T) == 2 || w_lds);
    // one node range per wave (per slice), at least ~2 edge tiles each; see NodeRange
    const int64_t ranges = std::max<int64_t>(1, std::min<int64_t>(cdiv(p.E, bwd ? MDL_BWD_RANGE_EDGES : MDL_FWD_RANGE_EDGES), p.N));
    int64_t items = ranges * (all_slices ? 1 : d.NS);
    int64_t grid = cdiv(items, waves);
    // backward is register-allocated for MDL_BWD_WAVES waves per SIMD: 1 -> one 4-wave workgroup per CU
    const int64_t cap = 256 * ((bwd && MDL_BWD_WAVES == 1) ? 1 : wg_per_cu);
    if (grid > cap) grid = cap;
    const CgEnv& env = cg_env();
    if (env.grid_cap > 0 && grid > env.grid_cap) grid = env.grid_cap;   // experiments build
    if (det) { grid = d.NS; p.ctr = nullptr; }
    // edge-per-lane backward (cgconv_ep2.inc): bf16, C = 64, G = 50, target-sorted edge features, bf16 by-source sums
    if constexpr (sizeof(T) == 2) {
        int ep_sel = env.ep >= 0 ? env.ep : MDL_EP_DEFAULT;
        if (p.flags & MDL_K3_PER_WAVE) ep_sel = 0;
        const bool force2 = env.ep == 2 || (p.flags & MDL_K3_EDGE_LANE) != 0;
        if (force2) ep_sel = 2;
        if (ep_sel != 0 && !det && bwd && fast && !wsp && d.Cp == 64 && p.bias_col && p.E >= 64) {
            // (a workgroup of kernel 2 stages 51 KB of weights and clears 112 KB of tile buffers before its first tile: below a few
            // rounds per workgroup the per-wave kernel wins — 0.64 vs 0.71 ms per step at the reference's batch size 100)
            if (ep_sel == 2 && p.rs16 && (force2 || p.E >= MDL_EP2_MIN_EDGES)) { g_last_k3 = 2; return ep::launch2(p, st, env.ep_wgs, name); }
#if MDL_EXPERIMENTS
            if (ep_sel == 1 && !p.rs16) return ep::launch(p, st, env.ep_wgs, name);      // fp32 by-source sums (mdl_cgconv_bwd)
#endif
        }
    }
    if (bwd) g_last_k3 = det ? 3 : 1;
    // per-wave backward, 4 waves on 2 slices with the weights in LDS (>= 2 x 16 KB: the hand-over area): pair-wise dwe flush
    p.dwe_combine = (bwd && !det && waves == 4 && d.NS == 2 && w_lds && !p.w_slice && w_bytes >= 2 * 2 * 2 * 16 * 64 * 4) ? 1 : 0;
    // total waves must be a multiple of NS so that every wave keeps one channel slice (w_slice: whole workgroups)
    while ((grid * waves) % d.NS) ++grid;
    if (p.w_slice) while (grid % d.NS) ++grid;
    // dynamic group scheduling only pays when every wave gets several 32-node groups
    p.g_full = p.n_groups;
    if (bwd && p.ctr) {
        if ((int64_t)p.n_groups * d.NS >= 4 * grid * waves && d.NS <= 16) {
            if (hipMemsetAsync(p.ctr, 0, 64, st) != hipSuccess) { set_error("%s: workspace memset failed", name); return MDL_E_LAUNCH; }
            // the last two groups per wave of a slice are handed out as four half groups
            const int64_t tail = 2 * (grid * waves / d.NS);
            p.g_full = (int)std::max<int64_t>(0, (int64_t)p.n_groups - tail);
            if (env.no_half_groups) p.g_full = p.n_groups;
        } else {
            p.ctr = nullptr;
        }
    }

#if MDL_EXPERIMENTS
    // training forward that also stores the gate factors for cgconv_bwd_ab_kernel (static bf16 shapes only)
    if constexpr (sizeof(T) == 2) {
        if (!bwd && p.ab) {
            if (!(fast && all_slices && p.bias_col)) {
                set_error("%s: the saved-gate forward supports bf16, C in {32, 64}, G = 50, target-sorted edge features", name);
                return MDL_E_UNSUPP;
            }
            if (d.Cp == 64) {
                auto kf = cgconv_fwd_kernel<T, 64, 50, 9, 2, 1, true>;
                (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
                hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);
            } else {
                auto kf = cgconv_fwd_kernel<T, 32, 50, 9, 2, 1, true>;
                (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
                hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);
            }
            return check_launch(name);
        }
    }

    // cooperative column-block kernels (cgconv_cb.inc): bf16 static shapes
    if constexpr (sizeof(T) == 2) {
        const bool use_cb = env.cb_fwd >= 0 ? env.cb_fwd != 0 : (MDL_CG_CB_DEFAULT != 0);
        if (use_cb && fast && !wsp && !bwd && p.E >= 64 && p.bias_col) {   // (E >= 64: the kernels' edge-feature window is 1024 dwords)
            const int cb_wgs = env.cb_wgs > 0 ? env.cb_wgs : MDL_CB_FWD_WG_PER_CU;
            int64_t cb_grid = std::min<int64_t>(256 * cb_wgs, ranges);
            if (env.grid_cap > 0 && cb_grid > env.grid_cap) cb_grid = env.grid_cap;
            if (d.Cp == 64) {
                const int cb_lds = 2 * cb::Cfg<64>::BUF_FWD;
                hipLaunchKernelGGL(cb::fwd_kernel<64>, dim3((unsigned)cb_grid), dim3(cb::Cfg<64>::NT), cb_lds, st, p);
            } else {
                const int cb_lds = 2 * cb::Cfg<32>::BUF_FWD;
                hipLaunchKernelGGL(cb::fwd_kernel<32>, dim3((unsigned)cb_grid), dim3(cb::Cfg<32>::NT), cb_lds, st, p);
            }
            return check_launch(name);
        }
        const bool use_cbb = env.cb_bwd >= 0 ? env.cb_bwd != 0 : (MDL_CG_CB_BWD_DEFAULT != 0);
        if (use_cbb && fast && !wsp && !p.rs16 && bwd && p.E >= 64 && p.bias_col) {
            const int cb_wgs = env.cb_wgs > 0 ? env.cb_wgs : MDL_CB_BWD_OCC;
            const int64_t cb_grid = std::min<int64_t>(256 * cb_wgs, ranges);
            if (d.Cp == 64) {
                const int cb_lds = 2 * cb::BwdLds<64>::BUF + cb::Cfg<64>::NCB * 32 * (cb::Cfg<64>::KE + 8) * 2;
                auto kf = cb::bwd_kernel<64>;
                set_max_dynamic_lds(reinterpret_cast<const void*>(kf), cb_lds);
                hipLaunchKernelGGL(kf, dim3((unsigned)cb_grid), dim3(cb::Cfg<64>::NT), cb_lds, st, p);
            } else {
                const int cb_lds = 2 * cb::BwdLds<32>::BUF + cb::Cfg<32>::NCB * 32 * (cb::Cfg<32>::KE + 8) * 2;
                auto kf = cb::bwd_kernel<32>;
                set_max_dynamic_lds(reinterpret_cast<const void*>(kf), cb_lds);
                hipLaunchKernelGGL(kf, dim3((unsigned)cb_grid), dim3(cb::Cfg<32>::NT), cb_lds, st, p);
            }
            return check_launch(name);
        }
    }

#endif

    if constexpr (sizeof(T) == 2) {
        if (!bwd && p.bn_sums) {       // forward with the BatchNorm statistics in its epilogue: the static all-slices kernel only
            if (!(fast && all_slices && w_lds)) { set_error("%s: BatchNorm statistics need the static bf16 kernels", name); return MDL_E_UNSUPP; }
            auto kf = d.Cp == 64 ? cgconv_fwd_kernel<T, 64, 50, 9, 2, 1, false, 0, true> : cgconv_fwd_kernel<T, 32, 50, 9, 2, 1, false, 0, true>;
            hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
            if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
            hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);
            return check_launch(name);
        }
    } else if (!bwd && p.bn_sums) {
        set_error("%s: BatchNorm statistics are bf16 only", name);
        return MDL_E_UNSUPP;
    }
#define MDL_CG_LAUNCH(CP_, G_, VEC_, EW_, WM_)                                                               \
    do {                                                                                                     \
        auto kf = bwd ? cgconv_bwd_kernel<T, CP_, G_, VEC_, EW_, WM_> : cgconv_fwd_kernel<T, CP_, G_, VEC_, EW_, WM_>; \
        hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                 \
        if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, lds, hipGetErrorString(e)); return MDL_E_LAUNCH; } \
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);                           \
    } while (0)
#define MDL_CG_BY_WL(VEC_, EW_) do { if (w_lds) MDL_CG_LAUNCH(0, 0, VEC_, EW_, 1); else MDL_CG_LAUNCH(0, 0, VEC_, EW_, 0); } while (0)
#define MDL_CG_BY_EW(VEC_) do { if (EW == 2) MDL_CG_BY_WL(VEC_, 2); else MDL_CG_BY_WL(VEC_, 1); } while (0)

#if MDL_EXPERIMENTS
    if (wsp) {
        // W-split kernels: bf16, static shapes, W (edge part) in LDS
        if constexpr (sizeof(T) == 2) {
            if (!(fast && w_lds && p.bias_col && (bwd || all_slices))) {
                set_error("%s: the W-split kernels support bf16, C in {32, 64}, G = 50, target-sorted edge features", name);
                return MDL_E_UNSUPP;
            }
#define MDL_CG_LAUNCH_WSP(CP_)                                                                                         \
            do {                                                                                                       \
                hipError_t e;                                                                                          \
                if (bwd) {                                                                                             \
                    auto kf = cgconv_bwd_kernel<T, CP_, 50, 9, 2, 1, 1>;                                               \
                    e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                   \
                    if (e == hipSuccess) hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);   \
                } else {                                                                                               \
                    auto kf = cgconv_fwd_kernel<T, CP_, 50, 9, 2, 1, false, 1>;                                        \
                    e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                   \
                    if (e == hipSuccess) hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);   \
                }                                                                                                      \
                if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, lds, hipGetErrorString(e)); return MDL_E_LAUNCH; } \
            } while (0)
            if (d.Cp == 64) MDL_CG_LAUNCH_WSP(64); else MDL_CG_LAUNCH_WSP(32);
#undef MDL_CG_LAUNCH_WSP
            return check_launch(name);
        } else {
            set_error("%s: the W-split kernels are bf16 only", name);
            return MDL_E_UNSUPP;
        }
    }
#endif
#ifdef MDL_CG_FAST_ONLY   // compile-time experiments: only the bf16 C=64 G=50 instantiation
    if constexpr (sizeof(T) == 2) {
        if (fast && d.Cp == 64) MDL_CG_LAUNCH(64, 50, 9, 2, MDL_CG_WM);
        else if (fast128 && p.w_slice) MDL_CG_LAUNCH(128, 50, 9, 2, 1);
    }
#else
    if constexpr (sizeof(T) == 2) {
        if (fast && d.Cp == 64) MDL_CG_LAUNCH(64, 50, 9, 2, MDL_CG_WM);
        else if (fast && d.Cp == 32) MDL_CG_LAUNCH(32, 50, 9, 2, MDL_CG_WM);
        else if (fast128 && p.w_slice) MDL_CG_LAUNCH(128, 50, 9, 2, 1);
        else if (vec == 8) MDL_CG_BY_EW(8);
        else if (vec == 4) MDL_CG_BY_EW(4);
        else MDL_CG_BY_EW(1);
    } else {
        if (x3) {
            // fp32 storage, split-bf16 products (MDL_SPLIT_BF16): the static C = 64, G = 50 kernels with the weights in LDS only
            const bool x3_64 = fast && w_lds && d.Cp == 64;
            const bool x3_128 = !p.eperm && p.G == 50 && p.C == 128 && d.Cp == 128 && w_lds && p.w_slice;   // (caller pads C = 100 to 128)
            if (!((x3_64 || x3_128) && p.bias_col && !wsp && !det && reinterpret_cast<uintptr_t>(p.x) % 16 == 0)) {
                set_error("%s: MDL_SPLIT_BF16 needs fp32, C = 64 or 128, G = 50, edge features in CSR order, 16-byte aligned x", name);
                return MDL_E_UNSUPP;
            }
            hipError_t e;
#define MDL_CG_LAUNCH_X3(CP_)                                                                                                 \
            do {                                                                                                              \
                if (bwd) {                                                                                                    \
                    auto kf = cgconv_bwd_kernel<T, CP_, 50, 9, 1, 1, 0, true>;                                                \
                    e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                          \
                    if (e == hipSuccess) hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);          \
                } else {                                                                                                      \
                    auto kf = cgconv_fwd_kernel<T, CP_, 50, 9, 1, 1, false, 0, false, true>;                                  \
                    e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                          \
                    if (e == hipSuccess) hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);          \
                }                                                                                                             \
            } while (0)
            if (x3_64) MDL_CG_LAUNCH_X3(64); else MDL_CG_LAUNCH_X3(128);
#undef MDL_CG_LAUNCH_X3
            if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
        }
        else if (fast && w_lds) MDL_CG_LAUNCH(64, 50, 9, 1, 1);
        else MDL_CG_BY_WL(1, 1);
    }
#endif
#undef MDL_CG_BY_EW
#undef MDL_CG_BY_WL
#undef MDL_CG_LAUNCH
    return check_launch(name);
}

#if MDL_EXPERIMENTS
// saved-gate backward (cgconv_bwd_ab_kernel): bf16, C in {32, 64}, G = 50, target-sorted edge features
static int cg_launch_bwd_ab(CgParams& p, hipStream_t st, const char* name) {
    const CgDims d = cg_dims(p.C, p.G, MDL_BF16);
    p.Cp = d.Cp; p.KE = d.KE; p.KT = d.KT; p.WS = d.WS; p.EKS = d.EKS; p.NS = d.NS; p.GP = d.GP;
    p.w_elems = 0;
    p.bias_col = 1;
    p.n_groups = (int)cdiv(p.N, 32);
    if (p.n_groups == 0) return MDL_OK;
    p.GW = p.G / 2;
    p.gw_inv = (unsigned)((0x100000000ull + p.GW - 1) / p.GW);
    const int et_bytes = (32 * d.EKS * 2 + 15) & ~15;
    p.wave_lds_bytes = et_bytes + 32 + 128 + 32 + 16 + 128 * OHS * 2;
    const int waves = 4;
    const int lds = waves * p.wave_lds_bytes;
    const int64_t ranges = std::max<int64_t>(1, std::min<int64_t>(cdiv(p.E, 64), p.N));
    int64_t grid = cdiv(ranges * d.NS, waves);
    const CgEnv& env = cg_env();
    const int64_t cap = 256 * (env.ab_wgs > 0 ? env.ab_wgs : 1);
    if (grid > cap) grid = cap;
    while ((grid * waves) % d.NS) ++grid;
    if (p.ctr) {
        if ((int64_t)p.n_groups * d.NS >= 4 * grid * waves) {
            if (hipMemsetAsync(p.ctr, 0, 64, st) != hipSuccess) { set_error("%s: workspace memset failed", name); return MDL_E_LAUNCH; }
        } else {
            p.ctr = nullptr;
        }
    }
    if (d.Cp == 64) {
        auto kf = cgconv_bwd_ab_kernel<64, 50>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);
    } else {
        auto kf = cgconv_bwd_ab_kernel<32, 50>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(waves * 64), lds, st, p);
    }
    return check_launch(name);
}
#endif

static int cg_check(const char* name, const void* x, const void* ea, const int32_t* rowptr, const int32_t* src,
                    const int32_t* tgt, const void* wpack, const float* bpack, int64_t N, int64_t E, int C, int G,
                    int aggr, int dtype) {
    MDL_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) - 64 && E < (1ll << 31) - 64, MDL_E_ARG, "%s: bad N=%lld E=%lld", name,
                (long long)N, (long long)E);
    MDL_REQUIRE(C >= 1 && C <= 256, MDL_E_UNSUPP, "%s: unsupported channels C=%d (1..256)", name, C);
    MDL_REQUIRE(G >= 1 && G <= 64, MDL_E_UNSUPP, "%s: unsupported edge feature count G=%d (1..64)", name, G);
    MDL_REQUIRE(dtype == MDL_F32 || dtype == MDL_BF16, MDL_E_UNSUPP, "%s: unsupported dtype %d", name, dtype);
    MDL_REQUIRE(aggr == MDL_MEAN || aggr == MDL_SUM, MDL_E_UNSUPP, "%s: unsupported aggr %d", name, aggr);
    MDL_REQUIRE(N == 0 || (x && rowptr && wpack && bpack), MDL_E_ARG, "%s: null pointer", name);
    MDL_REQUIRE(E == 0 || (ea && src && tgt), MDL_E_ARG, "%s: null edge pointer", name);
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(wpack) % 16 == 0, MDL_E_ARG, "%s: wpack must be 16-byte aligned", name);
    return MDL_OK;
}

}  // namespace mdl

// x3 (MDL_SPLIT_BF16 OR-ed into dtype, fp32 only): the static shape the split-product kernels exist for
// (C in (96, 128]: the static 128-channel kernels on zero-padded rows, one channel slice of W per workgroup — the reference's
// default width 100, config.yml:123)
static bool cg_x3_ok(int C, int G, int dtype) { return dtype == MDL_F32 && (C == 64 || (C > 96 && C <= 128)) && G == 50; }

extern "C" size_t mdl_cgconv_wpack_bytes(int C, int G, int dtype) {
    using namespace mdl;
    const bool x3 = (dtype & MDL_SPLIT_BF16) != 0;
    dtype &= ~MDL_SPLIT_BF16;
    if (C < 1 || G < 1 || (dtype != MDL_F32 && dtype != MDL_BF16) || (x3 && !cg_x3_ok(C, G, dtype))) return 0;
    const CgDims d = cg_dims(C, G, dtype, x3);
    const size_t b = (size_t)2 * d.Cp * d.WS * (dtype == MDL_BF16 ? 2 : 4);
    return (b + 15) & ~(size_t)15;
}

static int cg_pack_weights(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C, int G, void* wpack,
                           float* bpack, void* wn_t, int dtype, mdlStream_t stream);
extern "C" int mdl_cgconv_pack_weights(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C,
                                       int G, void* wpack, float* bpack, int dtype, mdlStream_t stream) {
    return cg_pack_weights(w_f, b_f, w_s, b_s, C, G, wpack, bpack, nullptr, dtype, stream);
}
extern "C" int mdl_cgconv_pack_weights_node(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C,
                                            int G, void* wpack, float* bpack, void* wn_t, int dtype, mdlStream_t stream) {
    MDL_REQUIRE(wn_t && dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_pack_weights_node: bf16 only, wn_t required");
    return cg_pack_weights(w_f, b_f, w_s, b_s, C, G, wpack, bpack, wn_t, dtype, stream);
}
static int cg_pack_weights(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C, int G, void* wpack,
                           float* bpack, void* wn_t, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(w_f && w_s && wpack && bpack, MDL_E_ARG, "mdl_cgconv_pack_weights: null pointer");
    MDL_REQUIRE(C >= 1 && C <= 256 && G >= 1 && G <= 64, MDL_E_UNSUPP, "mdl_cgconv_pack_weights: unsupported C=%d G=%d", C, G);
    const bool x3 = (dtype & MDL_SPLIT_BF16) != 0;
    dtype &= ~MDL_SPLIT_BF16;
    MDL_REQUIRE(!x3 || cg_x3_ok(C, G, dtype), MDL_E_UNSUPP, "mdl_cgconv_pack_weights: MDL_SPLIT_BF16 needs fp32, C = 64, G = 50 (C=%d G=%d dtype=%d)", C, G, dtype);
    const CgDims d = cg_dims(C, G, dtype, x3);
    const int total = 2 * d.Cp * d.WS;
    dim3 grid((unsigned)cdiv(total, 256)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int bias_col = (G % 16) != 0;
    if (dtype == MDL_BF16)
        hipLaunchKernelGGL((cgconv_pack_kernel<bf16_t>), grid, block, 0, st, w_f, b_f, w_s, b_s, C, G, d, (bf16_t*)wpack, bpack,
                           Gate<true>::W_SCALE, bias_col, (bf16_t*)wn_t);
    else if (dtype == MDL_F32)
        hipLaunchKernelGGL((cgconv_pack_kernel<float>), grid, block, 0, st, w_f, b_f, w_s, b_s, C, G, d, (float*)wpack, bpack,
                           x3 ? Gate<true>::W_SCALE : Gate<false>::W_SCALE, bias_col, (bf16_t*)nullptr, x3 ? 1 : 0);
    else {
        set_error("mdl_cgconv_pack_weights: unsupported dtype %d", dtype);
        return MDL_E_UNSUPP;
    }
    return check_launch("mdl_cgconv_pack_weights");
}

extern "C" int mdl_cgconv_pack_weights_multi(int L, const float* const* w_f, const float* const* b_f, const float* const* w_s,
                                             const float* const* b_s, int C, int G, void* const* wpack, float* const* bpack,
                                             void* const* wn_t, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(L >= 1 && L <= PACK_MAXL && w_f && w_s && wpack && bpack, MDL_E_ARG, "mdl_cgconv_pack_weights_multi: 1..%d layers, non-null tables", PACK_MAXL);
    MDL_REQUIRE(C >= 1 && C <= 256 && G >= 1 && G <= 64, MDL_E_UNSUPP, "mdl_cgconv_pack_weights_multi: unsupported C=%d G=%d", C, G);
    const bool x3 = (dtype & MDL_SPLIT_BF16) != 0;
    dtype &= ~MDL_SPLIT_BF16;
    MDL_REQUIRE(dtype == MDL_BF16 || dtype == MDL_F32, MDL_E_UNSUPP, "mdl_cgconv_pack_weights_multi: unsupported dtype %d", dtype);
    MDL_REQUIRE(!x3 || cg_x3_ok(C, G, dtype), MDL_E_UNSUPP, "mdl_cgconv_pack_weights_multi: MDL_SPLIT_BF16 needs fp32, C = 64, G = 50");
    MDL_REQUIRE(!wn_t || dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_pack_weights_multi: wn_t is bf16 only");
    PackMulti a = {};
    for (int l = 0; l < L; ++l) {
        MDL_REQUIRE(w_f[l] && w_s[l] && wpack[l] && bpack[l], MDL_E_ARG, "mdl_cgconv_pack_weights_multi: null pointer (layer %d)", l);
        a.wf[l] = w_f[l]; a.bf[l] = b_f ? b_f[l] : nullptr; a.ws[l] = w_s[l]; a.bs[l] = b_s ? b_s[l] : nullptr;
        a.wpack[l] = wpack[l]; a.bpack[l] = bpack[l]; a.wn_t[l] = wn_t ? static_cast<bf16_t*>(wn_t[l]) : nullptr;
    }
    const CgDims d = cg_dims(C, G, dtype, x3);
    dim3 grid((unsigned)cdiv(2 * d.Cp * d.WS, 256), (unsigned)L), block(256);
    const int bias_col = (G % 16) != 0;
    if (dtype == MDL_BF16) hipLaunchKernelGGL((cgconv_pack_multi_kernel<bf16_t>), grid, block, 0, (hipStream_t)stream, a, C, G, d, Gate<true>::W_SCALE, bias_col);
    else hipLaunchKernelGGL((cgconv_pack_multi_kernel<float>), grid, block, 0, (hipStream_t)stream, a, C, G, d, x3 ? Gate<true>::W_SCALE : Gate<false>::W_SCALE, bias_col, x3 ? 1 : 0);
    return check_launch("mdl_cgconv_pack_weights_multi");
}

extern "C" int mdl_cgconv_fwd(const void* x, const void* edge_attr, const int32_t* rowptr, const int32_t* src,
                              const int32_t* tgt, const int32_t* eperm, const void* wpack, const float* bpack,
                              void* out, int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = cg_check("mdl_cgconv_fwd", x, edge_attr, rowptr, src, tgt, wpack, bpack, N, E, C, G, aggr, dtype);
    if (rc) return rc;
    MDL_REQUIRE(N == 0 || out, MDL_E_ARG, "mdl_cgconv_fwd: null out");
    CgParams p = {};
    p.x = x; p.ea = edge_attr; p.rowptr = rowptr; p.src = src; p.tgt = tgt; p.eperm = eperm;
    p.wpack = wpack; p.bpack = bpack; p.out = out; p.N = N; p.E = E; p.C = C; p.G = G; p.aggr = aggr;
    if (dtype == MDL_BF16) return cg_launch<bf16_t>(false, p, dtype, (hipStream_t)stream, "mdl_cgconv_fwd");
    return cg_launch<float>(false, p, dtype, (hipStream_t)stream, "mdl_cgconv_fwd");
}

static int cg_fwd_stats_ok(int C, int G, int dtype) {      // the static all-slices forward: the kernel that has the epilogue
    return (dtype == MDL_BF16 && G == 50 && (C == 32 || C == 64) && MDL_FWD_ALLSLICES && MDL_CG_WM == 1) ? 1 : 0;
}

extern "C" int mdl_cgconv_fwd_ex(const MdlCgConv* a, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(a && a->size == sizeof(MdlCgConv), MDL_E_ARG, "mdl_cgconv_fwd_ex: argument struct of another layout (size %u, expected %u)",
                a ? a->size : 0u, (unsigned)sizeof(MdlCgConv));
    MDL_REQUIRE((a->flags & ~(uint32_t)(MDL_DETERMINISTIC | MDL_SPLIT_BF16)) == 0, MDL_E_ARG, "mdl_cgconv_fwd_ex: unknown flag bits %#x", a->flags);
    MDL_REQUIRE(!(a->flags & MDL_SPLIT_BF16) || (a->dtype == MDL_F32 && !a->bn_sums), MDL_E_UNSUPP, "mdl_cgconv_fwd_ex: MDL_SPLIT_BF16 goes with MDL_F32 storage");
    int rc = cg_check("mdl_cgconv_fwd_ex", a->x, a->edge_attr, a->rowptr, a->src, a->tgt, a->wpack, a->bpack, a->N, a->E, a->C, a->G,
                      a->aggr, a->dtype);
    if (rc) return rc;
    MDL_REQUIRE(a->N == 0 || a->out, MDL_E_ARG, "mdl_cgconv_fwd_ex: null out");
    CgParams p = {};
    p.x = a->x; p.ea = a->edge_attr; p.rowptr = a->rowptr; p.src = a->src; p.tgt = a->tgt; p.eperm = a->eperm;
    p.wpack = a->wpack; p.bpack = a->bpack; p.out = a->out; p.N = a->N; p.E = a->E; p.C = a->C; p.G = a->G; p.aggr = a->aggr;
    p.flags = (int)(a->flags & MDL_SPLIT_BF16);
    if (a->bn_sums) {
        // BatchNorm statistics in the epilogue: the static all-slices kernel only (bf16, C in {32, 64}, G = 50, CSR-ordered edge features)
        MDL_REQUIRE(cg_fwd_stats_ok(a->C, a->G, a->dtype) && !a->eperm && a->E > 0 &&
                        reinterpret_cast<uintptr_t>(a->x) % 16 == 0 && reinterpret_cast<uintptr_t>(a->out) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(a->edge_attr) % 4 == 0,
                    MDL_E_UNSUPP, "mdl_cgconv_fwd_ex: BatchNorm statistics need bf16, C in {32, 64}, G = 50, edge features in CSR order, "
                                  "16-byte aligned x / out (C=%d G=%d dtype=%d)", a->C, a->G, a->dtype);
        p.bn_sums = a->bn_sums; p.bn_shift = a->bn_shift; p.bn_nrows = a->bn_rows;
    }
    if (a->dtype == MDL_BF16) return cg_launch<bf16_t>(false, p, a->dtype, (hipStream_t)stream, "mdl_cgconv_fwd_ex");
    return cg_launch<float>(false, p, a->dtype, (hipStream_t)stream, "mdl_cgconv_fwd_ex");
}

extern "C" size_t mdl_cgconv_workspace_bytes(int64_t, int64_t, int, int, int) { return 64; }

#if MDL_EXPERIMENTS   // saved-gate pair, W-split pair (declared in experiments/mdl_hip_experiments.h)
extern "C" size_t mdl_cgconv_gate_row_bytes(int C, int G, int dtype) {
    return (dtype == MDL_BF16 && G == 50 && (C == 32 || C == 64)) ? (size_t)4 * C : 0;
}

extern "C" int mdl_cgconv_fwd_save(const void* x, const void* edge_attr, const int32_t* rowptr, const int32_t* src,
                                   const int32_t* tgt, const void* wpack, const float* bpack, void* out, void* gate,
                                   int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = cg_check("mdl_cgconv_fwd_save", x, edge_attr, rowptr, src, tgt, wpack, bpack, N, E, C, G, aggr, dtype);
    if (rc) return rc;
    MDL_REQUIRE(mdl_cgconv_gate_row_bytes(C, G, dtype) != 0, MDL_E_UNSUPP,
                "mdl_cgconv_fwd_save: unsupported C=%d G=%d dtype=%d (bf16, C in {32, 64}, G = 50)", C, G, dtype);
    MDL_REQUIRE(N == 0 || out, MDL_E_ARG, "mdl_cgconv_fwd_save: null out");
    MDL_REQUIRE(E == 0 || (gate && reinterpret_cast<uintptr_t>(gate) % 4 == 0), MDL_E_ARG, "mdl_cgconv_fwd_save: bad gate buffer");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(edge_attr) % 4 == 0, MDL_E_ARG,
                "mdl_cgconv_fwd_save: x must be 16-byte and edge_attr 4-byte aligned");
    CgParams p = {};
    p.x = x; p.ea = edge_attr; p.rowptr = rowptr; p.src = src; p.tgt = tgt; p.eperm = nullptr;
    p.wpack = wpack; p.bpack = bpack; p.out = out; p.ab = E ? gate : nullptr; p.N = N; p.E = E; p.C = C; p.G = G; p.aggr = aggr;
    return cg_launch<bf16_t>(false, p, dtype, (hipStream_t)stream, "mdl_cgconv_fwd_save");
}

extern "C" int mdl_cgconv_bwd_saved(const void* edge_attr, const int32_t* rowptr, const int32_t* src, const int32_t* tgt,
                                    const void* gate, const void* grad_out, void* r_tgt, float* r_src, float* dwe, float* db,
                                    int64_t N, int64_t E, int C, int G, int aggr, int dtype, void* workspace, size_t ws_bytes,
                                    mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(mdl_cgconv_gate_row_bytes(C, G, dtype) != 0, MDL_E_UNSUPP,
                "mdl_cgconv_bwd_saved: unsupported C=%d G=%d dtype=%d (bf16, C in {32, 64}, G = 50)", C, G, dtype);
    MDL_REQUIRE(N >= 0 && E >= 0 && N < (1ll << 31) - 64 && E < (1ll << 31) - 64, MDL_E_ARG, "mdl_cgconv_bwd_saved: bad N=%lld E=%lld",
                (long long)N, (long long)E);
    MDL_REQUIRE(aggr == MDL_MEAN || aggr == MDL_SUM, MDL_E_UNSUPP, "mdl_cgconv_bwd_saved: unsupported aggr %d", aggr);
    MDL_REQUIRE(N == 0 || (rowptr && grad_out && r_tgt && r_src && dwe), MDL_E_ARG, "mdl_cgconv_bwd_saved: null pointer");
    MDL_REQUIRE(E == 0 || (edge_attr && src && tgt && gate), MDL_E_ARG, "mdl_cgconv_bwd_saved: null edge pointer");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(gate) % 4 == 0 && reinterpret_cast<uintptr_t>(edge_attr) % 4 == 0, MDL_E_ARG,
                "mdl_cgconv_bwd_saved: gate / edge_attr must be 4-byte aligned");
    CgParams p = {};
    p.ea = edge_attr; p.rowptr = rowptr; p.src = src; p.tgt = tgt; p.ab = const_cast<void*>(gate); p.gout = grad_out;
    p.r_tgt = r_tgt; p.r_src = r_src; p.dwe = dwe; p.db = db; p.N = N; p.E = E; p.C = C; p.G = G; p.aggr = aggr;
    p.ctr = (workspace && ws_bytes >= 64) ? static_cast<unsigned*>(workspace) : nullptr;
    return cg_launch_bwd_ab(p, (hipStream_t)stream, "mdl_cgconv_bwd_saved");
}

extern "C" size_t mdl_cgconv_wsplit_bytes(int C, int G, int dtype, int which) {
    using namespace mdl;
    if (dtype != MDL_BF16 || G != 50 || (C != 32 && C != 64)) return 0;
    const CgDims d = cg_dims(C, G, dtype);
    const size_t b = which == 0 ? (size_t)2 * d.Cp * d.EKS * 2 : (size_t)2 * 2 * d.Cp * d.Cp * 2;
    return (b + 15) & ~(size_t)15;
}

extern "C" int mdl_cgconv_pack_weights_split(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C,
                                             int G, void* wpack_e, void* wproj, float* bpack, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(w_f && w_s && wpack_e && wproj && bpack, MDL_E_ARG, "mdl_cgconv_pack_weights_split: null pointer");
    MDL_REQUIRE(mdl_cgconv_wsplit_bytes(C, G, dtype, 0) != 0, MDL_E_UNSUPP,
                "mdl_cgconv_pack_weights_split: unsupported C=%d G=%d dtype=%d (bf16, C in {32, 64}, G = 50)", C, G, dtype);
    const CgDims d = cg_dims(C, G, dtype);
    const int total = 2 * d.Cp * d.EKS + 4 * d.Cp * d.Cp;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cgconv_pack_split_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, w_f, b_f, w_s, b_s, C, G, d.Cp,
                       d.KE, d.EKS, (bf16_t*)wpack_e, (bf16_t*)wproj, Gate<true>::W_SCALE);
    if (hipMemsetAsync(bpack, 0, 2 * d.Cp * sizeof(float), st) != hipSuccess) { set_error("mdl_cgconv_pack_weights_split: memset failed"); return MDL_E_LAUNCH; }
    return check_launch("mdl_cgconv_pack_weights_split");
}

extern "C" int mdl_cgconv_fwd_p(const void* x, const void* p_tgt, const void* p_src, const void* edge_attr, const int32_t* rowptr,
                                const int32_t* src, const int32_t* tgt, const void* wpack_e, const float* bpack, void* out,
                                int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = cg_check("mdl_cgconv_fwd_p", x, edge_attr, rowptr, src, tgt, wpack_e, bpack, N, E, C, G, aggr, dtype);
    if (rc) return rc;
    MDL_REQUIRE(N == 0 || (out && p_tgt && p_src), MDL_E_ARG, "mdl_cgconv_fwd_p: null pointer");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(p_tgt) % 16 == 0 && reinterpret_cast<uintptr_t>(p_src) % 16 == 0, MDL_E_ARG,
                "mdl_cgconv_fwd_p: the projections must be 16-byte aligned");
    CgParams p = {};
    p.x = x; p.pt = p_tgt; p.ps = p_src; p.ea = edge_attr; p.rowptr = rowptr; p.src = src; p.tgt = tgt; p.eperm = nullptr;
    p.wpack = wpack_e; p.bpack = bpack; p.out = out; p.N = N; p.E = E; p.C = C; p.G = G; p.aggr = aggr;
    if (dtype != MDL_BF16) { set_error("mdl_cgconv_fwd_p: bf16 only"); return MDL_E_UNSUPP; }
    return cg_launch<bf16_t>(false, p, dtype, (hipStream_t)stream, "mdl_cgconv_fwd_p");
}

extern "C" int mdl_cgconv_bwd_p(const void* p_tgt, const void* p_src, const void* edge_attr, const int32_t* rowptr,
                                const int32_t* src, const int32_t* tgt, const void* wpack_e, const float* bpack,
                                const void* grad_out, void* r_tgt, float* r_src, float* dwe, float* db, int64_t N, int64_t E,
                                int C, int G, int aggr, int dtype, void* workspace, size_t ws_bytes, mdlStream_t stream) {
    using namespace mdl;
    int rc = cg_check("mdl_cgconv_bwd_p", p_tgt, edge_attr, rowptr, src, tgt, wpack_e, bpack, N, E, C, G, aggr, dtype);
    if (rc) return rc;
    MDL_REQUIRE(N == 0 || (grad_out && r_tgt && r_src && dwe && p_src), MDL_E_ARG, "mdl_cgconv_bwd_p: null pointer");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(p_tgt) % 16 == 0 && reinterpret_cast<uintptr_t>(p_src) % 16 == 0, MDL_E_ARG,
                "mdl_cgconv_bwd_p: the projections must be 16-byte aligned");
    CgParams p = {};
    p.x = p_tgt; p.pt = p_tgt; p.ps = p_src; p.ea = edge_attr; p.rowptr = rowptr; p.src = src; p.tgt = tgt; p.eperm = nullptr;
    p.wpack = wpack_e; p.bpack = bpack; p.gout = grad_out; p.r_tgt = r_tgt; p.r_src = r_src; p.dwe = dwe; p.db = db;
    p.N = N; p.E = E; p.C = C; p.G = G; p.aggr = aggr;
    p.ctr = (workspace && ws_bytes >= mdl_cgconv_workspace_bytes(N, E, C, G, dtype)) ? static_cast<unsigned*>(workspace) : nullptr;
    if (dtype != MDL_BF16) { set_error("mdl_cgconv_bwd_p: bf16 only"); return MDL_E_UNSUPP; }
    return cg_launch<bf16_t>(true, p, dtype, (hipStream_t)stream, "mdl_cgconv_bwd_p");
}
#endif

namespace mdl {
// per-node cost of the backward edge pass in quarter units: 4 per edge and node, + 5 per edge whose source lies 48 or more
// rows away from its target (such edges land in far blocks of the by-source window or outside it: +0.8 of an edge's time in the fit,
// fitted on per-workgroup end times of the bench batch — DESIGN.md section 4)
__global__ __launch_bounds__(256) void cgconv_balance_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ src,
                                                             int64_t N, int32_t* __restrict__ cost, int far_w, int far_t, int zero_w) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n == 0) cost[0] = 0;
    if (n >= N) return;
    const int b = rowptr[n], e = rowptr[n + 1];
    int far = 0;
    for (int k = b; k < e; ++k) {
        const int d = src[k] - (int)n;
        far += (d >= far_t || d <= -far_t) ? 1 : 0;
    }
    // (an edge-less node — the padding rows of a static batch come 32 to an empty tile that costs a full tile's time — is charged
    // like the share of an average tile: without it the workgroup that holds the padding gets a fifth more rounds than the others)
    cost[n + 1] = 4 * (e - b + 1) + far_w * far + (e == b ? zero_w : 0);
}
}  // namespace mdl

extern "C" int mdl_cgconv_balance(const int32_t* rowptr, const int32_t* src, int64_t N, int32_t* cost, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && (N == 0 || (rowptr && src && cost)), MDL_E_ARG, "mdl_cgconv_balance: bad arguments");
#if MDL_EXPERIMENTS
    // (MDL_BAL_W / MDL_BAL_T: weight in quarter units and distance threshold of a far edge — tools/fit_balance.py)
    static const int far_w = [] { const char* s = getenv("MDL_BAL_W"); return s ? atoi(s) : 5; }();
    static const int far_t = [] { const char* s = getenv("MDL_BAL_T"); return s ? atoi(s) : 48; }();
    static const int zero_w = [] { const char* s = getenv("MDL_BAL_Z"); return s ? atoi(s) : 4; }();
#else
    constexpr int far_w = 5, far_t = 48, zero_w = 4;
#endif
    hipLaunchKernelGGL(cgconv_balance_kernel, dim3((unsigned)cdiv(N + 1, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, src, N, cost,
                       far_w, far_t, zero_w);
    return check_launch("mdl_cgconv_balance");
}

extern "C" int mdl_cgconv_bwd_ex(const MdlCgConv* a, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(a && a->size == sizeof(MdlCgConv), MDL_E_ARG, "mdl_cgconv_bwd_ex: argument struct of another layout (size %u, expected %u)",
                a ? a->size : 0u, (unsigned)sizeof(MdlCgConv));
    MDL_REQUIRE((a->flags & ~(uint32_t)(MDL_DETERMINISTIC | MDL_K3_PER_WAVE | MDL_K3_EDGE_LANE | MDL_SPLIT_BF16)) == 0, MDL_E_ARG,
                "mdl_cgconv_bwd_ex: unknown flag bits %#x", a->flags);
    MDL_REQUIRE(!(a->flags & MDL_SPLIT_BF16) || (a->dtype == MDL_F32 && !(a->flags & MDL_DETERMINISTIC)), MDL_E_UNSUPP,
                "mdl_cgconv_bwd_ex: MDL_SPLIT_BF16 goes with MDL_F32 storage (and not with MDL_DETERMINISTIC)");
    const int dtype = a->dtype;
    int rc = cg_check("mdl_cgconv_bwd_ex", a->x, a->edge_attr, a->rowptr, a->src, a->tgt, a->wpack, a->bpack, a->N, a->E, a->C, a->G,
                      a->aggr, dtype);
    if (rc) return rc;
    MDL_REQUIRE(a->N == 0 || (a->grad_out && a->r_tgt && a->r_src && a->dwe), MDL_E_ARG, "mdl_cgconv_bwd_ex: null pointer");
    MDL_REQUIRE(a->r_src_dtype == MDL_F32 || a->r_src_dtype == MDL_BF16, MDL_E_ARG, "mdl_cgconv_bwd_ex: r_src_dtype must be MDL_F32 or MDL_BF16");
    CgParams p = {};
    p.flags = (int)a->flags;
    p.x = a->x; p.ea = a->edge_attr; p.rowptr = a->rowptr; p.src = a->src; p.tgt = a->tgt; p.eperm = a->eperm;
    p.wpack = a->wpack; p.bpack = a->bpack; p.gout = a->grad_out; p.r_tgt = a->r_tgt; p.r_src = static_cast<float*>(a->r_src);
    p.dwe = a->dwe; p.db = a->db;
    p.N = a->N; p.E = a->E; p.C = a->C; p.G = a->G; p.aggr = a->aggr;
    MDL_REQUIRE(a->ld_dwe == 0 || a->ld_dwe >= a->G, MDL_E_ARG, "mdl_cgconv_bwd_ex: ld_dwe (%d) below G", a->ld_dwe);
    p.ldwe = a->ld_dwe;
    // optional caller workspace: work counters for dynamic group scheduling (zeroed here, on the stream)
    p.ctr = (a->workspace && a->workspace_bytes >= mdl_cgconv_workspace_bytes(a->N, a->E, a->C, a->G, dtype))
                ? static_cast<unsigned*>(a->workspace) : nullptr;
    if (a->r_src_dtype == MDL_BF16) {
        // by-source sums in bf16 (packed bf16 atomics): the static bf16 kernels, edge features in CSR order
        MDL_REQUIRE(dtype == MDL_BF16 && a->G == 50 && (a->C == 32 || a->C == 64 || a->C == 128) && !a->eperm, MDL_E_UNSUPP,
                    "mdl_cgconv_bwd_ex: bf16 by-source sums need bf16, C in {32, 64, 128}, G = 50, no eperm (C=%d G=%d dtype=%d)",
                    a->C, a->G, dtype);
        MDL_REQUIRE(reinterpret_cast<uintptr_t>(a->x) % 16 == 0 && reinterpret_cast<uintptr_t>(a->edge_attr) % 4 == 0 &&
                        reinterpret_cast<uintptr_t>(a->r_src) % 4 == 0, MDL_E_ARG,
                    "mdl_cgconv_bwd_ex: x must be 16-byte, edge_attr / r_src 4-byte aligned");
        p.rs16 = 1;
        p.balance = a->balance;
    } else {
        MDL_REQUIRE(!a->balance && (a->flags & (MDL_K3_PER_WAVE | MDL_K3_EDGE_LANE)) == 0, MDL_E_ARG,
                    "mdl_cgconv_bwd_ex: the balance prefix and the MDL_K3_* flags go with bf16 by-source sums");
    }
    if (dtype == MDL_BF16) return cg_launch<bf16_t>(true, p, dtype, (hipStream_t)stream, "mdl_cgconv_bwd_ex");
    return cg_launch<float>(true, p, dtype, (hipStream_t)stream, "mdl_cgconv_bwd_ex");
}

extern "C" int mdl_debug_last_k3(void) { return mdl::g_last_k3; }
#endif   // !MDL_CG_EP_TU
