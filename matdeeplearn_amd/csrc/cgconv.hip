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
// Layout of the sources (round 6): cgconv_tiles.inc (shared machinery), cgconv_fwd.inc (K2), cgconv_bwd.inc (per-wave K3), this file
// (weight packing, launch logic, C ABI); cgconv_ep.hip = cgconv_tiles.inc + cgconv_ep_common.inc + cgconv_ep2.inc (edge-per-lane K3).
#include "cgconv_tiles.inc"
#include "cgconv_fwd.inc"
#include "cgconv_bwd.inc"

namespace mdl {
#if MDL_EXPERIMENTS
#include "../../experiments/csrc/cgconv_ab.inc"   // saved-gate backward edge pass: net zero against the recomputing pass
#include "../../experiments/csrc/cgconv_cb.inc"   // namespace mdl::cb (cooperative weight-stationary kernels: measured slower)
#endif
// The edge-per-lane backward (cgconv_ep2.inc, namespace mdl::ep) lives in its own translation unit, cgconv_ep.hip, compiled with
// -mllvm -amdgpu-mfma-vgpr-form=1 — its phase-A MFMA results then land in the VGPRs the gate arithmetic reads (the default AGPR
// form costs one v_accvgpr_read per value, 96 per tile), while the per-wave kernels of this unit, which live on 256 + 179
// registers, need the AGPR form.
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
