/* mdl_hip.h — C ABI of libmdl_hip.so, the MI355X (gfx950) message-passing engine for
 * MatDeepLearn-style crystal-graph networks.
 *
 * The reference (Fung-Lab/MatDeepLearn) is pure Python and has NO FFI of its own: every kernel on
 * its hot path lives in third-party wheels (torch_scatter, torch_geometric on ATen).  Each entry
 * point below therefore cites the reference call site (paths relative to /root/reference) whose
 * third-party operator it replaces.  The binding a maintainer adds on the reference side is a
 * ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless marked "host".  The caller owns every buffer,
 *     including scratch/outputs; the library never allocates, frees or synchronises.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - Return value: 0 = ok, negative = error (MDL_E_*); mdl_last_error_string() gives the text
 *     (thread-local).  No exceptions cross the boundary.
 *   - dtype: MDL_F32 (parity mode, exact-fp32 MFMA / VALU) or MDL_BF16 (bf16 storage, fp32
 *     accumulation).  Index arrays are int32.
 *   - Graph layout: edges are given in CSR order BY TARGET node:
 *       rowptr[N+1]  first CSR slot of every target node
 *       src[E]       source node of CSR slot k           (edge_index[0] permuted)
 *       tgt[E]       target node of CSR slot k           (edge_index[1] permuted, non-decreasing)
 *       eperm[E]     original edge id of CSR slot k, or NULL when the caller's per-edge arrays
 *                    (edge_attr, ...) are already stored in CSR order.
 *   - Stateless and re-entrant; safe to call from one thread per device/stream.
 */
#ifndef MDL_HIP_H
#define MDL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDL_VERSION 100 /* 0.1.0 */

enum { MDL_F32 = 0, MDL_BF16 = 1 };
enum { MDL_SUM = 0, MDL_MEAN = 1, MDL_MAX = 2 };
enum {
    MDL_OK = 0,
    MDL_E_ARG = -1,     /* bad shape / null pointer / misaligned buffer */
    MDL_E_UNSUPP = -2,  /* unsupported C / G / dtype combination        */
    MDL_E_LAUNCH = -3   /* hipGetLastError() after a launch             */
};

/* Execution flags (for the positional entry points: OR-ed into the `dtype` argument of those that say so).  The library keeps no mode state and
 * reads no environment variable; what a caller (or a test) wants, it passes with the call.
 *   MDL_DETERMINISTIC  run-to-run bit-reproducible results.  By default the gradient reductions combine per-workgroup
 *                      partial sums with floating-point atomics (r_src / dwe / db of the CGConv backward, dWn, the TN
 *                      GEMM, the BatchNorm sums, the fused head): the fastest form, but the order of the adds, hence the
 *                      last bits, varies between runs.  With this flag such a kernel is launched in a shape in which every
 *                      sum receives its terms from ONE wave in program order (one workgroup for the streaming kernels,
 *                      one wave per channel slice for the CGConv edge pass).  A few hundred times slower: meant for
 *                      HIP-vs-HIP regression checks (graph replay vs eager, padded rows, data-parallel exchange).
 *   MDL_K3_PER_WAVE /  MdlCgConv.flags of mdl_cgconv_bwd_ex with bf16 by-source sums: force the per-wave kernel / the
 *   MDL_K3_EDGE_LANE   edge-per-lane kernel 2 instead of the edge-count heuristic (kernel 2 from 4e5 edges).
 *   MDL_BN_SHIFT_ROW   mdl_bn_apply_n only: the sums were formed by the PRODUCER of the rows (mdl_linear_act_stats,
 *                      mdl_cgconv_fwd_ex) about the per-column shift it stored behind the totals rows of the sums buffer
 *                      (row 2 MDL_BN_REPLICAS + 2), instead of about the first row as mdl_bn_stats forms them.
 *   MDL_SPLIT_BF16     CGConv kernels with MDL_F32 storage (MdlCgConv.flags of mdl_cgconv_fwd_ex / mdl_cgconv_bwd_ex; OR-ed
 *                      into `dtype` for mdl_cgconv_wpack_bytes / mdl_cgconv_pack_weights[_multi], whose packed layout it
 *                      changes): the K = 2C + G product z W^T runs as THREE bf16 MFMAs on operands split into (hi, lo) bf16
 *                      pairs (hi hi + lo hi + hi lo: operands to 16 significant bits, fp32 accumulation) instead of the
 *                      exact-fp32 MFMA, which has 1/16 of the bf16 rate.  Relative error 2^-16 per product: a PARITY
 *                      mode between bf16 (2^-9) and exact fp32 — the reference computes fp32 throughout
 *                      (training.py:34-54).  C = 64, G = 50, edge features in CSR order only.
 *   MDL_MLP_F32_IO     mdl_mlp_head_fwd / _bwd only: the head's LAST output y (fwd) and its gradient gy (bwd) are fp32 rows —
 *                      the model's `out.float()` (the reference's fp32 prediction, cgcnn.py:169-174) and the cast of the
 *                      loss gradient back to bf16 cost a launch each, a tenth of the glue of a batch-100 step.  y holds the
 *                      bf16-rounded values the bf16 output would have held.
 * The struct entry points (MdlCgConv, MdlCgNode) carry their flags in a field of their own; the positional entry points of
 * the dense / BatchNorm kernels take them OR-ed into `dtype`. */
#define MDL_DTYPE_MASK 0xff
#define MDL_DETERMINISTIC 0x100
#define MDL_K3_PER_WAVE 0x200
#define MDL_K3_EDGE_LANE 0x400
#define MDL_BN_SHIFT_ROW 0x800
#define MDL_SPLIT_BF16 0x1000
#define MDL_MLP_F32_IO 0x2000

typedef void* mdlStream_t; /* hipStream_t */

int mdl_version(void);
const char* mdl_last_error_string(void);
/* Which backward edge pass the last mdl_cgconv_bwd* call of this process launched (a debug value outside the data path; the
 * callers' autograd engines run backward passes on threads of their own, so it is per process, not per thread):
 * 0 none yet, 1 per-wave kernel, 2 edge-per-lane kernel 2, 3 per-wave kernel in its deterministic shape.  For tests that must
 * know that the kernel they mean to check is the one that ran. */
int mdl_debug_last_k3(void);

/* ---- K1: Gaussian RBF edge expansion ------------------------------------------------------
 * Replaces GaussianSmearing.forward, matdeeplearn/process/process.py:588-590 (instantiated
 * (0, 1, graph_edge_length, 0.2) at :500-502, applied per graph at :506-509):
 *     out[e, k] = exp(coeff * (d[e] - offsets[k])^2)
 * d: [E] fp32 (normalised distance, process.py:647-653); offsets: [G] fp32 (the linspace buffer);
 * out: [E, G] row-major with leading dimension ld_out (>= G) in `out_dtype`. */
int mdl_rbf_expand(const float* d, const float* offsets, float coeff, void* out, int64_t E, int G,
                   int64_t ld_out, int out_dtype, mdlStream_t stream);

/* ---- CSR helpers ---------------------------------------------------------------------------
 * rowptr[n] = lower_bound(sorted_index, n) for n in [0, N]; sorted_index: [E] int32 non-decreasing.
 * Stands in for the implicit index handling inside torch_scatter / PyG propagate. */
int mdl_csr_rowptr(const int32_t* sorted_index, int64_t E, int64_t N, int32_t* rowptr, mdlStream_t stream);

/* ---- K5: segmented reduce (scatter with a sorted index) -------------------------------------
 * Replaces torch_scatter.scatter / scatter_mean as called at matdeeplearn/models/megnet.py:86,
 * 130-132,342-348 and inside PyG global_{mean,add,max}_pool (matdeeplearn/models/cgcnn.py:154):
 *     out[n, :] = reduce_{k in [rowptr[n], rowptr[n+1])} src[perm ? perm[k] : k, :]
 * mean divides by max(count,1); empty segments give 0 (also for max).  src: [E, C]; out: [N, C];
 * argmax: [N, C] int32 (row index into src, -1 for empty; required for MDL_MAX, else may be NULL). */
int mdl_segment_reduce_fwd(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                           int32_t* argmax, int64_t N, int64_t C, int reduce, int dtype, mdlStream_t stream);
/* grad_src[perm?perm[k]:k, :] = grad_out[seg[k], :] (/count for mean); for MDL_MAX grad goes to the
 * argmax row only (grad_src must be zero-filled by the caller).  seg: [E] int32 segment id of slot k. */
int mdl_segment_reduce_bwd(const void* grad_out, const int32_t* rowptr, const int32_t* seg,
                           const int32_t* perm, const int32_t* argmax, void* grad_src, int64_t N,
                           int64_t E, int64_t C, int reduce, int dtype, mdlStream_t stream);
/* same for MDL_SUM / MDL_MEAN with a second gradient of the same source rows added on the way out:
 * grad_src[r, :] = (scattered grad_out)[r, :] + addend[r, :] — the accumulation autograd forms with one more pass over
 * [E, C] when the reduced tensor also feeds a residual (matdeeplearn/models/megnet.py:86 with :321-336).  Rows must be a
 * multiple of 4 elements (bf16) and 8-byte aligned. */
int mdl_segment_reduce_bwd_add(const void* grad_out, const int32_t* rowptr, const int32_t* seg, const int32_t* perm,
                               const void* addend, void* grad_src, int64_t N, int64_t E, int64_t C, int reduce, int dtype,
                               mdlStream_t stream);

/* ---- K2/K3: fused CGConv ------------------------------------------------------------------
 * Replaces torch_geometric.nn.CGConv(channels=C, dim=G, aggr, batch_norm=False) as constructed at
 * matdeeplearn/models/cgcnn.py:80-83 and called at cgcnn.py:136-145:
 *     z_k   = [ x[tgt_k] | x[src_k] | edge_attr_k ]                      (2C+G)
 *     m_k   = sigmoid(W_f z_k + b_f) * softplus(W_s z_k + b_s)           (C)
 *     out_i = x_i + aggr_{k: tgt_k = i} m_k                             aggr in {MDL_MEAN, MDL_SUM}
 *
 * Weights are handed over pre-packed (mdl_cgconv_pack_weights) so the per-step cost of converting
 * the fp32 master weights is one tiny kernel.
 *   wpack: mdl_cgconv_wpack_bytes(C,G,dtype) bytes; bpack: [2*Cp] fp32, Cp = 32*ceil(C/32).
 *   w_f, w_s: [C, 2C+G] fp32 row-major (nn.Linear layout, column order target|source|edge);
 *   b_f, b_s: [C] fp32 or NULL. */
size_t mdl_cgconv_wpack_bytes(int C, int G, int dtype);
int mdl_cgconv_pack_weights(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C,
                            int G, void* wpack, float* bpack, int dtype, mdlStream_t stream);
/* The same launch also writes the operand of the backward node kernel (mdl_cgconv_pack_node_weights: wn_t [C][4Cp] bf16), so a
 * training step packs each layer's weights once (dtype MDL_BF16). */
int mdl_cgconv_pack_weights_node(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C,
                                 int G, void* wpack, float* bpack, void* wn_t, int dtype, mdlStream_t stream);

/* The same for every conv layer of a model in ONE launch: host tables of L <= 16 device pointers (b_f / b_s / wn_t tables or
 * single entries may be NULL); all layers share C, G, dtype.  A model's weights are all known before its first layer runs, and
 * at the reference's batch size (config.yml:136) a pack launch per layer is 5 us of a launch-bound step. */
int mdl_cgconv_pack_weights_multi(int L, const float* const* w_f, const float* const* b_f, const float* const* w_s,
                                  const float* const* b_s, int C, int G, void* const* wpack, float* const* bpack,
                                  void* const* wn_t, int dtype, mdlStream_t stream);

/* x: [N, C]; edge_attr: [E, G] (leading dim G); out: [N, C]; all in `dtype`. */
int mdl_cgconv_fwd(const void* x, const void* edge_attr, const int32_t* rowptr, const int32_t* src,
                   const int32_t* tgt, const int32_t* eperm, const void* wpack, const float* bpack,
                   void* out, int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream);

/* ---- CGConv, struct entry points (round 5) ------------------------------------------------------------------------------
 * The forward with its optional epilogue, the backward edge pass and the node kernel each take ONE argument struct: every
 * variant the positional entry points of rounds 1-4 had grown (`_bwd`, `_bwd_h`, `_bwd_hb`, `_bwd_node`, `_bwd_node_z`,
 * `_bwd_node_h`) is a field here, and the execution flags have a field of their own instead of riding in `dtype`.
 * `size` must be sizeof(the struct): a caller built against another layout is rejected (MDL_E_ARG), never misread; fields
 * a call does not use are zero.  Same reference call site as above (cgcnn.py:80-83,136-145 and its autograd backward). */
typedef struct MdlCgConv {
    uint32_t size;            /* sizeof(MdlCgConv) */
    int32_t dtype;            /* MDL_F32 | MDL_BF16: x, edge_attr, out, grad_out, r_tgt */
    uint32_t flags;           /* bwd: MDL_DETERMINISTIC | MDL_K3_PER_WAVE | MDL_K3_EDGE_LANE; fwd: 0 */
    int32_t aggr;             /* MDL_MEAN | MDL_SUM */
    int64_t N, E;
    int32_t C, G;
    const void* x;            /* [N, C] */
    const void* edge_attr;    /* [E, G], CSR order unless eperm is given */
    const int32_t* rowptr;    /* [N + 1] */
    const int32_t* src;       /* [E] */
    const int32_t* tgt;       /* [E] */
    const int32_t* eperm;     /* [E] or NULL: row of edge_attr for CSR slot k (generic kernels) */
    const void* wpack;        /* mdl_cgconv_pack_weights */
    const float* bpack;
    /* forward (mdl_cgconv_fwd_ex) */
    void* out;                /* [N, C] */
    float* bn_sums;           /* optional: statistics of `out` for the training-mode BatchNorm1d behind the layer
                               * (cgcnn.py:143), formed in the epilogue — per column sum (v - shift) and sum (v - shift)^2 of the
                               * ROUNDED outputs over the rows that exist, into one of the MDL_BN_REPLICAS copies (layout of
                               * mdl_bn_stats, caller zero-fills, (2 MDL_BN_REPLICAS + 3) * C floats: the kernel stores the shift
                               * it used in the row behind the totals rows); follow with
                               * mdl_bn_apply_n(..., dtype | MDL_BN_SHIFT_ROW).  bf16, C in {32, 64}, G = 50, no eperm. */
    const float* bn_shift;    /* [C] fp32 or NULL (zeros): any per-column value near the column mean, e.g. the beta of the
                               * BatchNorm in front of the layer — keeps E[v^2] - E[v]^2 from cancelling in fp32 */
    const int64_t* bn_rows;   /* device row count of a padded static batch, or NULL (N) */
    /* backward edge pass (mdl_cgconv_bwd_ex).  With dpre_k = d loss / d (W z_k + b) in R^{2Cp} (f half | s half):
     *     r_tgt[i, :] = sum_{k: tgt_k = i} dpre_k      [N, 2Cp] in `dtype`, written once per node (no atomics)
     *     r_src[j, :] += sum_{k: src_k = j} dpre_k     [N, 2Cp] in r_src_dtype, atomics; caller zero-fills
     *     dwe[c, g]   += sum_k dpre_k[c] * edge_attr_k[g]   [2Cp, Gp] fp32, Gp = 64*ceil(G/64); caller zero-fills
     *     db[c]       += sum_i r_tgt[i, c]                  [2Cp] fp32 bias gradient; caller zero-fills; may be NULL
     * from which the node level forms dx = g + r_tgt W_tgt + r_src W_src, dW_tgt = r_tgt^T x, dW_src = r_src^T x
     * (mdl_cgconv_bwd_node_ex, or library GEMMs).  The gate pre-activations are recomputed, not stored. */
    const void* grad_out;     /* [N, C] */
    void* r_tgt;
    void* r_src;
    int32_t r_src_dtype;      /* MDL_F32; or MDL_BF16 (dtype MDL_BF16, C in {32, 64, 128}, G = 50, no eperm): packed bf16 atomics,
                               * half the atomic operations and bytes; what the balance prefix and the MDL_K3_* flags go with */
    int32_t ld_dwe;           /* leading dimension of dwe in floats; 0 = Gp.  With ld_dwe = 2C + G and dwe = dW + 2C the rows land
                               * straight in the two Linears' stacked weight gradient dW [2C, 2C + G] (rows f | s, columns target |
                               * source | edge; C == Cp) and mdl_cgconv_assemble_grads is not needed */
    float* dwe;
    float* db;
    void* workspace;          /* optional scratch, mdl_cgconv_workspace_bytes (dynamic group scheduling of the per-wave kernel) */
    size_t workspace_bytes;
    const int32_t* balance;   /* optional [N + 1] inclusive cost prefix (mdl_cgconv_balance + cumsum): the workgroups of the
                               * edge-per-lane kernel take node ranges of equal COST; NULL: equal edge + node counts */
} MdlCgConv;
int mdl_cgconv_fwd_ex(const MdlCgConv* args, mdlStream_t stream);
int mdl_cgconv_bwd_ex(const MdlCgConv* args, mdlStream_t stream);

/* Work balance of the edge-per-lane backward: cost[0] = 0, cost[n + 1] = cost of node n in quarter units (4 per edge and node,
 * + 5 per edge whose source is 48 or more rows away from its target, + 4 for a node without edges: 32 of them make an empty
 * tile that costs a full tile's time).  The caller turns it into an inclusive prefix sum (int32, [N + 1]) for
 * MdlCgConv.balance (topology only: one prefix per batch serves every layer). */
int mdl_cgconv_balance(const int32_t* rowptr, const int32_t* src, int64_t N, int32_t* cost, mdlStream_t stream);

/* Optional scratch for mdl_cgconv_bwd_ex (caller-owned device memory, contents ignored; the library zeroes what it
 * uses, on the stream).  With it the per-wave backward hands 32-node groups to its waves dynamically (large problems);
 * without it (NULL / 0) every wave gets a fixed edge-balanced node range.  Same results up to the order of the
 * fp32 atomic adds into r_src / dwe / db. */
size_t mdl_cgconv_workspace_bytes(int64_t N, int64_t E, int C, int G, int dtype);

/* Node-level dense half of the CGConv backward (same reference call site), one pass over r_tgt / r_src:
 *     dx  [N, C]   = grad_out + [r_tgt | r_src] @ Wn          Wn = wn_t^T, wn_t: [C, 4Cp] in `dtype`
 *     dwn [4Cp, C] += [r_tgt | r_src]^T @ x                   fp32, caller zero-fills
 * Row blocks of Wn / dwn: (f_tgt, s_tgt, f_src, s_src), each Cp rows.  Supported: dtype MDL_BF16, C in {32, 64} (C == Cp);
 * otherwise MDL_E_UNSUPP and the caller uses library GEMMs.  zero_src != 0: r_src is handed back ZEROED (this kernel is its
 * only reader), so a caller that keeps ONE r_src buffer for all layers and steps never fills it again. */
typedef struct MdlCgNode {
    uint32_t size;            /* sizeof(MdlCgNode) */
    int32_t dtype;            /* MDL_BF16 */
    uint32_t flags;           /* MDL_DETERMINISTIC (one workgroup: every dwn element gets its terms from one wave) */
    int32_t zero_src;
    int64_t N;
    int32_t C;
    int32_t r_src_dtype;      /* MDL_F32 | MDL_BF16: what the edge pass accumulated */
    int32_t ld_dwn;           /* 0: dwn is [4Cp, C] as above.  > 0: dwn IS the stacked weight gradient dW [2C, ld_dwn] of the two Linears
                               * (ld_dwn = 2C + G): block b of the product's rows is added to rows (b & 1) C + c, columns (b >> 1) C + k */
    int32_t reserved;
    const void* x;            /* [N, C] */
    const void* grad_out;     /* [N, C] */
    const void* r_tgt;        /* [N, 2Cp] in dtype */
    void* r_src;              /* [N, 2Cp] in r_src_dtype */
    const void* wn_t;         /* [C, 4Cp] in dtype (mdl_cgconv_pack_node_weights / mdl_cgconv_pack_weights_node) */
    void* dx;                 /* [N, C] */
    float* dwn;               /* [4Cp, C] */
} MdlCgNode;
int mdl_cgconv_bwd_node_ex(const MdlCgNode* args, mdlStream_t stream);

/* Small layout helpers around the backward (replace the cat / transpose / cast / clone chain autograd would run):
 *   wn_t [C, 4Cp] (bf16) = Wn^T for mdl_cgconv_bwd_node from the two nn.Linear weights [C, 2C+G] (fp32);
 *   dw_f, dw_s [C, 2C+G], db_f, db_s [C] (fp32; db_* may be NULL) from dwn [4Cp, C], dwe [2Cp, GP], db [2Cp]. */
int mdl_cgconv_pack_node_weights(const float* w_f, const float* w_s, int C, int G, void* wn_t, int dtype, mdlStream_t stream);
int mdl_cgconv_assemble_grads(const float* dwn, const float* dwe, const float* db, int C, int G, float* dw_f, float* dw_s,
                              float* db_f, float* db_s, mdlStream_t stream);

/* ---- K8: device-side batch assembly ---------------------------------------------------------------
 * Builds one mini-batch from the device-resident flat dataset in a single launch (one workgroup per
 * graph).  Replaces the Python collate + H2D of the PyG DataLoader built at
 * matdeeplearn/training/training.py:300-325 / :39 (Batch.from_data_list semantics: per-key concatenation,
 * edge indices offset by the graph's first node, `batch` vector).
 * ids/noff/eoff: [B], [B+1], [B+1] int64 on the device (noff/eoff = exclusive prefix sums of the batch's node
 * and edge counts; noff[B] = N, eoff[B] = E).  Dataset arrays: node_ptr/edge_ptr [G+1] int64, x_all [Nt,F] fp32,
 * src_l/tgt_l [Et] graph-local int32 (sorted by target inside every graph), dist/dist_norm [Et] fp32,
 * lrowptr [Nt] int32 (exclusive in-degree prefix inside the node's graph), y_all [G,T] fp32.
 * Outputs: x [N,F] in `dtype`, batch [N] int64, rowptr [N+1], src/tgt [E] int32, ew/dn [E] fp32, y [B] fp32. */
int mdl_assemble_batch(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                       const int64_t* edge_ptr, const float* x_all, const int32_t* src_l, const int32_t* tgt_l,
                       const float* dist, const float* dist_norm, const int32_t* lrowptr, const float* y_all,
                       void* x, int64_t* batch, int32_t* rowptr, int32_t* src, int32_t* tgt, float* ew, float* dn,
                       float* y, int B, int F, int T, int target_index, int dtype, mdlStream_t stream);
/* The same into the PADDED buffers of a static batch (n_cap nodes, e_cap edge slots: HIP-graph replays of the training step at
 * the reference's batch size, config.yml:136) with mdl_pad_batch_tail and mdl_pad_edge_tail (below) done by extra workgroups of
 * the same launch, and pool_seg [n_cap] (may be NULL) = `batch` as int32 — four launches of a launch-bound step as one.
 * col_s / eid_s / src_s: the by-source arrays whose tails are padded too (all three or all NULL). */
int mdl_assemble_batch_padded(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                              const int64_t* edge_ptr, const float* x_all, const int32_t* src_l, const int32_t* tgt_l,
                              const float* dist, const float* dist_norm, const int32_t* lrowptr, const float* y_all,
                              void* x, int64_t* batch, int32_t* rowptr, int32_t* src, int32_t* tgt, float* ew, float* dn,
                              float* y, int B, int F, int T, int target_index, int dtype, int64_t n_cap, int64_t e_cap,
                              int32_t* pool_seg, int32_t* col_s, int32_t* eid_s, int32_t* src_s, mdlStream_t stream);

/* dx = g * sigmoid(pre) for y = softplus(pre) - ln 2 given y (the SchNet activation; sigmoid(pre) = 1 - exp(-(y + ln 2))): one
 * bf16 pass instead of six library elementwise launches.  n = number of elements (even). */
int mdl_ssp_bwd(const void* g, const void* y, void* dx, int64_t n, int dtype, mdlStream_t stream);

/* ---- K6: dense layer over edge rows with gathered addends (MEGNet edge block, megnet.py:41-56) ----------------------
 *     out[e, :] = act( x[e, :] W^T + bias + p1[idx1[e], :] + p2[idx2[e], :] + p3[idx3[e], :] )        bf16, act 0 none / 1 relu
 * x: [N, K] (the edge state), W: [M, K], p_i: [rows_i, M] per-node / per-graph projections of the OTHER column blocks of
 * the reference's concatenated input [x[row] | x[col] | e | u[batch]] (any p_i may be NULL).  The [E, 4d] concatenation and
 * its K = 4d product never exist.  Same shape limits as mdl_linear_act; with tables M <= 128 and rows_i * M * 2 < 2^31 bytes. */
/* mdl_linear_gather_act (tables optional) that also takes the BatchNorm statistics of its output on the way out: the
 * epilogue adds, per column, sum out and sum out^2 of the rounded bf16 values of the rows below *n_rows_dev (NULL = all N)
 * into one of the MDL_BN_REPLICAS copies of bn_sums (layout of mdl_bn_stats; caller zero-fills), so that
 * Linear -> ReLU -> BatchNorm1d (matdeeplearn/models/megnet.py:47-48) needs no statistics pass over [N, M]:
 * follow with mdl_bn_apply_n(..., dtype | MDL_BN_SHIFT_ROW): the sums are formed about output row 0, which the
 * kernel evaluates for itself and stores behind the totals rows (bn_sums holds (2 MDL_BN_REPLICAS + 3) * M floats).  Even 34 <= M <= 160, K <= 160. */
int mdl_linear_act_stats(const void* x, const void* w, const void* bias, const void* p1, const int32_t* idx1, const void* p2,
                         const int32_t* idx2, const void* p3, const int32_t* idx3, void* out, int64_t N, int K, int M, int act,
                         float* bn_sums, const int64_t* n_rows_dev, int dtype, mdlStream_t stream);
int mdl_linear_gather_act(const void* x, const void* w, const void* bias, const void* p1, const int32_t* idx1,
                          const void* p2, const int32_t* idx2, const void* p3, const int32_t* idx3, void* out,
                          int64_t N, int K, int M, int act, int dtype, mdlStream_t stream);

/* ---- K7: NNConv edge contraction without the E x C x C weight tensor --------------------------------------------------
 * torch_geometric.nn.NNConv(in, out, nn, aggr) at matdeeplearn/models/mpnn.py:83-88,148-157: m_e = x_j^T reshape(nn(e), [Ci, Co]).
 * With the last layer of `nn` = Linear(D3, Ci*Co) (weight W2, bias b2) the product is re-associated as
 *     m_e = Y_j (Co x D3) . h_e + Z_j,   Y = x @ W2.view(Ci, Co*D3),  Z = x @ b2.view(Ci, Co),  h_e = nn[:-1](e)
 * (two dense GEMMs over the NODES, done by the caller) and these kernels do the per-edge part, walking the edges BY SOURCE:
 *     fwd: m[eid, :]  = Y[j] . h[eid, :]                         for every slot s in [rowptr_s[j], rowptr_s[j+1]), eid = eid_s[s]
 *     bwd: dh[eid, :] = Y[j]^T . dm[eid, :];   dY[j] = sum_s dm[eid] (x) h[eid]    (dY rows of nodes without out-edges = 0)
 * Y, dY: [N, Co*D3]; h, dh: [E, D3]; m, dm: [E, Co]; all `dtype` (fp32 accumulation); eid_s may be NULL (slot = edge id). */
int mdl_nnconv_msg_fwd(const void* Y, const void* h, const int32_t* rowptr_s, const int32_t* eid_s, void* m, int64_t N,
                       int Co, int D3, int dtype, mdlStream_t stream);
int mdl_nnconv_msg_bwd(const void* Y, const void* h, const void* dm, const int32_t* rowptr_s, const int32_t* eid_s,
                       void* dh, void* dY, int64_t N, int Co, int D3, int dtype, mdlStream_t stream);

/* Tail of a padded static batch (buffers sized for n_cap nodes, the batch fills the first N = noff[B], read on the device):
 * padding nodes get rowptr = E (no edges) and batch = B (a dummy graph).  Part of the HIP-graph replay path. */
int mdl_pad_batch_tail(const int64_t* noff, const int64_t* eoff, int B, int64_t n_cap, int32_t* rowptr, int64_t* batch,
                       mdlStream_t stream);

/* CSR by SOURCE of the assembled batch from the dataset's per-graph by-source order (eperm_s [Et]: graph-local edge id at
 * every by-source position, stable; lrowptr_s [Nt]: exclusive out-degree prefix inside the node's graph): rowptr_s [N+1]
 * (n_cap+1 when padded), col_s / src_s / eid_s [E] = target, source and batch edge id per by-source slot.  This is the
 * index PyG's MessagePassing backward (scatter of the message gradient to x_j) and `scatter_mean(e, edge_index[0])`
 * (matdeeplearn/models/megnet.py:86,130) imply; taking it from the loader replaces one device sort per batch.
 * n_cap >= 0: also close the tail of a padded static batch (rowptr_s[n] = E for the padding nodes). */
int mdl_assemble_transposed(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                            const int64_t* edge_ptr, const int32_t* src_l, const int32_t* tgt_l, const int32_t* eperm_s,
                            const int32_t* lrowptr_s, int32_t* rowptr_s, int32_t* col_s, int32_t* eid_s, int32_t* src_s, int B,
                            int64_t n_cap, mdlStream_t stream);
/* Padded static batches: edge slots [E, e_cap) of src / tgt (and of the by-source arrays when given) point at the first
 * padding node, whose rows carry zero gradients — per-edge kernels that run over all e_cap slots then add nothing. */
int mdl_pad_edge_tail(const int64_t* noff, const int64_t* eoff, int B, int64_t n_cap, int64_t e_cap, int32_t* src, int32_t* tgt,
                      int32_t* col_s, int32_t* eid_s, int32_t* src_s, mdlStream_t stream);

/* ---- training loss + gradient in one launch ---------------------------------------------------
 * loss[0] = mean over n elements of |pred - y| (kind 0, F.l1_loss: the reference default, config.yml:117) or (pred - y)^2
 * (kind 1, F.mse_loss); grad[i] = d loss / d pred[i].  fp32.  Replaces the eight elementwise / reduction launches of
 * `getattr(F, loss)(output, data.y)` + its autograd (matdeeplearn/training/training.py:44-47). */
int mdl_loss_fwd_bwd(const float* pred, const float* y, int64_t n, int kind, float* loss, float* grad, mdlStream_t stream);
/* same over the first n of n_total predictions (y [n]; grad [n_total], exact zeros past n): the padded static batch of the
 * HIP-graph step carries a dummy graph behind the batch's B graphs, and `output[:B]` in front of the loss would put a slice
 * node (zero fill + copy in the backward) between the model and the loss. */
int mdl_loss_fwd_bwd_rows(const float* pred, const float* y, int64_t n, int64_t n_total, int kind, float* loss, float* grad,
                          mdlStream_t stream);

/* ---- training-mode BatchNorm1d over rows, x: [N, C] row-major ------------------------------------
 * Replaces torch.nn.BatchNorm1d as applied after every conv layer (matdeeplearn/models/cgcnn.py:85-87,143)
 * and inside the MEGNet MLPs (megnet.py:47-48).  `sums` is an fp32 scratch of mdl_bn_sums_rows() x C floats the
 * caller zero-fills before each *_stats call: MDL_BN_REPLICAS copies of a [2, C] accumulator (the reduction spreads
 * its atomics over them) followed by one [2, C] row pair of totals, which the *_apply call publishes
 * (sums + MDL_BN_REPLICAS*2*C: after bwd_apply = dbeta | dgamma).  `save` is [2, C] fp32 (mean | invstd) written
 * by mdl_bn_apply.
 * Supported: C a multiple of 8 (bf16) / 4 (fp32) with 256 % (C/W) == 0, C <= 256.
 *   stats:      copy[0] += sum(x - x[0,:]), copy[1] += sum((x - x[0,:])^2)   (shifted sums)
 *   apply:      y = (x - mean) * rsqrt(var_biased + eps) * gamma + beta; running_mean/var (unbiased) updated
 *               with `momentum` when non-NULL
 *   bwd_stats:  copy[0] += sum(dy), copy[1] += sum(dy * xhat)       ( = dbeta, dgamma )
 *   bwd_apply:  dx = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)) */
#define MDL_BN_REPLICAS 16
int mdl_bn_sums_rows(void);
/* The *_n variants take the number of rows that EXIST from device memory (n_rows_dev, may be NULL = all N): rows
 * [*n_rows_dev, N) are padding of a static batch — excluded from the statistics, written as zeros by the apply passes.
 * They exist for HIP-graph replays, where launch arguments are frozen at capture but the batch changes every step. */
int mdl_bn_stats_n(const void* x, float* sums, int64_t N, int C, const int64_t* n_rows_dev, int dtype, mdlStream_t stream);
int mdl_bn_apply_n(const void* x, float* sums, const float* gamma, const float* beta, float* save, float* running_mean,
                   float* running_var, void* y, int64_t N, int C, float eps, float momentum, const int64_t* n_rows_dev,
                   int dtype, mdlStream_t stream);
int mdl_bn_bwd_stats_n(const void* dy, const void* x, const float* save, float* sums, int64_t N, int C,
                       const int64_t* n_rows_dev, int dtype, mdlStream_t stream);
int mdl_bn_bwd_apply_n(const void* dy, const void* x, const float* save, float* sums, const float* gamma, void* dx,
                       int64_t N, int C, const int64_t* n_rows_dev, int dtype, mdlStream_t stream);
/* same, for Linear -> ReLU -> BatchNorm1d (the layer order of the reference's MEGNet blocks, matdeeplearn/models/megnet.py:47-48):
 * x is the ReLU's output and dx the gradient w.r.t. the ReLU's INPUT (zero where x <= 0), so that the dense layer's backward
 * behind it (mdl_dense_bwd with act = 0) needs neither the activation staging nor x's rows for the mask.  bf16 only. */
int mdl_bn_bwd_apply_relu_n(const void* dy, const void* x, const float* save, float* sums, const float* gamma, void* dx,
                            int64_t N, int C, const int64_t* n_rows_dev, int dtype, mdlStream_t stream);
/* ---- node-level dense layer forward, fused: out[N, M] = act(x[N, K] . w[M, K]^T + bias) ---------------
 * Replaces `getattr(F, act)(lin(out))` of the pre-FC / post-FC loops (matdeeplearn/models/cgcnn.py:124-130,155-166)
 * for the tall-skinny shapes of this path.  x: dense rows (leading dimension K), 16-byte aligned; w [M, K] and bias [M]
 * (may be NULL) in `dtype`; act: 0 = none, 1 = ReLU, 2 = shifted softplus (softplus(v) - ln 2, the SchNet filter
 * activation).  bf16 only, K even, 4 <= K <= 256, M <= 128 — or M <= 160 with K <= 160 (SchNet's 150-wide filters). */
int mdl_linear_act(const void* x, const void* w, const void* bias, void* out, int64_t N, int K, int M, int act, int dtype,
                   mdlStream_t stream);
/* same with x .* act'(y) in place of x: x = gradient w.r.t. the OUTPUT y[N, K] of an activated layer (xact 1: ReLU, y > 0;
 * 2: shifted softplus, 1 - exp(-(y + ln 2)); 0: none), applied while the tile is staged — the dX product of a fused
 * Linear + activation without a `threshold_backward` / softplus-backward pass (the reference's autograd runs one per
 * activation, e.g. matdeeplearn/models/megnet.py:41-56). */
int mdl_linear_act_in(const void* x, const void* y, int xact, const void* w, const void* bias, void* out, int64_t N, int K,
                      int M, int act, int dtype, mdlStream_t stream);

/* The post-FC head — post_lin_list + lin_out of the reference models on the pooled graph rows (matdeeplearn/models/cgcnn.py:
 * 155-174) — as ONE launch per direction: h_0 = relu(x w_0^T + b_0), ..., y = h_{NL-2} w_{NL-1}^T + b_{NL-1}.
 * bf16; 1 <= NL <= 4 dense layers, every width <= 64 (hidden widths and K0 even), ReLU between the layers, none after the last.
 * w / b / h / dw / db are HOST arrays of NL device pointers (b and db entries may be NULL).
 *   fwd: h[l] receives layer l's output [N, M[l]] (h[NL-1] = y, fp32 rows under dtype | MDL_MLP_F32_IO); the hidden ones are
 *        what the backward needs.
 *   bwd: h[l] (l < NL-1) = the saved hidden outputs, gy = dL/dy [N, M[NL-1]] (fp32 rows under MDL_MLP_F32_IO); dw[l] [M[l], K_l] and db[l] [M[l]] are fp32,
 *        zero-filled by the caller and accumulated with atomics; dx [N, K0] may be NULL. */
int mdl_mlp_head_fwd(const void* x, const void* const* w, const void* const* b, void* const* h, int64_t N, int K0, int NL,
                     const int* M, int dtype, mdlStream_t stream);
int mdl_mlp_head_bwd(const void* x, const void* const* w, const void* const* h, const void* gy, void* dx, float* const* dw,
                     float* const* db, int64_t N, int K0, int NL, const int* M, int dtype, mdlStream_t stream);

/* The gate arithmetic of one step of a single-layer GRU — the reference's MPNN runs `out, h = self.gru_list[i](m.unsqueeze(0), h)`
 * after every NNConv layer (matdeeplearn/models/mpnn.py:160-161): gi = W_ih m + b_ih and gh = W_hh h + b_hh are dense layers of this
 * library ([N, 3C] in `dtype`, torch's gate order r | z | n); these two entry points replace the ~12 forward and ~25 backward
 * elementwise / chunk / cat / cast launches of the gates written out with tensor operations:
 *   fwd: r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n), h_out = n + z (h - n)   (fp32 arithmetic;
 *        h, h_out [N, C] fp32; out_lp [N, C] in `dtype` = the copy of h_out the next layer reads, may be NULL);
 *   bwd: from g_h = dL/dh_out (fp32) and g_lp = dL/dout_lp (`dtype`) — either may be NULL, not both — with r, z, n recomputed:
 *        dgi, dgh [N, 3C] in `dtype` (the output gradients of the two dense layers), dh [N, C] fp32. */
int mdl_gru_gates_fwd(const void* gi, const void* gh, const float* h, float* h_out, void* out_lp, int64_t N, int C, int dtype,
                      mdlStream_t stream);
int mdl_gru_gates_bwd(const void* gi, const void* gh, const float* h, const float* g_h, const void* g_lp, void* dgi, void* dgh,
                      float* dh, int64_t N, int C, int dtype, mdlStream_t stream);

/* out[N, M] = x[N, K] w[M, K]^T for a wide output (M in the thousands; bf16, even K <= 160): NNConv's per-node operand
 * Y = x W2r of the re-associated message (matdeeplearn/models/mpnn.py:83-88 — C_out * d3 = 10^4 columns), a write stream of
 * N * M * 2 bytes that the library ran as a 256x256x32 macro-tile GEMM (771 us for 6.1e4 x 100 x 1e4). */
int mdl_linear_wide(const void* x, const void* w, void* out, int64_t N, int K, int64_t M, int dtype, mdlStream_t stream);

/* ---- tall-skinny TN GEMM: weight gradients of node-level Linear layers ---------------------------
 * c[M, K] (fp32, row-major, caller zero-fills) += a[N, M]^T . b[N, K]   a, b bf16 with leading dims lda, ldb.
 * Replaces the (out x N)(N x in) product autograd forms for dW of the reference's node-level Linears
 * (e.g. pre_lin_list, matdeeplearn/models/cgcnn.py:64-74,124-130).  1 <= M <= 128, 1 <= K <= 256 (or even M <= 160, K <= 160), bf16 only. */
int mdl_gemm_tn(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c, int64_t N, int dtype,
                mdlStream_t stream);
/* same, plus colsum[M] (fp32, caller zero-fills) += column sums of a — the bias gradient of that Linear
 * (`g.sum(0)` in the reference's autograd), out of the same pass.  Needs even M, K, lda, ldb and K <= 158. */
int mdl_gemm_tn_colsum(const void* a, int64_t lda, int M, const void* b, int64_t ldb, int K, float* c, float* colsum,
                       int64_t N, int dtype, mdlStream_t stream);
/* same with a .* act'(y) in place of a: a = gradient w.r.t. the OUTPUT y[N, M] (leading dim ldy) of an activated Linear,
 * act: 0 = none, 1 = ReLU (y > 0), 2 = shifted softplus (1 - exp(-(y + ln 2))); the factor is applied while the tile is
 * staged, so `threshold_backward` / the softplus backward of the reference's autograd need no pass of their own when the
 * Linear's input needs no gradient (first layer of SchNet's filter network, matdeeplearn/models/schnet.py:81 via
 * torch_geometric.nn.models.schnet.InteractionBlock.mlp).  colsum (may be NULL) = column sums of a .* act'(y).
 * Same shape limits as mdl_gemm_tn_colsum. */
int mdl_gemm_tn_act(const void* a, int64_t lda, int M, const void* y, int64_t ldy, int act, const void* b, int64_t ldb,
                    int K, float* c, float* colsum, int64_t N, int dtype, mdlStream_t stream);
/* The whole backward of a tall dense layer y = act(x w^T + b) in one streaming pass: with g' = g .* act'(y) (as above),
 *     dw[M, K] (fp32, caller zero-fills) += g'^T . x      db[M] (fp32, zero-filled; may be NULL) += column sums of g'
 *     dx[N, K] (bf16, leading dim lddx)   = g' . w        w: [M, K] bf16 row-major (the layer's weight)
 * xout: 0, or 1 / 2 when x is itself the relu / shifted-softplus OUTPUT of the layer in front and dx is wanted w.r.t. that
 * layer's pre-activation: dx .*= act_in'(x), taken from the staged x tile (the layer in front then needs act = 0 and no y).
 * gm (may be NULL): [N, M] bf16, dense — receives g' itself (for callers that reduce it further: the gathered tables of K6).
 * Replaces autograd's threshold_backward / softplus_backward + mm (dX) + mm (dW) + sum (db) of the edge-level Linears at
 * matdeeplearn/models/megnet.py:41-56,84-101 and matdeeplearn/models/schnet.py:81 (InteractionBlock.mlp): g, y, x are read
 * once.  Even 34 <= M <= 160, 34 <= K <= 160 (158 with db), even ldg / ldx / ldy, bf16 only; MDL_DETERMINISTIC: one workgroup. */
int mdl_dense_bwd(const void* g, int64_t ldg, int M, const void* y, int64_t ldy, int act, const void* x, int64_t ldx, int K,
                  const void* w, void* dx, int64_t lddx, int xout, void* gm, float* dw, float* db, int64_t N, int dtype,
                  mdlStream_t stream);
/* The streaming TN products end with one fp32 atomic per output element and WORKGROUP, and a launch's atomics retire at ~160 G/s
 * whatever their addresses (256 workgroups x 25 blocks: 42 us beside 16 us of product on 1.1e5 rows).  The _ex forms take
 * `scratch` = mdl_tn_scratch_bytes() bytes (16-byte aligned, contents undefined on entry and exit) or NULL: with it the blocks
 * leave as plain stores and a second launch adds them into dw / db (c / colsum).  Everything else as the forms without it, which
 * are the _ex forms with scratch = NULL. */
size_t mdl_tn_scratch_bytes(void);
int mdl_dense_bwd_ex(const void* g, int64_t ldg, int M, const void* y, int64_t ldy, int act, const void* x, int64_t ldx, int K,
                     const void* w, void* dx, int64_t lddx, int xout, void* gm, float* dw, float* db, void* scratch, int64_t N,
                     int dtype, mdlStream_t stream);
int mdl_gemm_tn_ex(const void* a, int64_t lda, int M, const void* y, int64_t ldy, int act, const void* b, int64_t ldb, int K,
                   float* c, float* colsum, void* scratch, int64_t N, int dtype, mdlStream_t stream);

/* ---- generic gather / edge-weighted gather-reduce (SchNet CFConv, GCNConv, MEGNet, NNConv) ------
 * Replace the index_select + elementwise + torch_scatter sequence of PyG MessagePassing.propagate at
 * matdeeplearn/models/schnet.py:134-143, gcn.py:135-144, megnet.py:41-56,84-101,129-147, mpnn.py:148-157.
 *
 * out[k, :] = src[idx[k], :]                                        src: [*, C]; idx: [E] int32; out: [E, C] */
int mdl_gather_rows(const void* src, const int32_t* idx, void* out, int64_t E, int64_t C, int dtype,
                    mdlStream_t stream);
/* out[i, :] = reduce_{k in [rowptr[i], rowptr[i+1])} h[col[k], :] * w[eid?eid[k]:k, :] * scale[eid?eid[k]:k]
 * h: [*, F]; w: [E, F] or NULL; scale: [E] fp32 or NULL; col: [E] int32 (node gathered by CSR slot k);
 * eid: [E] int32 original edge id of slot k or NULL; out: [N, F]; reduce in {MDL_SUM, MDL_MEAN}.
 * CFConv: rows = targets, col = sources, w = filter, scale = cosine cutoff.  Gradient w.r.t. h: the
 * same call on the transposed CSR (rows = sources, col = targets, h = grad_out). */
int mdl_gather_mul_reduce(const void* h, const void* w, const float* scale, const int32_t* rowptr,
                          const int32_t* col, const int32_t* eid, void* out, int64_t N, int64_t F, int reduce,
                          int dtype, mdlStream_t stream);
/* Backward of mdl_gather_mul_reduce (reduce = sum) in ONE walk over the transposed CSR (rowptr_s / col_s = target of the
 * slot / eid_s = edge id of the slot): dh[j,:] = sum_{k: src_k = j} g[tgt_k,:] * w[k,:] * scale[k] and, from the same operands,
 * dw[k,:] = g[tgt_k,:] * h[src_k,:] * scale[k] (the filter gradient of CFConv, schnet.py:134-143) — replaces the pair
 * mdl_gather_mul_reduce(transposed) + mdl_edge_mul.  bf16, even F <= 512.  Rows of dw whose edge belongs to no by-source
 * segment (unused slots of a padded static batch) are not written. */
int mdl_gather_mul_reduce_dw(const void* g, const void* w, const float* scale, const int32_t* rowptr_s,
                             const int32_t* col_s, const int32_t* eid_s, void* dh, const void* h, void* dw,
                             int64_t N, int64_t F, int dtype, mdlStream_t stream);
/* ---- K4: the fused forward of SchNet's continuous-filter convolution -----------------------------
 * Replaces, in ONE pass over the CSR-ordered edges, what torch_geometric's InteractionBlock / CFConv run at
 * matdeeplearn/models/schnet.py:131-145 (constructed at schnet.py:81) as filter network + propagate:
 *     a1[e, :]  = ssp(rbf[e, :] W1^T + b1)            ssp(v) = softplus(v) - ln 2          (mlp[0], mlp[1])
 *     w[e, :]   = a1[e, :] W2^T + b2                                                       (mlp[2]: the filter)
 *     out[i, :] = sum_{e in [rowptr[i], rowptr[i+1])} h[src[e], :] * w[e, :] * cut[e]      (CFConv.message + aggr = add)
 * rbf: [E, G] edge features in CSR order; cut: [E] fp32 cosine cutoff; h: [N, F] = lin1(x); src / tgt: [E] int32 node per
 * CSR slot; out: [N, F].  a1 / w: [E, F] — the two activations the backward consumes (mdl_gather_mul_reduce_dw, the dense
 * backward of the two layers) — or NULL for inference; rows past rowptr[N] (padded static batch) are zeroed.  The product uses
 * the bf16-ROUNDED filter, like the unfused sequence mdl_linear_act x 2 -> mdl_gather_mul_reduce.  bf16, G = 50, even F in
 * [64, 158] (SchNet_demo: 150; the kernel is static at the padded width 96 / 128 / 160); other shapes: MDL_E_UNSUPP (callers keep the unfused sequence).  wpack: the weights packed by
 * mdl_cfconv_pack_weights (mdl_cfconv_wpack_bytes() bytes, 16-byte aligned) from the fp32 masters W1 [F, G], b1 [F] or NULL,
 * W2 [F, F], b2 [F] or NULL. */
int mdl_cfconv_supported(int F, int G, int dtype);
size_t mdl_cfconv_wpack_bytes(void);
int mdl_cfconv_pack_weights(const float* w1, const float* b1, const float* w2, const float* b2, int F, int G, void* wpack,
                            mdlStream_t stream);
int mdl_cfconv_fwd(const void* rbf, const float* cut, const void* h, const int32_t* rowptr, const int32_t* src,
                   const int32_t* tgt, const void* wpack, void* out, void* a1, void* w, int64_t N, int64_t E, int F, int G,
                   int dtype, mdlStream_t stream);
/* K4b (csrc/cfconv_bwd.hip) — the backward of the same block (matdeeplearn/models/schnet.py:131-145 through autograd) with the
 * filter RECOMPUTED instead of stored.  The gradient w.r.t. h is mdl_cfconv_fwd itself on the by-source CSR with the output
 * gradient in the place of h (rbf / cut rows in by-source order, src := target per slot, tgt := source per slot).  This entry
 * point adds the parameter gradients of the filter network, fp32, into dw1 [F, G], db1 [F] or NULL, dw2 [F, F], db2 [F] or NULL
 * (the caller zero-fills or accumulates) in one pass over the edges: dw_e = g[tgt_e] * h[src_e] * cut_e, a1_e recomputed,
 * dW2 += dw^T a1, da = (dw W2) * ssp'(a1), dW1 += da^T rbf.  Nothing per edge is written.  g: [N, F] gradient w.r.t. the
 * aggregated messages; rowptr[N] = number of edges that exist (rows of the edge arrays past it are ignored); the shapes of
 * mdl_cfconv_supported; dtype | MDL_DETERMINISTIC: one workgroup, bit-reproducible sums.  scratch: NULL (the workgroups add
 * their partial sums into the outputs with atomics) or mdl_cfconv_bwd_w_scratch_bytes() bytes, 16-byte aligned, contents
 * undefined on entry and exit (the partial sums leave as plain stores and a second launch adds them up: 160 us less at
 * SchNet_demo's batch). */
size_t mdl_cfconv_bwd_w_scratch_bytes(void);
int mdl_cfconv_bwd_w(const void* rbf, const float* cut, const void* h, const void* g, const int32_t* rowptr, const int32_t* src,
                     const int32_t* tgt, const void* wpack, float* dw1, float* db1, float* dw2, float* db2, void* scratch, int64_t N,
                     int64_t E, int F, int G, int dtype, mdlStream_t stream);
/* out[e, :] = a[ia[e], :] * b[ib[e], :] * scale[e]   (gradient w.r.t. the per-edge filter w) */
int mdl_edge_mul(const void* a, const int32_t* ia, const void* b, const int32_t* ib, const float* scale, void* out,
                 int64_t E, int64_t F, int dtype, mdlStream_t stream);

/* ---- split-product ("bf16x3") parity mode: operand preparation for the TN GEMM -------------------------
 * hi[k] + lo[k] = src[k] to 16 significant bits (bf16 pair, both rounded to nearest even), so that a fp32 product runs on the
 * bf16 matrix core as a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation.  The CGConv kernels split their operands in
 * registers (MDL_SPLIT_BF16); this entry point prepares the operands of a node-level Linear's weight gradient dW = g^T x
 * (the pre-FC layer, matdeeplearn/models/cgcnn.py:64-74,124-130 through autograd) for three mdl_gemm_tn launches.  n even,
 * src 8-byte, hi / lo 4-byte aligned. */
int mdl_split_bf16(const float* src, void* hi, void* lo, int64_t n, mdlStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MDL_HIP_H */
