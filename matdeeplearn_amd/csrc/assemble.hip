// assemble.hip — K8: device-side batch assembly.  One launch builds a whole mini-batch from the
// device-resident flat dataset (matdeeplearn_amd/process/dataset.py): node features (converted to the
// compute dtype), the `batch` vector, the CSR-by-target index (rowptr / src / tgt with batch-global
// node ids), raw and normalised distances and the targets.  It replaces, on the hot path, the Python
// collate of the PyG DataLoader the reference builds at /root/reference/matdeeplearn/training/training.py:300-325
// (Batch.from_data_list: per-key concatenation, edge_index offset by the graph's first node, `batch`
// vector) and the host->device copy at training.py:39.
// One workgroup per graph: every copy is a contiguous run.  Algorithmic bytes: N*(F*4 + F*s + 12) + E*(4*4 + 4*4).
#include "mdl_common.h"

namespace mdl {

struct AsmParams {
    const int64_t* ids;        // [B] graph ids of the batch (device)
    const int64_t* noff;       // [B+1] first batch node of every graph
    const int64_t* eoff;       // [B+1] first batch edge of every graph
    const int64_t* node_ptr;   // dataset [G+1]
    const int64_t* edge_ptr;   // dataset [G+1]
    const float* x_all;        // dataset [Nt, F] fp32
    const int32_t* src_l;      // dataset [Et] graph-local source
    const int32_t* tgt_l;      // dataset [Et] graph-local target (sorted inside every graph)
    const float* dist;         // dataset [Et]
    const float* dist_norm;    // dataset [Et]
    const int32_t* lrowptr;    // dataset [Nt] exclusive in-degree prefix inside the node's graph
    const float* y_all;        // dataset [G, T]
    void* x;                   // out [N, F] in dtype
    int64_t* batch;            // out [N]
    int32_t* rowptr;           // out [N+1]
    int32_t* src;              // out [E]
    int32_t* tgt;              // out [E]
    float* ew;                 // out [E] raw distance (edge_weight)
    float* dn;                 // out [E] normalised distance
    float* y;                  // out [B]
    int F, T, target_index, B;
    // padded static batch (mdl_assemble_batch_padded; n_cap = 0 otherwise): the blocks past B close the tails in the same launch
    int64_t n_cap, e_cap;
    int32_t* pool_seg;         // out [n_cap] int32 copy of `batch` (the pooling index's segment ids) or null
    int32_t* col_s;            // by-source arrays [e_cap] whose tails are padded too, or null
    int32_t* eid_s;
    int32_t* src_s;
};

constexpr int ASM_TAIL_BLOCKS = 64;

template <typename T>
__global__ __launch_bounds__(256) void assemble_kernel(AsmParams p) {
    const int g = blockIdx.x;
    if (g >= p.B) {
        // tail of a padded static batch (what pad_tail_kernel / pad_edge_tail_kernel below do as launches of their own): padding
        // nodes own no edges and sit in the dummy graph B, unused edge slots point at the first padding node
        const int64_t N = p.noff[p.B], E = p.eoff[p.B];
        const int64_t t0 = (int64_t)(g - p.B) * blockDim.x + threadIdx.x, ts = (int64_t)ASM_TAIL_BLOCKS * blockDim.x;
        for (int64_t n = N + t0; n < p.n_cap; n += ts) {
            p.batch[n] = p.B;
            p.rowptr[n + 1] = (int32_t)E;
            if (p.pool_seg) p.pool_seg[n] = p.B;
        }
        const int32_t pad = (int32_t)(N < p.n_cap ? N : p.n_cap - 1);
        for (int64_t e = E + t0; e < p.e_cap; e += ts) {
            p.src[e] = pad;
            p.tgt[e] = pad;
            if (p.col_s) { p.col_s[e] = pad; p.src_s[e] = pad; p.eid_s[e] = (int32_t)e; }
        }
        return;
    }
    const int64_t gid = p.ids[g];
    const int64_t n0s = p.node_ptr[gid], nn = p.node_ptr[gid + 1] - n0s;
    const int64_t e0s = p.edge_ptr[gid], ne = p.edge_ptr[gid + 1] - e0s;
    const int64_t no = p.noff[g], eo = p.eoff[g];
    T* __restrict__ xo = static_cast<T*>(p.x) + no * p.F;
    const float* __restrict__ xi = p.x_all + n0s * p.F;
    // A block moves only ~3000 feature elements and ~330 edges: every loop of load -> store is a chain of memory round trips
    // (~2 us each on a cold 0.8-GB dataset), and at the reference's batch size the kernel IS that chain.  So the loads of the
    // first trip of every array — the edge arrays (two trips' worth), the row pointers and six feature elements per thread — are
    // all issued (clamped) before the first store; a typical graph then needs one more feature trip.
    constexpr int U = 6;
    const int64_t nx = nn * p.F;
    const int32_t shift = (int32_t)no;
    const int t = threadIdx.x, bd = blockDim.x;
    int32_t sv[2], tv[2];
    float dv[2], nv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t k = e0s + min((int64_t)t + u * bd, ne > 0 ? ne - 1 : 0);
        const bool ok = ne > 0;
        sv[u] = ok ? p.src_l[k] : 0; tv[u] = ok ? p.tgt_l[k] : 0; dv[u] = ok ? p.dist[k] : 0.0f; nv[u] = ok ? p.dist_norm[k] : 0.0f;
    }
    const int32_t lr = t < nn ? p.lrowptr[n0s + t] : 0;
    float v0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v0[u] = nx > 0 ? xi[min((int64_t)t + u * bd, nx - 1)] : 0.0f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t k = (int64_t)t + u * bd;
        if (k < ne) {
            p.src[eo + k] = sv[u] + shift;
            p.tgt[eo + k] = tv[u] + shift;
            p.ew[eo + k] = dv[u];
            p.dn[eo + k] = nv[u];
        }
    }
    if (t < nn) {
        p.batch[no + t] = g;
        p.rowptr[no + t] = (int32_t)(eo + lr);
        if (p.pool_seg) p.pool_seg[no + t] = g;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        if ((int64_t)t + u * bd < nx) Elem<T>::st(xo + t + u * bd, v0[u]);
    // the rest (large graphs)
    for (int64_t q0 = (int64_t)t + U * bd; q0 < nx; q0 += U * bd) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xi[min(q0 + u * bd, nx - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * bd < nx) Elem<T>::st(xo + q0 + u * bd, v[u]);
    }
    for (int64_t j = (int64_t)t + bd; j < nn; j += bd) {
        p.batch[no + j] = g;
        p.rowptr[no + j] = (int32_t)(eo + p.lrowptr[n0s + j]);
        if (p.pool_seg) p.pool_seg[no + j] = g;
    }
    for (int64_t k0 = (int64_t)t + 2 * bd; k0 < ne; k0 += 2 * bd) {
        int32_t sw[2], tw[2];
        float dw[2], nw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t k = e0s + min(k0 + u * bd, ne - 1);
            sw[u] = p.src_l[k]; tw[u] = p.tgt_l[k]; dw[u] = p.dist[k]; nw[u] = p.dist_norm[k];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t k = k0 + u * bd;
            if (k < ne) {
                p.src[eo + k] = sw[u] + shift;
                p.tgt[eo + k] = tw[u] + shift;
                p.ew[eo + k] = dw[u];
                p.dn[eo + k] = nw[u];
            }
        }
    }
    if (threadIdx.x == 0) {
        p.y[g] = p.y_all[gid * p.T + p.target_index];
        if (g == p.B - 1) p.rowptr[no + nn] = (int32_t)(eo + ne);
    }
}

// Tail of a PADDED static batch (HIP-graph replays need fixed shapes: the buffers hold n_cap nodes, the batch fills the
// first N = noff[B]): padding nodes get no edges (rowptr = E) and belong to a dummy graph B, so every kernel that walks
// rowptr / batch sees well-formed, empty rows.  N and E are read on the device — the launch is part of the graph.
__global__ __launch_bounds__(256) void pad_tail_kernel(const int64_t* __restrict__ noff, const int64_t* __restrict__ eoff,
                                                       int B, int64_t n_cap, int32_t* __restrict__ rowptr,
                                                       int64_t* __restrict__ batch) {
    const int64_t N = noff[B];
    const int32_t E = (int32_t)eoff[B];
    for (int64_t n = N + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_cap; n += (int64_t)gridDim.x * blockDim.x) {
        batch[n] = B;
        rowptr[n + 1] = E;
    }
}

// CSR by SOURCE of the batch (the transposed index the by-source reductions walk: CFConv / GCNConv backward, MEGNet's
// scatter at the source row, NNConv) from the dataset's per-graph by-source order, so that no batch needs a device sort:
// the dataset keeps, per graph, eperm_s (local edge id at every by-source position; stable, i.e. by target inside a
// source) and lrowptr_s (exclusive out-degree prefix).  One workgroup per graph; with n_cap >= 0 the tail of a padded
// static batch is closed as well (rowptr_s = E for the padding nodes).
__global__ __launch_bounds__(256) void assemble_transposed_kernel(const int64_t* __restrict__ ids, const int64_t* __restrict__ noff,
                                                                  const int64_t* __restrict__ eoff, const int64_t* __restrict__ node_ptr,
                                                                  const int64_t* __restrict__ edge_ptr, const int32_t* __restrict__ src_l,
                                                                  const int32_t* __restrict__ tgt_l, const int32_t* __restrict__ eperm_s,
                                                                  const int32_t* __restrict__ lrowptr_s, int32_t* __restrict__ rowptr_s,
                                                                  int32_t* __restrict__ col_s, int32_t* __restrict__ eid_s,
                                                                  int32_t* __restrict__ src_s, int B, int64_t n_cap) {
    const int g = blockIdx.x;
    const int64_t gid = ids[g];
    const int64_t n0s = node_ptr[gid], nn = node_ptr[gid + 1] - n0s;
    const int64_t e0s = edge_ptr[gid], ne = edge_ptr[gid + 1] - e0s;
    const int64_t no = noff[g], eo = eoff[g];
    for (int64_t j = threadIdx.x; j < nn; j += blockDim.x) rowptr_s[no + j] = (int32_t)(eo + lrowptr_s[n0s + j]);
    const int32_t shift = (int32_t)no;
    for (int64_t k = threadIdx.x; k < ne; k += blockDim.x) {
        const int32_t le = eperm_s[e0s + k];
        eid_s[eo + k] = (int32_t)(eo + le);
        col_s[eo + k] = tgt_l[e0s + le] + shift;
        src_s[eo + k] = src_l[e0s + le] + shift;
    }
    const int64_t N = noff[B];
    const int32_t E = (int32_t)eoff[B];
    if (g == B - 1 && threadIdx.x == 0) rowptr_s[N] = E;
    for (int64_t n = N + (int64_t)g * blockDim.x + threadIdx.x; n < n_cap; n += (int64_t)B * blockDim.x) rowptr_s[n + 1] = E;
}

// Edge slots past the batch's last edge (padded static batches) point at the first padding node: per-edge kernels that run
// over all e_cap slots then read rows whose gradient is exactly zero, so the unused slots contribute nothing.
__global__ __launch_bounds__(256) void pad_edge_tail_kernel(const int64_t* __restrict__ noff, const int64_t* __restrict__ eoff, int B,
                                                            int64_t n_cap, int64_t e_cap, int32_t* __restrict__ src,
                                                            int32_t* __restrict__ tgt, int32_t* __restrict__ col_s,
                                                            int32_t* __restrict__ eid_s, int32_t* __restrict__ src_s) {
    const int64_t N = noff[B], E = eoff[B];
    const int32_t pad = (int32_t)(N < n_cap ? N : n_cap - 1);
    for (int64_t e = E + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e_cap; e += (int64_t)gridDim.x * blockDim.x) {
        src[e] = pad;
        tgt[e] = pad;
        if (col_s) { col_s[e] = pad; src_s[e] = pad; eid_s[e] = (int32_t)e; }
    }
}

}  // namespace mdl

extern "C" int mdl_assemble_transposed(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                                       const int64_t* edge_ptr, const int32_t* src_l, const int32_t* tgt_l,
                                       const int32_t* eperm_s, const int32_t* lrowptr_s, int32_t* rowptr_s, int32_t* col_s,
                                       int32_t* eid_s, int32_t* src_s, int B, int64_t n_cap, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(B >= 1 && ids && noff && eoff && node_ptr && edge_ptr && src_l && tgt_l && eperm_s && lrowptr_s && rowptr_s &&
                    col_s && eid_s && src_s, MDL_E_ARG, "mdl_assemble_transposed: bad arguments");
    hipLaunchKernelGGL(assemble_transposed_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, ids, noff, eoff, node_ptr,
                       edge_ptr, src_l, tgt_l, eperm_s, lrowptr_s, rowptr_s, col_s, eid_s, src_s, B, n_cap);
    return check_launch("mdl_assemble_transposed");
}

extern "C" int mdl_pad_edge_tail(const int64_t* noff, const int64_t* eoff, int B, int64_t n_cap, int64_t e_cap, int32_t* src,
                                 int32_t* tgt, int32_t* col_s, int32_t* eid_s, int32_t* src_s, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(noff && eoff && src && tgt && B >= 1 && n_cap >= 1 && e_cap >= 0 && (!col_s || (eid_s && src_s)), MDL_E_ARG,
                "mdl_pad_edge_tail: bad arguments");
    hipLaunchKernelGGL(pad_edge_tail_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, noff, eoff, B, n_cap, e_cap, src, tgt,
                       col_s, eid_s, src_s);
    return check_launch("mdl_pad_edge_tail");
}

extern "C" int mdl_pad_batch_tail(const int64_t* noff, const int64_t* eoff, int B, int64_t n_cap, int32_t* rowptr,
                                  int64_t* batch, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(noff && eoff && rowptr && batch && B >= 1 && n_cap >= 1, MDL_E_ARG, "mdl_pad_batch_tail: bad arguments");
    hipLaunchKernelGGL(pad_tail_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, noff, eoff, B, n_cap, rowptr, batch);
    return check_launch("mdl_pad_batch_tail");
}

extern "C" int mdl_assemble_batch(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                                  const int64_t* edge_ptr, const float* x_all, const int32_t* src_l, const int32_t* tgt_l,
                                  const float* dist, const float* dist_norm, const int32_t* lrowptr, const float* y_all,
                                  void* x, int64_t* batch, int32_t* rowptr, int32_t* src, int32_t* tgt, float* ew, float* dn,
                                  float* y, int B, int F, int T, int target_index, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(B >= 1 && F >= 1 && T >= 1 && target_index >= 0 && target_index < T, MDL_E_ARG, "mdl_assemble_batch: bad sizes");
    MDL_REQUIRE(ids && noff && eoff && node_ptr && edge_ptr && x_all && lrowptr && y_all && x && batch && rowptr && y,
                MDL_E_ARG, "mdl_assemble_batch: null pointer");
    AsmParams p = {ids, noff, eoff, node_ptr, edge_ptr, x_all, src_l, tgt_l, dist, dist_norm, lrowptr, y_all,
                   x, batch, rowptr, src, tgt, ew, dn, y, F, T, target_index, B, 0, 0, nullptr, nullptr, nullptr, nullptr};
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_F32) hipLaunchKernelGGL((assemble_kernel<float>), dim3((unsigned)B), dim3(256), 0, st, p);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((assemble_kernel<bf16_t>), dim3((unsigned)B), dim3(256), 0, st, p);
    else { set_error("mdl_assemble_batch: unsupported dtype %d", dtype); return MDL_E_UNSUPP; }
    return check_launch("mdl_assemble_batch");
}

// mdl_assemble_batch + mdl_pad_batch_tail + mdl_pad_edge_tail (+ the int32 copy of `batch` the pooling index reads) as ONE launch
extern "C" int mdl_assemble_batch_padded(const int64_t* ids, const int64_t* noff, const int64_t* eoff, const int64_t* node_ptr,
                                         const int64_t* edge_ptr, const float* x_all, const int32_t* src_l, const int32_t* tgt_l,
                                         const float* dist, const float* dist_norm, const int32_t* lrowptr, const float* y_all,
                                         void* x, int64_t* batch, int32_t* rowptr, int32_t* src, int32_t* tgt, float* ew, float* dn,
                                         float* y, int B, int F, int T, int target_index, int dtype, int64_t n_cap, int64_t e_cap,
                                         int32_t* pool_seg, int32_t* col_s, int32_t* eid_s, int32_t* src_s, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(B >= 1 && F >= 1 && T >= 1 && target_index >= 0 && target_index < T && n_cap >= 1 && e_cap >= 0, MDL_E_ARG,
                "mdl_assemble_batch_padded: bad sizes");
    MDL_REQUIRE(ids && noff && eoff && node_ptr && edge_ptr && x_all && lrowptr && y_all && x && batch && rowptr && y && src && tgt &&
                    (!col_s || (eid_s && src_s)), MDL_E_ARG, "mdl_assemble_batch_padded: null pointer");
    AsmParams p = {ids, noff, eoff, node_ptr, edge_ptr, x_all, src_l, tgt_l, dist, dist_norm, lrowptr, y_all,
                   x, batch, rowptr, src, tgt, ew, dn, y, F, T, target_index, B, n_cap, e_cap, pool_seg, col_s, eid_s, src_s};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(B + ASM_TAIL_BLOCKS));
    if (dtype == MDL_F32) hipLaunchKernelGGL((assemble_kernel<float>), grid, dim3(256), 0, st, p);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((assemble_kernel<bf16_t>), grid, dim3(256), 0, st, p);
    else { set_error("mdl_assemble_batch_padded: unsupported dtype %d", dtype); return MDL_E_UNSUPP; }
    return check_launch("mdl_assemble_batch_padded");
}
