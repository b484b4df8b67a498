// nnconv.hip — K7: the per-edge contraction of NNConv without the E x C x C weight tensor.
//
// torch_geometric.nn.NNConv(in, out, nn, aggr="mean") as constructed at /root/reference/matdeeplearn/models/mpnn.py:83-88
// and called at :148-157 computes, per edge e = (j -> i), the message  m_e = x_j^T . reshape(nn(e_attr), [C_in, C_out])
// where the last layer of `nn` is Linear(d3, C_in*C_out): the reference path materialises a C_in x C_out matrix PER EDGE
// (40 KB at the MPNN_demo sizes C = 100; 400 MB per layer for the 10 k edges of one default batch).  Re-associated:
//
//     m_e[o] = sum_k h_e[k] * Y_j[o, k] + Z_j[o],      h_e = nn[:-1](e_attr)           [d3]
//     Y_j[o, k] = sum_i x_j[i] * W2[i*C_out + o, k]     = (x @ W2.view(C_in, C_out*d3))[j]     one dense GEMM over NODES
//     Z_j[o]    = sum_i x_j[i] * b2[i*C_out + o]        = (x @ b2.view(C_in, C_out))[j]
//
// so the per-edge work drops from 2*C_in*C_out*d3 flops (2 MFLOP) to 2*C_out*d3 (20 kFLOP): a batched mat-vec that is
// bound by reading Y (once per SOURCE node when the edges are walked by source).  The dense GEMMs stay library GEMMs
// (MFMA); this file holds the edge part:
//
//   mdl_nnconv_msg_fwd : m[eid, :] = Y[j] (C_out x d3) . h[eid, :]           for every out-edge eid of source node j
//   mdl_nnconv_msg_bwd : dh[eid, :] = Y[j]^T . dm[eid, :],   dY[j] = sum_{eid out of j} dm[eid] (x) h[eid]
//
// One workgroup per source node: Y_j is staged ONCE in LDS (row stride d3+1 words: conflict-free both along k and along
// o), the node's out-edges (<= 13 in the reference's graphs, any number here) are walked one by one with h / dm
// broadcast from LDS.  Plain VALU fp32 accumulation — "MFMA only in the dense GEMMs" (BASELINE north_star).
// Algorithmic bytes: N*Co*d3*s (Y, read once) + E*(d3 + Co)*s; backward adds the dY write and the dm / dh rows.
#include "mdl_common.h"

namespace mdl {

template <typename T>
__global__ __launch_bounds__(256) void nnconv_msg_fwd_kernel(const T* __restrict__ Y, const T* __restrict__ h,
                                                             const int32_t* __restrict__ rowptr_s,
                                                             const int32_t* __restrict__ eid_s, T* __restrict__ m, int Co,
                                                             int D3) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = D3 + 1;
    float* Ys = sm;                  // [Co][LD]
    float* hs = sm + Co * LD;        // [D3]
    const int j = blockIdx.x;
    const int b = rowptr_s[j], e = rowptr_s[j + 1];
    if (b == e) return;
    const T* Yj = Y + (int64_t)j * Co * D3;
    for (int q = threadIdx.x; q < Co * D3; q += blockDim.x) {
        const int o = q / D3, k = q - o * D3;
        Ys[o * LD + k] = Elem<T>::ld(Yj + q);
    }
    for (int s = b; s < e; ++s) {
        const int64_t eid = eid_s ? eid_s[s] : s;
        __syncthreads();                                   // Ys staged / previous edge's hs consumed
        for (int k = threadIdx.x; k < D3; k += blockDim.x) hs[k] = Elem<T>::ld(h + eid * D3 + k);
        __syncthreads();
        for (int o = threadIdx.x; o < Co; o += blockDim.x) {
            const float* yr = Ys + o * LD;
            float a0 = 0.0f, a1 = 0.0f;
            int k = 0;
            for (; k + 1 < D3; k += 2) { a0 = fmaf(yr[k], hs[k], a0); a1 = fmaf(yr[k + 1], hs[k + 1], a1); }
            if (k < D3) a0 = fmaf(yr[k], hs[k], a0);
            Elem<T>::st(m + eid * Co + o, a0 + a1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void nnconv_msg_bwd_kernel(const T* __restrict__ Y, const T* __restrict__ h,
                                                             const T* __restrict__ dm, const int32_t* __restrict__ rowptr_s,
                                                             const int32_t* __restrict__ eid_s, T* __restrict__ dh,
                                                             T* __restrict__ dY, int Co, int D3) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int LD = D3 + 1;
    float* Ys = sm;                  // [Co][LD]  Y_j, then the dY_j accumulator is kept in registers (below)
    float* hs = sm + Co * LD;        // [D3]
    float* gs = hs + D3;             // [Co]
    const int j = blockIdx.x;
    const int b = rowptr_s[j], e = rowptr_s[j + 1];
    T* dYj = dY + (int64_t)j * Co * D3;
    const int total = Co * D3;
    if (b == e) {                                          // node without out-edges: dY_j = 0
        for (int q = threadIdx.x; q < total; q += blockDim.x) Elem<T>::st(dYj + q, 0.0f);
        return;
    }
    const T* Yj = Y + (int64_t)j * Co * D3;
    for (int q = threadIdx.x; q < total; q += blockDim.x) {
        const int o = q / D3, k = q - o * D3;
        Ys[o * LD + k] = Elem<T>::ld(Yj + q);
    }
    // dY_j[o, k] accumulators: element q = threadIdx.x + r*blockDim.x (r < RMAX); larger problems take a second pass
    constexpr int RMAX = 48;                               // 256 threads x 48 = 12288 >= 100 x 100
    float acc[RMAX];
    for (int q0 = 0; q0 < total; q0 += RMAX * (int)blockDim.x) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) acc[r] = 0.0f;
        for (int s = b; s < e; ++s) {
            const int64_t eid = eid_s ? eid_s[s] : s;
            __syncthreads();
            for (int k = threadIdx.x; k < D3; k += blockDim.x) hs[k] = Elem<T>::ld(h + eid * D3 + k);
            for (int o = threadIdx.x; o < Co; o += blockDim.x) gs[o] = Elem<T>::ld(dm + eid * Co + o);
            __syncthreads();
            if (q0 == 0) {                                  // dh[eid, k] = sum_o Y_j[o, k] * dm[eid, o]   (first pass only)
                for (int k = threadIdx.x; k < D3; k += blockDim.x) {
                    float a0 = 0.0f, a1 = 0.0f;
                    int o = 0;
                    for (; o + 1 < Co; o += 2) { a0 = fmaf(Ys[o * LD + k], gs[o], a0); a1 = fmaf(Ys[(o + 1) * LD + k], gs[o + 1], a1); }
                    if (o < Co) a0 = fmaf(Ys[o * LD + k], gs[o], a0);
                    Elem<T>::st(dh + eid * D3 + k, a0 + a1);
                }
            }
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int q = q0 + threadIdx.x + r * (int)blockDim.x;
                if (q < total) {
                    const int o = q / D3, k = q - o * D3;
                    acc[r] = fmaf(gs[o], hs[k], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const int q = q0 + threadIdx.x + r * (int)blockDim.x;
            if (q < total) Elem<T>::st(dYj + q, acc[r]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16, Co <= 128, D3 <= 128 (the reference's sizes: 100 x 100): the same contraction on MFMA.  One workgroup (4 waves)
// per source node; Y_j (Co x D3, 20 KB) is staged ONCE in LDS as a zero-padded [128][136] bf16 tile with coalesced dword
// loads (wave = row, lane = dword), the node's out-edges are taken 32 at a time: their h (and dm) rows are gathered into
// [32][136] tiles, and
//   forward   m  [edges x Co] = H  . Y^T     A = H rows,  B = Y rows                      (wave w: output columns 32w..)
//   backward  dh [edges x D3] = DM . Y       A = DM rows, B = Y read k-major (transpose read)
//             dY [Co x D3]   += DM^T . H     both operands k-major over the EDGES (transpose reads), kept in registers
// The scalar kernels above move 2 bytes per load while staging Y and keep 100 of 256 threads busy in the dot products:
// 3.2 ms (forward) / 7.5 ms (backward) per layer on 6e4 nodes / 8e5 edges, 0.06 of the HBM roofline.
#ifndef MDL_K7_BWD_WGS
#define MDL_K7_BWD_WGS 2     // workgroups per CU the backward is register-allocated for (LDS allows 3)
#endif
#ifndef MDL_K7_BWD_SMALLC
#define MDL_K7_BWD_SMALLC 5  // chunks per thread of the small-block form (see mdl_nnconv_msg_bwd)
#endif
constexpr int NM_KP = 128, NM_LD = NM_KP + 8;
typedef __attribute__((ext_vector_type(4))) short nm_s16x4;
typedef __attribute__((address_space(3))) nm_s16x4* nm_lds4_t;

// fragment whose 8 K slots are ROWS 16*ks + 8*h + q of a row-major LDS tile, for column ct*32 + i  (see gemm_tn.hip)
__device__ __forceinline__ bf16x8 nm_tr_frag(const bf16_t* tile, int ks, int ct, int i, int h) {
    const int t = i & 15;
    const bf16_t* p = tile + (16 * ks + 8 * h + (t >> 2)) * NM_LD + ct * 32 + (i & 16) + 4 * (t & 3);
    const nm_s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nm_lds4_t)p);
    const nm_s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((nm_lds4_t)(p + 4 * NM_LD));
    return bf16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}
// fragment whose 8 K slots are COLUMNS 16*ks + 8*h + q of row `row`
__device__ __forceinline__ bf16x8 nm_row_frag(const bf16_t* tile, int row, int ks, int h) {
    return *reinterpret_cast<const bf16x8*>(tile + row * NM_LD + 16 * ks + 8 * h);
}
// rows [0, rows) x `width` bf16 of a dense row-major matrix at `src` -> zero-padded LDS tile [nrows][NM_LD]
// (wave wv takes rows wv, wv+4, ...; lane = dword of the row; rows >= rows and dwords >= width/2 become zeros)
__device__ __forceinline__ void nm_stage_dense(const bf16_t* __restrict__ src, int rows, int width, bf16_t* tile, int nrows,
                                               int wv, int lane) {
    const int w2 = width >> 1;
    for (int r0 = wv; r0 < nrows; r0 += 32) {
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            const bool ok = r < rows && lane < w2;
            v[u] = *reinterpret_cast<const unsigned*>(src + (int64_t)(ok ? r : 0) * width + 2 * (ok ? lane : 0));
            if (!ok) v[u] = 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            if (r < nrows) *reinterpret_cast<unsigned*>(tile + r * NM_LD + 2 * lane) = v[u];
        }
    }
}
// The same tile from a DENSE [rows][width] block whose bytes form one 16-byte-aligned run (Y_j: Co*D3*2 bytes at a multiple of
// it): the block is read as flat 16-byte chunks, ALL of a thread's chunks requested before the first one is used (one memory
// round trip per workgroup instead of one per 8-row trip: the kernel is a chain of such round trips, one workgroup per node),
// and scattered dword by dword into the padded rows; (row, column) of a dword from a multiplication by ceil(2^32 / (width / 2)).
// Rows [rows, nrows) and the columns [width, kpad) of every row are zero-filled here (MFMA operands past the matrix).
template <int MAXC>      // chunks per thread (256 threads x MAXC x 16 B >= rows * width * 2)
struct NmFlat {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    u4 v[MAXC];
    // request every chunk of the block (no use of the data: the caller issues other loads behind these)
    __device__ __forceinline__ void issue(const bf16_t* __restrict__ src, int rows, int width, int tid) {
        const int nd = rows * (width >> 1), nch = (nd + 3) >> 2;          // dwords, 16-byte chunks (the last one may be partial)
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const int c = tid + 256 * u;
            // (a chunk that would run past the block is re-read from its start and masked in commit: no load past the matrix)
            v[u] = *reinterpret_cast<const u4*>(src + 8 * (int64_t)(c < nch && 4 * c + 3 < nd ? c : 0));
        }
    }
    __device__ __forceinline__ void commit(const bf16_t* __restrict__ src, int rows, int width, unsigned w2_inv, bf16_t* tile,
                                           int nrows, int kpad, int tid) {
        const int w2 = width >> 1, nd = rows * w2, nch = (nd + 3) >> 2;
        // zero fill of the padding first (the loads are still flying)
        for (int q = tid; q < (nrows - rows) * (NM_LD / 2); q += 256)
            reinterpret_cast<unsigned*>(tile + rows * NM_LD)[q] = 0u;
        const int pw = (kpad - width) >> 1;
        for (int q = tid; q < rows * pw; q += 256) {
            const int r = q / max(pw, 1), c2 = q - r * pw;
            reinterpret_cast<unsigned*>(tile + r * NM_LD + width)[c2] = 0u;
        }
#pragma unroll
        for (int u = 0; u < MAXC; ++u) {
            const int c = tid + 256 * u;
            if (c < nch && 4 * c + 3 < nd) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned d = 4u * c + q;
                    const unsigned r = __umulhi(d, w2_inv), col = d - r * (unsigned)w2;
                    *reinterpret_cast<unsigned*>(tile + r * NM_LD + 2 * col) = v[u][q];
                }
            }
        }
        // a partial last chunk (rows * width not a multiple of 8): dword loads
        if ((nd & 3) && tid < (nd & 3)) {
            const unsigned d = (unsigned)(nd & ~3) + tid;
            const unsigned r = __umulhi(d, w2_inv), col = d - r * (unsigned)w2;
            *reinterpret_cast<unsigned*>(tile + r * NM_LD + 2 * col) = *reinterpret_cast<const unsigned*>(src + 2 * (int64_t)d);
        }
    }
};

// gathered rows: row r of the tile = src[ids[r], 0:width] (ids[r] < 0: zeros); 32 rows, 8 per wave
__device__ __forceinline__ void nm_stage_rows(const bf16_t* __restrict__ src, const int* ids, int width, bf16_t* tile, int wv,
                                              int lane) {
    const int w2 = width >> 1;
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int id = ids[wv + 4 * u];
        const bool ok = id >= 0 && lane < w2;
        v[u] = *reinterpret_cast<const unsigned*>(src + (int64_t)(ok ? id : 0) * width + 2 * (ok ? lane : 0));
        if (!ok) v[u] = 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<unsigned*>(tile + (wv + 4 * u) * NM_LD + 2 * lane) = v[u];
}

struct NmRows {
    unsigned v[8];
    __device__ __forceinline__ void issue(const bf16_t* __restrict__ src, const int* ids, int width, int wv, int lane) {
        const int w2 = width >> 1;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = ids[wv + 4 * u];
            const bool ok = id >= 0 && lane < w2;
            v[u] = *reinterpret_cast<const unsigned*>(src + (int64_t)(ok ? id : 0) * width + 2 * (ok ? lane : 0));
            if (!ok) v[u] = 0u;
        }
    }
    __device__ __forceinline__ void commit(bf16_t* tile, int wv, int lane) {
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<unsigned*>(tile + (wv + 4 * u) * NM_LD + 2 * lane) = v[u];
    }
};

__global__ __launch_bounds__(256, 2) void nnconv_msg_fwd_mfma_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ h,
                                                                     const int32_t* __restrict__ rowptr_s,
                                                                     const int32_t* __restrict__ eid_s, bf16_t* __restrict__ m,
                                                                     int Co, int D3, unsigned w2_inv, int flat, int yrows) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    // yrows: rows of the Y_j tile that are allocated — 128, or (flat staging) only the Co rows that exist: the fragments of the
    // last 32-column block then run over into the tiles behind (finite garbage in output columns >= Co, which are not stored),
    // and 36 KB instead of 44 KB of LDS let four workgroups share a CU instead of three
    bf16_t* Ys = reinterpret_cast<bf16_t*>(smem_c);          // [yrows][LD]
    bf16_t* Hs = Ys + yrows * NM_LD;                          // [32][LD]
    int* ids = reinterpret_cast<int*>(Hs + 32 * NM_LD);       // [32]
    const int j = blockIdx.x;
    const int b = rowptr_s[j], e = rowptr_s[j + 1];
    if (b == e) return;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, hh = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KS = (D3 + 15) >> 4;
    const bf16_t* const Yj = Y + (int64_t)j * Co * D3;
    // Round trips of a workgroup: the edge ids first, Y_j's chunks behind them, the gathered rows as soon as the ids are in
    // LDS — the rows travel while Y_j is still arriving, and Y_j is committed to LDS last (waits complete in issue order)
    int my_id = -1;
    if (tid < 32) my_id = tid < min(32, e - b) ? (eid_s ? eid_s[b + tid] : b + tid) : -1;
    NmFlat<7> yf;
    if (flat) yf.issue(Yj, Co, D3, tid);
    for (int c0 = b; c0 < e; c0 += 32) {
        const int cnt = min(32, e - c0);
        if (tid < 32) ids[tid] = c0 == b ? my_id : (tid < cnt ? (eid_s ? eid_s[c0 + tid] : c0 + tid) : -1);
        __syncthreads();                                      // ids visible
        NmRows hr;
        hr.issue(h, ids, D3, wv, lane);
        if (c0 == b) {
            if (flat) yf.commit(Yj, Co, D3, w2_inv, Ys, yrows, 16 * KS, tid);
            else nm_stage_dense(Yj, Co, D3, Ys, 128, wv, lane);
        }
        hr.commit(Hs, wv, lane);
        __syncthreads();
        if (wv * 32 < Co) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nm_row_frag(Hs, i, ks, hh), nm_row_frag(Ys, wv * 32 + i, ks, hh), acc, 0, 0, 0);
            const int col = wv * 32 + i;
            if (col < Co) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int id = ids[(r & 3) + 8 * (r >> 2) + 4 * hh];
                    if (id >= 0) m[(int64_t)id * Co + col] = f2bf(acc[r]);
                }
            }
        }
        __syncthreads();                                      // tile and ids are rewritten by the next chunk
    }
}

template <int MAXC>      // 16-byte chunks of Y_j per thread (5 covers 100 x 100, 7 the largest block: 128 x 112)
__global__ __launch_bounds__(256, MAXC <= 5 ? 3 : MDL_K7_BWD_WGS) void nnconv_msg_bwd_mfma_kernel(const bf16_t* __restrict__ Y, const bf16_t* __restrict__ h,
                                                                     const bf16_t* __restrict__ dm,
                                                                     const int32_t* __restrict__ rowptr_s,
                                                                     const int32_t* __restrict__ eid_s, bf16_t* __restrict__ dh,
                                                                     bf16_t* __restrict__ dY, int Co, int D3, unsigned w2_inv, int flat) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    bf16_t* Ys = reinterpret_cast<bf16_t*>(smem_c);          // [128][LD]
    bf16_t* Hs = Ys + 128 * NM_LD;                            // [32][LD]
    bf16_t* Ds = Hs + 32 * NM_LD;                             // [32][LD]  dm rows
    int* ids = reinterpret_cast<int*>(Ds + 32 * NM_LD);
    const int j = blockIdx.x;
    const int b = rowptr_s[j], e = rowptr_s[j + 1];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, hh = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16_t* dYj = dY + (int64_t)j * Co * D3;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    if (b == e) {                                             // node without out-edges: dY_j = 0
        if (flat) {
            for (int q = tid; q < (Co * D3) / 8; q += 256) reinterpret_cast<u4*>(dYj)[q] = u4{0u, 0u, 0u, 0u};
            for (int q = ((Co * D3) / 8) * 4 + tid; q < (Co * D3) / 2; q += 256) reinterpret_cast<unsigned*>(dYj)[q] = 0u;
        } else {
            for (int q = tid; q < (Co * D3) / 2; q += 256) reinterpret_cast<unsigned*>(dYj)[q] = 0u;
        }
        return;
    }
    const int OS = (Co + 15) >> 4;
    // (the transposed reads of Ys take the o index up to 16 OS: rows [Co, 16 OS) must read as zeros — the flat staging zeroes
    // rows [Co, 128) —, its column index up to 127: columns >= D3 only reach dh columns that are never stored)
    const bf16_t* const Yj = Y + (int64_t)j * Co * D3;
    int my_id = -1;
    if (tid < 32) my_id = tid < min(32, e - b) ? (eid_s ? eid_s[b + tid] : b + tid) : -1;
    NmFlat<MAXC> yf;
    if (flat) yf.issue(Yj, Co, D3, tid);
    f32x16 accY[4];                                           // wave wv: rows o = 32 wv + ..., column tiles kt = 0..3
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accY[kt][r] = 0.0f;
    for (int c0 = b; c0 < e; c0 += 32) {
        const int cnt = min(32, e - c0);
        if (tid < 32) ids[tid] = c0 == b ? my_id : (tid < cnt ? (eid_s ? eid_s[c0 + tid] : c0 + tid) : -1);
        __syncthreads();
        NmRows hr, dr;
        hr.issue(h, ids, D3, wv, lane);
        dr.issue(dm, ids, Co, wv, lane);
        if (c0 == b) {
            if (flat) yf.commit(Yj, Co, D3, w2_inv, Ys, 128, (D3 + 15) & ~15, tid);
            else nm_stage_dense(Yj, Co, D3, Ys, 128, wv, lane);
        }
        hr.commit(Hs, wv, lane);
        dr.commit(Ds, wv, lane);
        __syncthreads();
        if (wv * 32 < D3) {                                   // dh tile: columns k = 32 wv + i
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            for (int os = 0; os < OS; ++os)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nm_row_frag(Ds, i, os, hh), nm_tr_frag(Ys, os, wv, i, hh), acc, 0, 0, 0);
            const int col = wv * 32 + i;
            if (col < D3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int id = ids[(r & 3) + 8 * (r >> 2) + 4 * hh];
                    if (id >= 0) dh[(int64_t)id * D3 + col] = f2bf(acc[r]);
                }
            }
        }
        if (wv * 32 < Co) {                                   // dY rows o = 32 wv + ..: contraction over the chunk's edges
#pragma unroll
            for (int es = 0; es < 2; ++es) {
                const bf16x8 a = nm_tr_frag(Ds, es, wv, i, hh);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
                    if (kt * 32 < D3) accY[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, nm_tr_frag(Hs, es, kt, i, hh), accY[kt], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!flat) {
        if (wv * 32 < Co) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int k = kt * 32 + i;
                if (k < D3) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int o = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (o < Co) dYj[o * D3 + k] = f2bf(accY[kt][r]);
                    }
                }
            }
        }
        return;
    }
    // dY_j leaves as ONE contiguous run of 16-byte stores: the accumulators (rows o in registers, column k = lane) are laid
    // down densely ([Co][D3] bf16) in the LDS space of the Y tile (every wave is past its last read of it: the loop ends with
    // a barrier) and copied out chunk by chunk — 64 two-byte store instructions per lane before
    bf16_t* const od = Ys;
    if (wv * 32 < Co) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int k = kt * 32 + i;
            if (k < D3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (o < Co) od[o * D3 + k] = f2bf(accY[kt][r]);
                }
            }
        }
    }
    __syncthreads();
    const int nch = (Co * D3) >> 3;
    for (int c = tid; c < nch; c += 256) reinterpret_cast<u4*>(dYj)[c] = reinterpret_cast<const u4*>(od)[c];
    for (int q = nch * 4 + tid; q < (Co * D3) >> 1; q += 256) reinterpret_cast<unsigned*>(dYj)[q] = reinterpret_cast<const unsigned*>(od)[q];
}

static bool nm_ok(int Co, int D3, int dtype, const void* a, const void* b) {
    return dtype == MDL_BF16 && Co <= 128 && D3 <= 128 && Co >= 2 && D3 >= 2 && Co % 2 == 0 && D3 % 2 == 0 &&
           reinterpret_cast<uintptr_t>(a) % 4 == 0 && reinterpret_cast<uintptr_t>(b) % 4 == 0;
}

static int nn_check(const char* name, int64_t N, int Co, int D3, int dtype) {
    MDL_REQUIRE(dtype == MDL_F32 || dtype == MDL_BF16, MDL_E_UNSUPP, "%s: unsupported dtype %d", name, dtype);
    MDL_REQUIRE(N >= 0 && N < (1ll << 31) && Co >= 1 && D3 >= 1, MDL_E_ARG, "%s: bad sizes", name);
    MDL_REQUIRE(((int64_t)Co * (D3 + 1) + D3 + Co) * 4 <= 160 * 1024, MDL_E_UNSUPP,
                "%s: Y_j (%d x %d) does not fit in LDS", name, Co, D3);
    return MDL_OK;
}

}  // namespace mdl

extern "C" int mdl_nnconv_msg_fwd(const void* Y, const void* h, const int32_t* rowptr_s, const int32_t* eid_s, void* m,
                                  int64_t N, int Co, int D3, int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = nn_check("mdl_nnconv_msg_fwd", N, Co, D3, dtype);
    if (rc) return rc;
    if (N == 0) return MDL_OK;
    MDL_REQUIRE(Y && h && rowptr_s && m, MDL_E_ARG, "mdl_nnconv_msg_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (nm_ok(Co, D3, dtype, Y, h) && reinterpret_cast<uintptr_t>(m) % 2 == 0) {
        auto kf = nnconv_msg_fwd_mfma_kernel;
        // flat staging of Y_j: every node's block must start on a 16-byte boundary and fit the per-thread chunk budget
        // (D3 >= 4: the staging splits a dword index by w2 = D3 / 2 with a 32-bit reciprocal, and 2^32 / 1 does not fit one)
        const int flat = (D3 >= 4 && ((int64_t)Co * D3 * 2) % 16 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0 && Co * D3 <= 256 * 7 * 8) ? 1 : 0;
#ifndef MDL_K7_YROWS_FULL
        const int yrows = flat ? Co : 128;
#else
        const int yrows = 128;
#endif
        const int lds_m = (yrows + 32) * NM_LD * 2 + 32 * 4;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), (128 + 32) * NM_LD * 2 + 32 * 4);
        const unsigned w2_inv = (unsigned)((0x100000000ull + (D3 / 2) - 1) / (D3 / 2));
        hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds_m, st, (const bf16_t*)Y, (const bf16_t*)h, rowptr_s, eid_s, (bf16_t*)m, Co, D3,
                           w2_inv, flat, yrows);
        return check_launch("mdl_nnconv_msg_fwd");
    }
    const int lds = (Co * (D3 + 1) + D3) * 4;
    if (dtype == MDL_F32) {
        auto kf = nnconv_msg_fwd_kernel<float>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds, st, (const float*)Y, (const float*)h, rowptr_s, eid_s, (float*)m, Co, D3);
    } else {
        auto kf = nnconv_msg_fwd_kernel<bf16_t>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds, st, (const bf16_t*)Y, (const bf16_t*)h, rowptr_s, eid_s, (bf16_t*)m, Co, D3);
    }
    return check_launch("mdl_nnconv_msg_fwd");
}

extern "C" int mdl_nnconv_msg_bwd(const void* Y, const void* h, const void* dm, const int32_t* rowptr_s,
                                  const int32_t* eid_s, void* dh, void* dY, int64_t N, int Co, int D3, int dtype,
                                  mdlStream_t stream) {
    using namespace mdl;
    int rc = nn_check("mdl_nnconv_msg_bwd", N, Co, D3, dtype);
    if (rc) return rc;
    if (N == 0) return MDL_OK;
    MDL_REQUIRE(Y && h && dm && rowptr_s && dh && dY, MDL_E_ARG, "mdl_nnconv_msg_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (nm_ok(Co, D3, dtype, Y, h) && reinterpret_cast<uintptr_t>(dm) % 4 == 0 && reinterpret_cast<uintptr_t>(dY) % 4 == 0) {
        const int lds_m = (128 + 64) * NM_LD * 2 + 32 * 4;
        const int flat = (D3 >= 4 && ((int64_t)Co * D3 * 2) % 16 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0 &&
                          reinterpret_cast<uintptr_t>(dY) % 16 == 0 && Co * D3 <= 256 * 7 * 8) ? 1 : 0;
        const unsigned w2_inv = (unsigned)((0x100000000ull + (D3 / 2) - 1) / (D3 / 2));
        // blocks of up to 100 x 100 (MPNN_demo) fit five chunks per thread: 8 staging registers less, which is what lets the
        // kernel be register-allocated for the three workgroups per CU its LDS allows
        if (flat && Co * D3 <= 256 * MDL_K7_BWD_SMALLC * 8) {
            auto kf = nnconv_msg_bwd_mfma_kernel<MDL_K7_BWD_SMALLC>;
            (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds_m);
            hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds_m, st, (const bf16_t*)Y, (const bf16_t*)h, (const bf16_t*)dm, rowptr_s,
                               eid_s, (bf16_t*)dh, (bf16_t*)dY, Co, D3, w2_inv, flat);
        } else {
            auto kf = nnconv_msg_bwd_mfma_kernel<7>;
            (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds_m);
            hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds_m, st, (const bf16_t*)Y, (const bf16_t*)h, (const bf16_t*)dm, rowptr_s,
                               eid_s, (bf16_t*)dh, (bf16_t*)dY, Co, D3, w2_inv, flat);
        }
        return check_launch("mdl_nnconv_msg_bwd");
    }
    const int lds = (Co * (D3 + 1) + D3 + Co) * 4;
    if (dtype == MDL_F32) {
        auto kf = nnconv_msg_bwd_kernel<float>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds, st, (const float*)Y, (const float*)h, (const float*)dm, rowptr_s, eid_s, (float*)dh, (float*)dY, Co, D3);
    } else {
        auto kf = nnconv_msg_bwd_kernel<bf16_t>;
        (void)set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);
        hipLaunchKernelGGL(kf, dim3((unsigned)N), dim3(256), lds, st, (const bf16_t*)Y, (const bf16_t*)h, (const bf16_t*)dm, rowptr_s, eid_s, (bf16_t*)dh, (bf16_t*)dY, Co, D3);
    }
    return check_launch("mdl_nnconv_msg_bwd");
}
