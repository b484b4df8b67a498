// segment.hip — K5: CSR segmented reduce (sum / mean / max) forward + backward, and CSR rowptr build.
// Restates torch_scatter.scatter(src, index, dim=0, reduce=...) for a SORTED index, as called at
// /root/reference/matdeeplearn/models/megnet.py:86,130-132,342-348 and by PyG global_*_pool
// (/root/reference/matdeeplearn/models/cgcnn.py:154).  Atomic-free and deterministic: one thread
// owns VEC channels of one output row and walks the rows of its segment; consecutive lanes own
// consecutive channel vectors, so every source row is read as one contiguous run.
// Algorithmic bytes: E*C*s (read) + N*C*s (write) + 4*(N+1).
#include "mdl_common.h"

#ifndef MDL_SEG_SMALL
#define MDL_SEG_SMALL 1      // 0: the lane-group kernel for every size (A/B)
#endif

namespace mdl {

__global__ __launch_bounds__(256) void csr_rowptr_kernel(const int32_t* __restrict__ idx, int64_t E, int64_t N,
                                                         int32_t* __restrict__ rowptr) {
    // rowptr[n] = number of entries with idx < n  (lower_bound), n in [0, N]
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n > N) return;
    int64_t lo = 0, hi = E;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if ((int64_t)idx[mid] < n) lo = mid + 1; else hi = mid;
    }
    rowptr[n] = (int32_t)lo;
}

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void seg_fwd_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ perm, T* __restrict__ out,
                                                      int32_t* __restrict__ argmax, int64_t N, int C) {
    const int64_t total = N * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        const int b = rowptr[n], e = rowptr[n + 1];
        float acc = (REDUCE == MDL_MAX) ? -INFINITY : 0.0f;
        int best = -1;
        for (int k = b; k < e; ++k) {
            const int64_t r = perm ? perm[k] : k;
            const float v = Elem<T>::ld(src + r * C + c);
            if (REDUCE == MDL_MAX) {
                if (v > acc || best < 0) { acc = v; best = (int)r; }
            } else {
                acc += v;
            }
        }
        if (REDUCE == MDL_MEAN) acc = acc / (float)max(e - b, 1);
        if (REDUCE == MDL_MAX) {
            if (best < 0) acc = 0.0f;
            argmax[i] = best;
        }
        Elem<T>::st(out + i, acc);
    }
}

template <typename T, int REDUCE>
__global__ __launch_bounds__(256) void seg_bwd_kernel(const T* __restrict__ go, const int32_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ seg,
                                                      const int32_t* __restrict__ perm, T* __restrict__ gs,
                                                      int64_t E, int C) {
    const int64_t total = E * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / C;
        const int c = (int)(i - k * C);
        const int n = seg[k];
        float g = Elem<T>::ld(go + (int64_t)n * C + c);
        if (REDUCE == MDL_MEAN) g = g / (float)max(rowptr[n + 1] - rowptr[n], 1);
        const int64_t r = perm ? perm[k] : k;
        Elem<T>::st(gs + r * C + c, g);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void seg_bwd_max_kernel(const T* __restrict__ go,
                                                          const int32_t* __restrict__ argmax,
                                                          T* __restrict__ gs, int64_t N, int C) {
    const int64_t total = N * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int a = argmax[i];
        if (a >= 0) gs[(int64_t)a * C + (i % C)] = go[i];
    }
}

// 16-byte fast paths (sum / mean, rows in place, C a multiple of the vector width): one thread owns W channels of one
// segment and streams its rows with 16-byte loads, four in flight; the backward writes one 16-byte vector per thread.
template <typename T, int REDUCE, int W>
__global__ __launch_bounds__(256) void seg_fwd_vec_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ perm, T* __restrict__ out, int64_t N,
                                                          int CG) {
    // RL row lanes share one (segment, channel group): lane j takes rows b + j, b + j + RL, ... two at a time, and the
    // partial sums meet through two xor-shuffles.  (One thread per segment walks ~25 rows as a chain of round trips.)
    constexpr int U = 2;
    typedef VecW<T, W> V;
    const int RL = (CG <= 16 && (CG & (CG - 1)) == 0) ? 4 : 1;        // the CG*RL lanes of a group sit in one wavefront
    const int64_t total = N * CG * RL;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x; i0 < total; i0 += stride) {     // block-uniform trip count
        const int64_t i = min(i0 + threadIdx.x, total - 1);
        const int64_t n = i / (CG * RL);
        const int rem = (int)(i - n * (CG * RL));
        const int j = rem / CG, cg = rem - j * CG;
        const int b = rowptr[n], e = rowptr[n + 1];
        float acc[W];
#pragma unroll
        for (int q = 0; q < W; ++q) acc[q] = 0.0f;
        const T* base = src + (int64_t)cg * W;
        const int64_t ld = (int64_t)CG * W;
        for (int k = b + j; k < e; k += U * RL) {
            float v[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                                       // clamp, never guard
                const int kk = min(k + u * RL, e - 1);
                V::ld(base + (int64_t)(perm ? perm[kk] : kk) * ld, v[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k + u * RL < e) {
#pragma unroll
                    for (int q = 0; q < W; ++q) acc[q] += v[u][q];
                }
        }
        if (RL == 4) {
#pragma unroll
            for (int q = 0; q < W; ++q) {
                acc[q] += __shfl_xor(acc[q], CG);
                acc[q] += __shfl_xor(acc[q], 2 * CG);
            }
        }
        if (REDUCE == MDL_MEAN) {
            const float cnt = (float)max(e - b, 1);
#pragma unroll
            for (int q = 0; q < W; ++q) acc[q] = acc[q] / cnt;
        }
        if (j == 0 && i0 + threadIdx.x < total) V::st(out + (n * CG + cg) * W, acc);
    }
}

// FEW segments (a batch of ~100 graphs pooled to graph rows, the reference's batch size: config.yml:136): with one lane group per
// (segment, channel group) the launch is a few thousand threads that each walk their segment's rows as a chain of memory round
// trips (13 for a graph of 26 nodes, 160 for a graph's 330 edges).  Here a WORKGROUP owns a segment: 256 / CG row lanes per
// channel group take the rows b + r, b + r + RL, ... (two in flight each), the partial sums meet in LDS.  Same sums in another order.
template <typename T, int REDUCE, int W>
__global__ __launch_bounds__(256) void seg_fwd_small_kernel(const T* __restrict__ src, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ perm, T* __restrict__ out, int CG) {
    typedef VecW<T, W> V;
    __shared__ float red[256 * W];
    const int n = blockIdx.x, t = threadIdx.x;
    const int RL = 256 / CG;                                   // CG <= 256
    const int cg = t % CG, r = t / CG;
    const int b = rowptr[n], e = rowptr[n + 1];
    float acc[W];
#pragma unroll
    for (int q = 0; q < W; ++q) acc[q] = 0.0f;
    if (r < RL) {
        const T* base = src + (int64_t)cg * W;
        const int64_t ld = (int64_t)CG * W;
        for (int k = b + r; k < e; k += 2 * RL) {
            float v0[W], v1[W];
            const int k1 = min(k + RL, e - 1);
            V::ld(base + (int64_t)(perm ? perm[k] : k) * ld, v0);
            V::ld(base + (int64_t)(perm ? perm[k1] : k1) * ld, v1);
            const bool two = k + RL < e;
#pragma unroll
            for (int q = 0; q < W; ++q) acc[q] += v0[q] + (two ? v1[q] : 0.0f);
        }
    }
#pragma unroll
    for (int q = 0; q < W; ++q) red[t * W + q] = acc[q];
    __syncthreads();
    if (t < CG) {
#pragma unroll
        for (int q = 0; q < W; ++q) acc[q] = 0.0f;
        for (int j = 0; j < RL; ++j)
#pragma unroll
            for (int q = 0; q < W; ++q) acc[q] += red[(j * CG + t) * W + q];
        if (REDUCE == MDL_MEAN) {
            const float cnt = (float)max(e - b, 1);
#pragma unroll
            for (int q = 0; q < W; ++q) acc[q] = acc[q] / cnt;
        }
        V::st(out + ((int64_t)n * CG + t) * W, acc);
    }
}

template <typename T, int REDUCE, int W>
__global__ __launch_bounds__(256) void seg_bwd_vec_kernel(const T* __restrict__ go, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ seg, const int32_t* __restrict__ perm,
                                                          T* __restrict__ gs, int64_t E, int CG, const T* __restrict__ addend) {
    // addend (optional, [E, C] like gs): a second gradient of the same source rows, added on the way out — the sum autograd
    // would form with one more read-read-write pass over [E, C] (mdl_segment_reduce_bwd_add)
    typedef VecW<T, W> V;
    const int64_t total = E * CG;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / CG;
        const int cg = (int)(i - k * CG);
        const int n = seg[k];
        float g[W];
        V::ld(go + ((int64_t)n * CG + cg) * W, g);
        if (REDUCE == MDL_MEAN) {
            const float cnt = (float)max(rowptr[n + 1] - rowptr[n], 1);
#pragma unroll
            for (int j = 0; j < W; ++j) g[j] = g[j] / cnt;
        }
        const int64_t o = ((int64_t)(perm ? perm[k] : k) * CG + cg) * W;
        if (addend) {
            float a[W];
            V::ld(addend + o, a);
#pragma unroll
            for (int j = 0; j < W; ++j) g[j] += a[j];
        }
        V::st(gs + o, g);
    }
}

// vector width of the fast paths: 16 bytes when the row length allows it, else 4 elements (8 bytes of bf16: C = 100, 150)
template <typename T>
static int vec_width(const void* a, const void* b, int64_t C) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    if (C % Vec<T>::W == 0 && al % 16 == 0) return Vec<T>::W;
    if (sizeof(T) == 2 && C % 4 == 0 && al % 8 == 0) return 4;
    return 0;
}

static unsigned grid_for(int64_t total) {
    int64_t b = cdiv(total, 256);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}

template <typename T>
static int seg_fwd(const T* src, const int32_t* rowptr, const int32_t* perm, T* out, int32_t* argmax, int64_t N,
                   int64_t C, int reduce, hipStream_t st) {
    if (N * C == 0) return MDL_OK;
    const int vw = reduce != MDL_MAX ? vec_width<T>(src, out, C) : 0;
    if (vw) {
        const int CG = (int)(C / vw);
        dim3 gv(grid_for(N * CG * 4)), bv(256);
        if (reduce != MDL_SUM && reduce != MDL_MEAN) { set_error("mdl_segment_reduce_fwd: bad reduce %d", reduce); return MDL_E_ARG; }
        // few segments (fewer lane groups than the device has lanes for): a workgroup per segment (seg_fwd_small_kernel)
        if (MDL_SEG_SMALL && N * CG < 65536 && N <= 65535 && CG <= 256) {
            dim3 gs((unsigned)N);
            if constexpr (sizeof(T) == 2) {
                if (vw == 4) {
                    if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_fwd_small_kernel<T, MDL_SUM, 4>), gs, bv, 0, st, src, rowptr, perm, out, CG);
                    else hipLaunchKernelGGL((seg_fwd_small_kernel<T, MDL_MEAN, 4>), gs, bv, 0, st, src, rowptr, perm, out, CG);
                    return check_launch("mdl_segment_reduce_fwd");
                }
            }
            if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_fwd_small_kernel<T, MDL_SUM, Vec<T>::W>), gs, bv, 0, st, src, rowptr, perm, out, CG);
            else hipLaunchKernelGGL((seg_fwd_small_kernel<T, MDL_MEAN, Vec<T>::W>), gs, bv, 0, st, src, rowptr, perm, out, CG);
            return check_launch("mdl_segment_reduce_fwd");
        }
        if constexpr (sizeof(T) == 2) {
            if (vw == 4) {
                if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_fwd_vec_kernel<T, MDL_SUM, 4>), gv, bv, 0, st, src, rowptr, perm, out, N, CG);
                else hipLaunchKernelGGL((seg_fwd_vec_kernel<T, MDL_MEAN, 4>), gv, bv, 0, st, src, rowptr, perm, out, N, CG);
                return check_launch("mdl_segment_reduce_fwd");
            }
        }
        if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_fwd_vec_kernel<T, MDL_SUM, Vec<T>::W>), gv, bv, 0, st, src, rowptr, perm, out, N, CG);
        else hipLaunchKernelGGL((seg_fwd_vec_kernel<T, MDL_MEAN, Vec<T>::W>), gv, bv, 0, st, src, rowptr, perm, out, N, CG);
        return check_launch("mdl_segment_reduce_fwd");
    }
    dim3 g(grid_for(N * C)), b(256);
    switch (reduce) {
        case MDL_SUM: hipLaunchKernelGGL((seg_fwd_kernel<T, MDL_SUM>), g, b, 0, st, src, rowptr, perm, out, argmax, N, (int)C); break;
        case MDL_MEAN: hipLaunchKernelGGL((seg_fwd_kernel<T, MDL_MEAN>), g, b, 0, st, src, rowptr, perm, out, argmax, N, (int)C); break;
        case MDL_MAX: hipLaunchKernelGGL((seg_fwd_kernel<T, MDL_MAX>), g, b, 0, st, src, rowptr, perm, out, argmax, N, (int)C); break;
        default: set_error("mdl_segment_reduce_fwd: bad reduce %d", reduce); return MDL_E_ARG;
    }
    return check_launch("mdl_segment_reduce_fwd");
}

template <typename T>
static int seg_bwd(const T* go, const int32_t* rowptr, const int32_t* seg, const int32_t* perm,
                   const int32_t* argmax, T* gs, int64_t N, int64_t E, int64_t C, int reduce, hipStream_t st,
                   const T* addend = nullptr) {
    dim3 b(256);
    int vw = ((reduce == MDL_SUM || reduce == MDL_MEAN) && E * C != 0) ? vec_width<T>(go, gs, C) : 0;
    if (addend) {
        const int va = vw ? vec_width<T>(addend, gs, C) : 0;
        if (va < vw) vw = va;
        if (!vw) { set_error("mdl_segment_reduce_bwd_add: needs sum / mean and rows of a vector multiple (8-byte aligned)"); return MDL_E_UNSUPP; }
    }
    if (vw) {
        const int CG = (int)(C / vw);
        dim3 gv(grid_for(E * CG));
        if constexpr (sizeof(T) == 2) {
            if (vw == 4) {
                if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_bwd_vec_kernel<T, MDL_SUM, 4>), gv, b, 0, st, go, rowptr, seg, perm, gs, E, CG, addend);
                else hipLaunchKernelGGL((seg_bwd_vec_kernel<T, MDL_MEAN, 4>), gv, b, 0, st, go, rowptr, seg, perm, gs, E, CG, addend);
                return check_launch("mdl_segment_reduce_bwd");
            }
        }
        if (reduce == MDL_SUM) hipLaunchKernelGGL((seg_bwd_vec_kernel<T, MDL_SUM, Vec<T>::W>), gv, b, 0, st, go, rowptr, seg, perm, gs, E, CG, addend);
        else hipLaunchKernelGGL((seg_bwd_vec_kernel<T, MDL_MEAN, Vec<T>::W>), gv, b, 0, st, go, rowptr, seg, perm, gs, E, CG, addend);
        return check_launch("mdl_segment_reduce_bwd");
    }
    switch (reduce) {
        case MDL_SUM:
            if (E * C == 0) return MDL_OK;
            hipLaunchKernelGGL((seg_bwd_kernel<T, MDL_SUM>), dim3(grid_for(E * C)), b, 0, st, go, rowptr, seg, perm, gs, E, (int)C);
            break;
        case MDL_MEAN:
            if (E * C == 0) return MDL_OK;
            hipLaunchKernelGGL((seg_bwd_kernel<T, MDL_MEAN>), dim3(grid_for(E * C)), b, 0, st, go, rowptr, seg, perm, gs, E, (int)C);
            break;
        case MDL_MAX:
            if (N * C == 0) return MDL_OK;
            hipLaunchKernelGGL((seg_bwd_max_kernel<T>), dim3(grid_for(N * C)), b, 0, st, go, argmax, gs, N, (int)C);
            break;
        default: set_error("mdl_segment_reduce_bwd: bad reduce %d", reduce); return MDL_E_ARG;
    }
    return check_launch("mdl_segment_reduce_bwd");
}

}  // namespace mdl

extern "C" int mdl_csr_rowptr(const int32_t* sorted_index, int64_t E, int64_t N, int32_t* rowptr,
                              mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(E >= 0 && N >= 0 && rowptr && (E == 0 || sorted_index), MDL_E_ARG, "mdl_csr_rowptr: bad arguments");
    MDL_REQUIRE(E < (1ll << 31) && N < (1ll << 31), MDL_E_UNSUPP, "mdl_csr_rowptr: int32 index overflow");
    hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)cdiv(N + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                       sorted_index, E, N, rowptr);
    return check_launch("mdl_csr_rowptr");
}

extern "C" int mdl_segment_reduce_fwd(const void* src, const int32_t* rowptr, const int32_t* perm, void* out,
                                      int32_t* argmax, int64_t N, int64_t C, int reduce, int dtype,
                                      mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && C > 0 && C < (1 << 20), MDL_E_ARG, "mdl_segment_reduce_fwd: bad N=%lld C=%lld", (long long)N, (long long)C);
    MDL_REQUIRE(N == 0 || (rowptr && out), MDL_E_ARG, "mdl_segment_reduce_fwd: null pointer");
    MDL_REQUIRE(reduce != MDL_MAX || argmax || N == 0, MDL_E_ARG, "mdl_segment_reduce_fwd: MDL_MAX needs argmax");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_F32) return seg_fwd<float>((const float*)src, rowptr, perm, (float*)out, argmax, N, C, reduce, st);
    if (dtype == MDL_BF16) return seg_fwd<bf16_t>((const bf16_t*)src, rowptr, perm, (bf16_t*)out, argmax, N, C, reduce, st);
    set_error("mdl_segment_reduce_fwd: unsupported dtype %d", dtype);
    return MDL_E_UNSUPP;
}

extern "C" int mdl_segment_reduce_bwd(const void* grad_out, const int32_t* rowptr, const int32_t* seg,
                                      const int32_t* perm, const int32_t* argmax, void* grad_src, int64_t N,
                                      int64_t E, int64_t C, int reduce, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && E >= 0 && C > 0, MDL_E_ARG, "mdl_segment_reduce_bwd: bad sizes");
    MDL_REQUIRE(reduce == MDL_MAX ? (argmax != nullptr || N == 0) : (seg != nullptr || E == 0), MDL_E_ARG,
                "mdl_segment_reduce_bwd: missing seg/argmax");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_F32) return seg_bwd<float>((const float*)grad_out, rowptr, seg, perm, argmax, (float*)grad_src, N, E, C, reduce, st);
    if (dtype == MDL_BF16) return seg_bwd<bf16_t>((const bf16_t*)grad_out, rowptr, seg, perm, argmax, (bf16_t*)grad_src, N, E, C, reduce, st);
    set_error("mdl_segment_reduce_bwd: unsupported dtype %d", dtype);
    return MDL_E_UNSUPP;
}

extern "C" int mdl_segment_reduce_bwd_add(const void* grad_out, const int32_t* rowptr, const int32_t* seg, const int32_t* perm,
                                          const void* addend, void* grad_src, int64_t N, int64_t E, int64_t C, int reduce,
                                          int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && E >= 0 && C > 0, MDL_E_ARG, "mdl_segment_reduce_bwd_add: bad sizes");
    MDL_REQUIRE((reduce == MDL_SUM || reduce == MDL_MEAN) && (E == 0 || (seg && addend && grad_src)), MDL_E_ARG,
                "mdl_segment_reduce_bwd_add: sum / mean only, seg and addend required");
    if (E == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_F32) return seg_bwd<float>((const float*)grad_out, rowptr, seg, perm, nullptr, (float*)grad_src, N, E, C, reduce, st, (const float*)addend);
    if (dtype == MDL_BF16) return seg_bwd<bf16_t>((const bf16_t*)grad_out, rowptr, seg, perm, nullptr, (bf16_t*)grad_src, N, E, C, reduce, st, (const bf16_t*)addend);
    set_error("mdl_segment_reduce_bwd_add: unsupported dtype %d", dtype);
    return MDL_E_UNSUPP;
}
