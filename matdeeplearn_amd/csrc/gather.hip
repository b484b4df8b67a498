// gather.hip — generic gather / edge-weighted gather-reduce kernels used by the SchNet (CFConv), GCN,
// MEGNet and MPNN blocks.  They restate what PyG's MessagePassing.propagate does with ATen
// index_select + elementwise + torch_scatter (call sites: /root/reference/matdeeplearn/models/
// schnet.py:134-143, gcn.py:135-144, megnet.py:41-56,84-101,129-147, mpnn.py:148-157).
//
//   K4a  mdl_gather_mul_reduce:  out[i,:] = reduce_{k in row i} h[col[k],:] * w[eid(k),:] * scale[eid(k)]
//        (CSR rows = aggregation nodes).  One thread owns VEC channels of one output row and walks the
//        row's slots: atomic-free, deterministic; consecutive lanes = consecutive channels, so every
//        h row / w row is read as one contiguous run.  The SAME kernel on the transposed CSR gives the
//        gradient w.r.t. h; mdl_edge_mul gives the gradient w.r.t. w.
//        Algorithmic bytes: E*(2*F*s + 8) + N*(F*s + 4).
//   mdl_gather_rows: out[k,:] = src[idx[k],:]   (index_select along dim 0)
#include "mdl_common.h"

namespace mdl {

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                          T* __restrict__ out, int64_t E, int C) {
    const int64_t total = E * C;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t k = i / C;
        const int c = (int)(i - k * C);
        out[i] = src[(int64_t)idx[k] * C + c];
    }
}

template <typename T, bool MEAN>
__global__ __launch_bounds__(256) void gmr_kernel(const T* __restrict__ h, const T* __restrict__ w,
                                                  const float* __restrict__ scale, const int32_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                                                  T* __restrict__ out, int64_t N, int F) {
    const int64_t total = N * F;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t n = i / F;
        const int c = (int)(i - n * F);
        const int b = rowptr[n], e = rowptr[n + 1];
        float acc = 0.0f;
        for (int k = b; k < e; ++k) {
            const int64_t id = eid ? eid[k] : k;
            float v = Elem<T>::ld(h + (int64_t)col[k] * F + c);
            if (w) v *= Elem<T>::ld(w + id * F + c);
            if (scale) v *= scale[id];
            acc += v;
        }
        if (MEAN) acc = acc / (float)max(e - b, 1);
        Elem<T>::st(out + i, acc);
    }
}

// dw[e,:] = a[ia[e],:] * b[ib[e],:] * scale[e]      (per-edge product of two gathered rows)
template <typename T>
__global__ __launch_bounds__(256) void edge_mul_kernel(const T* __restrict__ a, const int32_t* __restrict__ ia,
                                                       const T* __restrict__ b, const int32_t* __restrict__ ib,
                                                       const float* __restrict__ scale, T* __restrict__ out, int64_t E,
                                                       int F) {
    const int64_t total = E * F;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t e = i / F;
        const int c = (int)(i - e * F);
        float v = Elem<T>::ld(a + (int64_t)ia[e] * F + c) * Elem<T>::ld(b + (int64_t)ib[e] * F + c);
        if (scale) v *= scale[e];
        Elem<T>::st(out + i, v);
    }
}

// ---- two-channel (dword) versions for bf16 rows with an even channel count (SchNet: F = 150 -> 75 dwords per row) ----
// One thread owns one dword of one output row; the threads of a row are consecutive, so every gathered h row and every w
// row is read as one contiguous run; rows are walked four slots at a time with all loads issued before the first use.
// DW (mdl_gather_mul_reduce_dw): the walk over the TRANSPOSED CSR that gives the gradient w.r.t. h also has, per slot,
// everything the gradient w.r.t. w needs — dw[eid,:] = g[tgt,:] * h[src,:] * scale[eid] with h[src] the walked row's OWN h
// row (one dword per thread, loaded once) — and writes it: the separate mdl_edge_mul pass (two more gathered rows per edge,
// 313 us per SchNet layer) disappears.
template <bool MEAN, bool DW = false>
__global__ __launch_bounds__(256) void gmr2_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ w,
                                                   const float* __restrict__ scale, const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col, const int32_t* __restrict__ eid,
                                                   bf16_t* __restrict__ out, int64_t N, int F2,
                                                   const bf16_t* __restrict__ own = nullptr, bf16_t* __restrict__ dw = nullptr) {
    const int npb = (int)blockDim.x / F2;                       // nodes per block
    const int ln = threadIdx.x / F2, d = threadIdx.x - ln * F2;
    const int64_t n = (int64_t)blockIdx.x * npb + ln;
    if (ln >= npb || n >= N) return;
    const unsigned* __restrict__ h2 = reinterpret_cast<const unsigned*>(h);
    const unsigned* __restrict__ w2 = reinterpret_cast<const unsigned*>(w);
    const int b = rowptr[n], e = rowptr[n + 1];
    float a0 = 0.0f, a1 = 0.0f;
    float o0 = 0.0f, o1 = 0.0f;
    if constexpr (DW) {
        const unsigned ov = reinterpret_cast<const unsigned*>(own)[n * F2 + d];
        o0 = __uint_as_float(ov << 16); o1 = __uint_as_float(ov & 0xffff0000u);
    }
    constexpr int U = 4;
    for (int k0 = b; k0 < e; k0 += U) {
        int64_t id[U];
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + u, e - 1);
            id[u] = eid ? eid[k] : k;
            c[u] = col[k];
        }
        unsigned hv[U], wv[U];
        float sc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            hv[u] = h2[(int64_t)c[u] * F2 + d];
            wv[u] = w ? w2[id[u] * F2 + d] : 0x3F803F80u;
            sc[u] = scale ? scale[id[u]] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < e) {
                a0 = fmaf(__uint_as_float(hv[u] << 16) * __uint_as_float(wv[u] << 16), sc[u], a0);
                a1 = fmaf(__uint_as_float(hv[u] & 0xffff0000u) * __uint_as_float(wv[u] & 0xffff0000u), sc[u], a1);
                if constexpr (DW)
                    reinterpret_cast<unsigned*>(dw)[id[u] * F2 + d] =
                        pk_bf16(__uint_as_float(hv[u] << 16) * o0 * sc[u], __uint_as_float(hv[u] & 0xffff0000u) * o1 * sc[u]);
            }
        }
    }
    if (MEAN) { const float inv = 1.0f / (float)max(e - b, 1); a0 *= inv; a1 *= inv; }
    reinterpret_cast<unsigned*>(out)[n * F2 + d] = pk_bf16(a0, a1);
}

__global__ __launch_bounds__(256) void edge_mul2_kernel(const bf16_t* __restrict__ a, const int32_t* __restrict__ ia,
                                                        const bf16_t* __restrict__ b, const int32_t* __restrict__ ib,
                                                        const float* __restrict__ scale, bf16_t* __restrict__ out, int64_t E,
                                                        int F2) {
    const int epb = (int)blockDim.x / F2;
    const int le = threadIdx.x / F2, d = threadIdx.x - le * F2;
    if (le >= epb) return;
    const unsigned* __restrict__ a2 = reinterpret_cast<const unsigned*>(a);
    const unsigned* __restrict__ b2 = reinterpret_cast<const unsigned*>(b);
    for (int64_t e = (int64_t)blockIdx.x * epb + le; e < E; e += (int64_t)gridDim.x * epb) {
        const unsigned av = a2[(int64_t)ia[e] * F2 + d], bv = b2[(int64_t)ib[e] * F2 + d];
        const float sc = scale ? scale[e] : 1.0f;
        reinterpret_cast<unsigned*>(out)[e * F2 + d] = pk_bf16(__uint_as_float(av << 16) * __uint_as_float(bv << 16) * sc,
                                                               __uint_as_float(av & 0xffff0000u) * __uint_as_float(bv & 0xffff0000u) * sc);
    }
}

// dx = g * sigmoid(pre) for y = softplus(pre) - ln2 given y:  sigmoid(pre) = 1 - exp(-(y + ln2))   (bf16 rows, one pass)
__global__ __launch_bounds__(256) void ssp_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ y,
                                                      bf16_t* __restrict__ dx, int64_t n2) {
    const unsigned* __restrict__ g2 = reinterpret_cast<const unsigned*>(g);
    const unsigned* __restrict__ y2 = reinterpret_cast<const unsigned*>(y);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned gv = g2[i], yv = y2[i];
        const float s0 = 1.0f - __expf(-(__uint_as_float(yv << 16) + 0.6931471805599453f));
        const float s1 = 1.0f - __expf(-(__uint_as_float(yv & 0xffff0000u) + 0.6931471805599453f));
        reinterpret_cast<unsigned*>(dx)[i] = pk_bf16(__uint_as_float(gv << 16) * s0, __uint_as_float(gv & 0xffff0000u) * s1);
    }
}

static unsigned g_grid(int64_t total) {
    int64_t b = cdiv(total, 256);
    if (b > 256 * 16) b = 256 * 16;
    return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace mdl

extern "C" int mdl_gather_rows(const void* src, const int32_t* idx, void* out, int64_t E, int64_t C, int dtype,
                               mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(E >= 0 && C > 0 && (E == 0 || (src && idx && out)), MDL_E_ARG, "mdl_gather_rows: bad arguments");
    if (E == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_F32)
        hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(g_grid(E * C)), dim3(256), 0, st, (const float*)src, idx, (float*)out, E, (int)C);
    else if (dtype == MDL_BF16)
        hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(g_grid(E * C)), dim3(256), 0, st, (const bf16_t*)src, idx, (bf16_t*)out, E, (int)C);
    else { set_error("mdl_gather_rows: unsupported dtype %d", dtype); return MDL_E_UNSUPP; }
    return check_launch("mdl_gather_rows");
}

extern "C" int mdl_gather_mul_reduce(const void* h, const void* w, const float* scale, const int32_t* rowptr,
                                     const int32_t* col, const int32_t* eid, void* out, int64_t N, int64_t F,
                                     int reduce, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && F > 0 && (N == 0 || (h && rowptr && out)), MDL_E_ARG, "mdl_gather_mul_reduce: bad arguments");
    MDL_REQUIRE(reduce == MDL_SUM || reduce == MDL_MEAN, MDL_E_UNSUPP, "mdl_gather_mul_reduce: reduce must be sum or mean");
    if (N == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_BF16 && F % 2 == 0 && F <= 512 && reinterpret_cast<uintptr_t>(h) % 4 == 0 &&
        reinterpret_cast<uintptr_t>(w) % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 4 == 0) {
        const int F2 = (int)(F / 2), npb = 256 / F2;
        const dim3 g2((unsigned)cdiv(N, npb)), b2((unsigned)(npb * F2));
        if (reduce == MDL_MEAN) hipLaunchKernelGGL((gmr2_kernel<true>), g2, b2, 0, st, (const bf16_t*)h, (const bf16_t*)w, scale, rowptr, col, eid, (bf16_t*)out, N, F2);
        else hipLaunchKernelGGL((gmr2_kernel<false>), g2, b2, 0, st, (const bf16_t*)h, (const bf16_t*)w, scale, rowptr, col, eid, (bf16_t*)out, N, F2);
        return check_launch("mdl_gather_mul_reduce");
    }
    dim3 g(g_grid(N * F)), b(256);
#define MDL_GMR(T_, M_) hipLaunchKernelGGL((gmr_kernel<T_, M_>), g, b, 0, st, (const T_*)h, (const T_*)w, scale, rowptr, col, eid, (T_*)out, N, (int)F)
    if (dtype == MDL_F32) { if (reduce == MDL_MEAN) MDL_GMR(float, true); else MDL_GMR(float, false); }
    else if (dtype == MDL_BF16) { if (reduce == MDL_MEAN) MDL_GMR(bf16_t, true); else MDL_GMR(bf16_t, false); }
    else { set_error("mdl_gather_mul_reduce: unsupported dtype %d", dtype); return MDL_E_UNSUPP; }
#undef MDL_GMR
    return check_launch("mdl_gather_mul_reduce");
}

extern "C" int mdl_gather_mul_reduce_dw(const void* g, const void* w, const float* scale, const int32_t* rowptr_s,
                                        const int32_t* col_s, const int32_t* eid_s, void* dh, const void* h, void* dw,
                                        int64_t N, int64_t F, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(N >= 0 && F > 0 && (N == 0 || (g && w && rowptr_s && col_s && eid_s && dh && h && dw)), MDL_E_ARG,
                "mdl_gather_mul_reduce_dw: bad arguments");
    MDL_REQUIRE(dtype == MDL_BF16 && F % 2 == 0 && F <= 512 &&
                    (reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dh) |
                     reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(dw)) % 4 == 0,
                MDL_E_UNSUPP, "mdl_gather_mul_reduce_dw: bf16 rows of an even width <= 512, 4-byte aligned");
    if (N == 0) return MDL_OK;
    const int F2 = (int)(F / 2), npb = 256 / F2;
    hipLaunchKernelGGL((gmr2_kernel<false, true>), dim3((unsigned)cdiv(N, npb)), dim3((unsigned)(npb * F2)), 0, (hipStream_t)stream,
                       (const bf16_t*)g, (const bf16_t*)w, scale, rowptr_s, col_s, eid_s, (bf16_t*)dh, N, F2, (const bf16_t*)h,
                       (bf16_t*)dw);
    return check_launch("mdl_gather_mul_reduce_dw");
}

extern "C" int mdl_edge_mul(const void* a, const int32_t* ia, const void* b, const int32_t* ib, const float* scale,
                            void* out, int64_t E, int64_t F, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(E >= 0 && F > 0 && (E == 0 || (a && ia && b && ib && out)), MDL_E_ARG, "mdl_edge_mul: bad arguments");
    if (E == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MDL_BF16 && F % 2 == 0 && F <= 512 && reinterpret_cast<uintptr_t>(a) % 4 == 0 &&
        reinterpret_cast<uintptr_t>(b) % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 4 == 0) {
        const int F2 = (int)(F / 2), epb = 256 / F2;
        int64_t grid = cdiv(E, epb);
        if (grid > 256 * 32) grid = 256 * 32;
        hipLaunchKernelGGL(edge_mul2_kernel, dim3((unsigned)grid), dim3((unsigned)(epb * F2)), 0, st, (const bf16_t*)a, ia, (const bf16_t*)b, ib, scale, (bf16_t*)out, E, F2);
        return check_launch("mdl_edge_mul");
    }
    if (dtype == MDL_F32)
        hipLaunchKernelGGL((edge_mul_kernel<float>), dim3(g_grid(E * F)), dim3(256), 0, st, (const float*)a, ia, (const float*)b, ib, scale, (float*)out, E, (int)F);
    else if (dtype == MDL_BF16)
        hipLaunchKernelGGL((edge_mul_kernel<bf16_t>), dim3(g_grid(E * F)), dim3(256), 0, st, (const bf16_t*)a, ia, (const bf16_t*)b, ib, scale, (bf16_t*)out, E, (int)F);
    else { set_error("mdl_edge_mul: unsupported dtype %d", dtype); return MDL_E_UNSUPP; }
    return check_launch("mdl_edge_mul");
}

extern "C" int mdl_ssp_bwd(const void* g, const void* y, void* dx, int64_t n, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16 && n % 2 == 0, MDL_E_UNSUPP, "mdl_ssp_bwd: bf16 with an even element count only");
    MDL_REQUIRE(n == 0 || (g && y && dx && (reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dx)) % 4 == 0),
                MDL_E_ARG, "mdl_ssp_bwd: null or misaligned pointer");
    if (n == 0) return MDL_OK;
    int64_t grid = cdiv(n / 2, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    hipLaunchKernelGGL(ssp_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, (const bf16_t*)y, (bf16_t*)dx, n / 2);
    return check_launch("mdl_ssp_bwd");
}
