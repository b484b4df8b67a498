// gru.hip — the gate arithmetic of ONE step of a single-layer GRU as one launch per direction.
// The reference's MPNN runs `out, h = self.gru_list[i](m.unsqueeze(0), h)` after every NNConv layer
// (/root/reference/matdeeplearn/models/mpnn.py:160-161): a sequence of length 1, i.e. two dense products
//     gi = W_ih m + b_ih,  gh = W_hh h + b_hh            ([N, 3C] each, torch's gate order r | z | n)
// (dense layers + TN-GEMM weight gradients of this library) and the gates
//     r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),  h' = n + z * (h - n).
// Written out with tensor operations the gates are ~12 elementwise launches forward and ~25 backward (chunk / cat / cast
// included) on [N, C] tensors of 6e4 x 100 — a tenth of the MPNN step.  Here: forward = h' (fp32) and its copy in the compute dtype
// (the next layer's input), backward = d gi, d gh (compute dtype, dense [N, 3C]: the operands of the two dense layers' backward) and
// d h (fp32) from the gradients of both outputs, with r, z, n recomputed from gi, gh (nothing saved per row but the inputs).
// Gate arithmetic in fp32 on the stored (rounded) gi / gh, like the written-out form.  HBM-bound: fwd N (6C s + 8C + C s) bytes.
#include "mdl_common.h"

namespace mdl {

__device__ __forceinline__ float gru_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }

template <typename T, int W> struct GruVec;
template <typename T> struct GruVec<T, 4> {
    __device__ static __forceinline__ void ld(const T* p, float* v) { VecW<T, 4>::ld(p, v); }
    __device__ static __forceinline__ void st(T* p, const float* v) { VecW<T, 4>::st(p, v); }
};
template <typename T> struct GruVec<T, 1> {
    __device__ static __forceinline__ void ld(const T* p, float* v) { v[0] = Elem<T>::ld(p); }
    __device__ static __forceinline__ void st(T* p, const float* v) { Elem<T>::st(p, v[0]); }
};

template <typename T, int W>
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const T* __restrict__ gi, const T* __restrict__ gh, const float* __restrict__ h,
                                                            float* __restrict__ h_out, T* __restrict__ out_lp, int64_t N, int C) {
    const int CG = C / W;
    const int64_t total = N * CG, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t n = q / CG;
        const int c = (int)(q - n * CG) * W;
        const int64_t g0 = n * 3 * C + c, h0 = n * C + c;
        float ir[W], iz[W], in_[W], hr[W], hz[W], hn[W], hv[W], o[W];
        GruVec<T, W>::ld(gi + g0, ir); GruVec<T, W>::ld(gi + g0 + C, iz); GruVec<T, W>::ld(gi + g0 + 2 * C, in_);
        GruVec<T, W>::ld(gh + g0, hr); GruVec<T, W>::ld(gh + g0 + C, hz); GruVec<T, W>::ld(gh + g0 + 2 * C, hn);
        GruVec<float, W>::ld(h + h0, hv);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float r = gru_sigmoid(ir[j] + hr[j]), z = gru_sigmoid(iz[j] + hz[j]);
            const float nn = tanhf(in_[j] + r * hn[j]);
            o[j] = nn + z * (hv[j] - nn);
        }
        GruVec<float, W>::st(h_out + h0, o);
        if (out_lp) GruVec<T, W>::st(out_lp + h0, o);
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const T* __restrict__ gi, const T* __restrict__ gh, const float* __restrict__ h,
                                                            const float* __restrict__ g_h, const T* __restrict__ g_lp,
                                                            T* __restrict__ dgi, T* __restrict__ dgh, float* __restrict__ dh,
                                                            int64_t N, int C) {
    const int CG = C / W;
    const int64_t total = N * CG, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += stride) {
        const int64_t n = q / CG;
        const int c = (int)(q - n * CG) * W;
        const int64_t g0 = n * 3 * C + c, h0 = n * C + c;
        float ir[W], iz[W], in_[W], hr[W], hz[W], hn[W], hv[W], g[W], g2[W];
        GruVec<T, W>::ld(gi + g0, ir); GruVec<T, W>::ld(gi + g0 + C, iz); GruVec<T, W>::ld(gi + g0 + 2 * C, in_);
        GruVec<T, W>::ld(gh + g0, hr); GruVec<T, W>::ld(gh + g0 + C, hz); GruVec<T, W>::ld(gh + g0 + 2 * C, hn);
        GruVec<float, W>::ld(h + h0, hv);
#pragma unroll
        for (int j = 0; j < W; ++j) { g[j] = 0.0f; g2[j] = 0.0f; }
        if (g_h) GruVec<float, W>::ld(g_h + h0, g);
        if (g_lp) GruVec<T, W>::ld(g_lp + h0, g2);
        float dr_[W], dz_[W], dn_[W], dnr[W], dhv[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float gg = g[j] + g2[j];
            const float r = gru_sigmoid(ir[j] + hr[j]), z = gru_sigmoid(iz[j] + hz[j]);
            const float nn = tanhf(in_[j] + r * hn[j]);
            const float dan = gg * (1.0f - z) * (1.0f - nn * nn);     // d (i_n + r h_n)
            const float daz = gg * (hv[j] - nn) * z * (1.0f - z);     // d (i_z + h_z)
            const float dar = dan * hn[j] * r * (1.0f - r);           // d (i_r + h_r)
            dr_[j] = dar; dz_[j] = daz; dn_[j] = dan; dnr[j] = dan * r; dhv[j] = gg * z;
        }
        GruVec<T, W>::st(dgi + g0, dr_); GruVec<T, W>::st(dgi + g0 + C, dz_); GruVec<T, W>::st(dgi + g0 + 2 * C, dn_);
        GruVec<T, W>::st(dgh + g0, dr_); GruVec<T, W>::st(dgh + g0 + C, dz_); GruVec<T, W>::st(dgh + g0 + 2 * C, dnr);
        GruVec<float, W>::st(dh + h0, dhv);
    }
}

static unsigned gru_grid(int64_t total) {
    int64_t b = cdiv(total, 256);
    if (b > 256 * 8) b = 256 * 8;
    return (unsigned)(b < 1 ? 1 : b);
}

static int gru_check(const char* name, int64_t N, int C, int dtype) {
    MDL_REQUIRE(dtype == MDL_F32 || dtype == MDL_BF16, MDL_E_UNSUPP, "%s: unsupported dtype %d", name, dtype);
    MDL_REQUIRE(N >= 0 && C >= 1 && C <= (1 << 16), MDL_E_ARG, "%s: bad N = %lld, C = %d", name, (long long)N, C);
    return MDL_OK;
}

template <typename T>
static bool gru_vec_ok(int C, std::initializer_list<const void*> ptrs) {
    if (C % 4) return false;
    for (const void* p : ptrs)
        if (p && reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T) < 16 ? 4 * sizeof(T) : 16)) return false;
    return true;
}

}  // namespace mdl

extern "C" int mdl_gru_gates_fwd(const void* gi, const void* gh, const float* h, float* h_out, void* out_lp, int64_t N, int C,
                                 int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = gru_check("mdl_gru_gates_fwd", N, C, dtype);
    if (rc) return rc;
    if (N == 0) return MDL_OK;
    MDL_REQUIRE(gi && gh && h && h_out, MDL_E_ARG, "mdl_gru_gates_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool f32ok = reinterpret_cast<uintptr_t>(h) % 16 == 0 && reinterpret_cast<uintptr_t>(h_out) % 16 == 0;
    if (dtype == MDL_BF16) {
        typedef bf16_t T;
        if (f32ok && gru_vec_ok<T>(C, {gi, gh, out_lp}))
            hipLaunchKernelGGL((gru_gates_fwd_kernel<T, 4>), dim3(gru_grid(N * (C / 4))), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, h_out, (T*)out_lp, N, C);
        else
            hipLaunchKernelGGL((gru_gates_fwd_kernel<T, 1>), dim3(gru_grid(N * C)), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, h_out, (T*)out_lp, N, C);
    } else {
        typedef float T;
        if (f32ok && gru_vec_ok<T>(C, {gi, gh, out_lp}))
            hipLaunchKernelGGL((gru_gates_fwd_kernel<T, 4>), dim3(gru_grid(N * (C / 4))), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, h_out, (T*)out_lp, N, C);
        else
            hipLaunchKernelGGL((gru_gates_fwd_kernel<T, 1>), dim3(gru_grid(N * C)), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, h_out, (T*)out_lp, N, C);
    }
    return check_launch("mdl_gru_gates_fwd");
}

extern "C" int mdl_gru_gates_bwd(const void* gi, const void* gh, const float* h, const float* g_h, const void* g_lp, void* dgi,
                                 void* dgh, float* dh, int64_t N, int C, int dtype, mdlStream_t stream) {
    using namespace mdl;
    int rc = gru_check("mdl_gru_gates_bwd", N, C, dtype);
    if (rc) return rc;
    if (N == 0) return MDL_OK;
    MDL_REQUIRE(gi && gh && h && dgi && dgh && dh && (g_h || g_lp), MDL_E_ARG, "mdl_gru_gates_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool f32ok = reinterpret_cast<uintptr_t>(h) % 16 == 0 && reinterpret_cast<uintptr_t>(dh) % 16 == 0 &&
                       reinterpret_cast<uintptr_t>(g_h) % 16 == 0;
    if (dtype == MDL_BF16) {
        typedef bf16_t T;
        if (f32ok && gru_vec_ok<T>(C, {gi, gh, g_lp, dgi, dgh}))
            hipLaunchKernelGGL((gru_gates_bwd_kernel<T, 4>), dim3(gru_grid(N * (C / 4))), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, g_h, (const T*)g_lp, (T*)dgi, (T*)dgh, dh, N, C);
        else
            hipLaunchKernelGGL((gru_gates_bwd_kernel<T, 1>), dim3(gru_grid(N * C)), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, g_h, (const T*)g_lp, (T*)dgi, (T*)dgh, dh, N, C);
    } else {
        typedef float T;
        if (f32ok && gru_vec_ok<T>(C, {gi, gh, g_lp, dgi, dgh}))
            hipLaunchKernelGGL((gru_gates_bwd_kernel<T, 4>), dim3(gru_grid(N * (C / 4))), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, g_h, (const T*)g_lp, (T*)dgi, (T*)dgh, dh, N, C);
        else
            hipLaunchKernelGGL((gru_gates_bwd_kernel<T, 1>), dim3(gru_grid(N * C)), dim3(256), 0, st, (const T*)gi, (const T*)gh, h, g_h, (const T*)g_lp, (T*)dgi, (T*)dgh, dh, N, C);
    }
    return check_launch("mdl_gru_gates_bwd");
}
