// cgconv_node.hip — K3c: node-level dense half of the CGConv backward (bf16 mode).
//
// After the edge pass (cgconv.hip) has reduced dpre by target (r_tgt) and by source (r_src),
//     dx  = grad_out + [r_tgt | r_src] (N x 4Cp)  @  Wn (4Cp x C)
//     dWn = [r_tgt | r_src]^T (4Cp x N)           @  x  (N x C)
// with Wn rows ordered (f_tgt, s_tgt, f_src, s_src) like the columns of [r_tgt | r_src].
// The reference gets these from autograd through eager cat/addmm ops
// (/root/reference/matdeeplearn/models/cgcnn.py:136-145 via PyG CGConv); a library GEMM handles the
// (4Cp x N)(N x C) product badly (K = N ~ 2e5, 256 x 64 output), so both products are done here in one
// pass over r_tgt/r_src: HBM-bound, algorithmic bytes N*(2*2Cp*4 + 3*C*2).
//
// Workgroup = 4 waves = 128 consecutive nodes.  Wave w: dx rows of its own 32 nodes (MFMA, K = 4Cp),
// and the dWn row block [w*Cp, (w+1)*Cp) over all 128 nodes (MFMA, K = nodes).  dWn partials stay
// in registers across the grid-stride loop and are flushed once per wave with fp32 atomics.
#include "mdl_common.h"

namespace mdl {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4_t r = {pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    return __builtin_bit_cast(bf16x8, r);
}

template <int CP>   // CP = padded channels = C (32 or 64)
__global__ __launch_bounds__(256, 2) void cgconv_node_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ gout,
                                                             const float* __restrict__ r_tgt,
                                                             const float* __restrict__ r_src,
                                                             const bf16_t* __restrict__ wn_t, bf16_t* __restrict__ dx,
                                                             float* __restrict__ dwn, int64_t N) {
    constexpr int K4 = 4 * CP;          // columns of [r_tgt | r_src]
    constexpr int LD = K4 + 8;          // LDS row stride of Wn^T (odd number of 16-byte slots)
    constexpr int NT = CP / 32;         // 32-wide feature tiles
    constexpr int MT = CP / 32;         // 32-row tiles of this wave's dWn row block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);
    for (int q = threadIdx.x; q < CP * (K4 / 8); q += blockDim.x) {
        const int row = q / (K4 / 8), c8 = q - row * (K4 / 8);
        *reinterpret_cast<bf16x8*>(wl + row * LD + c8 * 8) = *reinterpret_cast<const bf16x8*>(wn_t + row * K4 + c8 * 8);
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    f32x16 dw[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[a][b][r] = 0.0f;

    const int64_t n_super = (N + 127) / 128;
    for (int64_t sc = blockIdx.x; sc < n_super; sc += gridDim.x) {
        const int64_t nb = sc * 128;
        // ---- dx for this wave's 32 nodes -------------------------------------------------------
        {
            const int64_t node = nb + wv * 32 + i;
            const bool ok = node < N;
            const float* rt = r_tgt + node * (2 * CP);
            const float* rs = r_src + node * (2 * CP);
            f32x16 acc[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < K4 / 16; ++kk) {
                const int c0 = 16 * kk + 8 * h;
                const float* src = (c0 < 2 * CP) ? rt + c0 : rs + (c0 - 2 * CP);
                float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
                    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
                    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
                }
                const bf16x8 a = pack8(v);
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const bf16x8 bb = *reinterpret_cast<const bf16x8*>(wl + (b * 32 + i) * LD + 16 * kk + 8 * h);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t n = nb + wv * 32 + d_row(r, h);
                    if (n < N) {
                        const int64_t o = n * CP + b * 32 + i;
                        dx[o] = f2bf(bf2f(gout[o]) + acc[b][r]);
                    }
                }
        }
        // ---- dWn rows [wv*CP, (wv+1)*CP) over the 128 nodes ---------------------------------------
        {
            const float* rbase = (wv < 2) ? r_tgt : r_src;
            const int coff = (wv & 1) * CP;                 // f-half / s-half inside the 2CP row
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                bf16x8 bfr[NT];
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    bf16x8 t;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int64_t n = nb + 16 * ks + 8 * h + q;
                        t[q] = (n < N) ? (short)x[n * CP + b * 32 + i] : (short)0;
                    }
                    bfr[b] = t;
                }
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int64_t n = nb + 16 * ks + 8 * h + q;
                        v[q] = (n < N) ? rbase[n * (2 * CP) + coff + a * 32 + i] : 0.0f;
                    }
                    const bf16x8 af = pack8(v);
#pragma unroll
                    for (int b = 0; b < NT; ++b) dw[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr[b], dw[a][b], 0, 0, 0);
                }
            }
        }
    }
    // flush dWn partials: D rows = channel slot, cols = feature
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wv * CP + a * 32 + d_row(r, h);
                unsafeAtomicAdd(dwn + (int64_t)row * CP + b * 32 + i, dw[a][b][r]);
            }
}

}  // namespace mdl

extern "C" int mdl_cgconv_bwd_node(const void* x, const void* grad_out, const float* r_tgt, const float* r_src,
                                   const void* wn_t, void* dx, float* dwn, int64_t N, int C, int dtype,
                                   mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: bf16 only (fp32 parity mode uses library GEMMs)");
    MDL_REQUIRE(C == 32 || C == 64, MDL_E_UNSUPP, "mdl_cgconv_bwd_node: C must be 32 or 64 (got %d)", C);
    MDL_REQUIRE(N >= 0 && (N == 0 || (x && grad_out && r_tgt && r_src && wn_t && dx && dwn)), MDL_E_ARG,
                "mdl_cgconv_bwd_node: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(wn_t) % 16 == 0 && reinterpret_cast<uintptr_t>(r_tgt) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(r_src) % 16 == 0, MDL_E_ARG, "mdl_cgconv_bwd_node: 16-byte alignment required");
    if (N == 0) return MDL_OK;
    int64_t grid = cdiv(N, 128);
    if (grid > 512) grid = 512;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) {
        const int lds = 64 * (256 + 8) * 2;
        hipLaunchKernelGGL((cgconv_node_kernel<64>), dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)x,
                           (const bf16_t*)grad_out, r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
    } else {
        const int lds = 32 * (128 + 8) * 2;
        hipLaunchKernelGGL((cgconv_node_kernel<32>), dim3((unsigned)grid), dim3(256), lds, st, (const bf16_t*)x,
                           (const bf16_t*)grad_out, r_tgt, r_src, (const bf16_t*)wn_t, (bf16_t*)dx, dwn, N);
    }
    return check_launch("mdl_cgconv_bwd_node");
}
