// rbf.hip — K1: Gaussian RBF edge expansion (write-bandwidth bound).
// Restates GaussianSmearing.forward, /root/reference/matdeeplearn/process/process.py:588-590:
//   dist = d[:,None] - offset[None,:];  out = exp(coeff * dist^2)       (all fp32)
// Algorithmic bytes per edge: 4 (d) + G*sizeof(out)  ->  204 B fp32 / 104 B bf16 at G = 50.
#include "mdl_common.h"

namespace mdl {

// One thread produces VEC consecutive elements of the flattened [E*G] output (dense case
// ld_out == G) so every store is a full 16-byte vector (fp32 x4 / bf16 x8) and a wave writes 1 KiB contiguous.  exp() is the precise ocml expf: the kernel is bound by
// the store stream (50 exps per 104..204 bytes), not by VALU.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void rbf_dense_kernel(const float* __restrict__ d,
                                                        const float* __restrict__ offsets, float coeff,
                                                        T* __restrict__ out, int64_t total, int G) {
    __shared__ float s_off[256];
    for (int i = threadIdx.x; i < G; i += blockDim.x) s_off[i] = offsets[i];
    __syncthreads();
    // bf16 output keeps 8 mantissa bits: the hardware base-2 exponential (v_exp_f32, ~1 ulp fp32) is
    // exact enough; the fp32 (parity) output uses the precise expf.
    constexpr bool FAST = sizeof(T) == 2;
    const float c2 = coeff * LOG2E_F;
    const unsigned uG = (unsigned)G;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * VEC;
    for (int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC; base < total; base += stride) {
        int64_t e;
        int k;
        if (total < (1ll << 32)) {                      // 32-bit division is several times cheaper than 64-bit
            const unsigned ue = (unsigned)base / uG;
            e = ue;
            k = (int)((unsigned)base - ue * uG);
        } else {
            e = base / G;
            k = (int)(base - e * G);
        }
        float de = d[e];
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float diff = de - s_off[k];
            v[j] = FAST ? __builtin_amdgcn_exp2f(c2 * (diff * diff)) : expf(coeff * (diff * diff));
            if (++k == G) {
                k = 0;
                ++e;
                if (base + j + 1 < total) de = d[e];
            }
        }
        if (base + VEC <= total) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(out + base) = f32x4{v[0], v[1], v[2], v[3]};
            } else if constexpr (VEC == 8) {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                *reinterpret_cast<u32x4*>(out + base) =
                    u32x4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
            } else {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                *reinterpret_cast<u32x2*>(out + base) = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
            }
        } else {
            for (int j = 0; base + j < total; ++j) Elem<T>::st(out + base + j, v[j]);
        }
    }
}

// G known at compile time (the reference's 50 features, process.py:500-502), dense rows: a block owns EPB consecutive
// edges.  Their distances are read once, coalesced, into LDS; a thread produces 16-byte chunks c = tid, tid + 256, ... of
// the block's EPB*G contiguous outputs, and (edge, feature) of a chunk come from a division by a CONSTANT (multiply +
// shift).  The generic kernel above pays a 32-bit division, a dependent distance load and an LDS offset read per element
// per trip for one 16-byte store per thread and launches 2.6e5 waves of ~100 instructions (78 us on 2.65e6 edges, 3.5 TB/s).
template <typename T, int G_, int EPB>
__global__ __launch_bounds__(256) void rbf_rows_kernel(const float* __restrict__ d, const float* __restrict__ offsets, float coeff,
                                                       T* __restrict__ out, int64_t E) {
    constexpr int VEC = 16 / (int)sizeof(T);
    static_assert((EPB * G_) % VEC == 0, "a block's outputs are whole 16-byte chunks");
    constexpr int CH = EPB * G_ / VEC;                    // chunks per block
    __shared__ float s_off[G_];
    __shared__ float s_d[EPB];
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int ne = (int)min((int64_t)EPB, E - e0);
    for (int i = threadIdx.x; i < G_; i += 256) s_off[i] = offsets[i];
    for (int i = threadIdx.x; i < EPB; i += 256) s_d[i] = d[e0 + min(i, ne - 1)];
    __syncthreads();
    constexpr bool FAST = sizeof(T) == 2;
    const float c2 = coeff * LOG2E_F;
    T* __restrict__ ob = out + e0 * G_;
    const int n_el = ne * G_;                              // elements that exist in this block
    for (int c = threadIdx.x; c < CH; c += 256) {
        const int base = c * VEC;
        if (base >= n_el) break;
        int e = base / G_, k = base - e * G_;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float diff = s_d[min(e, EPB - 1)] - s_off[k];
            v[j] = FAST ? __builtin_amdgcn_exp2f(c2 * (diff * diff)) : expf(coeff * (diff * diff));
            if (++k == G_) { k = 0; ++e; }
        }
        if (base + VEC <= n_el) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4*>(ob + base) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                *reinterpret_cast<u32x4*>(ob + base) =
                    u32x4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
            }
        } else {
            for (int j = 0; base + j < n_el; ++j) Elem<T>::st(ob + base + j, v[j]);
        }
    }
}

// Strided rows (ld_out > G): one thread per element.
template <typename T>
__global__ __launch_bounds__(256) void rbf_strided_kernel(const float* __restrict__ d,
                                                          const float* __restrict__ offsets, float coeff,
                                                          T* __restrict__ out, int64_t E, int G, int64_t ld) {
    const int64_t total = E * G;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        int64_t e = i / G;
        int k = (int)(i - e * G);
        const float diff = d[e] - offsets[k];
        const float q = diff * diff;
        Elem<T>::st(out + e * ld + k, sizeof(T) == 2 ? __builtin_amdgcn_exp2f((coeff * LOG2E_F) * q) : expf(coeff * q));
    }
}

template <typename T>
static int launch_rbf(const float* d, const float* offsets, float coeff, T* out, int64_t E, int G, int64_t ld,
                      hipStream_t st) {
    if (E == 0) return MDL_OK;
    const int64_t total = E * G;
    if (ld == G && G == 50 && (reinterpret_cast<uintptr_t>(out) % 16) == 0) {
        constexpr int EPB = 128;                       // 128 edges x 50 features: 6400 outputs = 800 / 1600 chunks per block
        hipLaunchKernelGGL((rbf_rows_kernel<T, 50, EPB>), dim3((unsigned)cdiv(E, EPB)), dim3(256), 0, st, d, offsets, coeff, out, E);
    } else if (ld == G && (reinterpret_cast<uintptr_t>(out) % 16) == 0) {
        constexpr int VEC = 16 / (int)sizeof(T);       // 16-byte stores: 4 floats / 8 bf16 per thread
        int64_t blocks = cdiv(cdiv(total, VEC), 256);
        if (blocks > 256 * 256) blocks = 256 * 256;    // ~one trip per thread: a trip starts with a dependent load of d[e]
        hipLaunchKernelGGL((rbf_dense_kernel<T, VEC>), dim3((unsigned)blocks), dim3(256), 0, st, d, offsets, coeff,
                           out, total, G);
    } else {
        int64_t blocks = cdiv(total, 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((rbf_strided_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, d, offsets, coeff,
                           out, E, G, ld);
    }
    return check_launch("mdl_rbf_expand");
}

}  // namespace mdl

extern "C" int mdl_rbf_expand(const float* d, const float* offsets, float coeff, void* out, int64_t E, int G,
                              int64_t ld_out, int out_dtype, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(E >= 0 && G > 0 && G <= 256, MDL_E_ARG, "mdl_rbf_expand: bad E=%lld G=%d", (long long)E, G);
    MDL_REQUIRE(ld_out >= G, MDL_E_ARG, "mdl_rbf_expand: ld_out %lld < G %d", (long long)ld_out, G);
    MDL_REQUIRE(E == 0 || (d && offsets && out), MDL_E_ARG, "mdl_rbf_expand: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == MDL_F32) return launch_rbf<float>(d, offsets, coeff, (float*)out, E, G, ld_out, st);
    if (out_dtype == MDL_BF16) return launch_rbf<bf16_t>(d, offsets, coeff, (bf16_t*)out, E, G, ld_out, st);
    set_error("mdl_rbf_expand: unsupported dtype %d", out_dtype);
    return MDL_E_UNSUPP;
}
