// bn.hip — training-mode BatchNorm1d over the node dimension, x: [N, C] row-major (channels last).
// The reference applies torch.nn.BatchNorm1d after every conv layer
// (/root/reference/matdeeplearn/models/cgcnn.py:85-87,143; also on edges inside the MEGNet MLPs,
// megnet.py:47-48).  The library kernels for this shape (N ~ 2e5, C = 64) take ~220 us per layer for four
// passes over a 27 MB tensor; these kernels are plain HBM streams (16-byte loads, one partial sum per
// block, one fp32 atomic per channel per block): N*C*s bytes per reduction pass, 2*N*C*s per apply pass.
//
//   mdl_bn_stats : sums[0,c] += sum_n (x[n,c] - shift[c]),  sums[1,c] += sum_n (x[n,c] - shift[c])^2
//                  shift = x[0,:]  (shifted sums avoid the E[x^2]-E[x]^2 cancellation)
//   mdl_bn_apply : mean/var from the sums; y = (x - mean) * invstd * gamma + beta; writes save_mean/save_invstd
//                  and updates running_mean / running_var (unbiased) with `momentum`
//   mdl_bn_bwd_stats : sums[0,c] += sum dy,  sums[1,c] += sum dy * xhat
//   mdl_bn_bwd_apply : dx = gamma*invstd*(dy - mean(dy) - xhat*mean(dy*xhat));  dgamma = sums[1], dbeta = sums[0]
#include "mdl_common.h"

namespace mdl {

constexpr int BN_R = MDL_BN_REPLICAS;    // copies of the [2][C] sums the reduction scatters its atomics over

// totals of the BN_R copies into LDS (tot[2C]); block 0 also publishes them in row BN_R of `sums` for the host side
__device__ __forceinline__ void bn_totals(const float* __restrict__ sums, float* tot, int C, float* publish) {
    for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) {
        float v[BN_R];
#pragma unroll
        for (int r = 0; r < BN_R; ++r) v[r] = sums[(size_t)r * 2 * C + c];
        float t = 0.0f;
#pragma unroll
        for (int r = 0; r < BN_R; ++r) t += v[r];
        tot[c] = t;
        if (publish && blockIdx.x == 0) publish[c] = t;
    }
    __syncthreads();
}

// Block = 256 threads = (256 / CG) row lanes x CG channel groups of W channels (CG = C / W).
// MODE 0: forward stats of x.  MODE 1: backward stats (a = dy, b = x).
template <typename T, int MODE, int W>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        const float* __restrict__ save, float* __restrict__ sums,
                                                        int64_t Ncap, int C, const int64_t* __restrict__ n_dev) {
    typedef VecW<T, W> V;
    // rows that exist: all Ncap of them, or the device-side count of a padded static batch (HIP-graph replays: the
    // launch arguments are frozen at capture, the true row count changes with every batch)
    const int64_t N = n_dev ? max((int64_t)1, min(*n_dev, Ncap)) : Ncap;
    __shared__ float red[2][256 * 8 / 8 * 8];   // 2 x 256 x W floats max (W <= 8)
    const int CG = C / W;
    const int rows_per_block = (int)blockDim.x / CG;          // blockDim = rows_per_block * CG (<= 256)
    const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const bool active = rl < rows_per_block;
    float s0[W], s1[W], p0[W], p1[W];
#pragma unroll
    for (int j = 0; j < W; ++j) { s0[j] = 0.0f; s1[j] = 0.0f; }
    if (MODE == 0) {
        V::ld(a + cg * W, p0);                           // shift = first row
    } else {
#pragma unroll
        for (int j = 0; j < W; ++j) { p0[j] = save[cg * W + j]; p1[j] = save[C + cg * W + j]; }   // mean, invstd
    }
    if (active) {
        // four rows per trip, all loads issued before the first use (clamped row, masked contribution): a thread
        // makes only a handful of trips, so with one load in flight the kernel is bound by memory latency
        constexpr int U = 4;
        const int64_t stride = (int64_t)gridDim.x * rows_per_block;
        for (int64_t n0 = (int64_t)blockIdx.x * rows_per_block + rl; n0 < N; n0 += U * stride) {
            float va[U][W], vb[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t n = min(n0 + u * stride, N - 1);
                V::ld(a + n * C + cg * W, va[u]);
                if (MODE != 0) V::ld(b + n * C + cg * W, vb[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float m = (n0 + u * stride < N) ? 1.0f : 0.0f;
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < W; ++j) { const float d = (va[u][j] - p0[j]) * m; s0[j] += d; s1[j] += d * d; }
                } else {
#pragma unroll
                    for (int j = 0; j < W; ++j) { const float v = va[u][j] * m; s0[j] += v; s1[j] += v * ((vb[u][j] - p0[j]) * p1[j]); }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < W; ++j) { red[0][threadIdx.x * W + j] = s0[j]; red[1][threadIdx.x * W + j] = s1[j]; }
    __syncthreads();
    // threads 0 .. C-1 each own one channel: sum over the row lanes
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x, g = c / W, j = c % W;
        float t0 = 0.0f, t1 = 0.0f;
        for (int r = 0; r < rows_per_block; ++r) { t0 += red[0][(r * CG + g) * W + j]; t1 += red[1][(r * CG + g) * W + j]; }
        // 2C atomics per block on the same few cache lines serialise in L2 (10 of this kernel's 17 us with one copy):
        // BN_R copies of the sums, chosen by block, each in its own lines; the consumers add them up
        float* dst = sums + (size_t)(blockIdx.x % BN_R) * 2 * C;
        unsafeAtomicAdd(dst + c, t0);
        unsafeAtomicAdd(dst + C + c, t1);
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, float* __restrict__ sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ save, float* __restrict__ run_mean,
                                                       float* __restrict__ run_var, T* __restrict__ y, int64_t Ncap, int C,
                                                       float eps, float momentum, const int64_t* __restrict__ n_dev,
                                                       int shift_row) {
    typedef VecW<T, W> V;
    constexpr int U = 4;                                     // four independent 16-byte loads in flight per thread
    const int64_t N = n_dev ? max((int64_t)1, min(*n_dev, Ncap)) : Ncap;
    __shared__ float tot[512];
    const int CG = C / W;
    const int cg = threadIdx.x % CG;
    const int64_t total = N * CG;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // the first rows are requested BEFORE the statistics: the block's prologue (sums -> mean / invstd -> scale / shift) is
    // a chain of dependent loads, and a block streams only a few rows per thread — back to back they double its time
    float v[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) V::ld(x + (min(q0 + u * stride, total - 1) / CG) * C + cg * W, v[u]);
    float mean[W], scale[W], shiftv[W], sh[W];
    V::ld(x + cg * W, sh);
    if (shift_row) {                                          // MDL_BN_SHIFT_ROW: the producer of the sums (mdl_linear_act_stats,
        const float* srow = sums + (size_t)(2 * BN_R + 2) * C;   // mdl_cgconv_fwd_ex) left its shift behind the totals rows
#pragma unroll
        for (int j = 0; j < W; ++j) sh[j] = srow[cg * W + j];
    }
    bn_totals(sums, tot, C, sums + (size_t)BN_R * 2 * C);
    const float invn = 1.0f / (float)N;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int c = cg * W + j;
        const float m1 = tot[c] * invn;
        const float var = fmaxf(tot[C + c] * invn - m1 * m1, 0.0f);
        const float istd = rsqrtf(var + eps);
        mean[j] = sh[j] + m1;
        const float g = gamma ? gamma[c] : 1.0f;
        scale[j] = istd * g;
        shiftv[j] = (beta ? beta[c] : 0.0f) - mean[j] * scale[j];
        if (blockIdx.x == 0 && (int)threadIdx.x < CG) {
            save[c] = mean[j];
            save[C + c] = istd;
            if (run_mean) {
                const float unb = N > 1 ? var * ((float)N / (float)(N - 1)) : var;
                run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * mean[j];
                run_var[c] = (1.0f - momentum) * run_var[c] + momentum * unb;
            }
        }
    }
    // grid-stride over (row, channel group); blockDim is a multiple of CG so cg is loop invariant
    while (true) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * stride;
#pragma unroll
            for (int j = 0; j < W; ++j) v[u][j] = v[u][j] * scale[j] + shiftv[j];
            if (q < total) V::st(y + (q / CG) * C + cg * W, v[u]);
        }
        q0 += U * stride;
        if (q0 >= total) break;
#pragma unroll
        for (int u = 0; u < U; ++u) V::ld(x + (min(q0 + u * stride, total - 1) / CG) * C + cg * W, v[u]);
    }
    // padding rows of a static batch: exact zeros (they must not carry anything into the layers behind)
    if (N < Ncap) {
        float z[W];
#pragma unroll
        for (int j = 0; j < W; ++j) z[j] = 0.0f;
        for (int64_t q = total + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < Ncap * CG; q += stride)
            V::st(y + (q / CG) * C + cg * W, z);
    }
}

// MASK (mdl_bn_bwd_apply_relu_n): x is the OUTPUT of a ReLU (Linear -> ReLU -> BatchNorm, the layer order of the MEGNet blocks,
// megnet.py:47-48) and what is written is the gradient w.r.t. that ReLU's input: dx is zeroed where x <= 0, so the dense
// layer's backward behind it needs neither the activation staging nor a read of x's rows for the mask.
template <typename T, int W, bool MASK = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ save, float* __restrict__ sums,
                                                           const float* __restrict__ gamma, T* __restrict__ dx, int64_t Ncap,
                                                           int C, const int64_t* __restrict__ n_dev) {
    typedef VecW<T, W> V;
    constexpr int U = 2;
    const int64_t N = n_dev ? max((int64_t)1, min(*n_dev, Ncap)) : Ncap;
    __shared__ float tot[512];
    const int CG = C / W;
    const int cg = threadIdx.x % CG;
    const int64_t total = N * CG;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t q0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float vd[U][W], vx[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {                              // rows first, statistics second (see bn_apply_kernel)
        const int64_t n = min(q0 + u * stride, total - 1) / CG;
        V::ld(dy + n * C + cg * W, vd[u]);
        V::ld(x + n * C + cg * W, vx[u]);
    }
    bn_totals(sums, tot, C, sums + (size_t)BN_R * 2 * C);
    const float invn = 1.0f / (float)N;
    float mean[W], istd[W], k0[W], k1[W], gs[W];
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int c = cg * W + j;
        mean[j] = save[c];
        istd[j] = save[C + c];
        gs[j] = (gamma ? gamma[c] : 1.0f) * istd[j];
        k0[j] = tot[c] * invn;            // mean(dy)
        k1[j] = tot[C + c] * invn;        // mean(dy * xhat)
    }
    while (true) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * stride;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                const float d = gs[j] * (vd[u][j] - k0[j] - (vx[u][j] - mean[j]) * istd[j] * k1[j]);
                vd[u][j] = (!MASK || vx[u][j] > 0.0f) ? d : 0.0f;
            }
            if (q < total) V::st(dx + (q / CG) * C + cg * W, vd[u]);
        }
        q0 += U * stride;
        if (q0 >= total) break;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t n = min(q0 + u * stride, total - 1) / CG;
            V::ld(dy + n * C + cg * W, vd[u]);
            V::ld(x + n * C + cg * W, vx[u]);
        }
    }
    if (N < Ncap) {                                           // padding rows: no gradient
        float z[W];
#pragma unroll
        for (int j = 0; j < W; ++j) z[j] = 0.0f;
        for (int64_t q = total + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < Ncap * CG; q += stride)
            V::st(dx + (q / CG) * C + cg * W, z);
    }
}


// vector width: 16 bytes when the channel count allows it (bf16: C % 8), else 4 elements (bf16 C = 100, 150: 8 bytes)
static int bn_width(int C, int dtype) { return (dtype == MDL_BF16 && C % 8 == 0) ? 8 : 4; }
// threads per block: a whole number of rows, each CG = C / W lanes wide
static int bn_threads(int C, int W) { const int cg = C / W; return 256 / cg * cg; }

static int bn_check(const char* name, int64_t N, int C, int dtype, const void* p) {
    MDL_REQUIRE(dtype == MDL_F32 || dtype == MDL_BF16, MDL_E_UNSUPP, "%s: unsupported dtype %d", name, dtype);
    MDL_REQUIRE(N >= 1 && C >= 4 && C % 4 == 0 && C <= 256, MDL_E_UNSUPP,
                "%s: need N>=1 and C a multiple of 4, C<=256 (got N=%lld C=%d)", name, (long long)N, C);
    MDL_REQUIRE(p && reinterpret_cast<uintptr_t>(p) % (bn_width(C, dtype) * (dtype == MDL_BF16 ? 2 : 4)) == 0, MDL_E_ARG,
                "%s: null or misaligned tensor", name);
    return MDL_OK;
}

static unsigned bn_grid(int64_t N, int C, int W, bool det) {
    if (det) return 1;            // MDL_DETERMINISTIC: one workgroup, one add per sum
    const int rows = 256 / (C / W);
    int64_t g = cdiv(N, (int64_t)rows * 8);
    if (g > 1024) g = 1024;       // every block ends with 2C atomics, spread over BN_R copies of the sums
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace mdl

extern "C" int mdl_bn_sums_rows(void) { return 2 * (MDL_BN_REPLICAS + 1); }

extern "C" int mdl_bn_stats_n(const void* x, float* sums, int64_t N, int C, const int64_t* n_dev, int dtype, mdlStream_t stream) {
    using namespace mdl;
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;
    dtype &= MDL_DTYPE_MASK;
    int rc = bn_check("mdl_bn_stats", N, C, dtype, x);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = bn_width(C, dtype);
    if (dtype == MDL_BF16 && W == 8) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 0, 8>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 8)), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, sums, N, C, n_dev);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 0, 4>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 4)), 0, st, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, sums, N, C, n_dev);
    else hipLaunchKernelGGL((bn_reduce_kernel<float, 0, 4>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 4)), 0, st, (const float*)x, (const float*)nullptr, (const float*)nullptr, sums, N, C, n_dev);
    return check_launch("mdl_bn_stats");
}

extern "C" int mdl_bn_apply_n(const void* x, float* sums, const float* gamma, const float* beta, float* save,
                              float* running_mean, float* running_var, void* y, int64_t N, int C, float eps, float momentum,
                              const int64_t* n_dev, int dtype, mdlStream_t stream) {
    using namespace mdl;
    const int shift_row = (dtype & MDL_BN_SHIFT_ROW) ? 1 : 0;
    dtype &= MDL_DTYPE_MASK;
    int rc = bn_check("mdl_bn_apply", N, C, dtype, x);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    int64_t g = cdiv(N * (C / bn_width(C, dtype)), 256 * 4);
    if (g > 768) g = 768;          // 3 fat blocks per CU: the per-block statistics prologue amortises (measured 512..2048)
    if (g < 1) g = 1;
    const int W = bn_width(C, dtype);
    if (dtype == MDL_BF16 && W == 8) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, 8>), dim3((unsigned)g), dim3(bn_threads(C, 8)), 0, st, (const bf16_t*)x, sums, gamma, beta, save, running_mean, running_var, (bf16_t*)y, N, C, eps, momentum, n_dev, shift_row);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, 4>), dim3((unsigned)g), dim3(bn_threads(C, 4)), 0, st, (const bf16_t*)x, sums, gamma, beta, save, running_mean, running_var, (bf16_t*)y, N, C, eps, momentum, n_dev, shift_row);
    else hipLaunchKernelGGL((bn_apply_kernel<float, 4>), dim3((unsigned)g), dim3(bn_threads(C, 4)), 0, st, (const float*)x, sums, gamma, beta, save, running_mean, running_var, (float*)y, N, C, eps, momentum, n_dev, shift_row);
    return check_launch("mdl_bn_apply");
}

extern "C" int mdl_bn_bwd_stats_n(const void* dy, const void* x, const float* save, float* sums, int64_t N, int C,
                                  const int64_t* n_dev, int dtype, mdlStream_t stream) {
    using namespace mdl;
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;
    dtype &= MDL_DTYPE_MASK;
    int rc = bn_check("mdl_bn_bwd_stats", N, C, dtype, x);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int W = bn_width(C, dtype);
    if (dtype == MDL_BF16 && W == 8) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 1, 8>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 8)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, N, C, n_dev);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((bn_reduce_kernel<bf16_t, 1, 4>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 4)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, N, C, n_dev);
    else hipLaunchKernelGGL((bn_reduce_kernel<float, 1, 4>), dim3(bn_grid(N, C, W, det)), dim3(bn_threads(C, 4)), 0, st, (const float*)dy, (const float*)x, save, sums, N, C, n_dev);
    return check_launch("mdl_bn_bwd_stats");
}

static int bn_bwd_apply_launch(const char* name, bool mask, const void* dy, const void* x, const float* save, float* sums, const float* gamma,
                               void* dx, int64_t N, int C, const int64_t* n_dev, int dtype, mdlStream_t stream) {
    using namespace mdl;
    dtype &= MDL_DTYPE_MASK;
    int rc = bn_check(name, N, C, dtype, x);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    int64_t g = cdiv(N * (C / bn_width(C, dtype)), 256 * 4);
    if (g > 768) g = 768;          // 3 fat blocks per CU: the per-block statistics prologue amortises (measured 512..2048)
    if (g < 1) g = 1;
    const int W = bn_width(C, dtype);
    if (mask) {
        MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "%s: bf16 only", name);
        if (W == 8) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 8, true>), dim3((unsigned)g), dim3(bn_threads(C, 8)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, gamma, (bf16_t*)dx, N, C, n_dev);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4, true>), dim3((unsigned)g), dim3(bn_threads(C, 4)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, gamma, (bf16_t*)dx, N, C, n_dev);
        return check_launch(name);
    }
    if (dtype == MDL_BF16 && W == 8) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 8>), dim3((unsigned)g), dim3(bn_threads(C, 8)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, gamma, (bf16_t*)dx, N, C, n_dev);
    else if (dtype == MDL_BF16) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 4>), dim3((unsigned)g), dim3(bn_threads(C, 4)), 0, st, (const bf16_t*)dy, (const bf16_t*)x, save, sums, gamma, (bf16_t*)dx, N, C, n_dev);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 4>), dim3((unsigned)g), dim3(bn_threads(C, 4)), 0, st, (const float*)dy, (const float*)x, save, sums, gamma, (float*)dx, N, C, n_dev);
    return check_launch(name);
}

extern "C" int mdl_bn_bwd_apply_n(const void* dy, const void* x, const float* save, float* sums, const float* gamma,
                                  void* dx, int64_t N, int C, const int64_t* n_dev, int dtype, mdlStream_t stream) {
    return bn_bwd_apply_launch("mdl_bn_bwd_apply", false, dy, x, save, sums, gamma, dx, N, C, n_dev, dtype, stream);
}
extern "C" int mdl_bn_bwd_apply_relu_n(const void* dy, const void* x, const float* save, float* sums, const float* gamma,
                                       void* dx, int64_t N, int C, const int64_t* n_dev, int dtype, mdlStream_t stream) {
    return bn_bwd_apply_launch("mdl_bn_bwd_apply_relu", true, dy, x, save, sums, gamma, dx, N, C, n_dev, dtype, stream);
}
