// loss.hip — training loss and its gradient in ONE launch.
// The reference evaluates `getattr(F, loss)(output, data.y)` (matdeeplearn/training/training.py:44-47: l1_loss by default,
// config.yml:117) and lets autograd run sub / abs / mean and, backwards, sign / div / expand: eight launches that each
// touch B <= 8192 floats — at the reference's batch size a tenth of the whole step's launches.  Here one workgroup
// computes the mean loss and d loss / d pred together (fp32, reduction = "mean" over all elements).
#include "mdl_common.h"

namespace mdl {

// kind 0: l1 (|p - y|, gradient sign(p - y) / n, 0 at p == y like torch);  kind 1: mse ((p - y)^2, gradient 2 (p - y) / n)
// n_total >= n: pred / grad hold n_total elements of which the first n count (a padded static batch carries a dummy graph
// behind the B real ones): the gradient of the rest is written as exact zeros, so no slice sits between the model and the loss
__global__ __launch_bounds__(1024) void loss_fwd_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ y, int64_t n,
                                                            int64_t n_total, int kind, float* __restrict__ loss,
                                                            float* __restrict__ grad) {
    __shared__ float red[16];
    const float inv = 1.0f / (float)n;
    float s = 0.0f;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = pred[i] - y[i];
        if (kind == 0) {
            s += fabsf(d);
            grad[i] = d > 0.0f ? inv : (d < 0.0f ? -inv : 0.0f);
        } else {
            s += d * d;
            grad[i] = 2.0f * d * inv;
        }
    }
    for (int64_t i = n + threadIdx.x; i < n_total; i += blockDim.x) grad[i] = 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = threadIdx.x < (blockDim.x >> 6) ? red[threadIdx.x] : 0.0f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (threadIdx.x == 0) *loss = t * inv;
    }
}

}  // namespace mdl

extern "C" int mdl_loss_fwd_bwd(const float* pred, const float* y, int64_t n, int kind, float* loss, float* grad,
                                mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(n >= 1 && pred && y && loss && grad, MDL_E_ARG, "mdl_loss_fwd_bwd: bad arguments");
    MDL_REQUIRE(kind == 0 || kind == 1, MDL_E_UNSUPP, "mdl_loss_fwd_bwd: kind must be 0 (l1) or 1 (mse)");
    hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, y, n, n, kind, loss, grad);
    return check_launch("mdl_loss_fwd_bwd");
}

// the same over the first n of n_total predictions: grad [n_total], zero past n; y [n]
extern "C" int mdl_loss_fwd_bwd_rows(const float* pred, const float* y, int64_t n, int64_t n_total, int kind, float* loss,
                                     float* grad, mdlStream_t stream) {
    using namespace mdl;
    MDL_REQUIRE(n >= 1 && n_total >= n && pred && y && loss && grad, MDL_E_ARG, "mdl_loss_fwd_bwd_rows: bad arguments");
    MDL_REQUIRE(kind == 0 || kind == 1, MDL_E_UNSUPP, "mdl_loss_fwd_bwd_rows: kind must be 0 (l1) or 1 (mse)");
    hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, y, n, n_total, kind, loss, grad);
    return check_launch("mdl_loss_fwd_bwd_rows");
}
