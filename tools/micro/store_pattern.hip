// micro-benchmark: HBM write bandwidth of the dense layers' epilogue store pattern against 16-byte row stores.
// A 64-row x 160-column bf16 tile of a [N, LDO] output (LDO = 10000: NNConv's Y; LDO = 160: a 150-wide edge layer) is written
//   mode 0: like linear_act_kernel's epilogue — per 32x32 block 16 two-byte store instructions, lane = column, two row halves
//   mode 1: as 16-byte chunks of whole 320-byte row segments (what an LDS-transposed epilogue would issue)
// grid = column blocks x 64 row chunks as in mdl_linear_wide.  Prints GB/s of each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_store(unsigned short* __restrict__ out, long long N, int ldo, int ncb) {
    const int cb = blockIdx.x, rb = blockIdx.y, nrb = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = tid >> 6;
    unsigned short* o = out + cb * 160;
    const long long n_tiles = (N + 63) / 64;
    for (long long t = rb; t < n_tiles; t += nrb) {
        const long long r0 = t * 64;
        if (MODE == 0) {
            // wave wv: block row wv & 1, block columns (wv >> 1) + 2 j, j = 0..2 (5 column blocks over two waves: 3 + 2)
            const int mt = wv & 1;
            for (int nb = wv >> 1; nb < 5; nb += 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = r0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < N) o[row * ldo + nb * 32 + i] = (unsigned short)(r + lane);
                }
            }
        } else {
            typedef __attribute__((ext_vector_type(4))) unsigned u4;
            for (int c = tid; c < 64 * 20; c += 256) {
                const int row = c / 20, cc = c - row * 20;
                if (r0 + row < N) *reinterpret_cast<u4*>(o + (r0 + row) * ldo + cc * 8) = u4{(unsigned)c, 1u, 2u, 3u};
            }
        }
    }
}

int main() {
    const long long N = 61000;
    for (int ldo : {10000, 160}) {
        const int ncb = ldo / 160;
        const long long rows = ldo == 160 ? 1500000 : N;
        unsigned short* buf;
        hipMalloc(&buf, (size_t)rows * ldo * 2 + 4096);
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 6; ++it) {
                hipEventRecord(a);
                const int gy = ldo == 160 ? 512 : 64;
                if (mode == 0) hipLaunchKernelGGL(k_store<0>, dim3(ncb, gy), dim3(256), 0, 0, buf, rows, ldo, ncb);
                else hipLaunchKernelGGL(k_store<1>, dim3(ncb, gy), dim3(256), 0, 0, buf, rows, ldo, ncb);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (it > 0 && ms < best) best = ms;
            }
            printf("ldo %5d rows %8lld mode %d: %.1f us  %.2f TB/s\n", ldo, rows, mode, best * 1e3, (double)rows * ldo * 2 / (best * 1e-3) / 1e12);
        }
        hipFree(buf);
    }
    return 0;
}
