#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned u2;
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ int d_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
// acc[r] at lane (n = lane&31, h = lane>>5) encodes (edge = d_row(r,h), col = n): value = edge*100 + col
__global__ void k(float* out) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    float acc[16], v[16];
    for (int r = 0; r < 16; ++r) acc[r] = d_row(r, h) * 100.0f + n;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        float f = acc[r], s = acc[r + 8];
        swap16(f, s);
        float m = f * 10000.0f + s;     // f = edge*100 + ch (ch<16), s = edge*100 + 16 + ch  -> m encodes both
        float one = -1.0f;
        swap16(m, one);
        v[r] = m; v[r + 8] = one;
    }
    for (int r = 0; r < 16; ++r) out[r * 64 + lane] = v[r];
}
int main() {
    float* d; hipMalloc(&d, 16 * 64 * 4); k<<<1, 64>>>(d);
    float hbuf[16 * 64]; hipMemcpy(hbuf, d, sizeof(hbuf), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 16; ++r) for (int lane = 0; lane < 64; ++lane) {
        int n = lane & 31, h = lane >> 5; float got = hbuf[r * 64 + lane], exp;
        int edge = (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n < 16) exp = (edge * 100.0f + n) * 10000.0f + (edge * 100.0f + 16 + n); else exp = -1.0f;
        if (got != exp) { if (bad < 10) printf("r %d lane %d got %.1f exp %.1f\n", r, lane, got, exp); ++bad; }
    }
    printf("bad = %d\n", bad);
    return 0;
}
