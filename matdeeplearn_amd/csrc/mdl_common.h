// mdl_common.h — shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
// Internal to libmdl_hip.so; the public C ABI is include/mdl_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mdl_hip.h"

namespace mdl {

constexpr int WAVE = 64;

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one 32x32x16 A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator tile
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned short bf16_t;                               // raw bf16 bits

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) ---------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = MDL_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int dtype = MDL_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// ---- D-layout of the 32x32 MFMA accumulator (dtype independent on gfx950) ------------------
// lane l, register r  ->  row = (r&3) + 8*(r>>2) + 4*(l>>5),  col = l&31
__device__ __forceinline__ int d_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Make LDS traffic of ONE wave visible to its other lanes: the DS queue of a wave is in-order,
// so only the compiler has to be stopped from moving accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- gate math ---------------------------------------------------------------------------
// FAST: v_exp/v_log/v_rcp based (bf16 mode).  PRECISE: ocml expf/log1pf (fp32 parity mode).
template <bool FAST> __device__ __forceinline__ float sigmoidf_(float x) {
    if (FAST) return __frcp_rn(1.0f + __expf(-x));
    return 1.0f / (1.0f + expf(-x));
}
// softplus(x) = max(x,0) + log1p(exp(-|x|)); torch switches to identity above threshold 20,
// where the two agree to < 2.1e-9 absolute, i.e. below fp32 resolution of x itself.
template <bool FAST> __device__ __forceinline__ float softplusf_(float x) {
    float a = fabsf(x);
    if (FAST) return fmaxf(x, 0.0f) + __logf(1.0f + __expf(-a));
    return fmaxf(x, 0.0f) + log1pf(expf(-a));
}

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mdl

#define MDL_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            mdl::set_error(__VA_ARGS__);    \
            return (code);                  \
        }                                   \
    } while (0)
