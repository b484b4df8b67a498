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
__device__ __forceinline__ bf16_t f2bf(float f) {               // v_cvt_pk_bf16_f32 (gfx950): RNE, NaN quieted
    const __bf16 r = (__bf16)f;
    return __builtin_bit_cast(bf16_t, r);
}

// two floats -> packed bf16 pair (v_cvt_pk_bf16_f32 on gfx950, round-to-nearest-even)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    bf16x2_hw r = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_hw);
    return *reinterpret_cast<unsigned*>(&r);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int dtype = MDL_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float rnd(float v) { return v; }            // the value st() leaves in memory
};
template <> struct Elem<bf16_t> {
    static constexpr int dtype = MDL_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

// ---- 16-byte row vectors (W elements) <-> floats ------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<bf16_t> {
    static constexpr int W = 8;
    typedef bf16x8 raw;
    __device__ static __forceinline__ void ld(const bf16_t* p, float* v) {
        const raw r = *reinterpret_cast<const raw*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = bf2f((bf16_t)r[j]);
    }
    __device__ static __forceinline__ void st(bf16_t* p, const float* v) {
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        *reinterpret_cast<u4*>(p) = u4{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7])};
    }
};
template <> struct Vec<float> {
    static constexpr int W = 4;
    __device__ static __forceinline__ void ld(const float* p, float* v) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(p);
        v[0] = r[0]; v[1] = r[1]; v[2] = r[2]; v[3] = r[3];
    }
    __device__ static __forceinline__ void st(float* p, const float* v) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
};

// ---- row vectors of a chosen width: 16 bytes where the channel count allows it, 8 bytes (4 bf16) for widths like 100 / 150 ----
template <typename T, int W> struct VecW;
template <> struct VecW<bf16_t, 8> : Vec<bf16_t> {};
template <> struct VecW<float, 4> : Vec<float> {};
template <> struct VecW<bf16_t, 4> {
    static constexpr int W = 4;
    __device__ static __forceinline__ void ld(const bf16_t* p, float* v) {
        const bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bf2f((bf16_t)r[j]);
    }
    __device__ static __forceinline__ void st(bf16_t* p, const float* v) {
        typedef __attribute__((ext_vector_type(2))) unsigned u2;
        *reinterpret_cast<u2*>(p) = u2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
    }
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
// PRECISE (fp32 parity mode): ocml expf/log1pf on the true pre-activations.
// FAST (bf16 mode): the packed weights/biases are pre-scaled by log2(e), so the MFMA delivers
// t = pre*log2(e) and everything runs on the hardware base-2 transcendentals (v_exp/v_log/v_rcp):
//   sigmoid(pre)  = 1 / (1 + 2^-t)
//   softplus(pre) = ln2 * sp2(t),  sp2(t) = max(t,0) + log2(1 + 2^-|t|)
// The ln2 factor is folded into the per-node epilogue scale by the callers.
// torch's softplus switches to identity above threshold 20, where both forms agree to < 2.1e-9.
constexpr float LOG2E_F = 1.4426950408889634f;
constexpr float LN2_F = 0.6931471805599453f;

template <bool FAST> struct Gate;
template <> struct Gate<false> {
    static constexpr float W_SCALE = 1.0f;     // weight pre-scale
    static constexpr float M_SCALE = 1.0f;     // message post-scale
    __device__ static __forceinline__ float sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
    __device__ static __forceinline__ float softplus_u(float x) { return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x))); }
    // softplus (unscaled) and sigmoid of the same argument
    __device__ static __forceinline__ void softplus_sigmoid(float x, float& sp_u, float& sg) {
        const float ea = expf(-fabsf(x));
        sp_u = fmaxf(x, 0.0f) + log1pf(ea);
        const float r = 1.0f / (1.0f + ea);
        sg = x >= 0.0f ? r : ea * r;
    }
    // everything the backward needs: sigmoid(f), softplus_u(s), sigmoid(s)
    __device__ static __forceinline__ void deriv(float f, float sv, float& sf, float& sp_u, float& ss) {
        sf = sigmoid(f);
        softplus_sigmoid(sv, sp_u, ss);
    }
    __device__ static __forceinline__ void deriv2(float f, float sv, float& sf, float& sp_u, float& ss) { deriv(f, sv, sf, sp_u, ss); }
};
template <> struct Gate<true> {
    static constexpr float W_SCALE = LOG2E_F;
    static constexpr float M_SCALE = LN2_F;
    // NO inline asm in the gate: the compiler's hazard recogniser does not look inside asm statements, and both kinds of
    // producer here need wait states before a VALU consumer — MFMA results (the arguments of these functions) and the
    // results of the transcendental unit (v_exp / v_log / v_rcp).  An asm `v_max` right behind the MFMA chain read the
    // accumulator before the matrix core had written it back; an asm `v_min` behind a `v_exp` read lanes the
    // transcendental pipe had not finished.  fmaxf / fminf cost two instructions each (IEEE mode canonicalises first), so:
    //   max(t, 0) + c   =  0.5 * (t + |t|) + c     one add (|t| is a source modifier) + one fma, both exact
    //   2^-max(f, -126) =  min(2^-f, 2^126)        as an INTEGER min on the bit pattern (2^-f >= 0: same order), one op
    __device__ static __forceinline__ float relu_plus(float t, float c) { return fmaf(0.5f, t + fabsf(t), c); }
    // min(2^-f, 2^126): keeps (1 + 2^-f)(1 + 2^-|s|) finite
    __device__ static __forceinline__ float exp2_neg_capped(float f) {
        const unsigned e = __float_as_uint(__builtin_amdgcn_exp2f(-f));
        return __uint_as_float(e < 0x7e800000u ? e : 0x7e800000u);
    }
    __device__ static __forceinline__ float sigmoid(float t) {
        return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
    }
    __device__ static __forceinline__ float softplus_u(float t) {
        return relu_plus(t, __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(t))));
    }
    __device__ static __forceinline__ void softplus_sigmoid(float t, float& sp_u, float& sg) {
        const float ea = __builtin_amdgcn_exp2f(-fabsf(t));
        const float l = 1.0f + ea;
        sp_u = relu_plus(t, __builtin_amdgcn_logf(l));
        const float r = __builtin_amdgcn_rcpf(l);
        sg = t >= 0.0f ? r : ea * r;
    }
    // Transcendentals are the scarce resource (a wave64 v_exp/v_log/v_rcp occupies the SIMD ~8x longer
    // than an FMA): the two reciprocals 1/(1+2^-f) and 1/(1+2^-|s|) share ONE v_rcp of the product.
    __device__ static __forceinline__ void deriv(float f, float sv, float& sf, float& sp_u, float& ss) {
        const float a1 = 1.0f + exp2_neg_capped(f);
        const float ea = __builtin_amdgcn_exp2f(-fabsf(sv));
        const float l = 1.0f + ea;
        const float r = __builtin_amdgcn_rcpf(a1 * l);
        sf = r * l;
        const float rl = r * a1;
        sp_u = relu_plus(sv, __builtin_amdgcn_logf(l));
        ss = sv >= 0.0f ? rl : ea * rl;
    }
    // The same three values without a select, an |s| or a second add (edge-per-lane backward, where the gate arithmetic is
    // the instruction count that bounds the kernel):
    //   u = min(2^-f, 2^63), es = min(2^s, 2^63)      integer min on the bit patterns (both >= 0: same order); keeps the
    //                                                  product (1 + u)(1 + es) finite
    //   r = 1 / ((1 + u)(1 + es));  sf = r (1 + es) = sigmoid(f);  ss = es r (1 + u) = sigmoid(s)
    //   sp_u = max(log2(1 + es), s) = log2(1 + 2^s): the max restores the exact value where es was capped; taken as a SIGNED
    //          INTEGER max of the bit patterns (log2(1 + es) >= 0: a negative s is a negative integer, two non-negative floats
    //          order like their patterns) — fmaxf costs a canonicalising second instruction
    __device__ static __forceinline__ void deriv2(float f, float sv, float& sf, float& sp_u, float& ss) {
        const unsigned ub = __float_as_uint(__builtin_amdgcn_exp2f(-f));
        const unsigned eb = __float_as_uint(__builtin_amdgcn_exp2f(sv));
        const float a1 = 1.0f + __uint_as_float(ub < 0x5f000000u ? ub : 0x5f000000u);
        const float es = __uint_as_float(eb < 0x5f000000u ? eb : 0x5f000000u);
        const float l = 1.0f + es;
        const float r = __builtin_amdgcn_rcpf(a1 * l);
        sf = r * l;
        ss = es * (r * a1);
        const int lg = (int)__float_as_uint(__builtin_amdgcn_logf(l)), si = (int)__float_as_uint(sv);
        sp_u = __uint_as_float((unsigned)(lg > si ? lg : si));
    }
};

// ---- error plumbing ----------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);
hipError_t set_max_dynamic_lds(const void* kernel, int bytes);   // cached hipFuncSetAttribute(MaxDynamicSharedMemorySize)

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace mdl

#define MDL_REQUIRE(cond, code, ...)        \
    do {                                    \
        if (!(cond)) {                      \
            mdl::set_error(__VA_ARGS__);    \
            return (code);                  \
        }                                   \
    } while (0)
