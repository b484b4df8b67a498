// mlp.hip — the post-FC head of the reference models as ONE launch per direction:
//     h_1 = relu(x W_1^T + b_1), ..., h_L = relu(h_{L-1} W_L^T + b_L),  y = h_L W_out^T + b_out
// (/root/reference/matdeeplearn/models/cgcnn.py:155-174: post_lin_list + lin_out on the pooled graph rows; 8192 rows x 64
// columns at the bench batch).  As separate layers this is ~6 launches forward and ~15 backward (dense layer, dW + db, dX,
// casts) of 3-8 us each that move a megabyte apiece: launch-bound.  Here every layer's weights sit in LDS for the whole
// kernel, a 64-row tile walks through all layers (MFMA 32x32x16, activated tiles handed on through LDS) and the backward
// walks them in reverse: dZ_l -> dW_l (+)= dZ_l^T in_l (contraction over the rows: both operands through the LDS transpose
// read), db_l, dIn_l = dZ_l W_l, dZ_{l-1} = dIn_l .* (h_{l-1} > 0).
// Shapes: bf16, every width <= 64 (zero-padded to 64 in LDS), 1 <= L + 1 <= 4 dense layers, ReLU between them.
#include "mdl_common.h"

namespace mdl {

constexpr int MLP_MAXL = 4;
constexpr int MLP_LD = 72;          // LDS row stride (bf16): 64 + 8, an odd number of 16-byte slots

struct MlpArgs {
    const bf16_t* x;                // [N, K0]
    const bf16_t* w[MLP_MAXL];      // layer l: [M_l, K_l] row-major, K_l = l ? M_{l-1} : K0
    const bf16_t* b[MLP_MAXL];      // [M_l] or null
    bf16_t* h[MLP_MAXL];            // fwd: outputs of every layer (the last one is y); bwd: the SAVED hidden outputs (inputs)
    const bf16_t* gy;               // bwd: dL/dy [N, M_last]
    bf16_t* dx;                     // bwd: [N, K0] or null
    float* dw[MLP_MAXL];            // bwd: [M_l, K_l] fp32, accumulated with atomics (caller zero-fills)
    float* db[MLP_MAXL];            // bwd: [M_l] fp32 or null
    int64_t N;
    int K0, NL;
    int M[MLP_MAXL];
    int f32io;                      // MDL_MLP_F32_IO: fwd writes the LAST layer's output as fp32 rows, bwd reads gy as fp32 rows
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* mlp_lds4_t;

// rows [r0, r0 + 64) of src [N, K] (bf16, dense rows) -> tile [64][MLP_LD], zero-filled past N and past K.
// Every load of a staging call is issued before the first LDS store (clamped, branch-free addresses: element 0 stands in for
// what lies outside) — a loop of load -> wait -> store is a chain of 8-16 memory round trips per tile, which at the reference's
// batch size (two tiles of work per workgroup) was most of the two kernels' time.
__device__ __forceinline__ void mlp_stage(bf16_t* tile, const bf16_t* src, int64_t r0, int64_t N, int K, int tid) {
    if (K & 1) {                                                       // odd width (a 1-column output): element by element
        bf16_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = tid + 256 * u, row = q >> 6, c = q & 63;
            const bool ok = r0 + row < N && c < K;
            v[u] = src[ok ? (r0 + row) * (int64_t)K + c : 0];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = tid + 256 * u, row = q >> 6, c = q & 63;
            tile[row * MLP_LD + c] = (r0 + row < N && c < K) ? v[u] : (bf16_t)0;
        }
        return;
    }
    const int k2 = K >> 1;
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = tid + 256 * u, row = q >> 5, d = q & 31;
        const bool ok = r0 + row < N && d < k2;
        v[u] = *reinterpret_cast<const unsigned*>(src + (ok ? (r0 + row) * (int64_t)K + 2 * d : 0));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = tid + 256 * u, row = q >> 5, d = q & 31;
        *reinterpret_cast<unsigned*>(tile + row * MLP_LD + 2 * d) = (r0 + row < N && d < k2) ? v[u] : 0u;
    }
}

// the same from fp32 rows (the output gradient under MDL_MLP_F32_IO), rounded to bf16 as the cast in front of the kernel did
__device__ __forceinline__ void mlp_stage_f32(bf16_t* tile, const float* src, int64_t r0, int64_t N, int K, int tid) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int q = tid + 256 * u, row = q >> 6, c = q & 63;
        const bool ok = r0 + row < N && c < K;
        v[u] = src[ok ? (r0 + row) * (int64_t)K + c : 0];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int q = tid + 256 * u, row = q >> 6, c = q & 63;
        tile[row * MLP_LD + c] = (r0 + row < N && c < K) ? f2bf(v[u]) : (bf16_t)0;
    }
}

// out block (mt, nt) of  tile_in[64][64] . wl[64 (out)][64 (in)]^T : lane = output column nt*32 + i, registers = rows
__device__ __forceinline__ f32x16 mlp_block(const bf16_t* tile_in, const bf16_t* wl, int mt, int nt, int i, int h, float bias) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile_in + (mt * 32 + i) * MLP_LD + 16 * kk + 8 * h);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(wl + (nt * 32 + i) * MLP_LD + 16 * kk + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(256, 2) void mlp_head_fwd_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wl = reinterpret_cast<bf16_t*>(smem);                     // [NL][64][LD]
    bf16_t* ta = wl + p.NL * 64 * MLP_LD;
    bf16_t* tb = ta + 64 * MLP_LD;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wv & 1, nt = wv >> 1;
    {   // weights -> LDS: every layer's loads in flight before the first store (see mlp_stage)
        unsigned wv_[MLP_MAXL][8];
#pragma unroll
        for (int l = 0; l < MLP_MAXL; ++l) {
            const int K = l < p.NL ? (l ? p.M[l - 1] : p.K0) : 0, M = l < p.NL ? p.M[l] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + 256 * u, row = q >> 5, d = q & 31;
                unsigned v = 0u;
                if (row < M && 2 * d < K) v = *reinterpret_cast<const unsigned*>(p.w[l] + (int64_t)row * K + 2 * d);
                wv_[l][u] = v;
            }
        }
#pragma unroll
        for (int l = 0; l < MLP_MAXL; ++l) {
            if (l >= p.NL) break;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + 256 * u, row = q >> 5, d = q & 31;
                *reinterpret_cast<unsigned*>(wl + (l * 64 + row) * MLP_LD + 2 * d) = wv_[l][u];
            }
        }
    }
    const int64_t n_tiles = (p.N + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64;
        __syncthreads();                                               // previous tile done with both tiles; weights in place
        mlp_stage(ta, p.x, r0, p.N, p.K0, tid);
        __syncthreads();
        bf16_t* tin = ta;
        bf16_t* tout = tb;
        for (int l = 0; l < p.NL; ++l) {
            const int M = p.M[l], col = nt * 32 + i;
            const float bias = (p.b[l] && col < M) ? bf2f(p.b[l][col]) : 0.0f;
            const f32x16 acc = mlp_block(tin, wl + l * 64 * MLP_LD, mt, nt, i, h, bias);
            const bool relu = l + 1 < p.NL;
            bf16_t* const out = p.h[l];
            float* const out32 = (!relu && p.f32io) ? reinterpret_cast<float*>(p.h[l]) : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                float v = acc[r];
                if (relu) v = v > 0.0f ? v : 0.0f;
                const bf16_t hv = f2bf(v);
                tout[row * MLP_LD + col] = hv;                         // (columns past M: zero weights rows -> bias 0 -> relu(0) = 0)
                if (col < M && r0 + row < p.N) {
                    if (out32) out32[(r0 + row) * (int64_t)M + col] = bf2f(hv);     // (the value the bf16 output + cast gave)
                    else out[(r0 + row) * (int64_t)M + col] = hv;
                }
            }
            __syncthreads();
            bf16_t* t = tin; tin = tout; tout = t;
        }
    }
}

__global__ __launch_bounds__(256, 1) void mlp_head_bwd_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* wt = reinterpret_cast<bf16_t*>(smem);                     // [NL][64 (in)][LD]: TRANSPOSED weights (row = input column)
    bf16_t* tin = wt + p.NL * 64 * MLP_LD;                             // [NL][64][LD]: input tile of every layer (x, h_0, ...)
    bf16_t* da = tin + p.NL * 64 * MLP_LD;                             // [64][LD] dZ of the current layer
    bf16_t* dbuf = da + 64 * MLP_LD;                                   // [64][LD] dZ of the next (lower) layer
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wv & 1, nt = wv >> 1, t16 = i & 15;
    // wt[l][k][m] = W_l[m][k]: the rows are READ as coalesced dwords — every load of every layer in flight before the first is
    // used — and scattered into the transposed LDS copy with two-byte stores.  (The first version gathered element by element
    // from global memory, 16 dependent two-byte loads per thread and layer: most of the kernel's 31 us at the reference's batch
    // size, where a workgroup has two tiles of work, and a fixed ~15 us at 8192 rows.)
    {
        unsigned wv_[MLP_MAXL][8];
#pragma unroll
        for (int l = 0; l < MLP_MAXL; ++l) {
            const int K = l < p.NL ? (l ? p.M[l - 1] : p.K0) : 0, M = l < p.NL ? p.M[l] : 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + 256 * u, m = q >> 5, d = q & 31;
                unsigned v = 0u;
                if (m < M && 2 * d + 1 < K) v = *reinterpret_cast<const unsigned*>(p.w[l] + (int64_t)m * K + 2 * d);
                else if (m < M && 2 * d < K) v = (unsigned)p.w[l][(int64_t)m * K + 2 * d];        // (odd K: the last column alone)
                wv_[l][u] = v;
            }
        }
#pragma unroll
        for (int l = 0; l < MLP_MAXL; ++l) {
            if (l >= p.NL) break;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int q = tid + 256 * u, m = q >> 5, d = q & 31;
                wt[(l * 64 + 2 * d) * MLP_LD + m] = (bf16_t)(wv_[l][u] & 0xffffu);
                wt[(l * 64 + 2 * d + 1) * MLP_LD + m] = (bf16_t)(wv_[l][u] >> 16);
            }
        }
    }
    f32x16 dwacc[MLP_MAXL];
#pragma unroll
    for (int l = 0; l < MLP_MAXL; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwacc[l][r] = 0.0f;
    float dbacc[MLP_MAXL] = {0.0f, 0.0f, 0.0f, 0.0f};               // thread t: sum of column t & 63 of dZ_l over the 16 rows of row block t >> 6
    const int64_t n_tiles = (p.N + 63) / 64;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * 64;
        __syncthreads();
        for (int l = 0; l < p.NL; ++l)
            mlp_stage(tin + l * 64 * MLP_LD, l ? p.h[l - 1] : p.x, r0, p.N, l ? p.M[l - 1] : p.K0, tid);
        if (p.f32io) mlp_stage_f32(da, reinterpret_cast<const float*>(p.gy), r0, p.N, p.M[p.NL - 1], tid);
        else mlp_stage(da, p.gy, r0, p.N, p.M[p.NL - 1], tid);
        __syncthreads();
        bf16_t* dcur = da;
        bf16_t* dnext = dbuf;
#pragma unroll
        for (int lq = 0; lq < MLP_MAXL; ++lq) {
            const int l = p.NL - 1 - lq;
            if (l < 0) break;
            const bf16_t* const in = tin + l * 64 * MLP_LD;
            // dW_l block (mt: output rows m, nt: input columns k) += dZ_l^T . in_l over the tile's 64 rows
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int roff = 16 * ks + 8 * h + (t16 >> 2), coff = (i & 16) + 4 * (t16 & 3);
                const bf16_t* pa = dcur + roff * MLP_LD + mt * 32 + coff;
                const bf16_t* pb = in + roff * MLP_LD + nt * 32 + coff;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mlp_lds4_t)pa), a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mlp_lds4_t)(pa + 4 * MLP_LD));
                const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mlp_lds4_t)pb), b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((mlp_lds4_t)(pb + 4 * MLP_LD));
                const bf16x8 af = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                const bf16x8 bf = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                dwacc[lq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, dwacc[lq], 0, 0, 0);
            }
            {
                float s = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += bf2f(dcur[(16 * (tid >> 6) + r) * MLP_LD + (tid & 63)]);
                dbacc[lq] += s;
            }
            // dIn_l = dZ_l . W_l : lane = input column nt*32 + i, registers = rows
            if (l > 0 || p.dx) {
                const f32x16 acc = mlp_block(dcur, wt + l * 64 * MLP_LD, mt, nt, i, h, 0.0f);
                const int col = nt * 32 + i, K = l ? p.M[l - 1] : p.K0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mt * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
                    if (l > 0) {
                        const bool on = bf2f(in[row * MLP_LD + col]) > 0.0f;       // relu'(h_{l-1}) from its saved output
                        dnext[row * MLP_LD + col] = f2bf(on ? acc[r] : 0.0f);
                    } else if (col < K && r0 + row < p.N) {
                        p.dx[(r0 + row) * (int64_t)K + col] = f2bf(acc[r]);
                    }
                }
            }
            __syncthreads();
            bf16_t* t = dcur; dcur = dnext; dnext = t;
        }
    }
    // the four row blocks' bias sums -> one per column, added up in a fixed order (MDL_DETERMINISTIC runs one workgroup: every sum
    // then receives its terms from one thread in program order)
    {
        float* red = reinterpret_cast<float*>(da);                     // [MLP_MAXL][4][64] floats = 4 KB of the dZ tile (9 KB)
        __syncthreads();
#pragma unroll
        for (int lq = 0; lq < MLP_MAXL; ++lq) red[(lq * 4 + (tid >> 6)) * 64 + (tid & 63)] = dbacc[lq];
        __syncthreads();
#pragma unroll
        for (int lq = 0; lq < MLP_MAXL; ++lq)
            dbacc[lq] = tid < 64 ? ((red[(lq * 4 + 0) * 64 + tid] + red[(lq * 4 + 1) * 64 + tid]) + red[(lq * 4 + 2) * 64 + tid]) + red[(lq * 4 + 3) * 64 + tid] : 0.0f;
    }
    // flush: dW blocks (rows = m in registers, lane = input column) and the bias sums
#pragma unroll
    for (int lq = 0; lq < MLP_MAXL; ++lq) {
        const int l = p.NL - 1 - lq;
        if (l < 0) break;
        const int K = l ? p.M[l - 1] : p.K0, M = p.M[l], k = nt * 32 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + 4 * h + (r & 3) + 8 * (r >> 2);
            if (m < M && k < K) unsafeAtomicAdd(p.dw[l] + (int64_t)m * K + k, dwacc[lq][r]);
        }
        if (tid < 64 && tid < M && p.db[l]) unsafeAtomicAdd(p.db[l] + tid, dbacc[lq]);
    }
}

static int mlp_check(const char* name, const MlpArgs& a, int dtype) {
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "%s: bf16 only", name);
    MDL_REQUIRE(a.NL >= 1 && a.NL <= MLP_MAXL && a.K0 >= 2 && a.K0 <= 64 && a.K0 % 2 == 0, MDL_E_UNSUPP,
                "%s: 1..4 dense layers, even input width <= 64 (got %d layers, K0 = %d)", name, a.NL, a.K0);
    for (int l = 0; l < a.NL; ++l) {
        MDL_REQUIRE(a.M[l] >= 1 && a.M[l] <= 64 && (l + 1 == a.NL || a.M[l] % 2 == 0), MDL_E_UNSUPP,
                    "%s: widths <= 64, hidden widths even (layer %d: %d)", name, l, a.M[l]);
        MDL_REQUIRE(a.w[l] && reinterpret_cast<uintptr_t>(a.w[l]) % 4 == 0, MDL_E_ARG, "%s: null / misaligned weight %d", name, l);
    }
    MDL_REQUIRE(a.N >= 0 && (a.N == 0 || (a.x && reinterpret_cast<uintptr_t>(a.x) % 4 == 0)), MDL_E_ARG, "%s: bad arguments", name);
    return MDL_OK;
}

}  // namespace mdl

// w / b / h: NL pointers each (b entries may be NULL); h[l] receives layer l's output [N, M[l]] (h[NL-1] = y; with
// dtype | MDL_MLP_F32_IO that last one is an fp32 buffer — the prediction the fp32 loss reads, no cast launch behind the head)
extern "C" int mdl_mlp_head_fwd(const void* x, const void* const* w, const void* const* b, void* const* h, int64_t N, int K0,
                                int NL, const int* M, int dtype, mdlStream_t stream) {
    using namespace mdl;
    MlpArgs a = {};
    a.x = (const bf16_t*)x; a.N = N; a.K0 = K0; a.NL = NL;
    MDL_REQUIRE(NL >= 1 && NL <= MLP_MAXL && w && b && h && M, MDL_E_ARG, "mdl_mlp_head_fwd: bad arguments");
    for (int l = 0; l < NL; ++l) { a.w[l] = (const bf16_t*)w[l]; a.b[l] = (const bf16_t*)b[l]; a.h[l] = (bf16_t*)h[l]; a.M[l] = M[l]; }
    a.f32io = (dtype & MDL_MLP_F32_IO) != 0;
    dtype &= MDL_DTYPE_MASK;
    int rc = mlp_check("mdl_mlp_head_fwd", a, dtype);
    if (rc) return rc;
    for (int l = 0; l < NL; ++l) MDL_REQUIRE(N == 0 || a.h[l], MDL_E_ARG, "mdl_mlp_head_fwd: null output %d", l);
    if (N == 0) return MDL_OK;
    const int lds = (NL + 2) * 64 * MLP_LD * 2;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(mlp_head_fwd_kernel), lds);
    if (e != hipSuccess) { set_error("mdl_mlp_head_fwd: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    int64_t grid = cdiv(N, 64);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(mlp_head_fwd_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("mdl_mlp_head_fwd");
}

// h: the NL-1 saved hidden outputs (h[l] = output of layer l, l < NL-1); gy [N, M[NL-1]] (fp32 rows with MDL_MLP_F32_IO); dw[l] [M[l], K_l] / db[l] [M[l]] fp32,
// zero-filled by the caller (db entries and dx may be NULL)
extern "C" int mdl_mlp_head_bwd(const void* x, const void* const* w, const void* const* h, const void* gy, void* dx,
                                float* const* dw, float* const* db, int64_t N, int K0, int NL, const int* M, int dtype,
                                mdlStream_t stream) {
    using namespace mdl;
    MlpArgs a = {};
    a.x = (const bf16_t*)x; a.gy = (const bf16_t*)gy; a.dx = (bf16_t*)dx; a.N = N; a.K0 = K0; a.NL = NL;
    MDL_REQUIRE(NL >= 1 && NL <= MLP_MAXL && w && h && dw && db && M, MDL_E_ARG, "mdl_mlp_head_bwd: bad arguments");
    for (int l = 0; l < NL; ++l) {
        a.w[l] = (const bf16_t*)w[l]; a.M[l] = M[l]; a.dw[l] = dw[l]; a.db[l] = db[l];
        a.h[l] = l + 1 < NL ? (bf16_t*)const_cast<void*>(h[l]) : nullptr;
    }
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;      // one workgroup: one add per dw / db element
    a.f32io = (dtype & MDL_MLP_F32_IO) != 0;
    dtype &= MDL_DTYPE_MASK;
    int rc = mlp_check("mdl_mlp_head_bwd", a, dtype);
    if (rc) return rc;
    MDL_REQUIRE(N == 0 || (gy && reinterpret_cast<uintptr_t>(gy) % ((M[NL - 1] % 2 && !a.f32io) ? 2 : 4) == 0), MDL_E_ARG,
                "mdl_mlp_head_bwd: null / misaligned gy");
    for (int l = 0; l < NL; ++l) {
        MDL_REQUIRE(N == 0 || a.dw[l], MDL_E_ARG, "mdl_mlp_head_bwd: null dw %d", l);
        MDL_REQUIRE(N == 0 || l + 1 == NL || (a.h[l] && reinterpret_cast<uintptr_t>(a.h[l]) % 4 == 0), MDL_E_ARG, "mdl_mlp_head_bwd: null h %d", l);
    }
    if (N == 0) return MDL_OK;
    const int lds = (2 * NL + 2) * 64 * MLP_LD * 2;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(mlp_head_bwd_kernel), lds);
    if (e != hipSuccess) { set_error("mdl_mlp_head_bwd: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    int64_t grid = cdiv(N, 64);
    if (grid > 256) grid = 256;
    if (det) grid = 1;
    hipLaunchKernelGGL(mlp_head_bwd_kernel, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("mdl_mlp_head_bwd");
}
