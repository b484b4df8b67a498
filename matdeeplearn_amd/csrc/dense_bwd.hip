// dense_bwd.hip — backward of a tall dense layer y = act(x W^T + b) in ONE streaming pass (bf16 in, fp32 dW / db out):
//     g' = g .* act'(y)          (applied while the g tile is staged; relu / shifted softplus from the saved output y)
//     dW[M, K] += g'^T . x       db[M] += column sums of g'         dX[N, K] = g' . W   (optionally .* act_in'(x))
// Replaces the pair mdl_linear_act_in (dX) + mdl_gemm_tn_act (dW, db) for the edge-level Linears of the reference's
// MEGNet blocks and SchNet filter networks (/root/reference/matdeeplearn/models/megnet.py:41-56,84-101,
// /root/reference/matdeeplearn/models/schnet.py:81 via torch_geometric InteractionBlock.mlp), where autograd runs
// threshold_backward / softplus_backward, mm (dX) and mm (dW) as separate passes over [E, M] rows.  The pair read g and y
// twice; this kernel reads g, y, x once and writes dX: 4 row streams instead of 6 (3 instead of 6 with the activation
// hand-over `xout`, see gemm_tn_stream.inc).  HBM bound: N * (2M + 2K) * 2 bytes (ACT != 0), N * (M + 2K) * 2 otherwise.
#include "mdl_common.h"

namespace mdl {
#include "gemm_tn_stream.inc"
}  // namespace mdl

extern "C" int mdl_dense_bwd(const void* g, int64_t ldg, int M, const void* y, int64_t ldy, int act, const void* x, int64_t ldx,
                             int K, const void* w, void* dx, int64_t lddx, int xout, void* gm, float* dw, float* db,
                             int64_t N, int dtype, mdlStream_t stream) {
    return mdl_dense_bwd_ex(g, ldg, M, y, ldy, act, x, ldx, K, w, dx, lddx, xout, gm, dw, db, nullptr, N, dtype, stream);
}

extern "C" int mdl_dense_bwd_ex(const void* g, int64_t ldg, int M, const void* y, int64_t ldy, int act, const void* x, int64_t ldx,
                                int K, const void* w, void* dx, int64_t lddx, int xout, void* gm, float* dw, float* db, void* scratch,
                                int64_t N, int dtype, mdlStream_t stream) {
    using namespace mdl;
    const bool det = (dtype & MDL_DETERMINISTIC) != 0;      // one workgroup: every dw / db element gets one add from one wave
    dtype &= MDL_DTYPE_MASK;
    MDL_REQUIRE(dtype == MDL_BF16, MDL_E_UNSUPP, "mdl_dense_bwd: bf16 only");
    MDL_REQUIRE(act >= 0 && act <= 2 && (act == 0 || (y && ldy >= M && ldy % 2 == 0 && reinterpret_cast<uintptr_t>(y) % 4 == 0)),
                MDL_E_ARG, "mdl_dense_bwd: act must be 0, 1 (relu) or 2 (shifted softplus), with the saved output y for 1 / 2");
    MDL_REQUIRE(xout >= 0 && xout <= 2, MDL_E_ARG, "mdl_dense_bwd: xout must be 0, 1 (relu) or 2 (shifted softplus)");
    MDL_REQUIRE(M >= 34 && M <= 160 && K >= 34 && K <= (db ? 158 : 160) && M % 2 == 0 && K % 2 == 0, MDL_E_UNSUPP,
                "mdl_dense_bwd: need even 34<=M<=160, 34<=K<=160 (158 with db) (got %d, %d)", M, K);
    MDL_REQUIRE(N >= 0 && ldg >= M && ldx >= K && lddx >= K && ldg % 2 == 0 && ldx % 2 == 0 && (N == 0 || (g && x && w && dx && dw)),
                MDL_E_ARG, "mdl_dense_bwd: bad arguments");
    MDL_REQUIRE(reinterpret_cast<uintptr_t>(g) % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 4 == 0 &&
                    reinterpret_cast<uintptr_t>(w) % 2 == 0 && reinterpret_cast<uintptr_t>(dx) % 2 == 0 &&
                    reinterpret_cast<uintptr_t>(gm) % 4 == 0,
                MDL_E_ARG, "mdl_dense_bwd: misaligned pointer");
    if (N == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    int mt = (M + 31) / 32, nt = (K + (db ? 1 : 0) + 31) / 32;       // db rides in padding column K of the x tile
    if (mt == 3) mt = 4;
    if (nt == 3) nt = 4;
    const int grid_cap = det ? 1 : ((N >= (1 << 20) && mt * nt < 25) ? 512 : 256);
    int64_t grid = cdiv(N, 64);
    if (grid > grid_cap) grid = grid_cap;
    const int lds = (64 * (32 * mt + 8) + 64 * (32 * nt + 8) + 32 * nt * (32 * mt + 8)) * 2;
    // scratch: dW / db blocks as plain stores + a reduce launch instead of atomics from every workgroup (gemm_tn_stream.inc `part`)
    float* part = (scratch && !det && grid >= 32 && reinterpret_cast<uintptr_t>(scratch) % 16 == 0) ? static_cast<float*>(scratch) : nullptr;
#define MDL_DB_K(MT_, NT_, A_, NW_)                                                                                          \
    do {                                                                                                                     \
        auto kf = gemm_tn_stream_kernel<MT_, NT_, A_, true, NW_>;                                                            \
        hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), lds);                                          \
        if (e != hipSuccess) { set_error("mdl_dense_bwd: LDS attribute (%d B): %s", lds, hipGetErrorString(e)); return MDL_E_LAUNCH; } \
        hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(64 * NW_), lds, st, (const bf16_t*)g, (int)ldg, M, (const bf16_t*)x, \
                           (int)ldx, K, dw, db, N, (const bf16_t*)y, (int)ldy, (const bf16_t*)w, (bf16_t*)dx, (int)lddx, xout, (bf16_t*)gm, part); \
    } while (0)
// waves per workgroup (measured on 1.5e6 rows, tools/bench_dense.py): a 5-tile side needs 8 (one 4-wave workgroup per CU ran
// 150 x 150 in 611 us, 8 waves in 434); 4-tile shapes take 8 without an activation staging (288 -> 225 us) and 4 with one
// (the y tile's extra registers and staging work cost the 8-wave form its second workgroup's overlap: 285 vs 334 us)
#define MDL_DB(MT_, NT_)                                                                                                     \
    do {                                                                                                                     \
        constexpr bool wide = (MT_ > 4 || NT_ > 4);                                                                          \
        if (act == 0) MDL_DB_K(MT_, NT_, 0, 8);                                                                              \
        else if (act == 1) MDL_DB_K(MT_, NT_, 1, (wide ? 8 : 4));                                                            \
        else MDL_DB_K(MT_, NT_, 2, (wide ? 8 : 4));                                                                          \
    } while (0)
    if (mt == 2) { if (nt == 2) MDL_DB(2, 2); else if (nt == 4) MDL_DB(2, 4); else MDL_DB(2, 5); }
    else if (mt == 4) { if (nt == 2) MDL_DB(4, 2); else if (nt == 4) MDL_DB(4, 4); else MDL_DB(4, 5); }
    else { if (nt == 2) MDL_DB(5, 2); else if (nt == 4) MDL_DB(5, 4); else MDL_DB(5, 5); }
#undef MDL_DB
#undef MDL_DB_K
    if (part) hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(mt * nt * 4), 16), dim3(256), 0, st, part, (int)grid, nt, mt * nt, M, K, dw, db);
    return check_launch("mdl_dense_bwd");
}
