// cgconv_ep.hip — translation unit of the edge-per-lane CGConv backward edge pass (cgconv_ep2.inc: mdl::ep::bwd2_kernel and
// mdl::ep::launch2).  It shares the tile machinery of the per-wave kernels (CgParams, EWords, NodeRange, one-hot tables, ...) through
// cgconv_tiles.inc and is built with -mllvm -amdgpu-mfma-vgpr-form=1 (matdeeplearn_amd/_build.py; see the comment in cgconv.hip).
#define MDL_CG_EP_TU 1
#include "cgconv_tiles.inc"

namespace mdl {
#include "cgconv_ep_common.inc"
#if MDL_EXPERIMENTS
#include "../../experiments/csrc/cgconv_ep.inc"   // phases one after the other: measured slower than the per-wave kernel
#endif
#include "cgconv_ep2.inc"
namespace ep {
static bool cg_env_ep2_static() {          // experiments build, MDL_EP2_STATIC=1: kernel 2 without the dynamic tail (A/B)
#if MDL_EXPERIMENTS
    static const bool v = [] { const char* s = getenv("MDL_EP2_STATIC"); return s && atoi(s) != 0; }();
    return v;
#else
    return true;
#endif
}
int launch2(CgParams& p, hipStream_t st, int wgs, const char* name) {
    typedef Cfg2<64> F;
    const int64_t eg = std::min<int64_t>(wgs > 0 ? wgs : 256, std::max<int64_t>(1, cdiv(p.E, 32 * F::NA * 2)));
    auto kf = bwd2_kernel<64>;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), F::LDS);
    if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, F::LDS, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    // optional caller workspace: the chunk counter of the dynamic tail, zeroed on the stream
    if (MDL_EP2_TAIL == 0 || cg_env_ep2_static()) p.ctr = nullptr;      // (the dynamic tail: experiments build with -DMDL_EP2_TAIL=25)
    if (p.ctr && hipMemsetAsync(p.ctr, 0, 64, st) != hipSuccess) p.ctr = nullptr;
    hipLaunchKernelGGL(kf, dim3((unsigned)eg), dim3(F::NT), F::LDS, st, p);
    return check_launch(name);
}
#if MDL_EXPERIMENTS
int launch(CgParams& p, hipStream_t st, int wgs, const char* name) {
    typedef Cfg<64> F;
    // one workgroup per CU; small problems: at least two rounds of tiles per workgroup
    const int64_t eg = std::min<int64_t>(wgs > 0 ? wgs : 256, std::max<int64_t>(1, cdiv(p.E, 32 * F::NW * 2)));
    auto kf = bwd_kernel<64>;
    hipError_t e = set_max_dynamic_lds(reinterpret_cast<const void*>(kf), F::LDS);
    if (e != hipSuccess) { set_error("%s: LDS attribute (%d B): %s", name, F::LDS, hipGetErrorString(e)); return MDL_E_LAUNCH; }
    hipLaunchKernelGGL(kf, dim3((unsigned)eg), dim3(F::NT), F::LDS, st, p);
    return check_launch(name);
}
#endif
}  // namespace ep
}  // namespace mdl
