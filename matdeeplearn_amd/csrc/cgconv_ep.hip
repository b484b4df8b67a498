// cgconv_ep.hip — translation unit of the edge-per-lane CGConv backward edge pass (cgconv_ep.inc).
// It shares the tile machinery of cgconv.hip (CgParams, EWords, NodeRange, one-hot tables, ...) by including that file
// with MDL_CG_EP_TU defined: only mdl::ep::bwd_kernel and mdl::ep::launch are emitted here.  Built with
// -mllvm -amdgpu-mfma-vgpr-form=1 (matdeeplearn_amd/_build.py), see the comment at the include in cgconv.hip.
#define MDL_CG_EP_TU 1
#include "cgconv.hip"
