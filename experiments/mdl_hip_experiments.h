/* mdl_hip_experiments.h — entry points that exist only in the EXPERIMENTS build of the library
 * (experiments/build.py: the product sources compiled with -DMDL_EXPERIMENTS=1 -> experiments/lib/libmdl_hip_exp.so).
 * Each is a measured-negative variant of a product kernel, kept buildable and parity-tested (experiments/test_experiments.py)
 * so that the numbers in DESIGN.md section 4 can be reproduced; none of them is part of libmdl_hip.so or its header. */
#ifndef MDL_HIP_EXPERIMENTS_H
#define MDL_HIP_EXPERIMENTS_H
#include "../include/mdl_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Saved-gate variant of the pair above (dtype MDL_BF16, C in {32, 64}, G = 50, edge features in CSR order; row bytes 0 =
 * unsupported).  The training forward also writes, per edge and channel, the two factors the backward needs
 *     A = d m / d pre_f = sigmoid'(pre_f) softplus(pre_s),   B = d m / d pre_s = sigmoid(pre_f) sigmoid(pre_s)
 * as one packed bf16 pair: gate [E, C, 2] = mdl_cgconv_gate_row_bytes(C, G, dtype) (= 4C) bytes per edge, caller-owned.
 * mdl_cgconv_bwd_saved then produces the SAME r_tgt / r_src / dwe / db as mdl_cgconv_bwd from grad_out, the indices,
 * the edge features and `gate` alone — no x, no weights, no recompute of the gate: it trades 8C bytes of HBM traffic per
 * edge and layer for 24 of the 46 MFMAs, every transcendental and the x gathers of the recomputing pass. */
size_t mdl_cgconv_gate_row_bytes(int C, int G, int dtype);

/* W-split variant of the pair (dtype MDL_BF16, C in {32, 64}, G = 50, edge features in CSR order): the node parts of the two
 * Linear(2C+G, C) of PyG CGConv (cgcnn.py:80-83) leave the edge pass —
 *     z W^T = e W_e^T + P_t[i] + P_s[j],   P_t = x [W_f,tgt ; W_s,tgt]^T,   P_s = x [W_f,src ; W_s,src]^T   ([N, 2Cp] each)
 * — so that per edge only the K = 64 edge-feature product remains (4C G instead of 4C(2C+G) FLOP per edge).
 * mdl_cgconv_pack_weights_split fills wpack_e (mdl_cgconv_wsplit_bytes(.., 0) bytes: edge part + bias column) and wproj
 * (mdl_cgconv_wsplit_bytes(.., 1) bytes: two [2Cp, Cp] matrices, target then source, the `w` operand of mdl_linear_act with
 * M = 2Cp, K = C), both scaled like mdl_cgconv_pack_weights; the caller forms P_t / P_s with two mdl_linear_act launches.
 * mdl_cgconv_bwd_p produces the same r_tgt / r_src / dwe / db as mdl_cgconv_bwd: r_tgt = dL/dP_t and r_src = dL/dP_s are what
 * mdl_cgconv_bwd_node turns into dx and the node-weight gradients, unchanged. */
size_t mdl_cgconv_wsplit_bytes(int C, int G, int dtype, int which);
int mdl_cgconv_pack_weights_split(const float* w_f, const float* b_f, const float* w_s, const float* b_s, int C, int G,
                                  void* wpack_e, void* wproj, float* bpack, int dtype, mdlStream_t stream);
int mdl_cgconv_fwd_p(const void* x, const void* p_tgt, const void* p_src, const void* edge_attr, const int32_t* rowptr,
                     const int32_t* src, const int32_t* tgt, const void* wpack_e, const float* bpack, void* out,
                     int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream);
int mdl_cgconv_bwd_p(const void* p_tgt, const void* p_src, const void* edge_attr, const int32_t* rowptr, const int32_t* src,
                     const int32_t* tgt, const void* wpack_e, const float* bpack, const void* grad_out, void* r_tgt,
                     float* r_src, float* dwe, float* db, int64_t N, int64_t E, int C, int G, int aggr, int dtype,
                     void* workspace, size_t ws_bytes, mdlStream_t stream);
int mdl_cgconv_fwd_save(const void* x, const void* edge_attr, const int32_t* rowptr, const int32_t* src,
                        const int32_t* tgt, const void* wpack, const float* bpack, void* out, void* gate,
                        int64_t N, int64_t E, int C, int G, int aggr, int dtype, mdlStream_t stream);
int mdl_cgconv_bwd_saved(const void* edge_attr, const int32_t* rowptr, const int32_t* src, const int32_t* tgt,
                         const void* gate, const void* grad_out, void* r_tgt, float* r_src, float* dwe, float* db,
                         int64_t N, int64_t E, int C, int G, int aggr, int dtype, void* workspace, size_t ws_bytes,
                         mdlStream_t stream);

/* Two chained dense layers in one pass: h[N, M1] = act1(x[N, K] w1[M1, K]^T + b1), y[N, M2] = act2(h w2[M2, M1]^T + b2); both
 * results are written (the backward of the pair needs h), the rows of h reach the second product through LDS.  The filter
 * network of CFConv — Linear(num_gaussians, F) -> ShiftedSoftplus -> Linear(F, F) over the edges (matdeeplearn/models/
 * schnet.py:81 via torch_geometric.nn.models.schnet.InteractionBlock.mlp).  bf16; even K <= 64, even M1 <= 160, M2 <= 160. */
int mdl_mlp2(const void* x, const void* w1, const void* b1, int act1, const void* w2, const void* b2, int act2, void* h,
             void* y, int64_t N, int K, int M1, int M2, int dtype, mdlStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MDL_HIP_EXPERIMENTS_H */
