/*
 * dgcn.h -- C ABI of libdgcn.so, the MI355X (gfx950) message-passing library.
 *
 * Every entry point replaces one hot-path call site of lightaime/deep_gcns_torch
 * (paths below are relative to the reference root).  The library is torch-free:
 * callers hand over raw device pointers, sizes and a hipStream_t.  Rules that hold
 * for EVERY function:
 *
 *   - the caller owns all buffers (inputs, outputs, saved-for-backward, workspace);
 *     the library never allocates, frees or keeps a pointer after returning;
 *   - work is enqueued asynchronously on `stream`; nothing synchronises;
 *   - no mutable global state: safe from one host thread per GPU (nn.DataParallel);
 *   - return 0 on success, a negative DGCN_E_* for a rejected argument (nothing was
 *     launched), a positive hipError_t if a launch failed.  dgcn_strerror() names it.
 *
 * Index tensors are int32 on the device (E < 2^31); node features are fp32 rows.
 */
#ifndef DGCN_H
#define DGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGCN_VERSION 100 /* 0.1.0 */

/* error codes (negative = argument error, nothing launched) */
#define DGCN_OK 0
#define DGCN_E_NULL (-1)      /* required pointer is NULL */
#define DGCN_E_SHAPE (-2)     /* size/shape out of the supported range */
#define DGCN_E_ALIGN (-3)     /* pointer or stride not aligned as required */
#define DGCN_E_MODE (-4)      /* unknown mode / flag combination */
#define DGCN_E_WORKSPACE (-5) /* workspace too small */

/* aggregation modes: gcn_lib/sparse/torch_message.py:44-85 */
#define DGCN_AGGR_ADD 0     /* :46-47 -> PyG base scatter(reduce='add')  */
#define DGCN_AGGR_MEAN 1    /* :46-47 -> scatter(reduce='mean')          */
#define DGCN_AGGR_MAX 2     /* :46-47 -> scatter(reduce='max'), empty->0 */
#define DGCN_AGGR_SOFTMAX 3 /* :49-58  softmax / softmax_sg / softmax_sum (deg scaling done by caller) */
#define DGCN_AGGR_POWER 4   /* :68-74  power / power_sum                 */

/* message flags: how the per-edge message m_e is formed from the gathered row z_e */
#define DGCN_MSG_IDENTITY 0 /* m_e = z_e            (utils/pyg_util.py:26 scatter_ of an edge tensor) */
#define DGCN_MSG_RELU_EPS 1 /* m_e = relu(z_e)+eps  (gcn_lib/sparse/torch_vertex.py:78-85 GENConv.message) */

/* extra behaviour bits for dgcn_gen_aggr_{fwd,bwd}_f32 */
#define DGCN_FLAG_LEARN_T 1 /* softmax weights are differentiated (torch_message.py:51-52) */
#define DGCN_FLAG_LEARN_P 2 /* power exponent is differentiated  (torch_message.py:33-34) */
#define DGCN_FLAG_SHIFT_FLAG_IS_RANGE 8 /* backward: shift_ok points at the forward's range_flag (0 = safe), not at an "ok" flag */
#define DGCN_FLAG_ADD_ROOT 4 /* forward: out_i += x_i, the h = x + m of GENConv.forward (torch_vertex.py:74) fused in */
#define DGCN_FLAG_STATIC_ITEMS 32 /* per-edge-encoder entry points: deal the work items to the waves by index instead of
                                  * handing them out from the device-side counters.  The outputs and grad_x are the same
                                  * bits either way (an item's result does not depend on who computes it); the dW | db
                                  * partial sums are grouped per workgroup, so with the dynamic schedule (default) their
                                  * LAST BITS vary from run to run -- set this flag for bit-reproducible weight
                                  * gradients at 1.3 - 1.6x the launch time on power-law graphs. */
#define DGCN_FLAG_EA_IS_Z 16 /* backward: the rows of edge_attr are the pre-activations z_e themselves (saved by
                                dgcn_gen_aggr_egemm_fwd_f32), x is not gathered (may be NULL).  With DGCN_AGGR_MAX the
                                rows are not read either (edge_attr may be NULL, no z_save needed): that forward marks
                                the channels whose best neighbour has z <= 0 with arg-max id -1 */

/*
 * Graph structure, built once per distinct edge_index and reused by every layer
 * (SURVEY.md a16).  "CSR" is keyed by DESTINATION (edge_index[1]); "CSC" by SOURCE
 * (edge_index[0]).  Both are stable sorts of the original edge list, so edges of a
 * row keep their original relative order.  All arrays live on the device.
 */
typedef struct dgcn_graph {
  int32_t n_dst;         /* number of destination rows (dim_size of the scatter)          */
  int32_t n_src;         /* number of source rows (x.size(0)); == n_dst for square graphs */
  int32_t n_edges;       /* E                                                            */
  int32_t reserved;
  const int32_t* rowptr; /* [n_dst+1]  CSR offsets by destination                         */
  const int32_t* col;    /* [E]        source node of CSR position e                      */
  const int32_t* eperm;  /* [E] or NULL: original edge id of CSR position e (NULL=identity)*/
  const int32_t* t_rowptr; /* [n_src+1] CSC offsets by source        (backward only)      */
  const int32_t* t_col;    /* [E]       destination node of CSC position e                */
  const int32_t* t_eperm;  /* [E]       original edge id of CSC position e                */
  /* Optional work list that splits high-degree rows into chunks (deterministic hub
   * handling).  n_work == 0 means "one work item per row".  Layout contract (what
   * dgcn_graph_work_list produces and the merge kernels rely on): the pieces of a split
   * row are CONSECUTIVE items with CONSECUTIVE slot ids, in edge order, all pieces but
   * the last of equal length.                                                            */
  int32_t n_work;
  int32_t n_slots;         /* number of partial-result slots used by split rows            */
  const int32_t* work_row; /* [n_work] row of each item                                    */
  const int32_t* work_beg; /* [n_work] first CSR position                                  */
  const int32_t* work_end; /* [n_work] one past last CSR position                          */
  const int32_t* work_slot;/* [n_work] -1: item covers the whole row; >=0: partial slot id */
  /* same for the transposed walk */
  int32_t t_n_work;
  int32_t t_n_slots;
  const int32_t* t_work_row;
  const int32_t* t_work_beg;
  const int32_t* t_work_end;
  const int32_t* t_work_slot;
  /* index (into the work list) of the FIRST item of every split row: one merge wave per entry */
  int32_t n_split;
  int32_t t_n_split;
  const int32_t* split_item;   /* [n_split]   */
  const int32_t* t_split_item; /* [t_n_split] */
} dgcn_graph;

int dgcn_version(void);
const char* dgcn_strerror(int rc);

/* Device smoke test: y[i] = a*x[i] + y[i] on `stream`.  Used by the loader to prove the
 * library shares the caller's HIP runtime (same stream handles, same allocations). */
int dgcn_selftest_axpy_f32(float a, const float* x, float* y, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * Sparse generalized aggregation.
 * Replaces  GENConv.propagate -> message -> GenMessagePassing.aggregate
 *           (gcn_lib/sparse/torch_vertex.py:68,78-85; gcn_lib/sparse/torch_message.py:44-85)
 * and the torch_scatter kernels underneath (scatter / scatter_softmax / scatter_max).
 *
 *   z_e   = x[col[e]] (+ edge_attr[eperm[e]])           row gather, C channels
 *   m_e   = relu(z_e) + eps        (DGCN_MSG_RELU_EPS)  |  z_e  (DGCN_MSG_IDENTITY)
 *   out_i = AGGR_{e in row i} m_e                       per destination, per channel
 *
 *   x          [n_src, C] fp32, row stride x_stride floats
 *   edge_attr  [E, C] fp32 contiguous in ORIGINAL edge order, or NULL
 *   t_dev/p_dev  optional device scalars overriding t / p (learnable parameters: no host sync)
 *   out        [n_dst, C] contiguous
 *   aux1       [n_dst, C] or NULL.  SOFTMAX: logsumexp L_i = M_i + log D_i of t*m_e.
 *              POWER: the pre-clamp mean q_i.  MAX: int32 original edge id of the arg-max
 *              (-1 for an empty row, and with DGCN_MSG_RELU_EPS for a channel whose best message is the relu
 *              floor m = eps: no neighbour has z > 0, no edge receives a gradient).  ADD/MEAN: unused.
 *   aux2       [n_dst, C] or NULL.  SOFTMAX+LEARN_T: sum_e w_e m_e^2.
 *              POWER+LEARN_P: sum_e u_e^p ln u_e.  Otherwise unused.
 *   range_flag optional device int32, zeroed by the caller: SOFTMAX sets it to 1 when some |L_i| >= 80, i.e. when
 *              the single-gather backward with shift 0 would leave the fp32 range (decided on the device).
 *   workspace  >= dgcn_gen_aggr_fwd_workspace_bytes(g, C) bytes: the partial-state slots of split rows + 1 KiB of
 *              scheduling state for the per-edge-encoder entry points (dgcn_gen_aggr_enc_*: a work-item counter the
 *              entry point zeroes itself with a memset node on `stream`).  This plain entry point needs the slots only:
 *              on a graph without split rows NULL / 0 is accepted.
 */
size_t dgcn_gen_aggr_fwd_workspace_bytes(const dgcn_graph* g, int32_t channels);

int dgcn_gen_aggr_fwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                          const float* edge_attr, int32_t channels, int32_t mode,
                          int32_t msg, int32_t flags, float t, float p, float eps,
                          const float* t_dev, const float* p_dev, float* out, void* aux1,
                          float* aux2, int32_t* range_flag, void* workspace, size_t workspace_bytes,
                          void* stream);

/*
 * Backward of the above w.r.t. x (and edge_attr), one deterministic walk over the CSC:
 *   grad_x[s] = sum_{e: src(e)=s} dL/dz_e ,  grad_edge_attr[orig(e)] = dL/dz_e
 * with dL/dz_e = 1[z_e>0] * dL/dm_e (RELU_EPS) and dL/dm_e per SURVEY.md Appendix A.
 *
 *   gcoef   [n_dst, C] per-destination coefficient prepared by the caller:
 *             ADD: g_i        MEAN: g_i/max(deg_i,1)      MAX: g_i
 *             SOFTMAX: g_i    POWER: g_i * r_i^(1/p-1) / max(deg_i,1) * 1[lo<=q_i<=hi]
 *   aux1    as written by the forward (SOFTMAX: L_i, MAX: arg-max edge ids)
 *   out     forward output (only read for SOFTMAX with DGCN_FLAG_LEARN_T)
 *   gshift, kshift, shift_ok   optional single-gather form for SOFTMAX (see dgcn_softmax_bwd_prep_f32), or NULL
 *   groot   [n_src, C] contiguous or NULL: added to grad_x (the backward of DGCN_FLAG_ADD_ROOT: pass the raw
 *           upstream gradient)
 *   grad_x  [n_src, C] contiguous, fully overwritten
 *   grad_edge_attr [E, C] in original edge order or NULL
 */
size_t dgcn_gen_aggr_bwd_workspace_bytes(const dgcn_graph* g, int32_t channels);

int dgcn_gen_aggr_bwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride,
                          const float* edge_attr, int32_t channels, int32_t mode,
                          int32_t msg, int32_t flags, float t, float p, float eps,
                          const float* t_dev, const float* p_dev, const float* gcoef,
                          const void* aux1, const float* out, const float* gshift,
                          const float* kshift, const int32_t* shift_ok, const float* groot,
                          float* grad_x, float* grad_edge_attr, void* workspace, size_t workspace_bytes,
                          void* stream);

/*
 * Max aggregation without edge rows, the backward in two launches that move a fraction of the bytes
 * (scatter_('max') backward of torch_scatter under GenMessagePassing.aggregate, gcn_lib/sparse/torch_message.py:83;
 * MRConv / EdgConv, gcn_lib/sparse/torch_vertex.py:102,111):
 *   1. per-edge arg-max bit masks: bit c of mask[p] (p = CSR position, ceil(C/32) rounded up to 1|2|4|8 words) says
 *      whether edge p is the arg-max of its destination row in channel c -- one wave per destination row locates the
 *      rows' winners (work ~ rows x channels, not edges x channels);
 *   2. the CSC walk of dgcn_gen_aggr_bwd_f32 reading 4 mask bytes per (edge, 32 channels) -- fetched with the column
 *      ids one item ahead and parked in LDS -- instead of gathering the 4 C bytes of the destination's arg-max row, and
 *      fetching g only where a bit is set.
 *   t_cpos  [E] CSR position of every CSC position (the CSC walk's index into the masks)
 *   argmax  [n_dst, C] as written by the forward (aux1), gcoef = the upstream gradient, groot / grad_x as above
 *   mask    dgcn_gen_aggr_max_mask_bytes(E, C) bytes of scratch, 16-byte aligned; workspace as dgcn_gen_aggr_bwd_f32
 * Bit-identical to dgcn_gen_aggr_bwd_f32(mode = DGCN_AGGR_MAX).  channels <= 256.
 */
size_t dgcn_gen_aggr_max_mask_bytes(int32_t n_edges, int32_t channels);

int dgcn_gen_aggr_max_bwd_f32(const dgcn_graph* g, const int32_t* t_cpos, const float* x, int64_t x_stride,
                              int32_t channels, int32_t msg, int32_t flags, float eps, const float* gcoef,
                              const int32_t* argmax, const float* groot, float* grad_x, void* mask, size_t mask_bytes,
                              void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same aggregation with a NARROW edge encoder evaluated per edge: every edge recomputes its row
 *   e_e = enc_weight f_e + enc_bias       (n_feat = 8 raw features, 32 bytes per edge)
 * and no (E, C) or (E, hidden) edge array exists in either direction.  Call site: the reference's models with edge
 * features apply TWO Linear maps in a row to the 8 raw edge features -- the model-level
 * edge_encoder = Linear(8 -> hidden) (examples/ogb_eff/ogbn_proteins/model_rev.py:53,98; ogbn_proteins/model.py:74-78)
 * and every GENConv's edge_encoder = Linear(hidden -> C) (gcn_lib/sparse/torch_vertex.py:56-66) -- with nothing
 * between them: their composition W_l We (C x 8), W_l be + b_l is this entry point's (enc_weight, enc_bias); the host
 * side forms it with autograd ops (deep_gcns_torch_amd.blocks.ComposedEdgeEmbedding), so the gradients of both Linear
 * layers follow from the (C, 9) result of the backward by two tiny matrix products.
 *   enc_feat    [E, n_feat] fp32 contiguous, ORIGINAL edge order; n_feat must be 8
 *   enc_weight  [channels, n_feat], enc_bias [channels] or NULL
 *   channels % 4 == 0, channels <= 256, 16-byte aligned pointers; anything else returns DGCN_E_SHAPE / _ALIGN.
 * Backward: grad_x as above; the encoder gradients come out as per-workgroup partials
 *   enc_grad_partials [dgcn_gen_aggr_enc_bwd_num_partials(g, channels)][channels][n_feat + 1]
 * whose sum over the first axis is (d enc_weight | d enc_bias); every block is fully written.
 */
int dgcn_gen_aggr_enc_fwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride, const float* enc_feat,
                              const float* enc_weight, const float* enc_bias, int32_t n_feat, int32_t channels,
                              int32_t mode, int32_t msg, int32_t flags, float t, float p, float eps,
                              const float* t_dev, const float* p_dev, float* out, void* aux1, float* aux2,
                              int32_t* range_flag, void* workspace, size_t workspace_bytes, void* stream);

int32_t dgcn_gen_aggr_enc_bwd_num_partials(const dgcn_graph* g, int32_t channels);

int dgcn_gen_aggr_enc_bwd_f32(const dgcn_graph* g, const float* x, int64_t x_stride, const float* enc_feat,
                              const float* enc_weight, const float* enc_bias, int32_t n_feat, int32_t channels,
                              int32_t mode, int32_t msg, int32_t flags, float t, float p, float eps,
                              const float* t_dev, const float* p_dev, const float* gcoef, const void* aux1,
                              const float* out, const float* gshift, const float* kshift,
                              const int32_t* shift_ok, const float* groot, float* grad_x,
                              float* enc_grad_partials, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Weight gradient of the per-edge encoder (dgcn_gen_aggr_enc_fwd_f32) under DGCN_AGGR_MAX from the arg-max winners:
 * the forward's aux1 holds the ORIGINAL edge id of the winner of every (destination row, channel), -1 where no
 * neighbour passed the relu of GENConv.message (gcn_lib/sparse/torch_vertex.py:78-85) or the row is empty, so
 *   d enc_weight[c][f] = sum_r gcoef[r][c] * enc_feat[argmax[r][c]][f],   d enc_bias[c] = sum_{r: argmax >= 0} gcoef[r][c]
 * (what autograd builds for edge_emb = edge_encoder(edge_attr), torch_vertex.py:62-66, under scatter(reduce='max'),
 * gcn_lib/sparse/torch_message.py:46-47) costs n_dst * channels gathers of 32 bytes instead of a pass over the edges.
 * grad_x then comes from dgcn_gen_aggr_bwd_f32(edge_attr = NULL, flags | DGCN_FLAG_EA_IS_Z) (the ids carry the relu mask).
 *   enc_grad_partials [dgcn_enc_max_bwd_num_partials(n_dst)][channels][n_feat + 1], every block fully written; the sum
 *   over the first axis is (d enc_weight | d enc_bias), fixed summation order.  n_feat == 8, channels <= 256.
 *   n_edges = rows of enc_feat: an id outside [0, n_edges) counts as -1 (ids are never used as addresses unchecked;
 *   dgcn_egemm_max_bwd_f32 treats its ids the same way).
 */
int32_t dgcn_enc_max_bwd_num_partials(int32_t n_dst);
int dgcn_enc_max_bwd_weight_f32(const float* gcoef, const int32_t* argmax, int32_t n_dst, int32_t n_edges,
                                const float* enc_feat, int32_t n_feat, int32_t channels, float* enc_grad_partials,
                                void* stream);

/*
 * Composition of the model-level edge encoder Linear(F -> hidden) (examples/ogb_eff/ogbn_proteins/model_rev.py:53,98;
 * ogbn_proteins/model.py:90,107) with a GENConv's own edge_encoder = Linear(hidden -> C)
 * (gcn_lib/sparse/torch_vertex.py:56-66) into the Linear(F -> C) that dgcn_gen_aggr_enc_{fwd,bwd}_f32 evaluate per edge:
 *   out_w[c][f] = sum_h layer_w[c][h] enc_w[h][f],   out_b[c] = sum_h layer_w[c][h] enc_b[h] + layer_b[c]
 * and its backward (what autograd builds for the two Linear calls in a row), one launch each:
 *   grad_layer_w = grad_w enc_w^T + grad_b enc_b^T,  grad_enc_w = layer_w^T grad_w,  grad_enc_b = layer_w^T grad_b
 * (grad_layer_b = grad_b: the caller's).  Row-major contiguous fp32; n_feat <= 16; biases may be NULL (with out_b /
 * grad_b NULL when both are).  Fixed summation order: bit-reproducible.
 */
int dgcn_enc_compose_fwd_f32(const float* layer_w, const float* layer_b, const float* enc_w, const float* enc_b,
                             int32_t channels, int32_t hidden, int32_t n_feat, float* out_w, float* out_b, void* stream);
int dgcn_enc_compose_bwd_f32(const float* layer_w, const float* enc_w, const float* enc_b, const float* grad_w,
                             const float* grad_b, int32_t channels, int32_t hidden, int32_t n_feat, float* grad_layer_w,
                             float* grad_enc_w, float* grad_enc_b, void* stream);

/*
 * The edge encoder of GENConv on WIDE edge features as the reference's models use it: the model computes ONE
 * (E, hidden) edge embedding and every GENConv owns edge_encoder = Linear(edge_feat_dim = hidden -> C)
 * (gcn_lib/sparse/torch_vertex.py:56-66; examples/ogb_eff/ogbn_proteins/model_rev.py:45-55,98-107;
 * eff_gcn_modules/rev/rev_layer.py:53-75; examples/ogb/ogbn_proteins/model.py:74-107; ogbg_ppa/model.py:60).
 * dgcn_gen_aggr_egemm_fwd_f32 runs that E x n_feat x C GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: an
 * exact fp32 fma chain), adds the gathered x[src] and folds the tile straight into the aggregation: the (E, C)
 * edge embedding of the reference is never written.
 *   erow        [E] int32: destination row of every CSR position (= sorted edge_index[1])
 *   edge_feat   [E, n_feat] fp32, ORIGINAL edge order, row stride feat_stride floats (a torch.chunk view of the
 *               model-level embedding is consumed in place), 16-byte aligned rows
 *   enc_weight  [channels, n_feat] contiguous (nn.Linear.weight), enc_bias [channels] or NULL
 *   z_save      [E, channels] or NULL: receives z_e = x[src] + W f_e + b in ORIGINAL edge order; the backward is
 *               then dgcn_gen_aggr_bwd_f32(edge_attr = z_save, flags | DGCN_FLAG_EA_IS_Z), whose grad_edge_attr is
 *               dL/dz_e = the gradient of the (never materialised) edge embedding.  DGCN_AGGR_MAX: pass NULL here and
 *               NULL as the backward's edge_attr (aux1 is enough)
 *   workspace   dgcn_gen_aggr_egemm_fwd_workspace_bytes(E, n_src, n_feat, channels) bytes, 16-byte aligned (partial
 *               row states of the work items + the gather source x + bias)
 * Supported (dgcn_gen_aggr_egemm_supported): channels % 4 == 0, channels <= 128, n_feat % 16 == 0, n_feat <= 256,
 * weight tile + two wave tiles within the 160 KiB LDS; E >= 1.  Everything else as dgcn_gen_aggr_fwd_f32.
 */
int32_t dgcn_gen_aggr_egemm_supported(int32_t n_feat, int32_t channels);
size_t dgcn_gen_aggr_egemm_fwd_workspace_bytes(int32_t n_edges, int32_t n_src, int32_t n_feat, int32_t channels);
int dgcn_gen_aggr_egemm_fwd_f32(const dgcn_graph* g, const int32_t* erow, const float* x, int64_t x_stride,
                                const float* edge_feat, int64_t feat_stride, const float* enc_weight,
                                const float* enc_bias, int32_t n_feat, int32_t channels, int32_t mode, int32_t msg,
                                int32_t flags, float t, float p, float eps, const float* t_dev, const float* p_dev,
                                float* out, void* aux1, float* aux2, int32_t* range_flag, float* z_save,
                                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of the fused edge encoder under DGCN_AGGR_MAX without the (E, channels) gradient dz (what autograd builds
 * for edge_emb = edge_encoder(edge_feat) in GENConv.forward, gcn_lib/sparse/torch_vertex.py:62-66, aggregated by
 * scatter(reduce='max'), gcn_lib/sparse/torch_message.py:46-47; called per layer from the reversible backward,
 * eff_gcn_modules/rev/gcn_revop.py:121-133).  dz has one non-zero per (destination row, channel) -- g[r][c] at the
 * arg-max edge -- so the kernel walks the winners instead of the edges:
 *   grad_feat[e][:]       += sum over the channels c that edge e wins:  gcoef[r][c] * enc_weight[c][:]
 *   d enc_weight[c][:]     = sum over rows r:                           gcoef[r][c] * edge_feat[argmax[r][c]][:]
 *   gcoef   [n_dst, channels] gradient of the aggregated rows (for MAX: grad_out itself)
 *   argmax  [n_dst, channels] int32 = aux1 of dgcn_gen_aggr_egemm_fwd_f32(DGCN_AGGR_MAX): ORIGINAL edge id of the
 *           winner, -1 where no neighbour passed the relu (those channels pass no gradient)
 *   edge_feat / feat_stride, enc_weight: as in the forward (edge_feat may be NULL when grad_w_partials is NULL)
 *   grad_feat [n_edges, n_feat] row stride grad_feat_stride, ACCUMULATED in place (rows of edges that win nothing are
 *           not touched: pass a zeroed or running buffer), or NULL
 *   grad_w_partials [dgcn_egemm_max_bwd_num_partials(n_dst)][channels][n_feat], every block fully written, their sum
 *           over the first axis is d enc_weight; or NULL.  No atomics anywhere: both results are bit-reproducible.
 * grad_x and d enc_bias come from dgcn_gen_aggr_bwd_f32(edge_attr = NULL, grad_edge_attr = NULL, DGCN_FLAG_EA_IS_Z) as
 * before.  channels <= 128, n_feat % 4 == 0, n_feat <= 256, 16-byte aligned rows.
 */
int32_t dgcn_egemm_max_bwd_num_partials(int32_t n_dst);
int dgcn_egemm_max_bwd_f32(const float* gcoef, const int32_t* argmax, int32_t n_dst, int32_t n_edges,
                           const float* edge_feat, int64_t feat_stride, const float* enc_weight, int32_t n_feat,
                           int32_t channels, float* grad_feat, int64_t grad_feat_stride, float* grad_w_partials,
                           void* stream);

/* Per-destination coefficient of the POWER / MEAN backward in one pass (the `gcoef` of dgcn_gen_aggr_bwd_f32):
 *   out[i,c] = grad_out[i,c] * r^(1/p - 1) * [1e-7 <= q <= 10] / max(deg_i, 1),  r = clamp(q, 1e-7, 10)
 * q = aux1 of the POWER forward (torch_message.py:68-74); q == NULL gives the MEAN form grad_out / max(deg_i, 1).
 * p_dev (device scalar) overrides p when given. */
int dgcn_power_bwd_prep_f32(const dgcn_graph* g, const float* grad_out, const float* q, const float* p_dev, float p,
                            float* out, int32_t channels, void* stream);

/* Node-wise prologue of the single-gather softmax backward:
 *   out[i,c] = g[i,c] * exp(kshift[c] - L[i,c])          (channels % 4 == 0)
 * With it, dL/dm_e = g_i exp(t m_e - L_i) = out_i * exp(t m_e - kshift_c): the edge walk gathers ONE row
 * per edge instead of two.  The caller passes (gshift=out, kshift, shift_ok) to dgcn_gen_aggr_bwd_f32;
 * shift_ok is a DEVICE flag (1 when every |L_i - kshift_c| keeps both factors inside the fp32 range; with
 * kshift = 0 it is the negation of the forward's range_flag, so it is decided without a host sync or an
 * extra pass over L); when it is 0, or the three pointers are NULL, the kernel uses the two-gather form. */
int dgcn_softmax_bwd_prep_f32(const float* g, const float* L, const float* kshift, float* out,
                              int64_t n_rows, int32_t channels, void* stream);

/* Merge of two partial softmax aggregations of the same destination rows over DISJOINT edge sets (the local-source and
 * the remote-source edges of a destination partition, deep_gcns_torch_amd/dist.py SplitGraph): from
 * (out_a, lse_a) and (out_b, lse_b) as dgcn_gen_aggr_fwd_f32 writes them (out + aux1 = log-sum-exp; 0 for rows without
 * edges -- the two CSR row pointers say which rows those are)
 *   out = sigmoid(lse_a - lse_b) out_a + sigmoid(lse_b - lse_a) out_b,   lse = logaddexp(lse_a, lse_b)
 * exactly the aggregation over the union (gcn_lib/sparse/torch_message.py:54-58).  All arrays [n_rows, channels] fp32
 * contiguous and 16-byte aligned, channels % 4 == 0; out / lse may alias out_a / lse_a. */
int dgcn_softmax_state_merge_f32(const float* out_a, const float* lse_a, const int32_t* rowptr_a, const float* out_b,
                                 const float* lse_b, const int32_t* rowptr_b, float* out, float* lse, int64_t n_rows,
                                 int32_t channels, void* stream);

/* ------------------------------------------------------------------------------------
 * Device-side graph structure.  The reference keeps a COO edge_index and re-partitions it on
 * the host every epoch (utils/data_util.py:43-61 random_partition_graph / generate_sub_graphs: scipy CSR slicing;
 * examples/ogb/ogbn_proteins/dataset.py:87-151: a python dict lookup per edge for the edge ids).
 *
 * dgcn_graph_csr_build: stable order of the E edges by `key` (edge_index[1] for the forward CSR, edge_index[0] for
 *   the backward CSC), `other` = the opposite endpoint:
 *     rowptr [n_rows+1], col [E] = other endpoint of CSR position p, eperm [E] = original edge id of position p,
 *     erow [E] or NULL = key of position p (the sorted keys)
 *   status [8] int32 (device): [0] != 0 an id was out of range (such edges are dropped from the counts: treat the
 *   structure as invalid), [1] maximum row length, [2] != 0 the keys were NOT already non-decreasing, and -- for
 *   hub_chunk > 0 -- the sizes of the hub work list: [3] work items, [4] partial-result slots, [5] split rows.
 *   One host read of `status` replaces the min / max / is-sorted / bincount synchronisations of a host-driven build
 *   AND sizes the work-list arrays.
 * dgcn_graph_work_list: the work list of dgcn_graph (rows longer than 2 * hub_chunk edges cut into hub_chunk-edge items:
 *     work_row / work_beg / work_end / work_slot [status[3]], split_item [status[5]] = first item of every split row);
 *     slots are numbered in item order, work_slot = -1 for an item that covers its whole row.  No host read.
 * dgcn_graph_coalesce: the sorted (by row, then col), duplicate-free pairs of an edge list -- torch_sparse.coalesce on
 *     indices -- or, with both_directions != 0, of the list and its reverse: PyG to_undirected
 *     (examples/ogb/ogbn_arxiv/main.py:72-75).  out [2][out_stride] int64 (rows, then cols), out_stride >= number of keys
 *     (E or 2 E); counts [2] int64 (device) = {#pairs written, != 0 when an id was outside [0, n_nodes)}.
 * dgcn_subgraph_extract: the sub-graph induced by the nodes with parts[i] == cluster:
 *     node_ids [<= n_nodes] ascending, (sub_src, sub_dst) [<= E] relabelled to positions in node_ids, original edge
 *     order, edge_ids [<= E] = kept original edge ids (to slice edge_attr); counts [3] int64 (device) = {#nodes, #edges,
 *     != 0 when an edge endpoint was outside [0, n_nodes) -- such edges are never kept, nothing is read out of bounds}.
 * All int64 inputs are contiguous device arrays; workspaces from the *_workspace_bytes functions; asynchronous on
 * `stream`.
 */
size_t dgcn_graph_csr_workspace_bytes(int64_t n_edges, int32_t n_rows);
int dgcn_graph_csr_build(const int64_t* key, const int64_t* other, int64_t n_edges, int32_t n_rows, int32_t n_other,
                         int32_t hub_chunk, int32_t* rowptr, int32_t* col, int32_t* eperm, int32_t* erow,
                         int32_t* status, void* workspace, size_t workspace_bytes, void* stream);
size_t dgcn_graph_work_list_workspace_bytes(int32_t n_rows);
int dgcn_graph_work_list(const int32_t* rowptr, int32_t n_rows, int32_t hub_chunk, int32_t* work_row, int32_t* work_beg,
                         int32_t* work_end, int32_t* work_slot, int32_t* split_item, void* workspace,
                         size_t workspace_bytes, void* stream);
size_t dgcn_graph_coalesce_workspace_bytes(int64_t n_edges, int32_t n_nodes, int32_t both_directions);
int dgcn_graph_coalesce(const int64_t* src, const int64_t* dst, int64_t n_edges, int32_t n_nodes, int32_t both_directions,
                        int64_t* out, int64_t out_stride, int64_t* counts, void* workspace, size_t workspace_bytes,
                        void* stream);
size_t dgcn_subgraph_workspace_bytes(int64_t n_edges, int32_t n_nodes);
int dgcn_subgraph_extract(const int64_t* src, const int64_t* dst, int64_t n_edges, const int64_t* parts,
                          int32_t n_nodes, int64_t cluster, int64_t* node_ids, int64_t* sub_src, int64_t* sub_dst,
                          int64_t* edge_ids, int64_t* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Dense (B x C x N x 1) point-cloud path.
 * ------------------------------------------------------------------------------------ */

/* Fused pairwise distance + exact top-K + dilation.
 * Replaces pairwise_distance / dense_knn_matrix / DenseDilated.forward (deterministic branch)
 * (gcn_lib/dense/torch_edge.py:32-42, 45-58, 26-28) and knn_matrix (gcn_lib/sparse/torch_edge.py:66-91).
 *   x        (B, C, N) fp32 with element strides (sb, sc, sn)  -- channel slices / views are fine
 *   K        = k * dilation neighbours selected per point, self included, ascending distance
 *            D_ij = (|x_i|^2 + (-2 <x_i, x_j>)) + |x_j|^2 in fp32; equal distances ordered by index
 *   nn_out   [B, N, ceil(K/dilation)] int64: positions 0, d, 2d, ... of the sorted list
 *   ctr_out  same shape or NULL: the centre point id (edge_index[1])
 *   exclude_self != 0: the query point itself is never returned (torch_cluster.knn_graph(loop=False),
 *            gcn_lib/dense/torch_edge.py:97, gcn_lib/sparse/torch_edge.py:46); then K <= N-1
 *   workspace (optional, dgcn_knn_dense_workspace_bytes): enables the candidate-filter fast path for N >= 1024
 *            (pre-pass: |x_j|^2 and a per-row sampled threshold; 16 rows per workgroup on the matrix cores; the
 *            rows it cannot finish are listed on the device and redone by the exact path);
 *            without it every row takes the exact full-row path.  For C in {32, 64} the fast path evaluates the
 *            inner products on the bf16 matrix pipe from exact three-way bf16 splits (six products, fp32-faithful:
 *            csrc/bf16x6.h) instead of the channel-ordered fma chain: the same distances up to fp32 rounding, so two
 *            candidates closer than that may be ranked either way (as on any other fp32 evaluation of the reference's
 *            formula); every row is ranked by one evaluation only.  Other widths, or a workspace sized with C = 0
 *            (no room for the bf16 planes): the fp32-MFMA chain, identical results either way.
 *            For C in {32, 64} the size reported also holds 1024 (key, id) candidate pairs per point behind the planes
 *            (8 KiB per point): with them the filter pass runs 32 query rows per workgroup and keeps its candidate lists
 *            in the workspace; a workspace that is B*N*8192 bytes smaller selects the 16-row kernel with LDS lists
 *            (rounds 4 - 5; same ids).
 * Limits: N <= 4096, K <= 1024 (the candidate-filter fast path serves K <= 512), K <= N. */
size_t dgcn_knn_dense_workspace_bytes(int32_t B, int32_t N, int32_t C);
int dgcn_knn_dense_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B, int32_t C,
                       int32_t N, int32_t K, int32_t dilation, int32_t exclude_self, int64_t* nn_out,
                       int64_t* ctr_out, void* workspace, size_t workspace_bytes, void* stream);

/* Per-vertex GEMM on fp32 MFMA: out[(b*N+n)*M + m] = sum_c x[b,c,n] * W[c*M+m] + bias[m].
 * With W = [(W1-W2)^T | W2^T] this yields P and Q of the EdgeConv split
 * (replaces the 1x1 Conv2d over (B,2C,N,k) of BasicConv inside EdgeConv2d,
 *  gcn_lib/dense/torch_vertex.py:34 + gcn_lib/dense/torch_nn.py:52).  bias may be NULL. */
int dgcn_vertex_gemm_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B, int32_t C,
                         int32_t N, const float* W, const float* bias, int32_t M, float* out,
                         void* stream);

/* EdgeConv2d P/Q producer straight from the Conv2d parameters (no host-side weight re-packing):
 * conv_w [Cout][2C] = the (Cout,2C,1,1) weight, bias [Cout] or NULL;
 * out [B,N,2*Cout] = [ (W1-W2) x + b | W2 x ]  with W = [W1 | W2]. */
int dgcn_edgeconv_pq_f32(const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B, int32_t C, int32_t N,
                         const float* conv_w, const float* bias, int32_t Cout, float* out, void* stream);

/* Neighbourhood reduction  a_{bnl} = act(P[b,n,:] + Q[b, idx[b,n,l], :]),  l < k:
 *   vmax/vmin [B,N,C]  max_l / min_l a   (vmin, amin optional)
 *   amax/amin [B,N,C]  uint8 slot l attaining it (first on ties), saved for the backward
 *   stats     [dgcn_dense_edge_reduce_num_partials(B,N,C)][2][C] per-workgroup partial sums of a and
 *             a^2 (BatchNorm2d training statistics over B*N*k), fixed reduction order; optional
 * Replaces batched_index_select x2 + cat + act + torch.max (gcn_lib/dense/torch_nn.py:75-96,
 * gcn_lib/dense/torch_vertex.py:16-20,31-35).  P may be NULL (MRConv2d: Q = x point-major, act none).
 *   P/Q rows may be strided (ldp/ldq floats between consecutive points, multiples of 4): P and Q are the two
 *   halves of one vertex-GEMM output;  idx int64 (B,N,k) with element strides;  act: 0 none, 1 relu, 2 leaky-relu(slope);  C % 4 == 0, k <= 255. */
int32_t dgcn_dense_edge_reduce_num_partials(int32_t B, int32_t N, int32_t C);

int dgcn_dense_edge_reduce_fwd_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq,
                                   const int64_t* idx, int64_t idx_sb, int64_t idx_sn, int64_t idx_sk, int32_t B, int32_t N, int32_t C,
                                   int32_t k, int32_t act, float slope, float* vmax, float* vmin,
                                   uint8_t* amax, uint8_t* amin, float* stats, void* stream);

/* Backward of the above for L(vmax, vmin, sum a, sum a^2):
 *   dL/da_e = gmax*[l==amax] + gmin*[l==amin] + gsum[c] + 2 a_e gsq[c];  dz = dL/da * act'(z)
 *   sel_scale [C] (optional): BatchNorm scale; gmax is then routed to the arg-max slot where scale >= 0 and to
 *   the arg-min slot where scale < 0, and gmin is ignored (needs amin).
 *   dP[b,n,:] = sum_l dz (overwritten, optional).  dQ, two forms:
 *   (a) dq_parts != NULL, nsplit = dgcn_dense_edge_reduce_bwd_nsplit(B,N,C) > 0: each workgroup accumulates an
 *       8-channel slice of dQ[b] in LDS (ds_add_f32) and writes dq_parts[s][b][j][c] (dense, fully overwritten);
 *       the caller sums over s in a fixed order -> deterministic, no global atomics;  dQ is ignored;
 *   (b) dq_parts == NULL: dQ[b,j,:] += dz with hardware fp32 global atomics (dQ zero-filled by the caller;
 *       the last bits may vary run to run). */
int dgcn_dense_edge_reduce_bwd_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq,
                                   const int64_t* idx, int64_t idx_sb, int64_t idx_sn, int64_t idx_sk, int32_t B, int32_t N, int32_t C,
                                   int32_t k, int32_t act, float slope, const uint8_t* amax,
                                   const uint8_t* amin, const float* gmax, const float* gmin,
                                   const float* gsum, const float* gsq, const float* sel_scale, float* dP,
                                   float* dQ, float* dq_parts, int32_t nsplit, void* stream);

/* Leading dimension of `dq_parts` for the atomic-free backward, or 0 when N*32 bytes exceed the LDS. */
int32_t dgcn_dense_edge_reduce_bwd_nsplit(int32_t B, int32_t N, int32_t C);

/* ------------------------------------------------------------------------------------
 * BatchNorm2d around the neighbourhood max (EdgeConv2d with norm='batch',
 * gcn_lib/dense/torch_nn.py:54-56 + gcn_lib/dense/torch_vertex.py:34).  BN is a per-channel affine map
 * y = scale*a + shift, so max_l y = scale*(scale >= 0 ? max_l a : min_l a) + shift: node-sized work.
 *   bnbuf [4][C] = scale, shift, mean, invstd (written by dgcn_bn_finalize_f32)
 * ------------------------------------------------------------------------------------ */

/* stats [nparts][2][C] (from dgcn_dense_edge_reduce_fwd_f32) -> bnbuf; training=1 uses the batch statistics
 * over `count` = B*N*k activations (biased variance) and updates running_mean/var (momentum, unbiased
 * variance) and *num_batches += 1 when given; training=0 uses the running statistics.  One workgroup, fixed
 * summation order, fp64 accumulation.  gamma/beta NULL = 1/0. */
int dgcn_bn_finalize_f32(const float* stats, int32_t nparts, int32_t C, double count, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                         int32_t training, float momentum, float eps, float* bnbuf, void* stream);

/* out[b,c,n] = scale_c * (scale_c >= 0 ? vmax : vmin)[b,n,c] + shift_c : (B,N,C) point-major extremes ->
 * (B,C,N,1) channel-major layer output (LDS tile transpose).  bnbuf NULL = identity (norm=None). */
int dgcn_bn_apply_f32(const float* vmax, const float* vmin, const float* bnbuf, float* out, int32_t B,
                      int32_t N, int32_t C, void* stream);
/* The same with the block's skip connection in the store: out += res_scale * res[b,c,n] (element strides rb, rc, rn; the
 * `self.body(x) + x * self.res_scale` of ResDynBlock2d, gcn_lib/dense/torch_vertex.py:100-101).  res NULL = the call above. */
int dgcn_bn_apply_res_f32(const float* vmax, const float* vmin, const float* bnbuf, const float* res, int64_t rb,
                          int64_t rc, int64_t rn, float res_scale, float* out, int32_t B, int32_t N, int32_t C,
                          void* stream);

/* Backward prologue: g (B,C,N) with element strides -> gsel[b,n,c] = g*scale (point-major) and per-workgroup
 * partial sums of g and g*sel, partial [dgcn_bn_bwd_num_partials(B,N)][2][C] (NULL to skip). */
int32_t dgcn_bn_bwd_num_partials(int32_t B, int32_t N);
int dgcn_bn_bwd_prep_f32(const float* g, int64_t gb, int64_t gc, int64_t gn, const float* vmax,
                         const float* vmin, const float* bnbuf, float* gsel, float* partial, int32_t B,
                         int32_t N, int32_t C, void* stream);

/* partial -> coef [4][C] = dgamma, dbeta, gsum, gsq where (gsum, gsq) are the per-channel coefficients that
 * make dgcn_dense_edge_reduce_bwd_f32 reproduce the exact BatchNorm backward (0 when training=0). */
int dgcn_bn_bwd_finalize_f32(const float* partial, int32_t nparts, int32_t C, double count,
                             const float* gamma, const float* bnbuf, int32_t training, float* coef,
                             void* stream);

/* The same backward with dQ produced through INVERSE NEIGHBOUR LISTS instead of atomics: the edge kernel writes each
 * edge's dz row once, a counting sort groups the edge ids by neighbour, a gather kernel sums every neighbour's rows.
 * dP and dQ are fully overwritten (no pre-zeroing); workspace >= dgcn_dense_edge_reduce_bwd_inv_workspace_bytes
 * (B*N*k*C floats + ~2*B*N + 2*B*N*k ints).  Sums follow the fill order: reproducible up to fp32 rounding. */
size_t dgcn_dense_edge_reduce_bwd_inv_workspace_bytes(int32_t B, int32_t N, int32_t C, int32_t k);
int dgcn_dense_edge_reduce_bwd_inv_f32(const float* P, int64_t ldp, const float* Q, int64_t ldq, const int64_t* idx,
                                       int64_t idx_sb, int64_t idx_sn, int64_t idx_sk, int32_t B, int32_t N,
                                       int32_t C, int32_t k, int32_t act, float slope, const uint8_t* amax,
                                       const uint8_t* amin, const float* gmax, const float* gmin, const float* gsum,
                                       const float* gsq, const float* sel_scale, float* dP, float* dQ,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* What is left of EdgeConv2d's backward once dPQ [B][N][2 Cout] = [dP | dQ] exists (P = (W1 - W2) x + b, Q = W2 x with
 * conv_w = [W1 | W2] the (Cout, 2C, 1, 1) Conv2d weight of BasicConv: gcn_lib/dense/torch_vertex.py:31-35,
 * gcn_lib/dense/torch_nn.py:48-60; the reference's autograd runs conv2d_backward over the (B, 2C, N, k) edge tensor).
 *   dgcn_edgeconv_bwd_input_f32:  dx [B][C][N] = (W1 - W2)^T dP + W2^T dQ (+ res_scale * g: the skip connection of
 *       ResDynBlock2d, torch_vertex.py:101; g = the upstream gradient (B, C, N) with element strides, or NULL).
 *   dgcn_edgeconv_bwd_weight_f32: partials [dgcn_edgeconv_bwd_weight_num_partials(B, N)][Cout * 2C + Cout], each block
 *       = [dW1 | dW2] in conv_w's layout (dW1 = dP^T x, dW2 = (dQ - dP)^T x) followed by db = sum dP, over that
 *       workgroup's points; x (B, C, N) with element strides.  Sum the blocks with dgcn_reduce_partials_f32.
 * fp32 MFMA (v_mfma_f32_16x16x4_f32: an fma chain), fixed order, any C / Cout. */
int dgcn_edgeconv_bwd_input_f32(const float* dpq, const float* conv_w, const float* g, int64_t gsb, int64_t gsc,
                                int64_t gsn, float res_scale, int32_t B, int32_t C, int32_t N, int32_t Cout, float* dx,
                                void* stream);
int32_t dgcn_edgeconv_bwd_weight_num_partials(int32_t B, int32_t N);
int dgcn_edgeconv_bwd_weight_f32(const float* dpq, const float* x, int64_t sb, int64_t sc, int64_t sn, int32_t B,
                                 int32_t C, int32_t N, int32_t Cout, float* partials, void* stream);

/* dst[row*ld + c] = sum_s parts[s][row][c]  (fixed order; C % 4 == 0): combines the dq_parts of the
 * atomic-free edge backward into the Q half of the vertex-GEMM gradient. */
int dgcn_reduce_parts_f32(const float* parts, int32_t nsplit, int64_t rows, int32_t C, float* dst, int64_t ld,
                          void* stream);

/* out[w] = sum_s parts[s][w], s < nparts, w < width: the fixed-order sum of the per-workgroup partial blocks the backward
 * entry points leave (enc_grad_partials of dgcn_gen_aggr_enc_bwd_f32 / dgcn_enc_max_bwd_weight_f32, grad_w_partials of
 * dgcn_egemm_max_bwd_f32, the (d beta | d gamma) partials of dgcn_rows_ln_act_bwd_f32, the column sums of
 * dgcn_rows_linear_f32's EPI = 2) -- what autograd's sum over the workgroup axis does in the reference's graph, in one launch
 * instead of torch's memset + reduce pair.  Bit-reproducible. */
int dgcn_reduce_partials_f32(const float* parts, int32_t nparts, int64_t width, float* out, void* stream);
/* The same sum over blocks of shape [rows][inner], leaving as two contiguous arrays: out [rows][inner - 1] (all columns but
 * the last) and out_last [rows] (the last column): the [dW | db] blocks of dgcn_gen_aggr_enc_bwd_f32 /
 * dgcn_enc_max_bwd_weight_f32 as the weight and bias gradients autograd hands on.  inner >= 2. */
int dgcn_reduce_partials_split_f32(const float* parts, int32_t nparts, int64_t rows, int32_t inner, float* out,
                                   float* out_last, void* stream);

/* ------------------------------------------------------------------------------------
 * Node-wise BatchNorm1d on row-major (rows, C) features, optional fused ReLU  (SURVEY.md §8 f1).
 * Replaces nn.BatchNorm1d from norm_layer('batch', C) (gcn_lib/sparse/torch_nn.py:23-34) and the
 * Lin -> BatchNorm1d -> ReLU run inside MLP (gcn_lib/sparse/torch_nn.py:50-71).
 *   forward : dgcn_rows_stats_f32 -> dgcn_bn_finalize_f32 (above; count = rows) -> dgcn_rows_bn_apply_f32
 *   backward: dgcn_rows_bn_bwd_stats_f32 -> dgcn_rows_bn_bwd_finalize_f32 -> dgcn_rows_bn_bwd_apply_f32
 * x has row stride ld (floats); g, y, dx are contiguous (rows, C).  C % 4 == 0 with 16-byte aligned pointers takes
 * the float4 path (C <= 1024), anything else a scalar path (C <= 256).  y (the forward output) is only read as the
 * ReLU mask [y > 0]; pass NULL when no ReLU was fused.
 * ------------------------------------------------------------------------------------ */
int32_t dgcn_rows_num_partials(int64_t rows, int32_t C);

/* partial [dgcn_rows_num_partials][2][C] = per-workgroup sum x, sum x^2 (fixed order). */
int dgcn_rows_stats_f32(const float* x, int64_t ld, int64_t rows, int32_t C, float* partial, void* stream);

/* y = scale*x + shift (bnbuf rows 0,1), then max(.,0) when relu != 0. */
int dgcn_rows_bn_apply_f32(const float* x, int64_t ld, const float* bnbuf, int32_t relu, float* y,
                           int64_t rows, int32_t C, void* stream);

/* partial [nparts][2][C] = sum g', sum g'*xhat with g' = g*[y>0] and xhat = (x-mean)*invstd (bnbuf rows 2,3). */
int dgcn_rows_bn_bwd_stats_f32(const float* g, const float* x, int64_t ld, const float* y, const float* bnbuf,
                               float* partial, int64_t rows, int32_t C, void* stream);

/* coef [4][C] = dgamma, dbeta, c1 = sum g'/count, c2 = sum g'*xhat/count (c1 = c2 = 0 when training == 0). */
int dgcn_rows_bn_bwd_finalize_f32(const float* partial, int32_t nparts, int32_t C, double count,
                                  int32_t training, float* coef, void* stream);

/* dx = scale*(g' - c1 - xhat*c2). */
int dgcn_rows_bn_bwd_apply_f32(const float* g, const float* x, int64_t ld, const float* y, const float* bnbuf,
                               const float* coef, float* dx, int64_t rows, int32_t C, void* stream);

/* The pre-activation run  norm -> ReLU -> dropout  of the 'res+' blocks (examples/ogb/ogbn_arxiv/model.py:90-106) and of
 * the reversible BasicBlock (eff_gcn_modules/rev/rev_layer.py:35-51) as ONE apply pass, and its backward.
 *   drop_mode 0: no dropout.
 *   drop_mode 1: element i of the flattened (rows, C) array is kept iff u16(i) >= drop_thr, where u16 are 16 bits of a
 *                murmur-style hash of (seed0, seed1, i >> 2) (csrc/rows_norm.hip: drop_rand4); drop probability
 *                drop_thr / 65536, kept values are scaled by 65536 / (65536 - drop_thr).  The mask is regenerated from
 *                the seed in the backward: nothing is stored.
 *   drop_mode 2: multiply by drop_mask (rows, C), row stride drop_mask_ld floats -- SharedDropout's mask tensor
 *                (rev_layer.py:12-24); a per-group chunk view of the model-level (N, hidden) mask is used in place.
 * The backward recomputes the ReLU mask [scale*x + shift > 0] from x and bnbuf (the forward output is not read);
 * gadd (rows, C) or NULL is added to dx (gradient of a skip connection around the block). */
int dgcn_rows_bn_act_apply_f32(const float* x, int64_t ld, const float* bnbuf, int32_t relu, int32_t drop_mode,
                               const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1, uint32_t drop_thr, float* y,
                               int64_t rows, int32_t C, void* stream);
int dgcn_rows_bn_act_bwd_stats_f32(const float* g, const float* x, int64_t ld, const float* bnbuf, int32_t relu,
                                   int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1,
                                   uint32_t drop_thr, float* partial, int64_t rows, int32_t C, void* stream);
int dgcn_rows_bn_act_bwd_apply_f32(const float* g, const float* x, int64_t ld, const float* bnbuf, const float* coef,
                                   int32_t relu, int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0,
                                   uint32_t seed1, uint32_t drop_thr, const float* gadd, float* dx, int64_t rows,
                                   int32_t C, void* stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the channels of row-major (rows, C) features, optional fused ReLU  (SURVEY.md §8 f1).
 * Replaces nn.LayerNorm from norm_layer('layer', C) (gcn_lib/sparse/torch_nn.py:23-34: the default norm of the
 * ogbn-proteins / ogbg-ppa / RevGCN configurations) and the Lin -> LayerNorm -> ReLU run of MLP (:50-71).
 *   C % 4 == 0, C <= 1024, 16-byte aligned pointers, ld % 4 == 0.  gamma / beta may be NULL (no affine).
 *   forward : y = [relu]((x - mean_r) * rstd_r * gamma + beta); mean[rows], rstd[rows] are written for the backward
 *   backward: dx (may be NULL) and, when partial != NULL, per-workgroup partial sums
 *             partial [dgcn_rows_ln_num_partials][2][C]: slot 0 = sum_rows g', slot 1 = sum_rows g' * xhat
 *             (g' = g * [y > 0] when y, the forward output, is given); their sum over the first axis is
 *             (dbeta | dgamma).
 * ------------------------------------------------------------------------------------ */
int32_t dgcn_rows_ln_num_partials(int64_t rows, int32_t C);

int dgcn_rows_ln_fwd_f32(const float* x, int64_t ld, const float* gamma, const float* beta, float eps, int32_t relu,
                         float* y, float* mean, float* rstd, int64_t rows, int32_t C, void* stream);

int dgcn_rows_ln_bwd_f32(const float* g, const float* x, int64_t ld, const float* y, const float* gamma,
                         const float* mean, const float* rstd, float* dx, float* partial, int64_t rows, int32_t C,
                         void* stream);

/* LayerNorm -> [ReLU] -> [dropout] in one pass (drop_mode as above); the backward recomputes the ReLU mask
 * [xhat*gamma + beta > 0] from x, adds gadd (or NULL) to dx. */
int dgcn_rows_ln_act_fwd_f32(const float* x, int64_t ld, const float* gamma, const float* beta, float eps, int32_t relu,
                             int32_t drop_mode, const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1,
                             uint32_t drop_thr, float* y, float* mean, float* rstd, int64_t rows, int32_t C,
                             void* stream);
int dgcn_rows_ln_act_bwd_f32(const float* g, const float* x, int64_t ld, const float* gamma, const float* beta,
                             const float* mean, const float* rstd, int32_t relu, int32_t drop_mode,
                             const float* drop_mask, int64_t drop_mask_ld, uint32_t seed0, uint32_t seed1, uint32_t drop_thr,
                             const float* gadd, float* dx, float* partial, int64_t rows, int32_t C, void* stream);

/* MsgNorm (gcn_lib/sparse/torch_message.py:88-99) fused with GENConv's residual (torch_vertex.py:70-74):
 *   y_r = [x_r +] m_r / max(||m_r||_2, 1e-12) * ||x_r||_2 * (*scale)      rows independent, C % 4 == 0, C <= 1024
 * backward: dx, dm (either may be NULL) and ds_partial [dgcn_rows_ln_num_partials(rows, C)] whose sum is d scale. */
int dgcn_rows_msgnorm_fwd_f32(const float* x, int64_t ldx, const float* m, const float* scale, int32_t add_x,
                              float* y, int64_t rows, int32_t C, void* stream);

int dgcn_rows_msgnorm_bwd_f32(const float* g, const float* x, int64_t ldx, const float* m, const float* scale,
                              int32_t add_x, float* dx, float* dm, float* ds_partial, int64_t rows, int32_t C,
                              void* stream);

/* ------------------------------------------------------------------------------------
 * Node-wise Linear with the neighbouring row passes folded in  (csrc/rows_linear.hip).
 * Replaces the Linear stages of MLP (gcn_lib/sparse/torch_nn.py:50-71), mlp(x + m) with the 'res+' residual
 * h = conv(h2) + h (gcn_lib/sparse/torch_vertex.py:70-76, examples/ogb/ogbn_arxiv/model.py:90-106), the statistics pass
 * of the following BatchNorm and, in the backward launch, the bias gradient:
 *     y[r, c] = [relu]( sum_k x[r, k] * W[c, k] + bias[c] + res[r, c] )        rows >> K, C
 * fp32-faithful on the bf16 matrix pipe (six-product split, csrc/bf16x6.h: max error / sum|x||w| = 1.7e-7).
 *   x (rows, K) row stride ldx (16-byte aligned rows); K % 4 == 0, 16 <= K <= 256; 1 <= C <= 2048
 *   w_trans 0: w is the nn.Linear weight (C, K), row stride ldw;  1: w is (K, C), row stride ldw -- the input
 *              gradient dX = G W of a Linear(C_in = C here ... ) is this call with x = G and w = its weight
 *   bias (C) or NULL; res (rows, C) row stride ldr or NULL; y (rows, C) row stride ldy
 *   relu      bit 0: the ReLU above.  Bits 1 and 2 (res required; no ReLU, col_stats or xcol_sum; DGCN_E_MODE
 *              otherwise) are the additive coupling of the reversible layers (eff_gcn_modules/rev/memgcn.py:36-52)
 *              written where the caller assembles its rows (ldy = the full row): bit 2: y = res + (x W^T + bias) with
 *              the residual joining BEHIND the product chain (one rounding at its magnitude, as the elementwise pass
 *              it replaces -- 112 stacked couplings accumulate the difference); bit 1: y = res - (x W^T + bias), the
 *              inverse x_i = y_i - F_i(.), likewise
 *   col_stats NULL or [dgcn_rows_linear_num_partials][2][C]: per-workgroup sum y | sum y^2 (what dgcn_bn_finalize_f32
 *              takes as `partial`, count = rows)
 *   xcol_sum  NULL or [dgcn_rows_linear_num_partials][K]: per-workgroup column sums of x (K <= 128, no res / col_stats):
 *              the bias gradient when x is the upstream gradient
 * ------------------------------------------------------------------------------------ */
int32_t dgcn_rows_linear_supported(int32_t K, int32_t C);
int32_t dgcn_rows_linear_num_partials(int64_t rows, int32_t K, int32_t C);
int dgcn_rows_linear_f32(const float* x, int64_t ldx, int64_t rows, const float* w, int64_t ldw, int32_t w_trans,
                         const float* bias, const float* res, int64_t ldr, float* y, int64_t ldy, int32_t K, int32_t C,
                         int32_t relu, float* col_stats, float* xcol_sum, void* stream);

/* Weight gradient of a row-wise Linear:  dW[c][k] = sum_r g[r][c] * x[r][k]  (g^T x; csrc/rows_tn.hip), fp32-faithful on
 * the bf16 matrix pipe.  Replaces the backward GEMMs with a 10^5 .. 10^6-long reduction: the gradient of
 * edge_encoder.weight (dz^T F; gcn_lib/sparse/torch_vertex.py:63-66 under eff_gcn_modules/rev/gcn_revop.py:121-133) and
 * of the MLP Linear weights (gcn_lib/sparse/torch_nn.py:50-71).
 *   g (rows, C) row stride ldg, x (rows, K) row stride ldx; 1 <= C <= 128, 1 <= K <= 256
 *   partials [dgcn_rows_tn_num_partials(rows, C, K)][C][K]: scratch, one partial per workgroup
 *   out (C, K) row stride ldo >= K: the sum of the partials in a fixed order (a second launch on the same stream): the
 *   result is deterministic.  Row strides are limited to 2^31 / 128 bytes (DGCN_E_SHAPE beyond). */
int32_t dgcn_rows_tn_supported(int32_t C, int32_t K);
int32_t dgcn_rows_tn_num_partials(int64_t rows, int32_t C, int32_t K);
int dgcn_rows_tn_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t C, int32_t K,
                     float* partials, float* out, int64_t ldo, void* stream);
/* The same pass also yields the bias gradient of that Linear (the `g.sum(0)` of gcn_lib/sparse/torch_nn.py:50-71's
 * backward): colsum_of = 1 -> colsum[c] = sum_r g[r][c] (C values), 2 -> colsum[k] = sum_r x[r][k] (K values), 0 -> none
 * (colsum may be null).  The loading waves already hold every element once; no extra pass over the rows, no extra
 * launch.  transposed != 0 writes out as (K, C) with row stride ldo >= C -- (x^T g)^T for callers whose wide operand
 * (> 128 columns) has to be the kernel's second one.
 *   partials: dgcn_rows_tn_num_partials(rows, C, K) x (C K + [colsum_of == 1 ? C : colsum_of == 2 ? K : 0]) floats,
 *   16-byte aligned when colsum_of != 0 (DGCN_E_ALIGN otherwise), colsum 16-byte aligned. */
int dgcn_rows_tn_colsum_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t rows, int32_t C, int32_t K,
                            float* partials, float* out, int64_t ldo, int32_t transposed, int32_t colsum_of,
                            float* colsum, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGCN_H */
