/*
 * hotformerloc_hip.h -- C-ABI of libhotformerloc_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the native side of the HOTFormerLoc hot path.  Every entry
 * point takes plain device pointers and sizes (no torch types), enqueues work on
 * the given HIP stream WITHOUT synchronising (the reference's extension launches
 * on the current stream: libs/dwconv/csrc/dwconv.cu:92,106,124) and returns 0 on
 * success or a hipError_t value (> 0) / a negative HFL_E* code on failure; the
 * reference aborts on launch errors (libs/dwconv/csrc/utils.h:12-24), the host
 * wrapper turns non-zero into a Python exception.
 *
 * All feature matrices are row-major float32.  "neigh" tables hold row indices,
 * -1 = no neighbour.  Tables may be int64 (the reference's dtype, dwconv.cu:27)
 * or int32 (what this library builds itself): `idx64` selects.
 *
 * The reference interface each function replaces is cited as file:line of
 * csiro-robotics/HOTFormerLoc; ocnn==2.2.2 is the un-vendored dependency
 * (requirements.txt:6) whose behaviour SURVEY.md Appendix A restates.
 */
#ifndef HOTFORMERLOC_HIP_H
#define HOTFORMERLOC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* hfl_stream_t; /* hipStream_t */

#define HFL_OK 0
#define HFL_EINVAL (-1)      /* unsupported shape / argument            */
#define HFL_ECAPACITY (-2)   /* input exceeds a documented kernel limit */
#define HFL_EBACKEND (-100)  /* hipBLASLt refused the problem: code = -100 - hipblasStatus_t */

/* library / device identification: returns the version as major*100+minor */
int hfl_version(void);
/* name of the code-object architecture this library was compiled for */
const char* hfl_arch(void);

/* ------------------------------------------------------------------------
 * 1. Octree depth-wise convolution  (replaces dwconv.core, libs/dwconv)
 * ---------------------------------------------------------------------- */

/* out[h,c] = sum_k [neigh[h,k] >= 0] * weight[k,c] * data[neigh[h,k], c]
 *   replaces  Tensor dwconv_forward_backward(Tensor data, Tensor weight, Tensor neigh)
 *             libs/dwconv/csrc/dwconv.h:13, dwconv.cu:24-42,99-113, pybind.cpp:11
 *   data (n_in,C) ; weight (K,1,C) ; neigh (n_out,K) ; out (n_out,C).  Also used for
 *   the input gradient with the inverse table (libs/dwconv/dwconv/nn.py:36-38). */
int hfl_dwconv_forward_backward(float* out, const float* data, const float* weight,
                                const void* neigh, int idx64, int64_t n_out,
                                int64_t channels, int kngh, hfl_stream_t stream);

/* out[h,c] = sum_k [slot[h,k] >= 0] part[slot[h,k], c] [+ bias[c]]  (round 6): the per-row sum of the partial products of an
 * octree convolution over its live (row, tap) pairs -- ocnn's octree2col + mm scatters nothing, it multiplies the zero-padded
 * column matrix (models/layers/octformer_layers.py:89-95); here every output row adds the products of its own live taps in tap
 * order -- and the convolution's bias.  part (P, C) f32, slot (n_out, K) int32, -1 = no such neighbour, bias (C) or NULL.
 * Equals hfl_dwconv_forward_backward with unit weights (what rounds 1-5 launched) bit for bit. */
int hfl_slot_sum(float* out, const float* part, const int32_t* slot, const float* bias, int64_t n_out, int64_t channels, int kngh,
                 hfl_stream_t stream);

/* gW[k,c] = sum_h [neigh[h,k] >= 0] data[neigh[h,k], c] * grad[h,c]
 *   replaces  Tensor dwconv_weight_backward(Tensor grad, Tensor data, Tensor neigh)
 *             libs/dwconv/csrc/dwconv.h:14, dwconv.cu:44-72,115-131, pybind.cpp:12
 *   out (K,1,C).  `workspace` must hold hfl_dwconv_weight_backward_workspace() bytes. */
int64_t hfl_dwconv_weight_backward_workspace(int64_t n_rows, int64_t channels, int kngh);
int hfl_dwconv_weight_backward(float* out, const float* grad, const float* data,
                               const void* neigh, int idx64, int64_t n_rows,
                               int64_t channels, int kngh, void* workspace,
                               hfl_stream_t stream);

/* ineigh[neigh[h,k], k] = h on a -1-filled (n_rows,K) table
 *   replaces  Tensor inverse_neigh(Tensor neigh)
 *             libs/dwconv/csrc/dwconv.h:15, dwconv.cu:74-97, pybind.cpp:13 */
int hfl_inverse_neigh(void* ineigh, const void* neigh, int idx64, int64_t n_rows,
                      int kngh, hfl_stream_t stream);

/* Fused conditional position encoding:
 *   y = LayerNorm_C( dwconv(x) ) * gamma + beta ;  out = residual ? x + y : y
 *   replaces  CPE.forward + the residual of its callers
 *             models/layers/octformer_layers.py:138-142,
 *             models/octformer_backbone.py:258, models/hotformerloc_backbone.py:204,351
 *   x,out (n,C) with C in {128,256} (C % 64 == 0, C <= 512); neigh int32 (n,27). */
int hfl_cpe_forward(float* out, const float* x, const float* weight, const float* gamma,
                    const float* beta, const int32_t* neigh, int64_t n_rows,
                    int64_t channels, int kngh, float eps, int residual,
                    hfl_stream_t stream);
/* The same launch for the training forward: additionally writes conv_out (n, C) = dwconv(x), the input of the LayerNorm
 * backward (hfl_layer_norm_bwd_add), so that loss.backward() through CPE (training/trainer.py:344-362) needs no second
 * pass over the neighbourhoods; conv_out = NULL is hfl_cpe_forward. */
int hfl_cpe_forward_save(float* out, float* conv_out, const float* x, const float* weight, const float* gamma,
                         const float* beta, const int32_t* neigh, int64_t n_rows,
                         int64_t channels, int kngh, float eps, int residual,
                         hfl_stream_t stream);
/* The convolution alone through the same gather (live taps compacted per row): out = dwconv(data, weight, neigh) [+ add].
 * The data gradient of CPE in loss.backward(): data = gradient of the convolution's output, neigh = the INVERSE neighbour
 * table (hfl_inverse_neigh, libs/dwconv/dwconv/nn.py:36-38), add = the skip connection's gradient (or NULL).
 * int32 tables, C in {32, 64, 128, 256}, kngh <= 27; out must not alias data. */
int hfl_dwconv_add(float* out, const float* data, const float* weight, const int32_t* neigh, const float* add,
                   int64_t n_out, int64_t channels, int kngh, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 2. Octree construction  (replaces ocnn.octree.Octree.build_octree /
 *    merge_octrees / construct_all_neigh; call sites datasets/dataset_utils.py:89-94,
 *    eval/pnv_evaluate.py:173-175,123, misc/torch_utils.py:47-51)
 * ---------------------------------------------------------------------- */

/* Limits of the one-workgroup-per-cloud builder. */
#define HFL_OCTREE_MAX_POINTS (1 << 20) /* points per cloud; <= 16384 sort entirely in LDS */
#define HFL_OCTREE_MAX_DEPTH 10         /* 30-bit shuffled keys                            */

/* Bytes of scratch hfl_octree_build_clouds needs. */
int64_t hfl_octree_scratch_bytes(int64_t total_points, int batch, int max_points, int depth);

/* Stage 1: one workgroup per cloud.  points (P,3) in [-1,1]; cloud_offsets (B+1)
 * int64 on the DEVICE; total_points = P and max_points = the largest cloud are the
 * host's copies of the same numbers (they size the scratch strides and the LDS sort).
 * Writes per-cloud sorted unique node keys and child slots for depths
 * full_depth..depth into `scratch`, the per-leaf point averages (scaled to
 * [0,2^depth], ocnn's `points[depth]`) into leaf_points (P,3) at the cloud's point
 * offset, and counts[(depth+1) * B] int32 = non-empty nodes per depth and cloud
 * (`batch_nnum_nempty`, layout (depth+1,B) as models/octree.py:163 reads it).
 * Returns HFL_ECAPACITY when max_points > HFL_OCTREE_MAX_POINTS. */
int hfl_octree_build_clouds(const float* points, const int64_t* cloud_offsets, int batch,
                            int64_t total_points, int max_points, int depth, int full_depth,
                            void* scratch, float* leaf_points, int32_t* counts,
                            hfl_stream_t stream);

/* Stage 2 (after the host has read `counts` and allocated exact-size outputs):
 * merge into batch arrays, ocnn `merge_octrees` layout.  For every depth d in
 * [0,depth], pointers indexed by d:
 *   keys[d]     int64 (nnum_d)      shuffled key | batch << 48      (may be NULL)
 *   children[d] int32 (nnum_d)      index among non-empty nodes of depth d, or -1
 *   nkeys[d]    int64 (nne_d)       keys of the non-empty nodes only
 *   nidx[d]     int32 (nne_d)       position of each non-empty node among all nodes
 * children[d] for d >= full_depth must be pre-filled with -1 by the caller.
 * points_out (nne_depth,3) receives the compacted leaf averages.
 * cum_nne (depth+1, B+1) int32 DEVICE: exclusive prefix of counts over clouds. */
int hfl_octree_merge(const void* scratch, const int64_t* cloud_offsets, const int32_t* cum_nne,
                     int batch, int depth, int full_depth, int64_t total_points,
                     int64_t* const* keys, int32_t* const* children,
                     int64_t* const* nkeys, int32_t* const* nidx,
                     const float* leaf_points, float* points_out, hfl_stream_t stream);

/* 27-neighbour table of the non-empty nodes of one depth, equal to ocnn
 * `get_neigh(depth,'333',stride=1,nempty=True)`:
 *   depth <= full_depth : by key arithmetic (children_d maps key -> non-empty rank)
 *   depth  > full_depth : parent walk through neigh_parent (nne_{d-1},27),
 *                         nidx_d and children_d.
 * neigh_out int32 (nne_d,27). */
int hfl_octree_neigh(int32_t* neigh_out, const int32_t* neigh_parent, const int32_t* nidx,
                     const int32_t* children, const int64_t* nkeys, int64_t nne, int depth,
                     int full_depth, hfl_stream_t stream);

/* tok_meta (n,2) uint32 per non-empty node of one depth: [x | y<<10 | z<<20, batch id]
 * (ocnn key2xyz / batch_id as read by models/octree.py:132,273-275). */
int hfl_token_meta(uint32_t* tok_meta, const int64_t* nkeys, int64_t n, int depth,
                   hfl_stream_t stream);

/* Raw-cloud pre-steps in front of the octree build, for a whole batch (eval/pnv_evaluate.py:158-171,
 * datasets/dataset_utils.py:84-90): bounding-box normalisation into [-1,1] (datasets/augmentation.py:213-223,
 * scale_factor=None, unit_sphere_norm=False, zero_mean=True) when `normalize`, the |coordinate| <= 1 mask, the
 * |xy| <= 1 mask when `cylindrical_mask`, and -- when `cylindrical_transform` -- (x,y,z) -> (rho,phi,z) rescaled
 * to [-1,1] (datasets/coordinate_utils.py:68-116).  points (P,3) with cloud b at rows
 * [cloud_offsets[b], cloud_offsets[b+1]); the kept points of cloud b are written, in order, from row
 * cloud_offsets[b] of out_points (P,3) on, and out_counts[b] says how many.  Normalisation and masks are
 * bit-exact with the reference's fp32 torch sequence; the transform's atan2f is the device library's (see
 * csrc/preprocess.hip).  out_points must not alias points. */
int hfl_prepare_clouds(float* out_points, int32_t* out_counts, const float* points,
                       const int64_t* cloud_offsets, int batch, int normalize, int cylindrical_mask,
                       int cylindrical_transform, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. Octree convolution gather  (ocnn.nn.OctreeConv's octree2col; call sites
 *    models/layers/octformer_layers.py:89-95, models/octformer_backbone.py:470-475)
 * ---------------------------------------------------------------------- */

/* out[m, k*C + c] = neigh[m,k] >= 0 ? data[neigh[m,k], c] : 0 ;  out (n_out, K*C) */
int hfl_octree_gather(float* out, const float* data, const int32_t* neigh, int64_t n_out,
                      int kngh, int64_t channels, hfl_stream_t stream);

/* Live-tap lists of a (rows, taps) int32 index table, taps <= 32 (ocnn.nn.OctreeConv multiplies an (N, K*Cin)
 * octree2col matrix that is 80-94 % zeros on surface-like clouds; the convolutions here run over the live
 * (row, tap) pairs only):
 *   src   (rows*taps capacity) input row of every live pair, pairs ordered by tap then by row; the first
 *         edges[taps] entries are written
 *   slot  (rows, taps)         position of (row, tap) in that list, -1 where the neighbour is missing
 *   edges (taps + 1)           pairs of tap k are [edges[k], edges[k+1])          -- all on the device
 * No atomics, fixed order, no host synchronisation; `workspace` holds hfl_tap_lists_workspace() bytes. */
int64_t hfl_tap_lists_workspace(int64_t rows, int taps);
int hfl_tap_lists(int32_t* src, int32_t* slot, int32_t* edges, const int32_t* table, int64_t rows, int taps,
                  void* workspace, hfl_stream_t stream);
/* The same for n <= 16 tables in THREE launches in all (count, scan, fill over every table: the convolution depths of one
 * batch); arrays of n pointers / sizes, workspace[i] = hfl_tap_lists_workspace(rows[i], taps[i]) bytes. */
int hfl_tap_lists_multi(int n, int32_t* const* src, int32_t* const* slot, int32_t* const* edges, const int32_t* const* table,
                        const int64_t* rows, const int32_t* taps, void* const* workspace, hfl_stream_t stream);
/* Row tiles of the grouped tap GEMM (section 9b, hfl_linear_x3_grouped) from the device-side tap edges hfl_tap_lists wrote:
 * tiles (n_tiles, 3) int32 = {first pair, pairs (<= tile_rows), tap * w_rows}, n_tiles = sum_k ceil(pairs_k / tile_rows). */
int hfl_tap_tiles(int32_t* tiles, const int32_t* edges, int taps, int tile_rows, int w_rows, hfl_stream_t stream);
/* Padded gather index of a ragged per-cloud row stream (pooling head, models/layers/pooling.py:209-233): out (batch * nmax)
 * int64, row_off[b] + j inside cloud b, row_off[batch] (a zero row the caller appends) beyond it. */
int hfl_pad_index(int64_t* out, const int64_t* row_off, int batch, int64_t nmax, hfl_stream_t stream);
/* The padded copy itself in one pass: out (batch, nmax, channels) f32, rows of cloud b = x[row_off[b] ..), zeros beyond
 * (the per-cloud split + zero padding of models/layers/pooling.py:209-233 without an index table or an appended zero row). */
int hfl_pad_rows(float* out, const float* x, const int64_t* row_off, int batch, int64_t nmax, int64_t channels,
                 hfl_stream_t stream);
/* Attentional pooling of a level's ragged per-cloud rows by learned queries as ONE launch (+ a combine when the rows of a
 * cloud are split over workgroups) -- AdaptivePooling.forward, models/layers/salsa.py:25-55, per level from
 * PyramidAttnPoolWrapper.forward, models/layers/pooling.py:209-233:
 *     out[b, q, :] = sum_r softmax_r(scale * <query[q], x[r]>) x[r],   r over rows row_off[b] .. row_off[b + 1] of x (n_rows, C)
 * out + b * out_cloud_stride + q * C is row q of cloud b (so a level can write its slice of the concatenated token matrix).
 * Products as bf16 (hi, lo) splits with fp32 accumulation, softmax in fp32.  channels in {128, 256} (hfl_attn_pool_ok);
 * workspace: hfl_attn_pool_workspace bytes (without it a cloud's rows stay in one workgroup per 64 queries). */
/* Tail of the Mixer aggregator, models/layers/salsa.py:104-111 (channel_proj over the token axis, row_proj over the channel
 * axis, flatten) in one launch: out (batch, k_out * out_d), x (batch, k_tokens, channels), channel_w (k_out, k_tokens),
 * row_w (out_d, channels); out_d <= 8.  row_proj is applied first (the two maps commute): same value up to fp32 summation
 * order. */
int hfl_mixer_tail(float* out, const float* x, const float* channel_w, const float* channel_b, const float* row_w,
                   const float* row_b, int batch, int k_tokens, int channels, int k_out, int out_d, hfl_stream_t stream);
int hfl_attn_pool_ok(int channels);
int64_t hfl_attn_pool_workspace(int batch, int n_queries, int channels, int64_t n_rows);
int hfl_attn_pool(float* out, int64_t out_cloud_stride, const float* x, const int64_t* row_off, const float* query, int batch,
                  int n_queries, int channels, int64_t n_rows, float scale, void* workspace, int64_t workspace_bytes,
                  hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 4. Windowed multi-head attention over z-order octree windows
 *    (replaces OctreeAttention.forward's bias build + SDPA,
 *     models/octformer_backbone.py:59-88, RPE models/layers/octformer_layers.py:159-170,
 *     masks models/octree.py:186-222,267-283, window (un)packing models/octree.py:346-386)
 * ---------------------------------------------------------------------- */
typedef struct {
  int64_t n_tokens;     /* N_t: real tokens of this depth (rows 0..N_t-1 of qkv/out)    */
  int64_t rt_row0;      /* first relay-token row in qkv/out (G=1), ignored when G=0     */
  int32_t n_windows;    /* W = ceil(N_t / (K*D_pad)) * D_pad   (models/octree.py:73-75) */
  int32_t patch_size;   /* K  (48 or 64)                                                 */
  int32_t dilation;     /* D  (1, or the stage dilation 4 for odd OctFormer blocks)     */
  int32_t n_relay;      /* G  (0 or 1)                                                   */
  int32_t n_heads;      /* H, head dim is fixed at 16                                    */
  int32_t pos_bnd;      /* int(0.8*K*sqrt(D)), rpe table has 3*(2*pos_bnd+1) rows        */
  int32_t batch_size;   /* B: batch id given to padding                                  */
  float scale;          /* 16^-0.5                                                       */
  int32_t depth;        /* octree depth of the tokens (coords < 2^depth); 0 = unknown    */
  const float* rpe_expanded; /* optional: hfl_window_rpe_expand(rpe_table, depth) output, enables
                              * the two-lookup forward kernel; NULL = three axis lookups     */
} hfl_window_attn_desc;

/* qkv (rows, 3*H*16): per row [q(H,16) | k(H,16) | v(H,16)], the layout produced by
 * `self.qkv(data).reshape(-1, K+G, 3, H, C//H)`.  tok_meta (N_t) uint2 per token:
 * .x = x | y<<10 | z<<20 (node coords at this depth), .y = batch id.
 * rpe_table (3*(2*pos_bnd+1), H) float32 or NULL (disable_RPE).
 * out (rows, H*16).  Window w covers tokens  w*K+k  (D=1)  or
 * (w/D)*K*D + k*D + (w%D)  (dilated), k in [0,K).  Relay token of window w is row
 * rt_row0 + w and carries the batch id of the window's first token. */
int hfl_window_attention_fwd(float* out, const float* qkv, const uint32_t* tok_meta,
                             const float* rpe_table, const hfl_window_attn_desc* desc,
                             hfl_stream_t stream);

/* As hfl_window_attention_fwd, for the split-precision Linear path: qkv comes from a bias-free GEMM
 * and `qkv_bias` (3*H*16) is added to q, k, v on load (NULL = none); with out_split3 != 0 `out` is the
 * bf16 A operand [hi | hi | lo] (rows, 3*H*16) of the output projection (section 9). */
/* 1 when hfl_window_attention_fwd_ex takes the fp16 (hi, lo) qkv layout of hfl_linear_x3_qkv for this configuration
 * (`out_split3 | 0x100`; depth <= 5 with coordinates inside pos_bnd, <= 72 KiB of LDS), else 0: use fp32 qkv. */
int hfl_window_attention_f16_ok(const hfl_window_attn_desc* desc, int64_t n_rows_total);
int hfl_window_attention_fwd_ex(void* out, const float* qkv, const float* qkv_bias,
                                const uint32_t* tok_meta, const float* rpe_table,
                                const hfl_window_attn_desc* desc, int out_split3, hfl_stream_t stream);
/* n (<= 4) independent attention problems on fp16 (hi, lo) qkv operands (out_split3 carries 0x100) as ONE launch when they
 * have one shape (patch_size, n_relay, n_heads, table form), else one launch each: arrays of the arguments of
 * hfl_window_attention_fwd_ex (no qkv_bias).  Small problems ride along with a large one instead of paying a launch each. */
int hfl_window_attention_fwd_multi(int n, void* const* out, const float* const* qkv, const uint32_t* const* tok_meta,
                                   const float* const* rpe_table, const hfl_window_attn_desc* const* desc, int out_split3,
                                   hfl_stream_t stream);

/* Expanded relative-position table for the forward kernels, scaled by log2(e); which form is built depends on the consumer:
 *   f16_operand = 0 (fp32 qkv kernel): per head the x-axis table restricted to |dx| <= R = 2^depth - 1 followed by the
 *     pre-added (dy, dz) table of (2R+1)^2 entries; valid when R <= pos_bnd and depth <= 5 (size() returns 0 otherwise:
 *     the three-lookup kernel reads rpe_table itself)
 *   f16_operand = 1 (fp16 (hi, lo) qkv kernel, flag 0x100 of hfl_window_attention_fwd_ex): that form up to depth 4; from
 *     depth 5 (and whenever R > pos_bnd) three 1-D tables over the full coordinate range with the reference's clamp baked
 *     in (depth <= 7)
 *   f16_operand = 2: the three 1-D tables at every depth <= 7 (what the fp16 kernel reads from depth 5 on, for any depth)
 * out holds hfl_window_rpe_expand_size() floats; rebuild it whenever rpe_table changes.  A table built for one consumer
 * must not be handed to the other. */
int64_t hfl_window_rpe_expand_size(int n_heads, int pos_bnd, int depth, int f16_operand);
int hfl_window_rpe_expand(float* out, const float* rpe_table, int n_heads, int pos_bnd, int depth, int f16_operand,
                          hfl_stream_t stream);

/* Linear on the split operands through hipBLASLt, with the block's epilogue in the same launch:
 *   out (n_rows, out_features) f32 = a (n_rows, k_concat) bf16 . w (out_features, k_concat)^T bf16
 *                                    [+ bias (out_features) f32] [+ residual (n_rows, out_features) f32]
 * a / w are the K-concatenated [hi|hi|lo] / [hi|lo|hi] operands of section 9 (k_concat = 3 * in_features),
 * fp32 accumulation.  Replaces `x = x + proj(attn)` / `x = x + mlp(x)` of
 * models/octformer_backbone.py:275-278 and models/hotformerloc_backbone.py:213-216 (Linear + bias +
 * residual add) without a separate pass over the residual stream.  bias / residual may be NULL;
 * residual must not alias out.  Returns HFL_EBACKEND - status when the library rejects the problem. */
int hfl_gemm_bf16(float* out, const uint16_t* a, const uint16_t* w, const float* bias, const float* residual,
                  int64_t n_rows, int out_features, int k_concat, hfl_stream_t stream);

/* Weight gradient of that Linear (training path): out (n_out, k_out) f32 = a^T b with the contraction over the
 * n_rows_stacked = 3M rows of  a = [dy_hi; dy_hi; dy_lo] (3M, n_out) bf16  and  b = [x_hi; x_lo; x_hi] (3M, k_out)
 * bf16, i.e. dW = dy^T x to the same 2^-17 operand accuracy as the forward (torch autograd of nn.Linear). */
int hfl_gemm_bf16_tn(float* out, const uint16_t* a, const uint16_t* b, int64_t n_rows_stacked, int n_out,
                     int k_out, hfl_stream_t stream);

/* Test / measurement hook: select a kernel variant at run time; "reset" restores every default.  Keys kept in the product
 * build: "window_attention" (4 = default, 2 = three-lookup fp32 kernel), "window_rpe_form1_max_depth" (table form of the fp16
 * kernel), "window_bwd", "window_heads_per_wg", "window_v4_wgs_per_cu" / "window_v2_wgs_per_cu", "window_debug",
 * "tail_split" / "dynamic_units" / "mlp_stagger" (left-over rows, work tickets and start stagger of the row-tile kernels),
 * "attn_fused_split", "cpe_variant", "cpe_chunk_rows", "x3_dbg".  The variants that lost their A/B measurements in rounds
 * 2-4 (per-round launches, relay rows first, CU-masked streams, device-flag hops, the x3 ring kernel, 4-wave row tiles) are
 * gone from the library; their logs are under profiles/.  Returns HFL_EINVAL for an unknown key. */
int hfl_set_variant(const char* key, int value);

/* ------------------------------------------------------------------------
 * 5. Relay-token self-attention, ragged per cloud
 *    (replaces concat_and_pad_rt + RTAttention SDPA + unpad_and_split_rt,
 *     models/relay_token_utils.py:12-79, models/hotformerloc_backbone.py:83-119,
 *     mask models/octree.py:229-265)
 * ---------------------------------------------------------------------- */
/* qkv (rows, 3*H*16), out (rows, H*16).  Cloud b attends over the rows
 * seq_rows[seq_off[b] .. seq_off[b+1]) (its relay tokens of all pyramid depths,
 * fine to coarse; relay tokens of pure-padding windows are in no sequence).  Rows not
 * listed in seq_rows are left untouched (the caller zero-fills `out`).  max_seq_len = the
 * longest per-cloud sequence (host copy; sizes the grid). */
int hfl_relay_attention_fwd(float* out, const float* qkv, const int32_t* seq_rows,
                            const int32_t* seq_off, int batch, int n_heads, float scale,
                            int max_seq_len, hfl_stream_t stream);
/* The same attention on the fp16 (hi, lo) operand rows hfl_ln_qkv_fused writes (queries pre-multiplied by 16^-0.5 * log2 e),
 * writing attention.proj's bf16 split2 operand (rows, 2C) directly; `orphan_rows` (n_orphans of them, may be 0): rows of no
 * sequence, written as zeros (models/hotformerloc_backbone.py:83-119 on the padded sequences gives those rows no consumer). */
int hfl_relay_attention_f16_fwd(void* out_split2, const void* qkv_f16, const int32_t* seq_rows, const int32_t* seq_off, int batch,
                                int n_heads, int max_seq_len, const int32_t* orphan_rows, int n_orphans, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 6. Relay-token initialisation and window statistics
 * ---------------------------------------------------------------------- */
/* rt[w,:] = mean over tokens k of window w with batch id == id of the window's first
 * token (models/hotformerloc_backbone.py:352-360, mask models/octree.py:142-145).
 * Windows past the last token give 0.  x (N_t,C), rt (W,C). */
int hfl_relay_token_init(float* rt, const float* x, const uint32_t* tok_meta,
                         int64_t n_tokens, int32_t n_windows, int32_t patch_size,
                         int64_t channels, hfl_stream_t stream);

/* ADaPE window statistics, mode 'cov' (models/octree.py:285-344): per window mean (3)
 * and upper-triangular Bessel covariance (6) of the owner's node coordinates rescaled
 * to [-1,1] (misc/utils.py:293-304).  stats (W,9). */
int hfl_window_stats(float* stats, const uint32_t* tok_meta, int64_t n_tokens,
                     int32_t n_windows, int32_t patch_size, int depth, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 7. Attentional pooling helpers (models/layers/salsa.py:25-55,
 *    models/layers/pooling.py:209-233)
 * ---------------------------------------------------------------------- */
/* In place, per cloud b and pooled token q: scores[r, q] over rows
 * r in [row_off[b], row_off[b+1]) -> softmax over r of scores * scale.
 * scores (rows, n_queries). */
int hfl_segment_softmax(float* scores, const int64_t* row_off, int batch, int n_queries,
                        float scale, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 8. LayerNorm over the channel axis, optionally fused with the residual add that precedes it
 *    (torch.nn.LayerNorm call sites: models/octformer_backbone.py:275-278,
 *     models/hotformerloc_backbone.py:213-216,288-291; eps 1e-5, two-pass statistics)
 * ---------------------------------------------------------------------- */
/* out = LN(x) * gamma + beta ;  channels in {16,32,64,128,256,512,1024} */
int hfl_layer_norm(float* out, const float* x, const float* gamma, const float* beta, int64_t n_rows,
                   int64_t channels, float eps, hfl_stream_t stream);
/* x_out = x + y (+ bias, may be NULL) ; h_out = LN(x_out) * gamma + beta.  x_out may alias x. */
int hfl_add_layer_norm(float* x_out, float* h_out, const float* x, const float* y, const float* bias,
                       const float* gamma, const float* beta, int64_t n_rows, int64_t channels,
                       float eps, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 9. Operand producers of the split-precision Linear path.
 *    An fp32 Linear y = x W^T (torch.nn.Linear call sites: models/octformer_backbone.py:70,91,
 *    models/layers/octformer_layers.py:54,57) is evaluated as ONE bf16 matrix-core GEMM with fp32
 *    accumulation and fp32 output over K-concatenated operands
 *        A3 = [x_hi | x_hi | x_lo]  (rows, 3K) bf16,   W3 = [w_hi | w_lo | w_hi]  (N, 3K) bf16,
 *    i.e. x_hi w_hi + x_hi w_lo + x_lo w_hi with x = x_hi + x_lo to 2^-17 (measured GEMM error
 *    4e-6 relative, descriptors 1.5e-5 vs the 1e-3 bar).  These kernels write A3 directly from the
 *    op that produces x, so the split costs no extra pass.
 * ---------------------------------------------------------------------- */
/* A3 = split3(LN(x)) */
int hfl_layer_norm_split3(uint16_t* out, const float* x, const float* gamma, const float* beta,
                          int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream);
/* x_out = x + y (+ bias) ; A3 = split3(LN(x_out)) */
int hfl_add_layer_norm_split3(float* x_out, uint16_t* h_out, const float* x, const float* y,
                              const float* bias, const float* gamma, const float* beta, int64_t n_rows,
                              int64_t channels, float eps, hfl_stream_t stream);
/* out = x + y + bias (fp32; out may alias x or y) */
int hfl_add_bias(float* out, const float* x, const float* y, const float* bias, int64_t n_rows,
                 int64_t channels, hfl_stream_t stream);
/* A3 = split3(gelu(x + bias)), exact erf GELU */
int hfl_bias_gelu_split3(uint16_t* out, const float* x, const float* bias, int64_t n_rows,
                         int64_t channels, hfl_stream_t stream);
/* A3 = split3(x) */
int hfl_split3(uint16_t* out, const float* x, int64_t n_rows, int64_t channels, hfl_stream_t stream);

/* 9b. Hand-written split-precision Linear (csrc/gemm_x3.hip): y = x W^T [+ bias] [GELU] [+ residual] with
 * fp32-equivalent accuracy on the bf16 matrix cores (x_lo w_hi + x_hi w_lo + x_hi w_hi, fp32 accumulation).
 * Replaces torch.nn.Linear and the element-wise op after it (models/octformer_backbone.py:70,91,275-278;
 * models/layers/octformer_layers.py:53-59; models/hotformerloc_backbone.py:213-216).
 * Operands are PRE-SPLIT in the "split2" layout (rows, K/32, 2, 32) bf16: per 32-wide k-block [32 x hi | 32 x lo]
 * (hi = RNE(v), lo = RNE(v - hi)); hfl_split2 converts an fp32 matrix, hfl_layer_norm_split2 and the attention
 * kernel (out_split = 2) write it directly.  in_features % 32 == 0, out_features % 128 == 0.
 *   gelu_split_out = 0: out is float (n_rows, out_features) = acc + bias [+ residual]  (residual may alias out)
 *   gelu_split_out = 1: out is split2 bf16 (n_rows, out_features/32, 2, 32) of gelu(acc + bias), exact erf GELU */
int hfl_linear_x3(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                  const float* residual, int64_t n_rows, int in_features, int out_features, int gelu_split_out,
                  hfl_stream_t stream);
int hfl_split2(uint16_t* out, const float* x, int64_t n_rows, int64_t channels, hfl_stream_t stream);
/* A logical (n_rows, C) f32 matrix whose rows live in up to four separate arrays: rows [row0[i], row0[i + 1]) at ptr[i] (C floats
 * per row, row-major), row0[0] = 0, 1 <= n <= 4.  The [tokens | relay rows] buffer of an H-OSA block, whose relay rows arrive
 * from the relay-token block, and the relay-token block's input, the concatenation of the pyramid levels' relay rows
 * (models/hotformerloc_backbone.py:593-633: `torch.cat` / index assignment there), are read through it where they are. */
typedef struct hfl_row_segments {
  int32_t n;
  const float* ptr[4];
  int64_t row0[4];
} hfl_row_segments;
/* hfl_linear_x3 (f32 output) with the residual rows read through a segment table. */
int hfl_linear_x3_seg(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                      const hfl_row_segments* residual, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream);
/* Grouped form: ONE launch over row tiles that use DIFFERENT weight blocks -- the per-tap products of an octree convolution
 * over its live (row, tap) pairs (models/layers/octformer_layers.py:89-95; model.OctreeConv._forward_live_taps), which were
 * 27 library GEMM launches.  tiles (n_tiles, 3) int32 = {first row, rows (1..128), first row of the tile's weight block in
 * w_split2}; out (n_rows, out_features) f32 = x W_block^T for the rows of every tile (rows outside all tiles are not
 * written).  out_features % 128 == 0, or out_features == 64 with every weight block padded to 128 rows (zeros). */
int hfl_linear_x3_grouped(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const int32_t* tiles,
                          int64_t n_tiles, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream);
/* The same with the octree convolution's gather done by the GEMM's tile loader: row m of the A operand is
 * x_split2[gather[m]] -- x_split2 (n_src_rows, 2 K) bf16 is the split2 form of the convolution's INPUT rows, gather (n_rows)
 * int32 the input row of every live (row, tap) pair (src of hfl_tap_lists).  Replaces hfl_octree_gather + hfl_linear_x3_grouped:
 * the gathered (pairs x Cin) matrix never exists in memory.  n_src_rows * K * 4 < 2^32. */
int hfl_linear_x3_grouped_gather(float* out, const uint16_t* x_split2, const int32_t* gather, int64_t n_src_rows,
                                 const uint16_t* w_split2, const int32_t* tiles, int64_t n_tiles, int64_t n_rows,
                                 int in_features, int out_features, hfl_stream_t stream);
/* Per-row scaled forms for per-cloud stochastic depth (OctreeDropPath, models/layers/octformer_layers.py:213-289) inside the
 * fused residual branches: out = (x W^T + bias) * row_scale[m] + residual, and split2(x * row_scale[row]) for the branch's
 * incoming gradient.  row_scale (n_rows) may be NULL (= 1). */
int hfl_linear_x3_rows(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                       const float* residual, const float* row_scale, int64_t n_rows, int in_features, int out_features,
                       hfl_stream_t stream);
int hfl_split2_rows(uint16_t* out, const float* x, const float* row_scale, int64_t n_rows, int64_t channels,
                    hfl_stream_t stream);
/* Training forms of the MLP's first Linear (models/layers/octformer_layers.py:53-59 under autograd):
 *   _gelu_fwd: out_split2 = split2(gelu(x W^T + b)) AND preact (n_rows, out_features) f32 = x W^T + b in one launch (GELU's
 *              backward needs the pre-activation);
 *   _gelu_bwd: out_split2 = split2((dy W) * gelu'(preact)): the input gradient of fc2 multiplied by the GELU derivative and
 *              written as the operand of fc1's gradient GEMMs -- wt_split2 is the split2 layout of W^T (in_features x
 *              out_features of the forward = (out_features, in_features) here). */
int hfl_linear_x3_gelu_fwd(uint16_t* out_split2, float* preact, const uint16_t* x_split2, const uint16_t* w_split2,
                           const float* bias, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream);
int hfl_linear_x3_gelu_bwd(uint16_t* out_split2, const uint16_t* dy_split2, const uint16_t* wt_split2,
                           const float* preact, int64_t n_rows, int in_features, int out_features,
                           hfl_stream_t stream);
/* Weight and bias gradient of the same Linear (csrc/wgrad_x3.hip): dw (N,K) = dy^T x, db (N) = column sums of dy, both
 * operands in the split2 layout (dy (n_rows, N) and x (n_rows, K)), three-term products, fp32 accumulation, slabs of rows
 * reduced in a fixed order (bitwise reproducible).  Replaces autograd's fp32 GEMM + bias reduction for torch.nn.Linear in
 * loss.backward() (training/trainer.py:335-352).  N % 128 == 0, K % 128 == 0; db may be NULL;
 * workspace >= hfl_wgrad_x3_workspace(n_rows, N, K) bytes. */
int64_t hfl_wgrad_x3_workspace(int64_t n_rows, int64_t out_features, int64_t in_features);
int hfl_wgrad_x3(float* dw, float* db, const uint16_t* dy_split2, const uint16_t* x_split2, int64_t n_rows,
                 int64_t out_features, int64_t in_features, void* workspace, hfl_stream_t stream);
/* The qkv projection written straight into the operand layout of the fp16-MFMA window attention kernel
 * (hfl_window_attention_fwd_ex with flag 0x100): out (n_rows, out_features) 4-byte cells, every row = [Q | K | V] regions
 * of C = out_features / 3 features, per head 16 dims stored as [16 x hi | 16 x lo] fp16 (hi = RTZ(v), lo = RTZ(v - hi):
 * 22 significant bits) = 64 B, v = acc + bias, the queries multiplied by q_scale (softmax scale * log2 e).
 * Replaces qkv = Linear(x) + the reshape/permute of models/octformer_backbone.py:70-73.  C % 128 == 0. */
int hfl_linear_x3_qkv(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                      int64_t n_rows, int in_features, int out_features, float q_scale, hfl_stream_t stream);
/* split2(LayerNorm(x)) in one pass (the LayerNorm in front of qkv / fc1: models/octformer_backbone.py:275-278) */
int hfl_layer_norm_split2(uint16_t* out, const float* x, const float* gamma, const float* beta,
                          int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream);
/* ReLU(LayerNorm(x)) in one pass: f32 rows (out_f32) or the split2 operand of the next GEMM (out_split2); exactly one of
 * the two is non-NULL.  Replaces norm -> relu behind every stem convolution (models/layers/octformer_layers.py:80-98). */
int hfl_layer_norm_relu(float* out_f32, uint16_t* out_split2, const float* x, const float* gamma, const float* beta,
                        int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream);

/* 9b'. MATCHED-PRECISION Linear (csrc/gemm_x6.hip): y = x W^T [+ bias] [GELU] [* row_scale] [+ residual], all f32 in memory,
 * with fp32-grade products on the bf16 matrix cores: every operand is the EXACT sum of three bf16 planes (h, m, l) and the six
 * plane products >= 2^-16 are accumulated smallest first in fp32 (what is dropped, m l + l m + l l <= 2^-23 |x w|, is the size of
 * one fp32 rounding of the product).  Replaces torch.nn.Linear (fp32) and the element-wise op after it in the transformer blocks
 * (models/octformer_backbone.py:70,91,275-278; models/layers/octformer_layers.py:53-59; models/hotformerloc_backbone.py:213-216)
 * for callers that want the reference's own arithmetic rather than the 16-bit-operand split of section 9b.
 *   w3 (3, out_features, Kp) bf16 = hfl_linear_x6_pack(w (out_features, in_features) f32), once per parameter;
 *      Kp = hfl_linear_x6_padded_k(in_features) = in_features rounded up to a multiple of 64 (zero padding);
 *   x (n_rows, in_features) f32 as any producer left it (the split happens on the way into LDS);
 *   gelu = 1: out = gelu(acc + bias), exact-erf GELU (residual and row_scale must be NULL);
 *   gelu = 0: out = (acc + bias) [* row_scale[row]] [+ residual] (residual may alias out).
 * in_features % 32 == 0, out_features % 128 == 0. */
int64_t hfl_linear_x6_padded_k(int64_t in_features);
int hfl_linear_x6_pack(uint16_t* w3, const float* w, int64_t out_features, int64_t in_features, hfl_stream_t stream);
int hfl_linear_x6(float* out, const float* x, const uint16_t* w3, const float* bias, const float* residual,
                  const float* row_scale, int64_t n_rows, int in_features, int out_features, int gelu, hfl_stream_t stream);

/* Grouped form of hfl_linear_x6 with the gather done by the tile loader: the per-tap products of an octree convolution over
 * its live (row, tap) pairs (models/layers/octformer_layers.py:89-95: ocnn's octree2col + mm) at matched precision -- the
 * stem / downsample convolutions of the matched-precision step.  As hfl_linear_x3_grouped_gather: tiles (n_tiles, 3) int32 =
 * {first row, rows (1..128), first row of the tile's weight block in w3}; row m of the x operand is x[gather[m]] (x: the
 * convolution's input rows, f32, in_features wide); w3 = hfl_linear_x6_pack of the stacked per-tap blocks W[k]^T, each padded
 * to a multiple of 128 rows (w_rows rows in all); out (n_rows, out_features) f32, out_features % 128 == 0 or == 64. */
int hfl_linear_x6_grouped_gather(float* out, const float* x, const int32_t* gather, const uint16_t* w3, int64_t w_rows,
                                 const int32_t* tiles, int64_t n_tiles, int64_t n_rows, int in_features, int out_features,
                                 hfl_stream_t stream);

/* 9e'. norm1 -> attention.qkv -> window attention of a RELAY-TOKEN block (models/hotformerloc_backbone.py:197-216,
 *      models/octformer_backbone.py:52-93: C = 256, 16 heads, K = 48 tokens + 1 relay token per window, dilation 1) as ONE launch
 *      with specialised waves (csrc/attn_ws.hip: six waves run the qkv GEMM of 96 rows head pair by head pair, six run the
 *      window attention of the pair before from LDS): q, k, v of the token rows never cross HBM.
 *        out_split2 (rows, 2C) bf16 split2: the attention output of the token rows [0, n_tokens) and of the relay rows
 *                   [rt_row0, rt_row0 + n_windows) -- the operand of attention.proj (hfl_linear_x3)
 *        x          (n_tokens, C) f32 token rows (after the CPE); gamma / beta / eps: norm1
 *        qkv_pack   hfl_qkv_fused_pack image of attention.qkv.weight; qkv_bias (3C); q_scale = head_dim^-0.5 * log2(e)
 *        relay_qkv  (n_windows, 3C) fp16 (hi, lo) operand rows of the relay tokens = hfl_ln_qkv_fused over the relay rows
 *        rpe_tables3  hfl_window_rpe_expand(..., f16_operand = 2) of the block's RPE table, or NULL (disable_RPE)
 *      hfl_attn_ws_ok: 1 when the configuration is taken (channels 256, 16 heads, n_relay 1, dilation 1, K = 48, depth <= 7),
 *      else 0 -- the caller then runs hfl_ln_qkv_fused + hfl_window_attention_fwd_ex. */
int hfl_attn_ws_ok(const hfl_window_attn_desc* desc, int channels);
int hfl_attn_ws_fwd(void* out_split2, const float* x, const float* gamma, const float* beta, float eps, const void* qkv_pack,
                    const float* qkv_bias, float q_scale, const void* relay_qkv, const uint32_t* tok_meta,
                    const float* rpe_tables3, const hfl_window_attn_desc* desc, hfl_stream_t stream);

/* 9c. The pre-norm MLP branch of a transformer block as ONE launch (csrc/mlp_fused.hip):
 *       out (n_rows, C) = x + fc2(gelu(fc1(LayerNorm(x; gamma, beta, eps)) + b1)) + b2        fc1: C -> 4C, fc2: 4C -> C
 * Replaces norm2 -> mlp.fc1 -> GELU -> mlp.fc2 -> residual add (models/octformer_backbone.py:275-278,
 * models/hotformerloc_backbone.py:213-216, models/layers/octformer_layers.py:38-59).  Same split-precision arithmetic as
 * hfl_linear_x3 (three-term bf16 products, fp32 accumulation, exact-erf GELU); the 4C-wide hidden activation never leaves the
 * register file.  C in {128, 256}; out must not alias x.
 * `pack` is the weight image hfl_mlp_fused_pack writes ONCE per parameter pair from the fp32 weights w1 (4C, C) and w2 (C, 4C)
 * (hfl_mlp_fused_pack_bytes(C) bytes; 0 = unsupported C): the (hi, lo) bf16 split of both matrices cut into 32-hidden-feature
 * stages in the order the kernel streams them through LDS. */
int64_t hfl_mlp_fused_pack_bytes(int channels);
int hfl_mlp_fused_pack(void* pack, const float* w1, const float* w2, int channels, hfl_stream_t stream);
int hfl_ln_mlp_fused(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                     const float* b1, const float* b2, int64_t n_rows, int channels, hfl_stream_t stream);
/*     The same launch with a workspace of hfl_ln_mlp_fused_workspace(n_rows, C) bytes (0: none needed): rows are dealt to
 *     the workgroups in whole passes (128 rows at C = 256), and the rows left over after the last whole round -- fewer than
 *     one pass per workgroup -- are computed with the HIDDEN dimension split over several workgroups per row set, whose fc2
 *     partial sums go through the workspace and are added in a fixed order (bitwise reproducible).  Without a workspace the
 *     left-over rows cost every workgroup that holds some a whole extra pass of the weight stream. */
int64_t hfl_ln_mlp_fused_workspace(int64_t n_rows, int channels);
int hfl_ln_mlp_fused_ws(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                        const float* b1, const float* b2, int64_t n_rows, int channels, void* workspace,
                        int64_t workspace_bytes, hfl_stream_t stream);
/*     The same launches for a hidden width other than 4C: `hidden` = rows of w1 = columns of w2.  Supported: 4C (C = 128,
 *     256: the transformer blocks, identical to the entry points above) and C = hidden = 256 (the Mixer layers of the pooling
 *     head, models/layers/salsa.py:58-75: LayerNorm -> Linear -> GELU -> Linear -> residual with mlp_ratio 1). */
int64_t hfl_mlp_fused_pack_bytes_h(int channels, int hidden);
int hfl_mlp_fused_pack_h(void* pack, const float* w1, const float* w2, int channels, int hidden, hfl_stream_t stream);
int64_t hfl_ln_mlp_fused_workspace_h(int64_t n_rows, int channels, int hidden);
int hfl_ln_mlp_fused_h(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                       const float* b1, const float* b2, int64_t n_rows, int channels, int hidden, void* workspace,
                       int64_t workspace_bytes, hfl_stream_t stream);

/* 9d. LayerNorm -> qkv projection as ONE launch, written as the fp16 (hi, lo) operand rows of the window kernel
 *     (= hfl_layer_norm_split2 + hfl_linear_x3_qkv: norm1 -> attention.qkv, models/octformer_backbone.py:70,
 *     models/hotformerloc_backbone.py:213-216; csrc/qkv_fused.hip).  C = 128 or 256.  `pack` = hfl_qkv_fused_pack_bytes(C)
 *     bytes laid out once per parameter by hfl_qkv_fused_pack from the f32 weight (3C, C); bias (3C); qkv_out (n_rows, 3C)
 *     4 B per element; q_scale (softmax scale x log2 e) is folded into the queries. */
int64_t hfl_qkv_fused_pack_bytes(int channels);
int hfl_qkv_fused_pack(void* pack, const float* w_qkv, int channels, hfl_stream_t stream);
int hfl_ln_qkv_fused(void* qkv_out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                     const float* bias, float q_scale, int64_t n_rows, int channels, hfl_stream_t stream);
/* ... with the input rows read through a segment table (hfl_row_segments above). */
int hfl_ln_qkv_fused_seg(void* qkv_out, const hfl_row_segments* x, const float* gamma, const float* beta, float eps,
                         const void* pack, const float* bias, float q_scale, int64_t n_rows, int channels, hfl_stream_t stream);

/* 9e. LayerNorm -> qkv projection -> window attention as ONE launch (csrc/attn_fused.hip): norm1 -> attention.qkv -> mask /
 *     RPE bias / SDPA of `x = x + attn(norm1(x))` (models/octformer_backbone.py:52-93,275-276) up to the attention output,
 *     q, k, v never in HBM.  Built for the OctFormer stage: C = 128 (8 heads of 16), patch_size 48, no relay tokens,
 *     dilation 1 / 2 / 4, octree depth <= 7, RPE through the three clamped 1-D expanded tables
 *     (hfl_window_rpe_expand with f16_operand = 1 at a depth where it builds that form) or none.  hfl_attn_fused_ok says
 *     whether a configuration is taken.  x (n_tokens, C) f32; qkv_pack = hfl_qkv_fused_pack image; out_split2 (n_tokens, 2 C)
 *     bf16 = the operand of the proj GEMM (hfl_linear_x3), bitwise what hfl_ln_qkv_fused + hfl_window_attention_fwd_ex
 *     (out_split3 = 2 | 0x100) produce. */
int hfl_attn_fused_ok(const hfl_window_attn_desc* desc, int channels, int has_rpe);
int hfl_attn_fused_fwd(void* out_split2, const float* x, const float* gamma, const float* beta, float eps, const void* qkv_pack,
                       const float* qkv_bias, float q_scale, const uint32_t* tok_meta, const float* rpe_table,
                       const hfl_window_attn_desc* desc, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 10. Backward kernels (training path; autograd glue in hotformerloc_amd/autograd.py).
 *     The reference gets these from PyTorch autograd over its materialised formulation and from
 *     dwconv.cu for the depth-wise conv (section 1 already covers dwconv's two gradients).
 * ---------------------------------------------------------------------- */
/* Gradient of hfl_window_attention_fwd.  dout (rows, H*16) -> dqkv (rows, 3*H*16), fully written for
 * every token and relay row; drpe_table (3*(2*pos_bnd+1), H) is ACCUMULATED into (zero it first; float
 * atomics, so its low bits are not run-to-run reproducible) and may be NULL when rpe_table is NULL. */
int hfl_window_attention_bwd(float* dqkv, float* drpe_table, const float* qkv, const float* dout,
                             const uint32_t* tok_meta, const float* rpe_table,
                             const hfl_window_attn_desc* desc, hfl_stream_t stream);
/* The same with a REPRODUCIBLE RPE-table gradient: every grid column writes its partial table to `workspace`
 * (hfl_window_attention_bwd_workspace(desc) bytes; 0 = this launch configuration has no deterministic form, use the entry
 * above) and a second launch adds the partials in a fixed order -- no float atomics; drpe_table need not be zeroed. */
int64_t hfl_window_attention_bwd_workspace(const hfl_window_attn_desc* desc);
int hfl_window_attention_bwd_det(float* dqkv, float* drpe_table, const float* qkv, const float* dout,
                                 const uint32_t* tok_meta, const float* rpe_table, const hfl_window_attn_desc* desc,
                                 void* workspace, hfl_stream_t stream);
/* (All three backward entry points, round 6: with desc->depth in 1..5 -- token coordinates < 32 -- the RPE-table gradient
 * of models/layers/octformer_layers.py:144-174 is computed on the matrix cores as the diagonal sums of OHQ^T dS OHK, one-hot
 * coordinate matrices of the window, summed over every window of a wave and reduced once in a fixed order; deeper levels or
 * depth 0 = unknown keep the fixed-point LDS scatter-add.  The grid is the resident workgroup count of the instantiation.)
 * The same (workspace = NULL: the float-atomic table gradient of hfl_window_attention_bwd) with dqkv written as the split2
 * operand (rows, 2 * 3 H 16) bf16 of the qkv layer's data- and weight-gradient GEMMs (round 6): bit for bit what hfl_split2
 * makes of the f32 gradient, which is never in memory (the training step spent 2 ms in that pass). */
int hfl_window_attention_bwd_split2(uint16_t* dqkv_split2, float* drpe_table, const float* qkv, const float* dout,
                                    const uint32_t* tok_meta, const float* rpe_table, const hfl_window_attn_desc* desc,
                                    void* workspace, hfl_stream_t stream);
/* Gradient of hfl_relay_attention_fwd.  dqkv (rows, 3*H*16): rows listed in seq_rows are written, the
 * others left untouched (the caller zero-fills).  max_seq_len * 268 B of LDS per workgroup: returns
 * HFL_ECAPACITY beyond 611 relay tokens per cloud. */
int hfl_relay_attention_bwd(float* dqkv, const float* qkv, const float* dout, const int32_t* seq_rows,
                            const int32_t* seq_off, int batch, int n_heads, float scale,
                            int max_seq_len, hfl_stream_t stream);
/* ----------------------------------------------------------------------
 * 12. A whole transformer block of the inference path in one call (csrc/capi.hip): OctFormerBlock / HOTFormerBlock forward
 *     (models/octformer_backbone.py:232-281, models/hotformerloc_backbone.py:130-236 with layer scale and stochastic depth
 *     off) = CPE -> [relay rows] -> LN1 -> qkv -> window attention -> proj + residual -> LN2 -> fc1 + GELU -> fc2 +
 *     residual, on the entry points above (split-precision GEMMs, fp16 (hi, lo) attention operands: the caller checks
 *     hfl_window_attention_f16_ok first).  Weights in the layouts those entry points take; `rpe_table` may be NULL.
 *     io->x_in: (n_rows, C) input buffer [tokens | relay rows]; io->relay: optional fresh relay rows (n_rows - n_tokens, C)
 *     that replace x_in's; io->out (n_rows, C); io->arena >= hfl_block_forward_x3_arena(n_rows, C) bytes of scratch.
 * ---------------------------------------------------------------------- */
typedef struct hfl_block_weights {
  int64_t channels;
  float eps, q_scale;
  const float *cpe_weight, *cpe_gamma, *cpe_beta;
  const float *norm1_gamma, *norm1_beta, *norm2_gamma, *norm2_beta;
  const uint16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;     /* split2 */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  const float* rpe_table;                              /* (3 (2 pos_bnd + 1), H) or NULL */
  const void* mlp_pack;                                /* hfl_mlp_fused_pack image of (fc1, fc2) or NULL: when set, LN2 -> fc1 ->
                                                          GELU -> fc2 -> residual run as ONE launch (hfl_ln_mlp_fused) and fc1_w /
                                                          fc2_w are not read */
  const void* qkv_pack;                                /* hfl_qkv_fused_pack image of qkv_w or NULL: when set, phase 1 runs LN1 -> qkv
                                                          of the token rows as ONE launch (hfl_ln_qkv_fused); qkv_w is still
                                                          read for the relay rows */
  int32_t fuse_attention;                              /* bit 0: a whole-block call (phase 0) of a block WITHOUT relay rows runs LN1
                                                          -> qkv -> window attention as ONE launch (hfl_attn_fused_fwd) when
                                                          hfl_attn_fused_ok takes the configuration; needs qkv_pack.
                                                          bit 1: a block WITH relay rows runs LN1 -> qkv -> window attention of its
                                                          token rows as ONE launch (hfl_attn_ws_fwd) when hfl_attn_ws_ok takes the
                                                          configuration; needs qkv_pack and rpe_tables3.  Phase 1 is then the CPE
                                                          alone (the launch reads the relay rows' q / k / v).
                                                          bit 2: copy the relay rows into the block's buffer first (the form before
                                                          proj's residual read them in place; for A/B runs) */
  const float* rpe_tables3;                            /* hfl_window_rpe_expand(..., f16_operand = 2) of rpe_table, or NULL */
} hfl_block_weights;
typedef struct hfl_block_io {
  const float* x_in;
  const float* relay;                                  /* also in phase 4: proj's residual reads the relay rows there */
  float* out;
  void* arena;
  const int32_t* neigh;                                /* (n_tokens, 27) */
  const uint32_t* tok_meta;
  int64_t n_rows, n_tokens;
  int32_t phase;                                       /* 0: the whole block.  1: only what does not depend on the relay rows --
                                                          CPE, LN1 and the qkv projection of the TOKEN rows (may run while the
                                                          relay-token self-attention of the iteration is still in flight);
                                                          2: the rest -- relay rows in, their LN1 / qkv, window attention, proj,
                                                          MLP.  3 + 4 split phase 2 around the attention: 3 = relay rows in and
                                                          their LN1 / qkv, 4 = proj and MLP; between them the caller runs the
                                                          attention of this and other blocks with hfl_block_attention_x3_multi.
                                                          All phases of a block share `arena`. */
} hfl_block_io;
int64_t hfl_block_forward_x3_arena(int64_t n_rows, int64_t channels);
int hfl_block_forward_x3(const hfl_block_weights* w, const hfl_block_io* io, const hfl_window_attn_desc* desc,
                         hfl_stream_t stream);
/* The window attention of n (<= 4) blocks that have finished phases 1 and 3, in ONE launch when they share an attention shape
 * (patch size, relay tokens, heads, table form: the pyramid levels of an H-OSA iteration,
 * models/hotformerloc_backbone.py:466-473), else one launch each; then run each block's phase 4. */
int hfl_block_attention_x3_multi(int n, const hfl_block_weights* const* w, const hfl_block_io* const* io,
                                 const hfl_window_attn_desc* const* desc, hfl_stream_t stream);

/* The relay-token transformer block (RTSA, models/hotformerloc_backbone.py:239-302) of the inference path as ONE call: LN1 ->
 * split2, qkv GEMM, ragged relay attention (hfl_relay_attention_fwd), split2, proj GEMM + residual, LN2 -> split2, fc1 GEMM +
 * GELU, fc2 GEMM + residual -- eight launches and a memset issued back to back from native code (the rows are the few
 * thousand relay tokens of a batch: the launches are tiny, the Python issue time was the cost).
 * arena >= hfl_relay_block_forward_x3_arena(n_rows, channels) bytes; out (n_rows, C) must not alias x_in. */
typedef struct hfl_relay_block_weights {
  int64_t channels;
  int32_t n_heads;
  float eps;
  const float *norm1_gamma, *norm1_beta, *norm2_gamma, *norm2_beta;
  const uint16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;     /* split2 */
  const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
  const void* mlp_pack;                                /* hfl_mlp_fused_pack image of (fc1, fc2) or NULL: when set (C = 128 / 256) the MLP
                                                          branch is ONE launch (hfl_ln_mlp_fused_ws) and fc1_w / fc2_w are not read */
  const void* qkv_pack;                                /* hfl_qkv_fused_pack image of qkv_w or NULL: when set, LN1 -> qkv is ONE launch
                                                          (hfl_ln_qkv_fused) and the attention reads its fp16 (hi, lo) rows and
                                                          writes proj's split2 operand itself (hfl_relay_attention_f16_fwd): three
                                                          launches up to proj's input instead of six */
} hfl_relay_block_weights;
typedef struct hfl_relay_block_io {
  const float* x_in;
  float* out;
  void* arena;
  const int32_t* seq_rows;                             /* as hfl_relay_attention_fwd */
  const int32_t* seq_off;
  int64_t n_rows;
  int32_t batch, max_seq_len;
  const int32_t* orphan_rows;                          /* rows that belong to no sequence (relay tokens of pure padding windows):
                                                          their attention output is zero; read when qkv_pack is set */
  int32_t n_orphans;
  const hfl_row_segments* x_segments;                  /* optional, needs qkv_pack: the input rows where the pyramid levels left
                                                          them (no concatenation launch); x_in is then not read */
} hfl_relay_block_io;
int64_t hfl_relay_block_forward_x3_arena(int64_t n_rows, int64_t channels);
int hfl_relay_block_forward_x3(const hfl_relay_block_weights* w, const hfl_relay_block_io* io, hfl_stream_t stream);

/* Weight gradient of an octree convolution over its live (row, tap) pairs (csrc/tapconv.hip; replaces autograd over
 * ocnn's octree2col + mm, models/layers/octformer_layers.py:89-95): dw[k] (cin, cout) = g_k^T dpart_k over the pairs of tap
 * k.  g (P, cin), dpart (P, cout) fp32 pair-major; chunks (n_chunks, 3) int32 = {tap, first pair, end pair}, ascending, no
 * chunk straddles a tap; tap_chunk_off (taps + 1) int32 = first chunk of every tap; workspace n_chunks * cin * cout floats.
 * cin % 64 == 0, cout % 64 == 0.  fp32 MFMA, fixed summation order. */
int hfl_tap_wgrad(float* dw, const float* g, const float* dpart, const int32_t* chunks, int n_chunks,
                  const int32_t* tap_chunk_off, int taps, int cin, int cout, float* workspace, hfl_stream_t stream);
/* The same with the pair-major operands read through row tables (round 6): pair p contracts row g_rows[p] of g (the layer
 * input, (n_src, cin)) with row d_rows[p] of dpart (the output gradient, (n_out, cout)); a NULL table = pair-major operand as
 * above.  With both tables neither octree2col copy of the backward is written (ocnn materialises both). */
int hfl_tap_wgrad_gather(float* dw, const float* g, const int32_t* g_rows, const float* dpart, const int32_t* d_rows,
                         const int32_t* chunks, int n_chunks, const int32_t* tap_chunk_off, int taps, int cin, int cout,
                         float* workspace, hfl_stream_t stream);
/* Inverse of a gather table whose source and destination row counts differ (stride-2 conv:
 * table (n_dst, K) with entries in [0, n_src)): inverse (n_src, K), inverse[table[m,k], k] = m, -1 else. */
int hfl_inverse_table(int32_t* inverse, int64_t n_src_rows, const int32_t* table, int64_t n_dst_rows,
                      int kngh, hfl_stream_t stream);
/* Gradient of hfl_octree_gather w.r.t. data through the INVERSE table (hfl_inverse_neigh of the
 * forward table, int32): ddata[n,c] = sum_k dcol[ineigh[n,k], k*C + c]. */
int hfl_octree_gather_bwd(float* ddata, const float* dcol, const int32_t* ineigh, int64_t n_rows,
                          int kngh, int64_t channels, hfl_stream_t stream);
/* Gradient of hfl_relay_token_init: dx (N_t,C) fully written. */
int hfl_relay_token_init_bwd(float* dx, const float* drt, const uint32_t* tok_meta, int64_t n_tokens,
                             int32_t n_windows, int32_t patch_size, int64_t channels,
                             hfl_stream_t stream);

/* Gradient of LayerNorm over the channel axis (training path; replaces torch's native_layer_norm_backward behind
 * the LayerNorms of models/octformer_backbone.py:275-278 etc.): dx (n_rows, C); dgamma_partial / dbeta_partial
 * (hfl_layer_norm_bwd_blocks(n_rows, C), C) per-workgroup partial sums the caller adds up (fixed order, no atomics).
 * Statistics are recomputed from x. */
int hfl_layer_norm_bwd_blocks(int64_t n_rows, int64_t channels);
int hfl_layer_norm_bwd(float* dx, float* dgamma_partial, float* dbeta_partial, const float* dy, const float* x,
                       const float* gamma, int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream);
/* The same with dx = dres + (LayerNorm input gradient): the skip path of a pre-norm residual branch y = x + f(LN(x))
 * (models/octformer_backbone.py:275-278) joins inside the kernel; dres (n_rows, channels) may be NULL and may alias dx. */
int hfl_layer_norm_bwd_add(float* dx, float* dgamma_partial, float* dbeta_partial, const float* dy, const float* x,
                           const float* gamma, const float* dres, int64_t n_rows, int64_t channels, float eps,
                           hfl_stream_t stream);
int hfl_layer_norm_bwd_finalize(float* dgamma, float* dbeta, const float* dgamma_partial, const float* dbeta_partial,
                                int n_blocks, int64_t channels, hfl_stream_t stream);

/* ------------------------------------------------------------------------
 * 11. TruncatedSmoothAP ranking core (replaces the (B,P,B) tensor algebra of
 *     models/losses/truncated_smoothap.py:44-93 and its autograd graph)
 * ---------------------------------------------------------------------- */
/* sim (B,B) f32 = E E^T, pos_mask / neg_mask (B,B) uint8 (0/1), closest_pos (B,P) int64 = top-P positives of
 * every query by similarity (entries that are not positives are skipped, as the reference zeroes them,
 * :85-87).  ap (B): sum_j valid r_j / n_valid (0 for a query without positives); dap_ds (B,B): d ap[q] / d sim[q,z]
 * with the clamped-sigmoid gradient of loss_utils.py:40-48.  HFL_ECAPACITY beyond 17066 rows (9 B of LDS each). */
int hfl_smoothap_rows(float* ap, float* dap_ds, const float* sim, const uint8_t* pos_mask,
                      const uint8_t* neg_mask, const int64_t* closest_pos, int batch, int positives_per_query,
                      float tau, hfl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* HOTFORMERLOC_HIP_H */
