// Library identification entry points of libhotformerloc_hip.so.
#include "hfl_common.h"


extern "C" {

// CUs a launch sizes its persistent grid for (every launcher asks through hfl_stream_cus): the device's.
int hfl_internal_stream_cus(void* stream) {
  (void)stream;
  return hfl_num_cus();
}

int hfl_version(void) { return 100; }   // 1.00

const char* hfl_arch(void) { return "gfx950"; }

// upper bound of hfl_ln_mlp_fused_workspace over every row count <= n_rows (the executor launches the fused MLP on all rows
// or on the token rows alone): parts x left-over rows <= min(16 n_rows, one round of the whole chip)
static int64_t mlp_ws_bound(int64_t n_rows, int64_t channels) {
  const int64_t round = (int64_t)hfl_num_cus() * (channels == 256 ? 8 : 16) * 16;
  const int64_t r = 16 * n_rows < round ? 16 * n_rows : round;
  return r * channels * 4;
}

// One transformer block of the inference path as ONE call: the launches a block makes (CPE, relay-row copy, LN1 -> split2,
// qkv GEMM into the fp16 attention operand, window attention, proj GEMM + residual, then either LN2 -> split2, fc1 GEMM + GELU,
// fc2 GEMM + residual or the fused MLP launch) issued back to back from native code.  Nothing new runs on the GPU; what goes away is eight Python launch wrappers
// per block (~22 us of host time each: the fresh-batch path is host-bound).  See include/hotformerloc_hip.h.
int hfl_block_forward_x3(const hfl_block_weights* w, const hfl_block_io* io, const hfl_window_attn_desc* desc,
                         hfl_stream_t stream) {
  if (w == nullptr || io == nullptr || desc == nullptr) return HFL_EINVAL;
  const int64_t C = w->channels, rows = io->n_rows, nt = io->n_tokens;
  if (C <= 0 || C % 128 != 0 || rows < nt || nt < 0) return HFL_EINVAL;
  if (rows == 0) return HFL_OK;
  // arena carve (16-B aligned: C % 128 == 0): x0 f32 | a2 split2 | qkv | o2 split2 | x1 f32 | h2 split2 | g2 split2 (4C)
  unsigned char* a = static_cast<unsigned char*>(io->arena);
  const size_t unit = (size_t)rows * C * 4;
  float* x0 = reinterpret_cast<float*>(a);
  uint16_t* a2 = reinterpret_cast<uint16_t*>(a + unit);
  float* qkv = reinterpret_cast<float*>(a + 2 * unit);
  uint16_t* o2 = reinterpret_cast<uint16_t*>(a + 5 * unit);
  float* x1 = reinterpret_cast<float*>(a + 6 * unit);
  uint16_t* h2 = reinterpret_cast<uint16_t*>(a + 7 * unit);
  uint16_t* g2 = reinterpret_cast<uint16_t*>(a + 8 * unit);
  int rc;
  const int phase = io->phase;
  if (phase < 0 || phase > 4) return HFL_EINVAL;
  // relay-token blocks of the shape csrc/attn_ws.hip takes: LN1 -> qkv -> window attention of the token rows are ONE launch
  // behind the relay rows' qkv (their q / k / v are operands of every window), so phase 1 is the CPE alone.  (Phase 3 -- the
  // attention goes out in a merged launch of several blocks -- keeps the separate qkv launch.)
  const bool ws = (w->fuse_attention & 2) != 0 && rows > nt && nt > 0 && w->qkv_pack != nullptr && w->rpe_tables3 != nullptr &&
                  w->rpe_table != nullptr && hfl_attn_ws_ok(desc, (int)C) != 0;
  // blocks without relay rows (the OctFormer stage), whole-block call: LN1 -> qkv -> window attention as ONE launch, q / k / v
  // never in HBM (csrc/attn_fused.hip), when the configuration is one it takes
  const bool fused_attn = phase == 0 && rows == nt && (w->fuse_attention & 1) != 0 && w->qkv_pack != nullptr &&
                          hfl_attn_fused_ok(desc, (int)C, w->rpe_table != nullptr) != 0;
  // The relay rows (what the relay-token block of the iteration produced, or x_in's) are read WHERE THEY ARE when the fused
  // LN1 -> qkv launch is available: by that launch and by proj's residual (a two-segment row table); the block's own buffer
  // holds them only on the unfused path, whose LayerNorm reads it.
  const float* relay_src = io->relay != nullptr ? io->relay : io->x_in + nt * C;
  const bool relay_in_place = rows > nt && w->qkv_pack != nullptr && (w->fuse_attention & 4) == 0;
  const bool part1 = phase <= 1, part3 = phase == 0 || phase == 2 || phase == 3, part_att = phase == 0 || phase == 2;
  // (what follows the attention -- proj, MLP -- runs in phases 0, 2 and 4: every phase that gets past the two returns below)
  auto ln_qkv = [&](int64_t r0, int64_t nr, const float* src) -> int {       // rows [r0, r0 + nr) of qkv from src (nr x C)
    if (nr <= 0) return HFL_OK;
    if (w->qkv_pack != nullptr)
      return hfl_ln_qkv_fused(qkv + r0 * 3 * C, src, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b, w->q_scale, nr,
                              (int)C, stream);
    int r = hfl_layer_norm_split2(a2 + r0 * 2 * C, src, w->norm1_gamma, w->norm1_beta, nr, C, w->eps, stream);
    if (r != HFL_OK) return r;
    return hfl_linear_x3_qkv(qkv + r0 * 3 * C, a2 + r0 * 2 * C, w->qkv_w, w->qkv_b, nr, (int)C, (int)(3 * C), w->q_scale, stream);
  };
  // ---- phase 1: everything that reads TOKEN rows only (independent of this iteration's relay-token self-attention): the CPE
  // and, per-row operators that they are, the token rows' LN1 and qkv
  if (part1 && nt > 0) {
    rc = hfl_cpe_forward(x0, io->x_in, w->cpe_weight, w->cpe_gamma, w->cpe_beta, io->neigh, nt, C, 27, w->eps, 1, stream);
    if (rc != HFL_OK) return rc;
    if (!ws && !fused_attn) {
      rc = ln_qkv(0, nt, x0);
      if (rc != HFL_OK) return rc;
    }
  }
  if (phase == 1) return HFL_OK;
  // ---- phase 2 (or the rest of the whole block) = phase 3 (the relay rows up to their qkv) + the window attention + phase 4
  // (proj, MLP); 3 and 4 exist so that the attention of several blocks can go out as one launch in between
  // (hfl_block_attention_x3_multi).  The relay rows' qkv goes out FIRST: the window attention waits for it.
  if (part3 && rows > nt) {
    if (!relay_in_place) {
      hipError_t e = hipMemcpyAsync(x0 + nt * C, relay_src, (size_t)(rows - nt) * C * 4, hipMemcpyDeviceToDevice,
                                    static_cast<hipStream_t>(stream));
      if (e != hipSuccess) return (int)e;
    }
    rc = ln_qkv(nt, rows - nt, relay_in_place ? relay_src : x0 + nt * C);
    if (rc != HFL_OK) return rc;
    if (ws && phase == 3) {
      rc = ln_qkv(0, nt, x0);
      if (rc != HFL_OK) return rc;
    }
  }
  if (phase == 3) return HFL_OK;
  if (part_att) {
    if (fused_attn) {
      rc = hfl_attn_fused_fwd(o2, x0, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b, w->q_scale, io->tok_meta,
                              w->rpe_table, desc, stream);
    } else if (ws) {
      rc = hfl_attn_ws_fwd(o2, x0, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b, w->q_scale, qkv + nt * 3 * C,
                           io->tok_meta, w->rpe_tables3, desc, stream);
    } else {
      rc = hfl_window_attention_fwd_ex(o2, qkv, nullptr, io->tok_meta, w->rpe_table, desc, 2 | 0x100, stream);
    }
    if (rc != HFL_OK) return rc;
  }
  if (relay_in_place) {
    hfl_row_segments res;
    res.n = 2;
    res.ptr[0] = x0; res.row0[0] = 0;
    res.ptr[1] = relay_src; res.row0[1] = nt;
    res.ptr[2] = res.ptr[3] = nullptr; res.row0[2] = res.row0[3] = 0;
    if (nt == 0) { res.n = 1; res.ptr[0] = relay_src; }
    rc = hfl_linear_x3_seg(x1, o2, w->proj_w, w->proj_b, &res, rows, (int)C, (int)C, stream);
  } else {
    rc = hfl_linear_x3(x1, o2, w->proj_w, w->proj_b, x0, rows, (int)C, (int)C, 0, stream);
  }
  if (rc != HFL_OK) return rc;
  if (w->mlp_pack != nullptr)       // the MLP branch in one launch: the 4C-wide hidden activation stays in registers
    // (workspace of the left-over rows' partial sums: behind the twelve units, sized by hfl_block_forward_x3_arena)
    return hfl_ln_mlp_fused_ws(io->out, x1, w->norm2_gamma, w->norm2_beta, w->eps, w->mlp_pack, w->fc1_b, w->fc2_b, rows, (int)C,
                               a + 12 * unit, mlp_ws_bound(rows, C), stream);
  rc = hfl_layer_norm_split2(h2, x1, w->norm2_gamma, w->norm2_beta, rows, C, w->eps, stream);
  if (rc != HFL_OK) return rc;
  rc = hfl_linear_x3(g2, h2, w->fc1_w, w->fc1_b, nullptr, rows, (int)C, (int)(4 * C), 1, stream);
  if (rc != HFL_OK) return rc;
  return hfl_linear_x3(io->out, g2, w->fc2_w, w->fc2_b, x1, rows, (int)C * 4, (int)C, 0, stream);
}

int64_t hfl_block_forward_x3_arena(int64_t n_rows, int64_t channels) {
  return n_rows * channels * 4 * 12 + mlp_ws_bound(n_rows, channels);
}

// The window attention of n blocks (between their phases 3 and 4) as ONE launch when the blocks have one attention shape
// (the pyramid levels of an H-OSA iteration), else one launch each: operands are found in the blocks' arenas.
int hfl_block_attention_x3_multi(int n, const hfl_block_weights* const* w, const hfl_block_io* const* io,
                                 const hfl_window_attn_desc* const* desc, hfl_stream_t stream) {
  if (n < 1 || n > 4 || w == nullptr || io == nullptr || desc == nullptr) return HFL_EINVAL;
  void* out[4];
  const float* qkv[4];
  const uint32_t* meta[4];
  const float* table[4];
  for (int i = 0; i < n; ++i) {
    if (w[i] == nullptr || io[i] == nullptr || desc[i] == nullptr) return HFL_EINVAL;
    const int64_t C = w[i]->channels, rows = io[i]->n_rows;
    if (C <= 0 || C % 128 != 0 || rows < io[i]->n_tokens) return HFL_EINVAL;
    unsigned char* a = static_cast<unsigned char*>(io[i]->arena);
    const size_t unit = (size_t)rows * C * 4;
    qkv[i] = reinterpret_cast<const float*>(a + 2 * unit);
    out[i] = a + 5 * unit;
    meta[i] = io[i]->tok_meta;
    table[i] = w[i]->rpe_table;
  }
  return hfl_window_attention_fwd_multi(n, out, qkv, meta, table, desc, 2 | 0x100, stream);
}

// The relay-token transformer block as one call (see include/hotformerloc_hip.h): the same launches, in the same order, as
// model.RelayTokenTransformerBlock issues through the Python wrappers.
int hfl_relay_block_forward_x3(const hfl_relay_block_weights* w, const hfl_relay_block_io* io, hfl_stream_t stream) {
  if (w == nullptr || io == nullptr) return HFL_EINVAL;
  const int64_t C = w->channels, rows = io->n_rows;
  if (C <= 0 || C % 128 != 0 || rows < 0 || w->n_heads * 16 != C) return HFL_EINVAL;
  if (rows == 0) return HFL_OK;
  // arena carve: a2 split2 | qkv f32 (3C) | att f32 | o2 split2 | x1 f32 | h2 split2 | g2 split2 (4C)
  unsigned char* a = static_cast<unsigned char*>(io->arena);
  const size_t unit = (size_t)rows * C * 4;
  uint16_t* a2 = reinterpret_cast<uint16_t*>(a);
  float* qkv = reinterpret_cast<float*>(a + unit);
  float* att = reinterpret_cast<float*>(a + 4 * unit);
  uint16_t* o2 = reinterpret_cast<uint16_t*>(a + 5 * unit);
  float* x1 = reinterpret_cast<float*>(a + 6 * unit);
  uint16_t* h2 = reinterpret_cast<uint16_t*>(a + 7 * unit);
  uint16_t* g2 = reinterpret_cast<uint16_t*>(a + 8 * unit);
  int rc;
  const hfl_row_segments* xs = io->x_segments;
  if (xs != nullptr && w->qkv_pack == nullptr) return HFL_EINVAL;
  if (xs == nullptr && io->x_in == nullptr) return HFL_EINVAL;
  if (w->qkv_pack != nullptr) {
    // LN1 -> qkv as one launch into the fp16 (hi, lo) operand rows, the attention from them straight into proj's operand
    if (xs != nullptr)          // the pyramid levels' relay rows where the levels left them
      rc = hfl_ln_qkv_fused_seg(qkv, xs, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b,
                                0.25f * 1.4426950408889634f, rows, (int)C, stream);
    else
      rc = hfl_ln_qkv_fused(qkv, io->x_in, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b,
                            0.25f * 1.4426950408889634f, rows, (int)C, stream);
    if (rc != HFL_OK) return rc;
    rc = hfl_relay_attention_f16_fwd(o2, qkv, io->seq_rows, io->seq_off, io->batch, w->n_heads, io->max_seq_len, io->orphan_rows,
                                     io->n_orphans, stream);
    if (rc != HFL_OK) return rc;
  } else {
    rc = hfl_layer_norm_split2(a2, io->x_in, w->norm1_gamma, w->norm1_beta, rows, C, w->eps, stream);
    if (rc != HFL_OK) return rc;
    rc = hfl_linear_x3(qkv, a2, w->qkv_w, w->qkv_b, nullptr, rows, (int)C, (int)(3 * C), 0, stream);
    if (rc != HFL_OK) return rc;
    hipError_t e = hipMemsetAsync(att, 0, unit, static_cast<hipStream_t>(stream));      // rows in no sequence -> 0
    if (e != hipSuccess) return (int)e;
    rc = hfl_relay_attention_fwd(att, qkv, io->seq_rows, io->seq_off, io->batch, w->n_heads, 0.25f, io->max_seq_len, stream);
    if (rc != HFL_OK) return rc;
    rc = hfl_split2(o2, att, rows, C, stream);
    if (rc != HFL_OK) return rc;
  }
  if (xs != nullptr)
    rc = hfl_linear_x3_seg(x1, o2, w->proj_w, w->proj_b, xs, rows, (int)C, (int)C, stream);
  else
    rc = hfl_linear_x3(x1, o2, w->proj_w, w->proj_b, io->x_in, rows, (int)C, (int)C, 0, stream);
  if (rc != HFL_OK) return rc;
  if (w->mlp_pack != nullptr)       // LN2 -> fc1 -> GELU -> fc2 -> residual in one launch, hidden dimension split over the chip
    return hfl_ln_mlp_fused_ws(io->out, x1, w->norm2_gamma, w->norm2_beta, w->eps, w->mlp_pack, w->fc1_b, w->fc2_b, rows, (int)C,
                               a + 12 * unit, hfl_ln_mlp_fused_workspace(rows, (int)C), stream);
  rc = hfl_layer_norm_split2(h2, x1, w->norm2_gamma, w->norm2_beta, rows, C, w->eps, stream);
  if (rc != HFL_OK) return rc;
  rc = hfl_linear_x3(g2, h2, w->fc1_w, w->fc1_b, nullptr, rows, (int)C, (int)(4 * C), 1, stream);
  if (rc != HFL_OK) return rc;
  return hfl_linear_x3(io->out, g2, w->fc2_w, w->fc2_b, x1, rows, (int)(4 * C), (int)C, 0, stream);
}

int64_t hfl_relay_block_forward_x3_arena(int64_t n_rows, int64_t channels) {
  return n_rows * channels * 4 * 12 + hfl_ln_mlp_fused_workspace(n_rows, (int)channels);
}

}  // extern "C"
