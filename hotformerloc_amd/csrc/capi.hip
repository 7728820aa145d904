// Library identification entry points of libhotformerloc_hip.so.
#include "hfl_common.h"

#include <mutex>
#include <utility>
#include <vector>

namespace {
std::mutex g_stream_mu;
std::vector<std::pair<void*, int>> g_stream_cus;       // streams made by hfl_stream_create_cu_mask -> CUs in their mask
int g_cu_reserve = 0;                                  // probe knob 'cu_reserve'
int g_round_launches = 0;                              // probe knob 'round_launches' (measured: -1 % of the step, off)
}  // namespace

extern "C" {

// The chip-filling row-tile launches of a block inside the multi-stream H-OSA schedule (fused LN1 -> qkv of the token rows,
// fused MLP) as one launch per ROUND of the grid (cus x rows per pass) instead of one persistent launch over all rounds.
// While such a kernel runs nothing else fits on any CU (8 waves x 256 VGPRs, 135-150 KB of LDS): the relay-token block's
// launches, on their high-priority stream, waited for its last workgroup -- the relay tokens' fused MLP took 105 us beside the
// finest level's qkv launch against 21 alone (kernel trace), and the finest level then waited ~58 us per iteration for the
// relay rows.  At a launch boundary every CU frees at once and the queue arbiter serves the higher-priority stream first.
// The rows of a round are whole passes (no tail), only the last launch has left-over rows: the same work, cut in time.
// Measured (profiles/r04_ag_ab_round_launches.log): the relay tokens' MLP drops to 63 us, but its reduce then starves beside
// the second round (61 us), every round pays its own ramp (depth-4 MLP 184 + 139 us against 276) and the step loses 1 %:
// 2789-2802 clouds/s against 2820-2829.  Off; `round_launches` = 1 switches it on.
void hfl_internal_set_round_launches(int v) { g_round_launches = v ? 1 : 0; }
int hfl_internal_stream_cus(void* stream);

static int qkv_by_rounds(void* qkv_out, const float* x, const hfl_block_weights* w, int64_t n_rows, int64_t C, int split,
                         hfl_stream_t stream) {
  const int64_t round = (int64_t)hfl_internal_stream_cus(stream) * (C == 256 ? 128 : 256);
  int64_t r0 = 0;
  while (split && g_round_launches && n_rows - r0 >= 2 * round) {      // (the last launch keeps at least one whole round)
    int rc = hfl_ln_qkv_fused(static_cast<float*>(qkv_out) + r0 * 3 * C, x + r0 * C, w->norm1_gamma, w->norm1_beta, w->eps,
                              w->qkv_pack, w->qkv_b, w->q_scale, round, (int)C, stream);
    if (rc != HFL_OK) return rc;
    r0 += round;
  }
  return hfl_ln_qkv_fused(static_cast<float*>(qkv_out) + r0 * 3 * C, x + r0 * C, w->norm1_gamma, w->norm1_beta, w->eps,
                          w->qkv_pack, w->qkv_b, w->q_scale, n_rows - r0, (int)C, stream);
}

static int mlp_by_rounds(float* out, const float* x, const hfl_block_weights* w, int64_t n_rows, int64_t C, void* ws,
                         int64_t ws_bytes, int split, hfl_stream_t stream) {
  const int64_t round = (int64_t)hfl_internal_stream_cus(stream) * (C == 256 ? 128 : 256);
  int64_t r0 = 0;
  while (split && g_round_launches && n_rows - r0 >= 2 * round) {
    int rc = hfl_ln_mlp_fused_ws(out + r0 * C, x + r0 * C, w->norm2_gamma, w->norm2_beta, w->eps, w->mlp_pack, w->fc1_b, w->fc2_b,
                                 round, (int)C, ws, ws_bytes, stream);
    if (rc != HFL_OK) return rc;
    r0 += round;
  }
  return hfl_ln_mlp_fused_ws(out + r0 * C, x + r0 * C, w->norm2_gamma, w->norm2_beta, w->eps, w->mlp_pack, w->fc1_b, w->fc2_b,
                             n_rows - r0, (int)C, ws, ws_bytes, stream);
}

// CUs a launch sizes its persistent grid for.  `cu_reserve` CUs are left out on purpose: the chip-filling kernels of the
// finest pyramid level occupy a CU completely (8 waves x 256 VGPRs, 135-150 KB of LDS), so while one of them runs no launch
// of another stream -- the coarse levels' and the relay tokens' short kernels -- finds a free slot anywhere; with a few CUs
// never claimed by the persistent grids those chains keep moving.
void hfl_internal_set_cu_reserve(int v) { g_cu_reserve = v < 0 ? 0 : v; }

int hfl_internal_stream_cus(void* stream) {
  int n = hfl_num_cus();
  if (stream != nullptr) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& e : g_stream_cus)
      if (e.first == stream) n = e.second;
  }
  return n - g_cu_reserve >= 8 ? n - g_cu_reserve : n;
}

// A HIP stream whose kernels run on a subset of the chip's CUs: mask bits [first_bit, first_bit + n_bits).  On gfx950 bit i
// of a CU mask is CU (i / 8) of XCD (i % 8) (tools/micro/cu_mask_census.hip), so a run of 8 k consecutive bits is k CUs of
// every XCD -- an even slice of every L2.  Streams with disjoint masks do not compete for CUs: persistent kernels of one do
// not wait for, or starve, the launches of the other.
int hfl_stream_create_cu_mask(hfl_stream_t* out, int first_bit, int n_bits) {
  const int total = hfl_num_cus();
  if (out == nullptr || first_bit < 0 || n_bits < 8 || first_bit % 8 != 0 || n_bits % 8 != 0 || first_bit + n_bits > total)
    return HFL_EINVAL;
  std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
  for (int b = first_bit; b < first_bit + n_bits; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) return (int)e;
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus.emplace_back(static_cast<void*>(s), n_bits);
  }
  *out = static_cast<hfl_stream_t>(s);
  return HFL_OK;
}

int hfl_stream_destroy(hfl_stream_t stream) {
  if (stream == nullptr) return HFL_EINVAL;
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (size_t i = 0; i < g_stream_cus.size(); ++i)
      if (g_stream_cus[i].first == stream) {
        g_stream_cus.erase(g_stream_cus.begin() + (long)i);
        break;
      }
  }
  return (int)hipStreamDestroy(static_cast<hipStream_t>(stream));
}

// ---- cross-stream ordering through a device flag instead of an event (probe; see tools/hop_latency.py).
// hfl_flag_set: a one-lane kernel that stores `value` to *flag (release) -- stream order puts it behind everything queued
// before it.  hfl_flag_wait: a one-lane kernel that polls *flag (acquire) until it is >= value, for at most `max_polls` polls
// (bounded: a lost signal costs a late, wrong result that the parity tests catch, never a hung queue); the launches behind it
// in its stream start when it exits.  The waiting side holds no CU resources to speak of (64 lanes, no LDS).
namespace {
__global__ void flag_set_kernel(unsigned int* flag, unsigned int value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void flag_wait_kernel(const unsigned int* flag, unsigned int value, int max_polls) {
  if (threadIdx.x == 0) {
    for (int i = 0; i < max_polls; ++i) {
      if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
}
}  // namespace

int hfl_flag_set(unsigned int* flag, unsigned int value, hfl_stream_t stream) {
  if (flag == nullptr) return HFL_EINVAL;
  flag_set_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(flag, value);
  HFL_RETURN_LAST_ERROR();
}

int hfl_flag_wait(const unsigned int* flag, unsigned int value, int max_polls, hfl_stream_t stream) {
  if (flag == nullptr || max_polls <= 0) return HFL_EINVAL;
  flag_wait_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(flag, value, max_polls);
  HFL_RETURN_LAST_ERROR();
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
  if (phase < 0 || phase > 6) return HFL_EINVAL;
  if (phase == 5 || phase == 6) {
    // proj + residual and the MLP branch of the RELAY rows (5) or of the TOKEN rows (6) alone: per-row operators, so the relay
    // rows -- all the next iteration's relay-token self-attention waits for -- can go first, as three small launches, and that
    // self-attention then runs beside the token rows' proj / MLP instead of after them
    const int64_t r0 = phase == 5 ? nt : 0, nr = phase == 5 ? rows - nt : nt;
    if (nr == 0) return HFL_OK;
    rc = hfl_linear_x3(x1 + r0 * C, o2 + r0 * 2 * C, w->proj_w, w->proj_b, x0 + r0 * C, nr, (int)C, (int)C, 0, stream);
    if (rc != HFL_OK) return rc;
    if (phase == 6 && w->mlp_pack != nullptr)
      return mlp_by_rounds(io->out + r0 * C, x1 + r0 * C, w, nr, C, a + 12 * unit, mlp_ws_bound(rows, C), 1, stream);
    if (w->fc1_w == nullptr || w->fc2_w == nullptr) return HFL_EINVAL;
    rc = hfl_layer_norm_split2(h2 + r0 * 2 * C, x1 + r0 * C, w->norm2_gamma, w->norm2_beta, nr, C, w->eps, stream);
    if (rc != HFL_OK) return rc;
    rc = hfl_linear_x3(g2 + r0 * 8 * C, h2 + r0 * 2 * C, w->fc1_w, w->fc1_b, nullptr, nr, (int)C, (int)(4 * C), 1, stream);
    if (rc != HFL_OK) return rc;
    return hfl_linear_x3(io->out + r0 * C, g2 + r0 * 8 * C, w->fc2_w, w->fc2_b, x1 + r0 * C, nr, (int)(4 * C), (int)C, 0, stream);
  }
  // ---- phase 1: everything that reads TOKEN rows only (independent of this iteration's relay-token self-attention)
  if (phase <= 1 && nt > 0) {
    rc = hfl_cpe_forward(x0, io->x_in, w->cpe_weight, w->cpe_gamma, w->cpe_beta, io->neigh, nt, C, 27, w->eps, 1, stream);
    if (rc != HFL_OK) return rc;
    if (phase == 1) {           // per-row operators: the token rows' LN1 and qkv do not wait for the relay rows either
      if (w->qkv_pack != nullptr) {
        rc = qkv_by_rounds(qkv, x0, w, nt, C, 1, stream);
      } else {
        rc = hfl_layer_norm_split2(a2, x0, w->norm1_gamma, w->norm1_beta, nt, C, w->eps, stream);
        if (rc != HFL_OK) return rc;
        rc = hfl_linear_x3_qkv(qkv, a2, w->qkv_w, w->qkv_b, nt, (int)C, (int)(3 * C), w->q_scale, stream);
      }
      if (rc != HFL_OK) return rc;
    }
  }
  if (phase == 1) return HFL_OK;
  // ---- phase 2 (or the whole block) = phase 3 (the relay rows up to their qkv) + the window attention + phase 4 (the rest);
  // 3 and 4 exist so that the attention of several blocks can go out as one launch in between (hfl_block_attention_x3_multi)
  // (phases 2 / 3 with the fused LN1 -> qkv launch available: the relay rows' qkv reads them where they are and goes out FIRST
  // -- the window attention waits for it, the copy of the rows into the block's buffer is only needed by proj's residual --
  // as one launch with the output features split over the workgroups instead of copy, LayerNorm, GEMM: three dependent small
  // launches of the finest level's critical chain per iteration)
  const bool relay_fused = (phase == 2 || phase == 3) && rows > nt && w->qkv_pack != nullptr;
  if (relay_fused) {
    const float* src = io->relay != nullptr ? io->relay : io->x_in + nt * C;
    rc = hfl_ln_qkv_fused(qkv + nt * 3 * C, src, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b, w->q_scale,
                          rows - nt, (int)C, stream);
    if (rc != HFL_OK) return rc;
  }
  if (phase != 4 && rows > nt) {
    const float* src = io->relay != nullptr ? io->relay : io->x_in + nt * C;
    hipError_t e = hipMemcpyAsync(x0 + nt * C, src, (size_t)(rows - nt) * C * 4, hipMemcpyDeviceToDevice,
                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return (int)e;
  }
  // blocks without relay rows (the OctFormer stage), whole-block call: LN1 -> qkv -> window attention as ONE launch, q / k / v
  // never in HBM (csrc/attn_fused.hip), when the configuration is one it takes
  const bool fused_attn = phase == 0 && rows == nt && w->fuse_attention != 0 && w->qkv_pack != nullptr &&
                          hfl_attn_fused_ok(desc, (int)C, w->rpe_table != nullptr) != 0;
  if (fused_attn) {
    rc = hfl_attn_fused_fwd(o2, x0, w->norm1_gamma, w->norm1_beta, w->eps, w->qkv_pack, w->qkv_b, w->q_scale, io->tok_meta,
                            w->rpe_table, desc, stream);
    if (rc != HFL_OK) return rc;
  }
  if (phase != 4 && !fused_attn && !relay_fused) {
    // LN1 + qkv of the rows phase 1 has not done: all of them (phase 0) or the relay rows (phases 2, 3)
    const int64_t r0 = phase != 0 ? nt : 0, nr = rows - r0;
    if (nr > 0) {
      rc = hfl_layer_norm_split2(a2 + r0 * 2 * C, x0 + r0 * C, w->norm1_gamma, w->norm1_beta, nr, C, w->eps, stream);
      if (rc != HFL_OK) return rc;
      rc = hfl_linear_x3_qkv(qkv + r0 * 3 * C, a2 + r0 * 2 * C, w->qkv_w, w->qkv_b, nr, (int)C, (int)(3 * C), w->q_scale,
                             stream);
      if (rc != HFL_OK) return rc;
    }
  }
  if (phase == 3) return HFL_OK;
  if (phase != 4 && !fused_attn) {
    rc = hfl_window_attention_fwd_ex(o2, qkv, nullptr, io->tok_meta, w->rpe_table, desc, 2 | 0x100, stream);
    if (rc != HFL_OK) return rc;
  }
  rc = hfl_linear_x3(x1, o2, w->proj_w, w->proj_b, x0, rows, (int)C, (int)C, 0, stream);
  if (rc != HFL_OK) return rc;
  if (w->mlp_pack != nullptr)       // the MLP branch in one launch: the 4C-wide hidden activation stays in registers
    // (workspace of the left-over rows' partial sums: behind the twelve units, sized by hfl_block_forward_x3_arena)
    return mlp_by_rounds(io->out, x1, w, rows, C, a + 12 * unit, mlp_ws_bound(rows, C), phase != 0, stream);
  rc = hfl_layer_norm_split2(h2, x1, w->norm2_gamma, w->norm2_beta, rows, C, w->eps, stream);
  if (rc != HFL_OK) return rc;
  rc = hfl_linear_x3(g2, h2, w->fc1_w, w->fc1_b, nullptr, rows, (int)C, (int)(4 * C), 1, stream);
  if (rc != HFL_OK) return rc;
  return hfl_linear_x3(io->out, g2, w->fc2_w, w->fc2_b, x1, rows, (int)(4 * C), (int)C, 0, stream);
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
  if (w->qkv_pack != nullptr) {
    // LN1 -> qkv as one launch into the fp16 (hi, lo) operand rows, the attention from them straight into proj's operand
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
