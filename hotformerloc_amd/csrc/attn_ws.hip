// The attention branch of a relay-token (H-OSA) transformer block up to the attention output as ONE kernel with
// SPECIALISED WAVES:
//
//     o (rows, C as bf16 split2) = window_attention( qkv( LayerNorm(x) ), relay q / k / v )      q, k, v of the token rows
//                                                                                                never leave the CU
// Replaces norm1 -> attention.qkv -> [hat_window_mask + padded RPE bias, SDPA] of the reference's OctreeAttention inside
// `data = data + attn(norm1(cat(rt, data)))` for the pyramid blocks (models/hotformerloc_backbone.py:197-216,
// models/octformer_backbone.py:52-93): C = 256, 16 heads of 16, K = 48 tokens + 1 relay token per window, dilation 1 -- the
// two launches hfl_ln_qkv_fused + window_attn_kernel_v5 (q, k, v crossed HBM as 24 of their 32 B per (row, channel)).  The
// relay rows (one per window, known only after the relay-token self-attention of the iteration) still get LayerNorm + qkv
// from hfl_ln_qkv_fused; their fp16 (hi, lo) operand rows are read here.
//
// Why specialised waves.  The first attempt (tools/experiments/attn_fused_rt) let every wave do both jobs: LayerNorm rows as
// MFMA B fragments (64 VGPRs) AND the softmax state of an attention unit in the 168 VGPRs a 12-wave workgroup allows -- hipcc
// spilled, and the spills' `s_waitcnt vmcnt(0)` drained the weight ring.  Here an 896-lane workgroup owns 96 token rows (two
// windows) and its waves have ONE job each:
//   * GEMM waves 0..5: wave w keeps LayerNorm(x) of rows 16 w .. 16 w + 15 as B fragments; Wqkv streams through a 3-slot LDS
//     ring in stages of one head's 16 features of one region (16 KiB of hfl_qkv_fused_pack's image); per HEAD PAIR six stages
//     (Q, Q', K, K', V, V') whose epilogues write the pair's fp16 (hi, lo) image [head][Q | K | V][rows][16 hi | 16 lo] in LDS;
//   * attention waves 6..13 (two per SIMD): the 12 (window, head, query tile) units and 4 relay queries of the PREVIOUS pair, from the other
//     image buffer: v5's arithmetic (two fp16 MFMAs per score tile, three 1-D RPE tables, exp2 softmax, V^T by transposing
//     LDS reads, split2 output).
// The GEMM half is matrix-pipe work, the attention half vector-pipe work: on a SIMD they overlap instead of queueing.  All
// fourteen waves meet at the six stage barriers of a step; the attention waves deal their units' phases over the six intervals.
// Token rows: the arithmetic of the two launches, operation for operation.
#include "hfl_common.h"
#include "x3_math.h"
#include "stage_stream.h"

#include <mutex>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ws_h8 __attribute__((ext_vector_type(8)));
typedef short ws_s4 __attribute__((ext_vector_type(4)));

constexpr float kWMask = -1e3f;      // models/octree.py:66
constexpr float kWDead = -1e30f;

constexpr int WC = 256;              // channels
constexpr int WH = 16;               // heads
constexpr int WK = 48;               // tokens per window
constexpr int WFT = 3;               // 16-row tiles per window
constexpr int WNWIN = 2;             // windows per workgroup tile
constexpr int WROWS = 96;            // token rows per workgroup tile
constexpr int WGW = 6;               // GEMM waves
constexpr int WAW = 8;               // attention waves (two per SIMD: a unit is a chain of LDS round trips, two waves cover each other)
constexpr int WW = WGW + WAW;
constexpr int WLAUNCH = 16;           // waves launched (four per SIMD): two of them leave at once, see WsParams::map
constexpr int WKS = WC / 32;         // k-steps of the qkv GEMM
constexpr int WSTAGE = WC * 64;      // bytes of a weight stage: 8 k-steps x 16 features x 128 B
constexpr int WNSLOT = 3;
constexpr int WSPR = WC / 32;        // 32-feature stages of the pack per region = head pairs
constexpr int WTSMAX = 768;          // floats of one head's expanded table (three 1-D tables, depth <= 7)
constexpr int WIMROWS = WROWS + 4;   // image rows: the tile's tokens + its windows' relay rows (+ 2 unused)
constexpr int WIMREG = WIMROWS * 64; // bytes of one region (Q, K or V) of one head: 64 B per row
constexpr int WIMHEAD = 3 * WIMREG;  // one head's image
constexpr int WIMG = 2 * WIMHEAD;    // a head pair's image

#ifdef HFL_PROBES
#define WS_DBG(bit) ((p.dbg & (bit)) != 0)
#else
#define WS_DBG(bit) false
#endif
struct WsParams {
  unsigned char* out;              // (rows, 2 C) bf16 split2: token rows, relay rows at rt_row0 + window
  const float* x;                  // (n_tokens, C) f32 token rows
  const float* gamma;
  const float* beta;
  const unsigned char* pack;       // hfl_qkv_fused_pack image of Wqkv
  const float* bias;               // (3 C)
  const uint32_t* meta;            // (n_tokens, 2): x | y << 10 | z << 20, batch id
  const float* rpe2;               // (H, TS) three clamped 1-D tables per head, log2e-prescaled, or null
  const unsigned char* relay_qkv;  // (n_windows, 3 C x 4 B): the relay rows' fp16 (hi, lo) operand rows (hfl_ln_qkv_fused)
  int64_t n_tokens;
  int64_t rt_row0;
  int n_windows;
  int n_tiles;
  // work units: tiles [0, full_tiles) whole (all 8 head pairs), then every later tile cut into `tail_parts` units of
  // 8 / tail_parts head pairs (no reduction: a part writes its own heads)
  int full_tiles;
  int tail_parts;
  int depth;
  int batch;
  float eps;
  float q_scale;
  int map;                         // wave -> role map (see the kernel)
#ifdef HFL_PROBES
  int dbg;                         // timing ablations (wrong results): 1 no attention work, 2 no GEMM k-loop, 4 no weight stream, 8 no relay units
#endif
};

__device__ __forceinline__ float ws_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float ws_rows_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float ws_rows_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ void ws_split_pair_f16(float p0, float p1, unsigned int& hi, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(p1));
  lo = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
__device__ __forceinline__ float ws_h2f(unsigned int packed, int half) {
  const unsigned short h = half ? (unsigned short)(packed >> 16) : (unsigned short)(packed & 0xFFFFu);
  return (float)__builtin_bit_cast(_Float16, h);
}

#define WS_DS_WRITE64(addr, val)                                                                                    \
  {                                                                                                                  \
    const u32x2 v__ = {(val).x, (val).y};                                                                            \
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v__) : "memory");                                            \
  }
#define WS_LDS_READ2(f0, f1, ahi, alo, off) \
  asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" : "=&v"(f0), "=&v"(f1) : "v"(ahi), "v"(alo), "n"(off))
#define WS_LDS_WAIT2(f0, f1, n) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(n))

// The weight ring is its OWN array: hipcc tracks LDS-DMA destinations by array (csrc/attn_fused.hip).
__shared__ __attribute__((aligned(1024))) unsigned char w_ring[WNSLOT * WSTAGE];
// w_lds: [tables 2 steps x 2 heads x TSMAX f32 | images 2 x WIMG | meta 2 x (s_qry int4 x 96 | s_key int2 x 96 | s_kbid int x 96) |
//         bias 3C f32 | gamma C | beta C | per-attention-wave output staging 6 x 1 KiB]
constexpr int WL_TAB = 0;
constexpr int WL_IMG = WL_TAB + 2 * 2 * WTSMAX * 4;
constexpr int WL_META = WL_IMG + 2 * WIMG;
constexpr int WMETA = WROWS * (16 + 8 + 4);
constexpr int WL_BIAS = WL_META + 2 * WMETA;
constexpr int WL_GAMMA = WL_BIAS + 3 * WC * 4;
constexpr int WL_BETA = WL_GAMMA + WC * 4;
constexpr int WL_STG = WL_BETA + WC * 4;
constexpr int WL_END = WL_STG + WAW * 1024;
__shared__ __attribute__((aligned(1024))) unsigned char w_lds[WL_END];
static_assert(WNSLOT * WSTAGE + WL_END <= 160 * 1024, "LDS budget of one workgroup per CU");

// 16-B chunk `ch` (0, 1: hi; 2, 3: lo) of image row r: slot ch ^ sw(r), sw = {0, 2, 3, 1}[(r >> 2) & 3] -- conflict-free for the
// operand reads (lane (c, g): row c, chunk g), the transposed V reads and the epilogue's 8-B writes
__device__ __forceinline__ int ws_sw(int r) { return (0x78 >> ((r >> 1) & 6)) & 3; }

// what an attention wave consumes in a step: the pair produced in the step before
struct WsJob {
  int valid;      // 0: nothing to do (the step before produced no image)
  int tile;
  int pair;       // head pair: heads 2 pair, 2 pair + 1
  int mpar;       // parity of the tile's metadata block
  int ipar;       // parity of the image / table buffers
};

template <int RPE>
__global__ void __launch_bounds__(WLAUNCH * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
attn_ws_kernel(const WsParams p) {
  typedef __attribute__((address_space(3))) const float lds_f32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // role of the hardware wave (waves go to SIMD id % 4 in launch order): map 0 = GEMM waves 0..5 (two on SIMDs 0 / 1, one on
  // 2 / 3), attention waves 6..13 (two per SIMD); map 1 = the attention waves where the GEMM waves are not: three on SIMDs 2 / 3
  // (hardware waves 6, 7, 10, 11, 14, 15), one on SIMDs 0 / 1 (8, 9).  The two waves without a role leave after the
  // first barrier.
  int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (p.map == 1) {
    const int hw = wave;
    wave = hw < 8 ? hw : hw < 10 ? hw + 4 : hw < 12 ? hw - 2 : hw < 14 ? -1 : hw - 4;
  } else if (wave >= WW) {
    wave = -1;
  }
  const bool gemm_wave = wave >= 0 && wave < WGW;
  const int fr = lane & 15, fq = lane >> 4;          // GEMM: row of the wave's tile, k / feature quarter
  const int c = lane & 15, g = lane >> 4;            // attention: column of a 16-tile, 4-row group
  float* s_tab = reinterpret_cast<float*>(w_lds + WL_TAB);
  unsigned char* s_img = w_lds + WL_IMG;
  float* bs = reinterpret_cast<float*>(w_lds + WL_BIAS);
  float* gms = reinterpret_cast<float*>(w_lds + WL_GAMMA);
  float* bts = reinterpret_cast<float*>(w_lds + WL_BETA);

  const int R = (1 << p.depth) - 1, W = 2 * R + 1;
  const int TS = RPE ? ((3 * W + 3) & ~3) : 0;
  const int n_tok = (int)p.n_tokens;

  for (int i = tid; i < 3 * WC / 4; i += WLAUNCH * 64) reinterpret_cast<float4*>(bs)[i] = reinterpret_cast<const float4*>(p.bias)[i];
  for (int i = tid; i < WC / 4; i += WLAUNCH * 64) {
    reinterpret_cast<float4*>(gms)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
    reinterpret_cast<float4*>(bts)[i] = reinterpret_cast<const float4*>(p.beta)[i];
  }
  __syncthreads();
  if (wave < 0) return;

  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  auto unit_of = [&](int unit, int& tile, int& pr0, int& npr) {
    tile = unit; pr0 = 0; npr = WSPR;
    if (unit >= p.full_tiles) {
      const int v = unit - p.full_tiles;
      tile = p.full_tiles + v / p.tail_parts;
      npr = WSPR / p.tail_parts;
      pr0 = (v % p.tail_parts) * npr;
    }
  };
  // a step's barrier for a wave without weight-ring duties at that point
  auto step_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  if (gemm_wave) {
    // =============================================================== GEMM waves
    // weight ring: a stage = 16 pieces of 1 KiB, four each from waves 0..3 (wave w: k-steps 2 w, 2 w + 1 -- the 2 KiB of the
    // head's 16 features inside the pack's 4-KiB k-step block of 32 features); two stages ahead of the consumed one
    const bool loader = wave < 4;
    const uint32_t lane_off = (uint32_t)lane * 16u;
    int pr0 = 0, nst_cur = 0;                  // current unit: first pair, stages (6 per pair)
    auto issue = [&](int n, int slot) {
      if (!loader || WS_DBG(4)) return;
      const int pr = pr0 + n / 6, j = n % 6;       // stage j of the pair: region j >> 1, head 2 pr + (j & 1)
      const unsigned char* s = p.pack + (int64_t)((j >> 1) * WSPR + pr) * (WC * 128) + wave * 8192 + (j & 1) * 2048 + lane_off;
      unsigned char* d = w_ring + slot * WSTAGE + wave * 4096;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                       (__attribute__((address_space(3))) void*)d, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + 4096),
                                       (__attribute__((address_space(3))) void*)(d + 2048), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + 4096),
                                       (__attribute__((address_space(3))) void*)(d + 2048), 16, 1024, 0);
    };
    uint32_t seq = 0;
    auto acquire = [&](int n) -> const unsigned char* {
      if (loader) {
        if (n + 1 < nst_cur) HFL_WAIT_VM(4);
        else HFL_WAIT_VM(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's image writes of the stage before are performed)
      __builtin_amdgcn_s_barrier();
      if (n + 2 < nst_cur) issue(n + 2, (int)((seq + 2) % WNSLOT));
      const unsigned char* st = w_ring + (seq % WNSLOT) * WSTAGE;
      ++seq;
      return st;
    };
    const int off_hi = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), off_lo = off_hi ^ 64;
    int t = 0, ui = 0;
    for (int unit = blockIdx.x;; unit += gridDim.x, ++ui) {
      const bool have = unit < n_units;
      int tile = 0, npr = 0;
      if (have) unit_of(unit, tile, pr0, npr);
      // ---- prologue step: LayerNorm rows of the unit's tile, its metadata, the first two stages of its weight stream
      step_barrier();
      bf16x8 xh[WKS], xl[WKS];
      if (have) {
        const int row0 = tile * WROWS;
        {
          int r = row0 + wave * 16 + fr;
          if (r >= n_tok) r = n_tok - 1;
          const float* xr = p.x + (int64_t)r * WC + fq * 8;
          float4 a[WKS][2];
          float sum = 0.f;
#pragma unroll
          for (int ks = 0; ks < WKS; ++ks) {
            a[ks][0] = *reinterpret_cast<const float4*>(xr + ks * 32);
            a[ks][1] = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
            sum += ((a[ks][0].x + a[ks][0].y) + (a[ks][0].z + a[ks][0].w)) + ((a[ks][1].x + a[ks][1].y) + (a[ks][1].z + a[ks][1].w));
          }
          sum += __shfl_xor(sum, 16, 64);
          sum += __shfl_xor(sum, 32, 64);
          const float mean = sum * (1.0f / (float)WC);
          float sq = 0.f;
#pragma unroll
          for (int ks = 0; ks < WKS; ++ks)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              a[ks][h].x -= mean; a[ks][h].y -= mean; a[ks][h].z -= mean; a[ks][h].w -= mean;
              sq += (a[ks][h].x * a[ks][h].x + a[ks][h].y * a[ks][h].y) + (a[ks][h].z * a[ks][h].z + a[ks][h].w * a[ks][h].w);
            }
          sq += __shfl_xor(sq, 16, 64);
          sq += __shfl_xor(sq, 32, 64);
          const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)WC) + p.eps);
#pragma unroll
          for (int ks = 0; ks < WKS; ++ks) {
            uint32_t hi[4], lo[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float4 gm = *reinterpret_cast<const float4*>(gms + ks * 32 + fq * 8 + h * 4);
              const float4 bt = *reinterpret_cast<const float4*>(bts + ks * 32 + fq * 8 + h * 4);
              const f32x2 v01 = {fmaf(a[ks][h].x * rstd, gm.x, bt.x), fmaf(a[ks][h].y * rstd, gm.y, bt.y)};
              const f32x2 v23 = {fmaf(a[ks][h].z * rstd, gm.z, bt.z), fmaf(a[ks][h].w * rstd, gm.w, bt.w)};
              x3_split_pair(v01, hi[2 * h], lo[2 * h]);
              x3_split_pair(v23, hi[2 * h + 1], lo[2 * h + 1]);
            }
            xh[ks] = __builtin_bit_cast(bf16x8, (u32x4){hi[0], hi[1], hi[2], hi[3]});
            xl[ks] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], lo[2], lo[3]});
            // (pinned: left alone hipcc sinks this arithmetic towards its first use and carries the raw rows + gamma / beta)
            asm volatile("" : "+v"(xh[ks]), "+v"(xl[ks]));
          }
        }
        // metadata of the tile's 96 rows into the block of this unit's parity: query side {4 x, 4 y | 4 z << 16, batch id,
        // global row}, key side {4 (R - x), 4 (W + R - y) | 4 (2 W + R - z) << 16}, batch id (-1: the row does not exist)
        if (tid < WROWS) {
          unsigned char* mb = w_lds + WL_META + (ui & 1) * WMETA;
          int4* s_qry = reinterpret_cast<int4*>(mb);
          int2* s_key = reinterpret_cast<int2*>(mb + WROWS * 16);
          int* s_kbid = reinterpret_cast<int*>(mb + WROWS * 24);
          const int tk = row0 + tid;
          int bid = -1, row = -1, x = 0, y = 0, z = 0;
          if (tk < n_tok) {
            const uint2 mt = *reinterpret_cast<const uint2*>(p.meta + 2 * (int64_t)tk);
            x = (int)(mt.x & 1023u); y = (int)((mt.x >> 10) & 1023u); z = (int)(mt.x >> 20);
            bid = (int)mt.y;
            row = tk;
          }
          s_key[tid] = make_int2(4 * (R - x), (4 * (W + R - y)) | ((4 * (2 * W + R - z)) << 16));
          s_qry[tid] = make_int4(4 * x, (4 * y) | ((4 * z) << 16), bid, row);
          s_kbid[tid] = bid;
        }
        nst_cur = 6 * npr;
        issue(0, (int)(seq % WNSLOT));
        issue(1, (int)((seq + 1) % WNSLOT));
      }
#pragma unroll 1
      for (int j = 1; j < 6; ++j) step_barrier();
      ++t;
      if (!have) break;
      // ---- pair steps: six stages each, every stage behind the step's next barrier
#pragma unroll 1
      for (int pp = 0; pp < npr; ++pp, ++t) {
        unsigned char* img = s_img + (t & 1) * WIMG;
#pragma unroll 1
        for (int j = 0; j < 6; ++j) {
          const unsigned char* st = acquire(6 * pp + j);
          const int reg = j >> 1, hl = j & 1, hd = 2 * (pr0 + pp) + hl;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const uint32_t ahi = (uint32_t)(uintptr_t)(st + off_hi), alo = (uint32_t)(uintptr_t)(st + off_lo);
          bf16x8 wf[4][2];
          // (the stage's bias rides in front of the first fragment reads: LDS operations return in order, so the k-loop's first
          // counted wait covers it and the epilogue starts without a round trip of its own)
          f32x4 b;
          const uint32_t baddr = (uint32_t)(uintptr_t)(bs + reg * WC + hd * 16 + fq * 4);
          asm volatile("ds_read_b128 %0, %1" : "=&v"(b) : "v"(baddr));
          if (!WS_DBG(2)) {
          WS_LDS_READ2(wf[0][0], wf[0][1], ahi, alo, 0);
          WS_LDS_READ2(wf[1][0], wf[1][1], ahi, alo, 2048);
          hfl_static_for(std::make_integer_sequence<int, WKS>{}, [&](auto kc) {
            constexpr int ks = decltype(kc)::value;
            if constexpr (ks + 2 < WKS) {
              WS_LDS_READ2(wf[(ks + 2) & 3][0], wf[(ks + 2) & 3][1], ahi, alo, (ks + 2) * 2048);
              WS_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 4);
            } else if constexpr (ks + 1 < WKS) {
              WS_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 2);
            } else {
              WS_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][0], xl[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][1], xh[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][0], xh[ks], acc, 0, 0, 0);
          });
          }
          // epilogue: bias, query scale, fp16 (hi, lo) split (csrc/qkv_fused.hip / gemm_x3's EPI 2) into the head's image: row
          // 16 wave + fr, features 4 fq .. 4 fq + 3: hi 8 B of chunk fq >> 1, lo of chunk 2 + (fq >> 1)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b), "+v"(acc));
          float v0 = acc[0] + b[0], v1 = acc[1] + b[1], v2 = acc[2] + b[2], v3 = acc[3] + b[3];
          if (reg == 0) { v0 *= p.q_scale; v1 *= p.q_scale; v2 *= p.q_scale; v3 *= p.q_scale; }
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(v0, v1), h23 = __builtin_amdgcn_cvt_pkrtz(v2, v3);
          const auto l01 = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1]);
          const auto l23 = __builtin_amdgcn_cvt_pkrtz(v2 - (float)h23[0], v3 - (float)h23[1]);
          const uint2 hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
          const uint2 lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
          const int trow = wave * 16 + fr;
          unsigned char* ir = img + hl * WIMHEAD + reg * WIMREG + trow * 64;
          const int sw = ws_sw(trow);
          WS_DS_WRITE64((uint32_t)(uintptr_t)(ir + (((fq >> 1) ^ sw) << 4) + (fq & 1) * 8), hi);
          WS_DS_WRITE64((uint32_t)(uintptr_t)(ir + (((2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8), lo);
        }
      }
    }
    return;
  }

  // ================================================================= attention waves
  // Eight waves, two per SIMD.  Per step a wave runs two of the pair's 16 units (12 (window, head, query tile) units + 4 relay
  // queries), each in three phases (scores, softmax, output): one phase per interval between the step's barriers.
  const int aw = wave - WGW;                           // 0 .. 7
  unsigned char* stg = w_lds + WL_STG + aw * 1024;
  const int st_quad = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const int st_row = c * 64, st_x = (c >> 2) & 3;
  const float mask2 = kWMask * 1.4426950408889634f;

  struct UnitState {
    f32x4 sc[WFT];      // scores, then the un-normalised probabilities of the token keys
    float srt;          // score of the relay key (lanes g == 0), then its probability
    float inv;
    int homog, q_bid, rt_bid;
  };
  // a unit = (window wl, head hl, query tile qt); qt == WFT: the relay query of the window (one live query column, no RPE)
  auto unit_scores = [&](const WsJob& jb, int wl, int hl, int qt, UnitState& us) {
    const unsigned char* img = s_img + jb.ipar * WIMG + hl * WIMHEAD;
    const unsigned char* mb = w_lds + WL_META + jb.mpar * WMETA;
    const int4* s_qry = reinterpret_cast<const int4*>(mb);
    const int2* s_key = reinterpret_cast<const int2*>(mb + WROWS * 16);
    const int* s_kbid = reinterpret_cast<const int*>(mb + WROWS * 24);
    auto iaddr = [&](int reg, int r, int ch) -> const unsigned char* { return img + reg * WIMREG + r * 64 + ((ch ^ ws_sw(r)) << 4); };
    const bool is_rt = qt == WFT;
    const int wbase = wl * WK, rrow = WROWS + wl;
    const int tabb = (int)(uintptr_t)(s_tab + (jb.ipar * 2 + hl) * WTSMAX);
    const int bid0 = s_kbid[wbase], bidl = s_kbid[wbase + WK - 1];
    us.homog = __builtin_amdgcn_readfirstlane((bidl >= 0 && bid0 == bidl) ? 1 : 0);
    us.rt_bid = bid0 >= 0 ? bid0 : p.batch;           // the relay token carries the id of the window's first token
    uint4 ka[WFT + 1], qh, ql;
#pragma unroll
    for (int kt = 0; kt < WFT; ++kt) ka[kt] = *reinterpret_cast<const uint4*>(iaddr(1, wbase + kt * 16 + c, g));
    ka[WFT] = make_uint4(0u, 0u, 0u, 0u);               // relay key tile: one live key, position 0
    if (c == 0) ka[WFT] = *reinterpret_cast<const uint4*>(iaddr(1, rrow, g));
    int4 qm = make_int4(0, 0, -1, -1);
    if (!is_rt) {
      const int r = wbase + qt * 16 + c;
      qh = *reinterpret_cast<const uint4*>(iaddr(0, r, g & 1));
      ql = *reinterpret_cast<const uint4*>(iaddr(0, r, 2 + (g & 1)));
      qm = s_qry[wbase + qt * 16 + c];
    } else {                                            // the relay query: column 0 of an otherwise empty query tile
      qh = ql = make_uint4(0u, 0u, 0u, 0u);
      if (c == 0) {
        qh = *reinterpret_cast<const uint4*>(iaddr(0, rrow, g & 1));
        ql = *reinterpret_cast<const uint4*>(iaddr(0, rrow, 2 + (g & 1)));
      }
    }
    us.q_bid = is_rt ? us.rt_bid : qm.z;
    const int qxa = qm.x + tabb, qyza = qm.y + tabb * 0x10001;
    f32x4 sc[WFT + 1];
#pragma unroll
    for (int kt = 0; kt <= WFT; ++kt) {
      const ws_h8 ak = __builtin_bit_cast(ws_h8, ka[kt]);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(ws_h8, ql), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(ws_h8, qh), acc, 0, 0, 0);
      sc[kt] = acc;
    }
    if (RPE && !is_rt) {     // no RPE for the relay row / column (octformer_backbone.py:78-80)
#pragma unroll
      for (int kt = 0; kt < WFT; ++kt) {
        f32x4 bx, by, bz;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int2 km = s_key[wbase + kt * 16 + 4 * g + r];
          const uint32_t tt = (uint32_t)(km.y + qyza);                  // both halves are LDS byte addresses
          bx[r] = *reinterpret_cast<lds_f32*>(km.x + qxa);
          by[r] = *reinterpret_cast<lds_f32*>((int)(tt & 0xFFFFu));
          bz[r] = *reinterpret_cast<lds_f32*>((int)(tt >> 16));
        }
        sc[kt] += (bx + by) + bz;
      }
    }
    if (!us.homog) {
#pragma unroll
      for (int kt = 0; kt < WFT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (s_kbid[wbase + kt * 16 + 4 * g + r] != us.q_bid) sc[kt][r] += mask2;
    }
#pragma unroll
    for (int kt = 0; kt < WFT; ++kt) us.sc[kt] = sc[kt];
    us.srt = sc[WFT][0] + (g == 0 ? 0.f : kWDead);      // relay key: tile FT, key 0 = register 0 of the g == 0 lanes
    if (!us.homog && us.rt_bid != us.q_bid) us.srt += mask2;
  };
  auto unit_softmax = [&](UnitState& us) {
    float mx = us.srt;
#pragma unroll
    for (int kt = 0; kt < WFT; ++kt) {
      mx = ws_max3(mx, us.sc[kt][0], us.sc[kt][1]);
      mx = ws_max3(mx, us.sc[kt][2], us.sc[kt][3]);
    }
    mx = ws_rows_max(mx);
    const f32x4 nmx4 = {-mx, -mx, -mx, -mx};
    f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < WFT; ++kt) {
      f32x4 e = us.sc[kt] + nmx4;
      e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
      e[2] = __builtin_amdgcn_exp2f(e[2]); e[3] = __builtin_amdgcn_exp2f(e[3]);
      us.sc[kt] = e;
      sum4 += e;
    }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    us.srt = __builtin_amdgcn_exp2f(us.srt - mx);          // zero in the lanes g != 0
    sum += us.srt;
    sum = ws_rows_sum(sum);
    us.inv = __builtin_amdgcn_rcpf(sum);
  };
  auto unit_output = [&](const WsJob& jb, int wl, int hl, int qt, const UnitState& us) {
    const unsigned char* img = s_img + jb.ipar * WIMG + hl * WIMHEAD;
    const unsigned char* mb = w_lds + WL_META + jb.mpar * WMETA;
    const int4* s_qry = reinterpret_cast<const int4*>(mb);
    auto iaddr = [&](int reg, int r, int ch) -> const unsigned char* { return img + reg * WIMREG + r * 64 + ((ch ^ ws_sw(r)) << 4); };
    const bool is_rt = qt == WFT;
    const int wbase = wl * WK, rrow = WROWS + wl;
    const int hd = 2 * jb.pair + hl;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      // V^T fragments of the pair of key tiles (0, 1) / (2, relay): lane 4 q + p of a 16-lane group addresses key q of its
      // 4-key block, dims 4 p .. 4 p + 3; the relay key's tile: key 0 only (lanes g == 0, element 0)
      ws_h8 vhi_p, vlo_p;
      {
        typedef __attribute__((address_space(3))) ws_s4 lds_s4;
        const int kk = 4 * g + (c >> 2), cp = c & 3;
        ws_s4 hh[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, ll[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int kt = 2 * pp + e;
          if (kt < WFT) {
            const int r = wbase + kt * 16 + kk;
            hh[e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r, cp >> 1) + (cp & 1) * 8));
            ll[e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r, 2 + (cp >> 1)) + (cp & 1) * 8));
          } else {
            const unsigned short vh = *reinterpret_cast<const unsigned short*>(iaddr(2, rrow, c >> 3) + (c & 7) * 2);
            const unsigned short vl = *reinterpret_cast<const unsigned short*>(iaddr(2, rrow, 2 + (c >> 3)) + (c & 7) * 2);
            hh[e][0] = g == 0 ? (short)vh : (short)0;
            ll[e][0] = g == 0 ? (short)vl : (short)0;
          }
        }
        const short __attribute__((ext_vector_type(8))) h8 = {hh[0][0], hh[0][1], hh[0][2], hh[0][3], hh[1][0], hh[1][1], hh[1][2], hh[1][3]};
        const short __attribute__((ext_vector_type(8))) l8 = {ll[0][0], ll[0][1], ll[0][2], ll[0][3], ll[1][0], ll[1][1], ll[1][2], ll[1][3]};
        vhi_p = __builtin_bit_cast(ws_h8, h8);
        vlo_p = __builtin_bit_cast(ws_h8, l8);
      }
      f32x4 oa = {0.f, 0.f, 0.f, 0.f};
      unsigned int hh0 = 0u, hh1 = 0u, hh2 = 0u, hh3 = 0u, ll0 = 0u, ll1 = 0u, ll2 = 0u, ll3 = 0u;
      ws_split_pair_f16(us.sc[2 * pp][0], us.sc[2 * pp][1], hh0, ll0);
      ws_split_pair_f16(us.sc[2 * pp][2], us.sc[2 * pp][3], hh1, ll1);
      if (2 * pp + 1 < WFT) {
        ws_split_pair_f16(us.sc[2 * pp + 1][0], us.sc[2 * pp + 1][1], hh2, ll2);
        ws_split_pair_f16(us.sc[2 * pp + 1][2], us.sc[2 * pp + 1][3], hh3, ll3);
      } else {
        ws_split_pair_f16(us.srt, 0.f, hh2, ll2);
        ll2 &= 0xFFFFu;
      }
      const u32x4 uh = {hh0, hh1, hh2, hh3}, ul = {ll0, ll1, ll2, ll3};
      const ws_h8 phi = __builtin_bit_cast(ws_h8, uh), plo = __builtin_bit_cast(ws_h8, ul);
      oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi_p, plo, oa, 0, 0, 0);
      oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vlo_p, phi, oa, 0, 0, 0);
      oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi_p, phi, oa, 0, 0, 0);
      o += oa;
    }
    o *= us.inv;
    uint2 hi, lo;
    x3_split_pair_scalar(o[0], o[1], hi.x, lo.x);
    x3_split_pair_scalar(o[2], o[3], hi.y, lo.y);
    WS_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((g >> 1) ^ st_x) * 16) + (g & 1) * 8), hi);
    WS_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((2 + (g >> 1)) ^ st_x) * 16) + (g & 1) * 8), lo);
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(stg + st_quad)) : "memory");
    const int rl = lane >> 2, ch = lane & 3;
    int orow_l;
    if (!is_rt) {
      orow_l = s_qry[wbase + qt * 16 + rl].w;
    } else {
      const int wglob = jb.tile * WNWIN + wl;
      orow_l = (rl == 0 && wglob < p.n_windows) ? (int)p.rt_row0 + wglob : -1;
    }
    if (orow_l >= 0)
      *reinterpret_cast<uint4*>(p.out + (size_t)orow_l * (uint32_t)(4 * WC) +
                                (uint32_t)((hd >> 1) * 128 + (hd & 1) * 32 + (ch & 1) * 16 + (ch >> 1) * 64)) = v;
  };

  // what the step's GEMM waves are producing: wave 0 fetches the pair's tables, wave 1 the relay rows of the tile's windows,
  // early in the step, and writes them into the buffers of the step's parity behind the step's last barrier
  auto side_fetch = [&](bool producing, int tile, int pair, uint4& held) {
    held = make_uint4(0u, 0u, 0u, 0u);
    if (!producing) return;
    if (aw == 0 && RPE && lane < 2 * TS / 4 && 2 * TS / 4 <= 64)
      held = *reinterpret_cast<const uint4*>(p.rpe2 + (size_t)(2 * pair) * TS + lane * 4);      // heads 2 pair, 2 pair + 1: contiguous
    if (aw == 1 && lane < 2 * WNWIN * 12) {      // head lane / 24, window (lane / 12) % 2, region (lane / 4) % 3, 16-B chunk lane % 4
      const int h2 = lane / 24, w2 = (lane / 12) % 2, rg = (lane >> 2) % 3, ch = lane & 3;
      const int wg2 = tile * WNWIN + w2;
      if (wg2 < p.n_windows)
        held = *reinterpret_cast<const uint4*>(p.relay_qkv + (size_t)wg2 * (3 * WC * 4) + rg * (WC * 4) + (2 * pair + h2) * 64 + ch * 16);
    }
  };
  auto side_store = [&](bool producing, int pair, int ipar, const uint4& held) {
    if (!producing) return;
    if (aw == 0 && RPE) {
      float* dst = s_tab + (ipar * 2) * WTSMAX;
      if (2 * TS / 4 <= 64) {
        // (the two heads' tables are TS floats apart in memory and WTSMAX apart in LDS)
        if (lane < 2 * TS / 4) {
          const int hh = (lane * 4) / TS, off = (lane * 4) % TS;
          *reinterpret_cast<uint4*>(dst + hh * WTSMAX + off) = held;
        }
      } else {
        for (int hh = 0; hh < 2; ++hh) {
          const float4* src = reinterpret_cast<const float4*>(p.rpe2 + (size_t)(2 * pair + hh) * TS);
          for (int i = lane; i < TS / 4; i += 64) reinterpret_cast<float4*>(dst + hh * WTSMAX)[i] = src[i];
        }
      }
    }
    if (aw == 1 && lane < 2 * WNWIN * 12) {
      const int h2 = lane / 24, w2 = (lane / 12) % 2, rg = (lane >> 2) % 3, ch = lane & 3;
      const int r = WROWS + w2;
      *reinterpret_cast<uint4*>(s_img + ipar * WIMG + h2 * WIMHEAD + rg * WIMREG + r * 64 + ((ch ^ ws_sw(r)) << 4)) = held;
    }
  };

  // a step of an attention wave: its two units' six phases, one per interval behind the step's six barriers.  Units of a pair:
  // v = 0 .. 11 the token units (window v / 6, head (v / 3) & 1, query tile v % 3), v = 12 .. 15 the relay queries (window
  // (v - 12) >> 1, head (v - 12) & 1); wave a takes v = a and a + 8.
  auto attn_step = [&](const WsJob& jb, bool producing, int ptile, int ppair, int pipar) {
    UnitState us;
    uint4 held;
    const bool work = jb.valid && !WS_DBG(1);
    const int v0 = aw, v1 = aw + 8;
    const int w0 = v0 / 6, h0_ = (v0 / 3) & 1, q0 = v0 % 3;
    const int w1 = v1 < 12 ? v1 / 6 : (v1 - 12) >> 1, h1_ = v1 < 12 ? (v1 / 3) & 1 : (v1 - 12) & 1, q1 = v1 < 12 ? v1 % 3 : WFT;
    const bool work1 = work && !(q1 == WFT && WS_DBG(8));
    step_barrier();                                   // B0: the image of the step before is complete
    side_fetch(producing, ptile, ppair, held);
    if (work) unit_scores(jb, w0, h0_, q0, us);
    step_barrier();                                   // B1
    if (work) unit_softmax(us);
    step_barrier();                                   // B2
    if (work) unit_output(jb, w0, h0_, q0, us);
    step_barrier();                                   // B3
    if (work1) unit_scores(jb, w1, h1_, q1, us);
    step_barrier();                                   // B4
    if (work1) unit_softmax(us);
    step_barrier();                                   // B5
    if (work1) unit_output(jb, w1, h1_, q1, us);
    side_store(producing, ppair, pipar, held);
  };

  {
    int t = 0, ui = 0;
    WsJob prev = {0, 0, 0, 0, 0};
    for (int unit = blockIdx.x;; unit += gridDim.x, ++ui) {
      const bool have = unit < n_units;
      int tile = 0, pr0 = 0, npr = 0;
      if (have) unit_of(unit, tile, pr0, npr);
      attn_step(prev, false, 0, 0, 0);               // prologue step of the GEMM waves: the last pair of the unit before
      prev.valid = 0;
      ++t;
      if (!have) break;
#pragma unroll 1
      for (int pp = 0; pp < npr; ++pp, ++t) {
        attn_step(prev, true, tile, pr0 + pp, t & 1);
        prev.valid = 1; prev.tile = tile; prev.pair = pr0 + pp; prev.mpar = ui & 1; prev.ipar = t & 1;
      }
    }
  }
}

}  // namespace

// per-launch HIP events for bench.py's roofline leg (the launches sit inside hfl_block_forward_x3: no Python timer sees them)
struct WsTimingRec {
  hipEvent_t e0, e1;
  double bytes, flops_gemm, flops_attn;
};
static int g_ws_timing = 0;
static int g_ws_map = 1;     // measured at depth 4: 160 us against 173 for map 0 (profiles/r05_v_attn_ws_wave_map.log)
extern "C" void hfl_internal_set_ws_map(int v) { g_ws_map = v == 1 ? 1 : 0; }
#ifdef HFL_PROBES
static int g_ws_dbg = 0;
extern "C" void hfl_internal_set_ws_dbg(int v) { g_ws_dbg = v; }
#endif
static std::vector<WsTimingRec> g_ws_recs;
static std::mutex g_ws_mu;

extern "C" {

/* 1 when hfl_attn_ws_fwd takes this configuration (see include/hotformerloc_hip.h), else 0 */
int hfl_attn_ws_ok(const hfl_window_attn_desc* d, int channels) {
  if (d == nullptr || channels != WC || d->n_heads != WH || d->n_relay != 1 || d->dilation != 1) return 0;
  if (d->patch_size != WK) return 0;
  if (d->depth < 1 || d->depth > 7) return 0;
  if (d->n_tokens <= 0 || d->n_windows <= 0 || d->rt_row0 < d->n_tokens) return 0;
  if ((d->rt_row0 + d->n_windows) >= ((int64_t)1 << 31) / (4 * WC)) return 0;
  if ((int64_t)d->n_windows * d->patch_size < d->n_tokens) return 0;
  return 1;
}

int hfl_attn_ws_fwd(void* out_split2, const float* x, const float* gamma, const float* beta, float eps, const void* qkv_pack,
                    const float* qkv_bias, float q_scale, const void* relay_qkv, const uint32_t* tok_meta,
                    const float* rpe_tables3, const hfl_window_attn_desc* d, hfl_stream_t stream) {
  if (out_split2 == nullptr || x == nullptr || gamma == nullptr || beta == nullptr || qkv_pack == nullptr ||
      qkv_bias == nullptr || relay_qkv == nullptr || tok_meta == nullptr || d == nullptr)
    return HFL_EINVAL;
  if (!hfl_attn_ws_ok(d, WC)) return HFL_EINVAL;
  WsParams p;
  p.out = static_cast<unsigned char*>(out_split2); p.x = x; p.gamma = gamma; p.beta = beta;
  p.pack = static_cast<const unsigned char*>(qkv_pack); p.bias = qkv_bias; p.meta = tok_meta;
  p.rpe2 = rpe_tables3;
  p.relay_qkv = static_cast<const unsigned char*>(relay_qkv);
  p.n_tokens = d->n_tokens; p.rt_row0 = d->rt_row0; p.n_windows = d->n_windows;
  // (tiles cover every window of the plan: the windows past the last token -- the reference pads the token stream to a multiple
  // of K x the stage dilation, models/octree.py:73-75 -- have a relay row too)
  const int64_t t_tok = hfl_cdiv(d->n_tokens, WROWS), t_win = hfl_cdiv(d->n_windows, WNWIN);
  p.n_tiles = (int)(t_tok > t_win ? t_tok : t_win); p.depth = d->depth;
  p.batch = d->batch_size; p.eps = eps; p.q_scale = q_scale; p.map = g_ws_map;
#ifdef HFL_PROBES
  p.dbg = g_ws_dbg;
#endif
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cus = hfl_stream_cus(s);
  // whole rounds of the grid take whole tiles; the tiles left over are cut by head pairs when that lets them share a round
  p.full_tiles = p.n_tiles;
  p.tail_parts = 1;
  {
    const int full = p.n_tiles / cus * cus, rem = p.n_tiles - full;
    int parts = 1;
    while (parts * 2 <= WSPR / 2 && rem * parts * 2 <= cus) parts *= 2;
    if (rem > 0 && parts > 1) {
      p.full_tiles = full;
      p.tail_parts = parts;
    }
  }
  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  const int grid = n_units < cus ? n_units : cus;
  WsTimingRec rec{};
  const bool timed = g_ws_timing != 0;
  if (timed) {
    const double L = d->patch_size + 1;
    rec.bytes = (double)d->n_tokens * WC * 8.0 + (double)d->n_tokens * 8.0 + (double)d->n_windows * WC * 16.0;
    rec.flops_gemm = 6.0 * (double)d->n_tokens * WC * WC;
    rec.flops_attn = 4.0 * L * L * WC * (double)d->n_windows;
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess || hipEventRecord(rec.e0, s) != hipSuccess) {
      if (rec.e0) (void)hipEventDestroy(rec.e0);
      if (rec.e1) (void)hipEventDestroy(rec.e1);
      return HFL_EINVAL;
    }
  }
  if (p.rpe2 != nullptr) attn_ws_kernel<1><<<grid, WLAUNCH * 64, 0, s>>>(p);
  else attn_ws_kernel<0><<<grid, WLAUNCH * 64, 0, s>>>(p);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_ws_mu);
    g_ws_recs.push_back(rec);
  }
  HFL_RETURN_LAST_ERROR();
}

// bench.py: per-launch timing of hfl_attn_ws_fwd on / off (both drop what was recorded) ...
int hfl_internal_ws_timing(int on) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (auto& r : g_ws_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_ws_recs.clear();
  g_ws_timing = on ? 1 : 0;
  return HFL_OK;
}
// ... and read it: per launch the duration (ms), algorithmic bytes, useful GEMM and attention flop; returns the launches recorded
int hfl_internal_ws_timing_read(double* ms, double* bytes, double* flops_gemm, double* flops_attn, int cap) {
  std::lock_guard<std::mutex> lk(g_ws_mu);
  int n = 0;
  for (auto& r : g_ws_recs) {
    if (n >= cap) break;
    if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
    ms[n] = t; bytes[n] = r.bytes; flops_gemm[n] = r.flops_gemm; flops_attn[n] = r.flops_attn;
    ++n;
  }
  return (int)g_ws_recs.size();
}

}  // extern "C"
