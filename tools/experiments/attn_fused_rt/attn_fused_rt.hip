// The attention branch of a relay-token (H-OSA) transformer block up to the attention output as ONE kernel on the matrix
// cores of gfx950:
//
//     o (rows, C as bf16 split2) = window_attention( qkv( LayerNorm(x) ), relay q / k / v )      q, k, v of the token rows
//                                                                                                never leave the CU
// Replaces norm1 -> attention.qkv -> [hat_window_mask + padded RPE bias, SDPA] of the reference's OctreeAttention inside
// `data = data + attn(norm1(cat(rt, data)))` for the pyramid blocks (models/hotformerloc_backbone.py:197-216,
// models/octformer_backbone.py:52-93): C = 256, 16 heads of 16, K = 48 or 64 tokens + 1 relay token per window, dilation 1.
// These ran as two launches (csrc/qkv_fused.hip, window_attn_kernel_v5 of csrc/attention.hip): q, k, v crossed HBM once each
// way as 4 B per element (24 of the two launches' 32 B per (row, channel)); here a token row costs x in (4 B), the split2
// attention output out (4 B) and 8 B of metadata.  The relay rows -- one per window, 2 % of the rows, available only after
// the relay-token self-attention of the iteration -- still get their LayerNorm + qkv from hfl_ln_qkv_fused (the fp16
// (hi, lo) operand rows it writes are read here, 3 x 64 B per window and head); their attention (one query row against the
// window's keys) runs here, on the VALU.
//
// Arithmetic of the token rows: that of the two launches operation for operation (csrc/attn_fused.hip's header); the three
// 1-D RPE tables (hfl_window_rpe_expand, f16_operand = 2) at every depth.
//
// Dataflow.  A 768-lane workgroup (12 waves, 3 per SIMD) owns 192 consecutive token rows = 4 (K = 48) or 3 (K = 64) windows.
// Wave w keeps LayerNorm(x) of rows 16 w .. 16 w + 15 as MFMA B fragments for the whole tile (64 VGPRs).  Wqkv streams through a
// 3-slot LDS ring in stages of ONE HEAD's 16 features of one region (16 KiB: the 2-KiB halves of hfl_qkv_fused_pack's k-step
// blocks), walked head by head: Q_h, K_h, V_h.  The GEMM epilogue of a stage (bias, query scale, fp16 (hi, lo) split) writes
// the head's LDS image [Q | K | V][192 + windows rows][16 hi | 16 lo]; the images are double-buffered by head parity, so the
// attention of head h -- 12 (window, query tile) units, one per wave, plus one relay-query unit per window on waves 8.. --
// runs behind the barrier of stage Q_(h+1) while nobody waits: one s_barrier per stage, no separate attention phase.
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
typedef _Float16 fr_h8 __attribute__((ext_vector_type(8)));
typedef short fr_s4 __attribute__((ext_vector_type(4)));

constexpr float kRMask = -1e3f;      // models/octree.py:66
constexpr float kRDead = -1e30f;

constexpr int RC = 256;              // channels
constexpr int RH = 16;               // heads
constexpr int RROWS = 192;           // token rows per workgroup tile
constexpr int RW = 12;               // waves
constexpr int RKS = RC / 32;         // k-steps of the qkv GEMM
constexpr int RSTAGE = RC * 64;      // bytes of a weight stage: 8 k-steps x 16 features x 128 B
constexpr int RNSLOT = 3;
constexpr int RSPR = RC / 32;        // 32-feature stages of the pack per region
constexpr int RTSMAX = 768;          // floats of one head's expanded table (three 1-D tables, depth <= 7)
constexpr int RIMROWS = RROWS + 4;   // image rows: the tile's tokens + its windows' relay rows
constexpr int RIMREG = RIMROWS * 64; // bytes of one region (Q, K or V) of a head's image: 64 B per row
constexpr int RIMG = 3 * RIMREG;

struct FusedRtParams {
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
  // work units: tiles [0, full_tiles) whole (all 16 heads), then every later tile cut into `tail_parts` units of
  // RH / tail_parts heads (csrc/attn_fused.hip: no reduction, a part writes its own heads)
  int full_tiles;
  int tail_parts;
  int depth;
  int batch;
  float eps;
  float q_scale;
  int dbg;                         // probe knob 'attn_fused_rt_dbg' (timing ablations, wrong results): 1 no token units, 2 no relay
                                   // units, 4 no GEMM k-loop, 8 no weight stream
};

__device__ __forceinline__ float fr_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float fr_rows_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float fr_rows_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ void fr_split_pair_f16(float p0, float p1, unsigned int& hi, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(p1));
  lo = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}
// fp16 (hi, lo) pair of packed halves -> two floats each
__device__ __forceinline__ float fr_h2f(unsigned int packed, int half) {
  const unsigned short h = half ? (unsigned short)(packed >> 16) : (unsigned short)(packed & 0xFFFFu);
  return (float)__builtin_bit_cast(_Float16, h);
}

#define FR_DS_WRITE64(addr, val)                                                                                    \
  {                                                                                                                  \
    const u32x2 v__ = {(val).x, (val).y};                                                                            \
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v__) : "memory");                                            \
  }
// two fragment reads (hi, lo of one k-step) and counted waits: LDS operations of a wave return in order
#define FR_LDS_READ2(f0, f1, ahi, alo, off) \
  asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" : "=&v"(f0), "=&v"(f1) : "v"(ahi), "v"(alo), "n"(off))
#define FR_LDS_WAIT2(f0, f1, n) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0), "+v"(f1) : "n"(n))

// The weight ring is its OWN array: hipcc tracks LDS-DMA destinations by array, so the plain C++ LDS reads of the attention
// code get no `s_waitcnt vmcnt(0)` in front of them (csrc/attn_fused.hip).
__shared__ __attribute__((aligned(1024))) unsigned char r_ring[RNSLOT * RSTAGE];
// r_lds: [tables of two heads 2 x TSMAX f32 | images of two heads 2 x RIMG | s_qry int4 x 192 | s_key int2 x 192 | s_kbid int x 192 |
//         bias 3C f32 | gamma C | beta C | per-wave output staging 12 x 1 KiB]
constexpr int RL_TAB = 0;
constexpr int RL_IMG = RL_TAB + 2 * RTSMAX * 4;
constexpr int RL_QRY = RL_IMG + 2 * RIMG;
constexpr int RL_KEY = RL_QRY + RROWS * 16;
constexpr int RL_KBID = RL_KEY + RROWS * 8;
constexpr int RL_BIAS = RL_KBID + RROWS * 4;
constexpr int RL_GAMMA = RL_BIAS + 3 * RC * 4;
constexpr int RL_BETA = RL_GAMMA + RC * 4;
constexpr int RL_STG = RL_BETA + RC * 4;
constexpr int RL_END = RL_STG + RW * 1024;
__shared__ __attribute__((aligned(1024))) unsigned char r_lds[RL_END];
static_assert(RNSLOT * RSTAGE + RL_END <= 160 * 1024, "LDS budget of one workgroup per CU");

// 16-B chunk `ch` (0, 1: hi; 2, 3: lo) of image row r: slot ch ^ sw(r), sw = {0, 2, 3, 1}[(r >> 2) & 3] -- conflict-free for the
// operand reads (lane (c, g): row c, chunk g), the transposed V reads (rows 4 g + (c >> 2), 8 B of the hi or lo half) and the
// epilogue's 8-B writes (16 consecutive rows, one chunk)
__device__ __forceinline__ int fr_sw(int r) { return (0x78 >> ((r >> 1) & 6)) & 3; }

template <int FK, int RPE>
__global__ void __launch_bounds__(RW * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
attn_fused_rt_kernel(const FusedRtParams p) {
  constexpr int FT = FK / 16;              // 16-row tiles per window
  constexpr int NWIN = RROWS / FK;         // windows per workgroup tile
  constexpr int NP = (FT + 2) / 2;         // pairs of key tiles, the relay key's tile included
  static_assert(NWIN * FT == RW, "one (window, query tile) unit per wave and head");
  typedef __attribute__((address_space(3))) const float lds_f32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;          // GEMM: row of the wave's tile, k / feature quarter
  const int c = lane & 15, g = lane >> 4;            // attention: column of a 16-tile, 4-row group
  float* s_tab = reinterpret_cast<float*>(r_lds + RL_TAB);
  unsigned char* s_img = r_lds + RL_IMG;
  int4* s_qry = reinterpret_cast<int4*>(r_lds + RL_QRY);
  int2* s_key = reinterpret_cast<int2*>(r_lds + RL_KEY);
  int* s_kbid = reinterpret_cast<int*>(r_lds + RL_KBID);
  float* bs = reinterpret_cast<float*>(r_lds + RL_BIAS);
  float* gms = reinterpret_cast<float*>(r_lds + RL_GAMMA);
  float* bts = reinterpret_cast<float*>(r_lds + RL_BETA);
  unsigned char* stg = r_lds + RL_STG + wave * 1024;

  const int R = (1 << p.depth) - 1, W = 2 * R + 1;
  const int TS = RPE ? ((3 * W + 3) & ~3) : 0;

  for (int i = tid; i < 3 * RC / 4; i += RW * 64) reinterpret_cast<float4*>(bs)[i] = reinterpret_cast<const float4*>(p.bias)[i];
  for (int i = tid; i < RC / 4; i += RW * 64) {
    reinterpret_cast<float4*>(gms)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
    reinterpret_cast<float4*>(bts)[i] = reinterpret_cast<const float4*>(p.beta)[i];
  }
  __syncthreads();

  // ---- weight ring: a stage = 16 pieces of 1 KiB, two each from waves 0..7 (wave w: k-step w, the 2 KiB of the head's 16
  // features inside the pack's 4-KiB k-step block of 32 features); two stages ahead of the consumed one
  const bool loader = wave < 8;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  int h0 = 0, nst_cur = 3 * RH;              // current unit: first head, stages (3 per head: Q, K, V)
  auto issue = [&](int n, int slot) {
    if (!loader || (p.dbg & 8)) return;
    const int hd = h0 + n / 3, reg = n % 3;
    const unsigned char* s = p.pack + (int64_t)(reg * RSPR + (hd >> 1)) * (RC * 128) + wave * 4096 + (hd & 1) * 2048 + lane_off;
    unsigned char* d = r_ring + slot * RSTAGE + wave * 2048;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)d, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                     (__attribute__((address_space(3))) void*)d, 16, 1024, 0);
  };
  uint32_t seq = 0;
  auto acquire = [&](int n) -> const unsigned char* {
    if (loader) {
      if (n + 1 < nst_cur) HFL_WAIT_VM(2);
      else HFL_WAIT_VM(0);
    }
    // (this wave's image / table writes of the previous stage are performed before the others read them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (n + 2 < nst_cur) issue(n + 2, (int)((seq + 2) % RNSLOT));
    const unsigned char* st = r_ring + (seq % RNSLOT) * RSTAGE;
    ++seq;
    return st;
  };
  // A fragment of a stage: row = feature fr, hi chunk fq, lo chunk 4 + fq of k-step ks (2 KiB each); 16-B slot t of row r
  // stored at slot t ^ ((r >> 1) & 7) (the pack's swizzle: rows 16 .. 31 of a 32-row block repeat the pattern of rows 0 .. 15)
  const int off_hi = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), off_lo = off_hi ^ 64;

  // attention-side constants (v5's mappings): per-wave staging block of 16 rows x 64 B, 16-B chunk j of row r in slot
  // j ^ ((r >> 2) & 3)
  const int st_quad = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const int st_row = c * 64, st_x = (c >> 2) & 3;
  const float mask2 = kRMask * 1.4426950408889634f;
  const int n_tok = (int)p.n_tokens;

  // token unit of this wave: window wl of the tile, query tile qt
  const int wl = wave / FT, qt = wave % FT;
  const int wbase = wl * FK;

  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    int tile = unit, nh = RH;
    h0 = 0;
    if (unit >= p.full_tiles) {
      const int v = unit - p.full_tiles;
      tile = p.full_tiles + v / p.tail_parts;
      nh = RH / p.tail_parts;
      h0 = (v % p.tail_parts) * nh;
    }
    const int nst = 3 * nh;
    const int row0 = tile * RROWS;
    // ---- LayerNorm of this wave's 16 rows -> B-operand fragments (lane: row fr, channels 32 ks + 8 fq + j)
    bf16x8 xh[RKS], xl[RKS];
    {
      int r = row0 + wave * 16 + fr;
      if (r >= n_tok) r = n_tok - 1;
      const float* xr = p.x + (int64_t)r * RC + fq * 8;
      float4 a[RKS][2];
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < RKS; ++ks) {
        a[ks][0] = *reinterpret_cast<const float4*>(xr + ks * 32);
        a[ks][1] = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        sum += ((a[ks][0].x + a[ks][0].y) + (a[ks][0].z + a[ks][0].w)) + ((a[ks][1].x + a[ks][1].y) + (a[ks][1].z + a[ks][1].w));
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / (float)RC);
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < RKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[ks][h].x -= mean; a[ks][h].y -= mean; a[ks][h].z -= mean; a[ks][h].w -= mean;
          sq += (a[ks][h].x * a[ks][h].x + a[ks][h].y * a[ks][h].y) + (a[ks][h].z * a[ks][h].z + a[ks][h].w * a[ks][h].w);
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)RC) + p.eps);
#pragma unroll
      for (int ks = 0; ks < RKS; ++ks) {
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
        // (pinned here: left alone hipcc SINKS this arithmetic below the barrier and the metadata block, towards its first use in
        // the stage loop, and carries the 64 row registers + 128 of gamma / beta there -- through scratch)
        asm volatile("" : "+v"(xh[ks]), "+v"(xl[ks]));
      }
    }
    // (every wave has left the previous unit's last attention -- its metadata, its images -- and its ring reads)
    __builtin_amdgcn_s_barrier();
    // ---- metadata of the tile's 192 rows: query side {4 x, 4 y | 4 z << 16, batch id, global row}, key side
    // {4 (R - x), 4 (W + R - y) | 4 (2 W + R - z) << 16}, batch id (-1: the row does not exist)
    if (tid < RROWS) {
      const int t = row0 + tid;
      int bid = -1, row = -1, x = 0, y = 0, z = 0;
      if (t < n_tok) {
        const uint2 mt = *reinterpret_cast<const uint2*>(p.meta + 2 * (int64_t)t);
        x = (int)(mt.x & 1023u); y = (int)((mt.x >> 10) & 1023u); z = (int)(mt.x >> 20);
        bid = (int)mt.y;
        row = t;
      }
      s_key[tid] = make_int2(4 * (R - x), (4 * (W + R - y)) | ((4 * (2 * W + R - z)) << 16));
      s_qry[tid] = make_int4(4 * x, (4 * y) | ((4 * z) << 16), bid, row);
      s_kbid[tid] = bid;
    }
    nst_cur = nst;
    issue(0, (int)(seq % RNSLOT));
    issue(1, (int)((seq + 1) % RNSLOT));

    // ---- the attention of head hd from image `img` and table `tab`: this wave's token unit, and on waves 8 .. the relay
    // query of window wave - 8
    auto attention = [&](int hd) {
      const unsigned char* img = s_img + (hd & 1) * RIMG;
      auto iaddr = [&](int reg, int r, int ch) -> const unsigned char* {
        return img + reg * RIMREG + r * 64 + ((ch ^ fr_sw(r)) << 4);
      };
      const int tabb = (int)(uintptr_t)(s_tab + (hd & 1) * RTSMAX);
      const int rrow = RROWS + wl;                          // image row of its relay token
      const int bid0 = s_kbid[wbase], bidl = s_kbid[wbase + FK - 1];
      const bool homog = __builtin_amdgcn_readfirstlane((bidl >= 0 && bid0 == bidl) ? 1 : 0) != 0;
      const int rt_bid = bid0 >= 0 ? bid0 : p.batch;        // the relay token carries the id of the window's first token
      if (!(p.dbg & 1)) {
        // ---------------- token unit: queries 16 qt .. 16 qt + 15 of window wl against its FK keys + the relay key
        __builtin_amdgcn_sched_barrier(0);
        uint4 ka[FT + 1], qh, ql;
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) ka[kt] = *reinterpret_cast<const uint4*>(iaddr(1, wbase + kt * 16 + c, g));
        ka[FT] = make_uint4(0u, 0u, 0u, 0u);               // relay key tile: one live key, position 0
        if (c == 0) ka[FT] = *reinterpret_cast<const uint4*>(iaddr(1, rrow, g));
        {
          const int r = wbase + qt * 16 + c;
          qh = *reinterpret_cast<const uint4*>(iaddr(0, r, g & 1));
          ql = *reinterpret_cast<const uint4*>(iaddr(0, r, 2 + (g & 1)));
        }
        const int4 qm = s_qry[wbase + qt * 16 + c];
        const int q_bid = qm.z;
        const int qxa = qm.x + tabb, qyza = qm.y + tabb * 0x10001;

        f32x4 sc[FT + 1];
#pragma unroll
        for (int kt = 0; kt <= FT; ++kt) {
          const fr_h8 ak = __builtin_bit_cast(fr_h8, ka[kt]);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(fr_h8, ql), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(fr_h8, qh), acc, 0, 0, 0);
          sc[kt] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (RPE) {               // no RPE for the relay column (octformer_backbone.py:78-80)
#pragma unroll
          for (int kt = 0; kt < FT; ++kt) {
            f32x4 bx, by, bz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int2 km = s_key[wbase + kt * 16 + 4 * g + r];
              const uint32_t t = (uint32_t)(km.y + qyza);                  // both halves are LDS byte addresses
              bx[r] = *reinterpret_cast<lds_f32*>(km.x + qxa);
              by[r] = *reinterpret_cast<lds_f32*>((int)(t & 0xFFFFu));
              bz[r] = *reinterpret_cast<lds_f32*>((int)(t >> 16));
            }
            sc[kt] += (bx + by) + bz;
          }
        }
        if (!homog) {
#pragma unroll
          for (int kt = 0; kt < FT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (s_kbid[wbase + kt * 16 + 4 * g + r] != q_bid) sc[kt][r] += mask2;
        }
        __builtin_amdgcn_sched_barrier(0);
        // relay key: tile FT, key 0 = register 0 of the g == 0 lanes
        float srt = sc[FT][0] + (g == 0 ? 0.f : kRDead);
        if (!homog && rt_bid != q_bid) srt += mask2;
        float mx = srt;
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
          mx = fr_max3(mx, sc[kt][0], sc[kt][1]);
          mx = fr_max3(mx, sc[kt][2], sc[kt][3]);
        }
        mx = fr_rows_max(mx);
        const f32x4 nmx4 = {-mx, -mx, -mx, -mx};
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
          f32x4 e = sc[kt] + nmx4;
          e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
          e[2] = __builtin_amdgcn_exp2f(e[2]); e[3] = __builtin_amdgcn_exp2f(e[3]);
          sc[kt] = e;
          sum4 += e;
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        const float ert = __builtin_amdgcn_exp2f(srt - mx);               // zero in the lanes g != 0
        sum += ert;
        sum = fr_rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          // V^T fragments of this pair of key tiles: lane 4 q + p of a 16-lane group addresses key q of its 4-key block, dims
          // 4 p .. 4 p + 3; the relay key's tile: key 0 only (lanes g == 0, element 0).  (Loaded here, pair by pair, and the
          // phases of the unit fenced for the scheduler: with every fragment of the unit requested up front the unit needs ~50
          // more registers than the 104 the row fragments leave, and hipcc then parks the row fragments in scratch.)
          fr_h8 vhi_p, vlo_p;
          {
            typedef __attribute__((address_space(3))) fr_s4 lds_s4;
            const int kk = 4 * g + (c >> 2), cp = c & 3;
            fr_s4 hh[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, ll[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int kt = 2 * pp + e;
              if (kt < FT) {
                const int r = wbase + kt * 16 + kk;
                hh[e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r, cp >> 1) + (cp & 1) * 8));
                ll[e] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r, 2 + (cp >> 1)) + (cp & 1) * 8));
              } else if (kt == FT) {
                // element 0 of the g == 0 lanes = V[relay][dim c] (hi: bytes 2 c of chunks 0, 1; lo: of chunks 2, 3)
                const unsigned short vh = *reinterpret_cast<const unsigned short*>(iaddr(2, rrow, c >> 3) + (c & 7) * 2);
                const unsigned short vl = *reinterpret_cast<const unsigned short*>(iaddr(2, rrow, 2 + (c >> 3)) + (c & 7) * 2);
                hh[e][0] = g == 0 ? (short)vh : (short)0;
                ll[e][0] = g == 0 ? (short)vl : (short)0;
              }
            }
            const short __attribute__((ext_vector_type(8))) h8 = {hh[0][0], hh[0][1], hh[0][2], hh[0][3], hh[1][0], hh[1][1], hh[1][2], hh[1][3]};
            const short __attribute__((ext_vector_type(8))) l8 = {ll[0][0], ll[0][1], ll[0][2], ll[0][3], ll[1][0], ll[1][1], ll[1][2], ll[1][3]};
            vhi_p = __builtin_bit_cast(fr_h8, h8);
            vlo_p = __builtin_bit_cast(fr_h8, l8);
          }
          f32x4 oa = {0.f, 0.f, 0.f, 0.f};
          unsigned int hh0 = 0u, hh1 = 0u, hh2 = 0u, hh3 = 0u, ll0 = 0u, ll1 = 0u, ll2 = 0u, ll3 = 0u;
          if (2 * pp < FT) {
            fr_split_pair_f16(sc[2 * pp][0], sc[2 * pp][1], hh0, ll0);
            fr_split_pair_f16(sc[2 * pp][2], sc[2 * pp][3], hh1, ll1);
          } else {                                       // the relay key's tile leads the pair
            fr_split_pair_f16(ert, 0.f, hh0, ll0);
            ll0 &= 0xFFFFu;
          }
          if (2 * pp + 1 < FT) {
            fr_split_pair_f16(sc[2 * pp + 1][0], sc[2 * pp + 1][1], hh2, ll2);
            fr_split_pair_f16(sc[2 * pp + 1][2], sc[2 * pp + 1][3], hh3, ll3);
          } else if (2 * pp + 1 == FT) {
            fr_split_pair_f16(ert, 0.f, hh2, ll2);
            ll2 &= 0xFFFFu;
          }
          const u32x4 uh = {hh0, hh1, hh2, hh3}, ul = {ll0, ll1, ll2, ll3};
          const fr_h8 phi = __builtin_bit_cast(fr_h8, uh), plo = __builtin_bit_cast(fr_h8, ul);
          oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi_p, plo, oa, 0, 0, 0);
          oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vlo_p, phi, oa, 0, 0, 0);
          oa = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi_p, phi, oa, 0, 0, 0);
          o += oa;
          __builtin_amdgcn_sched_barrier(0);
        }
        o *= inv;
        // split2 rows for the proj GEMM through the wave's staging block: one 16-B store per lane, a quad of lanes per row
        {
          uint2 hi, lo;
          x3_split_pair_scalar(o[0], o[1], hi.x, lo.x);
          x3_split_pair_scalar(o[2], o[3], hi.y, lo.y);
          FR_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((g >> 1) ^ st_x) * 16) + (g & 1) * 8), hi);
          FR_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((2 + (g >> 1)) ^ st_x) * 16) + (g & 1) * 8), lo);
          uint4 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(stg + st_quad)) : "memory");
          const int rl = lane >> 2, ch = lane & 3;
          const int orow_l = s_qry[wbase + qt * 16 + rl].w;
          if (orow_l >= 0)
            *reinterpret_cast<uint4*>(p.out + (size_t)orow_l * (uint32_t)(4 * RC) +
                                      (uint32_t)((hd >> 1) * 128 + (hd & 1) * 32 + (ch & 1) * 16 + (ch >> 1) * 64)) = v;
        }
      }
      if (wave >= 8 && wave - 8 < NWIN) {
        // ---------------- relay query of window wave - 8 on the VALU: lane j < FK = token key j; the relay key (the 49th / 65th
        // key of the window) is computed by every lane alike and enters the sums as a wave-uniform term
        const int w2 = wave - 8;
        const int wg2 = tile * NWIN + w2;
        const int wb2 = w2 * FK, rr2 = RROWS + w2;
        if (wg2 < p.n_windows && !(p.dbg & 2)) {
          const int b0 = s_kbid[wb2], bl = s_kbid[wb2 + FK - 1];
          const bool hom2 = (bl >= 0 && b0 == bl);
          const int rtb2 = b0 >= 0 ? b0 : p.batch;
          const int krow = wb2 + (lane < FK ? lane : 0);
          float s = 0.f, srt = 0.f;
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            const uint4 qh4 = *reinterpret_cast<const uint4*>(iaddr(0, rr2, hlf));
            const uint4 ql4 = *reinterpret_cast<const uint4*>(iaddr(0, rr2, 2 + hlf));
            const uint4 kh4 = *reinterpret_cast<const uint4*>(iaddr(1, krow, hlf));
            const uint4 kl4 = *reinterpret_cast<const uint4*>(iaddr(1, krow, 2 + hlf));
            const uint4 rh4 = *reinterpret_cast<const uint4*>(iaddr(1, rr2, hlf));
            const uint4 rl4 = *reinterpret_cast<const uint4*>(iaddr(1, rr2, 2 + hlf));
            const unsigned int qhw[4] = {qh4.x, qh4.y, qh4.z, qh4.w}, qlw[4] = {ql4.x, ql4.y, ql4.z, ql4.w};
            const unsigned int khw[4] = {kh4.x, kh4.y, kh4.z, kh4.w}, klw[4] = {kl4.x, kl4.y, kl4.z, kl4.w};
            const unsigned int rhw[4] = {rh4.x, rh4.y, rh4.z, rh4.w}, rlw[4] = {rl4.x, rl4.y, rl4.z, rl4.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float qv = fr_h2f(qhw[e >> 1], e & 1) + fr_h2f(qlw[e >> 1], e & 1);
              s = fmaf(qv, fr_h2f(khw[e >> 1], e & 1) + fr_h2f(klw[e >> 1], e & 1), s);
              srt = fmaf(qv, fr_h2f(rhw[e >> 1], e & 1) + fr_h2f(rlw[e >> 1], e & 1), srt);
            }
          }
          if (lane >= FK) s = kRDead;
          else if (!hom2 && s_kbid[wb2 + lane] != rtb2) s += mask2;
          const float mx = __builtin_fmaxf(hfl_group_max<64>(s), srt);       // (the relay key carries the relay query's own id: no mask)
          const float e = __builtin_amdgcn_exp2f(s - mx);
          const float ert = __builtin_amdgcn_exp2f(srt - mx);
          const float inv = __builtin_amdgcn_rcpf(hfl_group_sum<64>(e) + ert);
          __builtin_amdgcn_sched_barrier(0);
          float vv[16];
#pragma unroll
          for (int hlf = 0; hlf < 2; ++hlf) {
            const uint4 vh4 = *reinterpret_cast<const uint4*>(iaddr(2, krow, hlf));
            const uint4 vl4 = *reinterpret_cast<const uint4*>(iaddr(2, krow, 2 + hlf));
            const unsigned int vhw[4] = {vh4.x, vh4.y, vh4.z, vh4.w}, vlw[4] = {vl4.x, vl4.y, vl4.z, vl4.w};
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) vv[hlf * 8 + e2] = fr_h2f(vhw[e2 >> 1], e2 & 1) + fr_h2f(vlw[e2 >> 1], e2 & 1);
          }
          // O[d] = sum over the lanes of e v[d]: halving butterfly -- after the steps 32, 16, 8, 4 a lane holds ONE dim's partial
          // sum over its residue class, dim = 8 b5 + 4 b4 + 2 b3 + b2 (b_i = bit i of the lane); then the steps 2, 1
          float t8[8], t4[4], t2[2], t1;
          const bool u5 = (lane & 32) != 0, u4 = (lane & 16) != 0, u3 = (lane & 8) != 0, u2 = (lane & 4) != 0;
#pragma unroll
          for (int d = 0; d < 8; ++d) {
            const float mine = e * vv[u5 ? 8 + d : d], give = e * vv[u5 ? d : 8 + d];
            t8[d] = mine + __shfl_xor(give, 32, 64);
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float mine = t8[u4 ? 4 + d : d], give = t8[u4 ? d : 4 + d];
            t4[d] = mine + __shfl_xor(give, 16, 64);
          }
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            const float mine = t4[u3 ? 2 + d : d], give = t4[u3 ? d : 2 + d];
            t2[d] = mine + __shfl_xor(give, 8, 64);
          }
          {
            const float mine = t2[u2 ? 1 : 0], give = t2[u2 ? 0 : 1];
            t1 = mine + __shfl_xor(give, 4, 64);
          }
          t1 += __shfl_xor(t1, 2, 64);
          t1 += __shfl_xor(t1, 1, 64);
          if ((lane & 3) == 0) {
            const int d = (lane >> 2) & 15;              // = 8 b5 + 4 b4 + 2 b3 + b2
            const unsigned short rvh = *reinterpret_cast<const unsigned short*>(iaddr(2, rr2, d >> 3) + (d & 7) * 2);
            const unsigned short rvl = *reinterpret_cast<const unsigned short*>(iaddr(2, rr2, 2 + (d >> 3)) + (d & 7) * 2);
            t1 = fmaf(ert, (float)__builtin_bit_cast(_Float16, rvh) + (float)__builtin_bit_cast(_Float16, rvl), t1) * inv;
            const uint32_t hb = x3_bf16_rne(t1);
            const uint32_t lb = x3_bf16_rne(t1 - __uint_as_float(hb << 16));
            unsigned char* orow = p.out + (size_t)(p.rt_row0 + wg2) * (uint32_t)(4 * RC) + (uint32_t)((hd >> 1) * 128 + (hd & 1) * 32 + d * 2);
            *reinterpret_cast<unsigned short*>(orow) = (unsigned short)hb;
            *reinterpret_cast<unsigned short*>(orow + 64) = (unsigned short)lb;
          }
        }
      }
    };

#pragma unroll 1
    for (int hh = 0; hh < nh; ++hh) {
      const int hd = h0 + hh;
      unsigned char* img = s_img + (hd & 1) * RIMG;
#pragma unroll 1
      for (int reg = 0; reg < 3; ++reg) {
        const unsigned char* st = acquire(3 * hh + reg);
        if (reg == 0 && hh > 0) attention(hd - 1);          // (image and tables of the previous head are complete: the barrier)
        // ---- q / k / v of head hd for this wave's 16 rows: one stage of 16 features, 8 k-steps
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const uint32_t ahi = (uint32_t)(uintptr_t)(st + off_hi), alo = (uint32_t)(uintptr_t)(st + off_lo);
        bf16x8 wf[4][2];
        if (!(p.dbg & 4)) {
          FR_LDS_READ2(wf[0][0], wf[0][1], ahi, alo, 0);
          FR_LDS_READ2(wf[1][0], wf[1][1], ahi, alo, 2048);
        }
        if (!(p.dbg & 4))
        hfl_static_for(std::make_integer_sequence<int, RKS>{}, [&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          if constexpr (ks + 2 < RKS) {
            FR_LDS_READ2(wf[(ks + 2) & 3][0], wf[(ks + 2) & 3][1], ahi, alo, (ks + 2) * 2048);
            FR_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 4);
          } else if constexpr (ks + 1 < RKS) {
            FR_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 2);
          } else {
            FR_LDS_WAIT2(wf[ks & 3][0], wf[ks & 3][1], 0);
          }
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][0], xl[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][1], xh[ks], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 3][0], xh[ks], acc, 0, 0, 0);
        });
        // epilogue: bias, query scale, fp16 (hi, lo) split (csrc/qkv_fused.hip / gemm_x3's EPI 2) into the head's image: row
        // 16 wave + fr, features 4 fq .. 4 fq + 3: hi 8 B of chunk fq >> 1, lo of chunk 2 + (fq >> 1)
        {
          f32x4 b;
          const uint32_t baddr = (uint32_t)(uintptr_t)(bs + reg * RC + hd * 16 + fq * 4);
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b) : "v"(baddr), "v"(acc));
          float v0 = acc[0] + b[0], v1 = acc[1] + b[1], v2 = acc[2] + b[2], v3 = acc[3] + b[3];
          if (reg == 0) { v0 *= p.q_scale; v1 *= p.q_scale; v2 *= p.q_scale; v3 *= p.q_scale; }
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(v0, v1), h23 = __builtin_amdgcn_cvt_pkrtz(v2, v3);
          const auto l01 = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1]);
          const auto l23 = __builtin_amdgcn_cvt_pkrtz(v2 - (float)h23[0], v3 - (float)h23[1]);
          const uint2 hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
          const uint2 lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
          const int trow = wave * 16 + fr;
          unsigned char* ir = img + reg * RIMREG + trow * 64;
          const int sw = fr_sw(trow);
          FR_DS_WRITE64((uint32_t)(uintptr_t)(ir + (((fq >> 1) ^ sw) << 4) + (fq & 1) * 8), hi);
          FR_DS_WRITE64((uint32_t)(uintptr_t)(ir + (((2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8), lo);
        }
        // the head's table and the relay rows of the tile's windows (waves 8 .. 11: they carry no LDS-DMA, so their loads'
        // waits do not drain the ring): during the Q stage, into the buffers of this head's parity
        if (reg == 0 && !loader) {
          const int t2 = tid - 8 * 64;                      // 0 .. 255
          if (RPE) {
            const float4* src = reinterpret_cast<const float4*>(p.rpe2 + (size_t)hd * TS);
            float4* dst = reinterpret_cast<float4*>(s_tab + (hd & 1) * RTSMAX);
            for (int i = t2; i < TS / 4; i += (RW - 8) * 64) dst[i] = src[i];
          }
          if (t2 < NWIN * 12) {                             // window t2 / 12, region (t2 / 4) % 3, 16-B chunk t2 % 4
            const int w2 = t2 / 12, rg = (t2 >> 2) % 3, ch = t2 & 3;
            const int wg2 = tile * NWIN + w2;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (wg2 < p.n_windows)
              v = *reinterpret_cast<const uint4*>(p.relay_qkv + (size_t)wg2 * (3 * RC * 4) + rg * (RC * 4) + hd * 64 + ch * 16);
            const int r = RROWS + w2;
            *reinterpret_cast<uint4*>(img + rg * RIMREG + r * 64 + ((ch ^ fr_sw(r)) << 4)) = v;
          }
        }
      }
    }
    // the last head of the unit: publish its image, then its attention
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    attention(h0 + nh - 1);
  }
}

}  // namespace

// per-launch HIP events for bench.py's roofline leg (the launches sit inside hfl_block_forward_x3: no Python timer sees them)
struct FusedRtTimingRec {
  hipEvent_t e0, e1;
  double bytes, flops_gemm, flops_attn;
};
static int g_rt_timing = 0;
static std::vector<FusedRtTimingRec> g_rt_recs;
static std::mutex g_rt_mu;
static int g_rt_split = 1;      // probe knob 'attn_fused_rt_split'
static int g_rt_dbg = 0;        // probe knob 'attn_fused_rt_dbg'
extern "C" void hfl_internal_set_attn_fused_rt_dbg(int v) { g_rt_dbg = v; }
extern "C" void hfl_internal_set_attn_fused_rt_split(int v) { g_rt_split = v ? 1 : 0; }

extern "C" {

/* 1 when hfl_attn_fused_rt_fwd takes this configuration (see include/hotformerloc_hip.h), else 0 */
int hfl_attn_fused_rt_ok(const hfl_window_attn_desc* d, int channels) {
  if (d == nullptr || channels != RC || d->n_heads != RH || d->n_relay != 1 || d->dilation != 1) return 0;
  if (d->patch_size != 48 && d->patch_size != 64) return 0;
  if (d->depth < 1 || d->depth > 7) return 0;
  if (d->n_tokens <= 0 || d->n_windows <= 0 || d->rt_row0 < d->n_tokens) return 0;
  if ((d->rt_row0 + d->n_windows) >= ((int64_t)1 << 31) / (4 * RC)) return 0;
  if ((int64_t)d->n_windows * d->patch_size < d->n_tokens) return 0;
  return 1;
}

int hfl_attn_fused_rt_fwd(void* out_split2, const float* x, const float* gamma, const float* beta, float eps, const void* qkv_pack,
                          const float* qkv_bias, float q_scale, const void* relay_qkv, const uint32_t* tok_meta,
                          const float* rpe_tables3, const hfl_window_attn_desc* d, hfl_stream_t stream) {
  if (out_split2 == nullptr || x == nullptr || gamma == nullptr || beta == nullptr || qkv_pack == nullptr ||
      qkv_bias == nullptr || relay_qkv == nullptr || tok_meta == nullptr || d == nullptr)
    return HFL_EINVAL;
  if (!hfl_attn_fused_rt_ok(d, RC)) return HFL_EINVAL;
  FusedRtParams p;
  p.out = static_cast<unsigned char*>(out_split2); p.x = x; p.gamma = gamma; p.beta = beta;
  p.pack = static_cast<const unsigned char*>(qkv_pack); p.bias = qkv_bias; p.meta = tok_meta;
  p.rpe2 = rpe_tables3;
  p.relay_qkv = static_cast<const unsigned char*>(relay_qkv);
  p.n_tokens = d->n_tokens; p.rt_row0 = d->rt_row0; p.n_windows = d->n_windows;
  // (tiles cover every window of the plan: the windows past the last token -- the reference pads the token stream to a multiple
  // of K x the stage dilation, models/octree.py:73-75 -- have a relay row too)
  const int64_t t_tok = hfl_cdiv(d->n_tokens, RROWS), t_win = hfl_cdiv(d->n_windows, RROWS / d->patch_size);
  p.n_tiles = (int)(t_tok > t_win ? t_tok : t_win); p.depth = d->depth;
  p.batch = d->batch_size; p.eps = eps; p.q_scale = q_scale; p.dbg = g_rt_dbg;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cus = hfl_stream_cus(s);
  // whole rounds of the grid take whole tiles; the tiles left over are cut by heads when that lets them share a round
  p.full_tiles = p.n_tiles;
  p.tail_parts = 1;
  if (g_rt_split) {
    const int full = p.n_tiles / cus * cus, rem = p.n_tiles - full;
    int parts = 1;
    while (parts * 2 <= RH / 2 && rem * parts * 2 <= cus) parts *= 2;
    if (rem > 0 && parts > 1) {
      p.full_tiles = full;
      p.tail_parts = parts;
    }
  }
  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  const int grid = n_units < cus ? n_units : cus;
  FusedRtTimingRec rec{};
  const bool timed = g_rt_timing != 0;
  if (timed) {
    // algorithmic bytes: x in + split2 out = 8 B per (token row, channel) + 8 B of metadata per token + the relay rows' operand
    // rows in and split2 rows out; useful flop: the qkv GEMM 2 M C 3C of the token rows and the attention core 4 L^2 C per window
    const double L = d->patch_size + 1;
    rec.bytes = (double)d->n_tokens * RC * 8.0 + (double)d->n_tokens * 8.0 + (double)d->n_windows * RC * 16.0;
    rec.flops_gemm = 6.0 * (double)d->n_tokens * RC * RC;
    rec.flops_attn = 4.0 * L * L * RC * (double)d->n_windows;
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess || hipEventRecord(rec.e0, s) != hipSuccess) {
      if (rec.e0) (void)hipEventDestroy(rec.e0);
      if (rec.e1) (void)hipEventDestroy(rec.e1);
      return HFL_EINVAL;
    }
  }
  const bool rpe = p.rpe2 != nullptr;
  if (d->patch_size == 48) {
    if (rpe) attn_fused_rt_kernel<48, 1><<<grid, RW * 64, 0, s>>>(p);
    else attn_fused_rt_kernel<48, 0><<<grid, RW * 64, 0, s>>>(p);
  } else {
    if (rpe) attn_fused_rt_kernel<64, 1><<<grid, RW * 64, 0, s>>>(p);
    else attn_fused_rt_kernel<64, 0><<<grid, RW * 64, 0, s>>>(p);
  }
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_rt_mu);
    g_rt_recs.push_back(rec);
  }
  HFL_RETURN_LAST_ERROR();
}

// bench.py: per-launch timing of hfl_attn_fused_rt_fwd on / off (both drop what was recorded) ...
int hfl_internal_fused_rt_timing(int on) {
  std::lock_guard<std::mutex> lk(g_rt_mu);
  for (auto& r : g_rt_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_rt_recs.clear();
  g_rt_timing = on ? 1 : 0;
  return HFL_OK;
}
// ... and read it: per launch the duration (ms), algorithmic bytes, useful GEMM and attention flop; returns the launches recorded
int hfl_internal_fused_rt_timing_read(double* ms, double* bytes, double* flops_gemm, double* flops_attn, int cap) {
  std::lock_guard<std::mutex> lk(g_rt_mu);
  int n = 0;
  for (auto& r : g_rt_recs) {
    if (n >= cap) break;
    if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
    ms[n] = t; bytes[n] = r.bytes; flops_gemm[n] = r.flops_gemm; flops_attn[n] = r.flops_attn;
    ++n;
  }
  return (int)g_rt_recs.size();
}

}  // extern "C"
