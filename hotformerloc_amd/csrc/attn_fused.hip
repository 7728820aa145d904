// The attention branch of a transformer block up to the attention output as ONE kernel on the matrix cores of gfx950:
//
//     o (M, C as bf16 split2) = window_attention( qkv( LayerNorm(x) ) )          q, k, v never leave the CU
//
// Replaces norm1 -> attention.qkv -> [mask + RPE bias, SDPA] of the reference's OctreeAttention inside
// `x = x + attn(norm1(x))` (models/octformer_backbone.py:52-93,275-276) for the OctFormer stage (C = 128, 8 heads, K = 48,
// no relay tokens, dilation 1 / 2 / 4), which ran as two launches (csrc/qkv_fused.hip, window_attn_kernel_v5 of
// csrc/attention.hip): there q, k, v crossed HBM once each way as 4 B per element (12 of the branch's 16 B per (row,
// channel)); here the only HBM traffic is x in (4 B), the split2 attention output out (4 B) and 8 B of metadata per token.
//
// Arithmetic: identical to the two launches, operation for operation (LayerNorm with two-pass statistics; qkv as bf16 (hi, lo)
// products x_lo w_hi + x_hi w_lo + x_hi w_hi accumulated in fp32; q, k, v rounded to fp16 (hi, lo) pairs, q pre-scaled by
// scale * log2 e; scores S^T = K Q^T on v_mfma_f32_16x16x32_f16 with all four cross terms; RPE from the expanded tables;
// exp2-domain softmax in fp32; O^T = V^T P^T with P split in registers) -- the outputs are bitwise those of the two launches.
//
// Dataflow.  A 768-lane workgroup (12 waves, 3 per SIMD) owns a tile of 192 consecutive token rows = 4 windows (one K * D pad
// group at dilation 4, models/octree.py:354-369).  Wave w keeps LayerNorm(x) of rows 16 w .. 16 w + 15 as MFMA B fragments
// for the whole tile (32 VGPRs).  Wqkv streams through a 3-slot LDS ring in the stage format of csrc/qkv_fused.hip (32 output
// features x C k-values, 16 KiB), walked in HEAD-PAIR order: stages Q_p, K_p, V_p of heads 2 p, 2 p + 1, then the attention of
// those two heads on the 4 windows -- 24 (window, head, query tile) units over the 12 waves, operands read from a 72-KiB LDS
// image [Q | K | V][192 rows][2 heads x (16 hi | 16 lo) fp16] that the GEMM epilogue wrote -- while the ring already carries
// the next pair's stages.  The ring is its own __shared__ array: hipcc tracks LDS-DMA destinations by array, so the plain
// C++ LDS accesses of the attention phase get no `s_waitcnt vmcnt(0)` in front of them (the ring's fragment reads are inline
// asm, as in the other ring kernels).
#include "hfl_common.h"
#include "x3_math.h"
#include "stage_stream.h"

#include <mutex>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 fa_h8 __attribute__((ext_vector_type(8)));
typedef short fa_s4 __attribute__((ext_vector_type(4)));

constexpr float kFMask = -1e3f;      // models/octree.py:66
constexpr float kFDead = -1e30f;

constexpr int FC = 128;              // channels
constexpr int FH = 8;                // heads
constexpr int FK = 48;               // tokens per window
constexpr int FT = 3;                // 16-row tiles per window
constexpr int FROWS = 192;           // rows per workgroup tile (4 windows)
constexpr int FW = 12;               // waves
constexpr int FKS = FC / 32;         // k-steps of the qkv GEMM
constexpr int FSTAGE = FC * 128;     // bytes of a weight stage: C rows x 128 B (32 output features)
constexpr int FNSLOT = 3;
constexpr int FNST = 3 * FC / 32;    // stages: 32 output features each
constexpr int FSPR = FC / 32;        // stages per region (Q, K, V) = head pairs
constexpr int FTSMAX = 768;          // floats of one head's expanded table (form 2, depth <= 7)
constexpr int FREG = FROWS * 128;    // bytes of one region of the q / k / v image: 192 rows x (2 heads x 64 B)

struct FusedAttnParams {
  unsigned char* out;         // (n_tokens, 2 C) bf16 split2
  const float* x;             // (n_tokens, C) f32
  const float* gamma;
  const float* beta;
  const unsigned char* pack;  // hfl_qkv_fused_pack image of Wqkv
  const float* bias;          // (3 C)
  const uint32_t* meta;       // (n_tokens, 2): x | y << 10 | z << 20, batch id
  const float* rpe2;          // (H, TS) expanded tables, form 2 (three clamped 1-D tables, log2e-prescaled), or null
  int64_t n_tokens;
  int n_tiles;
  // work units: tiles [0, full_tiles) whole (all four head pairs), then every later tile cut into `tail_parts` units of
  // FSPR / tail_parts head pairs -- the tiles left over after the last whole round of the grid fill the chip once more at
  // a fraction of a tile's time instead of starting a round for a few of them (no reduction: a part writes its own heads)
  int full_tiles;
  int tail_parts;
  int D;
  int depth;
  int batch;
  float eps;
  float q_scale;
};

__device__ __forceinline__ float fa_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float fa_rows_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float fa_rows_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ void fa_split_pair_f16(float p0, float p1, unsigned int& hi, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(p1));
  lo = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

typedef unsigned int fa_u32x2 __attribute__((ext_vector_type(2)));
#define FA_DS_WRITE64(addr, val)                                                                                    \
  {                                                                                                                  \
    const fa_u32x2 v__ = {(val).x, (val).y};                                                                         \
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v__) : "memory");                                            \
  }

// The weight ring is its OWN array (see the header comment); everything else lives in f_lds.
__shared__ __attribute__((aligned(1024))) unsigned char f_ring[FNSLOT * FSTAGE];
// f_lds: [tables of the pair 2 x TSMAX f32 | Q, K, V images 3 x FREG | s_qry int4 x 192 | s_key int2 x 192 | s_kbid int x 192 |
//         bias 3C f32 | gamma C | beta C | per-wave output staging 12 x 1 KiB]
constexpr int FL_TAB = 0;
constexpr int FL_IMG = FL_TAB + 2 * FTSMAX * 4;
constexpr int FL_QRY = FL_IMG + 3 * FREG;
constexpr int FL_KEY = FL_QRY + FROWS * 16;
constexpr int FL_KBID = FL_KEY + FROWS * 8;
constexpr int FL_BIAS = FL_KBID + FROWS * 4;
constexpr int FL_GAMMA = FL_BIAS + 3 * FC * 4;
constexpr int FL_BETA = FL_GAMMA + FC * 4;
constexpr int FL_STG = FL_BETA + FC * 4;
constexpr int FL_END = FL_STG + FW * 1024;
__shared__ __attribute__((aligned(1024))) unsigned char f_lds[FL_END];

template <int RPE>
__global__ void __launch_bounds__(FW * 64) __attribute__((amdgpu_waves_per_eu(3, 3)))
attn_fused_kernel(const FusedAttnParams p) {
  typedef __attribute__((address_space(3))) const float lds_f32;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;          // GEMM: row of the wave's tile, k / feature quarter
  const int c = lane & 15, g = lane >> 4;            // attention: column of a 16-tile, 4-row group
  float* s_tab = reinterpret_cast<float*>(f_lds + FL_TAB);
  unsigned char* s_img = f_lds + FL_IMG;
  int4* s_qry = reinterpret_cast<int4*>(f_lds + FL_QRY);
  int2* s_key = reinterpret_cast<int2*>(f_lds + FL_KEY);
  int* s_kbid = reinterpret_cast<int*>(f_lds + FL_KBID);
  float* bs = reinterpret_cast<float*>(f_lds + FL_BIAS);
  float* gms = reinterpret_cast<float*>(f_lds + FL_GAMMA);
  float* bts = reinterpret_cast<float*>(f_lds + FL_BETA);
  unsigned char* stg = f_lds + FL_STG + wave * 1024;

  const int R = (1 << p.depth) - 1, W = 2 * R + 1;
  const int TS = RPE == 2 ? ((3 * W + 3) & ~3) : 0;

  for (int i = tid; i < 3 * FC / 4; i += FW * 64) reinterpret_cast<float4*>(bs)[i] = reinterpret_cast<const float4*>(p.bias)[i];
  for (int i = tid; i < FC / 4; i += FW * 64) {
    reinterpret_cast<float4*>(gms)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
    reinterpret_cast<float4*>(bts)[i] = reinterpret_cast<const float4*>(p.beta)[i];
  }
  __syncthreads();

  // ---- weight ring: 16 pieces of 1 KiB per stage, two each from waves 0..7 (csrc/qkv_fused.hip's protocol with two stages
  // ahead of the consumed one: at acquire(n) stage n has landed for every wave and the slot of stage n - 1 is free)
  constexpr int DPW = 2;
  const bool loader = wave < 8;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // stage n of the head-pair walk -> stage of the pack (features 32 s .. 32 s + 31 of [Q | K | V])
  int st0 = 0, nst_cur = FNST;               // current unit: first stage of the walk, stages
  auto sidx = [&](int n) -> int { return ((st0 + n) % 3) * FSPR + (st0 + n) / 3; };
  auto issue = [&](int n, int slot) {
    if (!loader) return;
    const unsigned char* s = p.pack + (int64_t)sidx(n) * FSTAGE + wave * (DPW * 1024);
    unsigned char* d = f_ring + slot * FSTAGE + wave * (DPW * 1024);
#pragma unroll
    for (int i = 0; i < DPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024 + lane_off),
                                       (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
  };
  uint32_t seq = 0;
  auto acquire = [&](int n) -> const unsigned char* {
    if (loader) {
      if (n + 1 < nst_cur) HFL_WAIT_VM(DPW);
      else HFL_WAIT_VM(0);
    }
    __builtin_amdgcn_s_barrier();
    if (n + 2 < nst_cur) issue(n + 2, (int)((seq + 2) % FNSLOT));
    const unsigned char* st = f_ring + (seq % FNSLOT) * FSTAGE;
    ++seq;
    return st;
  };
  // A fragment of a 16-feature block of a stage: row = block * 16 + fr, hi chunk fq, lo chunk 4 + fq (16-B slot t of row r
  // stored at slot t ^ ((r >> 1) & 7))
  const int off_hi = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), off_lo = off_hi ^ 64;

  // attention-side constants (v5's mappings): per-wave staging block of 16 rows x 64 B, 16-B chunk j of row r in slot
  // j ^ ((r >> 2) & 3): conflict-free for the operand-layout writes and the row-per-quad read-back
  const int st_quad = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const int st_row = c * 64, st_x = (c >> 2) & 3;
  const float mask2 = kFMask * 1.4426950408889634f;
  const int D = p.D;
  const int n_tok = (int)p.n_tokens;

  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  for (int unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    int tile = unit, pr0 = 0, npr = FSPR;
    if (unit >= p.full_tiles) {
      const int v = unit - p.full_tiles;
      tile = p.full_tiles + v / p.tail_parts;
      npr = FSPR / p.tail_parts;
      pr0 = (v % p.tail_parts) * npr;
    }
    const int nst = 3 * npr;                 // stages of this unit; stage n of the unit = stage 3 pr0 + n of the head-pair walk
    const int row0 = tile * FROWS;
    // ---- LayerNorm of this wave's 16 rows -> B-operand fragments (lane: row fr, channels 32 ks + 8 fq + j)
    bf16x8 xh[FKS], xl[FKS];
    {
      int r = row0 + wave * 16 + fr;
      if (r >= n_tok) r = n_tok - 1;
      const float* xr = p.x + (int64_t)r * FC + fq * 8;
      float4 a[FKS][2];
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) {
        a[ks][0] = *reinterpret_cast<const float4*>(xr + ks * 32);
        a[ks][1] = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        sum += ((a[ks][0].x + a[ks][0].y) + (a[ks][0].z + a[ks][0].w)) + ((a[ks][1].x + a[ks][1].y) + (a[ks][1].z + a[ks][1].w));
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / (float)FC);
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[ks][h].x -= mean; a[ks][h].y -= mean; a[ks][h].z -= mean; a[ks][h].w -= mean;
          sq += (a[ks][h].x * a[ks][h].x + a[ks][h].y * a[ks][h].y) + (a[ks][h].z * a[ks][h].z + a[ks][h].w * a[ks][h].w);
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)FC) + p.eps);
#pragma unroll
      for (int ks = 0; ks < FKS; ++ks) {
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
      }
    }
    // (every wave has left the previous tile's last attention phase -- its metadata, its images -- and its ring reads)
    __builtin_amdgcn_s_barrier();
    // ---- metadata of the tile's 192 rows (indexed by tile row): query side {4 x, 4 y | 4 z << 16, batch id, global row},
    // key side {4 (R - x), 4 (W + R - y) | 4 (2 W + R - z) << 16}, batch id (-1: the row does not exist)
    if (tid < FROWS) {
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
    // (the row loads above are consumed, so the ring may start moving; the metadata is published by the barriers of the
    // first stages, long before the first attention phase reads it)
    st0 = 3 * pr0;
    nst_cur = nst;
    issue(0, (int)(seq % FNSLOT));
    issue(1, (int)((seq + 1) % FNSLOT));

#pragma unroll 1
    for (int pr = pr0; pr < pr0 + npr; ++pr) {
      // ---- qkv of heads 2 pr, 2 pr + 1 for this wave's 16 rows: three stages (Q, K, V) of 32 features
#pragma unroll 1
      for (int reg = 0; reg < 3; ++reg) {
        const unsigned char* st = acquire(3 * (pr - pr0) + reg);
        f32x4 h[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const uint32_t ahi = (uint32_t)(uintptr_t)(st + off_hi), alo = (uint32_t)(uintptr_t)(st + off_lo);
        bf16x8 wf[2][4];
        HFL_LDS_READ4_FIRST(wf[0][0], wf[0][1], wf[0][2], wf[0][3], ahi, alo, 0, 2048);
        HFL_LDS_WAIT4(wf[0][0], wf[0][1], wf[0][2], wf[0][3]);
        hfl_static_for(std::make_integer_sequence<int, FKS>{}, [&](auto kc) {
          constexpr int ks = decltype(kc)::value;
          if constexpr (ks + 1 < FKS)
            HFL_LDS_READ4(wf[(ks + 1) & 1][0], wf[(ks + 1) & 1][1], wf[(ks + 1) & 1][2], wf[(ks + 1) & 1][3], ahi, alo,
                          (ks + 1) * 4096, (ks + 1) * 4096 + 2048, wf[ks & 1][0]);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            h[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][2 * i], xl[ks], h[i], 0, 0, 0);
            h[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][2 * i + 1], xh[ks], h[i], 0, 0, 0);
            h[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][2 * i], xh[ks], h[i], 0, 0, 0);
          }
          if constexpr (ks + 1 < FKS)
            HFL_LDS_WAIT4_AFTER(wf[(ks + 1) & 1][0], wf[(ks + 1) & 1][1], wf[(ks + 1) & 1][2], wf[(ks + 1) & 1][3], h[1]);
        });
        // epilogue: bias, query scale, fp16 (hi, lo) split (exactly csrc/qkv_fused.hip / gemm_x3's EPI 2), into the region's image:
        // row 16 wave + fr, head i of the pair: 64 B = [16 hi | 16 lo]; 16-B slot t of row r at slot t ^ ((r >> 1) & 7)
        const int trow = wave * 16 + fr;
        unsigned char* img = s_img + reg * FREG + trow * 128;
        const int sw = (trow >> 1) & 7;
        // (the bias through inline asm: hipcc hoists a plain read into the k-loop and waits vmcnt(0) in front of it there)
        f32x4 bias2[2];
        {
          const uint32_t baddr = (uint32_t)(uintptr_t)(bs + reg * FC + pr * 32 + fq * 4);
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(bias2[0]), "=&v"(bias2[1]) : "v"(baddr), "v"(h[1]));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 b = bias2[i];
          float v0 = h[i][0] + b[0], v1 = h[i][1] + b[1], v2 = h[i][2] + b[2], v3 = h[i][3] + b[3];
          if (reg == 0) { v0 *= p.q_scale; v1 *= p.q_scale; v2 *= p.q_scale; v3 *= p.q_scale; }
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(v0, v1), h23 = __builtin_amdgcn_cvt_pkrtz(v2, v3);
          const auto l01 = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1]);
          const auto l23 = __builtin_amdgcn_cvt_pkrtz(v2 - (float)h23[0], v3 - (float)h23[1]);
          const uint2 hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
          const uint2 lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
          // (inline asm: hipcc waits vmcnt(0) in front of every LDS WRITE it sees while an LDS-DMA is in flight, whatever array
          // it goes to -- only reads are disambiguated by array)
          FA_DS_WRITE64((uint32_t)(uintptr_t)(img + (((i * 4 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8), hi);
          FA_DS_WRITE64((uint32_t)(uintptr_t)(img + (((i * 4 + 2 + (fq >> 1)) ^ sw) << 4) + (fq & 1) * 8), lo);
        }
      }
      // the pair's expanded tables (waves 8..11 fetch them: they carry no LDS-DMA, so the loads' waits do not drain the ring)
      if (RPE == 2 && !loader) {
        const float4* src = reinterpret_cast<const float4*>(p.rpe2 + (size_t)(2 * pr) * TS);
        for (int i = tid - 8 * 64; i < 2 * TS / 4; i += (FW - 8) * 64) reinterpret_cast<float4*>(s_tab)[i] = src[i];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // q, k, v images and tables of the pair are complete (the ring keeps moving)
      asm volatile("" ::: "memory");

      // ---- attention of heads 2 pr, 2 pr + 1 on the tile's 4 windows: 24 (window, head, query tile) units, 2 per wave
#pragma unroll 1
      for (int u = wave; u < 4 * 2 * FT; u += FW) {
        const int qt = u % FT, hl = (u / FT) & 1, wl = u / (2 * FT);
        const int hd = 2 * pr + hl;
        // tile row of position j of window wl: group wl / D, member wl % D of the group, stride D
        const int wbase = (wl / D) * (FK * D) + (wl % D);
        auto trow_of = [&](int j) -> int { return wbase + j * D; };
        // 16-B chunk `ch` (0, 1: hi; 2, 3: lo) of head hl of tile row r in region reg
        auto iaddr = [&](int reg, int r, int ch) -> const unsigned char* {
          return s_img + reg * FREG + r * 128 + ((((4 * hl + ch) ^ ((r >> 1) & 7))) << 4);
        };
        // operands (v5's mappings): K tile kt: lane (c, g) = key c, chunk g; Q tile: [q_hi | q_hi] and [q_lo | q_lo]
        uint4 ka[FT], qh, ql;
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) ka[kt] = *reinterpret_cast<const uint4*>(iaddr(1, trow_of(kt * 16 + c), g));
        {
          const int r = trow_of(qt * 16 + c);
          qh = *reinterpret_cast<const uint4*>(iaddr(0, r, g & 1));
          ql = *reinterpret_cast<const uint4*>(iaddr(0, r, 2 + (g & 1)));
        }
        // V^T fragments of the key-tile pairs (0, 1) and (2, -): lane 4 q + p of a 16-lane group addresses key q of its
        // 4-key block, dims 4 p .. 4 p + 3
        fa_h8 vhi[2], vlo[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          typedef __attribute__((address_space(3))) fa_s4 lds_s4;
          const int kk = 4 * g + (c >> 2), cp = c & 3;
          const int r0 = trow_of((2 * pp) * 16 + kk);
          const fa_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r0, cp >> 1) + (cp & 1) * 8));
          const fa_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r0, 2 + (cp >> 1)) + (cp & 1) * 8));
          fa_s4 h1 = {0, 0, 0, 0}, l1 = {0, 0, 0, 0};
          if (2 * pp + 1 < FT) {
            const int r1 = trow_of((2 * pp + 1) * 16 + kk);
            h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r1, cp >> 1) + (cp & 1) * 8));
            l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(iaddr(2, r1, 2 + (cp >> 1)) + (cp & 1) * 8));
          }
          const short __attribute__((ext_vector_type(8))) hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
          const short __attribute__((ext_vector_type(8))) ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
          vhi[pp] = __builtin_bit_cast(fa_h8, hh);
          vlo[pp] = __builtin_bit_cast(fa_h8, ll);
        }
        const int bid0 = s_kbid[trow_of(0)], bidl = s_kbid[trow_of(FK - 1)];
        const bool homog = __builtin_amdgcn_readfirstlane((bidl >= 0 && bid0 == bidl) ? 1 : 0) != 0;
        const int4 qm = s_qry[trow_of(qt * 16 + c)];
        const int q_bid = qm.z;
        const int tabb = (int)(uintptr_t)(s_tab + hl * TS);
        const int qxa = qm.x + tabb, qyza = qm.y + tabb * 0x10001;

        f32x4 sc[FT];
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
          const fa_h8 ak = __builtin_bit_cast(fa_h8, ka[kt]);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(fa_h8, ql), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, __builtin_bit_cast(fa_h8, qh), acc, 0, 0, 0);
          sc[kt] = acc;
        }
        if (RPE == 2) {
#pragma unroll
          for (int kt = 0; kt < FT; ++kt) {
            f32x4 bx, by, bz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int2 km = s_key[trow_of(kt * 16 + 4 * g + r)];
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
              if (s_kbid[trow_of(kt * 16 + 4 * g + r)] != q_bid) sc[kt][r] += mask2;
        }
        float mx = kFDead;
#pragma unroll
        for (int kt = 0; kt < FT; ++kt) {
          mx = fa_max3(mx, sc[kt][0], sc[kt][1]);
          mx = fa_max3(mx, sc[kt][2], sc[kt][3]);
        }
        mx = fa_rows_max(mx);
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
        sum = fa_rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        f32x4 oacc[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          const f32x4 pa = sc[2 * pp];
          unsigned int h0, h1, h2 = 0u, h3 = 0u, l0, l1, l2 = 0u, l3 = 0u;
          fa_split_pair_f16(pa[0], pa[1], h0, l0);
          fa_split_pair_f16(pa[2], pa[3], h1, l1);
          if (2 * pp + 1 < FT) {
            const f32x4 pb = sc[2 * pp + 1];
            fa_split_pair_f16(pb[0], pb[1], h2, l2);
            fa_split_pair_f16(pb[2], pb[3], h3, l3);
          }
          const u32x4 uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
          const fa_h8 phi = __builtin_bit_cast(fa_h8, uh), plo = __builtin_bit_cast(fa_h8, ul);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi[pp], plo, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vlo[pp], phi, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi[pp], phi, o, 0, 0, 0);
          oacc[pp] = o;
        }
        f32x4 o = oacc[0];
        o += oacc[1];
        o *= inv;
        // split2 rows for the proj GEMM through the wave's staging block: one 16-B store per lane, a quad of lanes per row
        {
          uint2 hi, lo;
          x3_split_pair_scalar(o[0], o[1], hi.x, lo.x);
          x3_split_pair_scalar(o[2], o[3], hi.y, lo.y);
          FA_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((g >> 1) ^ st_x) * 16) + (g & 1) * 8), hi);
          FA_DS_WRITE64((uint32_t)(uintptr_t)(stg + st_row + (((2 + (g >> 1)) ^ st_x) * 16) + (g & 1) * 8), lo);
          uint4 v;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((uint32_t)(uintptr_t)(stg + st_quad)) : "memory");
          const int rl = lane >> 2, ch = lane & 3;
          const int orow_l = s_qry[trow_of(qt * 16 + rl)].w;
          if (orow_l >= 0)
            *reinterpret_cast<uint4*>(p.out + (size_t)orow_l * (uint32_t)(4 * FC) +
                                      (uint32_t)((hd >> 1) * 128 + (hd & 1) * 32 + (ch & 1) * 16 + (ch >> 1) * 64)) = v;
        }
      }
    }
  }
}

}  // namespace

// per-launch HIP events for bench.py's roofline leg (the launches sit inside hfl_block_forward_x3: no Python timer sees them)
struct FusedTimingRec {
  hipEvent_t e0, e1;
  double bytes, flops_gemm, flops_attn;
};
static int g_fused_timing = 0;
static std::vector<FusedTimingRec> g_fused_recs;
static std::mutex g_fused_mu;

extern "C" int hfl_internal_rpe_form(int depth, int bnd, int f16);
static int g_attn_fused_split = 1;      // probe knob 'attn_fused_split'
extern "C" void hfl_internal_set_attn_fused_split(int v) { g_attn_fused_split = v ? 1 : 0; }

extern "C" {

/* 1 when hfl_attn_fused_fwd takes this configuration (see include/hotformerloc_hip.h), else 0 */
int hfl_attn_fused_ok(const hfl_window_attn_desc* d, int channels, int has_rpe) {
  if (d == nullptr || channels != FC || d->n_heads != FH || d->patch_size != FK || d->n_relay != 0) return 0;
  if (d->dilation < 1 || FROWS % (FK * d->dilation) != 0) return 0;
  if (d->depth < 1 || d->depth > 7) return 0;
  if (has_rpe && hfl_internal_rpe_form(d->depth, d->pos_bnd, 1) != 2) return 0;
  if (d->n_tokens <= 0 || d->n_tokens >= ((int64_t)1 << 31) / (4 * FC)) return 0;
  return 1;
}

int hfl_attn_fused_fwd(void* out_split2, const float* x, const float* gamma, const float* beta, float eps, const void* qkv_pack,
                       const float* qkv_bias, float q_scale, const uint32_t* tok_meta, const float* rpe_table,
                       const hfl_window_attn_desc* d, hfl_stream_t stream) {
  if (out_split2 == nullptr || x == nullptr || gamma == nullptr || beta == nullptr || qkv_pack == nullptr ||
      qkv_bias == nullptr || tok_meta == nullptr || d == nullptr)
    return HFL_EINVAL;
  if (!hfl_attn_fused_ok(d, FC, rpe_table != nullptr)) return HFL_EINVAL;
  if (rpe_table != nullptr && d->rpe_expanded == nullptr) return HFL_EINVAL;
  FusedAttnParams p;
  p.out = static_cast<unsigned char*>(out_split2); p.x = x; p.gamma = gamma; p.beta = beta;
  p.pack = static_cast<const unsigned char*>(qkv_pack); p.bias = qkv_bias; p.meta = tok_meta;
  p.rpe2 = rpe_table != nullptr ? static_cast<const float*>(d->rpe_expanded) : nullptr;
  p.n_tokens = d->n_tokens; p.n_tiles = (int)hfl_cdiv(d->n_tokens, FROWS); p.D = d->dilation; p.depth = d->depth;
  p.batch = d->batch_size; p.eps = eps; p.q_scale = q_scale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cus = hfl_stream_cus(s);
  // whole rounds of the grid take whole tiles; the tiles left over are cut by head pairs when that lets them share a round
  p.full_tiles = p.n_tiles;
  p.tail_parts = 1;
  if (g_attn_fused_split) {
    const int full = p.n_tiles / cus * cus, rem = p.n_tiles - full;
    int parts = 1;
    while (parts * 2 <= FSPR && rem * parts * 2 <= cus) parts *= 2;
    if (rem > 0 && parts > 1) {
      p.full_tiles = full;
      p.tail_parts = parts;
    }
  }
  const int n_units = p.full_tiles + (p.n_tiles - p.full_tiles) * p.tail_parts;
  const int grid = n_units < cus ? n_units : cus;
  FusedTimingRec rec{};
  const bool timed = g_fused_timing != 0;
  if (timed) {
    // algorithmic bytes: x in + split2 out = 8 B per (row, channel) + 8 B of metadata per token; useful flop: the qkv GEMM
    // 2 M C 3C and the attention core 4 L^2 C per real window
    rec.bytes = (double)d->n_tokens * FC * 8.0 + (double)d->n_tokens * 8.0;
    rec.flops_gemm = 6.0 * (double)d->n_tokens * FC * FC;
    rec.flops_attn = 4.0 * FK * FK * FC * (double)((d->n_tokens + FK - 1) / FK);
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess || hipEventRecord(rec.e0, s) != hipSuccess) {
      if (rec.e0) (void)hipEventDestroy(rec.e0);
      if (rec.e1) (void)hipEventDestroy(rec.e1);
      return HFL_EINVAL;
    }
  }
  if (p.rpe2 != nullptr) attn_fused_kernel<2><<<grid, FW * 64, 0, s>>>(p);
  else attn_fused_kernel<0><<<grid, FW * 64, 0, s>>>(p);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_fused_mu);
    g_fused_recs.push_back(rec);
  }
  HFL_RETURN_LAST_ERROR();
}

// bench.py: per-launch timing of hfl_attn_fused_fwd on / off (both drop what was recorded) ...
int hfl_internal_fused_timing(int on) {
  std::lock_guard<std::mutex> lk(g_fused_mu);
  for (auto& r : g_fused_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_fused_recs.clear();
  g_fused_timing = on ? 1 : 0;
  return HFL_OK;
}
// ... and read it: per launch the duration (ms), algorithmic bytes, useful GEMM and attention flop; returns the launches recorded
int hfl_internal_fused_timing_read(double* ms, double* bytes, double* flops_gemm, double* flops_attn, int cap) {
  std::lock_guard<std::mutex> lk(g_fused_mu);
  int n = 0;
  for (auto& r : g_fused_recs) {
    if (n >= cap) break;
    if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
    ms[n] = t; bytes[n] = r.bytes; flops_gemm[n] = r.flops_gemm; flops_attn[n] = r.flops_attn;
    ++n;
  }
  return (int)g_fused_recs.size();
}

}  // extern "C"
