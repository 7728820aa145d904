// Windowed multi-head attention over z-order octree windows and ragged relay-token
// self-attention for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation).
//
// What the reference does for this step (models/octformer_backbone.py:59-88):
// materialise an int64 (N,K,K,3) relative-position tensor and an int64 (N,K,K) mask per
// depth (models/octree.py:186-222,272-283), gather the RPE table into an (N,H,K,K) fp32
// bias (models/layers/octformer_layers.py:159-170), pad/permute tokens into windows
// (models/octree.py:346-386) and call SDPA.  None of that is materialised here:
//
//   * one workgroup = one window, one wave = one head (head dim 16);
//   * K and V fragments of the wave's head are loaded once, straight into the MFMA
//     operand layout (16-B lane loads for Q/K), and stay in registers;
//   * S^T = K Q^T per 16-query tile, so a lane owns ONE query column and 4T keys:
//     the softmax reduction is over registers plus two 16-lane hops;
//   * the additive bias is computed in registers: -1e3 where the batch ids differ
//     (models/octree.py:66,267-270) + sum over axes of rpe_table[clamp(dx)+bnd+axis*n, h]
//     read from an LDS copy of the table (head-major so a wave stays on one row);
//   * P feeds the second MFMA directly as the A operand (key order inside a k-step is
//     a permutation shared with the V fragment), O is normalised and scattered to the
//     token rows: window (un)packing, dilation and padding are pure index arithmetic.
#include "hfl_common.h"
#include "x3_math.h"

#include <cstring>
#include <mutex>
#include <vector>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kMaskValue = -1e3f;     // models/octree.py:66
constexpr float kDeadValue = -1e30f;    // sequence positions that do not exist

struct WinParams {
  float* out;
  const float* qkv;
  const uint32_t* meta;
  const float* table;
  int64_t n_tokens;
  int64_t rt_row0;
  int n_windows;
  int K;
  int D;
  int H;
  int bnd;
  int batch;
  int clamp;      // 1 when a coordinate difference can exceed pos_bnd
  float scale;
  const float* qkv_bias;   // (3*H*16) added to q,k,v on load (bias-free GEMM upstream), or null
  int out_split;           // 1: out is bf16 [hi|hi|lo] rows of 3*H*16 (A operand of the K-concatenated split GEMM);
                           // 2: bf16 split2 rows of 2*H*16 (operand of hfl_linear_x3)
  int qkv_f16;             // 1: qkv rows are the fp16 (hi, lo) attention operand layout of hfl_linear_x3_qkv (v5 kernel)
  const float* rpe2;       // (H, TS) expanded table of hfl_window_rpe_expand (v4), or null
  int depth;               // octree depth of the tokens (0 = unknown)
  int dbg;                 // ablation bits (tools/kbench.py): 1 no softmax/MFMA, 2 no stores, 4 cached rows
};

// compiler-visible max of three (v5): the inline-asm form below is invisible to the hazard recogniser, and right behind a
// 16-cycle fp16 MFMA it read its operand before the matrix pipe had written it (wrong row maxima in the last query tile)
__device__ __forceinline__ float att_max3_c(float a, float b, float c) {
  return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

// max / sum over the four 16-lane rows of a wave (lanes c, c+16, c+32, c+48) on the VALU: `v_permlane32_swap` exchanges the upper
// half of one register with the lower half of another, `v_permlane16_swap` the odd rows of one with the even rows of the other;
// fed with two copies of v, a max (add) of the swapped pair is the xor-32 (xor-16) butterfly step.  Replaces two
// `ds_bpermute_b32` round trips through the LDS crossbar per reduction.  (Inline asm: the builtin's second result is folded
// into the first by this compiler.  The operands are VALU results, never raw MFMA outputs: see att_max3_c; `s_nop 1` is the
// VALU-write -> permlane-read wait.)
__device__ __forceinline__ float att_rows_max(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float att_rows_sum(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1\n\tv_mov_b32 %1, %0\n\t"
      "s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}

// two floats in [0, 1] -> packed fp16 (hi, lo): hi = RTZ(p), lo = RTZ(p - hi).  The residual is ONE `v_fma_mix_f32` per value
// (fma(hi as f16, -1, p): the f16 operand is widened by the instruction) instead of v_cvt_f32_f16 + v_sub_f32.  The asm's
// operands are the results of ordinary VALU instructions (the cvt_pkrtz in front of it), so the trans-use wait state of the
// v_exp_f32 that produced p has already been served.
__device__ __forceinline__ void att_split_pair_f16(float p0, float p1, unsigned int& hi, unsigned int& lo) {
  hi = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(p0, p1));
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(p0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(p1));
  lo = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// max of three for the softmax row maxima.  (An earlier inline-asm v_max3_f32 saved the canonicalising v_max(x, x) of
// fmaxf() but hid the operands from the compiler's hazard recogniser: see att_max3_c.)
__device__ __forceinline__ float att_max3(float a, float b, float c) { return att_max3_c(a, b, c); }

__device__ __forceinline__ uint16_t att_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// one channel of a split-precision output row: mode 1 = [hi | hi | lo] planes of C (row of 3C bf16), mode 2 = split2,
// per 32-channel block [32 x hi | 32 x lo] (row of 2C bf16, csrc/gemm_x3.hip)
__device__ __forceinline__ void att_store_split(uint16_t* out16, int64_t orow, int C, int ch, float v, int mode) {
  const uint16_t hi = att_bf16_rne(v);
  const uint16_t lo = att_bf16_rne(v - __uint_as_float((uint32_t)hi << 16));
  if (mode == 2) {
    uint16_t* o = out16 + orow * (2 * (int64_t)C) + (ch >> 5) * 64 + (ch & 31);
    o[0] = hi;
    o[32] = lo;
  } else {
    uint16_t* o = out16 + orow * (3 * (int64_t)C) + ch;
    o[0] = hi;
    o[C] = hi;
    o[2 * C] = lo;
  }
}
// FOUR consecutive channels ch .. ch+3 (ch % 4 == 0) of output row `orow`: one 16-B store (fp32) or 8-B stores of the
// packed hi / lo halves (split modes)
__device__ __forceinline__ void att_store_row4(char* out_b, uint32_t orow, int C, int ch, const f32x4 o, int mode) {
  if (mode == 0) {
    *reinterpret_cast<f32x4*>(out_b + ((size_t)orow * (uint32_t)C + (uint32_t)ch) * 4u) = o;
    return;
  }
  uint2 hi, lo;                       // v_cvt_pk_bf16_f32 (round to nearest even, as att_bf16_rne): 6 instructions per pair
  x3_split_pair_scalar(o[0], o[1], hi.x, lo.x);
  x3_split_pair_scalar(o[2], o[3], hi.y, lo.y);
  uint16_t* out16 = reinterpret_cast<uint16_t*>(out_b);
  if (mode == 2) {
    uint16_t* d = out16 + (size_t)orow * (uint32_t)(2 * C) + (uint32_t)((ch >> 5) * 64 + (ch & 31));
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + 32) = lo;
  } else {
    uint16_t* d = out16 + (size_t)orow * (uint32_t)(3 * C) + (uint32_t)ch;
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + C) = hi;
    *reinterpret_cast<uint2*>(d + 2 * C) = lo;
  }
}

// ----------------------------------------------------------------------------------
// v2: same decomposition, instruction diet.  The v1 body spends ~44 VALU per score (unpack,
// six min/max, address math, dead/relay branches) against 2 MFMA per 16 scores, so it is
// VALU-bound 20:1.  Here
//   * per-position metadata is pre-digested once per window into LDS as int4
//     {4*(bnd-x), 4*(bnd-y), 4*(bnd-z), batch id} for keys and {4x,4y,4z,id} for queries:
//     one ds_read_b128 per key, RPE byte offset = med3(q4 + k4, 0, 8*bnd) per axis;
//   * the RPE table copy in LDS is pre-multiplied by log2(e) and scores live in the exp2
//     domain (v_exp_f32 directly, no per-score multiply);
//   * dead positions are just "another batch" (-1e3 mask, exp2 underflows to exactly 0, the
//     same mechanism the reference relies on for padding), so no per-score dead test;
//   * the relay token sits alone in the last 16-tile: that key tile and that query tile are
//     special-cased statically (no RPE, one live key column), never tested per score;
//   * all Q fragments are fetched together with K and V (one memory round trip per window).
template <int T, int G, bool CLAMP, bool RPE>
__global__ void __launch_bounds__(256)
window_attn_kernel_v2(const WinParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LP = T * 16;
  constexpr int TW = T - G;                        // tiles that hold window tokens
  constexpr float kLog2e = 1.4426950408889634f;
  const int H = p.H, K = p.K;
  const int C = H * 16;
  const int nrpe = 2 * p.bnd + 1;
  int4* s_key = reinterpret_cast<int4*>(smem);                         // [LP]
  int4* s_qry = s_key + LP;                                            // [LP]
  int* s_row = reinterpret_cast<int*>(s_qry + LP);                     // [LP] qkv/out row, -1 dead
  float* s_tab = reinterpret_cast<float*>(s_row + LP);                 // [H][3*nrpe] * log2e

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hw = tid >> 6;                       // wave inside the workgroup
  const int nhw = blockDim.x >> 6;               // heads handled by this workgroup
  const int h = blockIdx.y * nhw + hw;           // head of this wave
  const int c = lane & 15, g = lane >> 4;
  constexpr bool rpe = RPE;

  if (rpe)
    for (int i = tid; i < 3 * nrpe * nhw; i += blockDim.x) {
      const int r = i / nhw, hh = i % nhw;       // table is (3*nrpe, H) row-major
      s_tab[hh * 3 * nrpe + r] = p.table[r * H + blockIdx.y * nhw + hh] * kLog2e;
    }
  // LDS byte addresses of this head's three axis tables; the query-side offsets carry them,
  // so `q' + k` is already the address of the table entry (clamped to the axis table)
  const int hu = __builtin_amdgcn_readfirstlane(hw);
  const int tabx = (int)(size_t)(s_tab + hu * 3 * nrpe) ;
  const int taby = tabx + nrpe * 4;
  const int tabz = taby + nrpe * 4;
  const int hi4 = 8 * p.bnd;                                           // 4 * (2*bnd)
  const float scale2 = p.scale * kLog2e;
  const float mask2 = kMaskValue * kLog2e;

  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), bk = bq;
  float bv = 0.f;
  if (p.qkv_bias != nullptr) {
    bq = *reinterpret_cast<const float4*>(p.qkv_bias + h * 16 + 4 * g);
    bk = *reinterpret_cast<const float4*>(p.qkv_bias + C + h * 16 + 4 * g);
    bv = p.qkv_bias[2 * C + h * 16 + c];
  }

  for (int w = blockIdx.x; w < p.n_windows; w += gridDim.x) {
    __syncthreads();
    for (int j = tid; j < LP; j += blockDim.x) {
      int bid = -1, row = -1;
      int x = 0, y = 0, z = 0;
      if (j < K) {
        const int64_t t = (p.D == 1) ? (int64_t)w * K + j
                                     : ((int64_t)(w / p.D) * K + j) * p.D + (w % p.D);
        if (t < p.n_tokens) {
          const uint32_t xyz = p.meta[2 * t];
          x = (int)(xyz & 1023u); y = (int)((xyz >> 10) & 1023u); z = (int)(xyz >> 20);
          bid = (int)p.meta[2 * t + 1];
          row = (int)t;
        }
      } else if (G > 0 && j == K) {
        const int64_t t0 = (int64_t)w * K;
        bid = t0 < p.n_tokens ? (int)p.meta[2 * t0 + 1] : p.batch;
        row = (int)(p.rt_row0 + w);
      }
      s_key[j] = make_int4(4 * (p.bnd - x), 4 * (p.bnd - y), 4 * (p.bnd - z), bid);
      s_qry[j] = make_int4(4 * x, 4 * y, 4 * z, row >= 0 ? bid : -2);
      s_row[j] = row;
    }
    __syncthreads();

    // ---- all fragments of this head in one round trip ---------------------------------
    float4 kf[T], qf[T];
    float vf[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = s_row[t * 16 + c];
      kf[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      qf[t] = kf[t];
      if (row >= 0) {
        const float* base = p.qkv + (int64_t)row * 3 * C + h * 16 + 4 * g;
        qf[t] = *reinterpret_cast<const float4*>(base);
        kf[t] = *reinterpret_cast<const float4*>(base + C);
        qf[t].x = (qf[t].x + bq.x) * scale2; qf[t].y = (qf[t].y + bq.y) * scale2;
        qf[t].z = (qf[t].z + bq.z) * scale2; qf[t].w = (qf[t].w + bq.w) * scale2;
        kf[t].x += bk.x; kf[t].y += bk.y; kf[t].z += bk.z; kf[t].w += bk.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (G > 0 && t == T - 1 && r > 0) { vf[t][r] = 0.f; continue; }   // only the relay key lives there
        const int rv = s_row[t * 16 + 4 * g + r];
        vf[t][r] = rv >= 0 ? p.qkv[(int64_t)rv * 3 * C + 2 * C + h * 16 + c] + bv : 0.f;
      }
    }

#pragma unroll
    for (int qt = 0; qt < TW; ++qt) {          // token queries; the relay query is handled below
      int4 q = s_qry[qt * 16 + c];
      const bool q_rpe = rpe;
      q.x += tabx; q.y += taby; q.z += tabz;

      f32x4 s[T];
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].x, qf[qt].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].y, qf[qt].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].z, qf[qt].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].w, qf[qt].w, acc, 0, 0, 0);
        s[kt] = acc;
      }

      float mx = kDeadValue;
#pragma unroll
      for (int kt = 0; kt < TW; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int4 k = s_key[kt * 16 + 4 * g + r];
          float v = s[kt][r];
          if (q_rpe) {
            int ox = q.x + k.x, oy = q.y + k.y, oz = q.z + k.z;
            if (CLAMP) {     // |delta| can exceed pos_bnd only when 2^depth - 1 > pos_bnd
              ox = min(max(ox, tabx), tabx + hi4);
              oy = min(max(oy, taby), taby + hi4);
              oz = min(max(oz, tabz), tabz + hi4);
            }
            typedef __attribute__((address_space(3))) const float lds_f32;
            v += (*reinterpret_cast<lds_f32*>(ox) + *reinterpret_cast<lds_f32*>(oy)) +
                 *reinterpret_cast<lds_f32*>(oz);
          }
          if (k.w != q.w) v += mask2;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      if (G > 0) {   // relay key: position K = tile T-1, k-slot group 0, register 0
        const int kb = s_key[K].w;
        float v = s[T - 1][0];
        if (g != 0 || kb != q.w) v += mask2;
        s[T - 1][0] = v;
        mx = fmaxf(mx, v);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < TW; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
          s[kt][r] = e;
          sum += e;
        }
      }
      if (G > 0) {
        const float e = __builtin_amdgcn_exp2f(s[T - 1][0] - mx);
        s[T - 1][0] = e;
        sum += e;
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;

      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < TW; ++kt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][r], s[kt][r] * inv, o, 0, 0, 0);   // O^T: see v4
      }
      if (G > 0) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[T - 1][0], s[T - 1][0] * inv, o, 0, 0, 0);
      const int orow = s_row[qt * 16 + c];          // the lane's accumulator: channels 4g .. 4g+3 of query c
      if (orow >= 0) att_store_row4(reinterpret_cast<char*>(p.out), (uint32_t)orow, C, h * 16 + 4 * g, o, p.out_split);
    }

    if (G > 0) {
      // ---- the relay token as a QUERY: one row, done on the VALU instead of padding a 16-row
      // MFMA tile with 15 dead queries.  No RPE for the relay row (octformer_backbone.py:78-80).
      const int rbid = s_qry[K].w;
      const float4 qr = qf[T - 1];                  // lanes c == 0 hold row K = the relay query ...
      float4 qrt;                                   // ... broadcast its 4g..4g+3 slice to the 16 lanes
      qrt.x = __shfl(qr.x, lane & 48, 64); qrt.y = __shfl(qr.y, lane & 48, 64);
      qrt.z = __shfl(qr.z, lane & 48, 64); qrt.w = __shfl(qr.w, lane & 48, 64);
      float sr[TW];
      float m = kDeadValue;
#pragma unroll
      for (int kt = 0; kt < TW; ++kt) {
        float part = (qrt.x * kf[kt].x + qrt.y * kf[kt].y) + (qrt.z * kf[kt].z + qrt.w * kf[kt].w);
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);           // score of key kt*16+c (replicated over g)
        if (s_key[kt * 16 + c].w != rbid) part += mask2;
        sr[kt] = part;
        m = fmaxf(m, part);
      }
      const float4 kr = kf[T - 1];                  // row K of K, held by lanes c == 0
      float srr = (qrt.x * __shfl(kr.x, lane & 48, 64) + qrt.y * __shfl(kr.y, lane & 48, 64)) +
                  (qrt.z * __shfl(kr.z, lane & 48, 64) + qrt.w * __shfl(kr.w, lane & 48, 64));
      srr += __shfl_xor(srr, 16, 64);
      srr += __shfl_xor(srr, 32, 64);
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
      m = fmaxf(m, srr);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < TW; ++kt) {
        sr[kt] = __builtin_amdgcn_exp2f(sr[kt] - m);
        sum += sr[kt];
      }
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d, 64);
      const float err = __builtin_amdgcn_exp2f(srr - m);
      const float inv = 1.0f / (sum + err);
      // O_rt[d = c] = sum_keys p V: this lane holds V[kt*16 + 4g + r][d = c]; the weight of that key
      // sits in lane (c' = 4g + r) of the same 16-lane group
      float acc = 0.f;
#pragma unroll
      for (int kt = 0; kt < TW; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc += __shfl(sr[kt], (lane & 48) | (4 * g + r), 64) * vf[kt][r];
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      const float vrt = __shfl(vf[T - 1][0], c, 64);     // V[relay][d = c] lives in the g == 0 lanes
      acc = (acc + err * vrt) * inv;
      const int orow = s_row[K];
      if (g == 0 && orow >= 0) {
        if (p.out_split) {
          att_store_split(reinterpret_cast<uint16_t*>(p.out), orow, C, h * 16 + c, acc, p.out_split);
        } else {
          p.out[(int64_t)orow * C + h * 16 + c] = acc;
        }
      }
    }
  }
}


// ----------------------------------------------------------------------------------
// v4: v2 after a second instruction diet.  The ISA of v2<5,1> is 3250 instructions per (window,
// head) for 148 MFMAs -- ~50 issue slots per score against a floor of ~8 -- so the kernel is
// VALU/issue bound, not memory bound (SQ counters: profiles/r01_b_summary.md).  What changes:
//   * RPE bias = X[dx] + YZ[dy,dz]: the y and z axis tables are pre-added into a (2R+1)^2 table
//     (R = 2^depth - 1, no clamp needed when R <= pos_bnd), built once per table by
//     hfl_window_rpe_expand and copied to LDS per workgroup: two lookups and one add per score
//     instead of three lookups, two adds and six clamps;
//   * a lane's 16 keys are the same for every query tile: their LDS table offsets live in
//     registers for the whole window instead of being re-read per query tile;
//   * windows that lie inside one cloud and have no padding (all but ~B of them) take a path
//     with no mask arithmetic at all; tokens are sorted by cloud, so comparing the batch id of the
//     first and last token of the window decides it (wave-uniform, two scalar loads);
//   * P stays un-normalised into the P V MFMAs; the 1/sum lands on the 4 output registers;
//   * v_rcp_f32 instead of an IEEE divide, 32-bit byte offsets from the qkv / out base pointers
//     (saddr + voffset addressing; the caller falls back to v2 when a buffer exceeds 4 GiB).
// Occupancy target: 3 waves per SIMD (168 VGPRs) when the window has at most 3 token tiles (K = 48),
// 2 otherwise (K = 64 needs ~190 VGPRs; forcing 3 spills).  Measured with back-to-back launches
// (tools/kbench.py): 3 waves 109 us vs 2 waves + next-window register prefetch 161 us at depth 4 --
// the prefetch variant lost to plain occupancy and was dropped.
constexpr int v4_waves_per_simd(int T, int G) { return (T - G) <= 3 ? 3 : 2; }

template <int T, int G, bool RPE>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(v4_waves_per_simd(T, G), v4_waves_per_simd(T, G))))
window_attn_kernel_v4(const WinParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LP = T * 16;
  constexpr int TW = T - G;
  constexpr float kLog2e = 1.4426950408889634f;
  typedef __attribute__((address_space(3))) const float lds_f32;
  const int H = p.H, K = p.K;
  const int C = H * 16;
  const int R = (1 << p.depth) - 1, W = 2 * R + 1;
  const int TS = RPE ? ((W + W * W + 3) & ~3) : 0;
  // metadata is double-buffered: window i+1's is written while window i's is still being read
  int4* s_qry0 = reinterpret_cast<int4*>(smem);                        // [2][LP] {4x, 4(yW+z), id, row}
  int2* s_key0 = reinterpret_cast<int2*>(s_qry0 + 2 * LP);             // [2][LP] {4(R-x), 4(W+(R-y)W+R-z)}
  int* s_kbid0 = reinterpret_cast<int*>(s_key0 + 2 * LP);              // [2][LP] batch id, -1 dead
  float* s_tab = reinterpret_cast<float*>(s_kbid0 + 2 * LP);           // [nhw][TS] * log2e

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hw = tid >> 6;
  const int nhw = blockDim.x >> 6;
  const int h = blockIdx.y * nhw + hw;
  const int c = lane & 15, g = lane >> 4;

  if (RPE) {
    const float4* src = reinterpret_cast<const float4*>(p.rpe2 + (size_t)blockIdx.y * nhw * TS);
    float4* dst = reinterpret_cast<float4*>(s_tab);
    for (int i = tid; i < nhw * TS / 4; i += blockDim.x) dst[i] = src[i];
  }
  const int hu = __builtin_amdgcn_readfirstlane(hw);
  const int tabb = (int)(size_t)(s_tab + hu * TS);
  const float scale2 = p.scale * kLog2e;
  const float mask2 = kMaskValue * kLog2e;
  const float rt_add = (g == 0) ? 0.f : kDeadValue;   // the relay key lives in the g == 0 lanes only
  const uint32_t row_q = (uint32_t)(3 * C) * 4u;       // bytes per qkv row
  const uint32_t row_o = p.out_split == 1 ? (uint32_t)(3 * C) * 2u : (uint32_t)C * 4u;   // split2 rows: 2C bf16 = 4C B
  const char* qkv_b = reinterpret_cast<const char*>(p.qkv);
  char* out_b = reinterpret_cast<char*>(p.out);

  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f), bk = bq;
  float bv = 0.f;
  if (p.qkv_bias != nullptr) {
    bq = *reinterpret_cast<const float4*>(p.qkv_bias + h * 16 + 4 * g);
    bk = *reinterpret_cast<const float4*>(p.qkv_bias + C + h * 16 + 4 * g);
    bv = p.qkv_bias[2 * C + h * 16 + c];
  }
  const uint32_t col_qk = (uint32_t)(h * 16 + 4 * g) * 4u;
  const uint32_t col_v = (uint32_t)(2 * C + h * 16 + c) * 4u;

  const int n_tok = (int)p.n_tokens;
  const bool owns = tid < LP;                          // the launcher guarantees blockDim.x >= LP

  // Request one window: its metadata word and every Q/K/V fragment of this wave's head.  Rows are
  // index arithmetic (token of window slot j = tok0 + j * D), so nothing here waits on anything.
  auto request = [&](float4 (&kf)[T], float4 (&qf)[T], float (&vf)[T][4], uint2& mt, int w) {
    const int tstep = p.D;
    const int tok0 = (p.D == 1) ? w * K : (w / p.D) * K * p.D + (w % p.D);
    const int rt_row = (int)p.rt_row0 + w;
    mt = make_uint2(0u, 0xFFFFFFFFu);                  // dead slot
    if (owns && tid < K) {
      const int t = tok0 + tid * tstep;
      if (t < n_tok) mt = *reinterpret_cast<const uint2*>(p.meta + 2 * (int64_t)t);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      kf[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      qf[t] = kf[t];
      int row;
      bool ok;
      if (G > 0 && t == T - 1) {
        row = rt_row;
        ok = (c == 0);
      } else {
        row = tok0 + (t * 16 + c) * tstep;
        ok = row < n_tok;
      }
      if ((p.dbg & 4) && ok) row &= 63;
      if (ok) {
        const char* base = qkv_b + ((uint32_t)row * row_q + col_qk);
        qf[t] = *reinterpret_cast<const float4*>(base);
        kf[t] = *reinterpret_cast<const float4*>(base + (uint32_t)C * 4u);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vf[t][r] = 0.f;
        int rv;
        bool okv;
        if (G > 0 && t == T - 1) {
          if (r > 0) continue;                         // only the relay key lives in that tile
          rv = rt_row;
          okv = (g == 0);
        } else {
          rv = tok0 + (t * 16 + 4 * g + r) * tstep;
          okv = rv < n_tok;
        }
        if ((p.dbg & 4) && okv) rv &= 63;
        if (okv) vf[t][r] = *reinterpret_cast<const float*>(qkv_b + ((uint32_t)rv * row_q + col_v));
      }
    }
  };

  float4 kf[T], qf[T];
  float vf[T][4];
  uint2 mt;
  int it = 0;
  for (int w = blockIdx.x; w < p.n_windows; w += gridDim.x, ++it) {
    request(kf, qf, vf, mt, w);

    const int tstep = p.D;
    const int tok0 = (p.D == 1) ? w * K : (w / p.D) * K * p.D + (w % p.D);
    const int rt_row = (int)p.rt_row0 + w;
    int4* s_qry = s_qry0 + (it & 1) * LP;
    int2* s_key = s_key0 + (it & 1) * LP;
    int* s_kbid = s_kbid0 + (it & 1) * LP;
    if (owns) {
      const int j = tid;
      int bid = -1, row = -1;
      int x = 0, y = 0, z = 0;
      if (j < K) {
        if (mt.y != 0xFFFFFFFFu) {
          x = (int)(mt.x & 1023u); y = (int)((mt.x >> 10) & 1023u); z = (int)(mt.x >> 20);
          bid = (int)mt.y;
          row = tok0 + j * tstep;
        }
      } else if (G > 0 && j == K) {
        row = rt_row;
      }
      s_key[j] = make_int2(4 * (R - x), 4 * (W + (R - y) * W + (R - z)));
      s_kbid[j] = bid;
      s_qry[j] = make_int4(4 * x, 4 * (y * W + z), bid, row);
    }
    __syncthreads();
    // the relay token carries the batch id of the window's first token (pad window: B)
    const int bid0 = s_kbid[0], bidl = s_kbid[K - 1];
    const int rt_bid = bid0 >= 0 ? bid0 : p.batch;
    const bool homog = __builtin_amdgcn_readfirstlane((bidl >= 0 && bid0 == bidl) ? 1 : 0) != 0;

    // q/k bias + softmax scale (after the loads so that nothing waits between their issue)
#pragma unroll
    for (int t = 0; t < T; ++t) {
      qf[t].x = (qf[t].x + bq.x) * scale2; qf[t].y = (qf[t].y + bq.y) * scale2;
      qf[t].z = (qf[t].z + bq.z) * scale2; qf[t].w = (qf[t].w + bq.w) * scale2;
      kf[t].x += bk.x; kf[t].y += bk.y; kf[t].z += bk.z; kf[t].w += bk.w;
#pragma unroll
      for (int r = 0; r < 4; ++r) vf[t][r] += bv;
    }

    int kxa[TW][4], kyza[TW][4];
    if (RPE) {
#pragma unroll
      for (int kt = 0; kt < TW; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int2 km = s_key[kt * 16 + 4 * g + r];
          kxa[kt][r] = km.x;
          kyza[kt][r] = km.y;
        }
    }

    auto body = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
      for (int qt = 0; qt < TW; ++qt) {          // token queries; the relay query is handled below
        const int4 qm = s_qry[qt * 16 + c];
        const int qxa = qm.x + tabb, qyza = qm.y + tabb;

        f32x4 s[T];
#pragma unroll
        for (int kt = 0; kt < T; ++kt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].x, qf[qt].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].y, qf[qt].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].z, qf[qt].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].w, qf[qt].w, acc, 0, 0, 0);
          s[kt] = acc;
        }
        if (RPE) {
#pragma unroll
          for (int kt = 0; kt < TW; ++kt) {
            f32x4 bx, byz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bx[r] = *reinterpret_cast<lds_f32*>(kxa[kt][r] + qxa);
              byz[r] = *reinterpret_cast<lds_f32*>(kyza[kt][r] + qyza);
            }
            s[kt] += bx + byz;
          }
        }
        if (MASKED) {
#pragma unroll
          for (int kt = 0; kt < TW; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (s_kbid[kt * 16 + 4 * g + r] != qm.z) s[kt][r] += mask2;
        }
        float mx = kDeadValue;
        float srt = kDeadValue;
        if (G > 0) {   // relay key: position K = tile T-1, k-slot group 0, register 0; no RPE
          srt = s[T - 1][0] + rt_add;
          if (MASKED && rt_bid != qm.z) srt += mask2;
          mx = srt;
        }
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          mx = att_max3(mx, s[kt][0], s[kt][1]);
          mx = att_max3(mx, s[kt][2], s[kt][3]);
        }
        mx = att_max3(mx, __shfl_xor(mx, 16, 64), mx);
        mx = att_max3(mx, __shfl_xor(mx, 32, 64), mx);
        const f32x4 nmx4 = {-mx, -mx, -mx, -mx};
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          f32x4 e = s[kt] + nmx4;
          e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
          e[2] = __builtin_amdgcn_exp2f(e[2]); e[3] = __builtin_amdgcn_exp2f(e[3]);
          s[kt] = e;
          sum4 += e;
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        float ert = 0.f;
        if (G > 0) {
          ert = __builtin_amdgcn_exp2f(srt - mx);
          sum += ert;
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = __builtin_amdgcn_rcpf(sum);

        // O^T = V^T P^T: with V as the A operand the accumulator holds 4 CONSECUTIVE CHANNELS (4g .. 4g+3) of query c,
        // whose 1/sum sits in this very lane: one wide store per lane, no shuffles
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kt][r], s[kt][r], o, 0, 0, 0);
        }
        if (G > 0) o = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[T - 1][0], ert, o, 0, 0, 0);
        o *= inv;
        const int orow = qm.w;
        if ((!MASKED || orow >= 0) && (!(p.dbg & 2) || o[0] == 1234.5f))
          att_store_row4(out_b, (uint32_t)orow, C, h * 16 + 4 * g, o, p.out_split);
      }

      if (G > 0) {
        // ---- the relay token as a QUERY: one row, done on the VALU instead of padding a 16-row
        // MFMA tile with 15 dead queries.  No RPE for the relay row (octformer_backbone.py:78-80).
        const int rbid = rt_bid;
        const float4 qr = qf[T - 1];                  // lanes c == 0 hold row K = the relay query ...
        float4 qrt;                                   // ... broadcast its 4g..4g+3 slice to the 16 lanes
        qrt.x = __shfl(qr.x, lane & 48, 64); qrt.y = __shfl(qr.y, lane & 48, 64);
        qrt.z = __shfl(qr.z, lane & 48, 64); qrt.w = __shfl(qr.w, lane & 48, 64);
        float sr[TW];
        float m = kDeadValue;
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          float part = (qrt.x * kf[kt].x + qrt.y * kf[kt].y) + (qrt.z * kf[kt].z + qrt.w * kf[kt].w);
          part += __shfl_xor(part, 16, 64);
          part += __shfl_xor(part, 32, 64);           // score of key kt*16+c (replicated over g)
          if (MASKED && s_kbid[kt * 16 + c] != rbid) part += mask2;
          sr[kt] = part;
          m = fmaxf(m, part);
        }
        const float4 kr = kf[T - 1];                  // row K of K, held by lanes c == 0
        float srr = (qrt.x * __shfl(kr.x, lane & 48, 64) + qrt.y * __shfl(kr.y, lane & 48, 64)) +
                    (qrt.z * __shfl(kr.z, lane & 48, 64) + qrt.w * __shfl(kr.w, lane & 48, 64));
        srr += __shfl_xor(srr, 16, 64);
        srr += __shfl_xor(srr, 32, 64);
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        m = fmaxf(m, srr);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          sr[kt] = __builtin_amdgcn_exp2f(sr[kt] - m);
          sum += sr[kt];
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) sum += __shfl_xor(sum, d, 64);
        const float err = __builtin_amdgcn_exp2f(srr - m);
        const float inv = __builtin_amdgcn_rcpf(sum + err);
        float acc = 0.f;
#pragma unroll
        for (int kt = 0; kt < TW; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc += __shfl(sr[kt], (lane & 48) | (4 * g + r), 64) * vf[kt][r];
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        const float vrt = __shfl(vf[T - 1][0], c, 64);     // V[relay][d = c] lives in the g == 0 lanes
        acc = (acc + err * vrt) * inv;
        const int orow = s_qry[K].w;
        if (g == 0 && orow >= 0) {
          char* ob = out_b + ((uint32_t)orow * row_o);
          if (p.out_split) {
            att_store_split(reinterpret_cast<uint16_t*>(out_b), orow, C, h * 16 + c, acc, p.out_split);
          } else {
            reinterpret_cast<float*>(ob)[h * 16 + c] = acc;
          }
        }
      }
    };
    if (p.dbg & 1) {            // ablation: memory pattern only
#pragma unroll
      for (int qt = 0; qt < TW; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = s_qry[qt * 16 + 4 * g + r].w;
          if (orow >= 0 && !(p.dbg & 2)) {
            const float v = qf[qt].x + kf[qt].y + vf[qt][r];
            char* ob = out_b + ((uint32_t)orow * row_o);
            if (p.out_split) {
              att_store_split(reinterpret_cast<uint16_t*>(out_b), orow, C, h * 16 + c, v, p.out_split);
            } else {
              reinterpret_cast<float*>(ob)[h * 16 + c] = v;
            }
          }
        }
      if ((p.dbg & 2) && qf[0].x + kf[T - 1].y + vf[0][1] == 1234.5f) p.out[0] = 1.f;
    } else if (homog)
      body(std::false_type{});
    else
      body(std::true_type{});
  }
}

// v5: v4 with the two contractions on the fp16 matrix cores at fp32-equivalent accuracy.
// v4 is bound by issue slots: 87 fp32 MFMAs (32 cycles each, dependent chains) + ~570 VALU per (window, head).  The
// 16x16x32 fp16 MFMA does a K = 32 step in 16 cycles, and a head's 16 dims as [16 x hi | 16 x lo] halves ARE a K = 32
// operand -- but splitting fp32 q, k, v into (hi, lo) inside this kernel would cost more VALU than the MFMAs save.
// So the producer does it: the qkv projection (csrc/gemm_x3.hip, EPI 2) writes every row as [Q | K | V] regions, per head
// 64 B = [16 x hi | 16 x lo] fp16 (hi = RTZ(v), lo = RTZ(v - hi): 22 significant bits), bias added, queries pre-multiplied
// by scale * log2 e.  Same bytes per row as fp32 qkv; every fragment below is ONE 16-B load:
//   S^T = K Q^T   A = [k_hi | k_lo] (lane group g takes chunk g of the head's 64 B), B = [q_hi | q_hi], then [q_lo | q_lo]:
//                 two MFMAs give all four cross terms (k_hi + k_lo)(q_hi + q_lo); fp32 accumulate
//   O^T = V^T P^T A = V columns by `ds_read_b64_tr_b16` from the rows staged in LDS as loaded (hardware transpose),
//                 B = the un-normalised P of two key tiles, split (hi, lo) in registers; 3 MFMAs per pair of key tiles
//                 (v_hi p_hi + v_lo p_hi + v_hi p_lo)
// 42 fp16 MFMAs (16 cycles) instead of 87 fp32 ones, 16 vector loads per window instead of 24, and the relay token
// as a query is simply a fourth query tile with one live column.  Everything else (index-arithmetic windows, double-
// buffered metadata, expanded RPE table, mask-free homogeneous windows, persistent grid) is v4's.
typedef _Float16 att_h8 __attribute__((ext_vector_type(8)));
typedef short att_s4 __attribute__((ext_vector_type(4)));

// RPE: 0 none, 1 expanded x + y-z tables (two lookups), 2 three clamped 1-D tables (deep octrees, three lookups)
// probe (window_debug bit 3): s_memtime stamps of ONE wave (workgroup 0, wave 0) at the phase boundaries of its first 16 windows;
// read back by hfl_internal_read_att_trace (tools/attn_trace.py)
// (compiled in only with -DHFL_ATT_TRACE=1: `HFL_EXTRA_HIPCC_FLAGS=-DHFL_ATT_TRACE=1 python -m hotformerloc_amd.build`; the
// branches cost the depth-5 launch 15 % even when the bit is off)
#ifndef HFL_ATT_TRACE
#define HFL_ATT_TRACE 0
#endif
__device__ unsigned long long g_att_trace[16 * 8];
__device__ unsigned long long g_att_wg[4096 * 2];      // probe: s_memrealtime at the start and end of every workgroup
#define HFL_ATT_STAMP(slot)                                                                      \
  if (HFL_ATT_TRACE && trace_on) {                                                               \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                  \
    if (it < 16 && lane == 0) g_att_trace[it * 8 + (slot)] = t_;                                 \
  }

// (A variant that requested the NEXT window's fragments into a second register set before the softmax of the current one --
// 2 waves per SIMD, <= 256 VGPRs, outputs held back so that no store sits in front of the next `s_waitcnt vmcnt(0)` -- was
// neutral: 65.7 / 81.5 us against 68 / 80.5 us on the depth-5 / depth-4 launches; removed.)
//
// ONE launch can serve several attention problems of the same shape (the pyramid levels of an H-OSA iteration: same K, G,
// heads and table form, different depths and row counts): blockIdx.x ranges are dealt to the problems in proportion to their
// window counts and each workgroup stays inside its problem.  The depth-2 / depth-3 launches of the bench (2 k / 14 k rows:
// 14 us and 21 us alone, one or two windows per workgroup, latency-bound) then ride along with the depth-4 one.
constexpr int kWinMulti = 4;
struct WinMultiParams {
  WinParams p[kWinMulti];
  int first[kWinMulti + 1];      // blockIdx.x range of problem i: [first[i], first[i + 1])
  int n;
};

template <int T, int G, int RPE>
__global__ void __launch_bounds__(256)
    __attribute__((amdgpu_waves_per_eu(v4_waves_per_simd(T, G), v4_waves_per_simd(T, G))))
window_attn_kernel_v5(const WinMultiParams m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LP = T * 16;
  constexpr int TW = T - G;
  constexpr int NP = (T + 1) / 2;                      // pairs of key tiles (K = 32 per PV MFMA)
  int prob = 0;
  while (prob + 1 < m.n && (int)blockIdx.x >= m.first[prob + 1]) ++prob;
  const WinParams& p = m.p[prob];
  const int wg_x = (int)blockIdx.x - m.first[prob], wg_nx = m.first[prob + 1] - m.first[prob];
  typedef __attribute__((address_space(3))) const float lds_f32;
  const int H = p.H, K = p.K;
  const int C = H * 16;
  const int R = (1 << p.depth) - 1, W = 2 * R + 1;
  const int TS = RPE == 1 ? ((W + W * W + 3) & ~3) : RPE == 2 ? ((3 * W + 3) & ~3) : 0;
  const int nhw = blockDim.x >> 6;
  // the tables come FIRST: form 2 packs two LDS byte addresses into the halves of one register, so they must stay below
  // 64 KiB (the launcher checks nhw * TS * 4 + 12 W < 65536)
  float* s_tab = reinterpret_cast<float*>(smem);                       // [nhw][TS] * log2e
  int4* s_qry0 = reinterpret_cast<int4*>(s_tab + nhw * TS);            // [2][LP] {4x, 4(yW+z) | packed 4y, 4z, id, row}
  int2* s_key0 = reinterpret_cast<int2*>(s_qry0 + 2 * LP);             // [2][LP] {4(R-x), 4(W+(R-y)W+R-z) | packed}
  int* s_kbid0 = reinterpret_cast<int*>(s_key0 + 2 * LP);              // [2][LP] batch id, -1 dead
  unsigned char* s_v0 = reinterpret_cast<unsigned char*>(s_kbid0 + 2 * LP);   // [nhw][2 NP * 16 rows][64 B]  V rows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int hw = tid >> 6;
  const int wg_lin = blockIdx.y * gridDim.x + blockIdx.x;                // dispatch order (probes)
  const int gy = blockIdx.y;
  const int h = gy * nhw + hw;
  const int c = lane & 15, g = lane >> 4;

  if (RPE) {
    const float4* src = reinterpret_cast<const float4*>(p.rpe2 + (size_t)gy * nhw * TS);
    float4* dst = reinterpret_cast<float4*>(s_tab);
    for (int i = tid; i < nhw * TS / 4; i += blockDim.x) dst[i] = src[i];
  }
  const int hu = __builtin_amdgcn_readfirstlane(hw);
  const int tabb = (int)(size_t)(s_tab + hu * TS);
  unsigned char* s_v = s_v0 + hu * (2 * NP * 16) * 64;                 // this wave's V image
  // rows past the real key tiles of the last pair must read as zeros (0 * garbage could be NaN)
  for (int i = lane; i < (2 * NP * 16 - LP) * 4; i += 64)
    reinterpret_cast<uint4*>(s_v + LP * 64)[i] = make_uint4(0u, 0u, 0u, 0u);
  const float mask2 = kMaskValue * 1.4426950408889634f;
  const float rt_add = (g == 0) ? 0.f : kDeadValue;   // the relay key lives in the g == 0 lanes only
  // (probe build 3: head-group-major addressing of the SAME byte volume -- [head group][row][Q | K | V][heads of the group] --
  // to price the operand layout; results are garbage)
  const uint32_t hgw = (uint32_t)nhw * 64u;             // bytes of one region of one head group
  const uint32_t row_q = HFL_ATT_TRACE == 3 ? 3u * hgw : (uint32_t)(3 * C) * 4u;       // bytes per qkv row (same as fp32 qkv)
  const char* qkv_b = reinterpret_cast<const char*>(p.qkv);
  char* out_b = reinterpret_cast<char*>(p.out);
  const uint32_t rows_all = (uint32_t)(G > 0 ? p.rt_row0 + p.n_windows : p.n_tokens);
  const uint32_t hg_base = HFL_ATT_TRACE == 3 ? (uint32_t)gy * rows_all * 3u * hgw : 0u;
  // this lane's 16-B chunk of the head's 64 B in the Q region (K and V: + C * 4, + C * 8)
  const uint32_t col_l = HFL_ATT_TRACE == 3 ? hg_base + (uint32_t)hw * 64u + (uint32_t)(lane & 3) * 16u
                                            : (uint32_t)h * 64u + (uint32_t)(lane & 3) * 16u;
  // LDS block of the wave, 64-B rows: 16-B chunk j of row r sits in slot j ^ ((r >> 2) & 3), so that both the row-per-quad
  // accesses (lane l: row l >> 2, chunk l & 3) and the operand accesses (lane (c, g): row c, chunk g) are conflict-free
  const uint32_t reg_k = HFL_ATT_TRACE == 3 ? hgw : (uint32_t)C * 4u;              // Q -> K -> V region stride
  const int st_quad = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  const int st_row = c * 64, st_x = (c >> 2) & 3;
  // transposed V reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of its 4-key block
  const int tr_off = ((4 * g + (c >> 2)) * 64) + (c & 3) * 8;

  const int n_tok = (int)p.n_tokens;
  const bool owns = tid < LP;                          // the launcher guarantees blockDim.x >= LP
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);

  const int tstep = p.D;
  const bool nt_loads = (p.dbg & 16) != 0;
  // ---- request a window: metadata word + every fragment of this wave's head (index arithmetic only) ----------------
  // A quad of lanes reads ONE row's 64 B (lane l: row l >> 2 of the tile, 16-B chunk l & 3): 16 requests per load
  // instruction instead of the 64 of the MFMA operand mapping (lane (c, g): row c, chunk g -- four rows per quad), and the
  // query row comes in once instead of as a [hi | hi] and a [lo | lo] load.  The operand mapping is restored through this
  // wave's LDS block (stage_in below).  Measured with the addresses alone swapped: -13 % on the depth-4 launch.
  auto request = [&](int w, uint2& mt, uint4 (&kr)[T], uint4 (&qr)[T], uint4 (&vr)[T]) {
    const int tok0 = (p.D == 1) ? w * K : (w / p.D) * K * p.D + (w % p.D);
    const int rt_row = (int)p.rt_row0 + w;
    mt = make_uint2(0u, 0xFFFFFFFFu);
    if (owns && tid < K) {
      const int t = tok0 + tid * tstep;
      if (t < n_tok) mt = *reinterpret_cast<const uint2*>(p.meta + 2 * (int64_t)t);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int row;
      bool ok;
      if (G > 0 && t == T - 1) {
        row = rt_row;
        ok = (lane >> 2) == 0;
      } else {
        row = tok0 + (t * 16 + (lane >> 2)) * tstep;
        ok = row < n_tok;
      }
      kr[t] = qr[t] = vr[t] = zero4;
      if (ok) {
        const char* base = qkv_b + (uint32_t)row * row_q + col_l;
        if (nt_loads) {        // probe (window_debug bit 4): the operand rows are read once -- non-temporal loads
          typedef unsigned int att_u4v __attribute__((ext_vector_type(4)));
          const att_u4v a = __builtin_nontemporal_load(reinterpret_cast<const att_u4v*>(base));
          const att_u4v b = __builtin_nontemporal_load(reinterpret_cast<const att_u4v*>(base + reg_k));
          const att_u4v c = __builtin_nontemporal_load(reinterpret_cast<const att_u4v*>(base + 2u * reg_k));
          qr[t] = make_uint4(a[0], a[1], a[2], a[3]);
          kr[t] = make_uint4(b[0], b[1], b[2], b[3]);
          vr[t] = make_uint4(c[0], c[1], c[2], c[3]);
        } else {
          qr[t] = *reinterpret_cast<const uint4*>(base);
          kr[t] = *reinterpret_cast<const uint4*>(base + reg_k);
          vr[t] = *reinterpret_cast<const uint4*>(base + 2u * reg_k);
        }
      }
    }
  };
  const bool trace_on = HFL_ATT_TRACE && (p.dbg & 8) && blockIdx.x == 0 && blockIdx.y == 0 && hw == 0;
  if (HFL_ATT_TRACE && (p.dbg & 8) && tid == 0 && wg_lin < 4096) g_att_wg[2 * wg_lin] = __builtin_amdgcn_s_memrealtime();
  int it = 0;
  for (int w = wg_x; w < p.n_windows; w += wg_nx, ++it) {
    const int tok0 = (p.D == 1) ? w * K : (w / p.D) * K * p.D + (w % p.D);
    const int rt_row = (int)p.rt_row0 + w;
    uint2 mt;
    uint4 kr[T], qr[T], vr[T];
    HFL_ATT_STAMP(0)
    if (HFL_ATT_TRACE && trace_on && it < 16 && lane == 0) g_att_trace[it * 8 + 7] = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    request(w, mt, kr, qr, vr);
    int4* s_qry = s_qry0 + (it & 1) * LP;
    int2* s_key = s_key0 + (it & 1) * LP;
    int* s_kbid = s_kbid0 + (it & 1) * LP;
    if (owns) {
      const int j = tid;
      int bid = -1, row = -1;
      int x = 0, y = 0, z = 0;
      if (j < K) {
        if (mt.y != 0xFFFFFFFFu) {
          x = (int)(mt.x & 1023u); y = (int)((mt.x >> 10) & 1023u); z = (int)(mt.x >> 20);
          bid = (int)mt.y;
          row = tok0 + j * tstep;
        }
      } else if (G > 0 && j == K) {
        row = rt_row;
      }
      if (RPE == 2) {   // byte offsets into [X | Y | Z]: low half y, high half z (each < 12 W < 65536, sums included)
        s_key[j] = make_int2(4 * (R - x), (4 * (W + R - y)) | ((4 * (2 * W + R - z)) << 16));
        s_qry[j] = make_int4(4 * x, (4 * y) | ((4 * z) << 16), bid, row);
      } else {
        s_key[j] = make_int2(4 * (R - x), 4 * (W + (R - y) * W + (R - z)));
        s_qry[j] = make_int4(4 * x, 4 * (y * W + z), bid, row);
      }
      s_kbid[j] = bid;
    }
    // row-per-quad registers -> MFMA operand registers through the wave's LDS block (LDS operations of one wave execute in
    // order, so the block is reused: Q, then K, then the V image that the transposed reads below need)
    uint4 ka[T], qh[T], ql[T];
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<uint4*>(s_v + t * 1024 + st_quad) = qr[t];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      qh[t] = *reinterpret_cast<const uint4*>(s_v + t * 1024 + st_row + (((g & 1) ^ st_x) * 16));         // [q_hi | q_hi]
      ql[t] = *reinterpret_cast<const uint4*>(s_v + t * 1024 + st_row + (((2 + (g & 1)) ^ st_x) * 16));   // [q_lo | q_lo]
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<uint4*>(s_v + t * 1024 + st_quad) = kr[t];
#pragma unroll
    for (int t = 0; t < T; ++t) ka[t] = *reinterpret_cast<const uint4*>(s_v + t * 1024 + st_row + ((g ^ st_x) * 16));
    // V rows as loaded, unswizzled: row (tile, key) = 64 B [16 x hi | 16 x lo]
#pragma unroll
    for (int t = 0; t < T; ++t) *reinterpret_cast<uint4*>(s_v + t * 1024 + lane * 16) = vr[t];
    __syncthreads();
    HFL_ATT_STAMP(1)
    const int bid0 = s_kbid[0], bidl = s_kbid[K - 1];
    const int rt_bid = bid0 >= 0 ? bid0 : p.batch;     // the relay token carries the id of the window's first token
    const bool homog = __builtin_amdgcn_readfirstlane((bidl >= 0 && bid0 == bidl) ? 1 : 0) != 0;

    // V^T fragments of every key-tile pair: elements 0..3 = keys 4g..4g+3 of tile 2p, 4..7 = of tile 2p+1, column d = c
    att_h8 vhi[NP], vlo[NP];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      typedef __attribute__((address_space(3))) att_s4 lds_s4;
      const unsigned char* b0 = s_v + (2 * pp) * 16 * 64 + tr_off;
      const att_s4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(b0));
      const att_s4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(b0 + 32));
      const att_s4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(b0 + 16 * 64));
      const att_s4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(b0 + 16 * 64 + 32));
      const short __attribute__((ext_vector_type(8))) hh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
      const short __attribute__((ext_vector_type(8))) ll = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
      vhi[pp] = __builtin_bit_cast(att_h8, hh);
      vlo[pp] = __builtin_bit_cast(att_h8, ll);
    }

    // table offsets of this lane's keys stay in registers (24 at K = 48).  (Re-reading them per query tile to fit 128 VGPRs and
    // a fourth wave per SIMD: depth 5 53 -> 51 us, depth 4 70 -> 78 us with 11 spilled dwords, depth 3 21 -> 29 us: dropped.)
    int kxa[TW][4], kyza[TW][4];
    if (RPE) {
#pragma unroll
      for (int kt = 0; kt < TW; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int2 km = s_key[kt * 16 + 4 * g + r];
          kxa[kt][r] = km.x;
          kyza[kt][r] = km.y;
        }
    }

    HFL_ATT_STAMP(2)

    auto body = [&](auto masked_tag) {
      constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
      for (int qt = 0; qt < T; ++qt) {           // TW token query tiles [+ the relay token as a tile with one live column]
        const bool is_rt = (G > 0 && qt == T - 1);
        // the relay tile needs no metadata (no RPE, its id is rt_bid, its row rt_row): do not read slots >= K
        const int4 qm = is_rt ? make_int4(0, 0, -1, -1) : s_qry[qt * 16 + c];
        const int q_bid = is_rt ? rt_bid : qm.z;
        const int qxa = qm.x + tabb, qyza = qm.y + (RPE == 2 ? tabb * 0x10001 : tabb);
        const att_h8 bqh = __builtin_bit_cast(att_h8, qh[qt]);
        const att_h8 bql = __builtin_bit_cast(att_h8, ql[qt]);
        if (HFL_ATT_TRACE >= 2) {      // probe build: the kernel's memory pattern alone (every load consumed, every row stored)
          const uint4 m = make_uint4(ka[qt].x ^ qh[qt].x ^ ql[qt].x ^ (uint32_t)vhi[qt / 2][0], ka[qt].y ^ qh[qt].y ^ ql[qt].y,
                                     ka[qt].z ^ qh[qt].z ^ ql[qt].z, ka[qt].w ^ qh[qt].w ^ ql[qt].w);
          const int orow_m = is_rt ? (c == 0 ? rt_row : -1) : qm.w;
          if (orow_m >= 0)
            att_store_row4(out_b, (uint32_t)orow_m, C, h * 16 + 4 * g, __builtin_bit_cast(f32x4, m), p.out_split);
          continue;
        }

        f32x4 s[T];
#pragma unroll
        for (int kt = 0; kt < T; ++kt) {
          const att_h8 ak = __builtin_bit_cast(att_h8, ka[kt]);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, bql, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ak, bqh, acc, 0, 0, 0);
          s[kt] = acc;
        }
        if (RPE == 1 && !is_rt) {                // no RPE for the relay row / column (octformer_backbone.py:78-80)
#pragma unroll
          for (int kt = 0; kt < TW; ++kt) {
            f32x4 bx, byz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              bx[r] = *reinterpret_cast<lds_f32*>(kxa[kt][r] + qxa);
              byz[r] = *reinterpret_cast<lds_f32*>(kyza[kt][r] + qyza);
            }
            s[kt] += bx + byz;
          }
        }
        if (RPE == 2 && !is_rt) {
#pragma unroll
          for (int kt = 0; kt < TW; ++kt) {
            f32x4 bx, by, bz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const uint32_t t = (uint32_t)(kyza[kt][r] + qyza);           // both halves are LDS byte addresses
              bx[r] = *reinterpret_cast<lds_f32*>(kxa[kt][r] + qxa);
              by[r] = *reinterpret_cast<lds_f32*>((int)(t & 0xFFFFu));
              bz[r] = *reinterpret_cast<lds_f32*>((int)(t >> 16));
            }
            s[kt] += (bx + by) + bz;
          }
        }
        if (MASKED) {
#pragma unroll
          for (int kt = 0; kt < TW; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (s_kbid[kt * 16 + 4 * g + r] != q_bid) s[kt][r] += mask2;
        }
        float mx = kDeadValue;
        float srt = kDeadValue;
        if (G > 0) {   // relay key: position K = tile T-1, k-slot group 0, register 0
          srt = s[T - 1][0] + rt_add;
          if (MASKED && rt_bid != q_bid) srt += mask2;
          mx = srt;
        }
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          mx = att_max3_c(mx, s[kt][0], s[kt][1]);
          mx = att_max3_c(mx, s[kt][2], s[kt][3]);
        }
        mx = att_rows_max(mx);
        const f32x4 nmx4 = {-mx, -mx, -mx, -mx};
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < TW; ++kt) {
          f32x4 e = s[kt] + nmx4;
          e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
          e[2] = __builtin_amdgcn_exp2f(e[2]); e[3] = __builtin_amdgcn_exp2f(e[3]);
          s[kt] = e;
          sum4 += e;
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
        if (G > 0) {
          const float ert = __builtin_amdgcn_exp2f(srt - mx);
          s[T - 1] = (f32x4){ert, 0.f, 0.f, 0.f};         // zero in the lanes g != 0 (rt_add)
          sum += ert;
        }
        sum = att_rows_sum(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);

        // O^T = V^T P^T over pairs of key tiles; P (un-normalised, in [0, 1]) split into fp16 (hi, lo) in registers
        // (one accumulator per pair: two independent chains of three MFMAs instead of one of six)
        f32x4 oacc[NP];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
          const f32x4 pa = s[2 * pp];
          const f32x4 pb = (2 * pp + 1 < T) ? s[2 * pp + 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
          typedef unsigned int att_u4 __attribute__((ext_vector_type(4)));
          unsigned int h0, h1, h2 = 0u, h3 = 0u, l0, l1, l2 = 0u, l3 = 0u;
          att_split_pair_f16(pa[0], pa[1], h0, l0);
          att_split_pair_f16(pa[2], pa[3], h1, l1);
          if (G > 0 && 2 * pp + 1 == T - 1) {            // the relay key tile: one live element (s[T - 1] above)
            att_split_pair_f16(pb[0], 0.f, h2, l2);
            l2 &= 0xFFFFu;
          } else if (2 * pp + 1 < T) {
            att_split_pair_f16(pb[0], pb[1], h2, l2);
            att_split_pair_f16(pb[2], pb[3], h3, l3);
          }
          const att_u4 uh = {h0, h1, h2, h3}, ul = {l0, l1, l2, l3};
          const att_h8 phi = __builtin_bit_cast(att_h8, uh);
          const att_h8 plo = __builtin_bit_cast(att_h8, ul);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi[pp], plo, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vlo[pp], phi, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vhi[pp], phi, o, 0, 0, 0);
          oacc[pp] = o;
        }
        f32x4 o = oacc[0];
#pragma unroll
        for (int pp = 1; pp < NP; ++pp) o += oacc[pp];
        o *= inv;
        // the accumulator holds channels 4g .. 4g+3 of query c (relay tile: only the column c == 0 is a row)
        const int orow = is_rt ? (c == 0 ? rt_row : -1) : qm.w;
        if (p.out_split == 2) {
          // split2 rows for the proj GEMM: the tile's 16 x [16 hi | 16 lo] bf16 goes through the LDS block (its V image is in
          // registers by now) and leaves as ONE 16-B store per lane, a quad of lanes per row (two 32-B segments of one line)
          uint2 hi, lo;
          x3_split_pair_scalar(o[0], o[1], hi.x, lo.x);
          x3_split_pair_scalar(o[2], o[3], hi.y, lo.y);
          *reinterpret_cast<uint2*>(s_v + st_row + (((g >> 1) ^ st_x) * 16) + (g & 1) * 8) = hi;
          *reinterpret_cast<uint2*>(s_v + st_row + (((2 + (g >> 1)) ^ st_x) * 16) + (g & 1) * 8) = lo;
          const uint4 v = *reinterpret_cast<const uint4*>(s_v + st_quad);
          const int rl = lane >> 2, ch = lane & 3;
          const int orow_l = is_rt ? (rl == 0 ? rt_row : -1) : s_qry[qt * 16 + rl].w;
          if (orow_l >= 0)
            *reinterpret_cast<uint4*>(out_b + (size_t)orow_l * (uint32_t)(4 * C) +
                                      (uint32_t)((h >> 1) * 128 + (h & 1) * 32 + (ch & 1) * 16 + (ch >> 1) * 64)) = v;
        } else if (orow >= 0) {
          att_store_row4(out_b, (uint32_t)orow, C, h * 16 + 4 * g, o, p.out_split);
        }
        if (HFL_ATT_TRACE && trace_on) {        // the stamp must not float above the tile: tie it to the tile's result
          const unsigned long long t_ = __builtin_amdgcn_s_memtime() + (o[0] == 1234.5f ? 1 : 0);
          if (it < 16 && lane == 0 && qt < 5) g_att_trace[it * 8 + 3 + qt] = t_;
        }
      }
    };
    if (homog)
      body(std::false_type{});
    else
      body(std::true_type{});
  }
  if (HFL_ATT_TRACE && (p.dbg & 8) && tid == 0 && wg_lin < 4096) g_att_wg[2 * wg_lin + 1] = __builtin_amdgcn_s_memrealtime();
}

// expanded RPE table of v4: out (H, TS), TS = (W + W*W + 3) & ~3, W = 2R+1, R = 2^depth - 1 <= pos_bnd:
//   out[h][i]             = table[(i - R + bnd), h] * log2e                       i in [0, W)   (x axis)
//   out[h][W + iy*W + iz] = (table[nrpe + iy - R + bnd, h] + table[2 nrpe + iz - R + bnd, h]) * log2e
__global__ void __launch_bounds__(256)
rpe_expand_kernel(float* __restrict__ out, const float* __restrict__ table, int H, int bnd, int R) {
  const int W = 2 * R + 1;
  const int TS = (W + W * W + 3) & ~3;
  const int nrpe = 2 * bnd + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * TS) return;
  const int h = i / TS, e = i % TS;
  float v = 0.f;
  if (e < W) {
    v = table[(e - R + bnd) * H + h];
  } else if (e < W + W * W) {
    const int iy = (e - W) / W, iz = (e - W) % W;
    v = table[(nrpe + iy - R + bnd) * H + h] + table[(2 * nrpe + iz - R + bnd) * H + h];
  }
  out[i] = v * 1.4426950408889634f;
}

// Expanded table, form 2 (deep octrees: depth 6-7, or coordinates beyond pos_bnd): three 1-D tables over the FULL
// coordinate difference range with the reference's clamp baked in,
//   out[h][a * W + i] = table[a * nrpe + clamp(i - R, -bnd, bnd) + bnd, h] * log2e,   a = x, y, z;  TS = (3 W + 3) & ~3
// -- three lookups and no clamp arithmetic per score (the (2R+1)^2 y-z table of form 1 would be 64 KB per head at depth 6)
__global__ void __launch_bounds__(256)
rpe_expand3_kernel(float* __restrict__ out, const float* __restrict__ table, int H, int bnd, int R) {
  const int W = 2 * R + 1;
  const int TS = (3 * W + 3) & ~3;
  const int nrpe = 2 * bnd + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * TS) return;
  const int h = i / TS, e = i % TS;
  float v = 0.f;
  if (e < 3 * W) {
    const int a = e / W;
    int d = e % W - R;
    d = d < -bnd ? -bnd : (d > bnd ? bnd : d);
    v = table[(a * nrpe + d + bnd) * H + h];
  }
  out[i] = v * 1.4426950408889634f;
}

// which expanded form a (depth, pos_bnd) pair takes: 1 = x table + y-z table (no clamp needed, depth <= 5), 2 = three
// clamped 1-D tables (depth <= 7), 0 = none (the three-lookup kernel v2 reads the original table)
// which expanded form a (depth, pos_bnd) pair takes, per consumer: the fp32 kernel (v4) knows form 1 only; the fp16 kernel (v5)
// takes form 1 up to depth 4 and form 2 beyond -- at depth 5 the (2R+1)^2 y-z table is 16 KB per head, which left room for
// 6 waves per CU only; the three 1-D tables (768 B per head) let 12 stay resident: 61.5 -> 53 us on the depth-5 bench launch
// in spite of the third lookup (at depth 4, 4 KB per head, form 1 is the faster one: 70 vs 73 us)
static int g_rpe_form1_max_depth = 4;       // probe knob 'window_rpe_form1_max_depth' (fp16 kernel)
static inline int rpe_form(int depth, int bnd, int f16) {
  if (depth < 1 || depth > 7) return 0;
  const bool fits = depth <= 5 && ((1 << depth) - 1) <= bnd;
  if (!f16) return fits ? 1 : 0;
  if (fits && depth <= g_rpe_form1_max_depth) return 1;
  return 2;
}
static inline size_t rpe_form_floats(int depth, int bnd, int f16) {      // per head
  const int W = 2 * ((1 << depth) - 1) + 1;
  const int f = rpe_form(depth, bnd, f16);
  return f == 1 ? (size_t)((W + W * W + 3) & ~3) : f == 2 ? (size_t)((3 * W + 3) & ~3) : 0;
}

static int g_window_variant = 4;
static int g_window_v4_wgs_per_cu = 1;   // multiples of the resident workgroup count
static int g_window_dbg = 0;
static int g_window_v2_wgs_per_cu = 16;
static int g_window_heads_per_wg = 4;
static int g_relay_fast = 1;            // probe (hfl_set_variant "relay_fast")
static int g_window_bwd_rt = -1;        // probe ("window_bwd_rt"): 0 forces the scatter-add table gradient of the backward

// fp16 (hi, lo) operand layout: only the v5 kernel reads it (callers ask hfl_window_attention_f16_ok first).  `n` problems of
// one shape in ONE launch (see WinMultiParams); HFL_EINVAL when they cannot share a launch (the caller then launches them
// one by one).  Heads per workgroup: 4, or 2 when the expanded RPE tables of 4 heads do not leave room in LDS.
// per-launch HIP events around the v5 launches (bench.py's roofline leg times the launches the product path really issues --
// including those inside hfl_block_forward_x3 / hfl_block_attention_x3_multi, which no Python-side timer sees)
struct AttnTimingRec {
  hipEvent_t e0, e1;
  double bytes, flops;
};
static int g_attn_timing = 0;
static std::vector<AttnTimingRec> g_attn_recs;
static std::mutex g_attn_mu;                     // launches may come from autograd / side-stream host threads
static void attn_rec_drop(AttnTimingRec& r) {
  if (r.e0) (void)hipEventDestroy(r.e0);
  if (r.e1) (void)hipEventDestroy(r.e1);
  r.e0 = r.e1 = nullptr;
}

template <int T, int G>
static int launch_window_v5(const WinParams* ps, int n, hipStream_t s) {
  constexpr int LP = T * 16;
  constexpr int np5 = (T + 1) / 2;
  if (n < 1 || n > kWinMulti) return HFL_EINVAL;
  int form0 = 0, hp0 = 0;
  size_t lds_max = 0;
  int64_t windows = 0;
  for (int i = 0; i < n; ++i) {
    const WinParams& p = ps[i];
    int hpw = g_window_heads_per_wg;             // heads (= waves) per workgroup, at most 4
    if (hpw < 1 || hpw > 4 || p.H % hpw != 0) hpw = (p.H % 4 == 0) ? 4 : (p.H % 2 == 0) ? 2 : 1;
    const int64_t rows_total = G > 0 ? p.rt_row0 + p.n_windows : p.n_tokens;
    const int form = p.table ? rpe_form(p.depth, p.bnd, 1) : 0;
    const size_t ts5 = p.table ? rpe_form_floats(p.depth, p.bnd, 1) : 0;
    int hp5 = hpw;
    size_t lds5 = 0;
    for (;; hp5 >>= 1) {
      lds5 = (size_t)2 * LP * (16 + 8 + 4) + (size_t)hp5 * ts5 * 4 + (size_t)hp5 * (2 * np5 * 16) * 64;
      if (lds5 <= 72 * 1024 || hp5 <= 2 || p.H % (hp5 / 2) != 0) break;
    }
    const int W5 = 2 * ((1 << (p.depth > 0 && p.depth <= 7 ? p.depth : 0)) - 1) + 1;
    if (!p.qkv_f16 || p.depth < 1 || p.depth > 7 || (p.table != nullptr && (p.rpe2 == nullptr || form == 0)) ||
        (p.table == nullptr && p.depth > 10) || (form == 2 && (size_t)hp5 * ts5 * 4 + 12 * (size_t)W5 >= 65536) ||
        rows_total * 3 * p.H * 16 * 4 >= (int64_t)1 << 32 || lds5 > 72 * 1024 || hp5 * 64 < LP || p.qkv_bias != nullptr ||
        p.n_windows < 1)
      return HFL_EINVAL;
    if (i == 0) {
      form0 = form;
      hp0 = hp5;
    } else if (form != form0 || hp5 != hp0 || p.H != ps[0].H || p.K != ps[0].K) {
      return HFL_EINVAL;
    }
    if (lds5 > lds_max) lds_max = lds5;
    windows += p.n_windows;
  }
  const int groups5 = ps[0].H / hp0;
  // persistent grid: exactly the workgroups that are resident at once (waves-per-SIMD target of the kernel, LDS), times
  // g_window_v4_wgs_per_cu, dealt to the problems in proportion to their windows (at least one each)
  int resident = v4_waves_per_simd(T, G) * 4 / hp0;
  const int lds_fit = (int)((size_t)160 * 1024 / (lds_max + 512));
  if (resident > lds_fit) resident = lds_fit;
  if (resident < 1) resident = 1;
  int64_t px = (int64_t)hfl_stream_cus(s) * resident * g_window_v4_wgs_per_cu / groups5;
  if (px < n) px = n;
  WinMultiParams m;
  m.n = n;
  m.first[0] = 0;
  for (int i = 0; i < n; ++i) {
    int64_t share = windows <= px ? ps[i].n_windows : (ps[i].n_windows * px + windows - 1) / windows;
    if (share < 1) share = 1;
    if (share > ps[i].n_windows) share = ps[i].n_windows;
    m.p[i] = ps[i];
    m.first[i + 1] = m.first[i] + (int)share;
  }
  for (int i = n; i < kWinMulti; ++i) {
    m.p[i] = ps[0];
    m.first[i + 1] = m.first[n];
  }
  dim3 grid5((unsigned)m.first[n], (unsigned)groups5);
  hipError_t e;
#define HFL_V5_LAUNCH(F)                                                                                  \
  {                                                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel_v5<T, G, F>),                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);                    \
    if (e != hipSuccess) return (int)e;                                                                   \
    window_attn_kernel_v5<T, G, F><<<grid5, hp0 * 64, lds_max, s>>>(m);                                   \
  }
  AttnTimingRec rec{};
  if (g_attn_timing) {
    // algorithmic bytes: q, k, v read + out written = 16 B per (row, channel); 4 L^2 C FLOP per window
    for (int i = 0; i < n; ++i) {
      const double rows = (double)(G > 0 ? ps[i].rt_row0 + ps[i].n_windows : ps[i].n_tokens);
      const double real_windows = (double)((ps[i].n_tokens + ps[i].K - 1) / ps[i].K);
      rec.bytes += rows * ps[i].H * 16 * 16.0;
      rec.flops += 4.0 * (ps[i].K + G) * (ps[i].K + G) * ps[i].H * 16 * real_windows;
    }
    if (hipEventCreate(&rec.e0) != hipSuccess || hipEventCreate(&rec.e1) != hipSuccess) {
      attn_rec_drop(rec);
      return HFL_EINVAL;
    }
    e = hipEventRecord(rec.e0, s);
    if (e != hipSuccess) {
      attn_rec_drop(rec);
      return (int)e;
    }
  }
  const bool timed = rec.e0 != nullptr;
#define HFL_V5_FAIL_IF(cond)    \
  if (cond) {                   \
    if (timed) attn_rec_drop(rec); \
    return (int)e;              \
  }
#undef HFL_V5_LAUNCH
#define HFL_V5_LAUNCH(F)                                                                                  \
  {                                                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel_v5<T, G, F>),                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);                    \
    HFL_V5_FAIL_IF(e != hipSuccess)                                                                       \
    window_attn_kernel_v5<T, G, F><<<grid5, hp0 * 64, lds_max, s>>>(m);                                   \
  }
  if (form0 == 0) HFL_V5_LAUNCH(0) else if (form0 == 1) HFL_V5_LAUNCH(1) else HFL_V5_LAUNCH(2)
#undef HFL_V5_LAUNCH
  if (timed) {
    e = hipEventRecord(rec.e1, s);
    HFL_V5_FAIL_IF(e != hipSuccess)
    std::lock_guard<std::mutex> lk(g_attn_mu);
    g_attn_recs.push_back(rec);
  }
#undef HFL_V5_FAIL_IF
  HFL_RETURN_LAST_ERROR();
}

template <int T, int G>
static int launch_window(const WinParams& p, hipStream_t s) {
  constexpr int LP = T * 16;
  const int nrpe = 2 * p.bnd + 1;
  int blocks = p.n_windows;
  const int cap = hfl_stream_cus(s) * 4;
  if (blocks > cap) blocks = cap;
  {
    int hpw = g_window_heads_per_wg;             // heads (= waves) per workgroup, at most 4
    if (hpw < 1 || hpw > 4 || p.H % hpw != 0) hpw = (p.H % 4 == 0) ? 4 : (p.H % 2 == 0) ? 2 : 1;
    const size_t lds = (p.table ? (size_t)hpw * 3 * nrpe * 4 : 0) + (size_t)LP * (16 + 16 + 4);
    const int groups = p.H / hpw;
    int bx = p.n_windows;
    const int capx = hfl_stream_cus(s) * g_window_v2_wgs_per_cu / groups;
    if (bx > capx) bx = capx;
    dim3 grid((unsigned)bx, (unsigned)groups);
    const int64_t rows_total = G > 0 ? p.rt_row0 + p.n_windows : p.n_tokens;
    const int R4 = (1 << (p.depth > 0 && p.depth <= 5 ? p.depth : 0)) - 1, W4 = 2 * R4 + 1;
    const size_t ts4 = p.table ? (size_t)((W4 + W4 * W4 + 3) & ~3) : 0;
    const size_t lds4 = (size_t)2 * LP * (16 + 8 + 4) + (size_t)hpw * ts4 * 4;
    if (p.qkv_f16) {
      return launch_window_v5<T, G>(&p, 1, s);
    } else if (g_window_variant == 4 && !p.clamp && p.depth >= 1 && p.depth <= 5 &&
        (p.table == nullptr || p.rpe2 != nullptr) && rows_total * 3 * p.H * 16 * 4 < (int64_t)1 << 32 &&
        lds4 <= 72 * 1024 && hpw * 64 >= LP) {
      // persistent grid: exactly the workgroups that are resident at once (waves-per-SIMD target of
      // the kernel, LDS), times g_window_v4_wgs_per_cu
      int resident = v4_waves_per_simd(T, G) * 4 / hpw;
      const int lds_fit = (int)((size_t)160 * 1024 / (lds4 + 512));
      if (resident > lds_fit) resident = lds_fit;
      if (resident < 1) resident = 1;
      int px = hfl_stream_cus(s) * resident * g_window_v4_wgs_per_cu / groups;
      if (px < 1) px = 1;
      if (px > p.n_windows) px = p.n_windows;
      dim3 grid4((unsigned)px, (unsigned)groups);
      if (lds4 > 48 * 1024) {
        hipError_t e;
        if (p.table == nullptr)
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel_v4<T, G, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        else
          e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_kernel_v4<T, G, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4);
        if (e != hipSuccess) return (int)e;
      }
      if (p.table == nullptr)
        window_attn_kernel_v4<T, G, false><<<grid4, hpw * 64, lds4, s>>>(p);
      else
        window_attn_kernel_v4<T, G, true><<<grid4, hpw * 64, lds4, s>>>(p);
    } else if (p.table == nullptr)
      window_attn_kernel_v2<T, G, false, false><<<grid, hpw * 64, lds, s>>>(p);
    else if (p.clamp)
      window_attn_kernel_v2<T, G, true, true><<<grid, hpw * 64, lds, s>>>(p);
    else
      window_attn_kernel_v2<T, G, false, true><<<grid, hpw * 64, lds, s>>>(p);
  }
  HFL_RETURN_LAST_ERROR();
}

// ------------------------------------------------------------------------------
// Relay-token self-attention.  One wave = (cloud, head, 16-query tile); keys are the
// cloud's relay tokens listed in seq_rows; two passes over the key tiles (statistics,
// then P V) so no accumulator rescaling is needed.
__global__ void __launch_bounds__(256)
relay_attn_kernel(float* __restrict__ out, const float* __restrict__ qkv,
                  const int32_t* __restrict__ seq_rows, const int32_t* __restrict__ seq_off,
                  int H, float scale) {
  const int b = blockIdx.x;
  const int r0 = seq_off[b];
  const int R = seq_off[b + 1] - r0;
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int C = H * 16;
  const int ntile = (R + 15) / 16;
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  // work items of this block: (head, query tile)
  for (int item = blockIdx.y * nwave + wave; item < H * ntile; item += gridDim.y * nwave) {
    const int h = item / ntile, qt = item % ntile;
    const int qi = qt * 16 + c;
    float4 qf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qi < R)
      qf = *reinterpret_cast<const float4*>(qkv + (int64_t)seq_rows[r0 + qi] * 3 * C + h * 16 + 4 * g);
    float m = kDeadValue, l = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      float inv = 0.f;
      if (pass == 1) {
        // combine the four k-slot groups' running (m, l)
        float m2 = fmaxf(m, __shfl_xor(m, 16, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, 32, 64));
        float l2 = l * __expf(m - m2);
        l2 += __shfl_xor(l2, 16, 64);
        l2 += __shfl_xor(l2, 32, 64);
        m = m2;
        inv = 1.0f / l2;
      }
      for (int kt = 0; kt < ntile; ++kt) {
        const int kj = kt * 16 + c;
        float4 kf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kj < R)
          kf = *reinterpret_cast<const float4*>(qkv + (int64_t)seq_rows[r0 + kj] * 3 * C + C + h * 16 + 4 * g);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf.w, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = kt * 16 + 4 * g + r;
          const float v = kk < R ? acc[r] * scale : kDeadValue;
          if (pass == 0) {
            const float mn = fmaxf(m, v);
            l = l * __expf(m - mn) + __expf(v - mn);
            m = mn;
          } else {
            const float pv = __expf(v - m) * inv;
            const float vv = kk < R ? qkv[(int64_t)seq_rows[r0 + kk] * 3 * C + 2 * C + h * 16 + c] : 0.f;
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vv, o, 0, 0, 0);
          }
        }
      }
      if (pass == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oq = qt * 16 + 4 * g + r;
          if (oq < R) out[(int64_t)seq_rows[r0 + oq] * C + h * 16 + c] = o[r];
        }
      }
    }
  }
}


// The same attention on the fp16 (hi, lo) operand rows hfl_ln_qkv_fused writes (per row [Q | K | V] regions of C * 4 bytes, per
// head 64 B = [16 hi | 16 lo] fp16, queries pre-multiplied by scale * log2 e), writing the proj GEMM's bf16 split2 operand
// directly: with LayerNorm -> qkv as one launch in front and no split pass behind, the relay-token block's attention branch is
// three launches instead of six (LayerNorm, qkv GEMM, memset, attention, split2, proj).  fp32 MFMA as above, exp2-domain
// softmax.  Block `batch` of the grid writes zeros to the rows that belong to no sequence (the relay tokens of pure padding
// windows, `orphan_rows`), which the memset did before.
__global__ void __launch_bounds__(256)
relay_attn_f16_kernel(unsigned char* __restrict__ out2, const unsigned char* __restrict__ qkv,
                      const int32_t* __restrict__ seq_rows, const int32_t* __restrict__ seq_off,
                      const int32_t* __restrict__ orphan_rows, int n_orphans, int batch, int H, int relay_fast) {
  const int b = blockIdx.x;
  const int C = H * 16;
  if (b >= batch) {
    const int per_row = C * 4 / 16;                       // 16-B cells of an output row
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < n_orphans * per_row; i += gridDim.y * blockDim.x)
      *reinterpret_cast<uint4*>(out2 + (size_t)orphan_rows[i / per_row] * (size_t)(C * 4) + (size_t)(i % per_row) * 16) =
          make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const int r0 = seq_off[b];
  const int R = seq_off[b + 1] - r0;
  const int lane = threadIdx.x & 63;
  const int c = lane & 15, g = lane >> 4;
  const int ntile = (R + 15) / 16;
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const size_t row_b = (size_t)C * 12;                    // bytes of an operand row: 3 regions of C * 4
  auto ld4 = [&](const unsigned char* p) -> float4 {      // dims 4 g .. 4 g + 3 of a head's 64 B at p: hi + lo
    const uint2 hi = *reinterpret_cast<const uint2*>(p + 8 * g), lo = *reinterpret_cast<const uint2*>(p + 32 + 8 * g);
    auto h = [](unsigned int w, int k) -> float {
      return (float)__builtin_bit_cast(_Float16, (unsigned short)(k ? w >> 16 : w & 0xFFFFu));
    };
    return make_float4(h(hi.x, 0) + h(lo.x, 0), h(hi.x, 1) + h(lo.x, 1), h(hi.y, 0) + h(lo.y, 0), h(hi.y, 1) + h(lo.y, 1));
  };
  if (R <= 0) return;
  if (ntile <= 4 && relay_fast) {
    // Short sequences (every shipped configuration: <= 64 relay tokens per cloud): a (head, query tile) item is ONE memory
    // round trip.  The lane's own row index first (one coalesced read; the others by cross-lane reads), then the query, the
    // four key fragments and the sixteen V elements of the lane are requested together from always-valid addresses (slots past
    // the sequence are clamped to its last row and masked below), scores and probabilities stay in registers (no second pass).
    // The general loop below re-reads the row table per key tile and per V element under a branch: ~40 dependent L2 round
    // trips per item, 23 us for 0.1 MFLOP, ten times per forward on the relay tokens' chain.
    const int myrow = seq_rows[r0 + (lane < R ? lane : R - 1)];
    for (int item = blockIdx.y * nwave + wave; item < H * ntile; item += gridDim.y * nwave) {
      const int h = item / ntile, qt = item % ntile;
      const float4 qf = ld4(qkv + (size_t)__shfl(myrow, qt * 16 + c, 64) * row_b + h * 64);
      float4 kf[4];
      unsigned short vh[4][4], vl[4][4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        kf[kt] = ld4(qkv + (size_t)__shfl(myrow, kt * 16 + c, 64) * row_b + (size_t)C * 4 + h * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned char* vp = qkv + (size_t)__shfl(myrow, kt * 16 + 4 * g + r, 64) * row_b + (size_t)C * 8 + h * 64 + c * 2;
          vh[kt][r] = *reinterpret_cast<const unsigned short*>(vp);
          vl[kt][r] = *reinterpret_cast<const unsigned short*>(vp + 32);
        }
      }
      f32x4 sc[4];
      float m = kDeadValue;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].x, qf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].y, qf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].z, qf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kt].w, qf.w, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[kt][r] = kt * 16 + 4 * g + r < R ? acc[r] : kDeadValue;
          m = fmaxf(m, sc[kt][r]);
        }
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[kt][r] = __builtin_amdgcn_exp2f(sc[kt][r] - m);
          l += sc[kt][r];
        }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vv = (float)__builtin_bit_cast(_Float16, vh[kt][r]) + (float)__builtin_bit_cast(_Float16, vl[kt][r]);
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[kt][r] * inv, vv, o, 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int oq = qt * 16 + 4 * g + r;
        const int orow = __shfl(myrow, oq < 64 ? oq : 63, 64);       // (outside the branch: a cross-lane read needs its source lane active)
        if (oq < R) {
          const uint32_t hb = x3_bf16_rne(o[r]);
          const uint32_t lb = x3_bf16_rne(o[r] - __uint_as_float(hb << 16));
          unsigned char* op = out2 + (size_t)orow * (size_t)(C * 4) + (h >> 1) * 128 + (h & 1) * 32 + c * 2;
          *reinterpret_cast<unsigned short*>(op) = (unsigned short)hb;
          *reinterpret_cast<unsigned short*>(op + 64) = (unsigned short)lb;
        }
      }
    }
    return;
  }
  for (int item = blockIdx.y * nwave + wave; item < H * ntile; item += gridDim.y * nwave) {
    const int h = item / ntile, qt = item % ntile;
    const int qi = qt * 16 + c;
    float4 qf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qi < R) qf = ld4(qkv + (size_t)seq_rows[r0 + qi] * row_b + h * 64);
    float m = kDeadValue, l = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      float inv = 0.f;
      if (pass == 1) {
        float m2 = fmaxf(m, __shfl_xor(m, 16, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, 32, 64));
        float l2 = l * __builtin_amdgcn_exp2f(m - m2);
        l2 += __shfl_xor(l2, 16, 64);
        l2 += __shfl_xor(l2, 32, 64);
        m = m2;
        inv = 1.0f / l2;
      }
      for (int kt = 0; kt < ntile; ++kt) {
        const int kj = kt * 16 + c;
        float4 kf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kj < R) kf = ld4(qkv + (size_t)seq_rows[r0 + kj] * row_b + (size_t)C * 4 + h * 64);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf.w, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kk = kt * 16 + 4 * g + r;
          const float v = kk < R ? acc[r] : kDeadValue;
          if (pass == 0) {
            const float mn = fmaxf(m, v);
            l = l * __builtin_amdgcn_exp2f(m - mn) + __builtin_amdgcn_exp2f(v - mn);
            m = mn;
          } else {
            const float pv = __builtin_amdgcn_exp2f(v - m) * inv;
            float vv = 0.f;
            if (kk < R) {
              const unsigned char* vp = qkv + (size_t)seq_rows[r0 + kk] * row_b + (size_t)C * 8 + h * 64 + c * 2;
              vv = (float)__builtin_bit_cast(_Float16, *reinterpret_cast<const unsigned short*>(vp)) +
                   (float)__builtin_bit_cast(_Float16, *reinterpret_cast<const unsigned short*>(vp + 32));
            }
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, vv, o, 0, 0, 0);
          }
        }
      }
      if (pass == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oq = qt * 16 + 4 * g + r;
          if (oq < R) {
            const uint32_t hb = x3_bf16_rne(o[r]);
            const uint32_t lb = x3_bf16_rne(o[r] - __uint_as_float(hb << 16));
            unsigned char* op = out2 + (size_t)seq_rows[r0 + oq] * (size_t)(C * 4) + (h >> 1) * 128 + (h & 1) * 32 + c * 2;
            *reinterpret_cast<unsigned short*>(op) = (unsigned short)hb;
            *reinterpret_cast<unsigned short*>(op + 64) = (unsigned short)lb;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" {

extern "C" void hfl_internal_set_cpe_chunk(int rows);
/* tuning / A-B hook: select kernel variants at run time (key "window_attention": 1 | 2) */
void hfl_internal_set_x3_dbg(int v);
void hfl_internal_set_window_bwd(int v);
void hfl_internal_set_mlp_stagger(int v);
#ifdef HFL_PROBES
void hfl_internal_set_mlp_dbg(int v);
#endif
void hfl_internal_set_mlp_tail_split(int v);
void hfl_internal_set_mlp_dynamic(int v);
void hfl_internal_set_attn_fused_split(int v);
void hfl_internal_set_ws_map(int v);
#ifdef HFL_PROBES
void hfl_internal_set_ws_dbg(int v);
#endif
void hfl_internal_set_qkv_tail_split(int v);
// bench.py: switch the per-launch timing of the fp16 window kernel on / off (both drop what was recorded) ...
int hfl_internal_rpe_form(int depth, int bnd, int f16) { return rpe_form(depth, bnd, f16); }     // (csrc/attn_fused.hip)

int hfl_internal_attn_timing(int on) {
  std::lock_guard<std::mutex> lk(g_attn_mu);
  for (auto& r : g_attn_recs) attn_rec_drop(r);
  g_attn_recs.clear();
  g_attn_timing = on ? 1 : 0;
  return HFL_OK;
}
// ... and read it: per recorded launch the duration (ms), algorithmic bytes and FLOP; returns the number of launches recorded
int hfl_internal_attn_timing_read(double* ms, double* bytes, double* flops, int cap) {
  std::lock_guard<std::mutex> lk(g_attn_mu);
  int n = 0;
  for (auto& r : g_attn_recs) {
    if (n >= cap) break;
    if (hipEventSynchronize(r.e1) != hipSuccess) return -1;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;
    ms[n] = t;
    bytes[n] = r.bytes;
    flops[n] = r.flops;
    ++n;
  }
  return (int)g_attn_recs.size();
}
#if HFL_ATT_TRACE          // (probe build only: tools/attn_trace.py)
int hfl_internal_read_att_trace(unsigned long long* host, int n) {
  if (n > 16 * 8) n = 16 * 8;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_att_trace), (size_t)n * 8);
}
int hfl_internal_read_att_wg(unsigned long long* host, int n) {
  if (n > 4096 * 2) n = 4096 * 2;
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_att_wg), (size_t)n * 8);
}
#endif
int hfl_set_variant(const char* key, int value) {
  if (key == nullptr) return HFL_EINVAL;
  auto is = [key](const char* name) { return strcmp(key, name) == 0; };
  if (is("reset")) {                             // every probe knob back to its default (tests call it around each case)
    g_window_variant = 4;
    g_window_v4_wgs_per_cu = 1;
    g_window_dbg = 0;
    hfl_internal_attn_timing(0);
    g_rpe_form1_max_depth = 4;
    g_window_v2_wgs_per_cu = 16;
    g_window_heads_per_wg = 4;
    g_relay_fast = 1;
    g_window_bwd_rt = -1;
    hfl_internal_set_window_bwd(2);
    hfl_internal_set_x3_dbg(0);
    hfl_internal_set_x3_dbg(0x100);
    hfl_internal_set_cpe_chunk(-1);
    hfl_internal_set_mlp_stagger(1 | (8 << 8));
#ifdef HFL_PROBES
    hfl_internal_set_mlp_dbg(0);
#endif
    hfl_internal_set_mlp_tail_split(1);
    hfl_internal_set_qkv_tail_split(1);
    hfl_internal_set_mlp_dynamic(1);
  } else if (is("attn_fused_split")) {
    hfl_internal_set_attn_fused_split(value);
  } else if (is("ws_map")) {
    hfl_internal_set_ws_map(value);
#ifdef HFL_PROBES
  } else if (is("ws_dbg")) {
    hfl_internal_set_ws_dbg(value);
#endif
  } else if (is("dynamic_units")) {
    hfl_internal_set_mlp_dynamic(value);
  } else if (is("tail_split")) {
    hfl_internal_set_mlp_tail_split(value);
    hfl_internal_set_qkv_tail_split(value);
#ifdef HFL_PROBES
  } else if (is("mlp_dbg")) {
    hfl_internal_set_mlp_dbg(value);
#endif
  } else if (is("mlp_stagger")) {
    hfl_internal_set_mlp_stagger(value);
  } else if (is("window_attention")) {
    g_window_variant = value;
  } else if (is("window_bwd")) {
    hfl_internal_set_window_bwd(value);
  } else if (is("x3_dbg")) {
    hfl_internal_set_x3_dbg(value);
  } else if (is("cpe_chunk_rows")) {
    hfl_internal_set_cpe_chunk(value);
  } else if (is("window_debug")) {
    g_window_dbg = value;
  } else if (is("window_rpe_form1_max_depth")) {
    g_rpe_form1_max_depth = value;
  } else if (is("window_v4_wgs_per_cu")) {
    g_window_v4_wgs_per_cu = value;
  } else if (is("window_v2_wgs_per_cu")) {
    g_window_v2_wgs_per_cu = value;
  } else if (is("window_heads_per_wg")) {
    g_window_heads_per_wg = value;
  } else if (is("relay_fast")) {                      // probe: 0 = the general loop of relay_attn_f16_kernel for every length
    g_relay_fast = value;
  } else if (is("window_bwd_rt")) {                   // probe: 0 = scatter-add table gradient in the attention backward
    g_window_bwd_rt = value;
  } else {
    return HFL_EINVAL;
  }
  return HFL_OK;
}

/* 1 when hfl_window_attention_fwd_ex accepts the fp16 (hi, lo) qkv operand layout (flag 0x100) for this launch
 * configuration: the v5 kernel needs the expanded RPE table (depth <= 5, no clamp) and <= 72 KiB of LDS */
int hfl_window_attention_f16_ok(const hfl_window_attn_desc* d, int64_t n_rows_total) {
  if (d == nullptr || d->n_heads <= 0 || d->patch_size % 16 != 0) return 0;
  if (rpe_form(d->depth, d->pos_bnd, 1) == 0) return 0;
  if (g_window_variant != 4) return 0;
  const int T = d->patch_size / 16 + d->n_relay;
  if (T < 1 || T > 5) return 0;
  int hpw = g_window_heads_per_wg;
  if (hpw < 1 || hpw > 4 || d->n_heads % hpw != 0) hpw = (d->n_heads % 4 == 0) ? 4 : (d->n_heads % 2 == 0) ? 2 : 1;
  const int LP = T * 16;
  const size_t ts5 = rpe_form_floats(d->depth, d->pos_bnd, 1);
  size_t lds5 = 0;
  for (;; hpw >>= 1) {        // as the launcher: 4 heads per workgroup, or 2 when their tables crowd the LDS
    lds5 = (size_t)2 * LP * (16 + 8 + 4) + (size_t)hpw * ts5 * 4 + (size_t)hpw * (2 * ((T + 1) / 2) * 16) * 64;
    if (lds5 <= 72 * 1024 || hpw <= 2 || d->n_heads % (hpw / 2) != 0) break;
  }
  const int W5 = 2 * ((1 << d->depth) - 1) + 1;
  if (rpe_form(d->depth, d->pos_bnd, 1) == 2 && (size_t)hpw * ts5 * 4 + 12 * (size_t)W5 >= 65536) return 0;
  if (lds5 > 72 * 1024 || hpw * 64 < LP) return 0;
  if (n_rows_total * 3 * d->n_heads * 16 * 4 >= (int64_t)1 << 32) return 0;
  return 1;
}

int hfl_window_attention_fwd(float* out, const float* qkv, const uint32_t* tok_meta,
                             const float* rpe_table, const hfl_window_attn_desc* d,
                             hfl_stream_t stream) {
  return hfl_window_attention_fwd_ex(out, qkv, nullptr, tok_meta, rpe_table, d, 0, stream);
}

}  // extern "C"

// argument checks + kernel parameters of one attention problem; HFL_OK with *empty = true when there is nothing to do
static int win_params(WinParams& p, bool* empty, void* out, const float* qkv, const float* qkv_bias, const uint32_t* tok_meta,
                      const float* rpe_table, const hfl_window_attn_desc* d, int out_split3) {
  *empty = false;
  if (d == nullptr || d->n_windows < 0 || d->n_heads <= 0 || d->n_heads > 16) return HFL_EINVAL;
  if ((qkv_bias != nullptr || out_split3) && g_window_variant < 2) return HFL_EINVAL;
  if ((out_split3 & 3) == 3 || (out_split3 & ~0x103)) return HFL_EINVAL;
  if (d->patch_size % 16 != 0 || d->dilation < 1 || d->n_relay < 0 || d->n_relay > 1) return HFL_EINVAL;
  if (d->n_relay == 1 && d->dilation != 1) return HFL_EINVAL;
  if (d->n_windows == 0) {
    *empty = true;
    return HFL_OK;
  }
  p.out = static_cast<float*>(out); p.qkv = qkv; p.meta = tok_meta; p.table = rpe_table;
  p.qkv_bias = qkv_bias; p.out_split = out_split3 & 3;
  p.qkv_f16 = (out_split3 >> 8) & 1;
  p.rpe2 = rpe_table != nullptr ? d->rpe_expanded : nullptr;
  p.depth = d->depth;
  p.dbg = g_window_dbg;
  p.n_tokens = d->n_tokens; p.rt_row0 = d->rt_row0; p.n_windows = d->n_windows;
  p.K = d->patch_size; p.D = d->dilation; p.H = d->n_heads; p.bnd = d->pos_bnd;
  p.batch = d->batch_size; p.scale = d->scale;
  // coordinates at octree depth `depth` are < 2^depth; 0 (unknown) keeps the clamp
  p.clamp = (d->depth <= 0 || d->depth > 10 || ((1 << d->depth) - 1) > d->pos_bnd) ? 1 : 0;
  return HFL_OK;
}

extern "C" {

// n attention problems (fp16 (hi, lo) qkv operands) in ONE launch when they have one shape, else one launch each
int hfl_window_attention_fwd_multi(int n, void* const* out, const float* const* qkv, const uint32_t* const* tok_meta,
                                   const float* const* rpe_table, const hfl_window_attn_desc* const* desc, int out_split3,
                                   hfl_stream_t stream) {
  if (n < 1 || n > kWinMulti || out == nullptr || qkv == nullptr || tok_meta == nullptr || rpe_table == nullptr ||
      desc == nullptr || !(out_split3 & 0x100))
    return HFL_EINVAL;
  WinParams ps[kWinMulti];
  int live = 0;
  bool same = true;
  for (int i = 0; i < n; ++i) {
    bool empty = false;
    const int rc = win_params(ps[live], &empty, out[i], qkv[i], nullptr, tok_meta[i], rpe_table[i], desc[i], out_split3);
    if (rc != HFL_OK) return rc;
    if (empty) continue;
    if (desc[i]->n_relay != desc[0]->n_relay || desc[i]->patch_size != desc[0]->patch_size) same = false;
    ++live;
  }
  if (live == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = HFL_EINVAL;
  if (same && live > 1) {
    const int T = ps[0].K / 16 + desc[0]->n_relay;
    if (desc[0]->n_relay == 0) {
      switch (T) {
        case 1: rc = launch_window_v5<1, 0>(ps, live, s); break;
        case 2: rc = launch_window_v5<2, 0>(ps, live, s); break;
        case 3: rc = launch_window_v5<3, 0>(ps, live, s); break;
        case 4: rc = launch_window_v5<4, 0>(ps, live, s); break;
        default: break;
      }
    } else {
      switch (T) {
        case 2: rc = launch_window_v5<2, 1>(ps, live, s); break;
        case 3: rc = launch_window_v5<3, 1>(ps, live, s); break;
        case 4: rc = launch_window_v5<4, 1>(ps, live, s); break;
        case 5: rc = launch_window_v5<5, 1>(ps, live, s); break;
        default: break;
      }
    }
    if (rc != HFL_EINVAL) return rc;
  }
  for (int i = 0; i < n; ++i) {          // not one shape (or a single problem): one launch each
    rc = hfl_window_attention_fwd_ex(out[i], qkv[i], nullptr, tok_meta[i], rpe_table[i], desc[i], out_split3, stream);
    if (rc != HFL_OK) return rc;
  }
  return HFL_OK;
}

int hfl_window_attention_fwd_ex(void* out, const float* qkv, const float* qkv_bias,
                                const uint32_t* tok_meta, const float* rpe_table,
                                const hfl_window_attn_desc* d, int out_split3, hfl_stream_t stream) {
  WinParams p;
  bool empty = false;
  const int rc0 = win_params(p, &empty, out, qkv, qkv_bias, tok_meta, rpe_table, d, out_split3);
  if (rc0 != HFL_OK || empty) return rc0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = d->patch_size / 16 + d->n_relay;
  if (d->n_relay == 0) {
    switch (T) {
      case 1: return launch_window<1, 0>(p, s);
      case 2: return launch_window<2, 0>(p, s);
      case 3: return launch_window<3, 0>(p, s);
      case 4: return launch_window<4, 0>(p, s);
      default: return HFL_EINVAL;
    }
  }
  switch (T) {
    case 2: return launch_window<2, 1>(p, s);
    case 3: return launch_window<3, 1>(p, s);
    case 4: return launch_window<4, 1>(p, s);
    case 5: return launch_window<5, 1>(p, s);
    default: return HFL_EINVAL;
  }
}

int64_t hfl_window_rpe_expand_size(int n_heads, int pos_bnd, int depth, int f16_operand) {
  if (n_heads <= 0 || pos_bnd < 0) return 0;
  if (f16_operand == 2) {                  // the fused attention kernels: three 1-D tables at every depth <= 7
    if (depth < 1 || depth > 7) return 0;
    return (int64_t)n_heads * (int64_t)((3 * (2 * ((1 << depth) - 1) + 1) + 3) & ~3);
  }
  return (int64_t)n_heads * (int64_t)rpe_form_floats(depth, pos_bnd, f16_operand ? 1 : 0);
}

int hfl_window_rpe_expand(float* out, const float* rpe_table, int n_heads, int pos_bnd, int depth, int f16_operand,
                          hfl_stream_t stream) {
  const int64_t n = hfl_window_rpe_expand_size(n_heads, pos_bnd, depth, f16_operand);
  if (n <= 0 || out == nullptr || rpe_table == nullptr) return HFL_EINVAL;
  if (f16_operand != 2 && rpe_form(depth, pos_bnd, f16_operand ? 1 : 0) == 1)
    rpe_expand_kernel<<<(unsigned)hfl_cdiv(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        out, rpe_table, n_heads, pos_bnd, (1 << depth) - 1);
  else
    rpe_expand3_kernel<<<(unsigned)hfl_cdiv(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
        out, rpe_table, n_heads, pos_bnd, (1 << depth) - 1);
  HFL_RETURN_LAST_ERROR();
}

int hfl_relay_attention_fwd(float* out, const float* qkv, const int32_t* seq_rows,
                            const int32_t* seq_off, int batch, int n_heads, float scale,
                            int max_seq_len, hfl_stream_t stream) {
  if (batch <= 0 || n_heads <= 0 || max_seq_len < 0) return HFL_EINVAL;
  // one wave per (head, 16-query tile) of the longest sequence; shorter clouds leave waves idle
  const int items = n_heads * ((max_seq_len + 15) / 16);
  dim3 grid((unsigned)batch, (unsigned)(items > 4 ? (items + 3) / 4 : 1));
  relay_attn_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(out, qkv, seq_rows, seq_off,
                                                                        n_heads, scale);
  HFL_RETURN_LAST_ERROR();
}

int hfl_relay_attention_f16_fwd(void* out_split2, const void* qkv_f16, const int32_t* seq_rows, const int32_t* seq_off, int batch,
                                int n_heads, int max_seq_len, const int32_t* orphan_rows, int n_orphans, hfl_stream_t stream) {
  if (out_split2 == nullptr || qkv_f16 == nullptr || seq_rows == nullptr || seq_off == nullptr || batch <= 0 || n_heads <= 0 ||
      max_seq_len < 0 || n_orphans < 0 || (n_orphans > 0 && orphan_rows == nullptr))
    return HFL_EINVAL;
  const int items = n_heads * ((max_seq_len + 15) / 16);
  dim3 grid((unsigned)batch + (n_orphans > 0 ? 1u : 0u), (unsigned)(items > 4 ? (items + 3) / 4 : 1));
  relay_attn_f16_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<unsigned char*>(out_split2), static_cast<const unsigned char*>(qkv_f16), seq_rows, seq_off, orphan_rows,
      n_orphans, batch, n_heads, g_relay_fast);
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"

// ======================================================================================
// Backward of the windowed attention (training path, SURVEY section 7 "backward obligations").
// Given dO it recomputes the softmax per (window, head) -- everything of a window lives in one
// workgroup -- and produces dQ, dK, dV for every token / relay row plus the RPE-table gradient.
//   P = softmax(S), S = scale q k^T + bias ;  dP = dO V^T ;  D_i = sum_j P_ij dP_ij
//   dS = P o (dP - D) ;  dQ = scale dS K ;  dK = scale dS^T Q ;  dV = P^T dO ;  dTable[idx] += dS
// Two register orientations of the same 16x16 score tile are used so that every contraction is
// an MFMA whose operands already sit in lanes: "A" (key rows, query columns: a lane owns a query;
// softmax statistics, dQ, table gradient) and "B" (query rows, key columns: dK, dV).
// Q, K, V, dO of the wave's head are staged once in LDS; the table gradient is accumulated in LDS
// by the persistent workgroup and flushed with one global atomic pass.
namespace {

struct WinBwdParams {
  float* dqkv;            // (rows, 3*H*16)
  float* dtable;          // (3*nrpe, H), accumulated (zero it before the launch)
  const float* qkv;       // (rows, 3*H*16), bias included
  const float* dout;      // (rows, H*16)
  const uint32_t* meta;
  const float* table;
  int64_t n_tokens;
  int64_t rt_row0;
  int n_windows, K, D, H, bnd, batch;
  float scale;
  float* dtable_part;     // second-generation kernel: per (grid column, head) partial tables (gridDim.x, H, 3*nrpe) instead of
                          // float atomics into dtable (fixed-order reduction afterwards: reproducible); null = atomics
  int out_split;          // 1: dqkv leaves as the split2 operand of the GEMMs behind it (hfl_window_attention_bwd_split2)
  int depth;              // octree depth of the token rows (coordinates < 2^depth), 0 = unknown
};

// one lane's four consecutive channels of a dqkv row.  split != 0: the row is written in the split2 layout of csrc/gemm_x3.hip
// (per 32 channels 32 bf16 hi then 32 bf16 lo; a row keeps its 4 * width bytes) -- what hfl_split2 would make of the f32 row,
// bit for bit, without the f32 row ever being in memory
__device__ __forceinline__ void bwd_store4(float* row, int col, float a, float b, float c, float d, int split) {
  if (!split) {
    *reinterpret_cast<float4*>(row + col) = make_float4(a, b, c, d);
    return;
  }
  uint2 hi, lo;
  x3_split_pair_scalar(a, b, hi.x, lo.x);
  x3_split_pair_scalar(c, d, hi.y, lo.y);
  uint16_t* o = reinterpret_cast<uint16_t*>(row) + (col >> 5) * 64 + (col & 31);
  *reinterpret_cast<uint2*>(o) = hi;
  *reinterpret_cast<uint2*>(o + 32) = lo;
}

template <int T, int G>
__global__ void __launch_bounds__(128)
window_attn_bwd_kernel(const WinBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LP = T * 16;
  constexpr int TW = T - G;
  constexpr int NHW = 2;                          // heads (waves) per workgroup
  constexpr float kLog2e = 1.4426950408889634f;
  const int H = p.H, K = p.K;
  const int C = H * 16;
  const int nrpe = 2 * p.bnd + 1;
  // LDS carve (per workgroup)
  int4* s_key = reinterpret_cast<int4*>(smem);                          // [LP]
  int4* s_qry = s_key + LP;                                             // [LP]
  int* s_row = reinterpret_cast<int*>(s_qry + LP);                      // [LP]
  float* s_tile = reinterpret_cast<float*>(s_row + LP);                 // [NHW][4][LP][16]  Q,K,V,dO
  float* s_stat = s_tile + NHW * 4 * LP * 16;                           // [NHW][3][LP]      m, 1/l, D
  float* s_tab = s_stat + NHW * 3 * LP;                                 // [NHW][3*nrpe] * log2e
  float* s_dtab = s_tab + NHW * 3 * nrpe;                               // [NHW][3*nrpe]

  const int tid = threadIdx.x;
  const int lane = tid & 63, hw = tid >> 6;
  const int h = blockIdx.y * NHW + hw;
  const int c = lane & 15, g = lane >> 4;
  const bool rpe = p.table != nullptr;

  if (rpe)
    for (int i = tid; i < 3 * nrpe * NHW; i += blockDim.x) {
      const int r = i / NHW, hh = i % NHW;
      s_tab[hh * 3 * nrpe + r] = p.table[r * H + blockIdx.y * NHW + hh] * kLog2e;
      s_dtab[hh * 3 * nrpe + r] = 0.f;
    }
  float* tq = s_tile + (hw * 4 + 0) * LP * 16;
  float* tk = s_tile + (hw * 4 + 1) * LP * 16;
  float* tv = s_tile + (hw * 4 + 2) * LP * 16;
  float* td = s_tile + (hw * 4 + 3) * LP * 16;
  float* st_m = s_stat + (hw * 3 + 0) * LP;
  float* st_il = s_stat + (hw * 3 + 1) * LP;
  float* st_d = s_stat + (hw * 3 + 2) * LP;
  const float* tabx = s_tab + hw * 3 * nrpe;
  float* dtabx = s_dtab + hw * 3 * nrpe;
  const int hi4 = 8 * p.bnd;
  const float scale2 = p.scale * kLog2e;
  const float mask2 = kMaskValue * kLog2e;

  // bias of score (key position kj, query position qi) in the exp2 domain; returns table offsets
  auto bias_of = [&](const int4 k, const int4 q, bool use_rpe, int& ox, int& oy, int& oz) -> float {
    float b = 0.f;
    ox = oy = oz = -1;
    if (use_rpe) {
      ox = min(max(q.x + k.x, 0), hi4) >> 2;
      oy = (min(max(q.y + k.y, 0), hi4) >> 2) + nrpe;
      oz = (min(max(q.z + k.z, 0), hi4) >> 2) + 2 * nrpe;
      b = (tabx[ox] + tabx[oy]) + tabx[oz];
    }
    if (k.w != q.w) b += mask2;
    return b;
  };

  for (int w = blockIdx.x; w < p.n_windows; w += gridDim.x) {
    __syncthreads();
    for (int j = tid; j < LP; j += blockDim.x) {
      int bid = -1, row = -1, x = 0, y = 0, z = 0;
      if (j < K) {
        const int64_t t = (p.D == 1) ? (int64_t)w * K + j
                                     : ((int64_t)(w / p.D) * K + j) * p.D + (w % p.D);
        if (t < p.n_tokens) {
          const uint32_t xyz = p.meta[2 * t];
          x = (int)(xyz & 1023u); y = (int)((xyz >> 10) & 1023u); z = (int)(xyz >> 20);
          bid = (int)p.meta[2 * t + 1];
          row = (int)t;
        }
      } else if (G > 0 && j == K) {
        const int64_t t0 = (int64_t)w * K;
        bid = t0 < p.n_tokens ? (int)p.meta[2 * t0 + 1] : p.batch;
        row = (int)(p.rt_row0 + w);
      }
      s_key[j] = make_int4(4 * (p.bnd - x), 4 * (p.bnd - y), 4 * (p.bnd - z), bid);
      s_qry[j] = make_int4(4 * x, 4 * y, 4 * z, row >= 0 ? bid : -2);
      s_row[j] = row;
    }
    __syncthreads();
    // stage Q, K, V, dO of this head: LP rows x 4 float4 each, one wave
    for (int i = lane; i < LP * 4; i += 64) {
      const int j = i >> 2, f = i & 3;
      const int row = s_row[j];
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, v4 = q4, d4 = q4;
      if (row >= 0) {
        const float* base = p.qkv + (int64_t)row * 3 * C + h * 16 + 4 * f;
        q4 = *reinterpret_cast<const float4*>(base);
        k4 = *reinterpret_cast<const float4*>(base + C);
        v4 = *reinterpret_cast<const float4*>(base + 2 * C);
        d4 = *reinterpret_cast<const float4*>(p.dout + (int64_t)row * C + h * 16 + 4 * f);
      }
      reinterpret_cast<float4*>(tq)[i] = q4;
      reinterpret_cast<float4*>(tk)[i] = k4;
      reinterpret_cast<float4*>(tv)[i] = v4;
      reinterpret_cast<float4*>(td)[i] = d4;
    }
    __builtin_amdgcn_s_waitcnt(0);   // this wave's LDS writes are complete before it reads them
    __builtin_amdgcn_wave_barrier();

    // ---------------- pass 1, orientation A: statistics, dQ, table gradient ----------------
#pragma unroll 1
    for (int qt = 0; qt < T; ++qt) {
      const int qi = qt * 16 + c;
      const int4 q = s_qry[qi];
      const bool q_rpe = rpe && !(G > 0 && qt == T - 1);
      const float4 qf = reinterpret_cast<const float4*>(tq)[qi * 4 + g];
      const float4 df = reinterpret_cast<const float4*>(td)[qi * 4 + g];
      f32x4 s[T], dp[T];
      int off[T][4][3];
      float mx = kDeadValue;
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
        const float4 kf = reinterpret_cast<const float4*>(tk)[(kt * 16 + c) * 4 + g];
        const float4 vf4 = reinterpret_cast<const float4*>(tv)[(kt * 16 + c) * 4 + g];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf.w, acc, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4.x, df.x, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4.y, df.y, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4.z, df.z, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(vf4.w, df.w, acd, 0, 0, 0);
        dp[kt] = acd;
        const bool t_rpe = q_rpe && kt < TW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int4 k = s_key[kt * 16 + 4 * g + r];
          const float v = acc[r] * scale2 + bias_of(k, q, t_rpe, off[kt][r][0], off[kt][r][1], off[kt][r][2]);
          acc[r] = v;
          mx = fmaxf(mx, v);
        }
        s[kt] = acc;
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < T; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
          s[kt][r] = e;
          sum += e;
        }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      float dsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < T; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] *= inv;                        // P
          dsum += s[kt][r] * dp[kt][r];
        }
      dsum += __shfl_xor(dsum, 16, 64);
      dsum += __shfl_xor(dsum, 32, 64);
      if (g == 0) {
        st_m[qi] = mx;
        st_il[qi] = inv;
        st_d[qi] = dsum;
      }
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kt = 0; kt < T; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = s[kt][r] * (dp[kt][r] - dsum);
          if (off[kt][r][0] >= 0 && q.w >= 0) {
            atomicAdd(dtabx + off[kt][r][0], ds);
            atomicAdd(dtabx + off[kt][r][1], ds);
            atomicAdd(dtabx + off[kt][r][2], ds);
          }
          // dQ^T[d][query] += K[key][d] * dS[key][query]  (A = K in (k-slot g, d = c) layout)
          dq = __builtin_amdgcn_mfma_f32_16x16x4f32(tk[(kt * 16 + 4 * g + r) * 16 + c], ds, dq, 0, 0, 0);
        }
      const int qrow = s_row[qi];
      if (qrow >= 0)
        bwd_store4(p.dqkv + (int64_t)qrow * 3 * C, h * 16 + 4 * g, dq[0] * p.scale, dq[1] * p.scale, dq[2] * p.scale,
                   dq[3] * p.scale, p.out_split);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // ---------------- pass 2, orientation B: dK, dV -----------------------------------------
    f32x4 dk[T], dv[T];
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      dk[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int qt = 0; qt < T; ++qt) {
      const float4 qf = reinterpret_cast<const float4*>(tq)[(qt * 16 + c) * 4 + g];
      const float4 df = reinterpret_cast<const float4*>(td)[(qt * 16 + c) * 4 + g];
      const bool q_rpe = rpe && !(G > 0 && qt == T - 1);
      int4 qm[4];
      float m_i[4], il_i[4], d_i[4], qv[4], dov[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qt * 16 + 4 * g + r;
        qm[r] = s_qry[qi];
        m_i[r] = st_m[qi];
        il_i[r] = st_il[qi];
        d_i[r] = st_d[qi];
        qv[r] = tq[qi * 16 + c];        // Q[query 4g+r][d = c]
        dov[r] = td[qi * 16 + c];       // dO[query 4g+r][d = c]
      }
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
        const float4 kf = reinterpret_cast<const float4*>(tk)[(kt * 16 + c) * 4 + g];
        const float4 vf4 = reinterpret_cast<const float4*>(tv)[(kt * 16 + c) * 4 + g];
        const int4 k = s_key[kt * 16 + c];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.x, kf.x, acc, 0, 0, 0);   // (query 4g+r, key c)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.y, kf.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.z, kf.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.w, kf.w, acc, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(df.x, vf4.x, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(df.y, vf4.y, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(df.z, vf4.z, acd, 0, 0, 0);
        acd = __builtin_amdgcn_mfma_f32_16x16x4f32(df.w, vf4.w, acd, 0, 0, 0);
        const bool t_rpe = q_rpe && kt < TW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int o0, o1, o2;
          const float v = acc[r] * scale2 + bias_of(k, qm[r], t_rpe, o0, o1, o2);
          const float pb = __builtin_amdgcn_exp2f(v - m_i[r]) * il_i[r];
          const float ds = pb * (acd[r] - d_i[r]);
          // dV^T[d][key c] += dO[query][d] * P[query][key] ; dK^T[d][key c] += Q[query][d] * dS[query][key]
          dv[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dov[r], pb, dv[kt], 0, 0, 0);
          dk[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv[r], ds, dk[kt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      const int krow = s_row[kt * 16 + c];
      if (krow >= 0) {
        float* row = p.dqkv + (int64_t)krow * 3 * C;
        bwd_store4(row, C + h * 16 + 4 * g, dk[kt][0] * p.scale, dk[kt][1] * p.scale, dk[kt][2] * p.scale, dk[kt][3] * p.scale,
                   p.out_split);
        bwd_store4(row, 2 * C + h * 16 + 4 * g, dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3], p.out_split);
      }
    }
  }
  __syncthreads();
  if (rpe && p.dtable != nullptr)
    for (int i = tid; i < 3 * nrpe * NHW; i += blockDim.x) {
      const int r = i / NHW, hh = i % NHW;
      const float v = s_dtab[hh * 3 * nrpe + r];
      if (v != 0.f) atomicAdd(p.dtable + r * H + blockIdx.y * NHW + hh, v);
    }
}

// ---- backward, second generation (the default): one pass per (window, head) on the 16-cycle bf16 MFMAs -------------
// Same semantics and interface as window_attn_bwd_kernel above, which spends 28 fp32 MFMAs of 32 cycles per
// (query tile, key tile) pair because it recomputes the softmax in two orientations.  Here:
//  * Q, K, dO of the head are staged ONCE as bf16 (hi, lo) rows [16 x hi | 16 x lo] = 64 B (hi = RNE(v), lo = RNE(v - hi):
//    2^-18 relative); a row is then directly a K = 32 MFMA operand ("[a_hi | a_lo] . [b_hi | b_hi]" + "... [b_lo | b_lo]" =
//    all four cross terms of a 16-dim dot product in two v_mfma_f32_16x16x32_bf16), and the same image read through
//    ds_read_b64_tr_b16 gives the transposed operands (K^T, Q^T, dO^T) of the second-stage products.  V stays in registers.
//  * orientation "keys x queries" only (a lane owns a query column, as in the forward kernels): S^T and dP^T by MFMA,
//    softmax / D / dS in registers, dQ^T = K^T dS^T straight from those registers (three-term split of dS).
//  * dK and dV contract over the QUERIES: P and dS of a pair of query tiles are split to bf16 (hi, lo), written to a 4-KiB
//    per-wave LDS block per key tile and read back transposed (the transposing read again) as the B operand of
//    dV^T = dO^T P, dK^T = Q^T dS -- no second softmax pass.
//  8.5 MFMAs of 16 cycles per tile pair instead of 28 of 32; the table gradient still goes through LDS atomics.
typedef short bwd_b8 __attribute__((ext_vector_type(8)));
typedef short bwd_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t bwd_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
// four floats -> packed bf16 hi (2 dwords) and lo (2 dwords)
// (v_cvt_pk_bf16_f32: round to nearest even as bwd_bf16_rne, 6 instructions per pair instead of ~14)
__device__ __forceinline__ void bwd_split4(const float a, const float b, const float c, const float d, uint2& hi, uint2& lo) {
  x3_split_pair_scalar(a, b, hi.x, lo.x);
  x3_split_pair_scalar(c, d, hi.y, lo.y);
}
__device__ __forceinline__ bwd_b8 bwd_cat(const uint2 a, const uint2 b) {
  const uint4 q = make_uint4(a.x, a.y, b.x, b.y);
  return __builtin_bit_cast(bwd_b8, q);
}

// RT > 0 (round 6): the RPE-table gradient on the matrix cores.  The token coordinates of a level of depth <= 5 are < R = 16 RT, so
//     dtable[axis][clamp(w - v + bnd)] = sum_{q, k} dS[q][k] [x_q = w] [x_k = v]  =  diagonal sums of  F = OHQ^T dS OHK
// with the one-hot matrices OHQ (L x R), OHK (L x R) of the window's query / key coordinates: E = dS OHK (the dS fragments that
// feed dQ, against a one-hot B operand built from the packed coordinates), F += OHQ^T E (E's accumulators of a query-tile pair,
// split to bf16 (hi, lo), ARE the B operand: no transposition).  F (3 axes x R x R, 12 or 48 accumulator registers) is summed over
// every window the wave visits; the 2R - 1 diagonals are added once, at the end, in a fixed order.  No LDS atomics: the
// fixed-point scatter-add below ran at the rate of its equal-address conflicts (64 lanes onto <= 31 offsets per axis at depth 4)
// and was 56-61 % of the launch (profiles/r06_o_attn_bwd_bench_before.log).  RT = 0: that path (deeper levels).
template <int T, int G, int NREP, int RT>
__global__ void __launch_bounds__(128)
window_attn_bwd2_kernel(const WinBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LP = T * 16;
  constexpr int TW = T - G;
  constexpr int NP = (T + 1) / 2;                 // tile pairs (a K = 32 contraction covers two 16-row tiles)
  constexpr int LR = NP * 32;                     // image rows (the pad rows of an odd T stay zero)
  constexpr int NHW = 2;                          // heads (waves) per workgroup
  constexpr float kLog2e = 1.4426950408889634f;
  typedef __attribute__((address_space(3))) bwd_s4 lds_s4;
  const int H = p.H, K = p.K;
  const int C = H * 16;
  const int nrpe = 2 * p.bnd + 1;
  const int tabf = (3 * nrpe + 3) & ~3;
  int4* s_key = reinterpret_cast<int4*>(smem);                          // [LP]
  int4* s_qry = s_key + LP;                                             // [LP]
  int* s_row = reinterpret_cast<int*>(s_qry + LP);                      // [LP]
  // one-hot image of the window's coordinates (RT > 0): [axis][LR token rows][R = 16 RT columns] bf16, 1.0 at the token's
  // coordinate, all-zero rows for slots without RPE (padding, the relay token); both waves of the workgroup read it
  constexpr int OHB = 32 * RT;                                           // bytes per image row
  unsigned char* s_oh = reinterpret_cast<unsigned char*>(s_row + LP);    // [3][LR][OHB]
  unsigned char* wave_base = s_oh + 3 * LR * OHB;
  const int scr_bytes = NREP * tabf * 8 > 4096 ? NREP * tabf * 8 : 4096;   // transposition block / fixed-point table: never live together
  const int wave_bytes = 3 * LR * 64 + scr_bytes + 2 * tabf * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63, hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.y * NHW + hw;
  const int c = lane & 15, g = lane >> 4;
  const bool rpe = p.table != nullptr;
  unsigned char* img_k = wave_base + hw * wave_bytes;
  unsigned char* img_q = img_k + LR * 64;
  unsigned char* img_d = img_q + LR * 64;
  unsigned char* scr = img_d + LR * 64;                                  // [Ph | Pl | dSh | dSl] x 32 rows x 32 B
  float* tabx = reinterpret_cast<float*>(scr + scr_bytes);              // [3 nrpe] * log2e
  float* dtabx = tabx + tabf;                                            // the wave's table gradient (f32, lane-owned entries)
  long long* itab = reinterpret_cast<long long*>(scr);                  // [NREP][tabf] fixed-point staging of one query tile (all zero between tiles)

  // Table gradient: ds_add_f32 costs ~190 cycles per wave-instruction on gfx950 whatever the addresses, ds_add_u64 ~5 + 2 per
  // conflicting lane (tools/micro/lds_atomic_rate.hip), so the scatter-add runs in FIXED POINT: per query tile the wave's
  // largest |dS| sets a power-of-two scale (values become 31-bit integers), the 3 x L^2/T adds go to an int64 table
  // (integer adds commute: the result does not depend on the order), and after the tile every lane converts its own entries
  // back and adds them to the f32 table.  NREP copies (lane (c, g) uses copy c % NREP) thin out equal-address conflicts.
  long long* itab_mine = itab + (c % NREP) * tabf;
  const int sh0 = 10 * (g % 3), sh1 = 10 * ((g + 1) % 3), sh2 = 10 * ((g + 2) % 3);
  if (rpe) {
    for (int i = lane; i < 3 * nrpe; i += 64) {
      tabx[i] = p.table[i * H + h] * kLog2e;
      dtabx[i] = 0.f;
    }
  }
  if (LR > LP)                                                           // pad rows of the three images: zeros, once
    for (int i = lane; i < (LR - LP) * 4 * 3; i += 64) {
      const int a = i / ((LR - LP) * 4), r = i % ((LR - LP) * 4);
      reinterpret_cast<uint4*>(img_k + a * LR * 64 + LP * 64)[r] = make_uint4(0u, 0u, 0u, 0u);
    }
  if (RT > 0 && LR > LP)
    for (int i = tid; i < 3 * (LR - LP) * OHB / 16; i += NHW * 64) {
      const int a = i / ((LR - LP) * OHB / 16), r = i % ((LR - LP) * OHB / 16);
      reinterpret_cast<uint4*>(s_oh + (a * LR + LP) * OHB)[r] = make_uint4(0u, 0u, 0u, 0u);
    }
  const int hi4 = 8 * p.bnd;
  const float scale2 = p.scale * kLog2e;
  const float mask2 = kMaskValue * kLog2e;
  // transposing reads: lane (c, g) addresses row 4g + (c >> 2), dims 4 (c & 3) .. +3 of a 64-B image row
  const int tr_img = (4 * g + (c >> 2)) * 64 + (c & 3) * 8;
  const int tr_scr = (4 * g + (c >> 2)) * 32 + (c & 3) * 8;

  // bias of (key, query) in the exp2 domain; `packed` = the three table offsets, 10 bits each (3 nrpe <= 1023 is checked
  // by the launcher), -1 without RPE
  auto bias_of = [&](const int4 k, const int4 q, bool use_rpe, int& packed) -> float {
    float b = 0.f;
    packed = -1;
    if (use_rpe) {
      const int ox = min(max(q.x + k.x, 0), hi4) >> 2;
      const int oy = (min(max(q.y + k.y, 0), hi4) >> 2) + nrpe;
      const int oz = (min(max(q.z + k.z, 0), hi4) >> 2) + 2 * nrpe;
      b = (tabx[ox] + tabx[oy]) + tabx[oz];
      packed = ox | (oy << 10) | (oz << 20);
    }
    if (k.w != q.w) b += mask2;
    return b;
  };
  auto tr_pair = [&](const unsigned char* base, int stride16) -> bwd_b8 {      // rows r..r+3 of two 16-row tiles
    const bwd_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base));
    const bwd_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(base + stride16));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  constexpr int RTA = RT > 0 ? RT : 1;
  f32x4 ftab[3][RTA][RTA];                         // F[axis][w tile][v tile]: lane (c, g) holds rows w = 4g + r, column v = c
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < RTA; ++i)
#pragma unroll
      for (int j = 0; j < RTA; ++j) ftab[a][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int w = blockIdx.x; w < p.n_windows; w += gridDim.x) {
    // Rows are index arithmetic (token of window slot j = tok0 + j * D, relay row = rt_row0 + w), so NOTHING below waits for
    // the metadata round trip: the metadata words, the Q / K / dO rows and the V fragments are requested together.
    auto row_of = [&](int j) -> int {
      if (j < K) {
        const int64_t t = (p.D == 1) ? (int64_t)w * K + j : ((int64_t)(w / p.D) * K + j) * p.D + (w % p.D);
        return t < p.n_tokens ? (int)t : -1;
      }
      return (G > 0 && j == K) ? (int)(p.rt_row0 + w) : -1;
    };
    static_assert(LP <= 128, "one metadata slot per thread");
    uint32_t m_xyz = 0u;
    int m_bid = -1, m_row = -1;
    if (tid < LP) {
      m_row = row_of(tid);
      if (tid < K) {
        if (m_row >= 0) {
          m_xyz = p.meta[2 * (int64_t)m_row];
          m_bid = (int)p.meta[2 * (int64_t)m_row + 1];
        }
      } else if (G > 0 && tid == K) {
        const int64_t t0 = (int64_t)w * K;
        m_bid = t0 < p.n_tokens ? (int)p.meta[2 * t0 + 1] : p.batch;
      }
    }
    // ---- stage Q, K, dO of this head as bf16 (hi, lo) rows (this wave's own images: no barrier involved) -------------
    for (int i = lane; i < LP * 4; i += 64) {
      const int j = i >> 2, f = i & 3;
      const int row = row_of(j);
      float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4, d4 = q4;
      if (row >= 0) {
        const float* base = p.qkv + (int64_t)row * 3 * C + h * 16 + 4 * f;
        q4 = *reinterpret_cast<const float4*>(base);
        k4 = *reinterpret_cast<const float4*>(base + C);
        d4 = *reinterpret_cast<const float4*>(p.dout + (int64_t)row * C + h * 16 + 4 * f);
      }
      uint2 hi, lo;
      bwd_split4(q4.x, q4.y, q4.z, q4.w, hi, lo);
      *reinterpret_cast<uint2*>(img_q + j * 64 + f * 8) = hi;
      *reinterpret_cast<uint2*>(img_q + j * 64 + 32 + f * 8) = lo;
      bwd_split4(k4.x, k4.y, k4.z, k4.w, hi, lo);
      *reinterpret_cast<uint2*>(img_k + j * 64 + f * 8) = hi;
      *reinterpret_cast<uint2*>(img_k + j * 64 + 32 + f * 8) = lo;
      bwd_split4(d4.x, d4.y, d4.z, d4.w, hi, lo);
      *reinterpret_cast<uint2*>(img_d + j * 64 + f * 8) = hi;
      *reinterpret_cast<uint2*>(img_d + j * 64 + 32 + f * 8) = lo;
    }
    // V rows as [v_hi | v_lo] A operands, in registers: lane (c, g) holds dims 8 (g & 1) .. +7, hi for g < 2, lo else
    bwd_b8 av[T];
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      const int row = row_of(kt * 16 + c);
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (row >= 0) {
        const float* base = p.qkv + (int64_t)row * 3 * C + 2 * C + h * 16 + 8 * (g & 1);
        a = *reinterpret_cast<const float4*>(base);
        b = *reinterpret_cast<const float4*>(base + 4);
      }
      uint2 h0, l0, h1, l1;
      bwd_split4(a.x, a.y, a.z, a.w, h0, l0);
      bwd_split4(b.x, b.y, b.z, b.w, h1, l1);
      av[kt] = (g < 2) ? bwd_cat(h0, h1) : bwd_cat(l0, l1);
    }
    __syncthreads();                 // every wave is done with the previous window's metadata
    if (tid < LP) {
      const int x = (int)(m_xyz & 1023u), y = (int)((m_xyz >> 10) & 1023u), z = (int)(m_xyz >> 20);
      s_key[tid] = make_int4(4 * (p.bnd - x), 4 * (p.bnd - y), 4 * (p.bnd - z), m_bid);
      s_qry[tid] = make_int4(4 * x, 4 * y, 4 * z, m_row >= 0 ? m_bid : -2);
      s_row[tid] = m_row;
      if constexpr (RT > 0) {
        if (rpe) {
          const bool live = tid < K && m_row >= 0;
          const int xyz[3] = {x, y, z};
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            unsigned char* row = s_oh + (a * LR + tid) * OHB;
#pragma unroll
            for (int i = 0; i < OHB / 16; ++i) reinterpret_cast<uint4*>(row)[i] = make_uint4(0u, 0u, 0u, 0u);
            if (live && xyz[a] < 16 * RT) reinterpret_cast<uint16_t*>(row)[xyz[a]] = 0x3F80;   // (same thread, LDS in order)
          }
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);   // this wave's LDS writes are complete before it reads them
    __builtin_amdgcn_wave_barrier();

    f32x4 dk[T], dv[T];
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      dk[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // one-hot operands of the window (RT > 0), straight from the image by transposing reads: lane (c, g) gets column c of a
    // 16-column block for the token slots 4g .. 4g + 3 of the two 16-row tiles of pair tp -- the contraction order of the dS / E
    // fragments.  Queries and keys are the same tokens: oh[tp] is OHK of key pair tp and OHQ^T of query pair tp.
    bwd_b8 oh[NP][3][RTA];
    if constexpr (RT > 0) {
      if (rpe) {
        const int tr_oh = (4 * g + (c >> 2)) * OHB + (c & 3) * 8;
#pragma unroll
        for (int tp = 0; tp < NP; ++tp)
#pragma unroll
          for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int t = 0; t < RT; ++t) oh[tp][a][t] = tr_pair(s_oh + (a * LR + 2 * tp * 16) * OHB + t * 32 + tr_oh, 16 * OHB);
      }
    }

#pragma unroll
    for (int qp = 0; qp < NP; ++qp) {
      uint2 ph[2][T], pl[2][T], sh[2][T], sl[2][T];           // P and dS of the pair's two query tiles, bf16 (hi, lo)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * qp + u;
        if (qt >= T) {
#pragma unroll
          for (int kt = 0; kt < T; ++kt) ph[u][kt] = pl[u][kt] = sh[u][kt] = sl[u][kt] = make_uint2(0u, 0u);
          continue;
        }
        const int qi = qt * 16 + c;
        const int4 q = s_qry[qi];
        const bool q_rpe = rpe && !(G > 0 && qt == T - 1);
        const bwd_b8 bq_h = *reinterpret_cast<const bwd_b8*>(img_q + qi * 64 + (g & 1) * 16);        // [q_hi | q_hi]
        const bwd_b8 bq_l = *reinterpret_cast<const bwd_b8*>(img_q + qi * 64 + 32 + (g & 1) * 16);   // [q_lo | q_lo]
        const bwd_b8 bd_h = *reinterpret_cast<const bwd_b8*>(img_d + qi * 64 + (g & 1) * 16);
        const bwd_b8 bd_l = *reinterpret_cast<const bwd_b8*>(img_d + qi * 64 + 32 + (g & 1) * 16);
        f32x4 s[T], dp[T];
        int off[T][4];
        float mx = kDeadValue;
#pragma unroll
        for (int kt = 0; kt < T; ++kt) {
          const bwd_b8 ak = *reinterpret_cast<const bwd_b8*>(img_k + (kt * 16 + c) * 64 + g * 16);   // [k_hi | k_lo]
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acd = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bq_l, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ak, bq_h, acc, 0, 0, 0);
          acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[kt], bd_l, acd, 0, 0, 0);
          acd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[kt], bd_h, acd, 0, 0, 0);
          dp[kt] = acd;
          const bool t_rpe = q_rpe && kt < TW;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int4 k = s_key[kt * 16 + 4 * g + r];
            const float v = acc[r] * scale2 + bias_of(k, q, t_rpe, off[kt][r]);
            acc[r] = v;
            mx = fmaxf(mx, v);
          }
          s[kt] = acc;
        }
        mx = att_rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
            s[kt][r] = e;
            sum += e;
          }
        sum = att_rows_sum(sum);
        const float inv = 1.0f / sum;
        float dsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[kt][r] *= inv;                        // P
            dsum += s[kt][r] * dp[kt][r];
          }
        dsum = att_rows_sum(dsum);
        f32x4 dsv[T];
        float dmax = 0.f;
#pragma unroll
        for (int kt = 0; kt < T; ++kt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dsv[kt][r] = s[kt][r] * (dp[kt][r] - dsum);
            dmax = fmaxf(dmax, fabsf(dsv[kt][r]));
          }
          bwd_split4(s[kt][0], s[kt][1], s[kt][2], s[kt][3], ph[u][kt], pl[u][kt]);
          bwd_split4(dsv[kt][0], dsv[kt][1], dsv[kt][2], dsv[kt][3], sh[u][kt], sl[u][kt]);
        }
        if (RT == 0 && q_rpe) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, o, 64));
          dmax = att_rows_max(dmax);
          int bexp = (int)((__float_as_uint(dmax) >> 23) & 0xffu);          // dmax in [2^(bexp-127), 2^(bexp-126))
          if (bexp >= 30) {                                                  // (smaller: the tile's gradient is zero in f32)
            if (bexp > 250) bexp = 250;
            const float sc = __uint_as_float((uint32_t)(283 - bexp) << 23);  // |ds| * sc < 2^30
            const float isc = __uint_as_float((uint32_t)(bexp - 29) << 23);
            const bool q_live = q.w >= 0;
            for (int i = lane; i < NREP * tabf; i += 64) itab[i] = 0;           // (the block doubles as the transposition buffer)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int kt = 0; kt < TW; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int fx32 = __float2int_rn(dsv[kt][r] * sc);
                if (off[kt][r] >= 0 && q_live && fx32 != 0) {
                  // the four lane groups walk the three axes in rotated order: one wave-instruction then spreads over
                  // the x, y and z parts of the table (an equal address costs 2 cycles per extra lane)
                  const unsigned long long fx = (unsigned long long)(long long)fx32;
                  atomicAdd(reinterpret_cast<unsigned long long*>(itab_mine + ((off[kt][r] >> sh0) & 1023)), fx);
                  atomicAdd(reinterpret_cast<unsigned long long*>(itab_mine + ((off[kt][r] >> sh1) & 1023)), fx);
                  atomicAdd(reinterpret_cast<unsigned long long*>(itab_mine + ((off[kt][r] >> sh2) & 1023)), fx);
                }
              }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < 3 * nrpe; i += 64) {
              long long acc = 0;
#pragma unroll
              for (int rep = 0; rep < NREP; ++rep) acc += itab[rep * tabf + i];
              dtabx[i] += (float)acc * isc;
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
        // dQ^T[d][query] = sum over keys K^T[d][key] dS^T[key][query]: K^T by transposing reads, dS^T from registers
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kp = 0; kp < NP; ++kp) {
          const bwd_b8 kt_h = tr_pair(img_k + (2 * kp) * 16 * 64 + tr_img, 16 * 64);
          const bwd_b8 kt_l = tr_pair(img_k + (2 * kp) * 16 * 64 + 32 + tr_img, 16 * 64);
          const uint2 z2 = make_uint2(0u, 0u);
          const bwd_b8 bs_h = bwd_cat(sh[u][2 * kp], (2 * kp + 1 < T) ? sh[u][(2 * kp + 1 < T) ? 2 * kp + 1 : 0] : z2);
          const bwd_b8 bs_l = bwd_cat(sl[u][2 * kp], (2 * kp + 1 < T) ? sl[u][(2 * kp + 1 < T) ? 2 * kp + 1 : 0] : z2);
          dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_h, bs_l, dq, 0, 0, 0);
          dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_l, bs_h, dq, 0, 0, 0);
          dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_h, bs_h, dq, 0, 0, 0);
        }
        const int qrow = s_row[qi];
        if (qrow >= 0)
          bwd_store4(p.dqkv + (int64_t)qrow * 3 * C, h * 16 + 4 * g, dq[0] * p.scale, dq[1] * p.scale, dq[2] * p.scale,
                     dq[3] * p.scale, p.out_split);
      }

      // ---- table gradient of the pair's 32 queries on the matrix cores (see the kernel's header) ----------------------
      if constexpr (RT > 0) {
        if (rpe) {
          const uint2 z2 = make_uint2(0u, 0u);
#pragma unroll
          for (int axis = 0; axis < 3; ++axis) {
#pragma unroll
            for (int vt = 0; vt < RT; ++vt) {
              f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = {0.f, 0.f, 0.f, 0.f};     // E[query 4g + r of tile u][v = 16 vt + c]
#pragma unroll
              for (int kp = 0; kp < NP; ++kp) {
                const bwd_b8 ohk = oh[kp][axis][vt];
                const bool two = 2 * kp + 1 < T;
                const bwd_b8 a0h = bwd_cat(sh[0][2 * kp], two ? sh[0][two ? 2 * kp + 1 : 0] : z2);
                const bwd_b8 a0l = bwd_cat(sl[0][2 * kp], two ? sl[0][two ? 2 * kp + 1 : 0] : z2);
                const bwd_b8 a1h = bwd_cat(sh[1][2 * kp], two ? sh[1][two ? 2 * kp + 1 : 0] : z2);
                const bwd_b8 a1l = bwd_cat(sl[1][2 * kp], two ? sl[1][two ? 2 * kp + 1 : 0] : z2);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, ohk, e0, 0, 0, 0);
                e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, ohk, e0, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, ohk, e1, 0, 0, 0);
                e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, ohk, e1, 0, 0, 0);
              }
              uint2 h0, l0, h1, l1;
              bwd_split4(e0[0], e0[1], e0[2], e0[3], h0, l0);
              bwd_split4(e1[0], e1[1], e1[2], e1[3], h1, l1);
              const bwd_b8 eb_h = bwd_cat(h0, h1), eb_l = bwd_cat(l0, l1);
#pragma unroll
              for (int wt = 0; wt < RT; ++wt) {
                const bwd_b8 ohq = oh[qp][axis][wt];
                ftab[axis][wt][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ohq, eb_l, ftab[axis][wt][vt], 0, 0, 0);
                ftab[axis][wt][vt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ohq, eb_h, ftab[axis][wt][vt], 0, 0, 0);
              }
            }
          }
        }
      }

      // ---- dK^T, dV^T: contraction over the 32 queries of the pair ---------------------------------------------------
      const bwd_b8 qT_h = tr_pair(img_q + (2 * qp) * 16 * 64 + tr_img, 16 * 64);
      const bwd_b8 qT_l = tr_pair(img_q + (2 * qp) * 16 * 64 + 32 + tr_img, 16 * 64);
      const bwd_b8 dT_h = tr_pair(img_d + (2 * qp) * 16 * 64 + tr_img, 16 * 64);
      const bwd_b8 dT_l = tr_pair(img_d + (2 * qp) * 16 * 64 + 32 + tr_img, 16 * 64);
#pragma unroll
      for (int kt = 0; kt < T; ++kt) {
        // rows = query (u * 16 + c), 4 consecutive keys 4g..4g+3 of tile kt per lane: 8 B
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          unsigned char* dst = scr + (u * 16 + c) * 32 + g * 8;
          *reinterpret_cast<uint2*>(dst) = ph[u][kt];
          *reinterpret_cast<uint2*>(dst + 1024) = pl[u][kt];
          *reinterpret_cast<uint2*>(dst + 2048) = sh[u][kt];
          *reinterpret_cast<uint2*>(dst + 3072) = sl[u][kt];
        }
        __builtin_amdgcn_wave_barrier();                 // LDS operations of one wave execute in order: no wait needed
        const bwd_b8 bp_h = tr_pair(scr + tr_scr, 16 * 32);
        const bwd_b8 bp_l = tr_pair(scr + 1024 + tr_scr, 16 * 32);
        const bwd_b8 bs_h = tr_pair(scr + 2048 + tr_scr, 16 * 32);
        const bwd_b8 bs_l = tr_pair(scr + 3072 + tr_scr, 16 * 32);
        dv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT_h, bp_l, dv[kt], 0, 0, 0);
        dv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT_l, bp_h, dv[kt], 0, 0, 0);
        dv[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dT_h, bp_h, dv[kt], 0, 0, 0);
        dk[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT_h, bs_l, dk[kt], 0, 0, 0);
        dk[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT_l, bs_h, dk[kt], 0, 0, 0);
        dk[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qT_h, bs_h, dk[kt], 0, 0, 0);
        __builtin_amdgcn_wave_barrier();                 // the block is rewritten for the next key tile
      }
    }
#pragma unroll
    for (int kt = 0; kt < T; ++kt) {
      const int krow = s_row[kt * 16 + c];
      if (krow >= 0) {
        float* row = p.dqkv + (int64_t)krow * 3 * C;
        bwd_store4(row, C + h * 16 + 4 * g, dk[kt][0] * p.scale, dk[kt][1] * p.scale, dk[kt][2] * p.scale, dk[kt][3] * p.scale,
                   p.out_split);
        bwd_store4(row, 2 * C + h * 16 + 4 * g, dv[kt][0], dv[kt][1], dv[kt][2], dv[kt][3], p.out_split);
      }
    }
  }
  if constexpr (RT > 0) {
    if (rpe) {
      // F leaves the accumulators through this wave's (now idle) image block; entry t of an axis = the diagonals w - v that
      // clamp to it, added in ascending (diagonal, w) order by the lane that owns the entry
      constexpr int R = 16 * RT;
      static_assert(3 * R * R * 4 <= 3 * LR * 64, "F must fit the wave's image block");
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      float* fl = reinterpret_cast<float*>(img_k);
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int wt = 0; wt < RT; ++wt)
#pragma unroll
          for (int vt = 0; vt < RT; ++vt)
#pragma unroll
            for (int r = 0; r < 4; ++r) fl[(a * R + 16 * wt + 4 * g + r) * R + 16 * vt + c] = ftab[a][wt][vt][r];
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < 3 * nrpe; i += 64) {
        const int a = i / nrpe, t = i - a * nrpe;
        int dlo = t - p.bnd, dhi = t - p.bnd;
        if (t == 0) dlo = -(R - 1);
        if (t == nrpe - 1) dhi = R - 1;
        if (dlo < -(R - 1)) dlo = -(R - 1);
        if (dhi > R - 1) dhi = R - 1;
        float sum = 0.f;
        for (int d = dlo; d <= dhi; ++d) {
          const int w0 = d > 0 ? d : 0, w1 = d < 0 ? R - 1 + d : R - 1;
          for (int w = w0; w <= w1; ++w) sum += fl[(a * R + w) * R + (w - d)];
        }
        dtabx[i] = sum;
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (rpe && p.dtable_part != nullptr) {          // this wave's whole partial table, zeros included: summed in a fixed order
    float* part = p.dtable_part + ((int64_t)blockIdx.x * H + h) * (3 * nrpe);
    for (int i = lane; i < 3 * nrpe; i += 64) part[i] = dtabx[i];
  } else if (rpe && p.dtable != nullptr)
    for (int i = lane; i < 3 * nrpe; i += 64) {
      const float v = dtabx[i];
      if (v != 0.f) atomicAdd(p.dtable + i * H + h, v);
    }
}

// dtable (3 nrpe, H) = sum over the grid columns of the partial tables, in column order
__global__ void __launch_bounds__(256) window_dtable_reduce_kernel(float* __restrict__ dtable, const float* __restrict__ part,
                                                                   int n_part, int H, int n3) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n3 * H) return;
  const int i = idx / H, h = idx % H;
  float acc = 0.f;
  for (int b = 0; b < n_part; ++b) acc += part[((int64_t)b * H + h) * n3 + i];
  dtable[idx] = acc;
}

static int g_window_bwd_variant = 2;

// table gradient on the matrix cores when the level's coordinates fit 16 or 32 (desc.depth is the octree depth of the token rows;
// 0 = not given) and F fits the wave's image block; the fixed-point LDS scatter-add otherwise
template <int T>
static int window_bwd2_rt(int depth, bool has_table) {
  constexpr int LR = ((T + 1) / 2) * 32;
  int rt = 0;
  if (has_table && depth >= 1 && depth <= 4) rt = 1;
  else if (has_table && depth == 5 && 3 * 32 * 32 * 4 <= 3 * LR * 64) rt = 2;
  if (g_window_bwd_rt >= 0 && g_window_bwd_rt < rt) rt = g_window_bwd_rt;
  return rt;
}

template <int T, int NREP>
static size_t window_bwd2_lds(int bnd, int rt) {
  constexpr int LP = T * 16, NP = (T + 1) / 2, LR = NP * 32, NHW = 2;
  const int nrpe = 2 * bnd + 1;
  const int tabf = (3 * nrpe + 3) & ~3;
  const size_t scr_bytes = (size_t)NREP * tabf * 8 > 4096 ? (size_t)NREP * tabf * 8 : 4096;
  return (size_t)LP * 36 + (size_t)3 * LR * 32 * rt + (size_t)NHW * (3 * LR * 64 + scr_bytes + 2 * tabf * 4);
}

// Workgroups of one instantiation that a CU holds at once -- registers AND LDS (hipOccupancyMaxActiveBlocksPerMultiprocessor).
// Rounds 2-5 sized the grid from the LDS alone and added one column: the K = 64 + relay kernel holds 2 workgroups per CU by its
// registers where its LDS admits 3, so a third of the grid ran as a second round (and the "+ 1" column made a second round of a
// few workgroups even where the count was right).  Every workgroup loops over windows / columns of them: the grid must not
// exceed what is resident.
template <int T, int G, int NREP, int RT>
static int window_bwd2_resident(size_t lds) {
  static size_t cached_lds = 0;
  static int cached = 0;
  if (cached_lds == lds) return cached;
  const void* fn = reinterpret_cast<const void*>(window_attn_bwd2_kernel<T, G, NREP, RT>);
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 128, lds) != hipSuccess) nb = 0;
  cached_lds = lds;
  cached = nb;
  return nb;
}

// grid columns of the second-generation backward (the launch and the workspace size both come here)
template <int T, int G, int NREP, int RT>
static int window_bwd2_columns_rt(int n_windows, int H, int bnd) {
  constexpr int NHW = 2;
  if (H % NHW != 0) return 0;
  int nb = window_bwd2_resident<T, G, NREP, RT>(window_bwd2_lds<T, NREP>(bnd, RT));
  if (nb < 1) return 0;
  if (nb > 8) nb = 8;
  int capx = hfl_num_cus() * nb / (H / NHW);
  if (capx < 1) capx = 1;
  return n_windows > capx ? capx : n_windows;
}

template <int T, int G, int NREP>
static int window_bwd2_columns(int n_windows, int H, int bnd, int depth) {
  constexpr int LR = ((T + 1) / 2) * 32;
  const int rt = window_bwd2_rt<T>(depth, true);
  if constexpr (3 * 32 * 32 * 4 <= 3 * LR * 64) {
    if (rt == 2) return window_bwd2_columns_rt<T, G, NREP, 2>(n_windows, H, bnd);
  }
  if (rt >= 1) return window_bwd2_columns_rt<T, G, NREP, 1>(n_windows, H, bnd);
  return window_bwd2_columns_rt<T, G, NREP, 0>(n_windows, H, bnd);
}

template <int T, int G, int NREP, int RT>
static int launch_window_bwd2_rt(const WinBwdParams& p, hipStream_t s) {
  constexpr int NHW = 2;
  const int nrpe = 2 * p.bnd + 1;
  const size_t lds = window_bwd2_lds<T, NREP>(p.bnd, RT);
  if (p.H % NHW != 0) return HFL_EINVAL;
  const int bx = window_bwd2_columns_rt<T, G, NREP, RT>(p.n_windows, p.H, p.bnd);     // (sets the kernel's LDS attribute)
  if (bx < 1) return HFL_ECAPACITY;
  dim3 grid((unsigned)bx, (unsigned)(p.H / NHW));
  window_attn_bwd2_kernel<T, G, NREP, RT><<<grid, NHW * 64, lds, s>>>(p);
  if (p.dtable_part != nullptr && p.dtable != nullptr && p.table != nullptr) {
    const int n3 = 3 * nrpe;
    window_dtable_reduce_kernel<<<(n3 * p.H + 255) / 256, 256, 0, s>>>(p.dtable, p.dtable_part, bx, p.H, n3);
  }
  HFL_RETURN_LAST_ERROR();
}

template <int T, int G, int NREP>
static int launch_window_bwd2(const WinBwdParams& p, hipStream_t s) {
  constexpr int LR = ((T + 1) / 2) * 32;
  const int rt = window_bwd2_rt<T>(p.depth, p.table != nullptr);
  if constexpr (3 * 32 * 32 * 4 <= 3 * LR * 64) {
    if (rt == 2) return launch_window_bwd2_rt<T, G, NREP, 2>(p, s);
  }
  if (rt >= 1) return launch_window_bwd2_rt<T, G, NREP, 1>(p, s);
  return launch_window_bwd2_rt<T, G, NREP, 0>(p, s);
}

template <int T, int G>
static int launch_window_bwd(const WinBwdParams& p, hipStream_t s) {
  constexpr int LP = T * 16;
  constexpr int NHW = 2;
  const int nrpe = 2 * p.bnd + 1;
  const size_t lds = (size_t)LP * (16 + 16 + 4) + (size_t)NHW * 4 * LP * 16 * 4 + (size_t)NHW * 3 * LP * 4 +
                     (p.table ? (size_t)2 * NHW * 3 * nrpe * 4 : 0);
  if (p.H % NHW != 0) return HFL_EINVAL;
  if (g_window_bwd_variant >= 2 && 3 * (2 * p.bnd + 1) <= 1023) {
    // (replicated fixed-point tables, NREP = 4 / 16, measured slower: the LDS they take costs more occupancy than the
    // equal-address conflicts they remove -- tools/attn_bwd_bench.py history in DESIGN.md)
    const int rc = launch_window_bwd2<T, G, 1>(p, s);
    if (rc != HFL_ECAPACITY) return rc;              // tables that do not fit LDS: first-generation kernel below
  }
  int bx = p.n_windows;
  const int capx = hfl_num_cus() * 8 / (p.H / NHW) + 1;
  if (bx > capx) bx = capx;
  dim3 grid((unsigned)bx, (unsigned)(p.H / NHW));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel<T, G>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  window_attn_bwd_kernel<T, G><<<grid, NHW * 64, lds, s>>>(p);
  HFL_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" void hfl_internal_set_window_bwd(int v) { g_window_bwd_variant = v; }
extern "C" void hfl_internal_set_window_bwd_rt(int v) { g_window_bwd_rt = v; }

static int window_bwd_columns(const hfl_window_attn_desc* d) {
  if (d == nullptr || d->n_heads <= 0 || d->patch_size % 16 != 0 || d->n_relay < 0 || d->n_relay > 1) return 0;
  if (g_window_bwd_variant < 2 || 3 * (2 * d->pos_bnd + 1) > 1023) return 0;
  const int T = d->patch_size / 16 + d->n_relay;
  const int W = d->n_windows, H = d->n_heads, b = d->pos_bnd;
  if (d->n_relay == 0) {
    switch (T) {
      case 1: return window_bwd2_columns<1, 0, 1>(W, H, b, d->depth);
      case 2: return window_bwd2_columns<2, 0, 1>(W, H, b, d->depth);
      case 3: return window_bwd2_columns<3, 0, 1>(W, H, b, d->depth);
      case 4: return window_bwd2_columns<4, 0, 1>(W, H, b, d->depth);
      default: return 0;
    }
  }
  switch (T) {
    case 2: return window_bwd2_columns<2, 1, 1>(W, H, b, d->depth);
    case 3: return window_bwd2_columns<3, 1, 1>(W, H, b, d->depth);
    case 4: return window_bwd2_columns<4, 1, 1>(W, H, b, d->depth);
    case 5: return window_bwd2_columns<5, 1, 1>(W, H, b, d->depth);
    default: return 0;
  }
}

extern "C" int64_t hfl_window_attention_bwd_workspace(const hfl_window_attn_desc* d) {
  const int cols = window_bwd_columns(d);
  return cols <= 0 ? 0 : (int64_t)cols * d->n_heads * 3 * (2 * d->pos_bnd + 1) * (int64_t)sizeof(float);
}

static int window_attention_bwd_impl(float* dqkv, float* drpe_table, const float* qkv, const float* dout,
                                     const uint32_t* tok_meta, const float* rpe_table, const hfl_window_attn_desc* d,
                                     void* workspace, hfl_stream_t stream, int out_split = 0);

extern "C" int hfl_window_attention_bwd(float* dqkv, float* drpe_table, const float* qkv,
                                        const float* dout, const uint32_t* tok_meta,
                                        const float* rpe_table, const hfl_window_attn_desc* d,
                                        hfl_stream_t stream) {
  return window_attention_bwd_impl(dqkv, drpe_table, qkv, dout, tok_meta, rpe_table, d, nullptr, stream);
}

extern "C" int hfl_window_attention_bwd_det(float* dqkv, float* drpe_table, const float* qkv, const float* dout,
                                            const uint32_t* tok_meta, const float* rpe_table,
                                            const hfl_window_attn_desc* d, void* workspace, hfl_stream_t stream) {
  if (workspace == nullptr || hfl_window_attention_bwd_workspace(d) <= 0) return HFL_EINVAL;
  return window_attention_bwd_impl(dqkv, drpe_table, qkv, dout, tok_meta, rpe_table, d, workspace, stream);
}

/* dqkv written as the split2 operand (rows, 2 * 3C bf16) of the qkv layer's data- and weight-gradient GEMMs: the f32 gradient
 * and the hfl_split2 pass over it are gone.  workspace as hfl_window_attention_bwd_det, or NULL for float atomics. */
extern "C" int hfl_window_attention_bwd_split2(uint16_t* dqkv_split2, float* drpe_table, const float* qkv, const float* dout,
                                               const uint32_t* tok_meta, const float* rpe_table,
                                               const hfl_window_attn_desc* d, void* workspace, hfl_stream_t stream) {
  if (d != nullptr && (3 * d->n_heads * 16) % 32 != 0) return HFL_EINVAL;
  return window_attention_bwd_impl(reinterpret_cast<float*>(dqkv_split2), drpe_table, qkv, dout, tok_meta, rpe_table, d,
                                   workspace, stream, 1);
}

static int window_attention_bwd_impl(float* dqkv, float* drpe_table, const float* qkv, const float* dout,
                                     const uint32_t* tok_meta, const float* rpe_table, const hfl_window_attn_desc* d,
                                     void* workspace, hfl_stream_t stream, int out_split) {
  if (d == nullptr || d->n_windows < 0 || d->n_heads <= 0 || d->n_heads > 16) return HFL_EINVAL;
  if (d->patch_size % 16 != 0 || d->dilation < 1 || d->n_relay < 0 || d->n_relay > 1) return HFL_EINVAL;
  if (d->n_relay == 1 && d->dilation != 1) return HFL_EINVAL;
  if (d->n_windows == 0) return HFL_OK;
  WinBwdParams p;
  p.dqkv = dqkv; p.dtable = drpe_table; p.qkv = qkv; p.dout = dout; p.meta = tok_meta; p.table = rpe_table;
  p.n_tokens = d->n_tokens; p.rt_row0 = d->rt_row0; p.n_windows = d->n_windows;
  p.K = d->patch_size; p.D = d->dilation; p.H = d->n_heads; p.bnd = d->pos_bnd; p.batch = d->batch_size;
  p.scale = d->scale;
  p.dtable_part = static_cast<float*>(workspace);
  p.out_split = out_split;
  p.depth = d->depth;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int T = d->patch_size / 16 + d->n_relay;
  if (d->n_relay == 0) {
    switch (T) {
      case 1: return launch_window_bwd<1, 0>(p, s);
      case 2: return launch_window_bwd<2, 0>(p, s);
      case 3: return launch_window_bwd<3, 0>(p, s);
      case 4: return launch_window_bwd<4, 0>(p, s);
      default: return HFL_EINVAL;
    }
  }
  switch (T) {
    case 2: return launch_window_bwd<2, 1>(p, s);
    case 3: return launch_window_bwd<3, 1>(p, s);
    case 4: return launch_window_bwd<4, 1>(p, s);
    case 5: return launch_window_bwd<5, 1>(p, s);
    default: return HFL_EINVAL;
  }
}

// ======================================================================================
// Backward of the ragged relay-token attention (training path).  One workgroup per (cloud, head):
// the cloud's Q, K, V, dO rows of that head are staged in LDS (R x 16 floats each).  Two orientations,
// no atomics: (A) one thread per QUERY row recomputes its softmax statistics (max, 1/sum, D = sum_j p dp),
// accumulates dQ in registers and leaves the statistics in LDS; (B) one thread per KEY row walks the
// queries (LDS broadcasts), rebuilds p_ij from the statistics and accumulates dK, dV in registers.
// Work is tiny (R ~ 56..250).
namespace {

__global__ void __launch_bounds__(256)
relay_attn_bwd_kernel(float* __restrict__ dqkv, const float* __restrict__ qkv,
                      const float* __restrict__ dout, const int32_t* __restrict__ seq_rows,
                      const int32_t* __restrict__ seq_off, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x, h = blockIdx.y;
  const int r0 = seq_off[b];
  const int R = seq_off[b + 1] - r0;
  if (R <= 0) return;
  const int C = H * 16;
  float* sq = reinterpret_cast<float*>(smem);      // [R][16]
  float* sk = sq + R * 16;
  float* sv = sk + R * 16;
  float* sd = sv + R * 16;
  float* s_m = sd + R * 16;                         // [R] row max (scaled scores)
  float* s_inv = s_m + R;                           // [R] 1 / sum exp
  float* s_D = s_inv + R;                           // [R] sum_j p_ij (dO_i . v_j)
  for (int i = threadIdx.x; i < R * 4; i += blockDim.x) {
    const int j = i >> 2, f = i & 3;
    const int64_t row = seq_rows[r0 + j];
    const float* base = qkv + row * 3 * C + h * 16 + 4 * f;
    reinterpret_cast<float4*>(sq)[i] = *reinterpret_cast<const float4*>(base);
    reinterpret_cast<float4*>(sk)[i] = *reinterpret_cast<const float4*>(base + C);
    reinterpret_cast<float4*>(sv)[i] = *reinterpret_cast<const float4*>(base + 2 * C);
    reinterpret_cast<float4*>(sd)[i] = *reinterpret_cast<const float4*>(dout + row * C + h * 16 + 4 * f);
  }
  __syncthreads();
  // ---- (A) thread = query row ------------------------------------------------------------------
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    float q[16], g[16], dq[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) { q[d] = sq[i * 16 + d]; g[d] = sd[i * 16 + d]; dq[d] = 0.f; }
    float m = -INFINITY;
    for (int j = 0; j < R; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) s = fmaf(q[d], sk[j * 16 + d], s);
      m = fmaxf(m, s * scale);
    }
    float l = 0.f, D = 0.f;
    for (int j = 0; j < R; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) { s = fmaf(q[d], sk[j * 16 + d], s); dp = fmaf(g[d], sv[j * 16 + d], dp); }
      const float e = __expf(s * scale - m);
      l += e;
      D += e * dp;
    }
    const float inv = 1.0f / l;
    D *= inv;
    for (int j = 0; j < R; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) { s = fmaf(q[d], sk[j * 16 + d], s); dp = fmaf(g[d], sv[j * 16 + d], dp); }
      const float ds = __expf(s * scale - m) * inv * (dp - D) * scale;
#pragma unroll
      for (int d = 0; d < 16; ++d) dq[d] = fmaf(ds, sk[j * 16 + d], dq[d]);
    }
    s_m[i] = m; s_inv[i] = inv; s_D[i] = D;
    float* o = dqkv + (int64_t)seq_rows[r0 + i] * 3 * C + h * 16;
#pragma unroll
    for (int d = 0; d < 16; d += 4) *reinterpret_cast<float4*>(o + d) = make_float4(dq[d], dq[d + 1], dq[d + 2], dq[d + 3]);
  }
  __syncthreads();
  // ---- (B) thread = key row ---------------------------------------------------------------------
  for (int j = threadIdx.x; j < R; j += blockDim.x) {
    float k[16], v[16], dk[16], dv[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) { k[d] = sk[j * 16 + d]; v[d] = sv[j * 16 + d]; dk[d] = 0.f; dv[d] = 0.f; }
    for (int i = 0; i < R; ++i) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < 16; ++d) { s = fmaf(sq[i * 16 + d], k[d], s); dp = fmaf(sd[i * 16 + d], v[d], dp); }
      const float pj = __expf(s * scale - s_m[i]) * s_inv[i];
      const float ds = pj * (dp - s_D[i]) * scale;
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        dk[d] = fmaf(ds, sq[i * 16 + d], dk[d]);
        dv[d] = fmaf(pj, sd[i * 16 + d], dv[d]);
      }
    }
    float* base = dqkv + (int64_t)seq_rows[r0 + j] * 3 * C + h * 16;
#pragma unroll
    for (int d = 0; d < 16; d += 4) {
      *reinterpret_cast<float4*>(base + C + d) = make_float4(dk[d], dk[d + 1], dk[d + 2], dk[d + 3]);
      *reinterpret_cast<float4*>(base + 2 * C + d) = make_float4(dv[d], dv[d + 1], dv[d + 2], dv[d + 3]);
    }
  }
}

}  // namespace

extern "C" int hfl_relay_attention_bwd(float* dqkv, const float* qkv, const float* dout,
                                       const int32_t* seq_rows, const int32_t* seq_off, int batch,
                                       int n_heads, float scale, int max_seq_len, hfl_stream_t stream) {
  if (batch <= 0 || n_heads <= 0 || max_seq_len < 0) return HFL_EINVAL;
  const size_t lds = (size_t)max_seq_len * (16 * 4 * 4 + 3 * 4);
  if (lds > 160 * 1024) return HFL_ECAPACITY;
  if (max_seq_len == 0) return HFL_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(relay_attn_bwd_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  dim3 grid((unsigned)batch, (unsigned)n_heads);
  relay_attn_bwd_kernel<<<grid, 256, lds, static_cast<hipStream_t>(stream)>>>(dqkv, qkv, dout, seq_rows,
                                                                            seq_off, n_heads, scale);
  HFL_RETURN_LAST_ERROR();
}
