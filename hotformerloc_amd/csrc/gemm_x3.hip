// fp32-equivalent Linear on the bf16 matrix cores of gfx950, hand-written (no library):
//
//     y (M,N) = x (M,K) . W (N,K)^T  [+ bias]  [GELU]  [+ residual]
//
// Replaces torch.nn.Linear + the element-wise op that follows it in every transformer block of the reference
// (qkv / proj: models/octformer_backbone.py:70,91; fc1 -> GELU -> fc2: models/layers/octformer_layers.py:53-59;
// residual adds: models/octformer_backbone.py:275-278, models/hotformerloc_backbone.py:213-216).
//
// Arithmetic: every fp32 operand is split into bf16 (hi, lo), hi = RNE(v), lo = RNE(v - hi) (v = hi + lo to 2^-17) and
// the product is x_lo w_hi + x_hi w_lo + x_hi w_hi in that order with fp32 accumulation (v_mfma_f32_16x16x32_bf16):
// 4e-6 relative per GEMM against fp64.  Both operands arrive PRE-SPLIT in the "split2" layout
//
//     (rows, K/32, 2, 32) bf16  =  per 32-wide k-block: [32 x hi | 32 x lo]  = one 128-B line per (row, k-block)
//
// so a row's k-block is fetched once (4 B per element, not the 6 B of a K-concatenated [hi|hi|lo] operand) and used
// by three MFMAs.  Producers write that layout directly (LayerNorm, the attention kernel, this kernel's own GELU
// epilogue), weights are laid out once per parameter on the host.
//
// Kernel: 128 (rows) x 128 (features) tile per 256-lane workgroup, 2 x 2 waves of 64 x 64, K step 32.
//  * global -> LDS by `global_load_lds_dwordx4` (no registers, no VALU): one wave-instruction moves 8 rows x 128 B.
//    LDS image per operand tile: 128 rows x 8 slots of 16 B, slot (t ^ ((row >> 1) & 7)): ds_read_b128 of an MFMA
//    fragment (16 rows x one 16-B chunk per 16-lane group) then touches 16 distinct bank slots.  LDS-DMA writes are
//    lane-linear, so the permutation is applied on the SOURCE address (lane -> which 16-B chunk of the line it fetches)
//    and again on the read.
//  * ONE 32-KiB LDS stage, 3 workgroups per CU: a k-step first pulls its 16 fragments into registers, then (second
//    barrier) the DMA of step t+1 is issued and lands while the 48 MFMAs of step t run.  Co-resident workgroups sit in
//    different phases (load / MFMA / store), which is what keeps HBM reads, the matrix pipe and HBM writes busy together.
//  * the MFMA takes W as its A operand and x as its B operand, so the accumulator holds 4 CONSECUTIVE FEATURES of one
//    row per lane; the epilogue transposes through LDS so that every store instruction writes whole 128-B lines
//    (bias / GELU / residual / re-split fused).
//  * 1-D grid, XCD-aware: the N/128 feature tiles of one row tile get consecutive slots on ONE XCD (blocks b and b+8
//    share an XCD), so the row tile's activations are fetched from HBM once and re-read from that XCD's L2.
#include "hfl_common.h"
#include "x3_math.h"
#include "stage_stream.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int XT = 128;                  // tile edge (rows and features)

struct X3Params {
  void* out;                // EPI 0: float (M, N); EPI 1: bf16 split2 (M, N/32, 2, 32)
  const uint16_t* x;        // (M, K/32, 2, 32) bf16
  const uint16_t* w;        // (N, K/32, 2, 32) bf16
  const float* bias;        // (N) or null
  const float* residual;    // (M, N) or null (EPI 0 only; may alias out)
  HflRowSeg res_seg;        // residual rows in several arrays (res_seg.n > 1; residual = res_seg.ptr[0])
  const int32_t* tiles;     // grouped launch (EPI 0): per row tile {first row, rows (<= 128), first row of its W block}, or null
  const int32_t* gather;    // grouped launch: row m of the A operand is x[gather[m]] (the live pairs' input rows: the octree
                            // convolution's gather done by the tile loader), or null: x[m]
  const float* row_scale;   // EPI 0: (M) or null: out = (acc + bias) * row_scale[m] + residual (per-cloud stochastic depth)
  float* aux;               // EPI 3: f32 (M, N) pre-activation written next to the split2 output; EPI 4: the same, read
  int64_t M;
  int N, K;
  int tiles_n;
  int64_t n_wg;
  int qk_channels;          // EPI 2: C (= out_features / 3); features < C are queries
  float q_scale;            // EPI 2: factor folded into the queries (softmax scale * log2 e)
  int nt;                   // bit 0: non-temporal stores of the f32 output, bit 1: of the split2 output
  int dbg;                  // unused (kept for the probe tools)
};

static int g_x3_dbg = 0;
                            // Off: alone it halves the time of the relay tokens' K = 1024 GEMM, inside the step it loses 3 %
                            // (2361 -> 2442 clouds/s without it): its 128 KB of LDS need a nearly empty CU, the one-stage kernel's
                            // 32 KB slip in beside the finest level's workgroups
static int g_x3_nt = 0;     // measured: no end-to-end difference (the consumer kernel re-reads the output anyway)

// Epilogue shared by the kernels below.  acc[i][j]: features 16 i + 4 fq .. +3 (registers) of row 16 j + frow of the
// (16 MT) x 64 tile at (m_tile, n_tile) that one wavefront owns; ep = that wavefront's private 8-KiB LDS region.
template <int EPI, int MT>
__device__ __forceinline__ void x3_epilogue(const X3Params& p, f32x4 (&acc)[4][MT], unsigned char* ep, int64_t m_tile,
                                            int n_tile, int lane, int64_t m_end) {
  const int frow = lane & 15, fq = lane >> 4;
  // The accumulator holds features n..n+3 (registers) of row m = lane & 15: stored as is, a 128-B line would be
  // written in two halves by different instructions.  Each wave transposes its (16 MT) x 64 tile through a private 8 KiB
  // LDS region, 32 rows at a time (16-B chunks XOR-swizzled by the row: conflict-free both ways), and reads it back so
  // that 16 consecutive lanes hold 256 contiguous bytes of one output row: every store instruction writes whole lines.
  const int N = p.N;
  const int ecol = lane & 15;                                                 // 16-B chunk (4 features) inside the row
  const int nbase = n_tile + ecol * 4;
  const bool n_ok = EPI != 0 || nbase < N;          // EPI 0 takes N = 64: the upper half of the tile is W's zero padding
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr && n_ok) b = *reinterpret_cast<const float4*>(p.bias + nbase);
  // EPI 0: every residual value of the tile is requested here, before the first store.  `residual` may be `out` (the block's
  // in-place x += proj(...)), so the compiler keeps a load behind every earlier store: in the loop below that was 8 MT / 2
  // dependent round trips per wave (load, wait, add, store, next load ...) -- ~10 us per tile, the whole duration of the
  // step's many one-round launches.  An element is read and written by the same lane only, so the order does not matter.
  float4 rs[EPI == 0 ? MT / 2 : 1][EPI == 0 ? 8 : 1];
  if (EPI == 0) {
#pragma unroll
    for (int h = 0; h < MT / 2; ++h)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int64_t m = m_tile + h * 32 + it * 4 + fq;
        rs[h][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual != nullptr && m < m_end && n_ok)
          rs[h][it] = *reinterpret_cast<const float4*>((p.res_seg.n > 1 ? hfl_seg_row(p.res_seg, m, N) : p.residual + m * N) + nbase);
      }
  }
#pragma unroll
  for (int h = 0; h < MT / 2; ++h) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = jj * 16 + frow;
        *reinterpret_cast<f32x4*>(ep + r * 256 + (((i * 4 + fq) ^ frow) << 4)) = acc[i][2 * h + jj];
      }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + fq;                                              // row inside the 32
      const f32x4 a = *reinterpret_cast<const f32x4*>(ep + r * 256 + ((ecol ^ (r & 15)) << 4));
      const int64_t m = m_tile + h * 32 + r;
      if (m >= m_end || !n_ok) continue;
      float4 v = make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
      if (EPI == 0) {
        if (p.row_scale != nullptr) {
          const float rsc = p.row_scale[m];
          v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
        }
        if (p.residual != nullptr) {
          const float4 r4 = rs[EPI == 0 ? h : 0][EPI == 0 ? it : 0];
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        // non-temporal: the output is not re-read by this kernel, and letting it allocate in L2 evicts the operand
        // tiles the co-resident workgroups are re-reading (measured: 160 -> 106 us for the depth-4 fc1 shape)
        const f32x4 vv = {v.x, v.y, v.z, v.w};
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + m * N + nbase);
        if (p.nt & 1) __builtin_nontemporal_store(vv, dst);
        else *dst = vv;
      } else if (EPI == 2) {
        if (nbase < p.qk_channels) { v.x *= p.q_scale; v.y *= p.q_scale; v.z *= p.q_scale; v.w *= p.q_scale; }
        const auto h01 = __builtin_amdgcn_cvt_pkrtz(v.x, v.y), h23 = __builtin_amdgcn_cvt_pkrtz(v.z, v.w);
        const auto l01 = __builtin_amdgcn_cvt_pkrtz(v.x - (float)h01[0], v.y - (float)h01[1]);
        const auto l23 = __builtin_amdgcn_cvt_pkrtz(v.z - (float)h23[0], v.w - (float)h23[1]);
        const uint32_t uh0 = __builtin_bit_cast(uint32_t, h01), uh1 = __builtin_bit_cast(uint32_t, h23);
        const uint32_t ul0 = __builtin_bit_cast(uint32_t, l01), ul1 = __builtin_bit_cast(uint32_t, l23);
        // lane pair (2k, 2k+1) holds 8 consecutive dims of one head: the even lane stores their hi halves (16 B), the
        // odd lane their lo halves
        const uint32_t mine0 = (lane & 1) ? uh0 : ul0, mine1 = (lane & 1) ? uh1 : ul1;
        const uint32_t got0 = __shfl_xor(mine0, 1, 64), got1 = __shfl_xor(mine1, 1, 64);
        const u32x4 qq = (lane & 1) ? (u32x4){got0, got1, ul0, ul1} : (u32x4){uh0, uh1, got0, got1};
        const int n8 = n_tile + (ecol & ~1) * 4;                       // first of the pair's 8 features
        const int dim0 = n8 & 15;                                              // 0 or 8
        unsigned char* o = reinterpret_cast<unsigned char*>(p.out) + m * (int64_t)N * 4 + n8 * 4 - dim0 * 2 +
                           ((lane & 1) ? 32 : 0);
        *reinterpret_cast<u32x4*>(o) = qq;
      } else {
        f32x2 v01 = {v.x, v.y}, v23 = {v.z, v.w};
        uint32_t h01, l01, h23, l23;                                      // packed bf16 pairs
        if (EPI == 4) {
          const float4 y = *reinterpret_cast<const float4*>(p.aux + m * N + nbase);
          v01 *= x3_gelu_grad2((f32x2){y.x, y.y});
          v23 *= x3_gelu_grad2((f32x2){y.z, y.w});
        } else {
          if (EPI == 3) *reinterpret_cast<float4*>(p.aux + m * N + nbase) = v;
          v01 = x3_gelu2(v01);
          v23 = x3_gelu2(v23);
        }
        x3_split_pair(v01, h01, l01);
        x3_split_pair(v23, h23, l23);
        // lane pair (2k, 2k+1) holds 8 consecutive features: the even lane stores their 8 hi values (16 B), the odd
        // lane their 8 lo values, so one instruction writes both halves of every [32 x hi | 32 x lo] line
        const uint32_t mine0 = (lane & 1) ? h01 : l01;                    // what the partner needs from me
        const uint32_t mine1 = (lane & 1) ? h23 : l23;
        const uint32_t got0 = __shfl_xor(mine0, 1, 64), got1 = __shfl_xor(mine1, 1, 64);
        uint4 q;
        if (lane & 1) q = make_uint4(got0, got1, l01, l23);               // lo of (partner, me)
        else          q = make_uint4(h01, h23, got0, got1);               // hi of (me, partner)
        const int nfeat = n_tile + (ecol & ~1) * 4;                      // first of the pair's 8 features
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + m * (2 * (int64_t)N) + (nfeat >> 5) * 64 + (nfeat & 31) +
                      ((lane & 1) ? 32 : 0);
        const u32x4 qq = {q.x, q.y, q.z, q.w};
        if (p.nt & 2) __builtin_nontemporal_store(qq, reinterpret_cast<u32x4*>(o));
        else *reinterpret_cast<u32x4*>(o) = qq;
      }
    }
  }
}

// EPI 0: out f32 = acc + bias [+ residual];  EPI 1: out split2 = split(gelu(acc + bias));
// EPI 3: EPI 1 + the pre-activation acc + bias also written as f32 to aux (training forward of fc1: GELU's backward needs it)
// EPI 4: out split2 = split((acc + bias) * gelu'(aux)) (training backward: dx of fc2 straight into the operand of fc1's gradients)
// EPI 2: out = the window-attention operand layout of acc + bias (qkv projection): per row [Q | K | V] regions of C
//        features, per head 16 dims as [16 x hi | 16 x lo] fp16 (hi = RTZ(v), lo = RTZ(v - hi): 22 significant bits), the
//        queries pre-multiplied by q_scale -- csrc/attention.hip, window_attn_kernel_v5 loads these as MFMA fragments
// MT = 16-row tiles of x per wave: 4 -> 128-row workgroup tile (3 workgroups / CU), 8 -> 256-row tile (2 / CU): the
// weight tile is then shared by twice the rows, 25 % less L2 -> LDS operand traffic per MAC -- the k-loop runs at the
// ~70 GB/s per CU an XCD's L2 serves, not at the matrix rate
template <int EPI, int MT>
__global__ void __launch_bounds__(256, MT == 4 ? 3 : 2)
gemm_x3_kernel(const X3Params p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];     // one stage: x tile | w tile
  constexpr int BM = 32 * MT;                                                 // rows of x per workgroup
  constexpr int XTILE_B = BM * 128;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wm = wave & 1;

  // ---- XCD-aware tile assignment (bijective remap: consecutive new ids share an XCD) -------------------------
  int64_t wg = blockIdx.x;
  {
    const int64_t q = p.n_wg >> 3, r = p.n_wg & 7;
    const int64_t xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  int64_t m0 = (wg / p.tiles_n) * BM;
  const int n0 = (int)(wg % p.tiles_n) * XT;
  const int K = p.K;
  const int nk = K >> 5;
  const int64_t row_b = (int64_t)K * 4;                       // bytes per split2 row (2K bf16)
  int64_t m_end = p.M;
  int64_t w_row0 = n0;
  if (p.tiles != nullptr) {                                   // grouped launch: this row tile's rows and weight block
    const int32_t* tt = p.tiles + 3 * (wg / p.tiles_n);
    m0 = tt[0];
    m_end = m0 + tt[1];
    w_row0 = (int64_t)tt[2] + n0;
  }

  // ---- staging: wave w moves rows [8 MT w, 8 MT (w+1)) of the x tile and [32w, 32w+32) of the w tile, 8 rows per
  // instruction
  // lane -> (row within the 8, physical slot); the slot it fills holds logical chunk t = slot ^ ((row >> 1) & 7)
  const int srow = lane >> 3, sslot = lane & 7;
  // uniform tile bases (SGPRs) + 32-bit per-lane offsets: tail rows fetch the last valid row, never stored
  const bool gathered = p.gather != nullptr;
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(p.x) + (gathered ? 0 : m0 * row_b);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + w_row0 * row_b;
  const int rows_valid = (int)((m_end - m0) < BM ? (m_end - m0) : BM);
  uint32_t xoff[MT], woff[4];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wave * (8 * MT) + i * 8 + srow;
    const int t = sslot ^ ((row >> 1) & 7);
    const int xr = row < rows_valid ? row : rows_valid - 1;
    // gathered: the pair's input row (the launcher checked rows * row bytes < 2^32)
    xoff[i] = (gathered ? (uint32_t)p.gather[m0 + xr] : (uint32_t)xr) * (uint32_t)row_b + t * 16;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + srow;
    woff[i] = (uint32_t)row * (uint32_t)row_b + (sslot ^ ((row >> 1) & 7)) * 16;
  }
  auto stage = [&](int kt) {
    const unsigned char* xk = xbase + (int64_t)kt * 128;
    const unsigned char* wk = wbase + (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xk + xoff[i]),
                                       (__attribute__((address_space(3))) void*)(smem + (wave * (8 * MT) + i * 8) * 128),
                                       16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wk + woff[i]),
                                       (__attribute__((address_space(3))) void*)(smem + XTILE_B + (wave * 32 + i * 8) * 128),
                                       16, 0, 0);
  };

  // ---- fragment addresses (bytes inside the stage): row = base + 16 i + (lane & 15), chunk q = lane >> 4 --------
  const int frow = lane & 15, fq = lane >> 4;
  // (16 i rows further on the swizzle term (row >> 1) & 7 is the same: one base per operand, immediates for i)
  const int rn0 = wn * 64 + frow, rm0 = wm * (16 * MT) + frow;
  const int offw_hi = XTILE_B + rn0 * 128 + ((fq ^ ((rn0 >> 1) & 7)) << 4), offw_lo = offw_hi ^ 64;
  const int offx_hi = rm0 * 128 + ((fq ^ ((rm0 >> 1) & 7)) << 4), offx_lo = offx_hi ^ 64;
  // the lo chunk is logical slot 4 + q: physical slot differs from the hi one in bit 2 only -> byte offset ^ 64

  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One LDS stage, two barriers per k-step: once every wave holds the step's 16 fragments in registers the stage is
  // free again, so the DMA of step t+1 is issued BEFORE the 48 MFMAs of step t and lands while they run.
  stage(0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                       // drains this wave's LDS-DMA (vmcnt) and publishes the stage
    bf16x8 wh[4], wl[4], xh[MT], xl[MT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wh[i] = *reinterpret_cast<const bf16x8*>(smem + offw_hi + i * 2048);
      wl[i] = *reinterpret_cast<const bf16x8*>(smem + offw_lo + i * 2048);
    }
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      xh[j] = *reinterpret_cast<const bf16x8*>(smem + offx_hi + j * 2048);
      xl[j] = *reinterpret_cast<const bf16x8*>(smem + offx_lo + j * 2048);
    }
    __syncthreads();                       // every wave has its fragments (lgkmcnt(0) precedes the barrier)
    if (kt + 1 < nk) stage(kt + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[i], xh[j], acc[i][j], 0, 0, 0);
      }
  }
  // ---- epilogue (the stage is free after the last barrier: every wave transposes through its own 8 KiB of it) ----------
  x3_epilogue<EPI, MT>(p, acc, smem + wave * 8192, m0 + wm * (16 * MT), n0 + wn * 64, lane, m_end);
}


// fp32 (rows, C) [* row_scale[row]] -> split2 (rows, C/32, 2, 32) bf16
__global__ void __launch_bounds__(256)
split2_kernel(uint16_t* __restrict__ out, const float* __restrict__ x, const float* __restrict__ row_scale, int64_t n_rows,
              int C) {
  const int cv = C / 4;
  const int64_t total = n_rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c = (int)(i % cv) * 4;
    float4 v = reinterpret_cast<const float4*>(x)[i];
    if (row_scale != nullptr) {
      const float rsc = row_scale[r];
      v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
    }
    const uint32_t h0 = x3_bf16_rne(v.x), h1 = x3_bf16_rne(v.y), h2 = x3_bf16_rne(v.z), h3 = x3_bf16_rne(v.w);
    const uint32_t l0 = x3_bf16_rne(v.x - __uint_as_float(h0 << 16));
    const uint32_t l1 = x3_bf16_rne(v.y - __uint_as_float(h1 << 16));
    const uint32_t l2 = x3_bf16_rne(v.z - __uint_as_float(h2 << 16));
    const uint32_t l3 = x3_bf16_rne(v.w - __uint_as_float(h3 << 16));
    uint16_t* o = out + r * (2 * (int64_t)C) + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<uint2*>(o) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
    *reinterpret_cast<uint2*>(o + 32) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
  }
}

}  // namespace

extern "C" {

void hfl_internal_set_x3_dbg(int v) {
  if (v >= 0x200) return;                   // (knobs of removed tile variants)
  if (v >= 0x100) g_x3_nt = v & 3;          // 0x100 | nt bits
  else g_x3_dbg = v;
}

static int x3_launch(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                     const float* residual, int64_t n_rows, int in_features, int out_features, int epi,
                     float q_scale, hfl_stream_t stream, float* aux = nullptr, const float* row_scale = nullptr,
                     const int32_t* tiles = nullptr, int64_t n_tiles = 0, const int32_t* gather = nullptr,
                     int64_t n_src_rows = 0, const HflRowSeg* res_seg = nullptr);

int hfl_linear_x3(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                  const float* residual, int64_t n_rows, int in_features, int out_features, int gelu_split_out,
                  hfl_stream_t stream) {
  return x3_launch(out, x_split2, w_split2, bias, residual, n_rows, in_features, out_features,
                   gelu_split_out ? 1 : 0, 1.0f, stream);
}

int hfl_linear_x3_seg(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                      const hfl_row_segments* residual, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream) {
  HflRowSeg seg;
  if (!hfl_seg_from(residual, n_rows, &seg)) return HFL_EINVAL;
  return x3_launch(out, x_split2, w_split2, bias, seg.ptr[0], n_rows, in_features, out_features, 0, 1.0f, stream, nullptr,
                   nullptr, nullptr, 0, nullptr, 0, &seg);
}

int hfl_linear_x3_rows(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                       const float* residual, const float* row_scale, int64_t n_rows, int in_features, int out_features,
                       hfl_stream_t stream) {
  return x3_launch(out, x_split2, w_split2, bias, residual, n_rows, in_features, out_features, 0, 1.0f, stream, nullptr,
                   row_scale);
}

int hfl_linear_x3_grouped(float* out, const uint16_t* x_split2, const uint16_t* w_split2, const int32_t* tiles,
                          int64_t n_tiles, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream) {
  if (tiles == nullptr || n_tiles < 0) return HFL_EINVAL;
  if (n_tiles == 0) return HFL_OK;
  return x3_launch(out, x_split2, w_split2, nullptr, nullptr, n_rows, in_features, out_features, 0, 1.0f, stream, nullptr,
                   nullptr, tiles, n_tiles);
}

/* hfl_linear_x3_grouped with the octree convolution's gather done by the tile loader: row m of the A operand is
 * x_split2[gather[m]] (x_split2 (n_src_rows, 2 K) bf16: the split2 form of the convolution's INPUT rows; gather (n_rows) int32
 * = the input row of every live (row, tap) pair).  The gathered (pairs x Cin) matrix never exists in memory. */
int hfl_linear_x3_grouped_gather(float* out, const uint16_t* x_split2, const int32_t* gather, int64_t n_src_rows,
                                 const uint16_t* w_split2, const int32_t* tiles, int64_t n_tiles, int64_t n_rows,
                                 int in_features, int out_features, hfl_stream_t stream) {
  if (tiles == nullptr || gather == nullptr || n_tiles < 0 || n_src_rows <= 0) return HFL_EINVAL;
  if (n_src_rows * (int64_t)in_features * 4 >= ((int64_t)1 << 32)) return HFL_ECAPACITY;       // 32-bit per-lane byte offsets
  if (n_tiles == 0) return HFL_OK;
  return x3_launch(out, x_split2, w_split2, nullptr, nullptr, n_rows, in_features, out_features, 0, 1.0f, stream, nullptr,
                   nullptr, tiles, n_tiles, gather, n_src_rows);
}

int hfl_linear_x3_qkv(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                      int64_t n_rows, int in_features, int out_features, float q_scale, hfl_stream_t stream) {
  if (out_features % 3 != 0 || (out_features / 3) % 128 != 0) return HFL_EINVAL;
  return x3_launch(out, x_split2, w_split2, bias, nullptr, n_rows, in_features, out_features, 2, q_scale, stream);
}

int hfl_linear_x3_gelu_fwd(uint16_t* out_split2, float* preact, const uint16_t* x_split2, const uint16_t* w_split2,
                           const float* bias, int64_t n_rows, int in_features, int out_features, hfl_stream_t stream) {
  if (preact == nullptr) return HFL_EINVAL;
  return x3_launch(out_split2, x_split2, w_split2, bias, nullptr, n_rows, in_features, out_features, 3, 1.0f, stream,
                   preact);
}

int hfl_linear_x3_gelu_bwd(uint16_t* out_split2, const uint16_t* dy_split2, const uint16_t* wt_split2,
                           const float* preact, int64_t n_rows, int in_features, int out_features,
                           hfl_stream_t stream) {
  if (preact == nullptr) return HFL_EINVAL;
  return x3_launch(out_split2, dy_split2, wt_split2, nullptr, nullptr, n_rows, in_features, out_features, 4, 1.0f,
                   stream, const_cast<float*>(preact));
}

static int x3_launch(void* out, const uint16_t* x_split2, const uint16_t* w_split2, const float* bias,
                     const float* residual, int64_t n_rows, int in_features, int out_features, int epi,
                     float q_scale, hfl_stream_t stream, float* aux, const float* row_scale, const int32_t* tiles,
                     int64_t n_tiles, const int32_t* gather, int64_t n_src_rows, const HflRowSeg* res_seg) {
  const int gelu_split_out = epi == 1 || epi == 3 || epi == 4;
  if (n_rows < 0 || in_features <= 0 || out_features <= 0) return HFL_EINVAL;
  const bool narrow = tiles != nullptr && epi == 0 && out_features == 64;       // W blocks padded to 128 rows by the caller
  if (in_features % 32 != 0 || (out_features % XT != 0 && !narrow)) return HFL_EINVAL;
  if (out == nullptr || x_split2 == nullptr || w_split2 == nullptr) return HFL_EINVAL;
  if (gelu_split_out && residual != nullptr) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  X3Params p;
  p.out = out; p.x = x_split2; p.w = w_split2; p.bias = bias; p.residual = residual; p.aux = aux;
  p.res_seg = res_seg != nullptr ? *res_seg : hfl_seg_single(residual);
  p.row_scale = row_scale;
  p.tiles = tiles;
  p.gather = gather;
  (void)n_src_rows;
  p.M = n_rows; p.N = out_features; p.K = in_features;
  p.tiles_n = narrow ? 1 : out_features / XT;
  // 128-row tiles, 3 workgroups per CU.  (A 256-row tile and a 128 x 256 eight-wave tile were built and measured 10 - 120 %
  // slower on every shape of the model: DESIGN.md section 4; they are gone from the library.)
  p.n_wg = (tiles != nullptr ? n_tiles : hfl_cdiv(n_rows, 128)) * p.tiles_n;
  p.dbg = g_x3_dbg;
  p.nt = g_x3_nt;
  p.qk_channels = out_features / 3;
  p.q_scale = q_scale;
  if (p.n_wg > 0x7fffffffLL) return HFL_ECAPACITY;
  const size_t lds = (size_t)(128 + XT) * 128 + (size_t)(g_x3_dbg & 0xFF) * 1024;   // (+ probe: extra KiB to cut occupancy)
  hipStream_t s = static_cast<hipStream_t>(stream);
#define HFL_X3_LAUNCH(E, M)                                                                              \
  {                                                                                                      \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3_kernel<E, M>),              \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    if (e != hipSuccess) return (int)e;                                                                  \
    gemm_x3_kernel<E, M><<<(unsigned)p.n_wg, 256, lds, s>>>(p);                                          \
  }
  if (epi == 4) HFL_X3_LAUNCH(4, 4) else if (epi == 3) HFL_X3_LAUNCH(3, 4) else if (epi == 2) HFL_X3_LAUNCH(2, 4)
  else if (epi == 1) HFL_X3_LAUNCH(1, 4) else HFL_X3_LAUNCH(0, 4)
#undef HFL_X3_LAUNCH
  HFL_RETURN_LAST_ERROR();
}

int hfl_split2_rows(uint16_t* out, const float* x, const float* row_scale, int64_t n_rows, int64_t channels,
                    hfl_stream_t stream) {
  if (n_rows < 0 || channels <= 0 || channels % 32 != 0) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  const int64_t need = hfl_cdiv(n_rows * (channels / 4), 256);
  const int64_t cap = (int64_t)hfl_stream_cus(static_cast<hipStream_t>(stream)) * 16;
  split2_kernel<<<(int)(need < cap ? need : cap), 256, 0, static_cast<hipStream_t>(stream)>>>(out, x, row_scale, n_rows,
                                                                                              (int)channels);
  HFL_RETURN_LAST_ERROR();
}

int hfl_split2(uint16_t* out, const float* x, int64_t n_rows, int64_t channels, hfl_stream_t stream) {
  return hfl_split2_rows(out, x, nullptr, n_rows, channels, stream);
}

}  // extern "C"
