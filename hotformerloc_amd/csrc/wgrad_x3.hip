// Weight gradient of a Linear layer on the bf16 matrix cores of gfx950, fp32-equivalent (three-term split):
//
//     dW (N,K) = dy (M,N)^T . x (M,K)          db (N) = column sums of dy
//
// Replaces what autograd does for torch.nn.Linear in the reference's training step (training/trainer.py:287-365 runs
// loss.backward() over models/octformer_backbone.py:70,91 and models/layers/octformer_layers.py:53-59): an fp32 GEMM
// contracting over the ~10^5..10^6 token rows into a small (N,K) matrix, plus a separate column reduction for the bias.
//
// Both operands arrive in the "split2" layout of csrc/gemm_x3.hip -- (rows, C/32, 2, 32) bf16 = per 32 channels
// [32 x hi | 32 x lo] -- which the backward pass has anyway (dy split for dx = dy W, x split by the forward).  The
// contraction index (the row m) is the SLOW index of both, i.e. both MFMA operands are needed transposed: tiles of
// 32 rows x 128 channels are DMA'd to LDS as they lie in memory (global_load_lds_dwordx4) and read back with
// ds_read_b64_tr_b16, which hands lane (c, g) the four rows 4g..4g+3 of column c -- two of them make the 8-deep k-slice
// of v_mfma_f32_16x16x32_bf16 (k order inside the 32-row step is permuted the same way for both operands, which a
// contraction does not see).
//
// Work split: the (N/128) x (K/128) output tiles times S row slabs (S chosen so that ~3 workgroups per CU exist);
// every workgroup writes its partial tile to a workspace (S, N, K) and a second kernel adds the slabs in a fixed order:
// bitwise reproducible, no atomics.  The bias gradient rides along as 8 extra MFMAs per step against a fragment of
// ones in the workgroups of the first K tile.
//
// LDS image of an operand tile: 32 rows x 512 B (4 channel blocks x [64 B hi | 64 B lo]); the 16-B chunk t of row r
// sits in slot t ^ ((r & 7) << 1): the 16 rows x 32 B one transposing read touches then cover all 64 banks twice
// (the minimum for 512 B).  As in gemm_x3 the permutation is applied on the DMA's SOURCE address.
#include "hfl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgParams {
  float* ws;                // (S, N, K) partial weight gradients
  float* wsb;               // (S, N) partial bias gradients or null
  const uint16_t* dy;       // (M, N/32, 2, 32) bf16
  const uint16_t* x;        // (M, K/32, 2, 32) bf16
  int64_t M;
  int N, K;
  int tiles_k, tiles;       // K / 128, (N / 128) * (K / 128)
  int64_t slab_rows;        // rows per slab (multiple of 32)
  int64_t n_wg;
};

constexpr int WG_TILE_B = 32 * 512;      // one operand tile of a 32-row step

__global__ void __launch_bounds__(256, 3)
wgrad_x3_kernel(const WgParams p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];      // dy tile | x tile
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int c = lane & 15, g = lane >> 4;

  // consecutive new ids share an XCD (bijective remap): the tiles of one slab re-read its rows from one L2
  int64_t wg = blockIdx.x;
  {
    const int64_t q = p.n_wg >> 3, r = p.n_wg & 7;
    const int64_t xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int64_t slab = wg / p.tiles;
  const int tile = (int)(wg % p.tiles);
  const int n0 = (tile / p.tiles_k) * 128, k0 = (tile % p.tiles_k) * 128;
  const int64_t m_begin = slab * p.slab_rows;
  const int64_t m_end = (m_begin + p.slab_rows < p.M) ? m_begin + p.slab_rows : p.M;
  const int nsteps = (int)((m_end - m_begin + 31) >> 5);
  const int64_t row_dy = (int64_t)p.N * 4, row_x = (int64_t)p.K * 4;          // bytes per split2 row

  // ---- staging: one wave-instruction moves 2 rows x 512 B; wave w owns rows [8w, 8w+8) of both tiles ------------------
  const int srow = lane >> 5, sslot = lane & 31;
  const unsigned char* dyb = reinterpret_cast<const unsigned char*>(p.dy) + (int64_t)n0 * 4;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + (int64_t)k0 * 4;
  // uniform bases (SGPRs) advanced by 32 rows per step + ONE 32-bit per-lane offset per operand: instruction i of the
  // wave covers rows +2i, whose swizzle term differs from instruction 0's in bits (i << 2) of the chunk index only
  // (row strides are multiples of 512 B, so the flip commutes with the row term)
  const int r0 = wave * 8 + srow;
  const uint32_t doff0 = (uint32_t)r0 * (uint32_t)row_dy + ((sslot ^ ((r0 & 7) << 1)) << 4);
  const uint32_t xoff0 = (uint32_t)r0 * (uint32_t)row_x + ((sslot ^ ((r0 & 7) << 1)) << 4);
  auto stage = [&](int st) {
    const int64_t m = m_begin + (int64_t)st * 32;
    const unsigned char* dk = dyb + m * row_dy;
    const unsigned char* xk = xb + m * row_x;
    if (m + 32 <= p.M) {
      uint32_t d0 = doff0, x0 = xoff0;
      asm volatile("" : "+v"(d0), "+v"(x0));           // derive the other offsets per step, not once into 14 registers
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(dk + 2 * i * row_dy + (d0 ^ (uint32_t)(i << 6))),
            (__attribute__((address_space(3))) void*)(smem + (wave * 8 + i * 2) * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(xk + 2 * i * row_x + (x0 ^ (uint32_t)(i << 6))),
            (__attribute__((address_space(3))) void*)(smem + WG_TILE_B + (wave * 8 + i * 2) * 512), 16, 0, 0);
      }
    } else {                                             // last step of the last slab: fetch valid rows only
      const int last = (int)(p.M - 1 - m);               // tail rows re-fetch row `last` and are zeroed after landing
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wave * 8 + i * 2 + srow;
        const int t = sslot ^ ((r & 7) << 1);
        const int rc = r < last ? r : last;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dk + rc * row_dy + t * 16),
                                         (__attribute__((address_space(3))) void*)(smem + (wave * 8 + i * 2) * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xk + rc * row_x + t * 16),
                                         (__attribute__((address_space(3))) void*)(smem + WG_TILE_B + (wave * 8 + i * 2) * 512),
                                         16, 0, 0);
      }
    }
  };

  // ---- fragment addresses: lane (c, g) addresses row 4g + (c >> 2), channels 4 (c & 3) .. +3 of its 16-channel tile --
  // 16-channel tile i of the wave's 64: logical 16-B chunk = 16 wn + 8 (i >> 1) + 2 (i & 1) + ((c & 3) >> 1) for the hi
  // half; the row's swizzle term flips bits 1..3, so tile i = tile 0 with byte-offset bits (i & 1) * 32 + (i >> 1) * 128
  // flipped as well
  const int fr = 4 * g + (c >> 2);
  const int offa0 = fr * 512 + (((wn * 16 + ((c & 3) >> 1)) ^ ((fr & 7) << 1)) << 4) + (c & 1) * 8;
  const int offb0 = WG_TILE_B + fr * 512 + (((wk * 16 + ((c & 3) >> 1)) ^ ((fr & 7) << 1)) << 4) + (c & 1) * 8;
  // lo half = logical chunk + 4 -> physical byte offset ^ 64; rows 16..31 of the step = + 16 * 512

  f32x4 acc[4][4];
  float accb[4] = {0.f, 0.f, 0.f, 0.f};                                        // column sums of dy (bias gradient)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.wsb != nullptr && k0 == 0 && wk == 0;                 // wave-uniform

  typedef __attribute__((address_space(3))) s16x4 lds_s4;
  auto frag = [&](int off) -> bf16x8 {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(smem + off));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(smem + off + 16 * 512));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  };

  if (nsteps > 0) stage(0);
  for (int st = 0; st < nsteps; ++st) {
    __syncthreads();                                   // the step's tiles have landed (vmcnt(0) precedes the barrier)
    const int64_t left = p.M - (m_begin + (int64_t)st * 32);
    if (left < 32) {                                   // last step of the last slab: rows past M contribute nothing
      const int dead = 32 - (int)left;
      for (int i = tid; i < dead * 32 * 2; i += 256) {
        const int tsel = i / (dead * 32), rem = i % (dead * 32);
        *reinterpret_cast<uint4*>(smem + tsel * WG_TILE_B + ((int)left + rem / 32) * 512 + (rem % 32) * 16) =
            make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    bf16x8 ah[4], al[4], bh[4], bl[4];
    int oa = offa0, ob = offb0;
    asm volatile("" : "+v"(oa), "+v"(ob));             // keep the 16 derived addresses out of loop-invariant registers
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fl = (i & 1) * 32 + (i >> 1) * 128;
      ah[i] = frag(oa ^ fl);
      al[i] = frag(oa ^ fl ^ 64);
      bh[i] = frag(ob ^ fl);
      bl[i] = frag(ob ^ fl ^ 64);
    }
    __syncthreads();                                   // every wave holds its fragments: the stage is free again
    if (st + 1 < nsteps) stage(st + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    if (do_bias) {                                     // VALU in the shadow of the MFMAs: 8 rows x (hi + lo) per tile
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 uh = __builtin_bit_cast(u32x4, ah[i]), ul = __builtin_bit_cast(u32x4, al[i]);
        float sacc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sacc += __uint_as_float(uh[e] << 16) + __uint_as_float(ul[e] << 16);
          sacc += __uint_as_float(uh[e] & 0xFFFF0000u) + __uint_as_float(ul[e] & 0xFFFF0000u);
        }
        accb[i] += sacc;
      }
    }
  }

  // ---- partial tile: lane (c, g) holds rows n = 4g..4g+3, column k = c of every 16 x 16 block --------------------------
  float* wsl = p.ws + slab * (int64_t)p.N * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * 64 + i * 16 + 4 * g + e;
      float* row = wsl + (int64_t)n * p.K + k0 + wk * 64 + c;
#pragma unroll
      for (int j = 0; j < 4; ++j) row[j * 16] = acc[i][j][e];
    }
  if (do_bias) {                                       // lane (c, g) summed rows {4g.., 16+4g..} of column c: add the 4 g
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = accb[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) p.wsb[slab * (int64_t)p.N + n0 + wn * 64 + i * 16 + c] = v;
    }
  }
}

// out[i] = sum over slabs (fixed order): 64 float4 columns x 4 slab groups per workgroup, LDS combine
// (the bias gradient's slabs ride in the same launch: workgroups past `blocks_a` reduce the second array -- 179 launches per
//  config-3 step fewer)
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(float* out, const float* ws, int64_t n4, int S, int blocks_a, float* out_b, const float* ws_b,
                    int64_t n4_b) {
  __shared__ float4 part[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  int64_t blk = blockIdx.x;
  if ((int)blockIdx.x >= blocks_a) {               // (workgroup-uniform)
    blk -= blocks_a;
    out = out_b;
    ws = ws_b;
    n4 = n4_b;
  }
  const int64_t i = blk * 64 + col;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* src = reinterpret_cast<const float4*>(ws) + i;
    for (int s = grp; s < S; s += 4) {
      const float4 v = src[(int64_t)s * n4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  part[grp][col] = a;
  __syncthreads();
  if (grp == 0 && i < n4) {
    float4 r = part[0][col];
#pragma unroll
    for (int k = 1; k < 4; ++k) { r.x += part[k][col].x; r.y += part[k][col].y; r.z += part[k][col].z; r.w += part[k][col].w; }
    reinterpret_cast<float4*>(out)[i] = r;
  }
}

struct WgSplit { int S; int64_t slab_rows; };

WgSplit wg_split(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = (N / 128) * (K / 128);
  int64_t S = hfl_cdiv(3 * (int64_t)hfl_num_cus(), tiles);
  const int64_t smax = hfl_cdiv(M, 256);               // at least 8 steps per slab
  if (S > smax) S = smax;
  if (S < 1) S = 1;
  int64_t slab = hfl_cdiv(hfl_cdiv(M, S), 32) * 32;
  S = hfl_cdiv(M, slab);
  return {(int)S, slab};
}

}  // namespace

extern "C" {

int64_t hfl_wgrad_x3_workspace(int64_t n_rows, int64_t out_features, int64_t in_features) {
  if (n_rows <= 0 || out_features <= 0 || in_features <= 0 || out_features % 128 != 0 || in_features % 128 != 0)
    return 0;
  const WgSplit sp = wg_split(n_rows, out_features, in_features);
  return (int64_t)sp.S * (out_features * in_features + out_features) * 4;
}

int hfl_wgrad_x3(float* dw, float* db, const uint16_t* dy2, const uint16_t* x2, int64_t n_rows, int64_t out_features,
                 int64_t in_features, void* workspace, hfl_stream_t stream) {
  if (n_rows <= 0 || out_features <= 0 || in_features <= 0 || out_features % 128 != 0 || in_features % 128 != 0 ||
      out_features > (1 << 20) || in_features > (1 << 20))
    return HFL_EINVAL;
  if (dw == nullptr || dy2 == nullptr || x2 == nullptr || workspace == nullptr) return HFL_EINVAL;
  const WgSplit sp = wg_split(n_rows, out_features, in_features);
  WgParams p;
  p.ws = static_cast<float*>(workspace);
  p.wsb = db != nullptr ? p.ws + (int64_t)sp.S * out_features * in_features : nullptr;
  p.dy = dy2;
  p.x = x2;
  p.M = n_rows;
  p.N = (int)out_features;
  p.K = (int)in_features;
  p.tiles_k = (int)(in_features / 128);
  p.tiles = (int)((out_features / 128) * (in_features / 128));
  p.slab_rows = sp.slab_rows;
  p.n_wg = (int64_t)sp.S * p.tiles;
  if (p.n_wg > 0x7fffffffLL) return HFL_ECAPACITY;
  hipStream_t s = static_cast<hipStream_t>(stream);
  wgrad_x3_kernel<<<(unsigned)p.n_wg, 256, 2 * WG_TILE_B, s>>>(p);
  const int64_t n4 = out_features * in_features / 4;
  const int blocks_a = (int)hfl_cdiv(n4, 64), blocks_b = db != nullptr ? (int)hfl_cdiv(out_features / 4, 64) : 0;
  wgrad_reduce_kernel<<<(unsigned)(blocks_a + blocks_b), 256, 0, s>>>(dw, p.ws, n4, sp.S, blocks_a, db, p.wsb, out_features / 4);
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
