// Attentional pooling of a pyramid level's ragged token rows by k learned queries, one launch (+ a small combine):
//
//     out[b, q, :] = sum_r softmax_r( scale * <query[q], x[r]> ) * x[r]          r over the rows of cloud b
//
// Replaces AdaptivePooling.forward of the reference (models/layers/salsa.py:25-55, called per level by
// PyramidAttnPoolWrapper, models/layers/pooling.py:209-233), which pads every cloud to the longest one, builds a key mask and
// runs softmax(Q X^T) X as two batched fp32 GEMMs; rounds 1-3 ran it as an fp32 GEMM for the scores, a segment softmax, two
// per-cloud padding copies and a batched GEMM (285 us of the 12 ms step, all of it after the last transformer block, on the
// critical path).  Here the scores never leave the register file and nothing is padded:
//
//   * workgroup = (cloud b, query group of 64, row chunk s of the cloud); wave w owns 16 queries as B-operand fragments
//     (bf16 (hi, lo), K = C: 64 VGPRs) and their output accumulators O (16 x C fp32: 64 VGPRs).
//   * the chunk's rows go through LDS 32 at a time as a bf16 (hi, lo) image (converted from fp32 on the way in, double
//     buffered); per step and wave
//         S^T (32 rows x 16 queries) = X Q^T        A = rows from LDS (ds_read_b128), B = queries from registers
//         online softmax over the rows: a lane holds 8 rows of ONE query (accumulator layout), the other 24 are two
//         butterflies away; running maximum m, running sum l, P = exp2(S - m)
//         O (16 queries x C) += P X                 A = P straight from the accumulator registers (the k order inside the
//         32-row step is permuted to the accumulator layout for both operands, which a contraction does not see),
//         B = columns of X by ds_read_b64_tr_b16 from the same LDS image
//   * arithmetic as everywhere on the default path: every fp32 operand as bf16 (hi, lo), three MFMAs per product
//     (v_mfma_f32_16x16x32_bf16), fp32 accumulation, softmax in fp32 (exp2 domain).
//   * S > 1 row chunks per cloud (so that ~two workgroups per CU exist): each writes (m, l, unnormalised O); a second kernel
//     merges them in fixed order (bitwise reproducible, no atomics).
#include "hfl_common.h"
#include "x3_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPoolWaves = 4;                 // 16 queries each: 64 queries per workgroup
constexpr float kPoolDead = -1.0e30f;

struct PoolParams {
  float* out;                 // (B, out_stride) : row q of cloud b at out + b * out_stride + q * C
  float* part_o;              // (B, G, S, 64, C) unnormalised partial sums   (S > 1)
  float* part_ml;             // (B, G, S, 64, 2) running maximum (exp2 domain), running sum
  const float* x;             // (N, C)
  const int64_t* row_off;     // (B + 1)
  const float* query;         // (k, C)
  int64_t out_stride;
  int k, G, S;
  float scale2;               // scale * log2(e)
};

// 16-B chunk t of image row r is stored at chunk t ^ swz(r): the 16 rows one ds_read_b128 quarter-wave touches (same logical
// chunk) land in 16 different chunks of the 256-B bank window, and the 8 rows x 32 B of one ds_read_b64_tr_b16 pass in 8
// different chunk pairs
__device__ __forceinline__ int pool_swz(int r) { return ((r & 7) << 1) | ((r >> 3) & 1); }

template <int C>
__global__ void __launch_bounds__(kPoolWaves * 64, 2)
attn_pool_kernel(const PoolParams p) {
  constexpr int KS = C / 32;                  // k-steps of the score product
  constexpr int FT = C / 16;                  // 16-channel tiles of the output
  constexpr int RS = C * 4;                   // bytes of an image row: C / 32 blocks of [64 B hi | 64 B lo]
  constexpr int IMG = 32 * RS;                // one 32-row image
  constexpr int NTHR = kPoolWaves * 64;
  constexpr int ITEMS = 32 * (C / 8);         // (row, 8-channel group) cells of an image
  constexpr int IPT = (ITEMS + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];      // two images

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;

  int wg = blockIdx.x;
  const int s = wg % p.S; wg /= p.S;
  const int grp = wg % p.G;
  const int b = wg / p.G;
  const int64_t r_begin = p.row_off[b], r_end = p.row_off[b + 1];
  const int64_t n_b = r_end - r_begin;
  // chunk s of the cloud: ceil(n_b / S) rows rounded up to whole 32-row steps
  const int64_t chunk = ((n_b + p.S - 1) / p.S + 31) / 32 * 32;
  const int64_t c_begin = r_begin + (int64_t)s * chunk;
  const int64_t c_end = c_begin + chunk < r_end ? c_begin + chunk : r_end;
  const int nsteps = c_begin < c_end ? (int)((c_end - c_begin + 31) / 32) : 0;

  const int q0 = grp * (kPoolWaves * 16) + wave * 16;         // first query of this wave
  const bool active = q0 < p.k;                               // wave-uniform

  // ---- the wave's queries as B-operand fragments: lane (c, g) holds query q0 + c, channels 32 ks + 8 g .. + 7
  bf16x8 qh[KS], ql[KS];
  {
    const int q = q0 + c;
    const bool have = q < p.k;
    const float* qr = p.query + (int64_t)(have ? q : 0) * C + g * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float4 a0 = *reinterpret_cast<const float4*>(qr + ks * 32);
      float4 a1 = *reinterpret_cast<const float4*>(qr + ks * 32 + 4);
      if (!have) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
      uint32_t hi[4], lo[4];
      x3_split_pair((f32x2){a0.x, a0.y}, hi[0], lo[0]);
      x3_split_pair((f32x2){a0.z, a0.w}, hi[1], lo[1]);
      x3_split_pair((f32x2){a1.x, a1.y}, hi[2], lo[2]);
      x3_split_pair((f32x2){a1.z, a1.w}, hi[3], lo[3]);
      qh[ks] = __builtin_bit_cast(bf16x8, (u32x4){hi[0], hi[1], hi[2], hi[3]});
      ql[ks] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], lo[2], lo[3]});
    }
  }

  // ---- staging: cell (row, 8-channel group) -> 16 B of hi and 16 B of lo in the image
  float4 st0[IPT], st1[IPT];
  auto stage_load = [&](int step) {
    const int64_t base = c_begin + (int64_t)step * 32;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int it = tid + i * NTHR;
      if (it < ITEMS) {
        const int r = it / (C / 8), grp8 = it % (C / 8);
        int64_t row = base + r;
        if (row >= c_end) row = c_end - 1;                     // rows past the chunk: a finite copy (their P is 0)
        const float* src = p.x + row * C + grp8 * 8;
        st0[i] = *reinterpret_cast<const float4*>(src);
        st1[i] = *reinterpret_cast<const float4*>(src + 4);
      }
    }
  };
  auto stage_store = [&](int buf) {
    unsigned char* img = smem + buf * IMG;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
      const int it = tid + i * NTHR;
      if (it < ITEMS) {
        const int r = it / (C / 8), grp8 = it % (C / 8);
        uint32_t hi[4], lo[4];
        x3_split_pair((f32x2){st0[i].x, st0[i].y}, hi[0], lo[0]);
        x3_split_pair((f32x2){st0[i].z, st0[i].w}, hi[1], lo[1]);
        x3_split_pair((f32x2){st1[i].x, st1[i].y}, hi[2], lo[2]);
        x3_split_pair((f32x2){st1[i].z, st1[i].w}, hi[3], lo[3]);
        const int t = (grp8 >> 2) * 8 + (grp8 & 3);            // logical chunk of the hi half; lo = + 4
        const int sw = pool_swz(r);
        *reinterpret_cast<uint4*>(img + r * RS + ((t ^ sw) << 4)) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(img + r * RS + (((t + 4) ^ sw) << 4)) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  };

  // ---- running state: lane (c, g) = query q0 + c; m is the same in the four g lanes, l is this lane's share of the sum
  float m_run = kPoolDead, l_run = 0.f;
  f32x4 oacc[FT];
#pragma unroll
  for (int i = 0; i < FT; ++i) oacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses inside an image
  // scores: lane (c, g) reads row 16 t + c, logical chunk 8 ks + g (hi), + 4 (lo)
  const int sw_a = pool_swz(c);                                // rows c and 16 + c share the swizzle term
  // columns: lane (c, g) addresses row 4 g + (c >> 2) (+ 16), channels 4 (c & 3) .. + 3 of 16-channel tile i:
  // logical chunk 8 (i >> 1) + 2 (i & 1) + ((c & 3) >> 1) (hi), + 4 (lo), byte (c & 1) * 8
  const int tr_row = 4 * g + (c >> 2);
  const int sw_t = pool_swz(tr_row);
  const int tr_base = tr_row * RS + (c & 1) * 8;
  typedef __attribute__((address_space(3))) s16x4 lds_s4;

  if (nsteps > 0) {
    stage_load(0);
    stage_store(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const unsigned char* img = smem + (step & 1) * IMG;
    if (step + 1 < nsteps) stage_load(step + 1);               // in flight across this step's products
    if (active) {
      // ---- S^T = X Q^T for the two 16-row tiles of the step
      f32x4 sc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const unsigned char* rowp = img + (16 * t + c) * RS;
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(rowp + (((8 * ks + g) ^ sw_a) << 4));
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(rowp + (((8 * ks + 4 + g) ^ sw_a) << 4));
          sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, qh[ks], sc[t], 0, 0, 0);
          sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ql[ks], sc[t], 0, 0, 0);
          sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, qh[ks], sc[t], 0, 0, 0);
        }
      }
      // ---- online softmax: lane (c, g) holds rows 16 t + 4 g + r of query c
      const int64_t row0 = c_begin + (int64_t)step * 32;
      float v[8];
      float mx = kPoolDead;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = row0 + 16 * t + 4 * g + r < c_end;
          v[4 * t + r] = ok ? sc[t][r] * p.scale2 : kPoolDead;
          mx = fmaxf(mx, v[4 * t + r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = __builtin_amdgcn_exp2f(v[e] - m_new);
        ps += v[e];
      }
      l_run = l_run * alpha + ps;
      uint32_t ph[4], pl[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) x3_split_pair((f32x2){v[2 * e], v[2 * e + 1]}, ph[e], pl[e]);
      const bf16x8 pH = __builtin_bit_cast(bf16x8, (u32x4){ph[0], ph[1], ph[2], ph[3]});
      const bf16x8 pL = __builtin_bit_cast(bf16x8, (u32x4){pl[0], pl[1], pl[2], pl[3]});
      // the accumulators hold queries 4 g + r on the lane: their rescale factors live in lanes 4 g + r
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {               // (wave-uniform; the maximum settles early)
        float ar[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * g + r, 64);
#pragma unroll
        for (int i = 0; i < FT; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[i][r] *= ar[r];
      }
      // ---- O += P X: columns of X by the transposing read (k slot (g, e) = row 16 (e >> 2) + 4 g + (e & 3), as P has it)
#pragma unroll
      for (int i = 0; i < FT; ++i) {
        const int lt = 8 * (i >> 1) + 2 * (i & 1) + ((c & 3) >> 1);
        const unsigned char* ah_p = img + tr_base + ((lt ^ sw_t) << 4);
        const unsigned char* al_p = img + tr_base + (((lt + 4) ^ sw_t) << 4);
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(ah_p));
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(ah_p + 16 * RS));
        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(al_p));
        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)(al_p + 16 * RS));
        const bf16x8 xh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 xl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pL, xh, oacc[i], 0, 0, 0);
        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pH, xl, oacc[i], 0, 0, 0);
        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pH, xh, oacc[i], 0, 0, 0);
      }
    }
    if (step + 1 < nsteps) stage_store((step + 1) & 1);        // (the other image: last read in step - 1, before the barrier below)
    __syncthreads();
  }

  if (!active) return;
  // ---- the cloud's (chunk's) result: lane (c, g) holds channel 16 i + c of queries 4 g + r
  float l_tot = l_run + __shfl_xor(l_run, 16, 64);
  l_tot += __shfl_xor(l_tot, 32, 64);
  float lr[4], mr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    lr[r] = __shfl(l_tot, 4 * g + r, 64);
    mr[r] = __shfl(m_run, 4 * g + r, 64);
  }
  if (p.S == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + 4 * g + r;
      if (q >= p.k) continue;
      const float inv = lr[r] > 0.f ? 1.0f / lr[r] : 0.f;
      float* orow = p.out + (int64_t)b * p.out_stride + (int64_t)q * C + c;
#pragma unroll
      for (int i = 0; i < FT; ++i) orow[i * 16] = oacc[i][r] * inv;
    }
  } else {
    const int64_t slot = ((int64_t)(b * p.G + grp) * p.S + s) * (kPoolWaves * 16) + wave * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* orow = p.part_o + (slot + 4 * g + r) * C + c;
#pragma unroll
      for (int i = 0; i < FT; ++i) orow[i * 16] = oacc[i][r];
      if (c == 0) {
        p.part_ml[(slot + 4 * g + r) * 2] = mr[r];
        p.part_ml[(slot + 4 * g + r) * 2 + 1] = lr[r];
      }
    }
  }
}

// out[b, q, :] = sum_s O_s 2^(m_s - M) / sum_s l_s 2^(m_s - M), s in fixed order; one float4 per lane
__global__ void __launch_bounds__(256)
attn_pool_combine_kernel(const PoolParams p, int C, int batch) {
  const int c4 = C / 4;
  const int64_t n = (int64_t)batch * p.k * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c4) * 4;
    const int q = (int)((i / c4) % p.k);
    const int b = (int)(i / ((int64_t)c4 * p.k));
    const int grp = q / (kPoolWaves * 16), ql = q % (kPoolWaves * 16);
    const int64_t slot0 = ((int64_t)(b * p.G + grp) * p.S) * (kPoolWaves * 16) + ql;
    float M = kPoolDead;
    for (int s = 0; s < p.S; ++s) M = fmaxf(M, p.part_ml[(slot0 + (int64_t)s * (kPoolWaves * 16)) * 2]);
    float L = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < p.S; ++s) {
      const int64_t slot = slot0 + (int64_t)s * (kPoolWaves * 16);
      const float l = p.part_ml[slot * 2 + 1];
      if (l <= 0.f) continue;                                  // an empty chunk
      const float w = __builtin_amdgcn_exp2f(p.part_ml[slot * 2] - M);
      const float4 o = *reinterpret_cast<const float4*>(p.part_o + slot * C + ch);
      L += l * w;
      acc.x += o.x * w; acc.y += o.y * w; acc.z += o.z * w; acc.w += o.w * w;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    *reinterpret_cast<float4*>(p.out + (int64_t)b * p.out_stride + (int64_t)q * C + ch) =
        make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
}

struct PoolPlan {
  int G, S;
};
static PoolPlan pool_plan(int batch, int n_queries, int64_t n_rows, int cus) {
  PoolPlan pl;
  pl.G = (n_queries + kPoolWaves * 16 - 1) / (kPoolWaves * 16);
  int s = 2 * cus / (batch * pl.G > 0 ? batch * pl.G : 1);     // two workgroups per CU (4 waves, 64 KB of LDS each)
  const int64_t avg = batch > 0 ? n_rows / batch : 0;
  while (s > 1 && avg / s < 96) --s;           // at least three 32-row steps per chunk
  pl.S = s < 1 ? 1 : (s > 16 ? 16 : s);
  return pl;
}

}  // namespace

extern "C" {

/* workspace of hfl_attn_pool for this shape (bytes; covers every CU-masked stream of the library) */
int64_t hfl_attn_pool_workspace(int batch, int n_queries, int channels, int64_t n_rows) {
  if (batch <= 0 || n_queries <= 0 || (channels != 128 && channels != 256)) return 0;
  int64_t need = 0;
  for (int cus = 8; cus <= hfl_num_cus(); cus += 8) {
    const PoolPlan pl = pool_plan(batch, n_queries, n_rows, cus);
    const int64_t b = pl.S > 1 ? (int64_t)batch * pl.G * pl.S * (kPoolWaves * 16) * (channels + 2) * 4 : 0;
    if (b > need) need = b;
  }
  return need;
}

int hfl_attn_pool_ok(int channels) { return channels == 128 || channels == 256; }

int hfl_attn_pool(float* out, int64_t out_cloud_stride, const float* x, const int64_t* row_off, const float* query, int batch,
                  int n_queries, int channels, int64_t n_rows, float scale, void* workspace, int64_t workspace_bytes,
                  hfl_stream_t stream) {
  if (out == nullptr || x == nullptr || row_off == nullptr || query == nullptr || batch < 0 || n_queries <= 0 || n_rows < 0)
    return HFL_EINVAL;
  if (channels != 128 && channels != 256) return HFL_EINVAL;
  if (batch == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  PoolPlan pl = pool_plan(batch, n_queries, n_rows, hfl_stream_cus(s));
  const int64_t need = pl.S > 1 ? (int64_t)batch * pl.G * pl.S * (kPoolWaves * 16) * (channels + 2) * 4 : 0;
  if (need > 0 && (workspace == nullptr || workspace_bytes < need)) pl.S = 1;
  PoolParams p;
  p.out = out; p.out_stride = out_cloud_stride; p.x = x; p.row_off = row_off; p.query = query;
  p.k = n_queries; p.G = pl.G; p.S = pl.S;
  p.scale2 = scale * 1.4426950408889634f;
  p.part_o = static_cast<float*>(workspace);
  p.part_ml = pl.S > 1 ? p.part_o + (int64_t)batch * pl.G * pl.S * (kPoolWaves * 16) * channels : nullptr;
  const int64_t grid = (int64_t)batch * pl.G * pl.S;
  if (grid > 0x7fffffffLL) return HFL_ECAPACITY;
  const size_t lds = (size_t)2 * 32 * channels * 4;
  if (channels == 256) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_pool_kernel<256>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attn_pool_kernel<256><<<(unsigned)grid, kPoolWaves * 64, lds, s>>>(p);
  } else {
    attn_pool_kernel<128><<<(unsigned)grid, kPoolWaves * 64, lds, s>>>(p);
  }
  if (pl.S > 1) {
    const int64_t n = (int64_t)batch * n_queries * (channels / 4);
    attn_pool_combine_kernel<<<(unsigned)hfl_cdiv(n, 256), 256, 0, s>>>(p, channels, batch);
  }
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Tail of the Mixer aggregator (models/layers/salsa.py:104-111): channel_proj over the token axis, then row_proj over the
// channel axis, flattened:
//     out[b, o * D + j] = sum_c ( sum_t Wc[o, t] x[b, t, c] + bc[o] ) Wr[j, c] + br[j]
// The reference (and rounds 1-4) ran it as permute -> Linear -> permute -> Linear -> flatten: a transposing copy of the
// (B, K, C) token matrix, two library GEMMs and two bias passes for 2 x 0.3 MFLOP per cloud.  row_proj shrinks C to D = 4
// first when the order of the two (linear) maps is exchanged:
//     U[t, j] = sum_c x[b, t, c] Wr[j, c]   (K x D, in LDS)      out = sum_t Wc[o, t] U[t, j] + bc[o] sum_c Wr[j, c] + br[j]
// -- the same value up to fp32 summation order, 13x less arithmetic, one launch: one workgroup per cloud.
namespace {

constexpr int kTailMaxD = 8;

// One workgroup per cloud, 16 waves (round 6; 4 waves walked 65 tokens each with the row_proj rows re-read per token and one
// lane per output looped over all tokens: 122 us at the very end of the forward, where nothing overlaps it).
__global__ void __launch_bounds__(1024)
mixer_tail_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ wc,
                  const float* __restrict__ bc, const float* __restrict__ wr, const float* __restrict__ br, int K, int C, int KO,
                  int D) {
  extern __shared__ float tail_lds[];                 // U (K x D) | sw (D)
  float* U = tail_lds;
  float* sw = tail_lds + K * D;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = 16;
  const float* xb = x + (int64_t)b * K * C;
  // ---- U: wave w takes tokens w, w + 16, ...; a lane holds 4 channels of every 256-channel slab.  C == 256 (every shipped
  // configuration): the lane's row_proj weights stay in registers and four tokens are in flight.
  if (C == 256) {
    float4 w[kTailMaxD];
#pragma unroll
    for (int j = 0; j < kTailMaxD; ++j)
      w[j] = j < D ? *reinterpret_cast<const float4*>(wr + (int64_t)j * C + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = wave; t0 < K; t0 += 4 * NW) {
      float4 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * NW;
        xv[u] = *reinterpret_cast<const float4*>(xb + (int64_t)(t < K ? t : t0) * C + lane * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * NW;
#pragma unroll
        for (int j = 0; j < kTailMaxD; ++j) {
          if (j < D) {
            float v = (xv[u].x * w[j].x + xv[u].y * w[j].y) + (xv[u].z * w[j].z + xv[u].w * w[j].w);
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
            if (lane == 0 && t < K) U[t * D + j] = v;
          }
        }
      }
    }
  } else {
    for (int t = wave; t < K; t += NW) {
      float acc[kTailMaxD];
#pragma unroll
      for (int j = 0; j < kTailMaxD; ++j) acc[j] = 0.f;
      for (int c0 = lane * 4; c0 < C; c0 += 256) {
        const float4 xv = *reinterpret_cast<const float4*>(xb + (int64_t)t * C + c0);
#pragma unroll
        for (int j = 0; j < kTailMaxD; ++j) {
          if (j < D) {
            const float4 w = *reinterpret_cast<const float4*>(wr + (int64_t)j * C + c0);
            acc[j] += (xv.x * w.x + xv.y * w.y) + (xv.z * w.z + xv.w * w.w);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kTailMaxD; ++j) {
        if (j < D) {
          float v = acc[j];
#pragma unroll
          for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
          if (lane == 0) U[t * D + j] = v;
        }
      }
    }
  }
  if (wave == NW - 1) {                              // sw[j] = sum_c Wr[j, c]
    for (int j = 0; j < D; ++j) {
      float v = 0.f;
      for (int c = lane; c < C; c += 64) v += wr[(int64_t)j * C + c];
#pragma unroll
      for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
      if (lane == 0) sw[j] = v;
    }
  }
  __syncthreads();
  // ---- out: wave w takes output tokens o = w, w + 16, ...; lanes run over the input tokens (one coalesced read of the
  // channel_proj row), the D sums by butterfly
  for (int o = wave; o < KO; o += NW) {
    float acc[kTailMaxD];
#pragma unroll
    for (int j = 0; j < kTailMaxD; ++j) acc[j] = 0.f;
    for (int t = lane; t < K; t += 64) {
      const float wv = wc[(int64_t)o * K + t];
#pragma unroll
      for (int j = 0; j < kTailMaxD; ++j)
        if (j < D) acc[j] = fmaf(wv, U[t * D + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kTailMaxD; ++j) {
      if (j < D) {
        float v = acc[j];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
        if (lane == 0) out[(int64_t)b * KO * D + o * D + j] = v + bc[o] * sw[j] + br[j];
      }
    }
  }
}

}  // namespace

extern "C" int hfl_mixer_tail(float* out, const float* x, const float* channel_w, const float* channel_b, const float* row_w,
                              const float* row_b, int batch, int k_tokens, int channels, int k_out, int out_d,
                              hfl_stream_t stream) {
  if (out == nullptr || x == nullptr || channel_w == nullptr || channel_b == nullptr || row_w == nullptr || row_b == nullptr)
    return HFL_EINVAL;
  if (batch < 0 || k_tokens <= 0 || channels <= 0 || channels % 4 != 0 || k_out <= 0 || out_d <= 0 || out_d > kTailMaxD)
    return HFL_EINVAL;
  const size_t lds = (size_t)(k_tokens * out_d + out_d) * sizeof(float);
  if (lds > 64 * 1024) return HFL_ECAPACITY;
  if (batch == 0) return HFL_OK;
  mixer_tail_kernel<<<(unsigned)batch, 1024, lds, static_cast<hipStream_t>(stream)>>>(out, x, channel_w, channel_b, row_w, row_b,
                                                                                   k_tokens, channels, k_out, out_d);
  HFL_RETURN_LAST_ERROR();
}
