// HBM-bound helpers of the HOTFormerLoc hot path on gfx950: octree-conv gather
// (octree2col), relay-token initialisation (masked window mean), ADaPE window
// statistics and the segment softmax of the attentional pooling head.
#include "hfl_common.h"

namespace {

// ------------------------------------------------------------------- gather
// out[m, k*C + c] = neigh[m,k] >= 0 ? data[neigh[m,k], c] : 0
// VEC=4: one lane moves 16 B; a (m,k) slab of C floats is contiguous in `out`.
template <int VEC>
__global__ void gather_kernel(float* __restrict__ out, const float* __restrict__ data,
                              const int32_t* __restrict__ neigh, int64_t n_out, int K, int C) {
  const int cv = C / VEC;
  const int64_t total = n_out * K * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t mk = i / cv;
    const int c = (int)(i % cv);
    const int32_t ni = neigh[mk];
    if (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ni >= 0) v = reinterpret_cast<const float4*>(data + (int64_t)ni * C)[c];
      reinterpret_cast<float4*>(out + mk * C)[c] = v;
    } else {
      out[mk * C + c] = ni >= 0 ? data[(int64_t)ni * C + c] : 0.f;
    }
  }
}


// ------------------------------------------------- backward of the octree-conv gather
// dcol (M, K*C) -> ddata (N, C):  ddata[n, c] = sum_k [ineigh[n,k] >= 0] dcol[ineigh[n,k], k*C + c]
// (the scatter-add of octree2col turned into a gather through the inverse table, which is
//  injective per column: no atomics, bitwise reproducible)
__global__ void gather_bwd_kernel(float* __restrict__ ddata, const float* __restrict__ dcol,
                                  const int32_t* __restrict__ ineigh, int64_t n_rows, int K, int C) {
  const int cv = C / 4;
  const int64_t total = n_rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / cv;
    const int c = (int)(i % cv);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const int32_t m = ineigh[n * K + k];
      if (m >= 0) {
        const float4 v = reinterpret_cast<const float4*>(dcol + ((int64_t)m * K + k) * C)[c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    reinterpret_cast<float4*>(ddata + n * C)[c] = acc;
  }
}

__global__ void gather_bwd_scalar_kernel(float* __restrict__ ddata, const float* __restrict__ dcol,
                                         const int32_t* __restrict__ ineigh, int64_t n_rows, int K, int C) {
  const int64_t total = n_rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / C;
    const int c = (int)(i % C);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int32_t m = ineigh[n * K + k];
      if (m >= 0) acc += dcol[((int64_t)m * K + k) * C + c];
    }
    ddata[i] = acc;
  }
}

// backward of relay-token init: dx[t] = drt[window(t)] / count(window) for the owner's tokens, else 0
__global__ void relay_init_bwd_kernel(float* __restrict__ dx, const float* __restrict__ drt,
                                      const uint32_t* __restrict__ meta, int64_t n_tokens,
                                      int32_t n_windows, int K, int C) {
  const int cv = C / 4;
  const int c = threadIdx.x;
  for (int w = blockIdx.x; w < n_windows; w += gridDim.x) {
    const int64_t t0 = (int64_t)w * K;
    if (t0 >= n_tokens) continue;
    const uint32_t owner = meta[2 * t0 + 1];
    int cnt = 0;
    for (int k = 0; k < K; ++k) {
      const int64_t t = t0 + k;
      if (t >= n_tokens || meta[2 * t + 1] != owner) break;
      ++cnt;
    }
    float4 gsc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cv) {
      const float4 gr = reinterpret_cast<const float4*>(drt + (int64_t)w * C)[c];
      const float n = (float)cnt;
      gsc = make_float4(gr.x / n, gr.y / n, gr.z / n, gr.w / n);
    }
    for (int k = 0; k < K; ++k) {
      const int64_t t = t0 + k;
      if (t >= n_tokens) break;
      if (c < cv)
        reinterpret_cast<float4*>(dx + t * C)[c] = k < cnt ? gsc : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// --------------------------------------------------------------- relay tokens
// one block per window; thread c4 owns 4 channels; owner = batch id of first token.
// The owner's tokens lead the window (batch ids are non-decreasing along the token stream): their count comes from ONE ballot
// over the window's batch ids, then the rows are read eight at a time from always-valid addresses and added in token order
// (the sum is bitwise the sequential one).  Rounds 1-5 walked the window token by token -- id, compare, branch, row -- i.e.
// K dependent L2 round trips per launch: 22 us for a (windows x C) mean.
__global__ void relay_init_kernel(float* __restrict__ rt, const float* __restrict__ x,
                                  const uint32_t* __restrict__ meta, int64_t n_tokens,
                                  int32_t n_windows, int K, int C) {
  const int cv = C / 4;
  const int c = threadIdx.x;
  const int lane = threadIdx.x & 63;
  for (int w = blockIdx.x; w < n_windows; w += gridDim.x) {
    const int64_t t0 = (int64_t)w * K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;
    if (t0 < n_tokens) {
      const uint32_t owner = meta[2 * t0 + 1];
      for (int k0 = 0; k0 < K; k0 += 64) {          // (every wave of the block computes the same count)
        const int k = k0 + lane;
        const int64_t t = t0 + k;
        const bool mine = k < K && t < n_tokens && meta[2 * (t < n_tokens ? t : t0) + 1] == owner;
        const unsigned long long lead = ~__ballot(mine);
        const int run = lead == 0ull ? 64 : __builtin_ctzll(lead);
        cnt += run;
        if (run < 64) break;
      }
      if (c < cv) {
        for (int k0 = 0; k0 < cnt; k0 += 8) {
          float4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4*>(x + (t0 + (k0 + j < cnt ? k0 + j : 0)) * C)[c];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (k0 + j < cnt) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
      }
    }
    if (c < cv) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cnt > 0) {
        const float n = (float)cnt;
        o = make_float4(acc.x / n, acc.y / n, acc.z / n, acc.w / n);
      }
      reinterpret_cast<float4*>(rt + (int64_t)w * C)[c] = o;
    }
  }
}

// ------------------------------------------------------------- window stats
// one thread per window: mean (3) + upper-triangular Bessel covariance (6)
__global__ void window_stats_kernel(float* __restrict__ stats, const uint32_t* __restrict__ meta,
                                    int64_t n_tokens, int32_t n_windows, int K, float scale) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_windows) return;
  float* o = stats + (int64_t)w * 9;
  const int64_t t0 = (int64_t)w * K;
  float out[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (t0 < n_tokens) {
    const uint32_t owner = meta[2 * t0 + 1];
    int cnt = 0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = 0; k < K; ++k) {
      const int64_t t = t0 + k;
      if (t >= n_tokens || meta[2 * t + 1] != owner) break;
      const uint32_t p = meta[2 * t];
      sx += (float)(p & 1023u) * scale - 1.0f;
      sy += (float)((p >> 10) & 1023u) * scale - 1.0f;
      sz += (float)((p >> 20) & 1023u) * scale - 1.0f;
      ++cnt;
    }
    const float c = (float)cnt;
    const float cs = c < 1.f ? 1.f : c;
    const float mx = sx / cs, my = sy / cs, mz = sz / cs;
    out[0] = mx; out[1] = my; out[2] = mz;
    if (cnt >= 2) {
      float cxx = 0.f, cxy = 0.f, cxz = 0.f, cyy = 0.f, cyz = 0.f, czz = 0.f;
      for (int k = 0; k < cnt; ++k) {
        const uint32_t p = meta[2 * (t0 + k)];
        const float dx = ((float)(p & 1023u) * scale - 1.0f) - mx;
        const float dy = ((float)((p >> 10) & 1023u) * scale - 1.0f) - my;
        const float dz = ((float)((p >> 20) & 1023u) * scale - 1.0f) - mz;
        cxx += dx * dx; cxy += dx * dy; cxz += dx * dz;
        cyy += dy * dy; cyz += dy * dz; czz += dz * dz;
      }
      const float den = c - 1.0f;
      out[3] = cxx / den; out[4] = cxy / den; out[5] = cxz / den;
      out[6] = cyy / den; out[7] = cyz / den; out[8] = czz / den;
    }
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = out[i];
}

// ---------------------------------------------------------- segment softmax
// block = 64 query columns x 16 row groups; one block per (cloud, 64-column chunk)
__global__ void __launch_bounds__(1024)
segment_softmax_kernel(float* __restrict__ scores, const int64_t* __restrict__ row_off,
                       int n_queries, float scale) {
  __shared__ float s_m[16][64];
  __shared__ float s_l[16][64];
  const int b = blockIdx.x;
  const int q = blockIdx.y * 64 + (threadIdx.x & 63);
  const int grp = threadIdx.x >> 6;
  const int64_t r0 = row_off[b], r1 = row_off[b + 1];
  const bool live = q < n_queries;
  float m = -INFINITY, l = 0.f;
  if (live) {
    for (int64_t r = r0 + grp; r < r1; r += 16) {
      const float s = scores[r * n_queries + q] * scale;
      const float mn = fmaxf(m, s);
      l = l * __expf(m - mn) + __expf(s - mn);
      m = mn;
    }
  }
  s_m[grp][threadIdx.x & 63] = m;
  s_l[grp][threadIdx.x & 63] = l;
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int g = 0; g < 16; ++g) M = fmaxf(M, s_m[g][threadIdx.x & 63]);
  float L = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) {
    const float mg = s_m[g][threadIdx.x & 63];
    if (mg > -INFINITY) L += s_l[g][threadIdx.x & 63] * __expf(mg - M);
  }
  if (live && L > 0.f) {
    const float inv = 1.0f / L;
    for (int64_t r = r0 + grp; r < r1; r += 16) {
      const int64_t idx = r * n_queries + q;
      scores[idx] = __expf(scores[idx] * scale - M) * inv;
    }
  }
}


// ------------------------------------------------------------------ LayerNorm
// One row = TPR lanes x VPL float4 (C = 4*TPR*VPL); 256-thread blocks, grid-stride over rows.
// Optional fused residual: xo = x + y (+ bias), h = LN(xo).  Pure streaming (HBM-bound).
// fp32 -> (hi, lo) bf16 with hi = RNE(v), lo = RNE(v - hi): v ~= hi + lo to 2^-17 relative.
__device__ __forceinline__ uint16_t hfl_bf16_rne(float v) {
  uint32_t u = __float_as_uint(v);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ void hfl_split4(const float4 v, uint2& hi, uint2& lo) {
  const uint16_t h0 = hfl_bf16_rne(v.x), h1 = hfl_bf16_rne(v.y), h2 = hfl_bf16_rne(v.z), h3 = hfl_bf16_rne(v.w);
  const uint16_t l0 = hfl_bf16_rne(v.x - __uint_as_float((uint32_t)h0 << 16));
  const uint16_t l1 = hfl_bf16_rne(v.y - __uint_as_float((uint32_t)h1 << 16));
  const uint16_t l2 = hfl_bf16_rne(v.z - __uint_as_float((uint32_t)h2 << 16));
  const uint16_t l3 = hfl_bf16_rne(v.w - __uint_as_float((uint32_t)h3 << 16));
  hi = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
  lo = make_uint2((uint32_t)l0 | ((uint32_t)l1 << 16), (uint32_t)l2 | ((uint32_t)l3 << 16));
}
// row of the split-GEMM A operand: [hi (C) | hi (C) | lo (C)] bf16, so that one bf16 GEMM against
// [w_hi | w_lo | w_hi] accumulates hi*hi + hi*lo + lo*hi in fp32
__device__ __forceinline__ void hfl_store_split3(uint16_t* row, int C, int c4, const float4 v) {
  uint2 hi, lo;
  hfl_split4(v, hi, lo);
  reinterpret_cast<uint2*>(row)[c4] = hi;
  reinterpret_cast<uint2*>(row + C)[c4] = hi;
  reinterpret_cast<uint2*>(row + 2 * C)[c4] = lo;
}

// row of the hand-written split GEMM's operand (csrc/gemm_x3.hip): per 32-channel block [32 x hi | 32 x lo]
__device__ __forceinline__ void hfl_store_split2(uint16_t* row, int c4, const float4 v) {
  uint2 hi, lo;
  hfl_split4(v, hi, lo);
  uint16_t* o = row + (c4 >> 3) * 64 + (c4 & 7) * 4;
  *reinterpret_cast<uint2*>(o) = hi;
  *reinterpret_cast<uint2*>(o + 32) = lo;
}

// SPLIT: 0 = fp32 output, 1 = bf16 [hi|hi|lo] (K-concatenated, hipBLASLt route), 2 = bf16 split2 (gemm_x3 route)
template <int TPR, int VPL, bool ADD, int SPLIT>
__global__ void __launch_bounds__(256)
layer_norm_kernel(float* __restrict__ h_out, float* x_out, const float* x, const float* __restrict__ y,
                  const float* __restrict__ bias, const float* __restrict__ gamma,
                  const float* __restrict__ beta, int64_t n_rows, float eps, int relu) {
  constexpr int C = TPR * VPL * 4;
  constexpr int RPB = 256 / TPR;
  const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
  float4 gm[VPL], bt[VPL], bs[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    gm[v] = reinterpret_cast<const float4*>(gamma)[v * TPR + tx];
    bt[v] = reinterpret_cast<const float4*>(beta)[v * TPR + tx];
    bs[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ADD && bias != nullptr) bs[v] = reinterpret_cast<const float4*>(bias)[v * TPR + tx];
  }
  const float inv_c = 1.0f / (float)C;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < n_rows; base += (int64_t)gridDim.x * RPB) {
    const int64_t r = base + ty;
    const bool live = r < n_rows;
    float4 a[VPL];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      a[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) {
        a[v] = reinterpret_cast<const float4*>(x + r * C)[v * TPR + tx];
        if (ADD) {
          const float4 b = reinterpret_cast<const float4*>(y + r * C)[v * TPR + tx];
          a[v].x += b.x + bs[v].x; a[v].y += b.y + bs[v].y;
          a[v].z += b.z + bs[v].z; a[v].w += b.w + bs[v].w;
          reinterpret_cast<float4*>(x_out + r * C)[v * TPR + tx] = a[v];
        }
      }
      sum += (a[v].x + a[v].y) + (a[v].z + a[v].w);
    }
    const float mean = hfl_group_sum<TPR>(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      a[v].x -= mean; a[v].y -= mean; a[v].z -= mean; a[v].w -= mean;
      sq += (a[v].x * a[v].x + a[v].y * a[v].y) + (a[v].z * a[v].z + a[v].w * a[v].w);
    }
    const float rstd = 1.0f / sqrtf(hfl_group_sum<TPR>(sq) * inv_c + eps);
    if (live) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        float4 o;
        o.x = fmaf(a[v].x * rstd, gm[v].x, bt[v].x);
        o.y = fmaf(a[v].y * rstd, gm[v].y, bt[v].y);
        o.z = fmaf(a[v].z * rstd, gm[v].z, bt[v].z);
        o.w = fmaf(a[v].w * rstd, gm[v].w, bt[v].w);
        if (relu) {                              // conv -> norm -> ReLU of the stem (octformer_layers.py:80-98) in one pass
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (SPLIT == 1)
          hfl_store_split3(reinterpret_cast<uint16_t*>(h_out) + r * 3 * C, C, v * TPR + tx, o);
        else if (SPLIT == 2)
          hfl_store_split2(reinterpret_cast<uint16_t*>(h_out) + r * 2 * C, v * TPR + tx, o);
        else
          reinterpret_cast<float4*>(h_out + r * C)[v * TPR + tx] = o;
      }
    }
  }
}

template <int TPR, int VPL>
static int launch_ln(float* h_out, float* x_out, const float* x, const float* y, const float* bias,
                     const float* gamma, const float* beta, int64_t n, float eps, int split_flags,
                     hipStream_t s) {
  const int split = split_flags & 0xF, relu = (split_flags >> 4) & 1;
  constexpr int RPB = 256 / TPR;
  const int64_t need = hfl_cdiv(n, RPB);
  const int64_t cap = (int64_t)hfl_stream_cus(s) * 16;
  const int blocks = (int)(need < cap ? need : cap);
  if (y != nullptr) {
    if (split == 2)
      layer_norm_kernel<TPR, VPL, true, 2><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
    else if (split)
      layer_norm_kernel<TPR, VPL, true, 1><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
    else
      layer_norm_kernel<TPR, VPL, true, 0><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
  } else {
    if (split == 2)
      layer_norm_kernel<TPR, VPL, false, 2><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
    else if (split)
      layer_norm_kernel<TPR, VPL, false, 1><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
    else
      layer_norm_kernel<TPR, VPL, false, 0><<<blocks, 256, 0, s>>>(h_out, x_out, x, y, bias, gamma, beta, n, eps, relu);
  }
  HFL_RETURN_LAST_ERROR();
}

static int dispatch_ln(float* h_out, float* x_out, const float* x, const float* y, const float* bias,
                       const float* gamma, const float* beta, int64_t n, int64_t C, float eps,
                       int split, hipStream_t s) {
  if (n == 0) return HFL_OK;
  switch (C) {
    case 16:   return launch_ln<4, 1>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 32:   return launch_ln<8, 1>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 64:   return launch_ln<16, 1>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 128:  return launch_ln<32, 1>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 256:  return launch_ln<64, 1>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 512:  return launch_ln<64, 2>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    case 1024: return launch_ln<64, 4>(h_out, x_out, x, y, bias, gamma, beta, n, eps, split, s);
    default:   return HFL_EINVAL;
  }
}

// ------------------------------------------------------------------ LayerNorm backward (training path)
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma,  xhat = (x - mean) * rstd   (statistics are
// recomputed from x: the forward keeps nothing but its input), and per-workgroup partial sums of
// dgamma = sum_rows dy * xhat, dbeta = sum_rows dy  (fixed order: lane registers over the workgroup's rows, then one LDS
// pass over the row slots; the (blocks, C) partials are summed by the caller -- no atomics, bitwise reproducible).
// Replaces torch's native_layer_norm_backward on the training path (three kernels, 40 % slower forward).
template <int TPR, int VPL>
__global__ void __launch_bounds__(256)
layer_norm_bwd_kernel(float* __restrict__ dx, float* __restrict__ dg_part, float* __restrict__ db_part,
                      const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                      const float* dres, int64_t n_rows, float eps) {
  constexpr int C = TPR * VPL * 4;
  constexpr int RPB = 256 / TPR;
  __shared__ float4 s_red[256];
  const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
  float4 gm[VPL], ag[VPL], ab[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    gm[v] = reinterpret_cast<const float4*>(gamma)[v * TPR + tx];
    ag[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[v] = ag[v];
  }
  const float inv_c = 1.0f / (float)C;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < n_rows; base += (int64_t)gridDim.x * RPB) {
    const int64_t r = base + ty;
    const bool live = r < n_rows;
    float4 a[VPL], d[VPL];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      a[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      d[v] = a[v];
      if (live) {
        a[v] = reinterpret_cast<const float4*>(x + r * C)[v * TPR + tx];
        d[v] = reinterpret_cast<const float4*>(dy + r * C)[v * TPR + tx];
      }
      sum += (a[v].x + a[v].y) + (a[v].z + a[v].w);
    }
    const float mean = hfl_group_sum<TPR>(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      a[v].x -= mean; a[v].y -= mean; a[v].z -= mean; a[v].w -= mean;
      sq += (a[v].x * a[v].x + a[v].y * a[v].y) + (a[v].z * a[v].z + a[v].w * a[v].w);
    }
    const float rstd = 1.0f / sqrtf(hfl_group_sum<TPR>(sq) * inv_c + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      a[v].x *= rstd; a[v].y *= rstd; a[v].z *= rstd; a[v].w *= rstd;             // xhat
      ab[v].x += d[v].x; ab[v].y += d[v].y; ab[v].z += d[v].z; ab[v].w += d[v].w;
      ag[v].x = fmaf(d[v].x, a[v].x, ag[v].x); ag[v].y = fmaf(d[v].y, a[v].y, ag[v].y);
      ag[v].z = fmaf(d[v].z, a[v].z, ag[v].z); ag[v].w = fmaf(d[v].w, a[v].w, ag[v].w);
      d[v].x *= gm[v].x; d[v].y *= gm[v].y; d[v].z *= gm[v].z; d[v].w *= gm[v].w;   // g
      s1 += (d[v].x + d[v].y) + (d[v].z + d[v].w);
      s2 += (d[v].x * a[v].x + d[v].y * a[v].y) + (d[v].z * a[v].z + d[v].w * a[v].w);
    }
    s1 = hfl_group_sum<TPR>(s1) * inv_c;
    s2 = hfl_group_sum<TPR>(s2) * inv_c;
    if (live) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        float4 o;
        o.x = rstd * (d[v].x - s1 - a[v].x * s2);
        o.y = rstd * (d[v].y - s1 - a[v].y * s2);
        o.z = rstd * (d[v].z - s1 - a[v].z * s2);
        o.w = rstd * (d[v].w - s1 - a[v].w * s2);
        if (dres != nullptr) {       // pre-norm residual branch y = x + f(LN(x)): the skip path's gradient joins here
          const float4 rs = reinterpret_cast<const float4*>(dres + r * C)[v * TPR + tx];
          o.x += rs.x; o.y += rs.y; o.z += rs.z; o.w += rs.w;
        }
        reinterpret_cast<float4*>(dx + r * C)[v * TPR + tx] = o;
      }
    }
  }
  // reduce the RPB row slots of the workgroup (fixed order), one channel group at a time
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();
      s_red[threadIdx.x] = which == 0 ? ag[v] : ab[v];
      __syncthreads();
      if (ty == 0) {
        float4 t = s_red[tx];
        for (int j = 1; j < RPB; ++j) {
          const float4 u = s_red[j * TPR + tx];
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        float* dst = (which == 0 ? dg_part : db_part) + (int64_t)blockIdx.x * C;
        reinterpret_cast<float4*>(dst)[v * TPR + tx] = t;
      }
    }
  }
}

static int ln_bwd_blocks(int64_t n, int rpb) {
  const int64_t need = hfl_cdiv(n, rpb);
  const int64_t cap = (int64_t)hfl_num_cus() * 8;
  return (int)(need < cap ? (need < 1 ? 1 : need) : cap);
}

template <int TPR, int VPL>
static int launch_ln_bwd(float* dx, float* dgp, float* dbp, const float* dy, const float* x, const float* gamma,
                         const float* dres, int64_t n, float eps, hipStream_t s) {
  layer_norm_bwd_kernel<TPR, VPL><<<ln_bwd_blocks(n, 256 / TPR), 256, 0, s>>>(dx, dgp, dbp, dy, x, gamma, dres, n, eps);
  HFL_RETURN_LAST_ERROR();
}

// ---------------------------------------------------------- elementwise producers
// MODE 0: out_f32 = a + b + bias            (residual add with the fc2 bias folded in)
// MODE 1: out_bf16x3 = split3(gelu(a + bias))   (exact erf GELU, models/layers/octformer_layers.py:49,55)
// MODE 2: out_bf16x3 = split3(a)
template <int MODE>
__global__ void __launch_bounds__(256)
eltwise_kernel(void* __restrict__ out, const float* a, const float* b,
               const float* __restrict__ bias, int64_t n_rows, int C) {
  const int cv = C / 4;
  const int64_t total = n_rows * cv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cv;
    const int c4 = (int)(i % cv);
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (bias != nullptr) {
      const float4 bb = reinterpret_cast<const float4*>(bias)[c4];
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    if (MODE == 0) {
      const float4 w = reinterpret_cast<const float4*>(b)[i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      reinterpret_cast<float4*>(out)[i] = v;
    } else {
      if (MODE == 1) {
        const float k = 0.70710678118654752440f;
        v.x = 0.5f * v.x * (1.0f + erff(v.x * k));
        v.y = 0.5f * v.y * (1.0f + erff(v.y * k));
        v.z = 0.5f * v.z * (1.0f + erff(v.z * k));
        v.w = 0.5f * v.w * (1.0f + erff(v.w * k));
      }
      hfl_store_split3(reinterpret_cast<uint16_t*>(out) + r * 3 * C, C, c4, v);
    }
  }
}

template <int MODE>
static int launch_eltwise(void* out, const float* a, const float* b, const float* bias, int64_t n,
                          int64_t C, hipStream_t s) {
  if (n < 0 || C <= 0 || C % 4 != 0) return HFL_EINVAL;
  if (n == 0) return HFL_OK;
  const int64_t need = hfl_cdiv(n * (C / 4), 256);
  const int64_t cap = (int64_t)hfl_stream_cus(s) * 16;
  eltwise_kernel<MODE><<<(int)(need < cap ? need : cap), 256, 0, s>>>(out, a, b, bias, n, (int)C);
  HFL_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

int hfl_octree_gather(float* out, const float* data, const int32_t* neigh, int64_t n_out, int kngh,
                      int64_t channels, hfl_stream_t stream) {
  if (n_out < 0 || kngh <= 0 || channels <= 0) return HFL_EINVAL;
  if (n_out == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int vec = (channels % 4 == 0) ? 4 : 1;
  const int64_t total = n_out * kngh * (channels / vec);
  const int64_t need = hfl_cdiv(total, 256);
  const int64_t cap = (int64_t)hfl_stream_cus(s) * 16;
  const int blocks = (int)(need < cap ? need : cap);
  if (vec == 4)
    gather_kernel<4><<<blocks, 256, 0, s>>>(out, data, neigh, n_out, kngh, (int)channels);
  else
    gather_kernel<1><<<blocks, 256, 0, s>>>(out, data, neigh, n_out, kngh, (int)channels);
  HFL_RETURN_LAST_ERROR();
}

int hfl_relay_token_init(float* rt, const float* x, const uint32_t* tok_meta, int64_t n_tokens,
                         int32_t n_windows, int32_t patch_size, int64_t channels,
                         hfl_stream_t stream) {
  if (n_windows < 0 || patch_size <= 0 || channels <= 0 || channels % 4 != 0 || channels > 4096)
    return HFL_EINVAL;
  if (n_windows == 0) return HFL_OK;
  int threads = (int)(channels / 4);
  threads = ((threads + 63) / 64) * 64;
  relay_init_kernel<<<n_windows, threads, 0, static_cast<hipStream_t>(stream)>>>(
      rt, x, tok_meta, n_tokens, n_windows, patch_size, (int)channels);
  HFL_RETURN_LAST_ERROR();
}

int hfl_window_stats(float* stats, const uint32_t* tok_meta, int64_t n_tokens, int32_t n_windows,
                     int32_t patch_size, int depth, hfl_stream_t stream) {
  if (n_windows < 0 || patch_size <= 0 || depth < 1) return HFL_EINVAL;
  if (n_windows == 0) return HFL_OK;
  const float scale = 1.0f / (float)(1 << (depth - 1));   // 2^(1-depth), misc/utils.py:301
  window_stats_kernel<<<(unsigned)hfl_cdiv(n_windows, 128), 128, 0, static_cast<hipStream_t>(stream)>>>(
      stats, tok_meta, n_tokens, n_windows, patch_size, scale);
  HFL_RETURN_LAST_ERROR();
}

int hfl_segment_softmax(float* scores, const int64_t* row_off, int batch, int n_queries, float scale,
                        hfl_stream_t stream) {
  if (batch <= 0 || n_queries <= 0) return HFL_EINVAL;
  dim3 grid((unsigned)batch, (unsigned)hfl_cdiv(n_queries, 64));
  segment_softmax_kernel<<<grid, 1024, 0, static_cast<hipStream_t>(stream)>>>(scores, row_off,
                                                                              n_queries, scale);
  HFL_RETURN_LAST_ERROR();
}

int hfl_layer_norm(float* out, const float* x, const float* gamma, const float* beta, int64_t n_rows,
                   int64_t channels, float eps, hfl_stream_t stream) {
  if (n_rows < 0) return HFL_EINVAL;
  return dispatch_ln(out, nullptr, x, nullptr, nullptr, gamma, beta, n_rows, channels, eps, 0,
                     static_cast<hipStream_t>(stream));
}

int hfl_add_layer_norm(float* x_out, float* h_out, const float* x, const float* y, const float* bias,
                       const float* gamma, const float* beta, int64_t n_rows, int64_t channels,
                       float eps, hfl_stream_t stream) {
  if (n_rows < 0 || y == nullptr || x_out == nullptr) return HFL_EINVAL;
  return dispatch_ln(h_out, x_out, x, y, bias, gamma, beta, n_rows, channels, eps, 0,
                     static_cast<hipStream_t>(stream));
}

int hfl_layer_norm_split3(uint16_t* out, const float* x, const float* gamma, const float* beta,
                          int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream) {
  if (n_rows < 0) return HFL_EINVAL;
  return dispatch_ln(reinterpret_cast<float*>(out), nullptr, x, nullptr, nullptr, gamma, beta, n_rows,
                     channels, eps, 1, static_cast<hipStream_t>(stream));
}

int hfl_layer_norm_split2(uint16_t* out, const float* x, const float* gamma, const float* beta,
                          int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream) {
  if (n_rows < 0 || channels % 32 != 0) return HFL_EINVAL;
  return dispatch_ln(reinterpret_cast<float*>(out), nullptr, x, nullptr, nullptr, gamma, beta, n_rows,
                     channels, eps, 2, static_cast<hipStream_t>(stream));
}

/* ReLU(LayerNorm(x)): the norm -> ReLU pair behind every stem convolution (models/layers/octformer_layers.py:80-98) in one
 * pass, as f32 rows (out_f32) or as the split2 operand of the next convolution's GEMM (out_split2); exactly one is set. */
int hfl_layer_norm_relu(float* out_f32, uint16_t* out_split2, const float* x, const float* gamma, const float* beta,
                        int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream) {
  if (n_rows < 0 || (out_f32 == nullptr) == (out_split2 == nullptr)) return HFL_EINVAL;
  if (out_split2 != nullptr && channels % 32 != 0) return HFL_EINVAL;
  return dispatch_ln(out_f32 != nullptr ? out_f32 : reinterpret_cast<float*>(out_split2), nullptr, x, nullptr, nullptr, gamma,
                     beta, n_rows, channels, eps, (out_split2 != nullptr ? 2 : 0) | 0x10, static_cast<hipStream_t>(stream));
}

int hfl_add_layer_norm_split3(float* x_out, uint16_t* h_out, const float* x, const float* y,
                              const float* bias, const float* gamma, const float* beta, int64_t n_rows,
                              int64_t channels, float eps, hfl_stream_t stream) {
  if (n_rows < 0 || y == nullptr || x_out == nullptr) return HFL_EINVAL;
  return dispatch_ln(reinterpret_cast<float*>(h_out), x_out, x, y, bias, gamma, beta, n_rows, channels,
                     eps, 1, static_cast<hipStream_t>(stream));
}

}  // extern "C" (reopened below)

namespace {
// out[a][c] = sum over the partial rows of array a, fixed order: 4 float4 columns x 64 row groups per workgroup, tree in LDS
__global__ void __launch_bounds__(256)
column_sum_kernel(float* __restrict__ out0, float* __restrict__ out1, const float* __restrict__ part0,
                  const float* __restrict__ part1, int n_rows, int C) {
  __shared__ float4 red[64][4];
  const float* part = blockIdx.y == 0 ? part0 : part1;
  float* out = blockIdx.y == 0 ? out0 : out1;
  const int col = threadIdx.x & 3, rg = threadIdx.x >> 2;
  const int c4 = blockIdx.x * 4 + col;                       // float4 column
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 * 4 < C)
    for (int r = rg; r < n_rows; r += 64) {
      const float4 v = reinterpret_cast<const float4*>(part + (int64_t)r * C)[c4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  red[rg][col] = a;
  __syncthreads();
  for (int half = 32; half > 0; half >>= 1) {
    if (rg < half) {
      const float4 b = red[rg + half][col];
      float4 m = red[rg][col];
      m.x += b.x; m.y += b.y; m.z += b.z; m.w += b.w;
      red[rg][col] = m;
    }
    __syncthreads();
  }
  if (rg == 0 && c4 * 4 < C) reinterpret_cast<float4*>(out)[c4] = red[0][col];
}
}  // namespace

extern "C" {

/* dgamma / dbeta (C) from the partial rows hfl_layer_norm_bwd wrote (fixed summation order). */
int hfl_layer_norm_bwd_finalize(float* dgamma, float* dbeta, const float* dgamma_partial, const float* dbeta_partial,
                                int n_blocks, int64_t channels, hfl_stream_t stream) {
  if (n_blocks <= 0 || channels <= 0 || channels % 4 != 0) return HFL_EINVAL;
  dim3 grid((unsigned)hfl_cdiv(channels, 16), 2);
  column_sum_kernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(dgamma, dbeta, dgamma_partial, dbeta_partial,
                                                                        n_blocks, (int)channels);
  HFL_RETURN_LAST_ERROR();
}

/* Number of (blocks, C) partial-sum rows hfl_layer_norm_bwd writes for this problem. */
int hfl_layer_norm_bwd_blocks(int64_t n_rows, int64_t channels) {
  switch (channels) {
    case 16: return ln_bwd_blocks(n_rows, 256 / 4);
    case 32: return ln_bwd_blocks(n_rows, 256 / 8);
    case 64: return ln_bwd_blocks(n_rows, 256 / 16);
    case 128: return ln_bwd_blocks(n_rows, 256 / 32);
    case 256: case 512: case 1024: return ln_bwd_blocks(n_rows, 256 / 64);
    default: return 0;
  }
}

int hfl_layer_norm_bwd_add(float* dx, float* dgamma_partial, float* dbeta_partial, const float* dy, const float* x,
                           const float* gamma, const float* dres, int64_t n_rows, int64_t channels, float eps,
                           hfl_stream_t stream) {
  if (n_rows <= 0) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (channels) {
    case 16:   return launch_ln_bwd<4, 1>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 32:   return launch_ln_bwd<8, 1>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 64:   return launch_ln_bwd<16, 1>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 128:  return launch_ln_bwd<32, 1>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 256:  return launch_ln_bwd<64, 1>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 512:  return launch_ln_bwd<64, 2>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    case 1024: return launch_ln_bwd<64, 4>(dx, dgamma_partial, dbeta_partial, dy, x, gamma, dres, n_rows, eps, s);
    default:   return HFL_EINVAL;
  }
}

int hfl_layer_norm_bwd(float* dx, float* dgamma_partial, float* dbeta_partial, const float* dy, const float* x,
                       const float* gamma, int64_t n_rows, int64_t channels, float eps, hfl_stream_t stream) {
  return hfl_layer_norm_bwd_add(dx, dgamma_partial, dbeta_partial, dy, x, gamma, nullptr, n_rows, channels, eps, stream);
}

int hfl_add_bias(float* out, const float* x, const float* y, const float* bias, int64_t n_rows,
                 int64_t channels, hfl_stream_t stream) {
  return launch_eltwise<0>(out, x, y, bias, n_rows, channels, static_cast<hipStream_t>(stream));
}

int hfl_bias_gelu_split3(uint16_t* out, const float* x, const float* bias, int64_t n_rows,
                         int64_t channels, hfl_stream_t stream) {
  return launch_eltwise<1>(out, x, nullptr, bias, n_rows, channels, static_cast<hipStream_t>(stream));
}

int hfl_split3(uint16_t* out, const float* x, int64_t n_rows, int64_t channels, hfl_stream_t stream) {
  return launch_eltwise<2>(out, x, nullptr, nullptr, n_rows, channels, static_cast<hipStream_t>(stream));
}

int hfl_octree_gather_bwd(float* ddata, const float* dcol, const int32_t* ineigh, int64_t n_rows,
                          int kngh, int64_t channels, hfl_stream_t stream) {
  if (n_rows < 0 || kngh <= 0 || channels <= 0) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t cap = (int64_t)hfl_num_cus() * 16;
  if (channels % 4 == 0) {
    const int64_t need = hfl_cdiv(n_rows * (channels / 4), 256);
    gather_bwd_kernel<<<(int)(need < cap ? need : cap), 256, 0, s>>>(ddata, dcol, ineigh, n_rows, kngh,
                                                                     (int)channels);
  } else {
    const int64_t need = hfl_cdiv(n_rows * channels, 256);
    gather_bwd_scalar_kernel<<<(int)(need < cap ? need : cap), 256, 0, s>>>(ddata, dcol, ineigh, n_rows,
                                                                            kngh, (int)channels);
  }
  HFL_RETURN_LAST_ERROR();
}

int hfl_relay_token_init_bwd(float* dx, const float* drt, const uint32_t* tok_meta, int64_t n_tokens,
                             int32_t n_windows, int32_t patch_size, int64_t channels,
                             hfl_stream_t stream) {
  if (n_windows < 0 || patch_size <= 0 || channels <= 0 || channels % 4 != 0 || channels > 4096)
    return HFL_EINVAL;
  if (n_windows == 0) return HFL_OK;
  int threads = (int)(channels / 4);
  threads = ((threads + 63) / 64) * 64;
  relay_init_bwd_kernel<<<n_windows, threads, 0, static_cast<hipStream_t>(stream)>>>(
      dx, drt, tok_meta, n_tokens, n_windows, patch_size, (int)channels);
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
