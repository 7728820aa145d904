// Octree depth-wise convolution for gfx950 (MI355X): forward / input-gradient,
// weight-gradient, inverse neighbour table, and the fused CPE
// (dwconv -> LayerNorm -> residual) used by the transformer blocks.
//
// Semantics: libs/dwconv/csrc/dwconv.cu:24-85 of the reference (3 CUDA kernels).
// This is not a translation of them: the reference maps one thread to one output
// float and re-reads the row's neighbour list per float.  Here a row is owned by
// C/4 lanes that move 16 B each (a 256-channel row is one 1 KiB wave access), the
// neighbour list is staged once per row through LDS, the 27xC tap weights live in
// LDS for the whole persistent block, and taps are issued in batches of independent
// 16-B gathers so a CU keeps tens of KiB in flight (HBM/L2-latency bound op).
#include "hfl_common.h"

namespace {

constexpr int kMaxTaps = 27;
constexpr int kMaxRowsPerBlock = 16;

struct RowGeom {
  int tpr;   // lanes per row = C/4
  int rpb;   // rows per block
};

static RowGeom row_geom(int64_t channels) {
  RowGeom g;
  g.tpr = (int)(channels / 4);
  g.rpb = 256 / g.tpr;
  if (g.rpb < 1) g.rpb = 1;
  if (g.rpb > kMaxRowsPerBlock) g.rpb = kMaxRowsPerBlock;
  return g;
}

// Workgroups of a persistent kernel per CU: a whole multiple of what is RESIDENT (registers and LDS, by the occupancy query), at
// most `want`.  With a grid that is not such a multiple the last, partial round costs a whole one: every workgroup owns an equal
// share of the rows (round 6: 8 per CU were launched where 5 fit -- 1.6 rounds).
static int persistent_per_cu(const void* kernel, int threads, size_t lds, int want, int* cache_nb, size_t* cache_lds) {
  if (*cache_lds != lds + 1) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds) != hipSuccess || nb < 1) nb = 0;
    *cache_nb = nb;
    *cache_lds = lds + 1;
  }
  const int nb = *cache_nb;
  if (nb < 1) return want;                         // (query failed: keep the old figure)
  if (nb >= want) return want;
  return (want / nb) * nb >= nb ? ((want / nb) * nb) : nb;
}

// ---------------------------------------------------------------- forward
// LDS: [K*C] weights | [rpb*K] row indices
template <typename IdxT, int KFIX>
__global__ void __launch_bounds__(256) dwconv_fwd_vec4(float* __restrict__ out, const float* __restrict__ data,
                                const float* __restrict__ weight, const IdxT* __restrict__ neigh,
                                int64_t n_out, int C, int kngh, int tpr, int rpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int K = KFIX > 0 ? KFIX : kngh;
  float4* s_w = reinterpret_cast<float4*>(smem);
  IdxT* s_idx = reinterpret_cast<IdxT*>(smem + (size_t)K * C * sizeof(float));
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  const int c4n = C / 4;
  for (int i = threadIdx.x; i < K * c4n; i += blockDim.x)
    s_w[i] = reinterpret_cast<const float4*>(weight)[i];

  for (int64_t base = (int64_t)blockIdx.x * rpb; base < n_out; base += (int64_t)gridDim.x * rpb) {
    const int64_t h = base + ty;
    const bool live = h < n_out;
    __syncthreads();   // previous iteration's readers are done (and weights are staged)
    if (live)
      for (int k = tx; k < K; k += tpr) s_idx[ty * K + k] = neigh[h * K + k];
    __syncthreads();
    if (!live) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int B = 9;   // taps in flight per lane
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += B) {
      float4 v[B];
      bool ok[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int k = k0 + j;
        int64_t ni = (k < K) ? (int64_t)s_idx[ty * K + k] : -1;
        ok[j] = ni >= 0;
        if (!ok[j]) ni = 0;
        v[j] = reinterpret_cast<const float4*>(data + ni * C)[tx];
      }
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int k = k0 + j;
        if (k < K && ok[j]) acc = hfl_fma4(s_w[k * c4n + tx], v[j], acc);
      }
    }
    reinterpret_cast<float4*>(out + h * C)[tx] = acc;
  }
}

// out[h] = sum over the row's live slots of part[slot[h, k]] [+ bias]: the second half of an octree convolution over its live
// (row, tap) pairs (model.OctreeConv._forward_live_taps: every output row adds its own partial products, in tap order, no
// atomics).  Rounds 1-5 ran it as the depth-wise convolution above with unit weights -- 27 x C ones staged in LDS and a
// multiply per element -- and added the convolution's bias in a separate pass; the sums are bitwise the same.
template <int B>
__global__ void __launch_bounds__(256) slot_sum_kernel(float* __restrict__ out, const float* __restrict__ part,
                                                        const int32_t* __restrict__ slot, const float* __restrict__ bias,
                                                        int64_t n_out, int C, int K, int tpr, int rpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* s_idx = reinterpret_cast<int32_t*>(smem);
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  const float4 bv = bias != nullptr ? reinterpret_cast<const float4*>(bias)[tx] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = (int64_t)blockIdx.x * rpb; base < n_out; base += (int64_t)gridDim.x * rpb) {
    const int64_t h = base + ty;
    const bool live = h < n_out;
    __syncthreads();
    if (live)
      for (int k = tx; k < K; k += tpr) s_idx[ty * K + k] = slot[h * K + k];
    __syncthreads();
    if (!live) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += B) {
      float4 v[B];
      bool ok[B];
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const int k = k0 + j;
        int64_t ni = (k < K) ? (int64_t)s_idx[ty * K + k] : -1;
        ok[j] = ni >= 0;
        if (!ok[j]) ni = 0;
        v[j] = reinterpret_cast<const float4*>(part + ni * C)[tx];
      }
#pragma unroll
      for (int j = 0; j < B; ++j)
        if (ok[j]) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    if (bias != nullptr) { acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w; }
    reinterpret_cast<float4*>(out + h * C)[tx] = acc;
  }
}

// any channel count (scalar lanes), used when C % 4 != 0 or C > 1024
template <typename IdxT>
__global__ void dwconv_fwd_scalar(float* __restrict__ out, const float* __restrict__ data,
                                  const float* __restrict__ weight, const IdxT* __restrict__ neigh,
                                  int64_t n_out, int64_t C, int K) {
  const int64_t total = n_out * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t h = i / C, c = i % C;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int64_t ni = (int64_t)neigh[h * K + k];
      if (ni >= 0) acc = fmaf(weight[k * C + c], data[ni * C + c], acc);
    }
    out[i] = acc;
  }
}

template <typename IdxT>
static int launch_fwd(float* out, const float* data, const float* weight, const IdxT* neigh,
                      int64_t n_out, int64_t C, int K, hipStream_t s) {
  if (n_out == 0) return HFL_OK;
  if (C % 4 == 0 && C <= 1024 && K <= kMaxTaps) {
    RowGeom g = row_geom(C);
    const size_t lds = (size_t)K * C * sizeof(float) + (size_t)g.rpb * K * sizeof(IdxT);
    const int64_t need = hfl_cdiv(n_out, g.rpb);
    static int nb27 = 0, nb0 = 0;
    static size_t l27 = 0, l0 = 0;
    const int per_cu = K == 27 ? persistent_per_cu(reinterpret_cast<const void*>(dwconv_fwd_vec4<IdxT, 27>), g.tpr * g.rpb, lds, 8,
                                                   &nb27, &l27)
                               : persistent_per_cu(reinterpret_cast<const void*>(dwconv_fwd_vec4<IdxT, 0>), g.tpr * g.rpb, lds, 8,
                                                   &nb0, &l0);
    const int64_t cap = (int64_t)hfl_stream_cus(s) * per_cu;
    const int blocks = (int)(need < cap ? need : cap);
    if (K == 27)
      dwconv_fwd_vec4<IdxT, 27><<<blocks, g.tpr * g.rpb, lds, s>>>(out, data, weight, neigh, n_out,
                                                                   (int)C, K, g.tpr, g.rpb);
    else
      dwconv_fwd_vec4<IdxT, 0><<<blocks, g.tpr * g.rpb, lds, s>>>(out, data, weight, neigh, n_out,
                                                                  (int)C, K, g.tpr, g.rpb);
  } else {
    const int64_t total = n_out * C;
    const int64_t need = hfl_cdiv(total, 256);
    const int blocks = (int)(need < 2048 ? need : 2048);
    dwconv_fwd_scalar<IdxT><<<blocks, 256, 0, s>>>(out, data, weight, neigh, n_out, C, K);
  }
  HFL_RETURN_LAST_ERROR();
}

// ---------------------------------------------------------- weight gradient
// partial[block, k, c] = sum over the rows the workgroup visited (per lane-row in visiting order, then over its lane-rows).
// The 27 gathers of a row are issued nine at a time WITHOUT a branch around each (a missing neighbour reads row 0 and is
// dropped by a select): with `if (ni >= 0) load; fma` every gather waited for the one before it and the launch ran at the
// latency of 27 dependent L2 round trips per row (187 us at depth 4 against the forward's 65).  Rows are dealt like the
// CPE's: XCD x = blockIdx & 7 walks its own contiguous eighth of the z-ordered rows, so that a row's neighbours are
// re-used through one L2.
// B taps of one row: gathers first, then the products (acc indices are compile-time: the 27 accumulators stay in VGPRs)
template <typename IdxT, int B, int K0>
__device__ __forceinline__ void wgrad_taps(float4 (&acc)[kMaxTaps], const float4 g, const float* __restrict__ data,
                                           const IdxT* idx, int K, int C, int tx) {
  constexpr int NB = K0 + B <= kMaxTaps ? B : kMaxTaps - K0;
  float4 d[NB];
  bool ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int64_t ni = K0 + j < K ? (int64_t)idx[K0 + j] : (int64_t)-1;
    ok[j] = ni >= 0;
    d[j] = reinterpret_cast<const float4*>(data + (ok[j] ? ni : (int64_t)0) * C)[tx];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const float4 a = hfl_fma4(d[j], g, acc[K0 + j]);          // (component selects: `ok ? a : acc` on the structs is a
    acc[K0 + j].x = ok[j] ? a.x : acc[K0 + j].x;              //  select of ADDRESSES and sends the array to scratch)
    acc[K0 + j].y = ok[j] ? a.y : acc[K0 + j].y;
    acc[K0 + j].z = ok[j] ? a.z : acc[K0 + j].z;
    acc[K0 + j].w = ok[j] ? a.w : acc[K0 + j].w;
  }
}

template <typename IdxT, int B, int K0 = 0>
__device__ __forceinline__ void wgrad_row(float4 (&acc)[kMaxTaps], const float4 g, const float* __restrict__ data,
                                          const IdxT* idx, int K, int C, int tx) {
  if constexpr (K0 < kMaxTaps) {
    if (K0 < K) wgrad_taps<IdxT, B, K0>(acc, g, data, idx, K, C, tx);
    wgrad_row<IdxT, B, K0 + B>(acc, g, data, idx, K, C, tx);
  }
}

template <typename IdxT, int B>
__global__ void __launch_bounds__(256) dwconv_wgrad_partial(float* __restrict__ partial, const float* __restrict__ grad,
                                     const float* __restrict__ data, const IdxT* __restrict__ neigh,
                                     int64_t n_rows, int C, int K, int tpr, int rpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  IdxT* s_idx = reinterpret_cast<IdxT*>(smem);
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  float4 acc[kMaxTaps];
#pragma unroll
  for (int k = 0; k < kMaxTaps; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t groups = (n_rows + rpb - 1) / rpb;
  const int64_t per_xcd = (groups + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int64_t xg0 = (int64_t)xcd * per_xcd;
  const int64_t xg1 = xg0 + per_xcd < groups ? xg0 + per_xcd : groups;
  const int nb = ((int)gridDim.x - xcd + 7) >> 3;                  // workgroups of this launch on the XCD
  for (int64_t grp = xg0 + (blockIdx.x >> 3); grp < xg1; grp += nb) {
    const int64_t h = grp * rpb + ty;
    const bool live = h < n_rows;
    __syncthreads();
    if (live)
      for (int k = tx; k < K; k += tpr) s_idx[ty * K + k] = neigh[h * K + k];
    __syncthreads();
    if (!live) continue;
    const float4 g = reinterpret_cast<const float4*>(grad + h * C)[tx];
    wgrad_row<IdxT, B>(acc, g, data, s_idx + ty * K, K, C, tx);
  }
  // the workgroup's rpb lane-rows are added in a fixed order (row 0 takes 1, 2, ...), nine taps at a time through LDS: ONE
  // partial slab per workgroup for the second kernel instead of rpb (85 -> 21 MB written and re-read per depth-4 launch)
  float4* s_red = reinterpret_cast<float4*>(smem + (((size_t)rpb * K * sizeof(IdxT) + 15) & ~(size_t)15));
#pragma unroll
  for (int k0 = 0; k0 < kMaxTaps; k0 += 9) {
    if (k0 < K && rpb > 1) {
      __syncthreads();
      if (ty > 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j)
          if (k0 + j < K) s_red[((ty - 1) * 9 + j) * tpr + tx] = acc[k0 + j];
      }
      __syncthreads();
      if (ty == 0) {
        for (int t = 1; t < rpb; ++t) {
#pragma unroll
          for (int j = 0; j < 9; ++j)
            if (k0 + j < K) {
              const float4 v = s_red[((t - 1) * 9 + j) * tpr + tx];
              acc[k0 + j].x += v.x; acc[k0 + j].y += v.y; acc[k0 + j].z += v.z; acc[k0 + j].w += v.w;
            }
        }
      }
    }
  }
  if (ty == 0) {
    float4* dst = reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * K * C);
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k)
      if (k < K) dst[k * (C / 4) + tx] = acc[k];
  }
}

// out[i] = sum_p partial[p, i] in a fixed order (bitwise reproducible)
__global__ void __launch_bounds__(1024) dwconv_wgrad_reduce(float* __restrict__ out, const float* __restrict__ partial,
                                                            int n_partial, int64_t kc) {
  __shared__ float s[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;                  // 16 groups walk the partial slabs
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (i < kc)
    for (int p = grp; p < n_partial; p += 16) acc += partial[(int64_t)p * kc + i];
  s[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && i < kc) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += s[g][lane];
    out[i] = v;
  }
}

template <typename IdxT>
__global__ void dwconv_wgrad_scalar(float* __restrict__ out, const float* __restrict__ grad,
                                    const float* __restrict__ data, const IdxT* __restrict__ neigh,
                                    int64_t n_rows, int64_t C, int K) {
  // one thread per (k,c): slow generic path for odd channel counts
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)K * C) return;
  const int k = (int)(i / C);
  const int64_t c = i % C;
  float acc = 0.f;
  for (int64_t h = 0; h < n_rows; ++h) {
    const int64_t ni = (int64_t)neigh[h * K + k];
    if (ni >= 0) acc = fmaf(data[ni * C + c], grad[h * C + c], acc);
  }
  out[i] = acc;
}

int g_wgrad_batch = 9;      // gathers in flight per lane (probe: hfl_internal_set_wgrad_batch)

static int wgrad_blocks(int64_t n_rows, int rpb) {
  const int64_t need = hfl_cdiv(n_rows, (int64_t)rpb * 8);
  // 3 workgroups per CU: the gathers of a row are only hidden by other waves (one workgroup per CU ran at a quarter of
  // the forward kernel's rate); the price is 3x the partial slabs for the fixed-order reduce
  const int64_t cap = 3 * (int64_t)hfl_num_cus();
  int64_t b = need < cap ? need : cap;
  if (b < 1) b = 1;
  return (int)((b + 7) & ~(int64_t)7);           // every XCD's eighth of the rows needs a workgroup (the kernel's row map)
}

template <typename IdxT>
static int launch_wgrad(float* out, const float* grad, const float* data, const IdxT* neigh,
                        int64_t n_rows, int64_t C, int K, void* workspace, hipStream_t s) {
  if (C % 4 == 0 && C <= 1024 && K <= kMaxTaps) {
    RowGeom g = row_geom(C);
    const int blocks = wgrad_blocks(n_rows, g.rpb);
    const size_t lds = (((size_t)g.rpb * K * sizeof(IdxT) + 15) & ~(size_t)15) +
                       (size_t)(g.rpb - 1) * 9 * C * sizeof(float);          // row tables | the row reduction's nine-tap block
    float* partial = static_cast<float*>(workspace);
    if (g_wgrad_batch == 6)
      dwconv_wgrad_partial<IdxT, 6><<<blocks, g.tpr * g.rpb, lds, s>>>(partial, grad, data, neigh, n_rows, (int)C, K, g.tpr,
                                                                       g.rpb);
    else if (g_wgrad_batch == 3)
      dwconv_wgrad_partial<IdxT, 3><<<blocks, g.tpr * g.rpb, lds, s>>>(partial, grad, data, neigh, n_rows, (int)C, K, g.tpr,
                                                                       g.rpb);
    else
      dwconv_wgrad_partial<IdxT, 9><<<blocks, g.tpr * g.rpb, lds, s>>>(partial, grad, data, neigh, n_rows, (int)C, K, g.tpr,
                                                                       g.rpb);
    const int64_t kc = (int64_t)K * C;
    dwconv_wgrad_reduce<<<(int)hfl_cdiv(kc, 64), 1024, 0, s>>>(out, partial, blocks, kc);
  } else {
    const int64_t kc = (int64_t)K * C;
    dwconv_wgrad_scalar<IdxT><<<(int)hfl_cdiv(kc, 256), 256, 0, s>>>(out, grad, data, neigh, n_rows, C, K);
  }
  HFL_RETURN_LAST_ERROR();
}

// ------------------------------------------------------------ inverse table
template <typename IdxT>
__global__ void inverse_neigh_kernel(IdxT* __restrict__ ineigh, const IdxT* __restrict__ neigh,
                                     int64_t n_rows, int K) {
  const int64_t total = n_rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t h = i / K;
    const int k = (int)(i % K);
    const int64_t j = (int64_t)neigh[i];
    if (j >= 0) ineigh[j * K + k] = (IdxT)h;   // injective per column: no atomics needed
  }
}

// ------------------------------------------------------------------ fused CPE
// TPR lanes own one row (C = 4*TPR); LayerNorm statistics by butterfly over TPR lanes.
// NORM = false: the convolution alone, out = dwconv(x) [+ add] (the data gradient of the training path: x = the incoming
// gradient, neigh = the inverse table, add = the skip connection's gradient).
template <int TPR, bool NORM>
__global__ void __launch_bounds__(256) cpe_fwd_kernel(float* __restrict__ out, float* __restrict__ conv_out,
                               const float* __restrict__ x, const float* __restrict__ add,
                               const float* __restrict__ weight, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const int32_t* __restrict__ neigh,
                               int64_t n_rows, int K, float eps, int residual, int chunk_rows) {
  constexpr int C = TPR * 4;
  constexpr int RPB = 256 / TPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* s_w = reinterpret_cast<float4*>(smem);
  int32_t* s_idx = reinterpret_cast<int32_t*>(smem + (size_t)K * C * sizeof(float));
  unsigned char* s_tap = reinterpret_cast<unsigned char*>(s_idx + (256 / TPR) * K);
  const int tx = threadIdx.x % TPR, ty = threadIdx.x / TPR;
  for (int i = threadIdx.x; i < K * TPR; i += blockDim.x)
    s_w[i] = reinterpret_cast<const float4*>(weight)[i];
  const float4 gm = NORM ? reinterpret_cast<const float4*>(gamma)[tx] : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 bt = NORM ? reinterpret_cast<const float4*>(beta)[tx] : make_float4(0.f, 0.f, 0.f, 0.f);

  // iteration space: `it` enumerates groups of RPB rows.  chunk_rows == 0: group it of block b is
  // b + it*gridDim (interleaved).  chunk_rows > 0: a block owns chunk_rows consecutive rows at a time
  // (z-order neighbours -> its gathers revisit lines still in L1/L2), chunks dealt round-robin.
  // chunk_rows < 0 (default): XCD-contiguous.  Workgroups b and b + 8 share an XCD (round-robin dispatch), so XCD
  // x = b & 7 walks its own contiguous eighth of the z-ordered rows with its gridDim / 8 workgroups interleaved inside it:
  // a row's 27 neighbours are z-order neighbours of the same cloud, so the lines a gather touches are (re)used through ONE
  // L2 instead of being fetched into all eight (the interleaved map: L2 hit 47 %, 3.7x the algorithmic fabric traffic).
  const int64_t groups = (n_rows + RPB - 1) / RPB;
  const int64_t gpc = chunk_rows > 0 ? (chunk_rows + RPB - 1) / RPB : 1;     // groups per chunk
  const int64_t per_xcd = (groups + 7) >> 3;
  const int64_t xg0 = (int64_t)(blockIdx.x & 7) * per_xcd;
  const int64_t xg1 = xg0 + per_xcd < groups ? xg0 + per_xcd : groups;
  for (int64_t it = 0;; ++it) {
    int64_t grp;
    if (chunk_rows < 0) {
      grp = xg0 + (blockIdx.x >> 3) + it * (gridDim.x >> 3);
      if (grp >= xg1) break;
    } else if (chunk_rows > 0) {
      const int64_t chunk = (it / gpc) * gridDim.x + blockIdx.x;
      grp = chunk * gpc + it % gpc;
      if (chunk * gpc >= groups) break;
    } else {
      grp = (int64_t)blockIdx.x + it * gridDim.x;
      if (grp >= groups) break;
    }
    const int64_t base = grp * RPB;
    const int64_t h = base + ty;
    const bool live = h < n_rows;
    __syncthreads();
    // Live taps of the row, compacted (tap, neighbour row): octree neighbourhoods are sparse (17.7 of 27 taps
    // at depth 4, 5.6 at depth 5) and the kernel is bound by the L1/TA request rate, so dead taps must not
    // cost a request.  TPR >= 27 lanes own the row: ballot + prefix popcount, order of k preserved.
    int n_live = 0;
    if (TPR >= kMaxTaps) {
      const int ni = (live && tx < K) ? neigh[h * K + tx] : -1;
      const unsigned long long bal = __ballot(ni >= 0);
      const int shift = (threadIdx.x & 63) - tx;                     // first lane of this row inside the wave
      const unsigned mask = (unsigned)((bal >> shift) & ((1ull << kMaxTaps) - 1ull));
      n_live = __popc(mask);
      if (ni >= 0) {
        const int rank = __popc(mask & ((1u << tx) - 1u));
        s_idx[ty * K + rank] = ni;
        s_tap[ty * K + rank] = (unsigned char)tx;
      }
    } else if (live) {
      for (int k = tx; k < K; k += TPR) s_idx[ty * K + k] = neigh[h * K + k];
    }
    __syncthreads();
    // all lanes of a wave stay in the loop body: the butterflies below need every lane
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int B = 9;
    if (TPR >= kMaxTaps) {
#pragma unroll 1
      for (int k0 = 0; k0 < n_live; k0 += B) {
        float4 v[B];
        int e[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {                 // past the end: repeat the last live tap (same line, weight 0)
          e[j] = ty * K + min(k0 + j, n_live - 1);
          v[j] = reinterpret_cast<const float4*>(x + (int64_t)s_idx[e[j]] * C)[tx];
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
          float4 w = s_w[(int)s_tap[e[j]] * TPR + tx];
          if (k0 + j >= n_live) w = make_float4(0.f, 0.f, 0.f, 0.f);
          acc = hfl_fma4(w, v[j], acc);
        }
      }
    } else {
#pragma unroll 1
      for (int k0 = 0; k0 < K; k0 += B) {
        float4 v[B];
        bool ok[B];
#pragma unroll
        for (int j = 0; j < B; ++j) {
          const int k = k0 + j;
          int64_t ni = (live && k < K) ? (int64_t)s_idx[ty * K + k] : -1;
          ok[j] = ni >= 0;
          if (!ok[j]) ni = 0;
          v[j] = reinterpret_cast<const float4*>(x + ni * C)[tx];
        }
#pragma unroll
        for (int j = 0; j < B; ++j) {
          const int k = k0 + j;
          if (k < K && ok[j]) acc = hfl_fma4(s_w[k * TPR + tx], v[j], acc);
        }
      }
    }
    float4 y = acc;
    if (NORM) {
      const float inv_c = 1.0f / (float)C;
      const float mean = hfl_group_sum<TPR>((acc.x + acc.y) + (acc.z + acc.w)) * inv_c;
      const float4 d = make_float4(acc.x - mean, acc.y - mean, acc.z - mean, acc.w - mean);
      const float var = hfl_group_sum<TPR>((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * inv_c;
      const float rstd = 1.0f / sqrtf(var + eps);
      y = make_float4(fmaf(d.x * rstd, gm.x, bt.x), fmaf(d.y * rstd, gm.y, bt.y),
                      fmaf(d.z * rstd, gm.z, bt.z), fmaf(d.w * rstd, gm.w, bt.w));
    }
    if (live) {
      // (training: the convolution's output is the LayerNorm backward's input)
      if (NORM && conv_out != nullptr) reinterpret_cast<float4*>(conv_out + h * C)[tx] = acc;
      if (residual) {
        const float4 xv = reinterpret_cast<const float4*>(add + h * C)[tx];
        y.x += xv.x; y.y += xv.y; y.z += xv.z; y.w += xv.w;
      }
      reinterpret_cast<float4*>(out + h * C)[tx] = y;
    }
  }
}


static int g_cpe_chunk_rows = -1;  // < 0: XCD-contiguous (default); 0: rows interleaved over blocks; > 0: contiguous chunk per block

template <int TPR, bool NORM>
static int launch_cpe(float* out, float* conv_out, const float* x, const float* add, const float* w, const float* gamma,
                      const float* beta, const int32_t* neigh, int64_t n, int K, float eps,
                      int residual, hipStream_t s) {
  constexpr int RPB = 256 / TPR;
  const size_t lds = (size_t)K * TPR * 16 + (size_t)RPB * K * (sizeof(int32_t) + 1) + 16;
  const int64_t need = hfl_cdiv(n, RPB);
  static int nb = 0;
  static size_t nb_lds = 0;
  const int per_cu = persistent_per_cu(reinterpret_cast<const void*>(cpe_fwd_kernel<TPR, NORM>), 256, lds, 8, &nb, &nb_lds);
  const int64_t cap = (int64_t)hfl_stream_cus(s) * per_cu;
  int blocks = (int)(need < cap ? need : cap);
  if (g_cpe_chunk_rows < 0) blocks = (blocks + 7) & ~7;          // the XCD map deals whole groups of eight workgroups
  cpe_fwd_kernel<TPR, NORM><<<blocks, 256, lds, s>>>(out, conv_out, x, add, w, gamma, beta, neigh, n, K, eps, residual,
                                                     g_cpe_chunk_rows);
  HFL_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

int hfl_dwconv_forward_backward(float* out, const float* data, const float* weight,
                                const void* neigh, int idx64, int64_t n_out, int64_t channels,
                                int kngh, hfl_stream_t stream) {
  if (n_out < 0 || channels <= 0 || kngh <= 0) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (idx64)
    return launch_fwd<int64_t>(out, data, weight, static_cast<const int64_t*>(neigh), n_out,
                               channels, kngh, s);
  return launch_fwd<int32_t>(out, data, weight, static_cast<const int32_t*>(neigh), n_out,
                             channels, kngh, s);
}

int64_t hfl_dwconv_weight_backward_workspace(int64_t n_rows, int64_t channels, int kngh) {
  if (channels % 4 != 0 || channels > 1024 || kngh > kMaxTaps) return 16;
  RowGeom g = row_geom(channels);
  return (int64_t)wgrad_blocks(n_rows, g.rpb) * kngh * channels * (int64_t)sizeof(float);
}

int hfl_dwconv_weight_backward(float* out, const float* grad, const float* data,
                               const void* neigh, int idx64, int64_t n_rows, int64_t channels,
                               int kngh, void* workspace, hfl_stream_t stream) {
  if (n_rows < 0 || channels <= 0 || kngh <= 0) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (idx64)
    return launch_wgrad<int64_t>(out, grad, data, static_cast<const int64_t*>(neigh), n_rows,
                                 channels, kngh, workspace, s);
  return launch_wgrad<int32_t>(out, grad, data, static_cast<const int32_t*>(neigh), n_rows,
                               channels, kngh, workspace, s);
}

int hfl_inverse_neigh(void* ineigh, const void* neigh, int idx64, int64_t n_rows, int kngh,
                      hfl_stream_t stream) {
  if (n_rows < 0 || kngh <= 0) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t esz = idx64 ? 8 : 4;
  hipError_t e = hipMemsetAsync(ineigh, 0xFF, (size_t)n_rows * kngh * esz, s);   // all -1
  if (e != hipSuccess) return (int)e;
  if (n_rows == 0) return HFL_OK;
  const int64_t need = hfl_cdiv(n_rows * kngh, 256);
  const int blocks = (int)(need < 4096 ? need : 4096);
  if (idx64)
    inverse_neigh_kernel<int64_t><<<blocks, 256, 0, s>>>(static_cast<int64_t*>(ineigh),
                                                         static_cast<const int64_t*>(neigh), n_rows, kngh);
  else
    inverse_neigh_kernel<int32_t><<<blocks, 256, 0, s>>>(static_cast<int32_t*>(ineigh),
                                                         static_cast<const int32_t*>(neigh), n_rows, kngh);
  HFL_RETURN_LAST_ERROR();
}

int hfl_cpe_forward_save(float* out, float* conv_out, const float* x, const float* weight, const float* gamma,
                         const float* beta, const int32_t* neigh, int64_t n_rows, int64_t channels,
                         int kngh, float eps, int residual, hfl_stream_t stream) {
  if (n_rows < 0 || kngh <= 0 || kngh > kMaxTaps) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (channels) {
    case 256: return launch_cpe<64, true>(out, conv_out, x, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 128: return launch_cpe<32, true>(out, conv_out, x, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 64:  return launch_cpe<16, true>(out, conv_out, x, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 32:  return launch_cpe<8, true>(out, conv_out, x, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    default:  return HFL_EINVAL;
  }
}

int hfl_dwconv_add(float* out, const float* data, const float* weight, const int32_t* neigh, const float* add,
                   int64_t n_out, int64_t channels, int kngh, hfl_stream_t stream) {
  if (n_out < 0 || kngh <= 0 || kngh > kMaxTaps || out == data || out == nullptr) return HFL_EINVAL;
  if (n_out == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int res = add != nullptr ? 1 : 0;
  switch (channels) {
    case 256: return launch_cpe<64, false>(out, nullptr, data, add, weight, nullptr, nullptr, neigh, n_out, kngh, 0.f, res, s);
    case 128: return launch_cpe<32, false>(out, nullptr, data, add, weight, nullptr, nullptr, neigh, n_out, kngh, 0.f, res, s);
    case 64:  return launch_cpe<16, false>(out, nullptr, data, add, weight, nullptr, nullptr, neigh, n_out, kngh, 0.f, res, s);
    case 32:  return launch_cpe<8, false>(out, nullptr, data, add, weight, nullptr, nullptr, neigh, n_out, kngh, 0.f, res, s);
    default:  return HFL_EINVAL;
  }
}

int hfl_cpe_forward(float* out, const float* x, const float* weight, const float* gamma,
                    const float* beta, const int32_t* neigh, int64_t n_rows, int64_t channels,
                    int kngh, float eps, int residual, hfl_stream_t stream) {
  return hfl_cpe_forward_save(out, nullptr, x, weight, gamma, beta, neigh, n_rows, channels, kngh, eps, residual, stream);
}

int hfl_inverse_table(int32_t* inverse, int64_t n_src_rows, const int32_t* table, int64_t n_dst_rows,
                      int kngh, hfl_stream_t stream) {
  if (n_src_rows < 0 || n_dst_rows < 0 || kngh <= 0) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(inverse, 0xFF, (size_t)n_src_rows * kngh * sizeof(int32_t), s);
  if (e != hipSuccess) return (int)e;
  if (n_dst_rows == 0) return HFL_OK;
  const int64_t need = hfl_cdiv(n_dst_rows * kngh, 256);
  const int blocks = (int)(need < 4096 ? need : 4096);
  inverse_neigh_kernel<int32_t><<<blocks, 256, 0, s>>>(inverse, table, n_dst_rows, kngh);
  HFL_RETURN_LAST_ERROR();
}

/* see include/hotformerloc_hip.h */
int hfl_slot_sum(float* out, const float* part, const int32_t* slot, const float* bias, int64_t n_out, int64_t channels, int kngh,
                 hfl_stream_t stream) {
  if (n_out < 0 || channels <= 0 || channels % 4 != 0 || channels > 1024 || kngh <= 0 || kngh > kMaxTaps) return HFL_EINVAL;
  if (n_out == 0) return HFL_OK;
  if (out == nullptr || part == nullptr || slot == nullptr) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const RowGeom g = row_geom(channels);
  const size_t lds = (size_t)g.rpb * kngh * sizeof(int32_t);
  static int nb = 0;
  static size_t nb_lds = 0;
  const int per_cu = persistent_per_cu(reinterpret_cast<const void*>(slot_sum_kernel<9>), g.tpr * g.rpb, lds, 8, &nb, &nb_lds);
  const int64_t need = hfl_cdiv(n_out, g.rpb), cap = (int64_t)hfl_stream_cus(s) * per_cu;
  slot_sum_kernel<9><<<(int)(need < cap ? need : cap), g.tpr * g.rpb, lds, s>>>(out, part, slot, bias, n_out, (int)channels, kngh,
                                                                               g.tpr, g.rpb);
  HFL_RETURN_LAST_ERROR();
}

/* internal tuning hook used by hfl_set_variant("cpe_chunk_rows", n) */
void hfl_internal_set_cpe_chunk(int rows) { g_cpe_chunk_rows = rows; }
/* probe: gathers in flight per lane of the depth-wise weight gradient (9, 6 or 3) */
void hfl_internal_set_wgrad_batch(int taps) { g_wgrad_batch = taps; }
}  // extern "C"
