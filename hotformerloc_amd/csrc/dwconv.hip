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
    const int blocks = (int)(need < (int64_t)hfl_num_cus() * 8 ? need : (int64_t)hfl_num_cus() * 8);
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
// partial[(block*rpb+ty), k, c] = sum over the rows this lane-row visited.
template <typename IdxT>
__global__ void __launch_bounds__(256) dwconv_wgrad_partial(float* __restrict__ partial, const float* __restrict__ grad,
                                     const float* __restrict__ data, const IdxT* __restrict__ neigh,
                                     int64_t n_rows, int C, int K, int tpr, int rpb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  IdxT* s_idx = reinterpret_cast<IdxT*>(smem);
  const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
  float4 acc[kMaxTaps];
#pragma unroll
  for (int k = 0; k < kMaxTaps; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t base = (int64_t)blockIdx.x * rpb; base < n_rows; base += (int64_t)gridDim.x * rpb) {
    const int64_t h = base + ty;
    const bool live = h < n_rows;
    __syncthreads();
    if (live)
      for (int k = tx; k < K; k += tpr) s_idx[ty * K + k] = neigh[h * K + k];
    __syncthreads();
    if (!live) continue;
    const float4 g = reinterpret_cast<const float4*>(grad + h * C)[tx];
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) {
      if (k < K) {
        const int64_t ni = (int64_t)s_idx[ty * K + k];
        if (ni >= 0) {
          const float4 d = reinterpret_cast<const float4*>(data + ni * C)[tx];
          acc[k] = hfl_fma4(d, g, acc[k]);
        }
      }
    }
  }
  float4* dst = reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * rpb + ty) * K * C);
#pragma unroll
  for (int k = 0; k < kMaxTaps; ++k)
    if (k < K) dst[k * (C / 4) + tx] = acc[k];
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

static int wgrad_blocks(int64_t n_rows, int rpb) {
  const int64_t need = hfl_cdiv(n_rows, (int64_t)rpb * 8);
  // 3 workgroups per CU: the gathers of a row are only hidden by other waves (one workgroup per CU ran at a quarter of
  // the forward kernel's rate); the price is 3x the partial slabs for the fixed-order reduce
  const int64_t cap = 3 * (int64_t)hfl_num_cus();
  int64_t b = need < cap ? need : cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename IdxT>
static int launch_wgrad(float* out, const float* grad, const float* data, const IdxT* neigh,
                        int64_t n_rows, int64_t C, int K, void* workspace, hipStream_t s) {
  if (C % 4 == 0 && C <= 1024 && K <= kMaxTaps) {
    RowGeom g = row_geom(C);
    const int blocks = wgrad_blocks(n_rows, g.rpb);
    const size_t lds = (size_t)g.rpb * K * sizeof(IdxT);
    float* partial = static_cast<float*>(workspace);
    dwconv_wgrad_partial<IdxT><<<blocks, g.tpr * g.rpb, lds, s>>>(partial, grad, data, neigh, n_rows,
                                                                  (int)C, K, g.tpr, g.rpb);
    const int64_t kc = (int64_t)K * C;
    dwconv_wgrad_reduce<<<(int)hfl_cdiv(kc, 64), 1024, 0, s>>>(out, partial, blocks * g.rpb, kc);
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
template <int TPR>
__global__ void __launch_bounds__(256) cpe_fwd_kernel(float* __restrict__ out, const float* __restrict__ x,
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
  const float4 gm = reinterpret_cast<const float4*>(gamma)[tx];
  const float4 bt = reinterpret_cast<const float4*>(beta)[tx];

  // iteration space: `it` enumerates groups of RPB rows.  chunk_rows == 0: group it of block b is
  // b + it*gridDim (interleaved).  chunk_rows > 0: a block owns chunk_rows consecutive rows at a time
  // (z-order neighbours -> its gathers revisit lines still in L1/L2), chunks dealt round-robin.
  const int64_t groups = (n_rows + RPB - 1) / RPB;
  const int64_t gpc = chunk_rows > 0 ? (chunk_rows + RPB - 1) / RPB : 1;     // groups per chunk
  for (int64_t it = 0;; ++it) {
    int64_t grp;
    if (chunk_rows > 0) {
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
    const float inv_c = 1.0f / (float)C;
    const float mean = hfl_group_sum<TPR>((acc.x + acc.y) + (acc.z + acc.w)) * inv_c;
    const float4 d = make_float4(acc.x - mean, acc.y - mean, acc.z - mean, acc.w - mean);
    const float var = hfl_group_sum<TPR>((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * inv_c;
    const float rstd = 1.0f / sqrtf(var + eps);
    float4 y = make_float4(fmaf(d.x * rstd, gm.x, bt.x), fmaf(d.y * rstd, gm.y, bt.y),
                           fmaf(d.z * rstd, gm.z, bt.z), fmaf(d.w * rstd, gm.w, bt.w));
    if (live) {
      if (residual) {
        const float4 xv = reinterpret_cast<const float4*>(x + h * C)[tx];
        y.x += xv.x; y.y += xv.y; y.z += xv.z; y.w += xv.w;
      }
      reinterpret_cast<float4*>(out + h * C)[tx] = y;
    }
  }
}


// ------------------------------------------------------------------ fused CPE, LDS-staged gathers
// The kernel above pulls 27 neighbour rows per output row through L2 (17.7 live taps x 1 KiB per row
// at depth 4): it is bound by L2->L1 gather bandwidth, not HBM (PMC: 1.5x the algorithmic HBM bytes,
// profiles/r01_c_summary.md).  Neighbours of z-order-adjacent rows overlap heavily: the 64 x 27 taps of
// 64 consecutive rows touch ~186 distinct rows at depth 4 (6.1x fewer), ~116 at depth 5 (3.1x).
// Here a workgroup owns 64 consecutive rows:
//   1. de-duplicates their neighbour indices in an LDS hash table (atomicCAS insert, slot = arrival order);
//   2. per 32-channel slice: stages the distinct rows' slice in LDS once (coalesced 128-B segments), then
//      every (row, tap) reads its operand from LDS; taps without a neighbour use weight 0 (no divergence);
//   3. keeps the C conv outputs of a row in the registers of its 4 lanes, then LayerNorm + residual.
// Rows beyond the LDS capacity (never at the shipped configs: max 240 of 256 slots) are read from global.
constexpr int kCpeRows = 64;          // rows per workgroup pass
constexpr int kCpeSlice = 32;         // channels per LDS pass
constexpr int kCpeCap = 256;          // distinct neighbour rows kept in LDS
constexpr int kCpeHash = 1024;        // hash table entries (load factor <= 0.25)
constexpr int kCpeStride = kCpeSlice + 4;   // floats per staged row (+16 B pad against bank conflicts)

template <int C>
__global__ void __launch_bounds__(256) cpe_fwd_lds_kernel(float* __restrict__ out, const float* __restrict__ x,
                                   const float* __restrict__ weight, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const int32_t* __restrict__ neigh,
                                   int64_t n_rows, int K, float eps, int residual) {
  constexpr int NS = C / kCpeSlice;                 // slices
  __shared__ __attribute__((aligned(16))) float s_buf[kCpeCap * kCpeStride];
  __shared__ __attribute__((aligned(16))) float s_w[kMaxTaps * kCpeSlice];
  __shared__ int s_key[kCpeHash];
  __shared__ unsigned short s_hslot[kCpeHash];
  __shared__ unsigned short s_slot[kCpeRows * kMaxTaps];
  __shared__ int s_list[kCpeCap];
  __shared__ int s_count;

  const int tid = threadIdx.x;
  const int r = tid >> 2, q = tid & 3;              // row inside the pass, quarter of the slice (8 channels)
  const int64_t n_pass = (n_rows + kCpeRows - 1) / kCpeRows;
  const int entries = kCpeRows * K;

  for (int64_t pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const int64_t base = pass * kCpeRows;
    const int64_t live_entries = (n_rows - base < kCpeRows ? n_rows - base : kCpeRows) * K;
    __syncthreads();                                 // previous pass is done with every LDS array
    for (int i = tid; i < kCpeHash; i += 256) s_key[i] = -1;
    if (tid == 0) s_count = 0;
    __syncthreads();
    // ---- 1. distinct neighbour rows of this pass --------------------------------------------------
    int pos[(kCpeRows * kMaxTaps + 255) / 256];
#pragma unroll
    for (int i = 0; i < (kCpeRows * kMaxTaps + 255) / 256; ++i) {
      const int e = tid + 256 * i;
      pos[i] = -1;
      if (e < entries && e < live_entries) {
        const int idx = neigh[base * K + e];
        if (idx >= 0) {
          unsigned h = ((unsigned)idx * 2654435761u) >> 22;      // top 10 bits
          for (;;) {
            const int prev = atomicCAS(&s_key[h], -1, idx);
            if (prev == -1) {
              const int slot = atomicAdd(&s_count, 1);
              if (slot < kCpeCap) {
                s_list[slot] = idx;
                s_hslot[h] = (unsigned short)slot;
              } else {
                s_hslot[h] = 0xFFFEu;                            // no room: that row stays in global
              }
              break;
            }
            if (prev == idx) break;
            h = (h + 1) & (kCpeHash - 1);
          }
          pos[i] = (int)h;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (kCpeRows * kMaxTaps + 255) / 256; ++i) {
      const int e = tid + 256 * i;
      if (e < entries) s_slot[e] = pos[i] >= 0 ? s_hslot[pos[i]] : (unsigned short)0xFFFFu;
    }
    __syncthreads();
    unsigned slots[kMaxTaps];                        // this row's 27 operand slots, for every slice
#pragma unroll
    for (int k = 0; k < kMaxTaps; ++k) slots[k] = k < K ? s_slot[r * K + k] : 0xFFFFu;
    const int n_uniq = s_count < kCpeCap ? s_count : kCpeCap;
    const int64_t h_row = base + r;
    const bool live = h_row < n_rows;

    // ---- 2. per channel slice: stage once, then 27 taps from LDS ----------------------------------
    float4 acc[NS][2];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) {
      __syncthreads();                               // slot table ready / previous slice consumed
      {
        // all requests of the slice first (one round trip), LDS stores after
        float wv[(kMaxTaps * kCpeSlice + 255) / 256];
        float4 xv[kCpeCap / 32];
#pragma unroll
        for (int i = 0; i < (kMaxTaps * kCpeSlice + 255) / 256; ++i) {
          const int e = tid + 256 * i;
          wv[i] = e < K * kCpeSlice ? weight[(e / kCpeSlice) * C + sl * kCpeSlice + (e % kCpeSlice)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < kCpeCap / 32; ++i) {
          const int u = (tid >> 3) + 32 * i;
          const int src = u < n_uniq ? s_list[u] : 0;          // unconditional: keeps xv[] in registers
          xv[i] = *reinterpret_cast<const float4*>(x + (int64_t)src * C + sl * kCpeSlice + 4 * (tid & 7));
        }
#pragma unroll
        for (int i = 0; i < (kMaxTaps * kCpeSlice + 255) / 256; ++i) {
          const int e = tid + 256 * i;
          if (e < K * kCpeSlice) s_w[e] = wv[i];
        }
#pragma unroll
        for (int i = 0; i < kCpeCap / 32; ++i) {
          const int u = (tid >> 3) + 32 * i;
          if (u < n_uniq) *reinterpret_cast<float4*>(&s_buf[u * kCpeStride + 4 * (tid & 7)]) = xv[i];
        }
      }
      __syncthreads();
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
#pragma unroll
      for (int k = 0; k < kMaxTaps; ++k) {
        if (k >= K) break;
        const unsigned slot = slots[k];
        float4 w0 = *reinterpret_cast<const float4*>(&s_w[k * kCpeSlice + 8 * q]);
        float4 w1 = *reinterpret_cast<const float4*>(&s_w[k * kCpeSlice + 8 * q + 4]);
        float4 v0, v1;
        if (slot == 0xFFFEu) {                       // overflow row (rare): straight from global
          const int64_t ni = neigh[h_row * K + k];
          v0 = *reinterpret_cast<const float4*>(x + ni * C + sl * kCpeSlice + 8 * q);
          v1 = *reinterpret_cast<const float4*>(x + ni * C + sl * kCpeSlice + 8 * q + 4);
        } else {
          const unsigned a = slot < (unsigned)kCpeCap ? slot : 0u;
          v0 = *reinterpret_cast<const float4*>(&s_buf[a * kCpeStride + 8 * q]);
          v1 = *reinterpret_cast<const float4*>(&s_buf[a * kCpeStride + 8 * q + 4]);
          if (slot == 0xFFFFu) { w0 = make_float4(0.f, 0.f, 0.f, 0.f); w1 = w0; }
        }
        a0 = hfl_fma4(w0, v0, a0);
        a1 = hfl_fma4(w1, v1, a1);
      }
      acc[sl][0] = a0;
      acc[sl][1] = a1;
    }

    // ---- 3. LayerNorm over the row (4 lanes x NS x 8 channels) + residual ---------------------------
    float sum = 0.f;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
      sum += ((acc[sl][0].x + acc[sl][0].y) + (acc[sl][0].z + acc[sl][0].w)) +
             ((acc[sl][1].x + acc[sl][1].y) + (acc[sl][1].z + acc[sl][1].w));
    const float inv_c = 1.0f / (float)C;
    const float mean = hfl_group_sum<4>(sum) * inv_c;
    float sq = 0.f;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float4& a = acc[sl][j];
        a.x -= mean; a.y -= mean; a.z -= mean; a.w -= mean;
        sq += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w);
      }
    const float rstd = 1.0f / sqrtf(hfl_group_sum<4>(sq) * inv_c + eps);
    if (live) {
#pragma unroll
      for (int sl = 0; sl < NS; ++sl)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ch = sl * kCpeSlice + 8 * q + 4 * j;
          const float4 gm = *reinterpret_cast<const float4*>(gamma + ch);
          const float4 bt = *reinterpret_cast<const float4*>(beta + ch);
          const float4 a = acc[sl][j];
          float4 y = make_float4(fmaf(a.x * rstd, gm.x, bt.x), fmaf(a.y * rstd, gm.y, bt.y),
                                 fmaf(a.z * rstd, gm.z, bt.z), fmaf(a.w * rstd, gm.w, bt.w));
          if (residual) {
            const float4 xv = *reinterpret_cast<const float4*>(x + h_row * C + ch);
            y.x += xv.x; y.y += xv.y; y.z += xv.z; y.w += xv.w;
          }
          *reinterpret_cast<float4*>(out + h_row * C + ch) = y;
        }
    }
  }
}

static int g_cpe_variant = 0;          // 0: direct gathers (default), 1: LDS-staged gathers (C = 128 / 256)
static int g_cpe_lds_wgs_per_cu = 3;

template <int C>
static int launch_cpe_lds(float* out, const float* x, const float* w, const float* gamma,
                          const float* beta, const int32_t* neigh, int64_t n, int K, float eps,
                          int residual, hipStream_t s) {
  const int64_t need = hfl_cdiv(n, kCpeRows);
  const int64_t cap = (int64_t)hfl_num_cus() * g_cpe_lds_wgs_per_cu;
  const int blocks = (int)(need < cap ? need : cap);
  cpe_fwd_lds_kernel<C><<<blocks, 256, 0, s>>>(out, x, w, gamma, beta, neigh, n, K, eps, residual);
  HFL_RETURN_LAST_ERROR();
}

static int g_cpe_chunk_rows = 0;   // 0: rows interleaved over blocks; >0: contiguous chunk per block

template <int TPR>
static int launch_cpe(float* out, const float* x, const float* w, const float* gamma,
                      const float* beta, const int32_t* neigh, int64_t n, int K, float eps,
                      int residual, hipStream_t s) {
  constexpr int RPB = 256 / TPR;
  const size_t lds = (size_t)K * TPR * 16 + (size_t)RPB * K * (sizeof(int32_t) + 1) + 16;
  const int64_t need = hfl_cdiv(n, RPB);
  const int blocks = (int)(need < (int64_t)hfl_num_cus() * 8 ? need : (int64_t)hfl_num_cus() * 8);
  cpe_fwd_kernel<TPR><<<blocks, 256, lds, s>>>(out, x, w, gamma, beta, neigh, n, K, eps, residual,
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
  return (int64_t)wgrad_blocks(n_rows, g.rpb) * g.rpb * kngh * channels * (int64_t)sizeof(float);
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

int hfl_cpe_forward(float* out, const float* x, const float* weight, const float* gamma,
                    const float* beta, const int32_t* neigh, int64_t n_rows, int64_t channels,
                    int kngh, float eps, int residual, hfl_stream_t stream) {
  if (n_rows < 0 || kngh <= 0 || kngh > kMaxTaps) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g_cpe_variant == 1 && channels == 256)
    return launch_cpe_lds<256>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
  if (g_cpe_variant == 1 && channels == 128)
    return launch_cpe_lds<128>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
  switch (channels) {
    case 256: return launch_cpe<64>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 128: return launch_cpe<32>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 64:  return launch_cpe<16>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    case 32:  return launch_cpe<8>(out, x, weight, gamma, beta, neigh, n_rows, kngh, eps, residual, s);
    default:  return HFL_EINVAL;
  }
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

/* internal tuning hook used by hfl_set_variant("cpe_chunk_rows", n) */
void hfl_internal_set_cpe_chunk(int rows) { g_cpe_chunk_rows = rows; }
void hfl_internal_set_cpe_variant(int v, int wgs) {
  if (v >= 0) g_cpe_variant = v;
  if (wgs > 0) g_cpe_lds_wgs_per_cu = wgs;
}

}  // extern "C"
