// Weight gradient of an octree convolution in its live-tap form (model.OctreeConv._forward_live_taps):
//
//     dW[k] (Cin, Cout) = g_k^T . dpart_k       for every tap k, over the pairs [edges[k], edges[k+1]) of that tap
//
// g (P, Cin) holds the gathered input row of every live (row, tap) pair, dpart (P, Cout) the output gradient of the
// pair's row.  Replaces what autograd does for ocnn's OctreeConv in the reference (octree2col + mm:
// models/layers/octformer_layers.py:89-95, models/octformer_backbone.py:470-475): a (27 Cin, N) x (N, Cout) GEMM over a
// column matrix that is 80-94 % zeros.  Per tap the product contracts thousands of pairs into a 64 x 64 .. 128 x 128
// matrix -- a shape the BLAS library runs at a few TF/s -- so it is done here: fp32 MFMA (v_mfma_f32_16x16x4_f32, the
// reference's arithmetic), operands straight from global memory in their natural row-major layout (lane (c, q) of an
// MFMA holds element [pair q][a channel of lane c]: 16 B = four channels, one per feature block, per lane), no LDS staging.
//
// Work split: the pair list is cut into chunks of <= 2048 pairs that never straddle a tap (host-built table, cached with
// the tap lists); workgroup = (chunk, 64 x 64 output tile), its 4 waves take interleaved 16-pair groups and are combined in
// LDS in a fixed order; partial tiles go to a workspace and a second kernel adds each tap's chunks in ascending order:
// bitwise reproducible, no atomics.
#include "hfl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool GATHER_G, bool GATHER_D>
__global__ void __launch_bounds__(256)
tap_wgrad_kernel(float* __restrict__ ws, const float* __restrict__ g, const float* __restrict__ dpart,
                 const int32_t* __restrict__ g_rows, const int32_t* __restrict__ d_rows,
                 const int32_t* __restrict__ chunks, int cin, int cout) {
  __shared__ __attribute__((aligned(16))) float red[3][64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int chunk = blockIdx.x;
  const int tiles_o = cout >> 6;
  const int ci0 = (blockIdx.y / tiles_o) << 6, co0 = (blockIdx.y % tiles_o) << 6;
  const int begin = chunks[3 * chunk + 1], end = chunks[3 * chunk + 2];

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // wave w takes the 16-pair groups w, w + 4, ... of the chunk; inside a group four MFMA steps of 4 pairs.
  // Lane (c, q) of an MFMA operand holds [pair q][one channel of lane c]; WHICH channel is ours to choose, so lane c takes the
  // four consecutive channels 4c .. 4c + 3 of the tile for the four feature blocks i (j) -- one 16-B load per pair and operand
  // instead of four 4-B loads 64 B apart (the launch was bound by the number of vector-memory instructions).
  // Every address below is valid (a pair index past the chunk's end is clamped to its last pair and the loaded values
  // dropped by a select), so nothing is loaded under a branch, and the group's operands are requested one group AHEAD, its
  // table entries two: the 64 MFMAs of a group run while the next group's sixteen 16-B reads are in flight (left to itself
  // hipcc sinks every read to its first use and waits for each in turn).
  auto rows_of = [&](int p0, int (&rg)[4], int (&rd)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = p0 + 4 * s + q;
      const int pc = p < end ? p : end - 1;
      // (hfl_tap_wgrad_gather: pair p reads row g_rows[p] of the layer input / d_rows[p] of the output gradient -- the
      //  pair-major copies of both never exist)
      rg[s] = GATHER_G ? g_rows[pc] : pc;
      rd[s] = GATHER_D ? d_rows[pc] : pc;
    }
  };
  auto operands_of = [&](const int (&rg)[4], const int (&rd)[4], f32x4 (&a)[4], f32x4 (&b)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      a[s] = *reinterpret_cast<const f32x4*>(g + (int64_t)rg[s] * cin + ci0 + 4 * c);
      b[s] = *reinterpret_cast<const f32x4*>(dpart + (int64_t)rd[s] * cout + co0 + 4 * c);
    }
  };
  if (begin < end) {
    int p0 = begin + wave * 16;
    f32x4 a[4], b[4], an[4], bn[4];
    int rg[4], rd[4];
    rows_of(p0, rg, rd);
    operands_of(rg, rd, a, b);
    rows_of(p0 + 64, rg, rd);
    for (; p0 < end; p0 += 64) {
      operands_of(rg, rd, an, bn);                  // group p0 + 64 (its rows were requested one group ago)
      rows_of(p0 + 128, rg, rd);                    // raw table entries only: nothing below the barrier uses them
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool ok = p0 + 4 * s + q < end;
#pragma unroll
        for (int i = 0; i < 4; ++i) a[s][i] = ok ? a[s][i] : 0.f;
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = an[s];
        b[s] = bn[s];
      }
    }
  }

  // combine the four waves in a fixed order: waves 1..3 park their tiles in LDS, wave 0 adds them 1, 2, 3
  // accumulator layout: lane (c, q), element e of acc[i][j] = input channel 4 (4 q + e) + i, output channel 4 c + j
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<f32x4*>(&red[wave - 1][(4 * (4 * q + e) + i) * 64 + 4 * c]) =
            (f32x4){acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
  }
  __syncthreads();
  if (wave == 0) {
    float* out = ws + (int64_t)chunk * cin * cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * (4 * q + e) + i;
        f32x4 v = (f32x4){acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
        v += *reinterpret_cast<const f32x4*>(&red[0][r * 64 + 4 * c]);
        v += *reinterpret_cast<const f32x4*>(&red[1][r * 64 + 4 * c]);
        v += *reinterpret_cast<const f32x4*>(&red[2][r * 64 + 4 * c]);
        *reinterpret_cast<f32x4*>(out + (int64_t)(ci0 + r) * cout + co0 + 4 * c) = v;
      }
  }
}

// dw[k] = sum of the tap's chunk partials in ascending chunk order (taps without pairs get zeros)
__global__ void __launch_bounds__(256)
tap_wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ ws, const int32_t* __restrict__ tap_off,
                        int n4) {
  const int k = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ch = tap_off[k]; ch < tap_off[k + 1]; ++ch) {
    const float4 v = reinterpret_cast<const float4*>(ws)[(int64_t)ch * n4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  reinterpret_cast<float4*>(dw)[(int64_t)k * n4 + i] = a;
}

}  // namespace

extern "C" int hfl_tap_wgrad_gather(float* dw, const float* g, const int32_t* g_rows, const float* dpart,
                                    const int32_t* d_rows, const int32_t* chunks, int n_chunks,
                                    const int32_t* tap_chunk_off, int taps, int cin, int cout, float* workspace,
                                    hfl_stream_t stream) {
  if (taps <= 0 || cin <= 0 || cout <= 0 || cin % 64 != 0 || cout % 64 != 0 || n_chunks < 0) return HFL_EINVAL;
  if (dw == nullptr || tap_chunk_off == nullptr) return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_chunks > 0) {
    if (g == nullptr || dpart == nullptr || chunks == nullptr || workspace == nullptr) return HFL_EINVAL;
    dim3 grid((unsigned)n_chunks, (unsigned)((cin / 64) * (cout / 64)));
    if (g_rows != nullptr && d_rows != nullptr)
      tap_wgrad_kernel<true, true><<<grid, 256, 0, s>>>(workspace, g, dpart, g_rows, d_rows, chunks, cin, cout);
    else if (g_rows != nullptr)
      tap_wgrad_kernel<true, false><<<grid, 256, 0, s>>>(workspace, g, dpart, g_rows, d_rows, chunks, cin, cout);
    else if (d_rows != nullptr)
      tap_wgrad_kernel<false, true><<<grid, 256, 0, s>>>(workspace, g, dpart, g_rows, d_rows, chunks, cin, cout);
    else
      tap_wgrad_kernel<false, false><<<grid, 256, 0, s>>>(workspace, g, dpart, g_rows, d_rows, chunks, cin, cout);
  }
  const int n4 = cin * cout / 4;
  dim3 rgrid((unsigned)hfl_cdiv(n4, 256), (unsigned)taps);
  tap_wgrad_reduce_kernel<<<rgrid, 256, 0, s>>>(dw, workspace, tap_chunk_off, n4);
  HFL_RETURN_LAST_ERROR();
}

extern "C" int hfl_tap_wgrad(float* dw, const float* g, const float* dpart, const int32_t* chunks, int n_chunks,
                             const int32_t* tap_chunk_off, int taps, int cin, int cout, float* workspace,
                             hfl_stream_t stream) {
  return hfl_tap_wgrad_gather(dw, g, nullptr, dpart, nullptr, chunks, n_chunks, tap_chunk_off, taps, cin, cout, workspace,
                              stream);
}
