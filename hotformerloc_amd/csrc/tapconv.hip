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
// MFMA wants element [pair q][channel c]: 16 consecutive floats of a row per 16 lanes), no LDS staging.
//
// Work split: the pair list is cut into chunks of <= 2048 pairs that never straddle a tap (host-built table, cached with
// the tap lists); workgroup = (chunk, 64 x 64 output tile), its 4 waves take interleaved 16-pair groups and are combined in
// LDS in a fixed order; partial tiles go to a workspace and a second kernel adds each tap's chunks in ascending order:
// bitwise reproducible, no atomics.
#include "hfl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
tap_wgrad_kernel(float* __restrict__ ws, const float* __restrict__ g, const float* __restrict__ dpart,
                 const int32_t* __restrict__ g_rows, const int32_t* __restrict__ d_rows,
                 const int32_t* __restrict__ chunks, int cin, int cout) {
  __shared__ float red[3][64 * 64];
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

  // wave w takes the 16-pair groups w, w + 4, ... of the chunk; inside a group four MFMA steps of 4 pairs
  for (int p0 = begin + wave * 16; p0 < end; p0 += 64) {
    float a[4][4], b[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = p0 + 4 * s + q;
      const bool ok = p < end;
      // (hfl_tap_wgrad_gather: pair p reads row g_rows[p] of the layer input / d_rows[p] of the output gradient -- the
      //  pair-major copies of both never exist)
      const int64_t pg = g_rows != nullptr ? (int64_t)g_rows[ok ? p : begin] : (int64_t)p;
      const int64_t pd = d_rows != nullptr ? (int64_t)d_rows[ok ? p : begin] : (int64_t)p;
      const float* gr = g + pg * cin + ci0 + c;
      const float* dr = dpart + pd * cout + co0 + c;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[s][i] = ok ? gr[16 * i] : 0.f;
        b[s][i] = ok ? dr[16 * i] : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i][j], 0, 0, 0);
  }

  // combine the four waves in a fixed order: waves 1..3 park their tiles in LDS, wave 0 adds them 1, 2, 3
  // accumulator layout: lane (c, q) holds rows ci = 16 i + 4 q + e, column co = 16 j + c
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][(16 * i + 4 * q + e) * 64 + 16 * j + c] = acc[i][j][e];
  }
  __syncthreads();
  if (wave == 0) {
    float* out = ws + (int64_t)chunk * cin * cout;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 16 * i + 4 * q + e, cc = 16 * j + c;
          float v = acc[i][j][e];
          v += red[0][r * 64 + cc];
          v += red[1][r * 64 + cc];
          v += red[2][r * 64 + cc];
          out[(int64_t)(ci0 + r) * cout + co0 + cc] = v;
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
    tap_wgrad_kernel<<<grid, 256, 0, s>>>(workspace, g, dpart, g_rows, d_rows, chunks, cin, cout);
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
