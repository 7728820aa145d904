// Raw-cloud pre-steps in front of the octree build, batched on the device:
//
//   normalise into [-1,1] by the bounding box   datasets/augmentation.py:213-223 (Normalize, scale_factor=None,
//                                                unit_sphere_norm=False, zero_mean=True)
//   drop points with any |coordinate| > 1        eval/pnv_evaluate.py:163-164
//   cylindrical configs: drop |xy| > 1           eval/pnv_evaluate.py:166-169
//   optionally (x,y,z) -> (rho,phi,z) in [-1,1]  datasets/coordinate_utils.py:30-45,68-91,104-116
//
// One 1024-lane workgroup per cloud: a min/max pass, then an order-preserving compaction of the kept points
// into the cloud's own slot of the output (no cross-cloud scan, no atomics).  The arithmetic is the reference's
// fp32 torch sequence with every rounding kept (explicit _rn intrinsics: no FMA contraction), so normalised
// coordinates and both masks are BIT-EXACT; the |xy| norm is torch's CPU reduction sqrt(fma(y, y, x*x)).
// The optional transform evaluates np.interp's `slope*(x - xp0) + fp0` in float64 like the reference, but atan2f
// here is the device library's (<= 1-2 ulp from torch's CPU atan2), which can move a point on a cell boundary:
// callers that need bit-parity with the reference run the transform on the host (hotformerloc_amd/preprocess.py).
#include "hfl_common.h"

namespace {

constexpr int kPrepThreads = 1024;

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const float o = __shfl_xor(v, m, 64);
    v = is_max ? fmaxf(v, o) : fminf(v, o);
  }
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < kPrepThreads / 64; ++w) r = is_max ? fmaxf(r, red[w]) : fminf(r, red[w]);
  return r;
}

__device__ __forceinline__ double interp2(double x, double x0, double x1, double y0, double y1) {
  // numpy arr_interp with two knots: clamp outside, exact knot values, else slope*(x - x0) + y0 (mul, add: 2 roundings)
  if (x > x1) return y1;
  if (x < x0) return y0;
  if (x == x1) return y1;
  if (x == x0) return y0;
  const double slope = __ddiv_rn(__dsub_rn(y1, y0), __dsub_rn(x1, x0));
  return __dadd_rn(__dmul_rn(slope, __dsub_rn(x, x0)), y0);
}

__global__ void __launch_bounds__(kPrepThreads)
prepare_clouds_kernel(float* __restrict__ out, int32_t* __restrict__ counts, const float* __restrict__ pts,
                      const int64_t* __restrict__ off, int normalize, int cyl_mask, int cyl_transform) {
  __shared__ float red[kPrepThreads / 64];
  __shared__ int wave_cnt[kPrepThreads / 64];
  __shared__ int running;
  const int b = blockIdx.x;
  const int64_t p0 = off[b];
  const int64_t n = off[b + 1] - p0;
  const float* p = pts + p0 * 3;
  float* o = out + p0 * 3;
  float cx = 0.f, cy = 0.f, cz = 0.f, factor = 1.f;
  if (normalize) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = threadIdx.x; i < n; i += kPrepThreads) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float v = p[i * 3 + a];
        mn[a] = fminf(mn[a], v);
        mx[a] = fmaxf(mx[a], v);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = block_reduce(mn[a], false, red);
      mx[a] = block_reduce(mx[a], true, red);
    }
    cx = __fmul_rn(__fadd_rn(mn[0], mx[0]), 0.5f);
    cy = __fmul_rn(__fadd_rn(mn[1], mx[1]), 0.5f);
    cz = __fmul_rn(__fadd_rn(mn[2], mx[2]), 0.5f);
    const float ext = fmaxf(fmaxf(__fsub_rn(mx[0], mn[0]), __fsub_rn(mx[1], mn[1])), __fsub_rn(mx[2], mn[2]));
    factor = __fdiv_rn(2.0f, __fadd_rn(ext, 1.0e-6f));
  }
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int64_t base = 0; base < n; base += kPrepThreads) {
    const int64_t i = base + threadIdx.x;
    float x = 0.f, y = 0.f, z = 0.f;
    bool keep = false;
    if (i < n) {
      x = p[i * 3 + 0];
      y = p[i * 3 + 1];
      z = p[i * 3 + 2];
      if (normalize) {
        x = __fmul_rn(__fsub_rn(x, cx), factor);
        y = __fmul_rn(__fsub_rn(y, cy), factor);
        z = __fmul_rn(__fsub_rn(z, cz), factor);
      }
      keep = fabsf(x) <= 1.0f && fabsf(y) <= 1.0f && fabsf(z) <= 1.0f;
      if (cyl_mask) keep = keep && __fsqrt_rn(__fmaf_rn(y, y, __fmul_rn(x, x))) <= 1.0f;
      if (cyl_transform && keep) {
        const float phi = atan2f(y, x);
        const float rho = __fsqrt_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
        const double kPi = 3.141592653589793;
        x = (float)interp2((double)rho, 0.0, 1.0, -1.0, 1.0);
        y = (float)interp2((double)phi, -kPi, kPi, -1.0, 1.0);
        x = fminf(fmaxf(x, -1.0f), 1.0f);
        y = fminf(fmaxf(y, -1.0f), 1.0f);
        z = fminf(fmaxf(z, -1.0f), 1.0f);
      }
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int pos = running;
    for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
    pos += __popcll(m & lt);
    if (keep) {
      o[(int64_t)pos * 3 + 0] = x;
      o[(int64_t)pos * 3 + 1] = y;
      o[(int64_t)pos * 3 + 2] = z;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
      for (int w = 0; w < kPrepThreads / 64; ++w) t += wave_cnt[w];
      running += t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[b] = running;
}

}  // namespace

extern "C" int hfl_prepare_clouds(float* out_points, int32_t* out_counts, const float* points,
                                  const int64_t* cloud_offsets, int batch, int normalize, int cylindrical_mask,
                                  int cylindrical_transform, hfl_stream_t stream) {
  if (batch < 0 || out_points == nullptr || out_counts == nullptr || points == nullptr || cloud_offsets == nullptr)
    return HFL_EINVAL;
  if (out_points == points) return HFL_EINVAL;        // compaction reads ahead of what it writes only per cloud
  if (batch == 0) return HFL_OK;
  prepare_clouds_kernel<<<batch, kPrepThreads, 0, static_cast<hipStream_t>(stream)>>>(
      out_points, out_counts, points, cloud_offsets, normalize, cylindrical_mask, cylindrical_transform);
  HFL_RETURN_LAST_ERROR();
}
