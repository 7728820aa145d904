// Shared helpers for the HOTFormerLoc gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hotformerloc_hip.h"

#define HFL_WAVE 64

#define HFL_RETURN_LAST_ERROR()            \
  do {                                     \
    hipError_t e__ = hipGetLastError();    \
    return e__ == hipSuccess ? HFL_OK : (int)e__; \
  } while (0)

static inline int hfl_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

// CUs a launch on `stream` can use: the device's, or the size of the stream's CU mask when it was made by
// hfl_stream_create_cu_mask (csrc/capi.hip keeps the registry).  Persistent grids are sized with this.
extern "C" int hfl_internal_stream_cus(void* stream);
static inline int hfl_stream_cus(hipStream_t s) { return hfl_internal_stream_cus(static_cast<void*>(s)); }

static inline int64_t hfl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A logical (rows, C) f32 matrix whose rows live in up to four separate arrays (hfl_row_segments of the C-ABI): rows
// [row[i], row[i + 1]) at ptr[i]; unused entries have row = INT64_MAX.  n == 1 is the plain pointer.
struct HflRowSeg {
  const float* ptr[4];
  long long row[4];
  int n;
};
static inline HflRowSeg hfl_seg_single(const float* p) {
  HflRowSeg s;
  for (int i = 0; i < 4; ++i) { s.ptr[i] = p; s.row[i] = i == 0 ? 0 : 0x7fffffffffffffffLL; }
  s.n = 1;
  return s;
}
// from the C-ABI struct; false when it is malformed for a matrix of n_rows rows
static inline bool hfl_seg_from(const hfl_row_segments* in, int64_t n_rows, HflRowSeg* out) {
  if (in == nullptr || in->n < 1 || in->n > 4 || in->row0[0] != 0) return false;
  *out = hfl_seg_single(in->ptr[0]);
  out->n = in->n;
  for (int i = 0; i < in->n; ++i) {
    if (in->ptr[i] == nullptr || (i > 0 && in->row0[i] < in->row0[i - 1]) || in->row0[i] > n_rows) return false;
    out->ptr[i] = in->ptr[i];
    out->row[i] = in->row0[i];
  }
  return true;
}
__device__ __forceinline__ const float* hfl_seg_row(const HflRowSeg& s, long long r, long long stride) {
  const float* p = s.ptr[0] + r * stride;
  if (s.n > 1) {
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (r >= s.row[i]) p = s.ptr[i] + (r - s.row[i]) * stride;
  }
  return p;
}

// index-table element access for int32 / int64 neighbour tables
template <typename IdxT>
__device__ __forceinline__ int64_t hfl_ld_idx(const IdxT* p, int64_t i) {
  return (int64_t)p[i];
}

// wave-wide butterfly reductions over the low `width` lanes groups (width power of 2)
template <int WIDTH>
__device__ __forceinline__ float hfl_group_sum(float v) {
#pragma unroll
  for (int m = WIDTH / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, HFL_WAVE);
  return v;
}
template <int WIDTH>
__device__ __forceinline__ float hfl_group_max(float v) {
#pragma unroll
  for (int m = WIDTH / 2; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, HFL_WAVE));
  return v;
}

__device__ __forceinline__ float4 hfl_fma4(float4 a, float4 b, float4 c) {
  c.x = fmaf(a.x, b.x, c.x);
  c.y = fmaf(a.y, b.y, c.y);
  c.z = fmaf(a.z, b.z, c.z);
  c.w = fmaf(a.w, b.w, c.w);
  return c;
}
