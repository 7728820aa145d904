// fp32-GRADE Linear on the bf16 matrix cores of gfx950, hand-written: the matched-precision leg of the transformer blocks.
//
//     y (M,N) = x (M,K) . W (N,K)^T  [+ bias]  [GELU]  [* row_scale]  [+ residual]        (all f32 in memory)
//
// Replaces torch.nn.Linear + the element-wise op that follows it where the caller wants the reference's own arithmetic
// (fp32 Linear layers: models/octformer_backbone.py:70,91; fc1 -> GELU -> fc2: models/layers/octformer_layers.py:53-59;
// residual adds: models/octformer_backbone.py:275-278, models/hotformerloc_backbone.py:213-216) instead of the 16-bit-operand
// split of csrc/gemm_x3.hip.
//
// Arithmetic: every fp32 operand is split into THREE bf16 planes, h = RNE(v), m = RNE(v - h), l = RNE(v - h - m).  Both
// subtractions are exact in fp32 and 3 x 8 significand bits cover fp32's 24, so v = h + m + l EXACTLY.  Of the nine
// plane products the six of relative size >= 2^-16 are computed, smallest first, with fp32 accumulation
// (v_mfma_f32_16x16x32_bf16):  h l + l h + m m  +  h m + m h  +  h h.   Dropped: m l, l m (<= 2^-24 |x w| each), l l
// (2^-32) -- the size of ONE fp32 rounding of the product, i.e. what an fp32 FMA chain commits per term anyway.  Measured
// against fp64 the result is as close as the fp32 library GEMM's (tests/test_gpu_kernels.py::test_linear_x6_*).
// Rate: 6 bf16 MFMA products per fp32 product = 2.5 PF / 6 = 416 TF/s fp32-equivalent peak, 2.6x the fp32 matrix rate
// (157 TF/s, v_mfma_f32_16x16x4_f32).
//
// Kernel: 256 (rows) x 128 (features) tile per 512-lane workgroup, 4 x 2 waves of 64 x 64, K step 32, ONE workgroup per CU
// (two waves per SIMD, <= 256 VGPRs), two 72-KiB LDS stages (separate __shared__ arrays: hipcc tracks LDS-DMA per array, so
// the fragment reads of the current stage do not wait for the DMA that fills the other one), one barrier per k-step.
//  * x arrives as plain f32: global -> registers (issued one k-step ahead) -> split into the three planes -> ds_write_b64
//    into the next stage.  No producer has to know the operand layout.
//  * W is pre-split once per parameter (hfl_linear_x6_pack: (3, N, K) bf16) and moves global -> LDS by
//    `global_load_lds_dwordx4` (16 rows x 64 B of one plane per wave-instruction).
//  * LDS images: per plane [row][32 k] = 64 B per row; the 16-B chunk q of row r sits at slot q ^ f((r >> 2) & 3),
//    f = (0, 2, 3, 1): a ds_read_b128 fragment read (16 rows x one chunk per 16-lane group) touches all 64 banks once.
//    For the DMA the permutation is applied to the SOURCE address (LDS-DMA writes are lane-linear).
//  * operand traffic at the matrix rate: 56 KiB per k-step per CU (18 B/clk; a 128 x 128 tile would need 26, above what an
//    XCD's L2 sustains per CU).
//  * W is the MFMA's A operand, x its B operand: a lane ends up with 4 consecutive features of one row; the epilogue
//    transposes through LDS so that every store instruction writes whole 128-B lines (bias / GELU / residual fused).
//  * 1-D grid, XCD-aware as csrc/gemm_x3.hip: the N/128 feature tiles of one row tile run on ONE XCD.
#include "hfl_common.h"
#include "x3_math.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int X6_BM = 256, X6_BN = 128;
constexpr int X6_APLANE = X6_BM * 64;                       // bytes of one plane of the x tile
constexpr int X6_WPLANE = X6_BN * 64;
constexpr int X6_WBASE = 3 * X6_APLANE;
constexpr int X6_STAGE = 3 * X6_APLANE + 3 * X6_WPLANE;     // 73 728 B

struct X6Params {
  float* out;               // (M, N) f32
  const float* x;           // (M, K) f32
  const uint16_t* w3;       // (3, N, K) bf16: planes h, m, l
  const float* bias;        // (N) or null
  const float* residual;    // (M, N) or null (may alias out)
  const float* row_scale;   // (M) or null: out = (acc + bias) * row_scale[m] + residual
  int64_t M;
  int N, K;
  int tiles_n;
  int64_t n_wg;
};

__device__ __forceinline__ int x6_swz(int row) { return (0x1320 >> (((row >> 2) & 3) << 2)) & 3; }

// four floats -> their three bf16 planes (4 x 16 bit = 8 B per plane)
__device__ __forceinline__ void x6_split4(const float4 v, uint2& h, uint2& m, uint2& l) {
  uint32_t h01, m01, h23, m23;
  x3_split_pair_scalar(v.x, v.y, h01, m01);
  x3_split_pair_scalar(v.z, v.w, h23, m23);
  // second residual: exact again (v - h is a multiple of ulp(v) with at most 16 significant bits)
  const float r0 = (v.x - __uint_as_float(h01 << 16)) - __uint_as_float(m01 << 16);
  const float r1 = (v.y - __uint_as_float(h01 & 0xffff0000u)) - __uint_as_float(m01 & 0xffff0000u);
  const float r2 = (v.z - __uint_as_float(h23 << 16)) - __uint_as_float(m23 << 16);
  const float r3 = (v.w - __uint_as_float(h23 & 0xffff0000u)) - __uint_as_float(m23 & 0xffff0000u);
  const uint32_t l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, x3_bf16x2));
  const uint32_t l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r2, r3}, x3_bf16x2));
  h = make_uint2(h01, h23);
  m = make_uint2(m01, m23);
  l = make_uint2(l01, l23);
}

// acc[i][j]: features 16 i + 4 fq .. +3 (registers) of row 16 j + frow of the 64 x 64 tile at (m_tile, n_tile) one wavefront
// owns; ep = that wavefront's private 8-KiB LDS region.  The wave transposes 32 rows at a time (16-B chunks XOR-swizzled by
// the row: conflict-free both ways) so that 16 consecutive lanes hold 256 contiguous bytes of one output row.
template <int GELU>
__device__ __forceinline__ void x6_epilogue(const X6Params& p, f32x4 (&acc)[4][4], unsigned char* ep, int64_t m_tile, int n_tile,
                                            int lane) {
  const int frow = lane & 15, fq = lane >> 4;
  const int N = p.N;
  const int ecol = lane & 15;
  const int nbase = n_tile + ecol * 4;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr) b = *reinterpret_cast<const float4*>(p.bias + nbase);
  // every residual value of the tile is requested before the first store (`residual` may be `out`: a load behind a store of
  // the same lane's element would otherwise wait for it, one round trip per 4 rows)
  float4 rs[GELU ? 1 : 2][GELU ? 1 : 8];
  if (!GELU) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int64_t m = m_tile + h * 32 + it * 4 + fq;
        rs[h][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.residual != nullptr && m < p.M) rs[h][it] = *reinterpret_cast<const float4*>(p.residual + m * N + nbase);
      }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = jj * 16 + frow;
        *reinterpret_cast<f32x4*>(ep + r * 256 + (((i * 4 + fq) ^ frow) << 4)) = acc[i][2 * h + jj];
      }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = it * 4 + fq;
      const f32x4 a = *reinterpret_cast<const f32x4*>(ep + r * 256 + ((ecol ^ (r & 15)) << 4));
      const int64_t m = m_tile + h * 32 + r;
      if (m >= p.M) continue;
      float4 v = make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
      if (GELU) {
        const f32x2 g01 = x3_gelu2((f32x2){v.x, v.y}), g23 = x3_gelu2((f32x2){v.z, v.w});
        v = make_float4(g01[0], g01[1], g23[0], g23[1]);
      } else {
        if (p.row_scale != nullptr) {
          const float rsc = p.row_scale[m];
          v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc;
        }
        if (p.residual != nullptr) {
          const float4 r4 = rs[GELU ? 0 : h][GELU ? 0 : it];
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
      }
      *reinterpret_cast<f32x4*>(p.out + m * N + nbase) = (f32x4){v.x, v.y, v.z, v.w};
    }
  }
}

template <int GELU>
__global__ void __launch_bounds__(512, 2)
gemm_x6_kernel(const X6Params p) {
  __shared__ __attribute__((aligned(1024))) unsigned char s0[X6_STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char s1[X6_STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;

  // XCD-aware tile assignment (bijective remap: consecutive new ids share an XCD)
  int64_t wg = blockIdx.x;
  {
    const int64_t q = p.n_wg >> 3, r = p.n_wg & 7;
    const int64_t xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int64_t m0 = (wg / p.tiles_n) * X6_BM;
  const int n0 = (int)(wg % p.tiles_n) * X6_BN;
  const int K = p.K;
  const int nk = K >> 5;

  // ---- x: lane -> (row within 8, 16-B chunk); instruction i covers rows wave * 32 + 8 i .. + 7
  const int arow = lane >> 3, ac = lane & 7;
  const float* xrow[4];
  int a_lds[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 32 + i * 8 + arow;
    int64_t m = m0 + r;
    if (m >= p.M) m = p.M - 1;                                   // tail rows read the last valid row, never stored
    xrow[i] = p.x + m * K + ac * 4;
    a_lds[i] = r * 64 + (((ac >> 1) ^ x6_swz(r)) << 4) + (ac & 1) * 8;
  }
  // ---- W: three 1-KiB pieces per wave and k-step; piece P = wave * 3 + t: plane P >> 3, rows 16 (P & 7) .. + 15
  uint32_t w_src[3];
  int w_lds[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int P = wave * 3 + t;
    const int plane = P >> 3, rblk = P & 7;
    const int row = rblk * 16 + (lane >> 2);
    const int q = (lane & 3) ^ x6_swz(row);
    w_src[t] = (uint32_t)(((int64_t)plane * p.N + n0 + row) * K * 2 + q * 16);      // (the launcher checked 3 N K 2 < 2^32)
    w_lds[t] = X6_WBASE + plane * X6_WPLANE + rblk * 1024;
  }
  const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w3);

  // ---- fragment addresses (bytes inside a stage)
  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (fq ^ x6_swz(frow)) << 4;
  const int offx = (wm * 64 + frow) * 64 + fsw;                    // + plane * X6_APLANE + j * 1024
  const int offw = X6_WBASE + (wn * 64 + frow) * 64 + fsw;         // + plane * X6_WPLANE + i * 1024

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 av[4];
  auto load_a = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const float4*>(xrow[i] + kt * 32);
  };
  auto dma_w = [&](unsigned char* st, int kt) {
#pragma unroll
    for (int t = 0; t < 3; ++t)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbytes + w_src[t] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(st + w_lds[t]), 16, 0, 0);
  };
  auto store_a = [&](unsigned char* st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 h, m, l;
      x6_split4(av[i], h, m, l);
      *reinterpret_cast<uint2*>(st + a_lds[i]) = h;
      *reinterpret_cast<uint2*>(st + X6_APLANE + a_lds[i]) = m;
      *reinterpret_cast<uint2*>(st + 2 * X6_APLANE + a_lds[i]) = l;
    }
  };
  auto compute = [&](const unsigned char* st) {
    bf16x8 xf[3][4];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[pl][j] = *reinterpret_cast<const bf16x8*>(st + offx + pl * X6_APLANE + j * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(st + offw + i * 1024);
      const bf16x8 wmid = *reinterpret_cast<const bf16x8*>(st + offw + X6_WPLANE + i * 1024);
      const bf16x8 wl = *reinterpret_cast<const bf16x8*>(st + offw + 2 * X6_WPLANE + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xf[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, xf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, xf[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xf[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xf[0][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // One barrier per k-step: step kt computes from its stage while x (kt + 1) is on its way to registers and W (kt + 1) to the
  // other stage; the split of x (kt + 1) is written behind the MFMAs.
  load_a(0);
  dma_w(s0, 0);
  store_a(s0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) { load_a(kt + 1); dma_w(s1, kt + 1); }
    compute(s0);
    if (kt + 1 < nk) store_a(s1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 2 < nk) { load_a(kt + 2); dma_w(s0, kt + 2); }
    compute(s1);
    if (kt + 2 < nk) store_a(s0);
    __syncthreads();
  }
  // the stages are free after the last barrier: every wave transposes through its own 8 KiB of s0
  x6_epilogue<GELU>(p, acc, s0 + wave * 8192, m0 + wm * 64, n0 + wn * 64, lane);
}

// (N, K) f32 -> (3, N, K) bf16 planes
__global__ void __launch_bounds__(256)
x6_pack_kernel(uint16_t* __restrict__ w3, const float* __restrict__ w, int64_t total4, int64_t plane_elems) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    uint2 h, m, l;
    x6_split4(reinterpret_cast<const float4*>(w)[i], h, m, l);
    *reinterpret_cast<uint2*>(w3 + i * 4) = h;
    *reinterpret_cast<uint2*>(w3 + plane_elems + i * 4) = m;
    *reinterpret_cast<uint2*>(w3 + 2 * plane_elems + i * 4) = l;
  }
}

}  // namespace

extern "C" {

int hfl_linear_x6_pack(uint16_t* w3, const float* w, int64_t out_features, int64_t in_features, hfl_stream_t stream) {
  if (w3 == nullptr || w == nullptr || out_features <= 0 || in_features <= 0 || in_features % 4 != 0) return HFL_EINVAL;
  const int64_t total4 = out_features * in_features / 4;
  const int64_t need = hfl_cdiv(total4, 256);
  x6_pack_kernel<<<(int)(need < 4096 ? need : 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(w3, w, total4,
                                                                                                 out_features * in_features);
  HFL_RETURN_LAST_ERROR();
}

int hfl_linear_x6(float* out, const float* x, const uint16_t* w3, const float* bias, const float* residual,
                  const float* row_scale, int64_t n_rows, int in_features, int out_features, int gelu, hfl_stream_t stream) {
  if (n_rows < 0 || in_features <= 0 || out_features <= 0) return HFL_EINVAL;
  if (in_features % 32 != 0 || out_features % X6_BN != 0) return HFL_EINVAL;
  if (out == nullptr || x == nullptr || w3 == nullptr) return HFL_EINVAL;
  if (gelu && (residual != nullptr || row_scale != nullptr)) return HFL_EINVAL;
  if ((int64_t)3 * out_features * in_features * 2 >= ((int64_t)1 << 32)) return HFL_ECAPACITY;   // 32-bit DMA source offsets
  if (n_rows == 0) return HFL_OK;
  X6Params p;
  p.out = out; p.x = x; p.w3 = w3; p.bias = bias; p.residual = residual; p.row_scale = row_scale;
  p.M = n_rows; p.N = out_features; p.K = in_features;
  p.tiles_n = out_features / X6_BN;
  p.n_wg = hfl_cdiv(n_rows, X6_BM) * p.tiles_n;
  if (p.n_wg > 0x7fffffffLL) return HFL_ECAPACITY;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (gelu) gemm_x6_kernel<1><<<(unsigned)p.n_wg, 512, 0, s>>>(p);
  else gemm_x6_kernel<0><<<(unsigned)p.n_wg, 512, 0, s>>>(p);
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
