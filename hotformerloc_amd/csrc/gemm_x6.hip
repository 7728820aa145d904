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
// against fp64 the result is closer than the fp32 library GEMM's on the step's shapes (2.3e-7 against 2.9e-7 at K = 256;
// tests/test_gpu_kernels.py::test_linear_x6_is_as_accurate_as_the_fp32_library_gemm).
// Rate: 6 bf16 MFMA products per fp32 product = 2.5 PF / 6 = 416 TF/s fp32-equivalent peak, 2.6x the fp32 matrix rate
// (157 TF/s, v_mfma_f32_16x16x4_f32).
//
// Kernel: (64 MT) rows x 128 features per tile (MT = 1 .. 4; or (32 MT) x 256), 8 waves of (16 MT) x 64, K step 32, ONE
// persistent 512-lane workgroup per CU (two waves per SIMD, <= 256 VGPRs), two LDS stages of <= 72 KiB (separate __shared__
// arrays: hipcc tracks LDS-DMA per array, so the fragment reads of the current stage do not wait for the DMA that fills the
// other one); the two waves of a SIMD run half a k-step apart (see the kernel): one is in its MFMA phase while the other reads
// fragments and splits x.
//  * x arrives as plain f32: global -> registers (requested one MFMA phase ahead) -> split into the three planes ->
//    ds_write_b64 into the next stage.  No producer has to know the operand layout.
//  * W is pre-split once per parameter (hfl_linear_x6_pack: (3, N, Kp) bf16, Kp = K rounded up to 64 with zeros) and moves
//    global -> LDS by `global_load_lds_dwordx4` (16 rows x 64 B of one plane per wave-instruction).
//  * LDS images: per plane [row][32 k] = 64 B per row; the 16-B chunk q of row r sits at slot q ^ f((r >> 2) & 3),
//    f = (0, 2, 3, 1): a ds_read_b128 fragment read (16 rows x one chunk per 16-lane group) touches all 64 banks once
//    (SQ_LDS_BANK_CONFLICT = 0).  For the DMA the permutation is applied to the SOURCE address (LDS-DMA writes are lane-linear).
//  * W is the MFMA's A operand, x its B operand: a lane ends up with 4 consecutive features of one row; the epilogue
//    transposes through LDS so that every store instruction writes whole 128-B lines (bias / GELU / residual fused).
//  * persistent tile loop, XCD-aware as csrc/gemm_x3.hip: the feature tiles of one row tile run on ONE XCD.
//
// Where it stands (round 6): 0.30-0.40 of the 2.5 PF bf16 peak issued = 125-165 TF/s fp32-equivalent on
// the step's shapes, 1.05-1.45x the fp32 library GEMM, MFMA pipe 34-40 % busy.  What bounds it is the per-wave ISSUE time of
// everything that is not an MFMA -- LDS-DMA pieces (~100-150 cycles each), 6 (MT + 4) fragment reads, the split's ~28 vector
// instructions per four values -- in the half-step beside the SIMD partner's MFMAs (s_memtime stamps: 2200-2900 cycles
// against 1536).  Six arrangements measured within +-10 % of one another: lock-step with one barrier per k-step; this one;
// 128 x 256 tiles (half the split per MFMA); raised priority for the read / split half; v_mfma_f32_32x32x16 with the DMA
// pieces between the MFMAs; x pre-split into planes by its producer and both operands by LDS-DMA, no vector arithmetic in the
// loop (bitwise the same results; parked under tools/experiments/).
#include "hfl_common.h"
#include "x3_math.h"

#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// Tile geometry: 8 waves as (8 / NWN) x NWN blocks of (16 MT) rows x 64 features: NWN = 2 -> (64 MT) rows x 128 features,
// NWN = 4 -> (32 MT) rows x 256 features.  The wide tile halves the x rows a workgroup splits per MFMA (the split's ~28
// vector instructions per four values compete with the partner wave's MFMAs for the SIMD's issue slots: with 256 x 128
// tiles the non-MFMA half-step took ~2400 cycles against the 1536 of the MFMA half) and re-splits a row tile N / 256 times
// instead of N / 128 times.  LDS stage: three planes of the x tile, three of the W tile.
template <int MT, int NWN> struct X6Geo {
  static constexpr int NWM = 8 / NWN;
  static constexpr int BM = NWM * 16 * MT;
  static constexpr int BN = NWN * 64;
  static constexpr int MTL = BM / 64;                          // float4 of x per lane and k-step
  static constexpr int APLANE = BM * 64;                       // bytes of one plane of the x tile
  static constexpr int WPLANE = BN * 64;
  static constexpr int WPIECES = 3 * BN / 16;                  // 1-KiB LDS-DMA pieces of a W stage
  static_assert(BM % 64 == 0, "the x tile is loaded 64 rows per pass of the workgroup");
};

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
  int64_t n_wg;             // tiles of the launch
  int nk;                   // k-steps of 32 (even: W's K is zero-padded to a multiple of 64)
  int64_t w_plane_bytes;    // bytes of one W plane: (rows of W) x 32 nk x 2
  // grouped launch (the per-tap products of an octree convolution over its live (row, tap) pairs): per row tile {first row,
  // rows (<= the tile height), first row of its W block}; row m of the x operand is x[gather[m]]; n_valid <= 128 features
  // of the (zero-padded) 128-row W blocks are stored
  const int32_t* tiles;
  const int32_t* gather;
  int n_valid;
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

// a wave's private 4-KiB transpose region of the epilogue (16 rows x 256 B).  It lies in the arrays of stage 1: a wave that is
// already in the NEXT tile's prologue writes stage 0 only, and nobody writes stage 1 before the next tile's first barrier, which
// every wave passes after its epilogue.
template <int MT, int NWN>
__device__ __forceinline__ unsigned char* x6_ep_region(unsigned char* sx1, unsigned char* sw1, int wave) {
  constexpr int in_x = (3 * X6Geo<MT, NWN>::APLANE) / 4096;       // regions in the x stage (>= 3), >= 6 more in the W stage
  return wave < in_x ? sx1 + wave * 4096 : sw1 + (wave - in_x) * 4096;
}

// acc[i][j]: features 16 i + 4 fq .. +3 (registers) of row 16 j + frow of the (16 MT) x 64 tile at (m_tile, n_tile) one
// wavefront owns.  The wave transposes 16 rows at a time through its LDS region (16-B chunks XOR-swizzled by the row:
// conflict-free both ways) so that 16 consecutive lanes hold 256 contiguous bytes of one output row: every store instruction
// writes whole 128-B lines.
template <int GELU, int MT>
__device__ __forceinline__ void x6_epilogue(const X6Params& p, f32x4 (&acc)[4][MT], unsigned char* ep, int64_t m_tile, int n_tile,
                                            int lane, int64_t m_end) {
  const int frow = lane & 15, fq = lane >> 4;
  const int N = p.N;
  const int ecol = lane & 15;
  const int nbase = n_tile + ecol * 4;
  float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr) b = *reinterpret_cast<const float4*>(p.bias + nbase);
  // every residual value of the tile is requested before the first store (`residual` may be `out`: a load behind a store of
  // the same lane's element would otherwise wait for it, one round trip per 4 rows)
  float4 rs[GELU ? 1 : MT][GELU ? 1 : 4];
  float rsc[GELU ? 1 : MT][GELU ? 1 : 4];
  if (!GELU) {
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int64_t m = m_tile + j * 16 + it * 4 + fq;
        rs[j][it] = make_float4(0.f, 0.f, 0.f, 0.f);
        rsc[j][it] = 1.0f;
        if (p.residual != nullptr && m < m_end) rs[j][it] = *reinterpret_cast<const float4*>(p.residual + m * N + nbase);
        if (p.row_scale != nullptr && m < m_end) rsc[j][it] = p.row_scale[m];
      }
  }
#pragma unroll
  for (int j = 0; j < MT; ++j) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(ep + frow * 256 + (((i * 4 + fq) ^ frow) << 4)) = acc[i][j];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 4 + fq;
      const f32x4 a = *reinterpret_cast<const f32x4*>(ep + r * 256 + ((ecol ^ r) << 4));
      const int64_t m = m_tile + j * 16 + r;
      if (m >= m_end || nbase >= p.n_valid) continue;
      float4 v = make_float4(a[0] + b.x, a[1] + b.y, a[2] + b.z, a[3] + b.w);
      if (GELU) {
        const f32x2 g01 = x3_gelu2((f32x2){v.x, v.y}), g23 = x3_gelu2((f32x2){v.z, v.w});
        v = make_float4(g01[0], g01[1], g23[0], g23[1]);
      } else {
        if (p.row_scale != nullptr) {
          const float sc = rsc[GELU ? 0 : j][GELU ? 0 : it];
          v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        if (p.residual != nullptr) {
          const float4 r4 = rs[GELU ? 0 : j][GELU ? 0 : it];
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
      }
      *reinterpret_cast<f32x4*>(p.out + m * N + nbase) = (f32x4){v.x, v.y, v.z, v.w};
    }
  }
}

// probe build (-DHFL_X6_STAMPS): s_memtime of wave 0 (group A) and wave 4 (group B) of workgroup 0 at every barrier arrival and
// release of its first tile (tools/x6_stamps.py)
#ifdef HFL_X6_STAMPS
__device__ unsigned long long g_x6_stamps[2][512];
#define X6_STAMP()                                                                                         \
  do {                                                                                                     \
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4) && stamp_n < 512)                         \
      g_x6_stamps[wave >> 2][stamp_n++] = __builtin_amdgcn_s_memtime();                                    \
  } while (0)
#else
#define X6_STAMP() do { } while (0)
#endif

// s_waitcnt with the gfx9 immediate: vmcnt(v) lgkmcnt(l), no wait on expcnt
#define X6_WAIT(v, l) __builtin_amdgcn_s_waitcnt(((v) & 15) | (((v) >> 4) << 14) | (7 << 4) | ((l) << 8))
#define X6_WAIT_LGKM0() X6_WAIT(63, 0)
#define X6_WAIT_ALL() X6_WAIT(0, 0)

// Kernel: persistent, one 512-lane workgroup per CU, tiles of (64 MT) rows x 128 features dealt round-robin; wave w owns the
// (16 MT) x 64 block (w & 3, w >> 2).  The two waves of a SIMD (w and w + 4) work HALF A K-STEP APART: while group A (waves
// 0-3) runs the 24 MT MFMAs of step t from fragments it holds in registers, group B (waves 4-7) reads its fragments of step t
// from LDS, splits its share of x (t + 1) into the other stage and -- half a step later -- runs its own MFMAs while group A
// reads, splits and issues the LDS-DMA of W (t + 1).  Two barriers per k-step; the matrix pipe of every SIMD always has one
// of its two waves in an MFMA phase (in lock-step, as one barrier per k-step has them, the fragment reads, the split and the
// barrier skew of BOTH waves lie between the MFMA phases: the pipe was 34 % busy, PMC).
template <int GELU, int MT, int NWN>
__global__ void __launch_bounds__(512, 2)
gemm_x6_kernel(const X6Params p) {
  using G = X6Geo<MT, NWN>;
  constexpr int MTL = G::MTL;
  // four arrays: hipcc orders LDS accesses against pending LDS-DMA per ARRAY, so the x planes (written by ds_write) and the W
  // planes (written by LDS-DMA) of the two stages are separate objects
  __shared__ __attribute__((aligned(1024))) unsigned char sx0[3 * G::APLANE];
  __shared__ __attribute__((aligned(1024))) unsigned char sx1[3 * G::APLANE];
  __shared__ __attribute__((aligned(1024))) unsigned char sw0[3 * G::WPLANE];
  __shared__ __attribute__((aligned(1024))) unsigned char sw1[3 * G::WPLANE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % G::NWM, wn = wave / G::NWM;
  const bool grp_a = wave < 4;                                       // (wave-uniform)
  const int K = p.K;
  const int nk = p.nk;                                               // even, >= 2 (W is zero-padded to a multiple of 64)
  const int64_t kw = (int64_t)nk * 32;                               // row stride of the W planes (elements)

  // ---- x: lane -> (row within 8, 16-B chunk); instruction i covers rows wave * 8 MTL + 8 i .. + 7 of the tile
  const int arow = lane >> 3, ac = lane & 7;
  int a_lds[MTL];
#pragma unroll
  for (int i = 0; i < MTL; ++i) {
    const int r = wave * (8 * MTL) + i * 8 + arow;
    a_lds[i] = r * 64 + (((ac >> 1) ^ x6_swz(r)) << 4) + (ac & 1) * 8;
  }
  // ---- W (group A): six 1-KiB pieces per wave and k-step; piece P = wave * 6 + t: plane P >> 3, rows 16 (P & 7) .. + 15.
  // The lane's part of the source offset is the same for every piece (row lane >> 2 of the piece, chunk q: the swizzle term
  // depends on (row >> 2) & 3 = (lane >> 4) & 3 only); plane and row block are wave-uniform.
  const uint32_t w_lane = (uint32_t)((lane >> 2) * kw * 2 + (((lane & 3) ^ x6_swz(lane >> 2)) << 4));
  const unsigned char* wbytes = reinterpret_cast<const unsigned char*>(p.w3);

  // ---- fragment addresses (bytes inside a stage)
  const int frow = lane & 15, fq = lane >> 4;
  const int fsw = (fq ^ x6_swz(frow)) << 4;
  const int offx = (wm * (16 * MT) + frow) * 64 + fsw;             // + plane * APLANE + j * 1024
  const int offw = (wn * 64 + frow) * 64 + fsw;                    // + plane * WPLANE + i * 1024

  // persistent tile loop; XCD-aware: the workgroups of one XCD (blockIdx.x & 7) take consecutive tiles, i.e. the feature tiles
  // of one row tile, in every round
  const int64_t grid = gridDim.x;
  int64_t slot = blockIdx.x;
  if ((grid & 7) == 0) slot = (int64_t)(blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3);

  f32x4 acc[4][MT];
  f32x4 av[MTL];
  bf16x8 xf[3][MT], wf[3][4];
  const float* xrow[MTL];
  const unsigned char* wtile = wbytes;
  int64_t m0 = 0;
  int n0 = 0;

  int64_t m_end = p.M;
  auto tile_setup = [&](int64_t tile) {
    m0 = (tile / p.tiles_n) * G::BM;
    n0 = (int)(tile % p.tiles_n) * G::BN;
    int64_t w_row0 = n0;
    if (p.tiles != nullptr) {                                      // grouped launch: this row tile's rows and weight block
      const int32_t* tt = p.tiles + 3 * (tile / p.tiles_n);
      m0 = tt[0];
      m_end = m0 + tt[1];
      w_row0 = (int64_t)tt[2] + n0;
    }
#pragma unroll
    for (int i = 0; i < MTL; ++i) {
      int64_t m = m0 + wave * (8 * MTL) + i * 8 + arow;
      if (m >= m_end) m = m_end - 1;                               // tail rows read the last valid row, never stored
      if (p.gather != nullptr) m = p.gather[m];                    // the pair's input row
      xrow[i] = p.x + m * K + ac * 4;
    }
    wtile = wbytes + w_row0 * kw * 2;
  };
  // x (kt + 2) is requested at the END of a group's non-MFMA half, travels during that group's MFMA phase and is split at the
  // top of its next non-MFMA half.  What keeps hipcc from undoing that: the requests cannot cross the s_barrier that follows
  // them (sched_barrier + the barrier's memory semantics); pin_a() -- an empty asm that "rewrites" the registers, placed behind
  // the NEXT step's barrier -- keeps everything computed from the data below that barrier (unpinned, the split arithmetic is
  // hoisted up to the loads and both sink into the MFMA phase behind a vmcnt(0)); the barriers in between wait with
  // vmcnt(MT), which leaves exactly these MT requests in flight.
  auto load_a = [&](int kt) {
    // (a k-step of W's zero padding: any valid address will do, the products vanish)
    const int ko = kt * 32 + 32 <= K ? kt * 32 : K - 32;
#pragma unroll
    for (int i = 0; i < MTL; ++i) av[i] = *reinterpret_cast<const f32x4*>(xrow[i] + ko);
  };
  auto pin_a = [&]() {
#pragma unroll
    for (int i = 0; i < MTL; ++i) asm volatile("" : "+v"(av[i]));
  };
  auto dma_w = [&](unsigned char* st, int kt) {
    constexpr int PPW = G::WPIECES / 4;                                        // pieces per wave of group A
    constexpr int RB = G::BN / 16;                                             // 16-row blocks of the W tile
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
      const int P = (wave & 3) * PPW + t;                                      // (wave-uniform)
      const int plane = P / RB, rblk = P % RB;
      const unsigned char* src = wtile + (int64_t)plane * p.w_plane_bytes + (int64_t)rblk * 16 * kw * 2 + kt * 64;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w_lane),
                                       (__attribute__((address_space(3))) void*)(st + plane * G::WPLANE + rblk * 1024), 16, 0, 0);
    }
  };
  auto store_a = [&](unsigned char* st) {
    pin_a();
#pragma unroll
    for (int i = 0; i < MTL; ++i) {
      uint2 h, m, l;
      x6_split4(make_float4(av[i][0], av[i][1], av[i][2], av[i][3]), h, m, l);
      *reinterpret_cast<uint2*>(st + a_lds[i]) = h;
      *reinterpret_cast<uint2*>(st + G::APLANE + a_lds[i]) = m;
      *reinterpret_cast<uint2*>(st + 2 * G::APLANE + a_lds[i]) = l;
    }
  };
  auto read_frags = [&](const unsigned char* stx, const unsigned char* stw) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int j = 0; j < MT; ++j) xf[pl][j] = *reinterpret_cast<const bf16x8*>(stx + offx + pl * G::APLANE + j * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[pl][i] = *reinterpret_cast<const bf16x8*>(stw + offw + pl * G::WPLANE + i * 1024);
    }
  };
  auto mfma_all = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][i], xf[0][j], acc[i][j], 0, 0, 0);     // l h
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], xf[2][j], acc[i][j], 0, 0, 0);     // h l
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], xf[1][j], acc[i][j], 0, 0, 0);     // m m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], xf[0][j], acc[i][j], 0, 0, 0);     // m h
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], xf[1][j], acc[i][j], 0, 0, 0);     // h m
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], xf[0][j], acc[i][j], 0, 0, 0);     // h h
      }
  };
  // Barrier forms.  FULL: every memory operation of this wave done (its LDS-DMA pieces have landed, its LDS writes are
  // visible).  LDS: only the LDS operations -- group A leaves its LDS-DMA of the next W tile in flight across the half-step.
  // (sched_barrier: the MFMAs are register-only instructions, hipcc's scheduler would otherwise move them across the
  // s_barrier into the other half-step, where the SIMD's partner wave has ITS MFMA phase)
  // Priority: the wave in its read / split half runs beside its SIMD partner's back-to-back MFMAs and, at equal priority, gets
  // about ONE instruction issued per MFMA (16 cycles): its ~180 instructions took ~2900 cycles against the partner's 1536
  // (s_memtime stamps, probe build).  Raised to priority 1 for that half it issues at once; the MFMA stream needs one issue slot
  // per 16 cycles and loses nothing.
  auto prio_hi = [&]() { __builtin_amdgcn_s_setprio(1); };
  auto prio_lo = [&]() { __builtin_amdgcn_s_setprio(0); };
  int stamp_n = 0;
  (void)stamp_n;
  auto bar_full = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    X6_WAIT_ALL();
    X6_STAMP();
    __builtin_amdgcn_s_barrier();
    X6_STAMP();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar_lds = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    X6_WAIT_LGKM0();
    X6_STAMP();
    __builtin_amdgcn_s_barrier();
    X6_STAMP();
    __builtin_amdgcn_sched_barrier(0);
  };
  // all but this wave's MTL youngest vector-memory operations done (its x requests stay in flight, its LDS-DMA has landed)
  auto bar_keep = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    X6_WAIT(MTL, 0);
    X6_STAMP();
    __builtin_amdgcn_s_barrier();
    X6_STAMP();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int64_t n_tiles = p.n_wg;
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // Both groups pass 2 nk + 1 barriers per tile: B1 (t) "stage t complete", B2 (t) between the half-steps, and one after the
  // last step ("every fragment is in registers, both stages are free").
  int64_t tile = slot;
  if (grp_a) {
    // ------------------------------------------------------------------ group A: [fragment reads, split | W DMA, x request, MFMA]
    // (the LDS-DMA pieces and the x requests leave at the TOP of the MFMA phase: among MFMAs an LDS-DMA piece costs ~60 issue
    // cycles, inside the read / split half 100-185, and six of them in front of the fragment reads made that half -- a chain of
    // latencies, not of throughputs -- twice as long as the partner's MFMA phase)
    auto mfma_phase = [&](unsigned char* stw_next, int kt_next_w, int kt_next_x) {
      if (kt_next_w >= 0) dma_w(stw_next, kt_next_w);
      // (the x requests must be the wave's YOUNGEST vector-memory operations: the next barrier waits with vmcnt(MTL))
      __builtin_amdgcn_sched_barrier(0);
      if (kt_next_x >= 0) load_a(kt_next_x);
      // no MFMA above this point (each reads an accumulator), no request below it
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]) : : "memory");
      mfma_all();
    };
    for (; tile < n_tiles; tile += grid) {
      tile_setup(tile);
      zero_acc();
      dma_w(sw0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_a(0);
      store_a(sx0);
      load_a(1);
      for (int kt = 0; kt + 2 < nk; kt += 2) {
        bar_keep();                              // B1 (kt): W (kt) has landed
        prio_hi();
        read_frags(sx0, sw0);
        store_a(sx1);                            // x (kt + 1)
        prio_lo();
        bar_lds();                               // B2 (kt)
        mfma_phase(sw1, kt + 1, kt + 2);         // W (kt + 1) -> stage 1 (free since B1 (kt)), wanted at B1 (kt + 1)
        bar_keep();                              // B1 (kt + 1)
        prio_hi();
        read_frags(sx1, sw1);
        store_a(sx0);                            // x (kt + 2)
        prio_lo();
        bar_lds();                               // B2 (kt + 1)
        mfma_phase(sw0, kt + 2, kt + 3);
      }
      bar_keep();                                // B1 (nk - 2)
      prio_hi();
      read_frags(sx0, sw0);
      store_a(sx1);                              // x (nk - 1)
      prio_lo();
      bar_lds();                                 // B2 (nk - 2)
      mfma_phase(sw1, nk - 1, -1);
      bar_full();                                // B1 (nk - 1)
      prio_hi();
      read_frags(sx1, sw1);
      prio_lo();
      bar_lds();                                 // B2 (nk - 1)
      mfma_all();
      bar_full();                                // end of tile
      x6_epilogue<GELU, MT>(p, acc, x6_ep_region<MT, NWN>(sx1, sw1, wave), m0 + wm * (16 * MT), n0 + wn * 64, lane, m_end);
    }
  } else {
    // ------------------------------------------------------------------ group B: [MFMA of the step before | reads, split, x request]
    for (; tile < n_tiles; tile += grid) {
      tile_setup(tile);
      zero_acc();
      load_a(0);
      store_a(sx0);
      load_a(1);
      bar_keep();                                // B1 (0)
      bar_keep();                                // B2 (0)
      prio_hi();
      read_frags(sx0, sw0);
      store_a(sx1);                              // x (1)
      load_a(2);
      prio_lo();
      for (int kt = 1; kt + 2 < nk; kt += 2) {
        bar_keep();                              // B1 (kt), kt odd: stage 1
        mfma_all();                              // step kt - 1
        bar_keep();                              // B2 (kt)
        prio_hi();
        read_frags(sx1, sw1);
        store_a(sx0);                            // x (kt + 1)
        load_a(kt + 2);
        prio_lo();
        bar_keep();                              // B1 (kt + 1): stage 0
        mfma_all();                              // step kt
        bar_keep();                              // B2 (kt + 1)
        prio_hi();
        read_frags(sx0, sw0);
        store_a(sx1);                            // x (kt + 2)
        load_a(kt + 3);                          // (kt + 3 <= nk - 1; at nk - 1 ... nk the requests repeat the last block)
        prio_lo();
      }
      bar_keep();                                // B1 (nk - 1)
      mfma_all();                                // step nk - 2
      bar_full();                                // B2 (nk - 1)
      prio_hi();
      read_frags(sx1, sw1);
      prio_lo();
      bar_full();                                // end of tile
      mfma_all();                                // step nk - 1
      x6_epilogue<GELU, MT>(p, acc, x6_ep_region<MT, NWN>(sx1, sw1, wave), m0 + wm * (16 * MT), n0 + wn * 64, lane, m_end);
    }
  }
}

// (N, K) f32 -> (3, N, Kp) bf16 planes, Kp = K rounded up to a multiple of 64, the padding zero
__global__ void __launch_bounds__(256)
x6_pack_kernel(uint16_t* __restrict__ w3, const float* __restrict__ w, int64_t n_rows, int K, int Kp) {
  const int64_t per_row = Kp / 4;
  const int64_t total4 = n_rows * per_row;
  const int64_t plane_elems = n_rows * Kp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / per_row;
    const int k = (int)(i % per_row) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) v = *reinterpret_cast<const float4*>(w + r * K + k);
    uint2 h, m, l;
    x6_split4(v, h, m, l);
    *reinterpret_cast<uint2*>(w3 + i * 4) = h;
    *reinterpret_cast<uint2*>(w3 + plane_elems + i * 4) = m;
    *reinterpret_cast<uint2*>(w3 + 2 * plane_elems + i * 4) = l;
  }
}

}  // namespace

static int g_x6_mt = 0;      // 0: tile shape chosen per launch; probe: 1 .. 4 = (64 MT) x 128, 12 / 14 = 64 / 128 rows x 256

extern "C" {

void hfl_internal_set_x6_mt(int v) { g_x6_mt = v; }
#ifdef HFL_X6_STAMPS
int hfl_internal_x6_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x6_stamps), sizeof(g_x6_stamps));
}
#endif

int64_t hfl_linear_x6_padded_k(int64_t in_features) { return (in_features + 63) / 64 * 64; }

int hfl_linear_x6_pack(uint16_t* w3, const float* w, int64_t out_features, int64_t in_features, hfl_stream_t stream) {
  if (w3 == nullptr || w == nullptr || out_features <= 0 || in_features <= 0 || in_features % 32 != 0) return HFL_EINVAL;
  const int64_t kp = hfl_linear_x6_padded_k(in_features);
  const int64_t need = hfl_cdiv(out_features * kp / 4, 256);
  x6_pack_kernel<<<(int)(need < 4096 ? need : 4096), 256, 0, static_cast<hipStream_t>(stream)>>>(w3, w, out_features,
                                                                                                 (int)in_features, (int)kp);
  HFL_RETURN_LAST_ERROR();
}

static int x6_launch(float* out, const float* x, const uint16_t* w3, const float* bias, const float* residual,
                     const float* row_scale, int64_t n_rows, int in_features, int out_features, int gelu, hfl_stream_t stream,
                     const int32_t* tiles, int64_t n_tiles, const int32_t* gather, int64_t w_rows) {
  if (n_rows < 0 || in_features <= 0 || out_features <= 0) return HFL_EINVAL;
  const bool grouped = tiles != nullptr;
  const bool narrow = grouped && out_features == 64;                 // W blocks padded to 128 rows by the caller
  if (in_features % 32 != 0 || (out_features % 128 != 0 && !narrow)) return HFL_EINVAL;
  if (out == nullptr || x == nullptr || w3 == nullptr) return HFL_EINVAL;
  if (gelu && (residual != nullptr || row_scale != nullptr)) return HFL_EINVAL;
  const int64_t kp = hfl_linear_x6_padded_k(in_features);
  if (!grouped) w_rows = out_features;
  if (w_rows * kp * 2 >= ((int64_t)1 << 31)) return HFL_ECAPACITY;          // 32-bit offsets inside a W plane
  if (n_rows == 0 || (grouped && n_tiles == 0)) return HFL_OK;
  X6Params p;
  p.out = out; p.x = x; p.w3 = w3; p.bias = bias; p.residual = residual; p.row_scale = row_scale;
  p.tiles = tiles; p.gather = gather; p.n_valid = out_features;
  p.M = n_rows; p.N = out_features; p.K = in_features;
  p.nk = (int)(kp / 32);
  p.w_plane_bytes = w_rows * kp * 2;
  // Tile shape per launch.  Row tiles of 16 MT x (8 / NWN) rows, one persistent workgroup per CU: a launch costs
  // ceil(tiles / CUs) rounds of (rows + c) units; a taller tile re-uses the W tile over more rows, a shorter one wastes less of
  // the last round.  (Grouped launches: the caller's tile table is cut for 128-row tiles.)
  const int cus = hfl_stream_cus(static_cast<hipStream_t>(stream));
  int shape = g_x6_mt;                   // probe: 1 .. 4 = 64 MT x 128, 12 / 14 = 64 / 128 x 256
  static const int kShapes[6][3] = {{4, 2, 256}, {3, 2, 192}, {2, 2, 128}, {1, 2, 64}, {4, 4, 128}, {2, 4, 64}};   // MT, NWN, rows
  int pick = -1;
  if (grouped) pick = 2;
  else if (shape >= 1 && shape <= 4) pick = 4 - shape;
  else if (shape == 14) pick = 4;
  else if (shape == 12) pick = 5;
  if (pick >= 4 && out_features % 256 != 0) pick = -1;
  if (pick < 0) {
    double best = 0.0;
    for (int c = 0; c < 6; ++c) {
      const int bn = kShapes[c][1] * 64, bm = kShapes[c][2];
      if (out_features % bn != 0) continue;
      const int64_t tiles_c = hfl_cdiv(n_rows, bm) * (out_features / bn);
      // units of 64 rows x 128 features (measured: a round of 64 MT x 128 tiles costs ~(MT + 0.6) units, a 256-feature tile as
      // much as the 128-feature tile of twice its rows; ties go to the narrow tile, listed first)
      const double cost = (double)hfl_cdiv(tiles_c, cus) * ((bm / 64.0) * (bn / 128.0) + 0.6);
      if (pick < 0 || cost < best) { best = cost; pick = c; }
    }
  }
  const int mt = kShapes[pick][0], nwn = kShapes[pick][1];
  p.tiles_n = narrow ? 1 : out_features / (nwn * 64);
  p.n_wg = (grouped ? n_tiles : hfl_cdiv(n_rows, kShapes[pick][2])) * p.tiles_n;
  if (p.n_wg > 0x7fffffffLL) return HFL_ECAPACITY;
  const unsigned grid = (unsigned)(p.n_wg < cus ? p.n_wg : cus);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define HFL_X6_LAUNCH(G, M, W) gemm_x6_kernel<G, M, W><<<grid, 512, 0, s>>>(p)
#define HFL_X6_SHAPES(G)                                                                            \
  if (nwn == 4) { if (mt == 4) HFL_X6_LAUNCH(G, 4, 4); else HFL_X6_LAUNCH(G, 2, 4); }                \
  else if (mt == 4) HFL_X6_LAUNCH(G, 4, 2); else if (mt == 3) HFL_X6_LAUNCH(G, 3, 2);               \
  else if (mt == 2) HFL_X6_LAUNCH(G, 2, 2); else HFL_X6_LAUNCH(G, 1, 2);
  if (gelu) { HFL_X6_SHAPES(1) } else { HFL_X6_SHAPES(0) }
#undef HFL_X6_SHAPES
#undef HFL_X6_LAUNCH
  HFL_RETURN_LAST_ERROR();
}

int hfl_linear_x6(float* out, const float* x, const uint16_t* w3, const float* bias, const float* residual,
                  const float* row_scale, int64_t n_rows, int in_features, int out_features, int gelu, hfl_stream_t stream) {
  return x6_launch(out, x, w3, bias, residual, row_scale, n_rows, in_features, out_features, gelu, stream, nullptr, 0, nullptr, 0);
}

/* Grouped form with the gather done by the tile loader (see include/hotformerloc_hip.h): the per-tap products of an octree
 * convolution over its live (row, tap) pairs at matched precision. */
int hfl_linear_x6_grouped_gather(float* out, const float* x, const int32_t* gather, const uint16_t* w3, int64_t w_rows,
                                 const int32_t* tiles, int64_t n_tiles, int64_t n_rows, int in_features, int out_features,
                                 hfl_stream_t stream) {
  if (tiles == nullptr || gather == nullptr || n_tiles < 0 || w_rows <= 0 || w_rows % 128 != 0) return HFL_EINVAL;
  return x6_launch(out, x, w3, nullptr, nullptr, nullptr, n_rows, in_features, out_features, 0, stream, tiles, n_tiles, gather,
                   w_rows);
}

}  // extern "C"
