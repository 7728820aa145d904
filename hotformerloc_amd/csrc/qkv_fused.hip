// LayerNorm -> qkv projection of a transformer block as ONE kernel on the bf16 matrix cores of gfx950, writing the window
// kernel's operand layout:
//
//     qkv (M, 3C as fp16 (hi, lo)) = attention_operand( LayerNorm(x) Wqkv^T + b ),  queries pre-multiplied by q_scale
//
// Replaces norm1 -> attention.qkv of every transformer block of the reference (models/octformer_backbone.py:70,
// models/hotformerloc_backbone.py:213-216), which ran as two launches (hfl_layer_norm_split2, hfl_linear_x3_qkv): the
// normalised rows crossed HBM once each way as 4 B per element (2 of the 6.5 M*C*4-byte units those launches move) and the
// GEMM re-read them per 128-feature column tile through L2.  Here a row tile's LayerNorm output lives in registers as MFMA
// B fragments for all 3C output features.
//
// Built like csrc/mlp_fused.hip's first GEMM (same arithmetic: bf16 (hi, lo) operands, x_lo w_hi + x_hi w_lo + x_hi w_hi, fp32
// accumulation): a 512-lane workgroup owns 128 (C = 256) or 256 (C = 128) rows per pass, wave w keeps NT tiles of 16 rows;
// the weight streams through a 4-slot LDS ring in stages of 32 output features (C rows x 128 B, laid out once per parameter
// by hfl_qkv_fused_pack), one s_barrier and one counted s_waitcnt per stage, the refill of a slot issued piecewise between
// the MFMA steps.  Epilogue per stage: bias, query scale, fp16 (hi, lo) split (hi = RTZ(v), lo = RTZ(v - hi), exactly as
// gemm_x3's EPI 2), then through a 2-KiB block of LDS per wave and row tile so that the 16 x 128 B (two heads x [16 hi | 16
// lo]) leave as whole 128-B lines, eight lanes per row.
#include "hfl_common.h"
#include "x3_math.h"
#include "stage_stream.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct QkvFusedParams {
  unsigned char* out;         // (M, 3C) attention operand rows: 3C * 4 B per row
  const float* x;             // (M, C) f32
  HflRowSeg xseg;             // ... or its rows in several arrays (xseg.n > 1; x = xseg.ptr[0])
  const float* gamma;         // (C)
  const float* beta;          // (C)
  const unsigned char* pack;  // hfl_qkv_fused_pack image of Wqkv
  const float* bias;          // (3C)
  int64_t M;
  float eps;
  float q_scale;
  int n_tiles;                // ceil(M / 16)
  int stagger;
  int stagger_groups;
  // Tail split (csrc/mlp_fused.hip): `full_passes` whole passes per workgroup; the tiles left over are cut into `tail_sets`
  // sets of up to one pass, each computed by `tail_parts` workgroups that share the OUTPUT features: part k runs the stages
  // [k, k + 1) * NST / tail_parts (32 features each) of its set -- no reduction, the parts write disjoint columns.
  int full_passes;
  int tail_tile0;
  int tail_sets;
  int tail_parts;
};

// LDS accesses of the epilogue as inline asm: while an LDS-DMA is in flight hipcc puts `s_waitcnt vmcnt(0)` in front of every
// LDS access it knows about (it cannot tell the DMA's destination from the address read), which would expose the weight
// stream's latency once per stage.  One wave's LDS operations execute in order; HFL_DS_WAIT names what was read.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define HFL_DS_READ128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#define HFL_DS_WRITE64(addr, val) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(val) : "memory")
#define HFL_DS_WAIT2(a, b) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b))

// staging blocks (2 KiB, one row tile) per wave: as many of the wave's NT tiles as the 160 KiB of LDS leave room for
constexpr int qkv_staging_blocks(int C, int NT, int W) {
  int nb = NT;
  while (nb > 1 && (size_t)4 * C * 128 + (size_t)C * 20 + (size_t)W * nb * 2048 > (size_t)160 * 1024) nb /= 2;
  return nb;
}

template <int C, int NT, int W, int PF>
__global__ void __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(W / 4, W / 4)))
ln_qkv_fused_kernel(const QkvFusedParams p) {
  static_assert(PF == 2 || PF == 3, "stages in flight ahead of the one being consumed (csrc/mlp_fused.hip)");
  constexpr int KS = C / 32;               // k-steps
  constexpr int NST = 3 * C / 32;          // stages per pass: 32 output features each
  constexpr int SPR = C / 32;              // stages per region (Q, K, V)
  constexpr int STAGE_B = C * 128;         // bytes of one stage: C rows x 128 B
  constexpr int NSLOT = 4;
  constexpr int DPW = STAGE_B / 1024 / W;  // LDS-DMA instructions per wave and stage
  constexpr int TPP = W * NT;              // 16-row tiles per pass
  constexpr int NB = qkv_staging_blocks(C, NT, W);     // staging blocks per wave
  constexpr int ROUNDS = NT / NB;          // epilogue rounds per stage (NB tiles each)
  static_assert(NT % NB == 0 && 3 + 2 * ROUNDS <= KS + 1, "epilogue schedule");
  constexpr int PPH = DPW / 2;             // pieces per half stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* bs = reinterpret_cast<float*>(smem + NSLOT * STAGE_B);        // (3C) bias | (C) gamma | (C) beta
  float* gms = bs + 3 * C;
  float* bts = gms + C;
  unsigned char* stg_all = reinterpret_cast<unsigned char*>(bts + C);  // [W waves][NB][16 rows][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  unsigned char* stg = stg_all + wave * (NB * 2048);

  const int G = gridDim.x, g = blockIdx.x;
  int tile0, tile_end;
  int tail_lo = 0, tail_hi = 0, ts0 = 0, tnst = 0;       // tail work: tiles [tail_lo, tail_hi), stages [ts0, ts0 + tnst)
  if (p.tail_sets > 0) {
    tile0 = g * p.full_passes * TPP;
    tile_end = tile0 + p.full_passes * TPP;
    if (g < p.tail_sets * p.tail_parts) {
      const int tset = g % p.tail_sets;
      tail_lo = p.tail_tile0 + tset * TPP;
      tail_hi = tail_lo + TPP < p.n_tiles ? tail_lo + TPP : p.n_tiles;
      tnst = NST / p.tail_parts;
      ts0 = (g / p.tail_sets) * tnst;
    }
  } else {
    const int base = p.n_tiles / G, extra = p.n_tiles % G;
    tile0 = g * base + (g < extra ? g : extra);
    tile_end = tile0 + base + (g < extra ? 1 : 0);
  }
  int s0 = 0, nst = NST;                    // current pass: first stage of the pack, stages (even)

  for (int i = tid; i < 3 * C / 4; i += W * 64) reinterpret_cast<float4*>(bs)[i] = reinterpret_cast<const float4*>(p.bias)[i];
  for (int i = tid; i < C / 4; i += W * 64) {
    reinterpret_cast<float4*>(gms)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
    reinterpret_cast<float4*>(bts)[i] = reinterpret_cast<const float4*>(p.beta)[i];
  }
  __syncthreads();

  const uint32_t lane_off = (uint32_t)lane * 16u;
  uint32_t seq = 0;                         // stages acquired so far (all passes): slot = seq % NSLOT
  auto issue = [&](int n, uint32_t slot) {
    const unsigned char* s = p.pack + (int64_t)(s0 + n) * STAGE_B + wave * (DPW * 1024);
    unsigned char* d = smem + slot * STAGE_B + wave * (DPW * 1024);
#pragma unroll
    for (int i = 0; i < DPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024 + lane_off),
                                       (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
  };
  // stage protocol of csrc/mlp_fused.hip: at acquire(n) stage n has landed for every wave and the slot of stage n - 2 is
  // free; its refill with stage n + 2 goes out piece by piece between the MFMA steps of stage n.  The epilogue's stores
  // sit on the same counter: loads return in order among themselves, so "at most DPW operations outstanding" can only be the
  // youngest DPW loads and / or stores -- stage n has landed whatever the stores do (they may complete out of order with the
  // loads: a count of DPW + stores would NOT be safe).  The price: outstanding stores eat into the look-ahead of the ring.
  int dma_n = 0;
  uint32_t dma_slot = 0;
  auto acquire = [&](int n) -> const unsigned char* {
    if (n + PF - 1 < nst) HFL_WAIT_VM((PF - 1) * DPW);
    else if (n + 1 < nst) HFL_WAIT_VM((PF - 2) * DPW > 0 ? (PF - 2) * DPW : 0);
    else HFL_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    dma_n = n + PF;
    dma_slot = (seq + PF) % NSLOT;
    const unsigned char* st = smem + (seq % NSLOT) * STAGE_B;
    ++seq;
    return st;
  };
  auto dma_piece = [&](int i) {
    if (dma_n < nst)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(p.pack + (int64_t)(s0 + dma_n) * STAGE_B + wave * (DPW * 1024) +
                                                          i * 1024 + lane_off),
          (__attribute__((address_space(3))) void*)(smem + dma_slot * STAGE_B + wave * (DPW * 1024) + i * 1024), 16, 0, 0);
  };
  // A fragment of a 16-row block of a stage: row = block * 16 + fr, hi chunk fq, lo chunk 4 + fq (slot t of row r at t ^ ((r >> 1) & 7))
  const int off_hi = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), off_lo = off_hi ^ 64;
  // output staging: row fr of the tile = 128 B = heads (2 k, 2 k + 1) x [16 hi | 16 lo] fp16; this lane's 4 features of head i:
  // hi at i * 64 + 8 fq, lo at i * 64 + 32 + 8 fq; 16-B slot s of row r stored at slot s ^ (r & 7)
  const int stg_row = fr * 128, stg_x = fr & 7;
  // read back: lane l takes 16-B slot l & 7 of row l >> 3 (and of row 8 + (l >> 3) in the second instruction)
  const int rd_row = lane >> 3, rd_slot = lane & 7;

  {
    const int naps = (int)(blockIdx.x % (unsigned)p.stagger_groups) * p.stagger;
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(63);
  }
  bool tail = false;
  for (;;) {
    if (tile0 >= tile_end) {
      if (tail || tnst == 0) break;
      tail = true;                          // the last pass: this workgroup's columns of a tail set
      tile0 = tail_lo;
      tile_end = tail_hi;
      s0 = ts0;
      nst = tnst;
    }
    const int ntile = tile_end - tile0 < TPP ? tile_end - tile0 : TPP;
    // ---- LayerNorm of this wave's rows -> B-operand fragments (lane: row fr of the tile, channels 32 ks + 8 fq + j)
    bf16x8 xh[NT][KS], xl[NT][KS];
    bool have[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = t * W + wave;
      have[t] = k < ntile;
      int64_t r = (int64_t)(tile0 + k) * 16 + fr;
      if (r >= p.M) r = p.M - 1;
      if (!have[t]) r = 0;
      const float* xr = (p.xseg.n > 1 ? hfl_seg_row(p.xseg, r, C) : p.x + r * C) + fq * 8;
      float4 a[KS][2];
      float sum = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a[ks][0] = *reinterpret_cast<const float4*>(xr + ks * 32);
        a[ks][1] = *reinterpret_cast<const float4*>(xr + ks * 32 + 4);
        sum += ((a[ks][0].x + a[ks][0].y) + (a[ks][0].z + a[ks][0].w)) + ((a[ks][1].x + a[ks][1].y) + (a[ks][1].z + a[ks][1].w));
      }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / (float)C);
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a[ks][h].x -= mean; a[ks][h].y -= mean; a[ks][h].z -= mean; a[ks][h].w -= mean;
          sq += (a[ks][h].x * a[ks][h].x + a[ks][h].y * a[ks][h].y) + (a[ks][h].z * a[ks][h].z + a[ks][h].w * a[ks][h].w);
        }
      sq += __shfl_xor(sq, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / (float)C) + p.eps);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 gm = *reinterpret_cast<const float4*>(gms + ks * 32 + fq * 8 + h * 4);
          const float4 bt = *reinterpret_cast<const float4*>(bts + ks * 32 + fq * 8 + h * 4);
          const f32x2 v01 = {fmaf(a[ks][h].x * rstd, gm.x, bt.x), fmaf(a[ks][h].y * rstd, gm.y, bt.y)};
          const f32x2 v23 = {fmaf(a[ks][h].z * rstd, gm.z, bt.z), fmaf(a[ks][h].w * rstd, gm.w, bt.w)};
          x3_split_pair(v01, hi[2 * h], lo[2 * h]);
          x3_split_pair(v23, hi[2 * h + 1], lo[2 * h + 1]);
        }
        xh[t][ks] = __builtin_bit_cast(bf16x8, (u32x4){hi[0], hi[1], hi[2], hi[3]});
        xl[t][ks] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], lo[2], lo[3]});
      }
    }
    const bool active = have[0];
    // (the first two stages start moving only now: while an LDS-DMA is in flight hipcc waits vmcnt(0) before every use of an
    // ordinary load's result -- the row loads above would become dependent round trips)
    __builtin_amdgcn_s_barrier();
    issue(0, seq % NSLOT);
    issue(1, (seq + 1) % NSLOT);
    if (PF == 3 && nst > 2) issue(2, (seq + 2) % NSLOT);

    // The epilogue of the PREVIOUS stage rides between the k-steps of the current one; the fragment waits of the k-loop
    // (`s_waitcnt lgkmcnt(0)` after every k-step and at the start of a half) are what its own LDS round trips wait on, so it
    // adds no wait of its own.  NB row tiles per round (one staging block each):
    //   slot 0: request the bias | slot 1 + 2 r: bias, scale, fp16 split, write round r's tiles to the staging blocks (and
    //   store round r - 1) | slot 2 + 2 r: request them back a row per eight lanes | slot 3 + 2 r: store whole 128-B lines
    f32x4 ep_b0, ep_b1;
    u32x4 ep_va[NB], ep_vb[NB];
    auto ep_write = [&](f32x4 (&hp)[2][NT], int pn, int r) {
      const bool is_q = pn < SPR;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int t = r * NB + u;
        const uint32_t sb = (uint32_t)(uintptr_t)(stg + u * 2048);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 b = i == 0 ? ep_b0 : ep_b1;
          float v0 = hp[i][t][0] + b[0], v1 = hp[i][t][1] + b[1], v2 = hp[i][t][2] + b[2], v3 = hp[i][t][3] + b[3];
          if (is_q) { v0 *= p.q_scale; v1 *= p.q_scale; v2 *= p.q_scale; v3 *= p.q_scale; }
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(v0, v1), h23 = __builtin_amdgcn_cvt_pkrtz(v2, v3);
          const auto l01 = __builtin_amdgcn_cvt_pkrtz(v0 - (float)h01[0], v1 - (float)h01[1]);
          const auto l23 = __builtin_amdgcn_cvt_pkrtz(v2 - (float)h23[0], v3 - (float)h23[1]);
          const u32x2 hi = {__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23)};
          const u32x2 lo = {__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23)};
          HFL_DS_WRITE64(sb + (uint32_t)(stg_row + (((i * 4 + (fq >> 1)) ^ stg_x) << 4) + (fq & 1) * 8), hi);
          HFL_DS_WRITE64(sb + (uint32_t)(stg_row + (((i * 4 + 2 + (fq >> 1)) ^ stg_x) << 4) + (fq & 1) * 8), lo);
        }
      }
    };
    auto ep_read = [&]() {
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const uint32_t sb = (uint32_t)(uintptr_t)(stg + u * 2048);
        HFL_DS_READ128(ep_va[u], sb + (uint32_t)(rd_row * 128 + ((rd_slot ^ (rd_row & 7)) << 4)));
        HFL_DS_READ128(ep_vb[u], sb + (uint32_t)((8 + rd_row) * 128 + ((rd_slot ^ (rd_row & 7)) << 4)));
      }
    };
    auto ep_store = [&](int pn, int r) {
      const int64_t col = (int64_t)(pn / SPR) * (C * 4) + (int64_t)(pn % SPR) * 128;     // byte offset inside an output row
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        const int t = r * NB + u;
        asm volatile("" : "+v"(ep_va[u]), "+v"(ep_vb[u]));        // valid from here on: the k-loop's wait is behind us
        const int64_t r0 = (int64_t)(tile0 + t * W + wave) * 16;
        if (have[t] && r0 + rd_row < p.M)
          *reinterpret_cast<u32x4*>(p.out + (r0 + rd_row) * (int64_t)(3 * C * 4) + col + rd_slot * 16) = ep_va[u];
        if (have[t] && r0 + 8 + rd_row < p.M)
          *reinterpret_cast<u32x4*>(p.out + (r0 + 8 + rd_row) * (int64_t)(3 * C * 4) + col + rd_slot * 16) = ep_vb[u];
      }
    };
    auto ep_phase = [&](int slot, f32x4 (&hp)[2][NT], int pn) {
      if (slot == 0) {
        const uint32_t baddr = (uint32_t)(uintptr_t)(bs + pn * 32 + fq * 4);
        HFL_DS_READ128(ep_b0, baddr);
        HFL_DS_READ128(ep_b1, baddr + 64u);
        return;
      }
      if (slot == 1) asm volatile("" : "+v"(ep_b0), "+v"(ep_b1));
#pragma unroll
      for (int r = 0; r <= ROUNDS; ++r) {
        if (r > 0 && slot == 1 + 2 * r) ep_store(pn, r - 1);            // (before this round's writes reuse the blocks: the
        if (r < ROUNDS && slot == 1 + 2 * r) ep_write(hp, pn, r);        //  values are in registers by now)
        if (r < ROUNDS && slot == 2 + 2 * r) ep_read();
      }
    };
    // one half of a stage: 32 output features x this wave's rows over k-steps [hf KS/2, (hf + 1) KS/2)
    auto gemm_half = [&](const unsigned char* st, auto hfc, f32x4 (&h)[2][NT], f32x4 (&hp)[2][NT], int pn, bool with_ep) {
      constexpr int hf = decltype(hfc)::value;
      if (hf == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) h[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const uint32_t ahi = (uint32_t)(uintptr_t)(st + off_hi) + hf * ((KS / 2) * 4096);
      const uint32_t alo = (uint32_t)(uintptr_t)(st + off_lo) + hf * ((KS / 2) * 4096);
      bf16x8 wf[2][4];
      HFL_LDS_READ4_FIRST(wf[0][0], wf[0][1], wf[0][2], wf[0][3], ahi, alo, 0, 2048);
      HFL_LDS_WAIT4(wf[0][0], wf[0][1], wf[0][2], wf[0][3]);
      hfl_static_for(std::make_integer_sequence<int, KS / 2>{}, [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        constexpr int ks = hf * (KS / 2) + kk;
        if constexpr (kk + 1 < KS / 2)
          HFL_LDS_READ4(wf[(kk + 1) & 1][0], wf[(kk + 1) & 1][1], wf[(kk + 1) & 1][2], wf[(kk + 1) & 1][3], ahi, alo,
                        (kk + 1) * 4096, (kk + 1) * 4096 + 2048, wf[kk & 1][0]);
        if constexpr (ks < 2 + 2 * ROUNDS) {
          if (with_ep) ep_phase(ks, hp, pn);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i], xl[t][ks], h[i][t], 0, 0, 0);
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i + 1], xh[t][ks], h[i][t], 0, 0, 0);
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i], xh[t][ks], h[i][t], 0, 0, 0);
          }
        if constexpr ((kk + 1) % ((KS / 2) / PPH) == 0) dma_piece(hf * PPH + (kk + 1) / ((KS / 2) / PPH) - 1);
        if constexpr (kk + 1 < KS / 2)
          HFL_LDS_WAIT4_AFTER(wf[(kk + 1) & 1][0], wf[(kk + 1) & 1][1], wf[(kk + 1) & 1][2], wf[(kk + 1) & 1][3], h[1][NT - 1]);
      });
    };
    // the last stage's epilogue, on its own (features 32 n + 16 i + 4 fq + r of row fr (tile t) in h[i][t][r])
    auto epilogue = [&](f32x4 (&h)[2][NT], int n) {
      const uint32_t baddr = (uint32_t)(uintptr_t)(bs + n * 32 + fq * 4);
      HFL_DS_READ128(ep_b0, baddr);
      HFL_DS_READ128(ep_b1, baddr + 64u);
      HFL_DS_WAIT2(ep_b0, ep_b1);
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        ep_write(h, n, r);
        ep_read();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ep_store(n, r);
      }
    };

    auto idle_pieces = [&]() {
#pragma unroll
      for (int i = 0; i < DPW; ++i) dma_piece(i);
    };
    f32x4 hA[2][NT], hB[2][NT];                 // accumulators of two consecutive stages
    const unsigned char* st = acquire(0);
    auto stage = [&](f32x4 (&h)[2][NT], f32x4 (&hp)[2][NT], int n) {       // n: stage of this pass; s0 + n: of the pack
      if (active) {
        gemm_half(st, std::integral_constant<int, 0>{}, h, hp, s0 + n - 1, n > 0);
        gemm_half(st, std::integral_constant<int, 1>{}, h, hp, s0 + n - 1, n > 0);
      } else {
        idle_pieces();
      }
      if (n + 1 < nst) st = acquire(n + 1);
    };
    static_assert(NST % 2 == 0, "stages come in pairs");
#pragma unroll 1
    for (int n = 0; n < nst; n += 2) {
      stage(hA, hB, n);
      stage(hB, hA, n + 1);
    }
    if (active) epilogue(hB, s0 + nst - 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // tell the compiler's wait-count pass: nothing in flight (vmcnt(0) was waited)
    tile0 += ntile;
  }
}

// ---- pack builder: fp32 Wqkv (3C, C) -> the stage stream.  Stage n, row (ks * 32 + r): Wqkv[32 n + r][32 ks .. 32 ks + 31]
// as [32 hi | 32 lo] bf16; 16-B slot t of a row stored at slot t ^ ((row >> 1) & 7) (the stage format of csrc/mlp_fused.hip)
__global__ void __launch_bounds__(256)
qkv_pack_kernel(unsigned char* __restrict__ pack, const float* __restrict__ w, int C) {
  const int nst = 3 * C / 32;
  const int64_t cells = (int64_t)nst * C * 8;
  for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(cell & 7);
    const int r = (int)((cell >> 3) % C);
    const int n = (int)(cell / ((int64_t)C * 8));
    const int t = slot ^ ((r >> 1) & 7);
    const int q = t & 3;
    const int ks = r >> 5, rr = r & 31;
    const float* src = w + (int64_t)(32 * n + rr) * C + ks * 32 + q * 8;
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t a = x3_bf16_rne(src[2 * e]), b = x3_bf16_rne(src[2 * e + 1]);
      if (t >= 4) {
        a = x3_bf16_rne(src[2 * e] - __uint_as_float(a << 16));
        b = x3_bf16_rne(src[2 * e + 1] - __uint_as_float(b << 16));
      }
      o[e] = a | (b << 16);
    }
    *reinterpret_cast<uint4*>(pack + cell * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

static int g_qkv_tail_split = 1;    // probe knob 'tail_split'

extern "C" {

void hfl_internal_set_qkv_tail_split(int v) { g_qkv_tail_split = v ? 1 : 0; }

int64_t hfl_qkv_fused_pack_bytes(int channels) {
  if (channels != 128 && channels != 256) return 0;
  return (int64_t)(3 * channels / 32) * channels * 128;
}

int hfl_qkv_fused_pack(void* pack, const float* w_qkv, int channels, hfl_stream_t stream) {
  if (pack == nullptr || w_qkv == nullptr || hfl_qkv_fused_pack_bytes(channels) == 0) return HFL_EINVAL;
  const int64_t cells = hfl_qkv_fused_pack_bytes(channels) / 16;
  qkv_pack_kernel<<<(unsigned)hfl_cdiv(cells, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<unsigned char*>(pack), w_qkv, channels);
  HFL_RETURN_LAST_ERROR();
}

static int ln_qkv_fused_launch(void* qkv_out, const float* x, const HflRowSeg* xseg, const float* gamma, const float* beta,
                               float eps, const void* pack, const float* bias, float q_scale, int64_t n_rows, int channels,
                               hfl_stream_t stream);

int hfl_ln_qkv_fused(void* qkv_out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                     const float* bias, float q_scale, int64_t n_rows, int channels, hfl_stream_t stream) {
  return ln_qkv_fused_launch(qkv_out, x, nullptr, gamma, beta, eps, pack, bias, q_scale, n_rows, channels, stream);
}

int hfl_ln_qkv_fused_seg(void* qkv_out, const hfl_row_segments* x, const float* gamma, const float* beta, float eps,
                         const void* pack, const float* bias, float q_scale, int64_t n_rows, int channels, hfl_stream_t stream) {
  HflRowSeg seg;
  if (n_rows < 0 || !hfl_seg_from(x, n_rows, &seg)) return HFL_EINVAL;
  return ln_qkv_fused_launch(qkv_out, seg.ptr[0], &seg, gamma, beta, eps, pack, bias, q_scale, n_rows, channels, stream);
}

static int ln_qkv_fused_launch(void* qkv_out, const float* x, const HflRowSeg* xseg, const float* gamma, const float* beta,
                               float eps, const void* pack, const float* bias, float q_scale, int64_t n_rows, int channels,
                               hfl_stream_t stream) {
  if (qkv_out == nullptr || x == nullptr || gamma == nullptr || beta == nullptr || pack == nullptr || bias == nullptr ||
      n_rows < 0)
    return HFL_EINVAL;
  if (channels != 128 && channels != 256) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  if (hfl_cdiv(n_rows, 16) > 0x7fffffffLL) return HFL_ECAPACITY;
  QkvFusedParams p;
  p.out = static_cast<unsigned char*>(qkv_out); p.x = x; p.gamma = gamma; p.beta = beta;
  p.xseg = xseg != nullptr ? *xseg : hfl_seg_single(x);
  p.pack = static_cast<const unsigned char*>(pack); p.bias = bias; p.M = n_rows; p.eps = eps; p.q_scale = q_scale;
  p.n_tiles = (int)hfl_cdiv(n_rows, 16);
  int cus = hfl_stream_cus(static_cast<hipStream_t>(stream));
  int grid = p.n_tiles < cus ? p.n_tiles : cus;
  // (8 waves, two per SIMD; C = 256: one 16-row tile per wave -- two spill; 4 waves x 512 registers lost, profiles/r04_waves_probe.log)
  const int waves = 8, nt = channels == 256 ? 1 : 2;
  p.stagger = p.n_tiles > (int64_t)grid * waves * nt ? 1 : 0;          // only when a workgroup walks several passes
  p.stagger_groups = 8;
  p.full_passes = 0; p.tail_tile0 = 0; p.tail_sets = 0; p.tail_parts = 0;
  {
    // whole passes for everybody, then the left-over tiles with the output features split over the workgroups of a set
    const int tpp = waves * nt, nst_all = 3 * channels / 32;
    const int64_t full = p.n_tiles / ((int64_t)cus * tpp);
    const int64_t rem = p.n_tiles - full * cus * tpp;
    // (full == 0: fewer rows than one round -- the relay rows of a block, 44 .. 1.4 k of them: their few row sets would each
    // stream all of W on one CU; with the output features split the launch takes two stages per workgroup)
    if (g_qkv_tail_split && rem > 0) {
      const int sets = (int)hfl_cdiv(rem, tpp);
      int parts = 1;
      for (int c = 2; c <= cus / sets && c <= nst_all / 2; ++c)
        if (nst_all % c == 0 && (nst_all / c) % 2 == 0) parts = c;      // parts | stages, an even number of stages each
      if (parts >= 2) {
        p.full_passes = (int)full; p.tail_tile0 = (int)(full * cus * tpp); p.tail_sets = sets; p.tail_parts = parts;
        if (full == 0) grid = sets * parts;
      }
    }
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)4 * channels * 128 + (size_t)channels * 20 + (size_t)waves * qkv_staging_blocks(channels, nt, waves) * 2048;
#define HFL_QKV_LAUNCH(CC, NT, WW, PF)                                                                          \
  {                                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_qkv_fused_kernel<CC, NT, WW, PF>),      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    ln_qkv_fused_kernel<CC, NT, WW, PF><<<grid, WW * 64, lds, s>>>(p);                                          \
  }
  if (channels == 256) HFL_QKV_LAUNCH(256, 1, 8, 3) else HFL_QKV_LAUNCH(128, 2, 8, 3)
#undef HFL_QKV_LAUNCH
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
