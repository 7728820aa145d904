// The pre-norm MLP branch of a transformer block as ONE kernel on the bf16 matrix cores of gfx950:
//
//     out (M, C) = x + fc2( gelu( fc1( LayerNorm(x) ) ) )        fc1: C -> 4C, fc2: 4C -> C
//
// Replaces norm2 -> mlp.fc1 -> GELU -> mlp.fc2 -> residual add of every transformer block of the reference
// (models/octformer_backbone.py:275-278, models/hotformerloc_backbone.py:213-216, models/layers/octformer_layers.py:38-59),
// which round 2 ran as three launches (hfl_layer_norm_split2, hfl_linear_x3 with the GELU epilogue, hfl_linear_x3 with the
// residual epilogue): there the M x 4C hidden activation crossed HBM twice (8 of the 13 M*C*4-byte units those three
// launches move); here it never leaves the register file.
//
// Arithmetic: identical to csrc/gemm_x3.hip -- every fp32 operand split into bf16 (hi, lo), products x_lo w_hi + x_hi w_lo +
// x_hi w_hi with fp32 accumulation (v_mfma_f32_16x16x32_bf16), exact-erf GELU in fp32, LayerNorm with two-pass statistics.
//
// Dataflow (C = 256; C = 128 in brackets).  A 512-lane workgroup owns up to 128 [256] rows per pass; wave w owns NT = 1 [2]
// tiles of 16 rows and keeps, for the whole pass,
//   * LayerNorm(x) of its rows as MFMA B-operand fragments, bf16 (hi, lo), K = C: 64 VGPRs,
//   * the fc2 accumulators out^T[C features x its rows]: 64 VGPRs.
// The weights stream through LDS in chunks of 32 hidden features: stage A_j = W1[32 j .. 32 j + 31][0 .. C) and stage
// B_j = W2[0 .. C)[32 j .. 32 j + 31], C rows x 128 B each, laid out ONCE per parameter on the host as the byte image the LDS
// wants (hfl_mlp_fused_pack_*: stages in consumption order, 16-B slots of every row XOR-swizzled), so a stage is a linear
// LDS-DMA copy (global_load_lds_dwordx4) with no address arithmetic.  Per chunk a wave runs
//   GEMM1  h^T[32 hidden x 16 rows] = W1_chunk . LN(x)^T     (A from LDS, B from registers)
//   GELU   in registers; the accumulator layout of GEMM1 (4 consecutive hidden features per lane, row on the lane) IS a
//          B-operand layout of GEMM2 once W2's k-order inside the 32-chunk is permuted to match (done in the pack): the
//          hidden activation is converted to (hi, lo) bf16 fragments in place -- no LDS round trip, no barrier
//   GEMM2  out^T[C x 16 rows] += W2_chunk . g^T
// software-pipelined by one chunk (stage order A0, A1, B0, A2, B1, ...): the GELU of chunk j-1 is VALU work between the
// MFMAs of GEMM1 of chunk j.  Four LDS slots, two stages in flight: one s_barrier and one counted s_waitcnt vmcnt per
// stage, nothing else on the vector-memory counter inside the loop; the refill of a slot is issued piecewise between the
// MFMA steps of a stage.
// Work split: persistent-style, every workgroup takes an equal contiguous share of the 16-row tiles and walks it in passes
// (the weight stream restarts per pass and is served by the XCD's L2: 2 MB [0.5 MB] per parameter pair).
#include "hfl_common.h"
#include "x3_math.h"
#include "stage_stream.h"

#include <mutex>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct MlpFusedParams {
  float* out;                 // (M, C) f32
  const float* x;             // (M, C) f32: LayerNorm input and residual
  const float* gamma;         // (C)
  const float* beta;          // (C)
  const unsigned char* pack;  // hfl_mlp_fused_pack image of (W1, W2)
  const float* b1;            // (4C)
  const float* b2;            // (C)
  int64_t M;
  float eps;
  int n_tiles;                // ceil(M / 16)
  int stagger;                // workgroup g starts (g % stagger_groups) * stagger naps of ~4000 cycles late (0: all together)
  int stagger_groups;
  // Tail split (tail_sets > 0): every workgroup walks `full_passes` whole passes; the tiles left over (fewer than one pass
  // per workgroup) are cut into `tail_sets` sets of up to one pass, and each set is computed by `tail_parts` workgroups that
  // share the hidden dimension: part k runs chunks [k, k + 1) * NCH / tail_parts and writes its fc2 partial sums to
  // part[k] (rows of the tail region x C, f32); mlp_tail_reduce_kernel adds them in fixed order (+ bias + residual).
  int full_passes;
  int tail_tile0;             // first tile of the tail region
  int tail_sets;
  int tail_parts;
  float* part;                // (tail_parts, tail rows, C)
  // Work units: whole passes first (full_units of them when the tail is split, else ceil(n_tiles / tiles per pass)), then
  // the tail units (set, part).  ticket == nullptr: workgroup g takes the units g, g + G, ...; else the units are handed out
  // by an atomic counter (ticket[0]; ticket[1] counts finished workgroups, the last one resets both): a workgroup that
  // starts late or shares its CU with another stream's kernels simply takes fewer units -- beside the coarse levels' launches
  // the static split let the slowest workgroup set the launch's time (410 us against 258 us alone).
  int full_units;
  unsigned int* ticket;
};

static int g_mlp_stagger = 1;      // probe knob 'mlp_stagger': naps per group step | groups << 8
static int g_mlp_stagger_groups = 8;
#ifdef HFL_PROBES
static int g_mlp_dbg = 0;          // probe knob 'mlp_dbg': timing ablations (see the kernel), 0 = off
#endif

// DBG (probe knob 'mlp_dbg', timing ablations only -- results are wrong): 1 GELU -> identity, 2 no bias / GELU / split at all,
// 4 no refill of the weight ring, 8 no barrier at the stage boundaries
template <int C, int NT, int PF, int NW, int LAG, int DBG = 0, int HID = 4 * C>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 1)
ln_mlp_fused_kernel(const MlpFusedParams p) {
  static_assert(LAG == 0 || (PF == 2 && NW == 8), "the late half keeps a stage one step longer: two stages ahead, not three");
  static_assert(PF == 2 || PF == 3, "stages in flight ahead of the one being consumed");
  static_assert(NW == 8 || NW == 4, "waves per workgroup: two per SIMD with 256 registers each, or one with 512");
  constexpr int NTHR = NW * 64;
  constexpr int KS = C / 32;               // k-steps of GEMM1
  constexpr int FT = C / 16;               // 16-feature tiles of the output
  constexpr int NCH = HID / 32;            // chunks of 32 hidden features (hidden = 4 C in the transformer blocks, C in the Mixer)
  constexpr int STAGE_B = C * 128;         // bytes of one stage: C rows x 128 B
  constexpr int NSLOT = 4;
  constexpr int DPW = STAGE_B / 1024 / NW; // LDS-DMA instructions per wave and stage
  constexpr int NST = 2 * NCH;             // stages per pass
  constexpr int TPP = NW * NT;             // 16-row tiles per pass
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  float* b1s = reinterpret_cast<float*>(smem + NSLOT * STAGE_B);       // (4C) fc1 bias | (C) gamma | (C) beta | (C) fc2 bias
  float* gms = b1s + HID;
  float* bts = gms + C;
  float* b2s = bts + C;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- work units of this launch (see MlpFusedParams)
  const int n_full = p.tail_sets > 0 ? p.full_units : (p.n_tiles + TPP - 1) / TPP;
  const int n_units = n_full + (p.tail_sets > 0 ? p.tail_sets * p.tail_parts : 0);
  __shared__ int s_unit;

  // small vectors live in LDS for the whole kernel: read per use with ds_read (a global load per use would be a dependent
  // L2 round trip each -- the register file has no room to hold them)
  for (int i = tid; i < HID / 4; i += NTHR) reinterpret_cast<float4*>(b1s)[i] = reinterpret_cast<const float4*>(p.b1)[i];
  for (int i = tid; i < C / 4; i += NTHR) {
    reinterpret_cast<float4*>(gms)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
    reinterpret_cast<float4*>(bts)[i] = reinterpret_cast<const float4*>(p.beta)[i];
    reinterpret_cast<float4*>(b2s)[i] = reinterpret_cast<const float4*>(p.b2)[i];
  }
  __syncthreads();

  // ---- stage stream: stage n of a pass lives at pack + n * STAGE_B; wave w copies bytes [w, w+1) * DPW KiB of it
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // LAG: waves NW/2 .. NW-1 (the second wave of every SIMD) run one stage behind the first half.  Stages alternate GEMM1 +
  // GELU (VALU between the MFMAs) and GEMM2 (MFMAs only): in lock-step both waves of a SIMD are in the same kind of stage
  // and the GELU's vector instructions queue behind the MFMA issue of BOTH (PMC: VALU under a running MFMA 4 % of the MFMA
  // time); one stage apart, one wave's GELU fills the issue gaps of the other's GEMM2.
  const bool lagw = LAG != 0 && wave >= NW / 2;
  uint32_t seq = 0;                         // stages acquired so far (all passes): slot = seq % NSLOT
  // A pass over the hidden chunks [c0, c0 + nch) (all of them except in the tail) consumes the stages A_c0, A_c0+1, B_c0, ...,
  // A_last, B_last-1, B_last: in the pack's stage order (A0, A1, B0, A2, B1, ...) that is index 2 c0 - 1 (0 for c0 = 0), then
  // 2 c0 + 1 ... 2 (c0 + nch) - 2 contiguously, then 2 (c0 + nch) (2 NCH - 1 for the last chunk of all).
  int c0 = 0, nst = NST;                    // current pass: first chunk, stages
  auto sidx = [&](int n) -> int {
    if (n == 0) return c0 == 0 ? 0 : 2 * c0 - 1;
    if (n == nst - 1) return 2 * c0 + nst == NST ? NST - 1 : 2 * c0 + nst;
    return 2 * c0 + n;
  };
  auto issue = [&](int n, uint32_t slot) {
    const unsigned char* s = p.pack + (int64_t)sidx(n) * STAGE_B + wave * (DPW * 1024);      // wave-uniform
    unsigned char* d = smem + slot * STAGE_B + wave * (DPW * 1024);
#pragma unroll
    for (int i = 0; i < DPW; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + i * 1024 + lane_off),
                                       (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
  };
  // Boundary step of stage n: it has landed for every wave (each waited for its own share, then the barrier) and every
  // wave has left stage n - 1, hence also stage n - 2, whose slot takes stage n + 2.  That refill is NOT issued here: eight
  // waves pushing 4 LDS-DMA instructions each through the CU's one address path at the same moment block each other at
  // issue for hundreds of cycles with the matrix pipe idle; the pieces are issued one at a time between the MFMA steps of
  // stage n (dma_piece).  Two stages in flight, four slots.
  int dma_n = 0;                            // stage whose pieces the current stage body issues (NST: none)
  uint32_t dma_slot = 0;
  const unsigned char* dma_src = p.pack;    // this lane's source of piece 0 of that stage, and the wave's LDS destination
  unsigned char* dma_dst = smem;
  auto acquire = [&](int n) -> const unsigned char* {
    // stages issued after stage n so far: n + 1 .. n + PF - 1 (those that exist); this wave's pieces of them may stay in flight
    if (n + PF - 1 < nst) HFL_WAIT_VM((PF - 1) * DPW);
    else if (n + 1 < nst) HFL_WAIT_VM((PF - 2) * DPW > 0 ? (PF - 2) * DPW : 0);
    else HFL_WAIT_VM(0);
    if constexpr (!(DBG & 8)) __builtin_amdgcn_s_barrier();
    dma_n = n + PF;
    dma_slot = (seq + PF) % NSLOT;
    // the refill's addresses once per stage: its pieces differ in the instruction's immediate offset only (which advances
    // the global and the LDS address alike) -- per piece this was a stage-index select, a 64-bit multiply-add and an M0 write
    dma_src = p.pack + (int64_t)sidx(dma_n < nst ? dma_n : 0) * STAGE_B + wave * (DPW * 1024) + lane_off;
    dma_dst = smem + dma_slot * STAGE_B + wave * (DPW * 1024);
    const unsigned char* st = smem + ((seq - (lagw ? 1u : 0u)) % NSLOT) * STAGE_B;   // (late half: the stage before)
    ++seq;
    return st;
  };
  auto dma_piece = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (DBG & 4) return;
    if (dma_n < nst)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)dma_src,
                                       (__attribute__((address_space(3))) void*)dma_dst, 16, i * 1024, 0);
  };
  constexpr int PPH = DPW / 2;              // pieces per half stage

  // fragment byte offsets inside a stage (rows of 128 B, 16-B slot t of row r stored at slot t ^ ((r >> 1) & 7)):
  // A fragment of a 16-row block: row = block * 16 + fr, hi chunk fq, lo chunk 4 + fq
  const int off_hi = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), off_lo = off_hi ^ 64;

  // Every workgroup has the same work, so without help they all sit in their memory-only phases (row loads + LayerNorm at
  // the top of a pass, residual loads + stores at its end: ~67 MB each way per pass over the chip) at the same time, with
  // every matrix pipe idle, and then all compute with the memory system idle.  Half of them start a fraction of that phase
  // later: from then on the groups alternate.  (Only when a workgroup has more than one pass: p.stagger = 0 otherwise.)
  {
    const int naps = (int)(blockIdx.x % (unsigned)p.stagger_groups) * p.stagger;
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(63);          // ~4000 cycles each
  }
  for (int unit = blockIdx.x; unit < n_units;) {
    int tile0, tile_end, tpart = 0;
    const bool tail = unit >= n_full;
    if (!tail) {                            // a whole pass: TPP tiles, every hidden chunk
      tile0 = unit * TPP;
      tile_end = tile0 + TPP < p.n_tiles ? tile0 + TPP : p.n_tiles;
      c0 = 0;
      nst = NST;
    } else {                                // part `tpart` of a tail set: its share of the hidden chunks
      const int v = unit - n_full;
      const int tset = v % p.tail_sets;
      tpart = v / p.tail_sets;
      tile0 = p.tail_tile0 + tset * TPP;
      tile_end = tile0 + TPP < p.n_tiles ? tile0 + TPP : p.n_tiles;
      const int tnch = NCH / p.tail_parts;
      c0 = tpart * tnch;
      nst = 2 * tnch;
    }
    int next_unit = unit + (int)gridDim.x;      // (wave-uniform: the ticket travels in `ticket_val` of thread 0 only)
    int ticket_val = 0;
    const int nch = nst / 2;                // hidden chunks of this pass (even)
    const int ntile = tile_end - tile0 < TPP ? tile_end - tile0 : TPP;    // tiles of this pass
    // ---- LayerNorm of this wave's rows -> B-operand fragments (lane: row fr of the tile, channels 32 ks + 8 fq + j)
    bf16x8 xh[NT][KS], xl[NT][KS];
    int64_t row[NT];
    bool have[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int k = t * NW + wave;                                  // tile k of the pass -> wave k % NW, slot k / NW
      have[t] = k < ntile;
      int64_t r = (int64_t)(tile0 + k) * 16 + fr;
      row[t] = r;
      if (r >= p.M) r = p.M - 1;
      if (!have[t]) r = 0;
      const float* xr = p.x + r * C + fq * 8;
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
    const bool active = have[0];                 // wave-uniform: tiles are dealt to the waves slot by slot
    // The first two stages start moving only now: while an LDS-DMA is in flight hipcc waits vmcnt(0) before every use of
    // an ordinary load's result, which would turn the row loads above into 16 dependent round trips.
    // (Every wave has left the previous pass's last stage: the barrier orders it.)
    __builtin_amdgcn_s_barrier();
    // the next unit's ticket: requested here (older than this pass's weight stream on the memory counter, so the stream's
    // counted waits cover it), read at the bottom of the pass
    if (p.ticket != nullptr && tid == 0) ticket_val = (int)atomicAdd(p.ticket, 1u);
    issue(0, seq % NSLOT);
    issue(1, (seq + 1) % NSLOT);
    if (PF == 3 && nst > 2) issue(2, (seq + 2) % NSLOT);

    f32x4 oacc[FT][NT];
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
      for (int t = 0; t < NT; ++t) oacc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 hA[2][NT], hB[2][NT];                  // fc1 accumulators of two consecutive chunks
    u32x4 gh[NT], gl[NT];                        // gelu(fc1) of one chunk as GEMM2 B fragments, (hi, lo) bf16 pairs per dword

    // one pair of hidden values through bias + GELU + split: dword d of row tile t's fragment = features 2 d, 2 d + 1 of the
    // lane's 8 (hidden block i = d >> 1, accumulator registers 2 (d & 1), 2 (d & 1) + 1); 4 NT pairs per chunk
    auto gelu_pair = [&](int pidx, f32x4 (&hp)[2][NT], int chunk) {
      const int t = pidx >> 2, d = pidx & 3;
      const int i = d >> 1, r0 = 2 * (d & 1);
      if constexpr (DBG & 2) {
        gh[t][d] = __float_as_uint(hp[i][t][r0]);
        gl[t][d] = __float_as_uint(hp[i][t][r0 + 1]);
        return;
      }
      (void)chunk;                  // (the bias is in the accumulators already: gemm1_half)
      // SCALAR f32 math on purpose (and -fno-slp-vectorize for this file): beside MFMAs a v_pk_*_f32 costs ~12 extra
      // cycles of the matrix pipe each (MI355X_MICROARCH.md, cycle constants), a plain VALU op hides in the MFMA's shadow
      const float g0 = (DBG & 1) ? hp[i][t][r0] : x3_gelu(hp[i][t][r0]);
      const float g1 = (DBG & 1) ? hp[i][t][r0 + 1] : x3_gelu(hp[i][t][r0 + 1]);
      uint32_t hi, lo;
      x3_split_pair_scalar(g0, g1, hi, lo);
      gh[t][d] = hi;
      gl[t][d] = lo;
    };
    // half `hf` of GEMM1 of one chunk from stage `st` into h; between its k-steps the GELU of the previous chunk (hp -> gh, gl)
    auto gemm1_half = [&](const unsigned char* st, auto hfc, f32x4 (&h)[2][NT], f32x4 (&hp)[2][NT], int prev_chunk, bool with_gelu) {
      constexpr int hf = decltype(hfc)::value;
      if (hf == 0) {
        // the accumulators start from the fc1 bias (lane: hidden 16 i + 4 fq .. + 3 of the chunk): read here, in front of the
        // first fragment read and waited for with it -- as a ds_read inside the GELU (rounds 3-4) its lgkmcnt(0) also waited
        // for the fragment prefetch issued just before, i.e. it serialised every k-step's LDS round trip with its MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4 bi = (DBG & 2) ? (f32x4){0.f, 0.f, 0.f, 0.f}
                                     : *reinterpret_cast<const f32x4*>(b1s + (c0 + prev_chunk + (with_gelu ? 1 : 0)) * 32 + i * 16 + fq * 4);
#pragma unroll
          for (int t = 0; t < NT; ++t) h[i][t] = bi;
        }
      }
      // fragments of k-step kk + 1 are requested before the MFMAs of k-step kk (two register sets): the LDS round trip
      // (~200 cycles) would otherwise be exposed in front of every 6 NT MFMAs
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
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i], xl[t][ks], h[i][t], 0, 0, 0);
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i + 1], xh[t][ks], h[i][t], 0, 0, 0);
            h[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk & 1][2 * i], xh[t][ks], h[i][t], 0, 0, 0);
          }
        // the previous chunk's bias + GELU + split, a pair of hidden values at a time between the MFMAs (4 NT pairs over the
        // KS k-steps of the stage: few live temporaries, the register file is what limits the fragment prefetch)
        if (with_gelu) {
          if constexpr (4 * NT >= KS) {
#pragma unroll
            for (int u = 0; u < 4 * NT / KS; ++u) gelu_pair(ks * (4 * NT / KS) + u, hp, prev_chunk);
          } else {
            if (ks % (KS / (4 * NT)) == 0) gelu_pair(ks / (KS / (4 * NT)), hp, prev_chunk);
          }
        }
        if constexpr ((kk + 1) % ((KS / 2) / PPH) == 0)
          dma_piece(std::integral_constant<int, hf * PPH + (kk + 1) / ((KS / 2) / PPH) - 1>{});
        if constexpr (kk + 1 < KS / 2)
          HFL_LDS_WAIT4_AFTER(wf[(kk + 1) & 1][0], wf[(kk + 1) & 1][1], wf[(kk + 1) & 1][2], wf[(kk + 1) & 1][3], h[1][NT - 1]);
      });
    };
    auto gelu_only = [&](f32x4 (&hp)[2][NT], int prev_chunk) {
#pragma unroll
      for (int pidx = 0; pidx < 4 * NT; ++pidx) gelu_pair(pidx, hp, prev_chunk);
    };
    auto gemm2_half = [&](const unsigned char* st, auto hfc) {
      constexpr int hf = decltype(hfc)::value;
      // two output tiles per step, the next step's 4 fragments requested before this step's MFMAs
      const uint32_t ahi = (uint32_t)(uintptr_t)(st + off_hi) + hf * ((FT / 2) * 2048);
      const uint32_t alo = (uint32_t)(uintptr_t)(st + off_lo) + hf * ((FT / 2) * 2048);
      bf16x8 wf[2][4];
      HFL_LDS_READ4_FIRST(wf[0][0], wf[0][1], wf[0][2], wf[0][3], ahi, alo, 0, 2048);
      HFL_LDS_WAIT4(wf[0][0], wf[0][1], wf[0][2], wf[0][3]);
      hfl_static_for(std::make_integer_sequence<int, FT / 4>{}, [&](auto sc) {
        constexpr int ss = decltype(sc)::value;
        if constexpr (ss + 1 < FT / 4)
          HFL_LDS_READ4(wf[(ss + 1) & 1][0], wf[(ss + 1) & 1][1], wf[(ss + 1) & 1][2], wf[(ss + 1) & 1][3], ahi, alo,
                        (ss + 1) * 4096, (ss + 1) * 4096 + 2048, wf[ss & 1][0]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          constexpr int i0 = hf * (FT / 2) + 2 * ss;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            oacc[i0 + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ss & 1][2 * u], __builtin_bit_cast(bf16x8, gl[t]), oacc[i0 + u][t], 0, 0, 0);
            oacc[i0 + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ss & 1][2 * u + 1], __builtin_bit_cast(bf16x8, gh[t]), oacc[i0 + u][t], 0, 0, 0);
            oacc[i0 + u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ss & 1][2 * u], __builtin_bit_cast(bf16x8, gh[t]), oacc[i0 + u][t], 0, 0, 0);
          }
        }
        if constexpr ((ss + 1) % ((FT / 4) / PPH) == 0)
          dma_piece(std::integral_constant<int, hf * PPH + (ss + 1) / ((FT / 4) / PPH) - 1>{});
        if constexpr (ss + 1 < FT / 4)
          HFL_LDS_WAIT4_AFTER(wf[(ss + 1) & 1][0], wf[(ss + 1) & 1][1], wf[(ss + 1) & 1][2], wf[(ss + 1) & 1][3],
                              oacc[hf * (FT / 2) + 2 * ss + 1][NT - 1]);
      });
    };

    // ---- the chunk pipeline: A0 | A1+gelu(0) B0 | A2+gelu(1) B1 | ... | A_{NCH-1}+gelu(NCH-2) B_{NCH-2} | gelu(NCH-1) B_{NCH-1}
    // (waves without a row tile in this pass still carry their share of the weight stream)
    int n = 0;
    const unsigned char* st = acquire(n++);
    auto idle_pieces = [&]() { hfl_static_for(std::make_integer_sequence<int, DPW>{}, [&](auto ic) { dma_piece(ic); }); };
    // step boundaries: the early half crosses one AFTER each stage body (the next stage), the late half BEFORE it (step 0 of
    // the late half is empty; its last stage landed at the previous boundary) -- the same number of barriers for both
    if (lagw) idle_pieces();
    auto enter = [&]() {
      if (lagw) {
        if (n < nst) {
          st = acquire(n++);
        } else {
          st = smem + ((seq - 1u) % NSLOT) * STAGE_B;
          dma_n = nst;
        }
      }
    };
    auto stage1 = [&](f32x4 (&h)[2][NT], f32x4 (&hp)[2][NT], int prev_chunk, bool with_gelu) {
      enter();
      if (active) {
        gemm1_half(st, std::integral_constant<int, 0>{}, h, hp, prev_chunk, with_gelu);
        gemm1_half(st, std::integral_constant<int, 1>{}, h, hp, prev_chunk, with_gelu);
      } else {
        idle_pieces();
      }
      if (!lagw) st = acquire(n++);
    };
    auto stage2 = [&](bool more) {
      enter();
      if (active) {
        gemm2_half(st, std::integral_constant<int, 0>{});
        gemm2_half(st, std::integral_constant<int, 1>{});
      } else {
        idle_pieces();
      }
      if (!lagw && more) st = acquire(n++);
    };
    stage1(hA, hB, 0, false);                                             // A0
#pragma unroll 1
    for (int j = 1; j < nch; j += 2) {           // nch is even: j odd here, j + 1 even
      stage1(hB, hA, j - 1, true);                                        // A_j + gelu(j - 1)
      stage2(true);                                                       // B_{j-1}
      if (j + 1 < nch) {
        stage1(hA, hB, j, true);                                          // A_{j+1} + gelu(j)
        stage2(true);                                                     // B_j
      }
    }
    if (active) gelu_only(hB, nch - 1);          // nch - 1 is odd: its accumulators are hB
    stage2(false);                                                        // B_{nch-1}

    // ---- epilogue: out = acc + b2 + x (lane: row fr of the tile, features 16 i + 4 fq .. + 3)
    // (tell the compiler's wait-count pass what the inline-asm wait of the last stage did: nothing is in flight any more,
    // so the residual loads below get ordinary counted waits)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
    if (tail) {                                  // partial sums of this part's chunks, no bias, no residual
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!have[t] || row[t] >= p.M) continue;
        const int64_t tail_rows = p.M - (int64_t)p.tail_tile0 * 16;
        float* orow = p.part + ((int64_t)tpart * tail_rows + (row[t] - (int64_t)p.tail_tile0 * 16)) * C + fq * 4;
#pragma unroll
        for (int i = 0; i < FT; ++i) *reinterpret_cast<f32x4*>(orow + i * 16) = oacc[i][t];
      }
    } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (!have[t] || row[t] >= p.M) continue;
      const float* xr = p.x + row[t] * C + fq * 4;
      float* orow = p.out + row[t] * C + fq * 4;
      f32x4 res[FT];               // every residual load in flight before the first store (the compiler cannot prove that
#pragma unroll                     // out and x do not alias and would not hoist them itself)
      for (int i = 0; i < FT; ++i) res[i] = *reinterpret_cast<const f32x4*>(xr + i * 16);
#pragma unroll
      for (int i = 0; i < FT; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(b2s + i * 16 + fq * 4);
        *reinterpret_cast<f32x4*>(orow + i * 16) = oacc[i][t] + b + res[i];
      }
    }
    }
    if (p.ticket != nullptr) {              // thread 0 holds the next unit: through LDS to everybody
      __syncthreads();
      if (tid == 0) s_unit = (int)gridDim.x + ticket_val;
      __syncthreads();
      next_unit = __builtin_amdgcn_readfirstlane(s_unit);
    }
    unit = next_unit;
  }
  if (p.ticket != nullptr && tid == 0) {    // the last workgroup out resets the counters for the next launch on this slot
    if (atomicAdd(p.ticket + 1, 1u) == gridDim.x - 1) {
      p.ticket[0] = 0u;
      p.ticket[1] = 0u;
    }
  }
}

// out[r] = x[r] + b2 + sum_k part[k][r] over the tail rows, k in fixed order (bitwise reproducible); one float4 per lane
__global__ void __launch_bounds__(256)
mlp_tail_reduce_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ b2,
                       const float* __restrict__ part, int64_t row0, int64_t tail_rows, int C, int parts) {
  const int64_t n4 = tail_rows * C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)((i * 4) % C);
    float4 a = reinterpret_cast<const float4*>(x + row0 * C)[i];
    const float4 b = *reinterpret_cast<const float4*>(b2 + col);
    float4 sum = reinterpret_cast<const float4*>(part)[i];
    for (int k = 1; k < parts; ++k) {
      const float4 v = reinterpret_cast<const float4*>(part + (int64_t)k * tail_rows * C)[i];
      sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    a.x = sum.x + b.x + a.x; a.y = sum.y + b.y + a.y; a.z = sum.z + b.z + a.z; a.w = sum.w + b.w + a.w;
    reinterpret_cast<float4*>(out + row0 * C)[i] = a;
  }
}

// ---- host-side pack builder kernel: fp32 weights -> the stage stream --------------------------------------------------
// Stage order of a pass (NCH = C / 8 chunks): A0, A1, B0, A2, B1, ..., A_{NCH-1}, B_{NCH-2}, B_{NCH-1}.
// Stage A_j, row (ks * 32 + r):  W1[32 j + r][32 ks .. 32 ks + 31]              as [32 hi | 32 lo] bf16
// Stage B_j, row n:              W2[n][32 j + perm(0 .. 31)]                    as [32 hi | 32 lo] bf16,
//   perm: position 8 q + e  <->  hidden offset 16 (e >> 2) + 4 q + (e & 3)   (the accumulator layout of GEMM1 read as a
//   B fragment of GEMM2).  16-B slot t of a row is stored at slot t ^ ((row >> 1) & 7).
__global__ void __launch_bounds__(256)
mlp_pack_kernel(unsigned char* __restrict__ pack, const float* __restrict__ w1, const float* __restrict__ w2, int C, int hidden) {
  const int nch = hidden / 32;
  const int64_t cells = (int64_t)2 * nch * C * 8;                 // 16-B slots of the pack
  for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < cells; cell += (int64_t)gridDim.x * blockDim.x) {
    const int slot = (int)(cell & 7);
    const int r = (int)((cell >> 3) % C);
    const int stage = (int)(cell / ((int64_t)C * 8));
    // stage index -> (kind, chunk)
    int kind, j;
    if (stage == 0) { kind = 0; j = 0; }
    else if (stage == 2 * nch - 1) { kind = 1; j = nch - 1; }
    else if (stage & 1) { kind = 0; j = (stage + 1) / 2; }
    else { kind = 1; j = stage / 2 - 1; }
    const int t = slot ^ ((r >> 1) & 7);                           // logical 16-B chunk: 0..3 hi, 4..7 lo; 8 elements each
    const int q = t & 3;
    float v[8];
    if (kind == 0) {
      const int ks = r >> 5, rr = r & 31;
      const float* src = w1 + (int64_t)(32 * j + rr) * C + ks * 32 + q * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[e];
    } else {
      const float* src = w2 + (int64_t)r * hidden + 32 * j;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = src[16 * (e >> 2) + 4 * q + (e & 3)];
    }
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t a = x3_bf16_rne(v[2 * e]), b = x3_bf16_rne(v[2 * e + 1]);
      if (t >= 4) {
        a = x3_bf16_rne(v[2 * e] - __uint_as_float(a << 16));
        b = x3_bf16_rne(v[2 * e + 1] - __uint_as_float(b << 16));
      }
      o[e] = a | (b << 16);
    }
    *reinterpret_cast<uint4*>(pack + cell * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

}  // namespace

static int grid_guess(int n_tiles, int cus) { return n_tiles < cus ? n_tiles : cus; }

extern "C" {

#ifdef HFL_PROBES
void hfl_internal_set_mlp_dbg(int v) { g_mlp_dbg = v; }      // timing ablations (tools/mlp_ablate.py), probe builds only
#endif

void hfl_internal_set_mlp_stagger(int v) {
  g_mlp_stagger = v & 0xFF;
  g_mlp_stagger_groups = (v >> 8) > 0 ? (v >> 8) : 2;
}

static bool mlp_shape_ok(int channels, int hidden) {
  return (channels == 128 || channels == 256) && (hidden == 4 * channels || (channels == 256 && hidden == 256));
}

int64_t hfl_mlp_fused_pack_bytes_h(int channels, int hidden) {
  if (!mlp_shape_ok(channels, hidden)) return 0;
  return (int64_t)2 * (hidden / 32) * channels * 128;
}
int64_t hfl_mlp_fused_pack_bytes(int channels) { return hfl_mlp_fused_pack_bytes_h(channels, 4 * channels); }

int hfl_mlp_fused_pack_h(void* pack, const float* w1, const float* w2, int channels, int hidden, hfl_stream_t stream) {
  if (pack == nullptr || w1 == nullptr || w2 == nullptr || hfl_mlp_fused_pack_bytes_h(channels, hidden) == 0) return HFL_EINVAL;
  const int64_t cells = hfl_mlp_fused_pack_bytes_h(channels, hidden) / 16;
  mlp_pack_kernel<<<(unsigned)hfl_cdiv(cells, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<unsigned char*>(pack), w1, w2, channels, hidden);
  HFL_RETURN_LAST_ERROR();
}
int hfl_mlp_fused_pack(void* pack, const float* w1, const float* w2, int channels, hfl_stream_t stream) {
  return hfl_mlp_fused_pack_h(pack, w1, w2, channels, 4 * channels, stream);
}

// Tail plan of a launch over n_rows: {full passes per workgroup, first tail tile, sets, parts}; parts == 0: no tail split
// (the rows are whole passes, or fewer than one round, or the split is switched off).
struct MlpTailPlan {
  int full, tile0, sets, parts;
};
static int g_mlp_tail_split = 1;   // probe knob 'mlp_tail_split'
static int g_mlp_dynamic = 1;      // probe knob 'mlp_dynamic': work units by atomic ticket (0: static, strided by workgroup)

// Ticket slots (two counters each, one 128-B line per slot), one table per DEVICE, zero-initialised once; a device's launches
// take its slots round-robin and the last workgroup of a launch leaves its slot zeroed.  A slot comes round again 256 ticketed
// launches later on its device; the launches of one forward that take tickets (multi-round row-tile launches) are a few
// dozen, so two live launches never share a slot unless more than 256 of them are in flight on one device at once.
constexpr int kTicketSlots = 256;
constexpr int kTicketDevices = 16;
unsigned int* hfl_internal_ticket_slot() {
  static std::mutex mu;
  static unsigned int* base[kTicketDevices] = {};
  static unsigned int next[kTicketDevices] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kTicketDevices) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (base[dev] == nullptr) {
    unsigned int* b = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&b), (size_t)kTicketSlots * 128) != hipSuccess) return nullptr;
    if (hipMemset(b, 0, (size_t)kTicketSlots * 128) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    base[dev] = b;
  }
  return base[dev] + (size_t)(next[dev]++ % kTicketSlots) * 32;
}
static unsigned int* ticket_slot() { return hfl_internal_ticket_slot(); }
static MlpTailPlan mlp_tail_plan(int64_t n_rows, int channels, int cus, int hidden) {
  MlpTailPlan t{0, 0, 0, 0};
  const int64_t n_tiles = hfl_cdiv(n_rows, 16);
  const int tpp = channels == 256 ? 8 : 16, nch = hidden / 32;
  if (!g_mlp_tail_split) return t;
  // (full == 0: fewer rows than one round -- relay tokens, the coarse pyramid levels.  Their few row sets would leave most of
  // the chip idle while every workgroup streams all of W1 and W2; split over the hidden dimension the same launch fills it.)
  const int64_t full = n_tiles / ((int64_t)cus * tpp);
  const int64_t rem = n_tiles - full * cus * tpp;
  if (rem == 0) return t;
  const int sets = (int)hfl_cdiv(rem, tpp);
  int parts = 1;
  while (parts * 2 <= cus / sets && nch / (parts * 2) >= 2) parts *= 2;      // parts | NCH, at least 2 chunks (an even count) each
  if (parts < 2) return t;                                                   // a tail of more than half a round: plain passes
  t.full = (int)full; t.tile0 = (int)(full * cus * tpp); t.sets = sets; t.parts = parts;
  return t;
}

extern "C" void hfl_internal_set_mlp_tail_split(int v) { g_mlp_tail_split = v ? 1 : 0; }
extern "C" void hfl_internal_set_mlp_dynamic(int v) { g_mlp_dynamic = v ? 1 : 0; }

static int64_t mlp_tail_bytes(const MlpTailPlan& t, int64_t n_rows, int channels) {
  return t.parts == 0 ? 0 : (int64_t)t.parts * (n_rows - (int64_t)t.tile0 * 16) * channels * 4;
}

/* Workspace of hfl_ln_mlp_fused_ws for this shape (0: none needed): enough for a launch on the whole chip or on any
 * CU-masked stream of this library (hfl_stream_create_cu_mask: 8 .. all CUs in steps of 8) -- the plan depends on the CUs the
 * launch's stream can use; parts x left-over rows never exceeds one round, cus x rows per pass. */
extern "C" int64_t hfl_ln_mlp_fused_workspace_h(int64_t n_rows, int channels, int hidden) {
  if (n_rows <= 0 || !mlp_shape_ok(channels, hidden)) return 0;
  int64_t need = 0;
  for (int cus = 8; cus <= hfl_num_cus(); cus += 8) {
    const int64_t b = mlp_tail_bytes(mlp_tail_plan(n_rows, channels, cus, hidden), n_rows, channels);
    if (b > need) need = b;
  }
  return need;
}

extern "C" int64_t hfl_ln_mlp_fused_workspace(int64_t n_rows, int channels) {
  return hfl_ln_mlp_fused_workspace_h(n_rows, channels, 4 * channels);
}

extern "C" int hfl_ln_mlp_fused_ws(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                        const float* b1, const float* b2, int64_t n_rows, int channels, void* workspace,
                        int64_t workspace_bytes, hfl_stream_t stream);
extern "C" int hfl_ln_mlp_fused_h(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                       const float* b1, const float* b2, int64_t n_rows, int channels, int hidden, void* workspace,
                       int64_t workspace_bytes, hfl_stream_t stream);

int hfl_ln_mlp_fused(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                     const float* b1, const float* b2, int64_t n_rows, int channels, hfl_stream_t stream) {
  return hfl_ln_mlp_fused_ws(out, x, gamma, beta, eps, pack, b1, b2, n_rows, channels, nullptr, 0, stream);
}

int hfl_ln_mlp_fused_ws(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                        const float* b1, const float* b2, int64_t n_rows, int channels, void* workspace,
                        int64_t workspace_bytes, hfl_stream_t stream) {
  return hfl_ln_mlp_fused_h(out, x, gamma, beta, eps, pack, b1, b2, n_rows, channels, 4 * channels, workspace, workspace_bytes,
                            stream);
}

int hfl_ln_mlp_fused_h(float* out, const float* x, const float* gamma, const float* beta, float eps, const void* pack,
                       const float* b1, const float* b2, int64_t n_rows, int channels, int hidden, void* workspace,
                       int64_t workspace_bytes, hfl_stream_t stream) {
  if (out == nullptr || x == nullptr || gamma == nullptr || beta == nullptr || pack == nullptr || b1 == nullptr ||
      b2 == nullptr || n_rows < 0)
    return HFL_EINVAL;
  if (!mlp_shape_ok(channels, hidden)) return HFL_EINVAL;
  if (out == x) return HFL_EINVAL;                       // rows are re-read for the residual after other rows were written
  if (n_rows == 0) return HFL_OK;
  if (hfl_cdiv(n_rows, 16) > 0x7fffffffLL) return HFL_ECAPACITY;
  MlpFusedParams p;
  p.out = out; p.x = x; p.gamma = gamma; p.beta = beta; p.pack = static_cast<const unsigned char*>(pack);
  p.b1 = b1; p.b2 = b2; p.M = n_rows; p.eps = eps;
  p.n_tiles = (int)hfl_cdiv(n_rows, 16);
  // stagger only launches in which a workgroup walks several passes (a single pass has nothing to alternate with)
  int cus = hfl_stream_cus(static_cast<hipStream_t>(stream));
  p.stagger = p.n_tiles > (int64_t)grid_guess(p.n_tiles, cus) * 8 * (channels == 256 ? 1 : 2) ? g_mlp_stagger : 0;
  p.stagger_groups = g_mlp_stagger_groups;
  int grid = p.n_tiles < cus ? p.n_tiles : cus;
  p.full_passes = 0; p.tail_tile0 = 0; p.tail_sets = 0; p.tail_parts = 0; p.part = nullptr;
  MlpTailPlan tp = mlp_tail_plan(n_rows, channels, cus, hidden);
  if (tp.parts > 0 && workspace != nullptr && workspace_bytes >= mlp_tail_bytes(tp, n_rows, channels)) {
    p.full_passes = tp.full; p.tail_tile0 = tp.tile0; p.tail_sets = tp.sets; p.tail_parts = tp.parts;
    p.part = static_cast<float*>(workspace);
    if (grid < tp.sets * tp.parts) grid = tp.sets * tp.parts;          // (full == 0: more workgroups than row sets)
  } else {
    tp.parts = 0;
  }
  const int tpp = channels == 256 ? 8 : 16;
  p.full_units = tp.parts > 0 ? tp.full * cus : 0;
  const int n_units = tp.parts > 0 ? p.full_units + tp.sets * tp.parts : (int)hfl_cdiv(p.n_tiles, tpp);
  if (grid > n_units) grid = n_units;
  if (tp.parts > 0 && tp.full > 0 && grid < cus) grid = cus < n_units ? cus : n_units;
  // tickets only when a workgroup can get more than one unit (else the static deal is the same thing without the atomics)
  p.ticket = (g_mlp_dynamic && n_units > grid) ? ticket_slot() : nullptr;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)4 * channels * 128 + (size_t)(hidden + 3 * channels) * 4;
#define HFL_MLP_LAUNCH(CC, NT, PF, NW, LAG)                                                                     \
  {                                                                                                             \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_mlp_fused_kernel<CC, NT, PF, NW, LAG>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    ln_mlp_fused_kernel<CC, NT, PF, NW, LAG><<<grid, NW * 64, lds, s>>>(p);                                     \
  }
  if (hidden != 4 * channels) {               // the Mixer's layers: hidden = C = 256
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_mlp_fused_kernel<256, 1, 3, 8, 0, 0, 256>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    ln_mlp_fused_kernel<256, 1, 3, 8, 0, 0, 256><<<grid, 512, lds, s>>>(p);
#ifdef HFL_PROBES
  } else if (g_mlp_dbg && channels == 256) {
#define HFL_MLP_DBG(D)                                                                                          \
  {                                                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ln_mlp_fused_kernel<256, 1, 3, 8, 0, D>),           \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    ln_mlp_fused_kernel<256, 1, 3, 8, 0, D><<<grid, 512, lds, s>>>(p);                                          \
  }
    switch (g_mlp_dbg) {
      case 1: HFL_MLP_DBG(1) break;
      case 3: HFL_MLP_DBG(3) break;
      case 4: HFL_MLP_DBG(4) break;
      case 7: HFL_MLP_DBG(7) break;
      case 8: HFL_MLP_DBG(8) break;
      default: HFL_MLP_DBG(15) break;
    }
#undef HFL_MLP_DBG
#endif
  } else {
    // 8 waves (two per SIMD, 256 registers each), three stages of the weight stream ahead of the consumed one: the measured
    // best of the variants tried in rounds 3-4 (4 waves x 512 registers, two stages ahead, the second wave of a SIMD one stage
    // late: profiles/r04_waves_probe.log, r04_mlp_lag_probe.log)
    if (channels == 256) HFL_MLP_LAUNCH(256, 1, 3, 8, 0) else HFL_MLP_LAUNCH(128, 2, 3, 8, 0)
  }
#undef HFL_MLP_LAUNCH
  if (tp.parts > 0) {
    const int64_t row0 = (int64_t)tp.tile0 * 16, tail_rows = n_rows - row0;
    const int64_t n4 = tail_rows * channels / 4;
    mlp_tail_reduce_kernel<<<(unsigned)hfl_cdiv(n4, 256), 256, 0, s>>>(out, x, b2, p.part, row0, tail_rows, channels, tp.parts);
  }
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
