// Batched octree construction on gfx950 (MI355X).
//
// What it replaces (reference call sites): ocnn `Octree.build_octree` per cloud in
// dataloader workers + `merge_octrees` (datasets/dataset_utils.py:89-94,
// eval/pnv_evaluate.py:173-175,123) and `construct_all_neigh` after the H2D copy
// (misc/torch_utils.py:47-51).  Semantics restated in SURVEY.md Appendix A and pinned
// by ocnn's fixtures (tests/golden/ocnn).
//
// MI355X-first design: one 1024-lane workgroup owns one cloud.  The cloud's
// (shuffled key << 32 | point index) pairs are sorted by a bitonic network that
// lives in the CU's 160 KiB LDS (up to 16384 points = 128 KiB resident; larger clouds run
// the strides >= 16384 of the same network through L2), leaves are
// found with a workgroup scan, per-leaf point averages are accumulated in original
// point order (the order of a sequential scatter-add), and every coarser level is
// produced by one more scan over the previous level's unique keys.  A second launch
// merges the per-cloud pieces into batch arrays once the host has sized them.
#include "hfl_common.h"

namespace {

constexpr int kBuildThreads = 1024;
constexpr int kLdsChunk = 16384;   // 128 KiB of (key,index) pairs per workgroup
constexpr int kMaxDepthSlots = HFL_OCTREE_MAX_DEPTH + 1;

__device__ __forceinline__ uint32_t shuffle_key(uint32_t x, uint32_t y, uint32_t z, int depth) {
  uint32_t key = 0;
  for (int i = 0; i < depth; ++i)
    key |= (((x >> i) & 1u) << (3 * i + 2)) | (((y >> i) & 1u) << (3 * i + 1)) |
           (((z >> i) & 1u) << (3 * i));
  return key;
}

__device__ __forceinline__ void unshuffle_key(uint32_t key, int depth, int& x, int& y, int& z) {
  x = y = z = 0;
  for (int i = 0; i < depth; ++i) {
    x |= ((key >> (3 * i + 2)) & 1u) << i;
    y |= ((key >> (3 * i + 1)) & 1u) << i;
    z |= ((key >> (3 * i)) & 1u) << i;
  }
}

// exclusive scan of one int per thread over a 1024-thread block; returns the
// exclusive prefix, *total = block sum.  s_wave: 17 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  __syncthreads();                       // s_wave may still be read from a previous call
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < kBuildThreads / 64; ++w) {
      const int t = s_wave[w];
      s_wave[w] = run;
      run += t;
    }
    s_wave[16] = run;
  }
  __syncthreads();
  *total = s_wave[16];
  return s_wave[wave] + incl - v;
}

struct ScratchView {
  uint32_t* skey;   // [(depth-full_depth+1)][P]  sorted unique keys per level, per cloud
  uint32_t* slot;   // [(depth-full_depth)][P]    8*parent_rank + octant
  unsigned long long* spairs;   // [B][np_stride]  sorted (key << 32 | point index) per cloud
  int64_t np_stride;
  int64_t P;
  int full_depth;
  __device__ __host__ uint32_t* key_at(int d) const { return skey + (int64_t)(d - full_depth) * P; }
  __device__ __host__ uint32_t* slot_at(int d) const { return slot + (int64_t)(d - full_depth - 1) * P; }
};

static int next_pow2(int64_t n) {
  int p = 2;
  while (p < n) p <<= 1;
  return p;
}

static int64_t level_words(int64_t P, int depth) {
  // (depth - full_depth + 1) key levels + (depth - full_depth) slot levels <= 2*depth + 1,
  // rounded up to an even count so the u64 pair region behind it stays 8-byte aligned
  return (int64_t)(2 * depth + 2) * P + (((int64_t)(2 * depth + 2) * P) & 1);
}

static ScratchView make_view(void* scratch, int64_t P, int max_points, int depth, int full_depth) {
  ScratchView v;
  v.skey = static_cast<uint32_t*>(scratch);
  v.slot = v.skey + (int64_t)(depth - full_depth + 1) * P;
  v.spairs = reinterpret_cast<unsigned long long*>(v.skey + level_words(P, depth));
  v.np_stride = next_pow2(max_points);
  v.P = P;
  v.full_depth = full_depth;
  return v;
}

// ------------------------------------------------------------------ stage 1
__global__ void __launch_bounds__(kBuildThreads)
build_cloud_kernel(const float* __restrict__ points, const int64_t* __restrict__ cloud_off,
                   int batch, int depth, int full_depth, ScratchView sv,
                   float* __restrict__ leaf_points, int32_t* __restrict__ counts, int ch_cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(smem);
  int* s_wave = reinterpret_cast<int*>(smem + (size_t)ch_cap * 8);

  const int b = blockIdx.x;
  const int64_t base = cloud_off[b];
  const int n = (int)(cloud_off[b + 1] - base);
  const int tid = threadIdx.x;
  const float scale = (float)(1 << (depth - 1));
  const uint32_t mask = (1u << depth) - 1u;
  unsigned long long* gp = sv.spairs + (int64_t)b * sv.np_stride;
  int np2 = 2;
  while (np2 < n) np2 <<= 1;
  const int CH = np2 < ch_cap ? np2 : ch_cap;        // LDS-resident chunk of the bitonic network

  // 1+2a. keys, and a full bitonic sort of every CH-chunk inside LDS.
  //       p = (x + 1) * 2^(depth-1), truncated, masked to `depth` bits (ocnn xyz2key).
  for (int c0 = 0; c0 < np2; c0 += CH) {
    for (int i = tid; i < CH; i += kBuildThreads) {
      const int gi = c0 + i;
      unsigned long long pr = ~0ull;
      if (gi < n) {
        const float* p = points + (base + gi) * 3;
        const uint32_t ix = (uint32_t)(int)((p[0] + 1.0f) * scale) & mask;
        const uint32_t iy = (uint32_t)(int)((p[1] + 1.0f) * scale) & mask;
        const uint32_t iz = (uint32_t)(int)((p[2] + 1.0f) * scale) & mask;
        pr = ((unsigned long long)shuffle_key(ix, iy, iz, depth) << 32) | (uint32_t)gi;
      }
      pairs[i] = pr;
    }
    __syncthreads();
    for (int k = 2; k <= CH; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (CH >> 1); t += kBuildThreads) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = ((c0 + lo) & k) == 0;
          const unsigned long long a = pairs[lo], c = pairs[hi];
          if ((a > c) == up) {
            pairs[lo] = c;
            pairs[hi] = a;
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < CH; i += kBuildThreads) gp[c0 + i] = pairs[i];
    __syncthreads();
  }
  // 2b. clouds larger than one chunk: the remaining network stages, strides >= CH through
  //     global memory (L2-resident, one workgroup), strides < CH back inside LDS.
  for (int k = 2 * CH; k <= np2; k <<= 1) {
    for (int j = k >> 1; j >= CH; j >>= 1) {
      for (int t = tid; t < (np2 >> 1); t += kBuildThreads) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const unsigned long long a = gp[lo], c = gp[hi];
        if ((a > c) == up) {
          gp[lo] = c;
          gp[hi] = a;
        }
      }
      __syncthreads();
    }
    for (int c0 = 0; c0 < np2; c0 += CH) {
      for (int i = tid; i < CH; i += kBuildThreads) pairs[i] = gp[c0 + i];
      __syncthreads();
      for (int j = CH >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (CH >> 1); t += kBuildThreads) {
          const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int hi = lo | j;
          const bool up = ((c0 + lo) & k) == 0;
          const unsigned long long a = pairs[lo], c = pairs[hi];
          if ((a > c) == up) {
            pairs[lo] = c;
            pairs[hi] = a;
          }
        }
        __syncthreads();
      }
      for (int i = tid; i < CH; i += kBuildThreads) gp[c0 + i] = pairs[i];
      __syncthreads();
    }
  }

  // 3. leaves: heads of equal-key runs; averages in original point order
  const int per = (np2 + kBuildThreads - 1) / kBuildThreads;
  const int i0 = tid * per, i1 = min(i0 + per, n);
  int heads = 0;
  for (int i = i0; i < i1; ++i) {
    const uint32_t ki = (uint32_t)(gp[i] >> 32);
    heads += (i == 0 || (uint32_t)(gp[i - 1] >> 32) != ki) ? 1 : 0;
  }
  int total = 0;
  int rank = block_excl_scan(heads, s_wave, &total);
  uint32_t* key_d = sv.key_at(depth) + base;
  for (int i = i0; i < i1; ++i) {
    const uint32_t ki = (uint32_t)(gp[i] >> 32);
    if (i == 0 || (uint32_t)(gp[i - 1] >> 32) != ki) {
      key_d[rank] = ki;
      float sx = 0.f, sy = 0.f, sz = 0.f;
      int cnt = 0;
      for (int j = i; j < n && (uint32_t)(gp[j] >> 32) == ki; ++j) {
        const float* p = points + (base + (uint32_t)gp[j]) * 3;
        sx += (p[0] + 1.0f) * scale;
        sy += (p[1] + 1.0f) * scale;
        sz += (p[2] + 1.0f) * scale;
        ++cnt;
      }
      const float c = (float)cnt;
      float* o = leaf_points + (base + rank) * 3;
      o[0] = sx / c;
      o[1] = sy / c;
      o[2] = sz / c;
      ++rank;
    }
  }
  int cur = total;
  if (tid == 0) counts[depth * batch + b] = cur;
  __syncthreads();   // key_d written by this workgroup, read below by other lanes

  // 4. coarser levels: parents of the unique keys of level d
  for (int d = depth; d > full_depth; --d) {
    const uint32_t* kd = sv.key_at(d) + base;
    uint32_t* kp = sv.key_at(d - 1) + base;
    uint32_t* sl = sv.slot_at(d) + base;
    const int perd = (cur + kBuildThreads - 1) / kBuildThreads;
    const int a0 = min(tid * perd, cur), a1 = min(a0 + perd, cur);
    int h = 0;
    for (int i = a0; i < a1; ++i) h += (i == 0 || (kd[i - 1] >> 3) != (kd[i] >> 3)) ? 1 : 0;
    int tot = 0;
    int r = block_excl_scan(h, s_wave, &tot);   // heads before this lane's chunk
    for (int i = a0; i < a1; ++i) {
      const uint32_t ki = kd[i];
      if (i == 0 || (kd[i - 1] >> 3) != (ki >> 3)) {
        kp[r] = ki >> 3;
        ++r;
      }
      sl[i] = (uint32_t)(r - 1) * 8u + (ki & 7u);   // parent rank = heads in [0..i] - 1
    }
    cur = tot;
    if (tid == 0) counts[(d - 1) * batch + b] = cur;
    __syncthreads();
  }
  if (tid < full_depth) counts[tid * batch + b] = 1 << (3 * tid);
}

// ------------------------------------------------------------------ stage 2
struct MergeArgs {
  int64_t* keys[kMaxDepthSlots];
  int32_t* children[kMaxDepthSlots];
  int64_t* nkeys[kMaxDepthSlots];
  int32_t* nidx[kMaxDepthSlots];
};

__device__ __forceinline__ int find_cloud(const int64_t* __restrict__ off, int batch, int64_t t) {
  int lo = 0, hi = batch;   // largest b with off[b] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= t) lo = mid; else hi = mid;
  }
  return lo;
}

// grid.y = level index (d = full_depth + y); one thread per scratch slot
__global__ void merge_sparse_kernel(ScratchView sv, const int64_t* __restrict__ cloud_off,
                                    const int32_t* __restrict__ cum_nne, int batch, int depth,
                                    int full_depth, MergeArgs a,
                                    const float* __restrict__ leaf_points,
                                    float* __restrict__ points_out) {
  const int d = full_depth + blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= sv.P) return;
  const int b = find_cloud(cloud_off, batch, t);
  const int64_t i = t - cloud_off[b];
  const int32_t* cum_d = cum_nne + (int64_t)d * (batch + 1);
  const int64_t cnt = cum_d[b + 1] - cum_d[b];
  if (i >= cnt) return;
  const int64_t g = cum_d[b] + i;
  const uint32_t key = sv.key_at(d)[t];
  const int64_t bbits = (int64_t)b << 48;
  a.nkeys[d][g] = bbits | (int64_t)key;
  int64_t pos;
  if (d == full_depth) {
    pos = ((int64_t)b << (3 * d)) + key;
  } else {
    const int32_t* cum_p = cum_nne + (int64_t)(d - 1) * (batch + 1);
    pos = 8 * (int64_t)cum_p[b] + sv.slot_at(d)[t];
  }
  a.nidx[d][g] = (int32_t)pos;
  a.children[d][pos] = (int32_t)g;
  if (d < depth && a.keys[d + 1] != nullptr) {
    int64_t* kc = a.keys[d + 1] + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) kc[j] = bbits | (int64_t)(key * 8u + j);
  }
  if (d == depth) {
    const float* src = leaf_points + t * 3;
    float* dst = points_out + g * 3;
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
  }
}

// depths <= full_depth are full: B * 8^d nodes
__global__ void merge_full_kernel(int batch, int full_depth, MergeArgs a) {
  const int d = blockIdx.y;
  const int64_t per = (int64_t)1 << (3 * d);
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= per * batch) return;
  const int64_t b = t / per, m = t % per;
  if (a.keys[d] != nullptr) a.keys[d][t] = (b << 48) | m;
  if (d < full_depth) {
    a.children[d][t] = (int32_t)t;
    a.nkeys[d][t] = (b << 48) | m;
    a.nidx[d][t] = (int32_t)t;
  }
}

// ---------------------------------------------------------------- neighbours
__global__ void neigh_full_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ children,
                                  const int64_t* __restrict__ nkeys, int64_t nne, int depth) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nne * 27) return;
  const int64_t g = t / 27;
  const int o = (int)(t % 27);
  const int64_t key = nkeys[g];
  const int64_t b = key >> 48;
  int x, y, z;
  unshuffle_key((uint32_t)(key & 0xFFFFFFFFll), depth, x, y, z);
  x += o / 9 - 1;
  y += (o / 3) % 3 - 1;
  z += o % 3 - 1;
  const int bound = 1 << depth;
  int32_t res = -1;
  if (x >= 0 && y >= 0 && z >= 0 && x < bound && y < bound && z < bound)
    res = children[(b << (3 * depth)) + shuffle_key((uint32_t)x, (uint32_t)y, (uint32_t)z, depth)];
  out[t] = res;
}

__global__ void neigh_walk_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ neigh_parent,
                                  const int32_t* __restrict__ nidx,
                                  const int32_t* __restrict__ children, int64_t nne) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nne * 27) return;
  const int64_t g = t / 27;
  const int o = (int)(t % 27);
  const int32_t pos = nidx[g];
  const int64_t parent = pos >> 3;
  const int oct = pos & 7;
  const int tx = ((oct >> 2) & 1) + o / 9 - 1;        // in [-1, 2]
  const int ty = ((oct >> 1) & 1) + (o / 3) % 3 - 1;
  const int tz = (oct & 1) + o % 3 - 1;
  const int po = ((tx >> 1) + 1) * 9 + ((ty >> 1) + 1) * 3 + ((tz >> 1) + 1);   // floor(t/2)+1
  const int co = ((tx & 1) << 2) | ((ty & 1) << 1) | (tz & 1);
  const int32_t q = neigh_parent[parent * 27 + po];
  out[t] = q < 0 ? -1 : children[(int64_t)q * 8 + co];
}

__global__ void token_meta_kernel(uint32_t* __restrict__ meta, const int64_t* __restrict__ nkeys,
                                  int64_t n, int depth) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int64_t key = nkeys[t];
  int x, y, z;
  unshuffle_key((uint32_t)(key & 0xFFFFFFFFll), depth, x, y, z);
  meta[2 * t] = (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20);
  meta[2 * t + 1] = (uint32_t)(key >> 48);
}

// ---------------------------------------------------------------------------------------------------------
// Live-tap lists of a (rows, taps) index table (taps <= 32): the tap-major list of the (row, tap) pairs whose
// neighbour exists, for the octree convolutions that run over live taps only (model.py OctreeConv).
//   src   (P)          input row of every pair, pairs ordered by tap, then by row
//   slot  (rows, taps) position of (row, tap) in that list, -1 where the neighbour is missing
//   edges (taps + 1)   pairs of tap k are [edges[k], edges[k+1])
// Three launches, no atomics, fixed order: per-block per-tap counts (wave ballots), a scan over blocks
// (one wave per tap), then the fill pass recomputes the ballots and writes.
constexpr int kTapRows = 1024;         // rows per block (256 threads x 4)
constexpr int kTapMax = 32;
constexpr int kTapTables = 16;         // tables per launch (hfl_tap_lists_multi)

// Several tables per launch (every convolution depth of a batch at once): flat block index -> (table, block of the table).
struct TapTable {
  const int32_t* table;
  int32_t* src;
  int32_t* slot;
  int32_t* edges;
  int32_t* block_counts;      // (blocks of this table, taps): per-block per-tap counts, then their exclusive scan
  int64_t rows;
  int taps;
  int first_block;
};
struct TapMulti {
  TapTable t[kTapTables];
  int n;
  int total_blocks;
};

__device__ __forceinline__ const TapTable& tap_find(const TapMulti& m, int block, int& local) {
  int i = 0;
  while (i + 1 < m.n && block >= m.t[i + 1].first_block) ++i;
  local = block - m.t[i].first_block;
  return m.t[i];
}

// One wave's 64 rows of a (rows, TAPS) table into registers, lane = row.  The table is row-major with TAPS * 4 bytes per
// row: a lane reading its own row directly touches a different cache line per lane and tap (TAPS = 27: 108-B stride, 27
// requests of 64 lines each; the fill pass took 66 us for 130 k rows that way).  Instead the wave copies its contiguous
// 64 x TAPS words with coalesced loads into its LDS block (word j * 64 + lane per instruction) and every lane reads its row
// back (stride TAPS words: conflict-free for odd TAPS).  TAPS = 8 rows are 32 B: two 16-B loads per lane are coalesced as is.
template <int TAPS>
__device__ __forceinline__ void tap_load_rows(const int32_t* __restrict__ table, int64_t rows, int64_t row0, int32_t* lds_wave,
                                              int lane, int32_t (&v)[kTapMax], int taps_rt) {
  if constexpr (TAPS == 8) {
    const int64_t r = row0 + lane;
    int4 a = make_int4(-1, -1, -1, -1), b = a;
    if (r < rows) {
      a = reinterpret_cast<const int4*>(table)[r * 2];
      b = reinterpret_cast<const int4*>(table)[r * 2 + 1];
    }
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else if constexpr (TAPS > 0) {
    const int64_t w0 = row0 * TAPS, wend = rows * TAPS;
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      const int64_t w = w0 + j * 64 + lane;
      lds_wave[j * 64 + lane] = w < wend ? table[w] : -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < TAPS; ++k) v[k] = lds_wave[lane * TAPS + k];
    __builtin_amdgcn_wave_barrier();
  } else {
    const int64_t r = row0 + lane;
#pragma unroll
    for (int k = 0; k < kTapMax; ++k)
      if (k < taps_rt) v[k] = r < rows ? table[r * taps_rt + k] : -1;
  }
}

template <int TAPS>
__device__ __forceinline__ void tap_count_body(const TapTable& t, int block, int32_t* lds) {
  __shared__ int32_t wave_cnt[4][kTapMax];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int taps = TAPS > 0 ? TAPS : t.taps;
  int32_t acc[kTapMax];
#pragma unroll
  for (int k = 0; k < kTapMax; ++k) acc[k] = 0;
  const int64_t row0 = (int64_t)block * kTapRows;
  int32_t* lds_wave = lds + wave * (64 * (TAPS > 0 ? TAPS : 1));
  for (int it = 0; it < kTapRows / 256; ++it) {
    int32_t v[kTapMax];
    tap_load_rows<TAPS>(t.table, t.rows, row0 + it * 256 + wave * 64, lds_wave, lane, v, taps);
#pragma unroll
    for (int k = 0; k < kTapMax; ++k)
      if (k < taps) acc[k] += __popcll(__ballot(v[k] >= 0));
  }
  if (lane == 0)
    for (int k = 0; k < taps; ++k) wave_cnt[wave][k] = acc[k];
  __syncthreads();
  if ((int)threadIdx.x < taps)
    t.block_counts[(int64_t)block * taps + threadIdx.x] =
        wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
tap_count_kernel(const TapMulti m) {
  __shared__ int32_t lds[4 * 64 * 27];
  int block;
  const TapTable& t = tap_find(m, blockIdx.x, block);
  if (t.taps == 27) tap_count_body<27>(t, block, lds);
  else if (t.taps == 8) tap_count_body<8>(t, block, lds);
  else tap_count_body<0>(t, block, lds);
}

// one workgroup per table, one wave per tap (16 waves, taps k and k + 16): exclusive scan of that tap's block counts (in
// place), totals -> edges
__global__ void __launch_bounds__(1024)
tap_scan_kernel(const TapMulti m) {
  __shared__ int32_t total[kTapMax];
  const TapTable& t = m.t[blockIdx.x];
  const int taps = t.taps;
  const int nblocks = (blockIdx.x + 1 < (unsigned)m.n ? m.t[blockIdx.x + 1].first_block : m.total_blocks) - t.first_block;
  int32_t* block_counts = t.block_counts;
  const int lane = threadIdx.x & 63;
  for (int k = threadIdx.x >> 6; k < taps; k += 16) {
    int32_t run = 0;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
      const int b = b0 + lane;
      const int32_t v = b < nblocks ? block_counts[(int64_t)b * taps + k] : 0;
      int32_t inc = v;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int32_t u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
      }
      if (b < nblocks) block_counts[(int64_t)b * taps + k] = run + inc - v;
      run += __shfl(inc, 63, 64);
    }
    if (lane == 0) total[k] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t e = 0;
    for (int k = 0; k < taps; ++k) {
      t.edges[k] = e;
      e += total[k];
    }
    t.edges[taps] = e;
  }
}

template <int TAPS>
__device__ __forceinline__ void tap_fill_body(const TapTable& t, int block, int32_t* lds) {
  __shared__ int32_t wave_cnt[4][kTapMax];
  __shared__ int32_t base[kTapMax];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int taps = TAPS > 0 ? TAPS : t.taps;
  if ((int)threadIdx.x < taps) base[threadIdx.x] = t.edges[threadIdx.x] + t.block_counts[(int64_t)block * taps + threadIdx.x];
  const int64_t row0 = (int64_t)block * kTapRows;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  int32_t* lds_wave = lds + wave * (64 * (TAPS > 0 ? TAPS : 1));
  for (int it = 0; it < kTapRows / 256; ++it) {
    const int64_t rw = row0 + it * 256 + wave * 64;       // first row of this wave
    const int64_t r = rw + lane;
    const bool in = r < t.rows;
    int32_t v[kTapMax];
    unsigned long long mk[kTapMax];
    tap_load_rows<TAPS>(t.table, t.rows, rw, lds_wave, lane, v, taps);
#pragma unroll
    for (int k = 0; k < kTapMax; ++k)
      if (k < taps) mk[k] = __ballot(v[k] >= 0);
    __syncthreads();                       // base[] ready (first pass) / updated (later passes)
    if (lane == 0)
      for (int k = 0; k < taps; ++k) wave_cnt[wave][k] = __popcll(mk[k]);
    __syncthreads();
    int32_t pos[kTapMax];
#pragma unroll
    for (int k = 0; k < kTapMax; ++k) {
      if (k < taps) {
        int32_t off = base[k];
        for (int w = 0; w < wave; ++w) off += wave_cnt[w][k];
        pos[k] = v[k] >= 0 ? off + __popcll(mk[k] & lt) : -1;
        if (in && v[k] >= 0) t.src[pos[k]] = v[k];
      }
    }
    // slot rows leave the way the table rows came in: through the wave's LDS block, whole lines per store instruction
    if constexpr (TAPS == 8) {
      if (in) {
        reinterpret_cast<int4*>(t.slot)[r * 2] = make_int4(pos[0], pos[1], pos[2], pos[3]);
        reinterpret_cast<int4*>(t.slot)[r * 2 + 1] = make_int4(pos[4], pos[5], pos[6], pos[7]);
      }
    } else if constexpr (TAPS > 0) {
#pragma unroll
      for (int k = 0; k < TAPS; ++k) lds_wave[lane * TAPS + k] = pos[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int64_t w0 = rw * TAPS, wend = t.rows * TAPS;
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        const int64_t w = w0 + j * 64 + lane;
        if (w < wend) t.slot[w] = lds_wave[j * 64 + lane];
      }
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int k = 0; k < kTapMax; ++k)
        if (k < taps && in) t.slot[r * taps + k] = pos[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < taps)
      base[threadIdx.x] += wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] +
                           wave_cnt[3][threadIdx.x];
  }
}

__global__ void __launch_bounds__(256)
tap_fill_kernel(const TapMulti m) {
  __shared__ int32_t lds[4 * 64 * 27];
  int block;
  const TapTable& t = tap_find(m, blockIdx.x, block);
  if (t.taps == 27) tap_fill_body<27>(t, block, lds);
  else if (t.taps == 8) tap_fill_body<8>(t, block, lds);
  else tap_fill_body<0>(t, block, lds);
}

}  // namespace

extern "C" {

int64_t hfl_octree_scratch_bytes(int64_t total_points, int batch, int max_points, int depth) {
  return level_words(total_points, depth) * (int64_t)sizeof(uint32_t) +
         (int64_t)batch * next_pow2(max_points) * (int64_t)sizeof(unsigned long long) + 64;
}

int hfl_octree_build_clouds(const float* points, const int64_t* cloud_offsets, int batch,
                              int64_t total_points, int max_points, int depth, int full_depth,
                              void* scratch, float* leaf_points, int32_t* counts,
                              hfl_stream_t stream) {
  if (batch <= 0 || depth < 1 || depth > HFL_OCTREE_MAX_DEPTH || full_depth < 0 || full_depth > depth)
    return HFL_EINVAL;
  if (max_points > HFL_OCTREE_MAX_POINTS) return HFL_ECAPACITY;
  if (max_points < 1) return HFL_EINVAL;
  int ch_cap = next_pow2(max_points);
  if (ch_cap > kLdsChunk) ch_cap = kLdsChunk;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ScratchView sv = make_view(scratch, total_points, max_points, depth, full_depth);
  const size_t lds = (size_t)ch_cap * 8 + 32 * sizeof(int);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(build_cloud_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return (int)e;
  build_cloud_kernel<<<batch, kBuildThreads, lds, s>>>(points, cloud_offsets, batch, depth, full_depth,
                                                       sv, leaf_points, counts, ch_cap);
  HFL_RETURN_LAST_ERROR();
}

int hfl_octree_merge(const void* scratch, const int64_t* cloud_offsets, const int32_t* cum_nne,
                     int batch, int depth, int full_depth, int64_t total_points,
                     int64_t* const* keys, int32_t* const* children, int64_t* const* nkeys,
                     int32_t* const* nidx, const float* leaf_points, float* points_out,
                     hfl_stream_t stream) {
  if (batch <= 0 || depth < 1 || depth > HFL_OCTREE_MAX_DEPTH || full_depth < 0 || full_depth > depth)
    return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  MergeArgs a;
  for (int d = 0; d < kMaxDepthSlots; ++d) {
    const bool in = d <= depth;
    a.keys[d] = (in && keys != nullptr) ? keys[d] : nullptr;
    a.children[d] = in ? children[d] : nullptr;
    a.nkeys[d] = in ? nkeys[d] : nullptr;
    a.nidx[d] = in ? nidx[d] : nullptr;
  }
  ScratchView sv = make_view(const_cast<void*>(scratch), total_points, 2, depth, full_depth);
  {
    const int64_t per = ((int64_t)1 << (3 * full_depth)) * batch;
    dim3 grid((unsigned)hfl_cdiv(per, 256), (unsigned)(full_depth + 1));
    merge_full_kernel<<<grid, 256, 0, s>>>(batch, full_depth, a);
  }
  if (total_points > 0) {
    dim3 grid((unsigned)hfl_cdiv(total_points, 256), (unsigned)(depth - full_depth + 1));
    merge_sparse_kernel<<<grid, 256, 0, s>>>(sv, cloud_offsets, cum_nne, batch, depth, full_depth, a,
                                             leaf_points, points_out);
  }
  HFL_RETURN_LAST_ERROR();
}

int hfl_octree_neigh(int32_t* neigh_out, const int32_t* neigh_parent, const int32_t* nidx,
                     const int32_t* children, const int64_t* nkeys, int64_t nne, int depth,
                     int full_depth, hfl_stream_t stream) {
  if (nne < 0 || depth < 0) return HFL_EINVAL;
  if (nne == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned blocks = (unsigned)hfl_cdiv(nne * 27, 256);
  if (depth <= full_depth)
    neigh_full_kernel<<<blocks, 256, 0, s>>>(neigh_out, children, nkeys, nne, depth);
  else
    neigh_walk_kernel<<<blocks, 256, 0, s>>>(neigh_out, neigh_parent, nidx, children, nne);
  HFL_RETURN_LAST_ERROR();
}

/* tok_meta (n,2) uint32 from the non-empty node keys of one depth:
 * [x | y<<10 | z<<20, batch id]  (ocnn key2xyz / batch_id, models/octree.py:132,273-275) */
int hfl_token_meta(uint32_t* tok_meta, const int64_t* nkeys, int64_t n, int depth,
                   hfl_stream_t stream) {
  if (n < 0 || depth < 0 || depth > HFL_OCTREE_MAX_DEPTH) return HFL_EINVAL;
  if (n == 0) return HFL_OK;
  token_meta_kernel<<<(unsigned)hfl_cdiv(n, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(
      tok_meta, nkeys, n, depth);
  HFL_RETURN_LAST_ERROR();
}

int64_t hfl_tap_lists_workspace(int64_t rows, int taps) {
  return hfl_cdiv(rows > 0 ? rows : 1, kTapRows) * (int64_t)taps * (int64_t)sizeof(int32_t);
}

/* Live-tap lists of a (rows, taps) int32 index table (see above): src must hold rows*taps entries (only the
 * first edges[taps] are written), slot (rows, taps), edges (taps + 1), all on the device; no host sync. */
int hfl_tap_lists_multi(int n, int32_t* const* src, int32_t* const* slot, int32_t* const* edges, const int32_t* const* table,
                        const int64_t* rows, const int32_t* taps, void* const* workspace, hfl_stream_t stream) {
  if (n < 1 || n > kTapTables || src == nullptr || slot == nullptr || edges == nullptr || table == nullptr || rows == nullptr ||
      taps == nullptr || workspace == nullptr)
    return HFL_EINVAL;
  TapMulti m;
  m.n = n;
  int first = 0;
  for (int i = 0; i < n; ++i) {
    if (rows[i] < 0 || taps[i] < 1 || taps[i] > kTapMax) return HFL_EINVAL;
    if (rows[i] * taps[i] > 0x7fffffffLL) return HFL_ECAPACITY;
    m.t[i].table = table[i]; m.t[i].src = src[i]; m.t[i].slot = slot[i]; m.t[i].edges = edges[i];
    m.t[i].block_counts = static_cast<int32_t*>(workspace[i]);
    m.t[i].rows = rows[i]; m.t[i].taps = taps[i]; m.t[i].first_block = first;
    first += (int)hfl_cdiv(rows[i] > 0 ? rows[i] : 1, kTapRows);
  }
  for (int i = n; i < kTapTables; ++i) m.t[i] = m.t[0];
  m.total_blocks = first;
  hipStream_t s = static_cast<hipStream_t>(stream);
  tap_count_kernel<<<first, 256, 0, s>>>(m);
  tap_scan_kernel<<<n, 1024, 0, s>>>(m);
  tap_fill_kernel<<<first, 256, 0, s>>>(m);
  HFL_RETURN_LAST_ERROR();
}

int hfl_tap_lists(int32_t* src, int32_t* slot, int32_t* edges, const int32_t* table, int64_t rows, int taps,
                  void* workspace, hfl_stream_t stream) {
  const int32_t t = taps;
  return hfl_tap_lists_multi(1, &src, &slot, &edges, &table, &rows, &t, &workspace, stream);
}

/* Row-tile table of the grouped tap GEMM (hfl_linear_x3_grouped) built on the device from the tap edges hfl_tap_lists wrote:
 * tiles (n_tiles, 3) int32 = {first pair, pairs (<= tile_rows), tap * w_rows}, tap-major, no tile straddles a tap; n_tiles =
 * sum over taps of ceil(pairs_k / tile_rows) (the caller knows it from its host copy of the edges).  Replaces a host-side
 * numpy table + a blocking pageable copy per convolution and forward. */
__global__ void __launch_bounds__(64) tap_tiles_kernel(int32_t* __restrict__ tiles, const int32_t* __restrict__ edges, int taps,
                                                       int tile_rows, int w_rows) {
  __shared__ int first[kTapMax + 1];
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int k = 0; k < taps; ++k) {
      first[k] = acc;
      acc += (edges[k + 1] - edges[k] + tile_rows - 1) / tile_rows;
    }
    first[taps] = acc;
  }
  __syncthreads();
  for (int k = 0; k < taps; ++k) {
    const int a = edges[k], b = edges[k + 1];
    for (int t = threadIdx.x; t < first[k + 1] - first[k]; t += blockDim.x) {
      const int r0 = a + t * tile_rows;
      int32_t* o = tiles + 3 * (int64_t)(first[k] + t);
      o[0] = r0;
      o[1] = b - r0 < tile_rows ? b - r0 : tile_rows;
      o[2] = k * w_rows;
    }
  }
}

int hfl_tap_tiles(int32_t* tiles, const int32_t* edges, int taps, int tile_rows, int w_rows, hfl_stream_t stream) {
  if (tiles == nullptr || edges == nullptr || taps < 1 || taps > kTapMax || tile_rows < 1) return HFL_EINVAL;
  tap_tiles_kernel<<<1, 64, 0, static_cast<hipStream_t>(stream)>>>(tiles, edges, taps, tile_rows, w_rows);
  HFL_RETURN_LAST_ERROR();
}

/* Gather index of a ragged row stream padded per cloud (attentional pooling head, models/layers/pooling.py:209-233):
 * out (B * nmax) int64, out[b * nmax + j] = row_off[b] + j for j < row_off[b+1] - row_off[b], else row_off[B] (the caller's
 * zero row).  row_off (B + 1) int64 on the device. */
__global__ void __launch_bounds__(256) pad_index_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ row_off, int B,
                                                        int64_t nmax) {
  const int64_t total = (int64_t)B * nmax;
  const int64_t sentinel = row_off[B];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / nmax);
    const int64_t j = i - (int64_t)b * nmax;
    const int64_t a = row_off[b];
    out[i] = j < row_off[b + 1] - a ? a + j : sentinel;
  }
}

/* The padded copy itself (models/layers/pooling.py:209-233 splits the ragged stream per cloud and pads with zeros): out
 * (B, nmax, C) f32, out[b][j] = x[row_off[b] + j] for j < n_b, zeros beyond -- one pass, no index table, no appended zero row. */
__global__ void __launch_bounds__(256) pad_rows_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                       const int64_t* __restrict__ row_off, int B, int64_t nmax, int c4) {
  const int64_t total = (int64_t)B * nmax * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4;
    const int v = (int)(i - row * c4);
    const int b = (int)(row / nmax);
    const int64_t j = row - (int64_t)b * nmax;
    const int64_t a = row_off[b];
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < row_off[b + 1] - a) val = reinterpret_cast<const float4*>(x)[(a + j) * c4 + v];
    reinterpret_cast<float4*>(out)[i] = val;
  }
}

int hfl_pad_rows(float* out, const float* x, const int64_t* row_off, int batch, int64_t nmax, int64_t channels,
                 hfl_stream_t stream) {
  if (out == nullptr || x == nullptr || row_off == nullptr || batch < 0 || nmax < 0 || channels <= 0 || channels % 4 != 0)
    return HFL_EINVAL;
  if (batch == 0 || nmax == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t need = hfl_cdiv((int64_t)batch * nmax * (channels / 4), 256);
  const int64_t cap = (int64_t)hfl_stream_cus(s) * 16;
  pad_rows_kernel<<<(unsigned)(need < cap ? need : cap), 256, 0, s>>>(out, x, row_off, batch, nmax, (int)(channels / 4));
  HFL_RETURN_LAST_ERROR();
}

int hfl_pad_index(int64_t* out, const int64_t* row_off, int batch, int64_t nmax, hfl_stream_t stream) {
  if (out == nullptr || row_off == nullptr || batch < 0 || nmax < 0) return HFL_EINVAL;
  if (batch == 0 || nmax == 0) return HFL_OK;
  const int64_t need = hfl_cdiv((int64_t)batch * nmax, 256);
  pad_index_kernel<<<(unsigned)(need < 1024 ? need : 1024), 256, 0, static_cast<hipStream_t>(stream)>>>(out, row_off, batch, nmax);
  HFL_RETURN_LAST_ERROR();
}

}  // extern "C"
