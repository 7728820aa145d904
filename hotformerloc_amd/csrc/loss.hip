// Ranking core of the TruncatedSmoothAP listwise loss, forward and gradient in one pass over each
// query row (reference: models/losses/truncated_smoothap.py:44-93 with the clamped temperature sigmoid
// of models/losses/loss_utils.py:40-48).  The reference materialises five (B, P, B) float tensors
// (s_diff, s_sigmoid, two masked copies, the scatter mask: 5 x 67 MB at B = 2048, P = 4) and autograd
// keeps them for the backward.  Here one workgroup owns one query q: its similarity row and both mask
// rows sit in LDS, and for each of its P selected positives p_j
//     a_j(z) = sigmoid((s_qz - s_qp_j) / tau),   r_p = 1 + sum_{z in Pos, z != p_j} a_j(z),
//     r_w = r_p + sum_{z in Neg} a_j(z),         r_j = r_p / r_w
// are two block reductions; AP_q = sum_j valid_j r_j / n_valid_q.  d AP_q / d s_q. follows in the same
// kernel from  dr/dr_p = N/r_w^2,  dr/dN = -r_p/r_w^2,  da/ds_qz = a(1-a)/tau  (0 where the exponent is
// clamped, as torch.clamp's gradient), -sum_z of it for s_qp_j.  Everything else (E E^T, top-k of the
// positives, the mean over valid queries, dE = (dS + dS^T) E) is dense torch/hipBLASLt work.
#include "hfl_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  v = hfl_group_sum<64>(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
  return t;
}

__global__ void __launch_bounds__(256)
smoothap_rows_kernel(float* __restrict__ ap, float* __restrict__ dap_ds, const float* __restrict__ S,
                     const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg,
                     const int64_t* __restrict__ idx, int B, int P, float tau) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_row = reinterpret_cast<float*>(smem);          // [B] similarities of query q
  float* s_g = s_row + B;                                  // [B] d AP_q / d s_qz
  uint8_t* s_m = reinterpret_cast<uint8_t*>(s_g + B);     // [B] bit 0 positive, bit 1 negative
  __shared__ float s_red[4];
  const int q = blockIdx.x;
  for (int z = threadIdx.x; z < B; z += blockDim.x) {
    s_row[z] = S[(int64_t)q * B + z];
    s_g[z] = 0.f;
    s_m[z] = (uint8_t)((pos[(int64_t)q * B + z] ? 1 : 0) | (neg[(int64_t)q * B + z] ? 2 : 0));
  }
  __syncthreads();
  int n_valid = 0;
  for (int j = 0; j < P; ++j) {
    const int pj = (int)idx[(int64_t)q * P + j];
    if (pj >= 0 && pj < B && (s_m[pj] & 1)) ++n_valid;
  }
  float ap_q = 0.f;
  const float inv_tau = 1.0f / tau;
  for (int j = 0; j < P; ++j) {
    const int pj = (int)idx[(int64_t)q * P + j];
    if (!(pj >= 0 && pj < B && (s_m[pj] & 1))) continue;      // fewer than P true positives: slot unused
    const float sp = s_row[pj];
    float sum_p = 0.f, sum_n = 0.f;
    for (int z = threadIdx.x; z < B; z += blockDim.x) {
      const float e = fminf(fmaxf(-(s_row[z] - sp) * inv_tau, -50.f), 50.f);
      const float a = 1.0f / (1.0f + expf(e));
      const uint8_t m = s_m[z];
      if ((m & 1) && z != pj) sum_p += a;
      if (m & 2) sum_n += a;
    }
    const float rp = 1.0f + block_sum(sum_p, s_red);
    const float nn = block_sum(sum_n, s_red);
    const float rw = rp + nn;
    ap_q += rp / rw;
    const float c_p = nn / (rw * rw) / (float)n_valid;       // d(r_j / n_valid) / d r_p
    const float c_n = -rp / (rw * rw) / (float)n_valid;      // d(r_j / n_valid) / d N
    float to_pj = 0.f;
    for (int z = threadIdx.x; z < B; z += blockDim.x) {
      const float x = -(s_row[z] - sp) * inv_tau;
      if (x < -50.f || x > 50.f) continue;                    // clamped exponent: zero gradient
      const float a = 1.0f / (1.0f + expf(x));
      const uint8_t m = s_m[z];
      const float coef = (((m & 1) && z != pj) ? c_p : 0.f) + ((m & 2) ? c_n : 0.f);
      const float gz = coef * a * (1.0f - a) * inv_tau;
      s_g[z] += gz;                                            // thread-private z: no race
      to_pj -= gz;
    }
    const float tp = block_sum(to_pj, s_red);
    if (threadIdx.x == 0) s_g[pj] += tp;
    __syncthreads();
  }
  if (threadIdx.x == 0) ap[q] = n_valid > 0 ? ap_q / (float)n_valid : 0.f;
  for (int z = threadIdx.x; z < B; z += blockDim.x) dap_ds[(int64_t)q * B + z] = s_g[z];
}

}  // namespace

extern "C" int hfl_smoothap_rows(float* ap, float* dap_ds, const float* sim, const uint8_t* pos_mask,
                                  const uint8_t* neg_mask, const int64_t* closest_pos, int batch,
                                  int positives_per_query, float tau, hfl_stream_t stream) {
  if (batch <= 0 || positives_per_query <= 0 || !(tau > 0.f)) return HFL_EINVAL;
  const size_t lds = (size_t)batch * 9;
  if (lds > 150 * 1024) return HFL_ECAPACITY;               // B <= 17066
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(smoothap_rows_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  smoothap_rows_kernel<<<batch, 256, lds, static_cast<hipStream_t>(stream)>>>(
      ap, dap_ds, sim, pos_mask, neg_mask, closest_pos, batch, positives_per_query, tau);
  HFL_RETURN_LAST_ERROR();
}
