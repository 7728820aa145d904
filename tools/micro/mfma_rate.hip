// MFMA issue rate on gfx950 for the shapes the split-precision kernels use: v_mfma_f32_16x16x32_bf16 with (a) independent
// accumulators, (b) chains of 3 dependent MFMAs on one accumulator (the x_lo w_hi + x_hi w_lo + x_hi w_hi pattern),
// (c) one accumulator only; 1 or 2 waves per SIMD; with and without LDS fragment reads between them; random operands.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o tools/micro/bin/mfma_rate && tools/micro/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int LDSR>
__global__ void __launch_bounds__(512, 2) k(float* out, const uint32_t* seedbuf, int iters) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x)
    reinterpret_cast<uint32_t*>(lds)[i] = (seedbuf[i & 4095] & 0x3f803f80u) ^ 0x3c003c00u;   // small bf16 values
  __syncthreads();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(lds + ((lane * 16 + i * 1024) & 65535));
    b[i] = *reinterpret_cast<const bf16x8*>(lds + ((lane * 16 + i * 1024 + 8192) & 65535));
  }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int off = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (LDSR) {       // 4 fragment reads per 6 MFMAs, as the fused MLP kernel at C = 256
        a[0] = *reinterpret_cast<const bf16x8*>(lds + ((off + (it * 4 + g) * 4096) & 65535));
        a[1] = *reinterpret_cast<const bf16x8*>(lds + (((off ^ 64) + (it * 4 + g) * 4096) & 65535));
        a[2] = *reinterpret_cast<const bf16x8*>(lds + ((off + (it * 4 + g) * 4096 + 2048) & 65535));
        a[3] = *reinterpret_cast<const bf16x8*>(lds + (((off ^ 64) + (it * 4 + g) * 4096 + 2048) & 65535));
      }
      if (MODE == 0) {          // 6 independent accumulators
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j & 3], acc[j], 0, 0, 0);
      } else if (MODE == 1) {   // 2 chains of 3 dependent
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * c], b[1], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * c + 1], b[0], acc[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * c], b[0], acc[c], 0, 0, 0);
        }
      } else {                  // one chain of 6
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j & 3], acc[0], 0, 0, 0);
      }
    }
  }
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int MODE, int LDSR>
void run(const char* name, int threads) {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  uint32_t* seed; hipMalloc(&seed, 4096 * 4);
  uint32_t h[4096]; uint32_t x = 12345; for (int i = 0; i < 4096; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  k<MODE, LDSR><<<256, threads>>>(out, seed, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, LDSR><<<256, threads>>>(out, seed, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 24 * (threads / 256);
  const double tf = 256.0 * 4 * mfma_per_simd * 16384.0 / (ms * 1e-3) / 1e12;
  printf("%-34s %d waves/SIMD: %.3f ms, %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz), %.0f TF/s\n", name, threads / 256, ms,
         ms * 1e6 / mfma_per_simd, ms * 1e-3 * 2.4e9 / mfma_per_simd, tf);
  hipFree(out); hipFree(seed);
}
int main() {
  run<0, 0>("independent x6", 256); run<0, 0>("independent x6", 512);
  run<1, 0>("2 chains of 3 dependent", 256); run<1, 0>("2 chains of 3 dependent", 512);
  run<2, 0>("1 chain of 6 dependent", 256); run<2, 0>("1 chain of 6 dependent", 512);
  run<0, 1>("independent x6 + 4 ds_read_b128", 256); run<0, 1>("independent x6 + 4 ds_read_b128", 512);
  run<1, 1>("2 chains of 3 + 4 ds_read_b128", 256); run<1, 1>("2 chains of 3 + 4 ds_read_b128", 512);
  return 0;
}
