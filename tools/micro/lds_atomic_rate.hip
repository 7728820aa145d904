// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64, distinct vs shared addresses.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_rate.hip -o /tmp/lds_atomic_rate && /tmp/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE, int SPREAD>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ unsigned long long tab[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // SPREAD = number of distinct addresses the 64 lanes of a wave-instruction hit
  int idx = wave * 512 + (lane % SPREAD);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = idx + u * 64;
      if (MODE == 0) atomicAdd(reinterpret_cast<float*>(tab) + a, 1.0f);
      if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int*>(tab) + a, 1u);
      if (MODE == 2) atomicAdd(tab + (a & 4095), 1ull);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)tab[0];
}
template <int MODE, int SPREAD>
void run(const char* name) {
  float* out; hipMalloc(&out, 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, blocks = 256 * 2;
  k<MODE, SPREAD><<<blocks, 256>>>(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, SPREAD><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per CU: 2 blocks x 4 waves x iters x 8 wave-instructions
  const double instr_per_cu = 2.0 * 4 * iters * 8;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("%-10s %2d distinct addresses per wave-instruction: %.1f cycles per wave-instruction per CU (%.2f lane-ops per cycle)\n", name, SPREAD,
         cycles / instr_per_cu, 64.0 * instr_per_cu / cycles);
  hipFree(out);
}
int main() {
  run<0, 64>("ds_add_f32"); run<0, 16>("ds_add_f32"); run<0, 4>("ds_add_f32"); run<0, 1>("ds_add_f32");
  run<1, 64>("ds_add_u32"); run<1, 16>("ds_add_u32"); run<1, 4>("ds_add_u32"); run<1, 1>("ds_add_u32");
  run<2, 64>("ds_add_u64"); run<2, 16>("ds_add_u64"); run<2, 4>("ds_add_u64"); run<2, 1>("ds_add_u64");
  return 0;
}
