// Does hipExtStreamCreateWithCUMask partition the chip on this box, and how do mask bits map to (XCD, CU)?
// Launches a census kernel (every workgroup records its XCC id and hardware id) on streams with different masks, and times
// a bandwidth-bound and an ALU-bound kernel alone and concurrently on two disjoint masks.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/bin/cu_mask_census tools/micro/cu_mask_census.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
#include <map>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void census(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // stay resident a little so that the grid spreads over every CU the stream may use
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = xcc;
    out[2 * blockIdx.x + 1] = hw;
  }
}

__global__ void spin_kernel(float* out, int iters) {      // ALU-bound: time ~ grid / CUs
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  if (a == 123.f) out[0] = a;
}

static int run_census(hipStream_t s, uint32_t* dbuf, int nwg, const char* what) {
  std::vector<uint32_t> h(2 * nwg);
  census<<<nwg, 256, 0, s>>>(dbuf, 20000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), dbuf, h.size() * 4, hipMemcpyDeviceToHost));
  std::map<uint32_t, std::set<uint32_t>> per_xcc;
  for (int i = 0; i < nwg; ++i) {
    const uint32_t xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  printf("%-40s:", what);
  for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  -> %d CUs\n", total);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  const int nwg = 8192;
  uint32_t* dbuf;
  CK(hipMalloc(&dbuf, 2 * nwg * 4));
  hipStream_t s0;
  CK(hipStreamCreate(&s0));
  if (run_census(s0, dbuf, nwg, "plain stream")) return 1;
  struct M { const char* name; std::vector<uint32_t> mask; };
  std::vector<M> masks;
  masks.push_back({"bits 0..63", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0..191", {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0}});
  masks.push_back({"bits 192..255", {0, 0, 0, 0, 0, 0, 0xFFFFFFFFu, 0xFFFFFFFFu}});
  masks.push_back({"every 4th bit (64 bits)", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}});
  masks.push_back({"bits 0..7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}});
  std::vector<hipStream_t> streams;
  for (auto& m : masks) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.mask.size(), m.mask.data());
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s) failed: %s\n", m.name, hipGetErrorString(e)); return 2; }
    streams.push_back(s);
    if (run_census(s, dbuf, nwg, m.name)) return 1;
  }
  // concurrency: the same ALU-bound kernel (grid 4096 x 256) alone on the plain stream, alone on 192 / 64 CUs, and both at once
  float* dout;
  CK(hipMalloc(&dout, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_one = [&](hipStream_t s, int grid) -> float {
    spin_kernel<<<grid, 256, 0, s>>>(dout, 20000);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    spin_kernel<<<grid, 256, 0, s>>>(dout, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
  };
  printf("ALU kernel, grid 8192: plain %.1f us | 192-CU mask %.1f us | 64-CU mask %.1f us\n", time_one(s0, 8192),
         time_one(streams[1], 8192), time_one(streams[2], 8192));
  // both masks at once: wall time of (6144 WGs on 192 CUs) || (2048 WGs on 64 CUs) -- should be ~ the plain stream's 8192
  hipDeviceSynchronize();
  hipEvent_t a0, a1, b1;
  hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b1);
  hipEventRecord(a0, s0);
  hipStreamWaitEvent(streams[1], a0, 0);
  hipStreamWaitEvent(streams[2], a0, 0);
  spin_kernel<<<6144, 256, 0, streams[1]>>>(dout, 20000);
  spin_kernel<<<2048, 256, 0, streams[2]>>>(dout, 20000);
  hipEventRecord(a1, streams[1]);
  hipEventRecord(b1, streams[2]);
  hipDeviceSynchronize();
  float t1 = 0, t2 = 0;
  hipEventElapsedTime(&t1, a0, a1);
  hipEventElapsedTime(&t2, a0, b1);
  printf("concurrent on disjoint masks: 6144 WGs / 192 CUs done at %.1f us, 2048 WGs / 64 CUs done at %.1f us\n", t1 * 1e3f, t2 * 1e3f);
  return 0;
}
