// Pieces shared by the kernels that stream a weight image through an LDS slot ring while their row tiles stay in registers
// (csrc/mlp_fused.hip, csrc/qkv_fused.hip): counted vector-memory waits and the inline-asm fragment reads.
#pragma once
#include "hfl_common.h"

#include <type_traits>
#include <utility>

#define HFL_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Fragment reads as inline asm: hipcc schedules its own ds_read next to the first use when registers are scarce (here it
// re-used one fragment set and waited lgkmcnt(0) in front of every other MFMA); written out, the reads of step s + 1 are in
// flight behind the MFMAs of step s.  Four 16-B reads: (ahi, alo) + o0 and (ahi, alo) + o1.  `pin` is a fragment of the
// CURRENT step: as a read-write operand it keeps that step's MFMAs behind this statement.  The matching wait names the four
// destinations read-write, so no consumer can be scheduled above it (cdna_hip_programming.md 5.7, form (ii)).
#define HFL_LDS_READ4(f0, f1, f2, f3, ahi, alo, o0, o1, pin)                                                         \
  asm volatile("ds_read_b128 %0, %5 offset:%7\n\tds_read_b128 %1, %6 offset:%7\n\tds_read_b128 %2, %5 offset:%8\n\t" \
               "ds_read_b128 %3, %6 offset:%8"                                                                       \
               : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "+v"(pin)                                                \
               : "v"(ahi), "v"(alo), "n"(o0), "n"(o1))
#define HFL_LDS_READ4_FIRST(f0, f1, f2, f3, ahi, alo, o0, o1)                                                        \
  asm volatile("ds_read_b128 %0, %4 offset:%6\n\tds_read_b128 %1, %5 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t" \
               "ds_read_b128 %3, %5 offset:%7"                                                                       \
               : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)                                                          \
               : "v"(ahi), "v"(alo), "n"(o0), "n"(o1))
#define HFL_LDS_WAIT4(f0, f1, f2, f3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3))
// ... and `after` = the last accumulator of the current step: the wait stays behind that step's MFMAs (no instruction in the
// statement touches it)
#define HFL_LDS_WAIT4_AFTER(f0, f1, f2, f3, after) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(after))

template <int... Is, class F>
__device__ __forceinline__ void hfl_static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

