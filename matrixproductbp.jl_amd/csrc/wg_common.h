// Definitions shared by every workgroup-size variant of the building blocks (wg_blocks.h / engine.h are
// included once per variant, inside the variant's namespace, with WG_THREADS / WG_WAVES defined).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Explicit address spaces.  Pointers that cross a (noinline) function boundary as plain `double*` are "generic":
// the compiler then emits FLAT loads/stores, which (a) reach LDS through the slow flat path and (b) count on both
// vmcnt and lgkmcnt, so every use of a loaded value waits for ALL outstanding memory operations - no software
// pipelining survives.  Typed pointers give global_load / ds_read with counted waits.
typedef __attribute__((address_space(1))) double gdbl;   // HBM
typedef __attribute__((address_space(3))) double ldbl;   // LDS
typedef __attribute__((address_space(1))) d2 gd2;
typedef __attribute__((address_space(1))) d4 gd4;

namespace wgc {

// optional phase profile: lane 0 of the workgroup adds elapsed wall-clock ticks (100 MHz) per phase
struct Prof { unsigned long long t[24]; };
__device__ __forceinline__ void prof_mark(Prof* pr, unsigned long long& last, int phase) {
  if (pr && threadIdx.x == 0) {
    unsigned long long now = wall_clock64();
    atomicAdd(&pr->t[phase], now - last);
    last = now;
  }
}
enum { PH_STAGE = 0, PH_Y1, PH_Y2, PH_QR1_PANEL, PH_QR1_TRAIL, PH_LF, PH_N, PH_MT, PH_QR2_PANEL, PH_QR2_TRAIL,
       PH_JAC, PH_TRUNC, PH_CARRY, PH_NORM, PH_COUNT };

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// c - a b: on gfx940+ the BLGP field of the f64 MFMA negates A / B / C (bit 0: A).  Negating the operand instead is a VALU
// instruction per use inside the MFMA loops, and on gfx950 every VALU instruction takes its cycles from the fp64 MFMAs of
// the same SIMD (tools/probes/mfma_operand_probe.hip).
__device__ __forceinline__ d4 mfma_na(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
}

// Workgroup barrier that orders LDS traffic only: this wave's LDS operations are complete (lgkmcnt(0)), outstanding
// global loads stay in flight across it (a __syncthreads() drains vmcnt as well and serialises every prefetch that spans
// a barrier - cdna_hip_programming.md section 5, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xc07f);      // vmcnt = 63, expcnt = 7, lgkmcnt = 0
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

}  // namespace wgc
