// Stall attribution for the cooperative trailing update (wg::qr_trail4_coop_pass<2>): ONE workgroup, 16 tiles (two per
// wave), 1600 x (64 + 256) matrix, the pass repeated `reps` times.  Each variant removes one ingredient (results are then
// wrong on purpose; only the time matters):  bit 0: no C global loads / stores, bit 1: no V global loads, bit 2: no LDS
// reads, bit 3: no barriers, bit 4: no MFMAs.  Standalone tool, not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I matrixproductbp.jl_amd/csrc -o tools/_trail_probe.bin tools/probes/trail_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "wg_common.h"
using namespace wgc;
typedef __attribute__((address_space(3))) d2 ld2;
constexpr int VS_LD = 34, VS_STAGE = 64 * VS_LD;

template <int FL>
__device__ __forceinline__ void bar() { if (!(FL & 8)) lds_barrier(); }

template <int FL>
__device__ __forceinline__ void pass(gdbl* Y, long ld, int j0, int nst, int cq0, int cq1, const ldbl* Tq, ldbl* Vs) {
  constexpr int NR = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int sc = tid >> 3, sr = tid & 7;
  const gdbl* vsrc = Y + (long)(j0 + sc) * ld + j0 + 4 * sr;
  const int spanel = sc >> 4, scol = sc & 15;
  auto stage_load = [&](int s) -> d4 {
    if (FL & 2) return d4{1e-3, 2e-3, 3e-3, 4e-3};
    const int sclamp = min(s, nst - 1);
    return *reinterpret_cast<const gd4*>(vsrc + 32 * sclamp);
  };
  auto stage_store = [&](int s, d4 v) {
    ldbl* dst = Vs + (s & 1) * VS_STAGE + sc * VS_LD + 4 * sr;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rp = 32 * s + 4 * sr + e - 16 * spanel;
      double a = v[e];
      a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
      v[e] = (rp >= 0) ? a : 0.0;
    }
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  };
  const int cq[2] = {cq0, cq1};
  d4 w0[4][NR];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) w0[p][q] = d4{0, 0, 0, 0};
  {
    const gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + l15) * ld + j0 + 4 * g;
    d4 vreg = stage_load(0);
    __syncthreads();
    stage_store(0, vreg);
    d4 cc[NR][2];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      if (FL & 1) { cc[q][0] = d4{1, 2, 3, 4}; cc[q][1] = d4{4, 3, 2, 1}; }
      else {
        cc[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q]));
        cc[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 16));
      }
    }
    for (int s = 0; s < nst; s++) {
      vreg = stage_load(s + 1);
      const int sn = min(s + 1, nst - 1);
      d4 cn[NR][2];
#pragma unroll
      for (int q = 0; q < NR; q++) {
        if (FL & 1) { cn[q][0] = cc[q][1]; cn[q][1] = cc[q][0]; }
        else {
          cn[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn));
          cn[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn + 16));
        }
      }
      bar<FL>();
      const ldbl* vb = Vs + (s & 1) * VS_STAGE;
#pragma unroll
      for (int rb = 0; rb < 2; rb++) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const ldbl* vp = vb + (16 * p + l15) * VS_LD + 16 * rb + 4 * g;
          d2 va, vbb;
          if (FL & 4) { va = d2{cc[0][rb][0] + p, 1.0}; vbb = d2{2.0, cc[1][rb][1]}; }
          else { va = *reinterpret_cast<const ld2*>(vp); vbb = *reinterpret_cast<const ld2*>(vp + 2); }
          const double v4[4] = {va[0], va[1], vbb[0], vbb[1]};
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int q = 0; q < NR; q++) {
              if (FL & 16) w0[p][q][e] += v4[e] * cc[q][rb][e];
              else w0[p][q] = mfma(v4[e], cc[q][rb][e], w0[p][q]);
            }
        }
      }
      stage_store(s + 1, vreg);
#pragma unroll
      for (int q = 0; q < NR; q++) { cc[q][0] = cn[q][0]; cc[q][1] = cn[q][1]; }
    }
  }
  d4 w[4][NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[(4 * s + g) + 16 * l15], w0[p][q][s], o);
      w[p][q] = o;
    }
  }
  {
    d4 vreg = stage_load(0);
    __syncthreads();
    stage_store(0, vreg);
    gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + g) * ld + j0 + 2 * l15;
    d2 cc[NR][4];
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (FL & 1) cc[q][r] = d2{1.0 * r, 2.0 * q};
        else cc[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld));
      }
    for (int s = 0; s < nst; s++) {
      vreg = stage_load(s + 1);
      const int sn = min(s + 1, nst - 1);
      d2 cn[NR][4];
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (FL & 1) cn[q][r] = cc[q][3 - r];
          else cn[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld + 32 * sn));
        }
      bar<FL>();
      const ldbl* vb = Vs + (s & 1) * VS_STAGE;
      d4 acc[NR][2];
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int e = 0; e < 2; e++) acc[q][e] = d4{cc[q][0][e], cc[q][1][e], cc[q][2][e], cc[q][3][e]};
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          d2 v;
          if (FL & 4) v = d2{acc[0][0][0] * 1e-9 + p, 1.0 + s2};
          else v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * VS_LD + 2 * l15);
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int q = 0; q < NR; q++) {
              if (FL & 16) acc[q][e][s2] -= w[p][q][s2] * v[e];
              else acc[q][e] = mfma(-w[p][q][s2], v[e], acc[q][e]);
            }
        }
      if (FL & 1) {
        if (acc[0][0][0] == 12345.678 && acc[1][1][2] == 4.25 && acc[0][1][1] == 7.5 && acc[1][0][3] == -1.25) cp[0][s] = acc[0][0][1];
      } else {
#pragma unroll
        for (int q = 0; q < NR; q++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            __builtin_nontemporal_store(d2{acc[q][0][r], acc[q][1][r]}, reinterpret_cast<gd2*>(cp[q] + (long)(4 * r) * ld + 32 * s));
      }
      stage_store(s + 1, vreg);
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) cc[q][r] = cn[q][r];
    }
  }
}


// Variant with a three-deep register ring: the C fragments and the V stage are requested TWO stages ahead.
template <int FL, bool DEEPC, int NR = 2>
__device__ __forceinline__ void pass_deep(gdbl* Y, long ld, int j0, int nst, int cq0, int cq1, const ldbl* Tq, ldbl* Vs) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int sc = tid >> 3, sr = tid & 7;
  const gdbl* vsrc = Y + (long)(j0 + sc) * ld + j0 + 4 * sr;
  const int spanel = sc >> 4, scol = sc & 15;
  auto stage_load = [&](int s) -> d4 {
    const int sclamp = min(s, nst - 1);
    return *reinterpret_cast<const gd4*>(vsrc + 32 * sclamp);
  };
  auto stage_store = [&](int s, d4 v) {
    ldbl* dst = Vs + (s & 1) * VS_STAGE + sc * VS_LD + 4 * sr;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rp = 32 * s + 4 * sr + e - 16 * spanel;
      double a = v[e];
      a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
      v[e] = (rp >= 0) ? a : 0.0;
    }
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  };
  const int cq[2] = {cq0, cq1};
  d4 w0[4][NR];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) w0[p][q] = d4{0, 0, 0, 0};
  {
    const gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + l15) * ld + j0 + 4 * g;
    d4 vr[3];
    d4 c[3][NR][2];
    auto loadc = [&](d4 (&dst)[NR][2], int s) {
      const int sn = min(s, nst - 1);
#pragma unroll
      for (int q = 0; q < NR; q++) {
        dst[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn));
        dst[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn + 16));
      }
    };
    auto body = [&](int s, d4 (&cur)[NR][2], d4 (&nx2)[NR][2], d4& vnext, d4& vnx2) {
      vnx2 = stage_load(s + 2);
      loadc(nx2, s + 2);
      lds_barrier();
      const ldbl* vb = Vs + (s & 1) * VS_STAGE;
#pragma unroll
      for (int rb = 0; rb < 2; rb++) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const ldbl* vp = vb + (16 * p + l15) * VS_LD + 16 * rb + 4 * g;
          const d2 va = *reinterpret_cast<const ld2*>(vp), vbb = *reinterpret_cast<const ld2*>(vp + 2);
          const double v4[4] = {va[0], va[1], vbb[0], vbb[1]};
#pragma unroll
          for (int e = 0; e < 4; e++)
#pragma unroll
            for (int q = 0; q < NR; q++) w0[p][q] = mfma(v4[e], cur[q][rb][e], w0[p][q]);
        }
      }
      stage_store(s + 1, vnext);
    };
    vr[0] = stage_load(0); vr[1] = stage_load(1);
    loadc(c[0], 0); loadc(c[1], 1);
    __syncthreads();
    stage_store(0, vr[0]);
    for (int s = 0; s < nst; s += 3) {
      body(s, c[0], c[2], vr[1], vr[2]);
      if (s + 1 < nst) body(s + 1, c[1], c[0], vr[2], vr[0]);
      if (s + 2 < nst) body(s + 2, c[2], c[1], vr[0], vr[1]);
    }
  }
  d4 w[4][NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[(4 * s + g) + 16 * l15], w0[p][q][s], o);
      w[p][q] = o;
    }
  }
  {
    gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + g) * ld + j0 + 2 * l15;
    d4 vr[3];
    d2 c[3][NR][4];
    auto loadc = [&](d2 (&dst)[NR][4], int s) {
      const int sn = min(s, nst - 1);
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          dst[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld + 32 * sn));
    };
    auto body = [&](int s, d2 (&cur)[NR][4], d2 (&nx2)[NR][4], d4& vnext, d4& vnx2) {
      vnx2 = stage_load(s + 2);
      loadc(nx2, s + 2);
      lds_barrier();
      const ldbl* vb = Vs + (s & 1) * VS_STAGE;
      d4 acc[NR][2];
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int e = 0; e < 2; e++) acc[q][e] = d4{cur[q][0][e], cur[q][1][e], cur[q][2][e], cur[q][3][e]};
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          const d2 v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * VS_LD + 2 * l15);
#pragma unroll
          for (int e = 0; e < 2; e++)
#pragma unroll
            for (int q = 0; q < NR; q++) acc[q][e] = mfma(-w[p][q][s2], v[e], acc[q][e]);
        }
#pragma unroll
      for (int q = 0; q < NR; q++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          __builtin_nontemporal_store(d2{acc[q][0][r], acc[q][1][r]}, reinterpret_cast<gd2*>(cp[q] + (long)(4 * r) * ld + 32 * s));
      stage_store(s + 1, vnext);
    };
    vr[0] = stage_load(0); vr[1] = stage_load(1);
    loadc(c[0], 0); if (DEEPC) loadc(c[1], 1);
    __syncthreads();
    stage_store(0, vr[0]);
    if (DEEPC) {
      for (int s = 0; s < nst; s += 3) {
        body(s, c[0], c[2], vr[1], vr[2]);
        if (s + 1 < nst) body(s + 1, c[1], c[0], vr[2], vr[0]);
        if (s + 2 < nst) body(s + 2, c[2], c[1], vr[0], vr[1]);
      }
    } else {
      // one stage ahead, unrolled by two so that no register copies (and no wait for them) close a stage
      auto body1 = [&](int s, d2 (&cur)[NR][4], d2 (&nx)[NR][4], d4& vcur, d4& vnx) {
        vnx = stage_load(s + 2);
        loadc(nx, s + 1);
        lds_barrier();
        const ldbl* vb = Vs + (s & 1) * VS_STAGE;
        d4 acc[NR][2];
#pragma unroll
        for (int q = 0; q < NR; q++)
#pragma unroll
          for (int e = 0; e < 2; e++) acc[q][e] = d4{cur[q][0][e], cur[q][1][e], cur[q][2][e], cur[q][3][e]};
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
          for (int s2 = 0; s2 < 4; s2++) {
            const d2 v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * VS_LD + 2 * l15);
#pragma unroll
            for (int e = 0; e < 2; e++)
#pragma unroll
              for (int q = 0; q < NR; q++) acc[q][e] = mfma(-w[p][q][s2], v[e], acc[q][e]);
          }
#pragma unroll
        for (int q = 0; q < NR; q++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            __builtin_nontemporal_store(d2{acc[q][0][r], acc[q][1][r]}, reinterpret_cast<gd2*>(cp[q] + (long)(4 * r) * ld + 32 * s));
        stage_store(s + 1, vcur);
      };
      for (int s = 0; s < nst; s += 2) {
        body1(s, c[0], c[1], vr[1], vr[0]);
        if (s + 1 < nst) body1(s + 1, c[1], c[0], vr[0], vr[1]);
      }
    }
  }
}

// Variant with 64-row stages of V in LDS: one barrier per 64 rows instead of per 32 (C still prefetched 32 rows ahead).
constexpr int VS_LD2 = 66, VS_STAGE2 = 64 * VS_LD2;
template <int FL>
__device__ __forceinline__ void pass_ms(gdbl* Y, long ld, int j0, int nst, int cq0, int cq1, const ldbl* Tq, ldbl* Vs) {
  constexpr int NR = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int sc = tid >> 3, sr = tid & 7;
  const int nms = (nst + 1) >> 1;
  const gdbl* vsrc = Y + (long)(j0 + sc) * ld + j0 + 4 * sr;
  const int spanel = sc >> 4, scol = sc & 15;
  auto stage_load = [&](int m, int h) -> d4 {
    const int s = min(2 * min(m, nms - 1) + h, nst - 1);
    return *reinterpret_cast<const gd4*>(vsrc + 32 * s);
  };
  auto stage_store = [&](int m, int h, d4 v) {
    ldbl* dst = Vs + (m & 1) * VS_STAGE2 + sc * VS_LD2 + 32 * h + 4 * sr;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rp = 64 * m + 32 * h + 4 * sr + e - 16 * spanel;
      double a = v[e];
      a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
      v[e] = (rp >= 0 && 2 * m + h < nst) ? a : 0.0;
    }
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  };
  const int cq[2] = {cq0, cq1};
  d4 w0[4][NR];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) w0[p][q] = d4{0, 0, 0, 0};
  {
    const gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + l15) * ld + j0 + 4 * g;
    d4 v0 = stage_load(0, 0), v1 = stage_load(0, 1);
    __syncthreads();
    stage_store(0, 0, v0); stage_store(0, 1, v1);
    d4 cc[NR][2];
#pragma unroll
    for (int q = 0; q < NR; q++) {
      cc[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q]));
      cc[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 16));
    }
    for (int m = 0; m < nms; m++) {
      v0 = stage_load(m + 1, 0); v1 = stage_load(m + 1, 1);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int s = 2 * m + h;
        const int sn = min(s + 1, nst - 1);
        d4 cn[NR][2];
#pragma unroll
        for (int q = 0; q < NR; q++) {
          cn[q][0] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn));
          cn[q][1] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 32 * sn + 16));
        }
        if (h == 0) lds_barrier();
        const ldbl* vb = Vs + (m & 1) * VS_STAGE2 + 32 * h;
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const ldbl* vp = vb + (16 * p + l15) * VS_LD2 + 16 * rb + 4 * g;
            const d2 va = *reinterpret_cast<const ld2*>(vp), vbb = *reinterpret_cast<const ld2*>(vp + 2);
            const double v4[4] = {va[0], va[1], vbb[0], vbb[1]};
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
              for (int q = 0; q < NR; q++) w0[p][q] = mfma(v4[e], cc[q][rb][e], w0[p][q]);
          }
        }
#pragma unroll
        for (int q = 0; q < NR; q++) { cc[q][0] = cn[q][0]; cc[q][1] = cn[q][1]; }
      }
      stage_store(m + 1, 0, v0); stage_store(m + 1, 1, v1);
    }
  }
  d4 w[4][NR];
#pragma unroll
  for (int q = 0; q < NR; q++) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[(4 * s + g) + 16 * l15], w0[p][q][s], o);
      w[p][q] = o;
    }
  }
  {
    d4 v0 = stage_load(0, 0), v1 = stage_load(0, 1);
    __syncthreads();
    stage_store(0, 0, v0); stage_store(0, 1, v1);
    gdbl* cp[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + g) * ld + j0 + 2 * l15;
    d2 cc[NR][4];
#pragma unroll
    for (int q = 0; q < NR; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) cc[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld));
    for (int m = 0; m < nms; m++) {
      v0 = stage_load(m + 1, 0); v1 = stage_load(m + 1, 1);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int s = 2 * m + h;
        const int sn = min(s + 1, nst - 1);
        d2 cn[NR][4];
#pragma unroll
        for (int q = 0; q < NR; q++)
#pragma unroll
          for (int r = 0; r < 4; r++) cn[q][r] = __builtin_nontemporal_load(reinterpret_cast<const gd2*>(cp[q] + (long)(4 * r) * ld + 32 * sn));
        if (h == 0) lds_barrier();
        if (s < nst) {
          const ldbl* vb = Vs + (m & 1) * VS_STAGE2 + 32 * h;
          d4 acc[NR][2];
#pragma unroll
          for (int q = 0; q < NR; q++)
#pragma unroll
            for (int e = 0; e < 2; e++) acc[q][e] = d4{cc[q][0][e], cc[q][1][e], cc[q][2][e], cc[q][3][e]};
#pragma unroll
          for (int p = 0; p < 4; p++)
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
              const d2 v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * VS_LD2 + 2 * l15);
#pragma unroll
              for (int e = 0; e < 2; e++)
#pragma unroll
                for (int q = 0; q < NR; q++) acc[q][e] = mfma(-w[p][q][s2], v[e], acc[q][e]);
            }
#pragma unroll
          for (int q = 0; q < NR; q++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              __builtin_nontemporal_store(d2{acc[q][0][r], acc[q][1][r]}, reinterpret_cast<gd2*>(cp[q] + (long)(4 * r) * ld + 32 * s));
        }
#pragma unroll
        for (int q = 0; q < NR; q++)
#pragma unroll
          for (int r = 0; r < 4; r++) cc[q][r] = cn[q][r];
      }
      stage_store(m + 1, 0, v0); stage_store(m + 1, 1, v1);
    }
  }
}

// Fused pass: phase C of the PREVIOUS block (columns jp .. jp+63, W read from `Wg`) and phase A of the CURRENT block
// (columns j0 = jp + 64 ..) in one sweep over the rows: C is read once and written once per block instead of read twice
// and written once.  Phase C in the non-transposed form C -= V W: the accumulator fragment (reg e of lane (g, l15) = row
// 16 rb + 4 g + e, column l15) is the d4 a lane loads from C, and after the update it is exactly the B operand of phase A.
// The previous block's reflectors are staged row-major with permuted panel columns (VT: [row][16 p + 4 (c % 4) + c / 4],
// stride 66) so that the four A operands of a (row block, panel) are one 32-byte LDS read.
constexpr int VT_LD = 66, VT_STAGE = 32 * VT_LD;
template <int FL>
__device__ __forceinline__ void pass_fused(gdbl* Y, long ld, int jp, int nst, int cq0, int cq1, const gdbl* Wg, const ldbl* Tq,
                                           ldbl* Vs, ldbl* Vt) {
  constexpr int NR = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = lane >> 4, l15 = lane & 15;
  const int j0 = jp + 64;
  const int sc = tid >> 3, sr = tid & 7;
  const gdbl* vsrcA = Y + (long)(j0 + sc) * ld + jp + 4 * sr;     // current block, rows counted from jp
  const gdbl* vsrcT = Y + (long)(jp + sc) * ld + jp + 4 * sr;     // previous block
  const int spanel = sc >> 4, scol = sc & 15;
  const int pcp = 16 * spanel + 4 * (scol & 3) + (scol >> 2);
  auto stage_load = [&](const gdbl* src, int s) -> d4 { return *reinterpret_cast<const gd4*>(src + 32 * min(s, nst - 1)); };
  auto head = [&](d4 v, int s, int off) -> d4 {             // unit-lower-trapezoidal form, zero above the panel
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rp = 32 * s + 4 * sr + e - off - 16 * spanel;
      double a = v[e];
      a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
      v[e] = (rp >= 0) ? a : 0.0;
    }
    return v;
  };
  auto store_A = [&](int s, d4 v) {
    v = head(v, s, 64);
    ldbl* dst = Vs + (s & 1) * VS_STAGE + sc * VS_LD + 4 * sr;
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  };
  auto store_T = [&](int s, d4 v) {
    v = head(v, s, 0);
    ldbl* dst = Vt + (s & 1) * VT_STAGE + (4 * sr) * VT_LD + pcp;
#pragma unroll
    for (int e = 0; e < 4; e++) dst[e * VT_LD] = -v[e];      // negated: the update subtracts
  };
  const int cq[2] = {cq0, cq1};
  d4 w[4][NR], w0[4][NR];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NR; q++) {
      w0[p][q] = d4{0, 0, 0, 0};
      w[p][q] = *reinterpret_cast<const gd4*>(Wg + ((long)(cq[q] >> 4) * 4 + p) * 256 + 4 * lane);
    }
  gdbl* cp[NR];
#pragma unroll
  for (int q = 0; q < NR; q++) cp[q] = Y + (long)(cq[q] + l15) * ld + jp + 4 * g;
  const int mrow = 4 * (l15 & 3) + (l15 >> 2);              // matrix row (in its 16-row block) behind A-operand row l15
  d4 va = stage_load(vsrcA, 0), vt = stage_load(vsrcT, 0);
  __syncthreads();
  store_A(0, va); store_T(0, vt);
  d4 cc[NR];                                                // 16 rows x 16 columns per tile, 16 rows ahead
#pragma unroll
  for (int q = 0; q < NR; q++) cc[q] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q]));
  for (int s = 0; s < nst; s++) {
    va = stage_load(vsrcA, s + 1); vt = stage_load(vsrcT, s + 1);
#pragma unroll 1
    for (int rb = 0; rb < 2; rb++) {
      const int hn = min(2 * s + rb + 1, 2 * nst - 1);
      d4 cn[NR];
#pragma unroll
      for (int q = 0; q < NR; q++) cn[q] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp[q] + 16 * hn));
      if (rb == 0) lds_barrier();
      const ldbl* vtb = Vt + (s & 1) * VT_STAGE + (16 * rb + mrow) * VT_LD + 4 * g;
      const ldbl* vab = Vs + (s & 1) * VS_STAGE + l15 * VS_LD + 16 * rb + 4 * g;
      // phase C of the previous block on this 16-row block
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const d2 t0 = *reinterpret_cast<const ld2*>(vtb + 16 * p), t1 = *reinterpret_cast<const ld2*>(vtb + 16 * p + 2);
        const double v4[4] = {t0[0], t0[1], t1[0], t1[1]};
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int q = 0; q < NR; q++) cc[q] = mfma(v4[r], w[p][q][r], cc[q]);
      }
#pragma unroll
      for (int q = 0; q < NR; q++) __builtin_nontemporal_store(cc[q], reinterpret_cast<gd4*>(cp[q] + 16 * (2 * s + rb)));
      // phase A of the current block on the updated fragment
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const ldbl* vp = vab + 16 * p * VS_LD;
        const d2 a0 = *reinterpret_cast<const ld2*>(vp), a1 = *reinterpret_cast<const ld2*>(vp + 2);
        const double v4[4] = {a0[0], a0[1], a1[0], a1[1]};
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
          for (int q = 0; q < NR; q++) w0[p][q] = mfma(v4[e], cc[q][e], w0[p][q]);
      }
#pragma unroll
      for (int q = 0; q < NR; q++) cc[q] = cn[q];
    }
    store_A(s + 1, va); store_T(s + 1, vt);
  }
  // phase B stand-in + the W of this block to scratch
#pragma unroll
  for (int q = 0; q < NR; q++)
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[(4 * s + g) + 16 * l15], w0[p][q][s], o);
      *reinterpret_cast<gd4*>(const_cast<gdbl*>(Wg) + ((long)(cq[q] >> 4) * 4 + p) * 256 + 4 * lane) = o;
    }
}

template <int FL>
__global__ void __launch_bounds__(512) k(double* Y, int ld, int rows32, int reps, double* W) {
  __shared__ __attribute__((aligned(16))) double Vs[2 * VS_STAGE2];
  __shared__ double T[256];
  __shared__ __attribute__((aligned(16))) double Vt[2 * VT_STAGE];
  if (threadIdx.x < 256) T[threadIdx.x] = 0.0;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  gdbl* Yg = (gdbl*)Y + (long)blockIdx.x * ld * 400;
  for (int r = 0; r < reps; r++)
    if (FL == 512) {
      pass_deep<FL, true, 1>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave, 64 + 32 * wave, (const ldbl*)T, (ldbl*)Vs);
      pass_deep<FL, true, 1>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave + 16, 64 + 32 * wave + 16, (const ldbl*)T, (ldbl*)Vs);
    } else if (FL == 256) pass_fused<FL>(Yg, ld, 0, rows32 >> 5, 128 + 32 * wave, 128 + 32 * wave + 16, (const gdbl*)W + (long)blockIdx.x * 32 * 1024, (const ldbl*)T, (ldbl*)Vs, (ldbl*)Vt);
    else if (FL == 128) pass_ms<FL>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave, 64 + 32 * wave + 16, (const ldbl*)T, (ldbl*)Vs);
    else if (FL == 32) pass_deep<FL, true>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave, 64 + 32 * wave + 16, (const ldbl*)T, (ldbl*)Vs);
    else if (FL == 64) pass_deep<FL, false>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave, 64 + 32 * wave + 16, (const ldbl*)T, (ldbl*)Vs);
    else pass<FL>(Yg, ld, 0, rows32 >> 5, 64 + 32 * wave, 64 + 32 * wave + 16, (const ldbl*)T, (ldbl*)Vs);
}

static double* g_W = nullptr;
template <int FL>
void run(double* d, int ld, int rows32, int blocks, const char* what) {
  const int reps = 20;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<FL>, dim3(blocks), dim3(512), 0, 0, d, ld, rows32, 1, g_W);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); hipLaunchKernelGGL(k<FL>, dim3(blocks), dim3(512), 0, 0, d, ld, rows32, reps, g_W); (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double ideal = (double)(rows32 >> 5) * 4 * 64 * 64 / 2.4e6;     // ms per pass: 4 tile-stages per SIMD, 64 MFMAs of 64 cycles each
  printf("blocks %3d  %-44s %7.3f ms per pass  (MFMA-bound %.3f ms, %.0f %%)\n", blocks, what, ms / reps, ideal, 100.0 * ideal / (ms / reps));
}

int main() {
  const int rows32 = 1600, ld = 1600, cols = 400;
  double* d; (void)hipMalloc(&d, sizeof(double) * ld * cols * 256);
  std::vector<double> h((size_t)ld * cols);
  for (size_t i = 0; i < h.size(); i++) h[i] = (double)((i * 2654435761u) % 2001) / 1000.0 - 1.0;
  for (int b = 0; b < 256; b++) (void)hipMemcpy(d + (size_t)b * ld * cols, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  (void)hipMalloc(&g_W, sizeof(double) * 32 * 1024 * 256); (void)hipMemset(g_W, 0, sizeof(double) * 32 * 1024 * 256);
  for (int blocks : {1, 256}) {
    run<0>(d, ld, rows32, blocks, "as shipped");
    run<128>(d, ld, rows32, blocks, "64-row stages (half the barriers)");
    run<512>(d, ld, rows32, blocks, "one tile per wave, two passes, C two stages ahead");
    run<256>(d, ld, rows32, blocks, "FUSED: C of block k-1 + A of block k (= 2 passes of MFMAs)");
    run<32>(d, ld, rows32, blocks, "two stages of prefetch");
    run<64>(d, ld, rows32, blocks, "two stages ahead in phase A only");
    run<1>(d, ld, rows32, blocks, "no C loads/stores");
    run<2>(d, ld, rows32, blocks, "no V global loads");
    run<3>(d, ld, rows32, blocks, "no global traffic at all");
    run<4>(d, ld, rows32, blocks, "no LDS reads");
    run<8>(d, ld, rows32, blocks, "no barriers");
    run<7>(d, ld, rows32, blocks, "no global, no LDS reads");
    run<15>(d, ld, rows32, blocks, "no global, no LDS reads, no barriers");
    run<16>(d, ld, rows32, blocks, "no MFMA (scalar stand-in)");
  }
  return 0;
}
