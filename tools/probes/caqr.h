// Communication-avoiding R-only QR for ONE 512-thread workgroup (gfx950): the gauge-sweep factorisation of the
// compress engine (`compress!` inside `op`, reference src/recursive_bp_factor.jl:127; the right sweep of
// TensorTrains' orthogonalize_right! only changes the gauge, so only R = Lf^T is kept - engine.h).
//
// Why a second algorithm.  The blocked Householder QR of wg_blocks.h (`wg::qr_r`) applies a 64-column block reflector
// over the full height of Y (1600 rows at configs[1]): V (819 KB) does not fit on the CU, so every block streams V
// twice and C three times, and the 38 panel-sized round trips inside a block go through HBM too - 88 MB of traffic
// for a 5 MB matrix, which is what bounds the engine.  Here the rows are cut into CHUNKS of
// 256 and every block (64 columns) is reduced chunk by chunk with a FLAT TREE (CAQR / tiled-QR "TS" kernels):
//
//   chunk 0 of block k   rows [64k, 64k+256): ordinary Householder panel; its first 64 rows become R_kk / R_k,: (the "top")
//   chunk i >= 1         QR of [R_kk ; P_i] with R_kk upper triangular: reflectors [e_j ; v_j], v_j dense in the chunk
//
// so a chunk's reflectors V_i (256 x 64) are LOCAL to its rows: they sit in LDS (128 KB, XOR-swizzled, both MFMA
// operand shapes conflict free) for the whole trailing update of the chunk, and every trailing tile (256 x 16) is
// read ONCE into registers, updated (W0 = Top + V^T C; W = T^T-recurrence; Top -= W; C -= V W) and written ONCE.
// V never goes to memory.  Traffic per 1600 x 400 QR: ~39 MB instead of 88 MB.
//
// The price of chunking is 6x more sequential Householder column steps.  They are hidden by WAVE SPECIALISATION:
//   panel group  (waves 0-3): factor the NEXT chunk-panel entirely in registers, VALU only (rows across lanes in the MFMA
//                             B-operand layout, 16 rows per lane, so a column's dot products are in-lane sums + 2 butterfly levels)
//   update group (waves 4-7): apply the CURRENT chunk's reflectors (LDS) to the trailing tiles with MFMA, one tile per wave
// The two groups only meet at the hand-over of the LDS image (two LDS sequence counters); inside a group the waves meet at
// an LDS arrival counter.  No workgroup barrier inside the factorisation.
//
// Layouts (lane = 16 g + c):
//   B/D layout of a 16-row group: lane (g, c) holds rows 4g + e (e = 0..3, one d4 = 32 contiguous bytes) of column c.
//     As the B operand of k-step e the MFMA k index g stands for row 4g + e; as the accumulator D, register e of lane
//     (g, c) is MFMA row g + 4e, which therefore also stands for row 4g + e: sigma(i) = 4 (i & 3) + (i >> 2).
//   V image in LDS: V[row * 64 + (col ^ swz(row & 15))], swz(m) = (m & 3) | (m & 8) | ((m & 4) << 2).
//   operand images of the sixteen-column triangular factors: T_p (p = 0..3) and the cross Grams S_pr (r < p):
//     Timg[p][s][lane (g, c)] = T_p[4g + s][sigma(c)],  Simg[p,r][s][lane] = -S_pr[sigma(c)][4g + s]
#pragma once
#include "wg_common.h"

namespace caqr {
using namespace wgc;

#ifdef CAQR_PROF
__device__ unsigned long long* g_prof = nullptr;     // [64]: accumulated s_memtime ticks per phase (probe builds only)
#define CAQR_T0() unsigned long long t0_ = __builtin_readcyclecounter()
#define CAQR_ACC(slot) do { if ((threadIdx.x & 63) == 0 && g_prof) { unsigned long long t1_ = __builtin_readcyclecounter(); \
    atomicAdd(&g_prof[(slot) + 16 * (threadIdx.x >> 6 >= 4)], t1_ - t0_); t0_ = t1_; } } while (0)
#else
#define CAQR_T0() do {} while (0)
#define CAQR_ACC(slot) do {} while (0)
#endif

typedef __attribute__((address_space(3))) int lint;

constexpr int L_V = 0;                    // [256 * 64]
constexpr int L_OPS = 16384;              // [10][256]
constexpr int L_PART = L_OPS + 2560;      // [2][4][64]
constexpr int L_ROW = L_PART + 512;       // [2][64]
constexpr int L_TAU = L_ROW + 128;        // [64]
constexpr int L_SYNC = L_TAU + 64;        // ints: 0 panel-group arrivals, 1 update-group arrivals, 2 images published, 3 update-group chunk arrivals
constexpr int L_TOTAL = L_SYNC + 8;       // 19656 doubles = 157,248 bytes

__device__ __forceinline__ int swz(int m) { return (m & 3) | (m & 8) | ((m & 4) << 2); }
__device__ __forceinline__ int sig(int i) { return 4 * (i & 3) + (i >> 2); }

// ---------------------------------------------------------------------------------------------- group synchronisation
// Arrival counter in LDS shared by the four waves of a group: every wave adds one, lane 0 polls.  LDS operations of a
// wave execute in order, so everything the wave wrote to LDS before its add is visible to whoever sees the count.
__device__ __forceinline__ void group_arrive_wait(lint* cnt, int& target) {
  target += 4;
  if ((threadIdx.x & 63) == 0) {
    __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_at_least(lint* cnt, int value) {
  if ((threadIdx.x & 63) == 0) {
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(2);
  }
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------- plan (same on both sides)
struct Plan {
  int rows, cols, nblk;
  __device__ __forceinline__ int bw(int k) const { return min(64, cols - 64 * k); }
  __device__ __forceinline__ int nch(int k) const { return (rows - 64 * k + 255) >> 8; }
  __device__ __forceinline__ int ntl(int k) const { const int c = cols - 64 * k - 64; return c > 0 ? (c + 15) >> 4 : 0; }
};

// ---------------------------------------------------------------------------------------------- update group
// One trailing tile (NRB 16-row groups x 16 columns at col0) of the chunk at row0 against the chunk's LDS image.
// TOP: the chunk's reflectors are [e_j ; v_j]; the pivot rows of the block start at top0.
template <int NRB, bool TOP>
__device__ __forceinline__ void update_tile(gdbl* Y, long ld, int row0, int col0, int top0, int np, const ldbl* V,
                                            const ldbl* OPS) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  gdbl* cp = Y + (long)(col0 + c) * ld + row0 + 4 * g;
  d4 C[NRB];
#pragma unroll
  for (int rb = 0; rb < NRB; rb++) C[rb] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp + 16 * rb));
  // LDS offsets: p even / odd variants absorb the bit-4 part of the swizzle (see header)
  int aE[4], aO[4], cE[4], cO[4];
  {
    const int sc = sig(c);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int m = 4 * g + e, z = swz(m);
      const int lo = m * 64 + (sc ^ (z & 15)), hi = 16 * (z >> 4);
      aE[e] = lo + hi; aO[e] = lo - hi;
    }
    const int z = swz(sc);
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int lo = sc * 64 + ((4 * g + s) ^ (z & 15)), hi = 16 * (z >> 4);
      cE[s] = lo + hi; cO[s] = lo - hi;
    }
  }
  // ------------------------------------------------ phase A: W0_p = V_p^T C
  // The LDS operands are fetched one row group ahead by hand and the schedule is pinned per group: left alone, the
  // scheduler hoists dozens of ds_reads above the MFMA chain and spills the C tile (204 spills at NRB = 16).
  d4 w[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    d4 acc = d4{0, 0, 0, 0};
    if (p < np) {
      // two base pointers per operand: the row-group offset (8 KB per group) then fits the 16-bit ds_read immediate;
      // with one base the compiler materialises an address register per (group, e) - 204 of them, spilled
      const ldbl* Vlo[4];
      const ldbl* Vhi[4];
#pragma unroll
      for (int e = 0; e < 4; e++) { Vlo[e] = V + ((p & 1) ? aO[e] : aE[e]); Vhi[e] = Vlo[e] + 8192; }      // the same 16 registers for every p
      double a[4], an[4];
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] = Vlo[e][16 * p];
#pragma unroll
      for (int rb = 0; rb < NRB; rb++) {
        if (rb + 1 < NRB) {
#pragma unroll
          for (int e = 0; e < 4; e++) an[e] = (rb + 1 < 8) ? Vlo[e][16 * p + 1024 * (rb + 1)] : Vhi[e][16 * p + 1024 * (rb + 1 - 8)];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) acc = mfma(a[e], C[rb][e], acc);
#pragma unroll
        for (int e = 0; e < 4; e++) a[e] = an[e];
#ifndef CAQR_NO_SCHED_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    w[p] = acc;
  }
  // ------------------------------------------------ phase B: W_p = T_p^T (W0_p (+ Top_p) - sum_{r<p} S_pr W_r)
  gdbl* tp = Y + (long)(col0 + c) * ld + top0 + 4 * g;
#pragma unroll
  for (int p = 0; p < 4; p++) {
    if (p < np) {
      d4 t = w[p];
      d4 topv = d4{0, 0, 0, 0};
      if (TOP) { topv = *reinterpret_cast<const gd4*>(tp + 16 * p); t += topv; }
#pragma unroll
      for (int r = 0; r < p; r++) {
        const ldbl* S = OPS + (4 + p * (p - 1) / 2 + r) * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; s++) t = mfma(S[64 * s], w[r][s], t);
      }
      d4 o = d4{0, 0, 0, 0};
      const ldbl* T = OPS + p * 256 + lane;
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(T[64 * s], t[s], o);
      w[p] = o;
      if (TOP) *reinterpret_cast<gd4*>(tp + 16 * p) = topv - o;
    }
  }
#pragma unroll
  for (int p = 0; p < 4; p++) w[p] = -w[p];
  // ------------------------------------------------ phase C: C -= sum_p V_p W_p
#pragma unroll
  for (int rb = 0; rb < NRB; rb++) {
    d4 acc = C[rb];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (p < np) {
        double a[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const ldbl* vb = V + ((p & 1) ? cO[s] : cE[s]) + ((rb < 8) ? 0 : 8192);
          a[s] = vb[16 * p + 1024 * (rb & 7)];
        }
#pragma unroll
        for (int s = 0; s < 4; s++) acc = mfma(a[s], w[p][s], acc);
      }
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<gd4*>(cp + 16 * rb));
#ifndef CAQR_NO_SCHED_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

// T of a sixteen-reflector panel from its Gram matrix (dlarft): row i of T depends only on its own earlier entries.
//   T(i,j) = -tau_j sum_{i2=i}^{j-1} T(i,i2) G(i2,j)  (i < j),  T(j,j) = tau_j.     G, Tout: plain 16 x 16, [i + 16 j]
// Executed by lanes 0..15 of one wave; G is read from LDS, T written over it afterwards by the caller.
__device__ __forceinline__ void t_from_gram(const ldbl* G, const ldbl* tau, double (&trow)[16], int i) {
#pragma unroll
  for (int j = 0; j < 16; j++) {
    double gcol[16];
#pragma unroll
    for (int i2 = 0; i2 < 16; i2++) gcol[i2] = G[i2 + 16 * j];
    const double tj = tau[j];
    double sacc = 0.0;
#pragma unroll
    for (int i2 = 0; i2 < j; i2++) sacc += (i2 >= i) ? trow[i2] * gcol[i2] : 0.0;
    trow[j] = (j == i) ? tj : ((j > i) ? -tj * sacc : 0.0);
  }
}

// Gram blocks of the LDS image -> T_p and S_pr operand images.  Called by the four update waves (u = 0..3) together.
//   blocks: 0..3 = (p,p); 4 + p(p-1)/2 + r = (p,r).  G_pr[i][j] = sum_rows V[row][16p+i] V[row][16r+j] (+ delta for TOP, p == r)
template <bool TOP>
__device__ __forceinline__ void build_images(int u, int nrb, int np, const ldbl* V, ldbl* OPS, const ldbl* tauL, lint* ubar,
                                             int& utarget) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int nblocks = np * (np + 1) / 2;
  // block list in the order (0,0),(1,1),(2,2),(3,3),(1,0),(2,0),(2,1),(3,0),(3,1),(3,2): wave u takes blocks u, u+4, u+8
  for (int b = u; b < 10; b += 4) {
    int p, r;
    if (b < 4) { p = b; r = b; }
    else { const int q = b - 4; p = (q < 1) ? 1 : ((q < 3) ? 2 : 3); r = q - p * (p - 1) / 2; }
    if (p >= np) continue;
    d4 acc = d4{0, 0, 0, 0};
    for (int rb = 0; rb < nrb; rb++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int m = 4 * g + e, z = swz(m);
        const double a = V[(16 * rb + m) * 64 + ((16 * p + c) ^ z)];
        const double bb = V[(16 * rb + m) * 64 + ((16 * r + c) ^ z)];
        acc = mfma(a, bb, acc);
      }
    }
    // plain block: G[i = g + 4e][j = c]
    ldbl* G = OPS + b * 256;
#pragma unroll
    for (int e = 0; e < 4; e++) G[(g + 4 * e) + 16 * c] = acc[e] + ((TOP && p == r && (g + 4 * e) == c) ? 1.0 : 0.0);
  }
  (void)nblocks;
  group_arrive_wait(ubar, utarget);
  // diagonal blocks -> T_p (wave u builds T_u), cross blocks -> image in place
  for (int b = u; b < 10; b += 4) {
    int p, r;
    if (b < 4) { p = b; r = b; }
    else { const int q = b - 4; p = (q < 1) ? 1 : ((q < 3) ? 2 : 3); r = q - p * (p - 1) / 2; }
    if (p >= np) continue;
    ldbl* G = OPS + b * 256;
    double img[4];
    if (b < 4) {
      double trow[16];
      if (lane < 16) {
        t_from_gram(G, tauL + 16 * p, trow, lane);
#pragma unroll
        for (int j = 0; j < 16; j++) G[lane + 16 * j] = trow[j];       // every lane has finished reading G (same wave, in order)
      }
      // Timg[s][lane (g, c)] = T[4g + s][sig(c)]
#pragma unroll
      for (int s = 0; s < 4; s++) img[s] = G[(4 * g + s) + 16 * sig(c)];
    } else {
      // Simg[s][lane] = -S_pr[sig(c)][4g + s]
#pragma unroll
      for (int s = 0; s < 4; s++) img[s] = -G[sig(c) + 16 * (4 * g + s)];
    }
#pragma unroll
    for (int s = 0; s < 4; s++) G[64 * s + lane] = img[s];
  }
  group_arrive_wait(ubar, utarget);
}

// ---------------------------------------------------------------------------------------------- panel group
__device__ __forceinline__ double bperm(double x, int srclane) {
  const int lo = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor_add(double v, int mask) {
  const int lane = threadIdx.x & 63;
  return v + bperm(v, lane ^ mask);
}

struct Refl { double beta, tau, scale; };
__device__ __forceinline__ Refl dlarfg(double alpha, double ss) {
  Refl r;
  if (ss == 0.0) { r.beta = alpha; r.tau = 0.0; r.scale = 0.0; return r; }
  const double n2 = alpha * alpha + ss;
  double ri = __builtin_amdgcn_rsq(n2);
  ri = ri * (1.5 - 0.5 * n2 * ri * ri);
  ri = ri * (1.5 - 0.5 * n2 * ri * ri);              // 1 / ||x||
  double nrm = n2 * ri;
  nrm = nrm + 0.5 * ri * (n2 - nrm * nrm);           // ||x||
  r.beta = -copysign(nrm, alpha);
  r.tau = 1.0 + fabs(alpha) * ri;                    // (beta - alpha) / beta
  const double dd = alpha - r.beta;                  // |dd| = |alpha| + ||x||: no cancellation
  double rd = __builtin_amdgcn_rcp(dd);
  rd = rd * (2.0 - dd * rd);
  rd = rd * (2.0 - dd * rd);
  r.scale = rd;
  return r;
}

// The sixteen column steps of sub-panel PJ.  P[rb][e][p]: chunk row 64 w + 16 rb + 4 g + e, panel column 16 p + c.
// Pt[e][p] (TOP): top row 16 w + 4 g + e.  !TOP: the pivot rows are the chunk's first 64 rows (wave 0).
template <int PJ, bool TOP>
__device__ __forceinline__ void panel16(double (&P)[4][4][4], double (&Pt)[4][4], double (&mytau)[4], int w, ldbl* lds,
                                        int& ptarget, int& step) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  ldbl* part = lds + L_PART;
  ldbl* rowb = lds + L_ROW;
  lint* pbar = (lint*)(lds + L_SYNC);
  for (int cj = 0; cj < 16; cj++) {
    const int j = 16 * PJ + cj;
    const int par = step & 1;
    step++;
    const int src = (lane & 48) | cj;
    double x[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        double v = bperm(P[rb][e][PJ], src);
        if (!TOP) {
          const bool incl = (w > 0) || (rb > PJ) || (rb == PJ && 4 * g + e > cj);
          v = incl ? v : 0.0;
        }
        x[rb][e] = v;
      }
    // column sums of x .* P over this wave's rows
    double d[4];
#pragma unroll
    for (int p = PJ; p < 4; p++) {
      double s = 0.0;
#pragma unroll
      for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int e = 0; e < 4; e++) s += x[rb][e] * P[rb][e][p];
      s = xor_add(s, 16);
      s = xor_add(s, 32);
      d[p] = s;
    }
    if (g == 0) {
#pragma unroll
      for (int p = PJ; p < 4; p++) part[par * 256 + w * 64 + 16 * p + c] = d[p];
    }
    // the pivot (top) row of this step, columns of the panel from sub-panel PJ on
    {
      const int ow = TOP ? PJ : 0;                   // owner wave
      if (w == ow && g == (cj >> 2)) {
        const int ee = cj & 3;
#pragma unroll
        for (int p = PJ; p < 4; p++) {
          double v;
          if (TOP) v = (ee == 0) ? Pt[0][p] : (ee == 1) ? Pt[1][p] : (ee == 2) ? Pt[2][p] : Pt[3][p];
          else v = (ee == 0) ? P[PJ][0][p] : (ee == 1) ? P[PJ][1][p] : (ee == 2) ? P[PJ][2][p] : P[PJ][3][p];
          rowb[par * 64 + 16 * p + c] = v;
        }
      }
    }
    group_arrive_wait(pbar, ptarget);
    double ss = 0.0, dt[4], rv[4];
#pragma unroll
    for (int ww = 0; ww < 4; ww++) ss += part[par * 256 + ww * 64 + 16 * PJ + cj];
#pragma unroll
    for (int p = PJ; p < 4; p++) {
      double s = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ww++) s += part[par * 256 + ww * 64 + 16 * p + c];
      dt[p] = s;
      rv[p] = rowb[par * 64 + 16 * p + c];
    }
    const double alpha = rowb[par * 64 + j];
    const Refl h = dlarfg(alpha, ss);
    mytau[PJ] = (c == cj) ? h.tau : mytau[PJ];      // kept in registers: the update group may still be reading the previous taus
    double tw[4];
#pragma unroll
    for (int p = PJ; p < 4; p++) {
      const double t = h.tau * (rv[p] + h.scale * dt[p]);
      tw[p] = (p == PJ && c <= cj) ? 0.0 : t;
    }
    const bool iscj = (c == cj);
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const double v = x[rb][e] * h.scale;
        bool incl = true;
        if (!TOP) incl = (w > 0) || (rb > PJ) || (rb == PJ && 4 * g + e > cj);
        const double upd = P[rb][e][PJ] - v * tw[PJ];
        P[rb][e][PJ] = (iscj && incl) ? v : upd;
#pragma unroll
        for (int p = PJ + 1; p < 4; p++) P[rb][e][p] -= v * tw[p];
      }
    // the pivot row itself: v = 1 there
    {
      const int ow = TOP ? PJ : 0;
      if (w == ow && g == (cj >> 2)) {
        const int ee = cj & 3;
#pragma unroll
        for (int p = PJ; p < 4; p++) {
          const double nv = (p == PJ && iscj) ? h.beta : (rv[p] - tw[p]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (TOP) Pt[e][p] = (e == ee) ? nv : Pt[e][p];
            else P[PJ][e][p] = (e == ee) ? nv : P[PJ][e][p];
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- the factorisation
// Y: column-major, leading dimension ld (multiple of 16), `rows` a multiple of 64, rows >= cols; zero padding columns
// up to the next multiple of 16 must exist.  On exit the upper trapezoid of Y holds R (below: unspecified).
// lds: L_TOTAL doubles, 16-byte aligned.  All 512 threads of the workgroup must call; contains workgroup barriers at entry/exit.
#ifndef CAQR_QR_ATTR
#define CAQR_QR_ATTR __attribute__((noinline))
#endif
__device__ CAQR_QR_ATTR void qr(gdbl* Y, long ld, int rows, int cols, ldbl* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  Plan pl;
  pl.rows = rows; pl.cols = cols; pl.nblk = (cols + 63) >> 6;
  lint* sync = (lint*)(lds + L_SYNC);
  if (tid < 8) sync[tid] = 0;
  __syncthreads();
  ldbl* V = lds + L_V;
  ldbl* OPS = lds + L_OPS;
  ldbl* tauL = lds + L_TAU;

  if (wave < 4) {
    // ================================================================ panel group
    const int w = wave;
    CAQR_T0();
    int ptarget = 0, step = 0;
    int seq = 0;                  // chunk-blocks published so far
    int seqblk = 0;               // sequence number of chunk 0 of the previous block
    for (int k = 0; k < pl.nblk; k++) {
      const int j0 = 64 * k, top0 = 64 * k;
      const int np = (pl.bw(k) + 15) >> 4;
      const int nch = pl.nch(k);
      double Pt[4][4];
      for (int i = 0; i < nch; i++) {
        const int row0 = 64 * k + 256 * i;
        const int nr = min(256, rows - row0);
        // the columns of this chunk-panel were last written by the update group in the previous block, chunks i and i+1
        if (k > 0) {
          const int prev_nch = pl.nch(k - 1);
          const int need = seqblk + min(i + 1, prev_nch - 1) + 1;
          wait_at_least(sync + 3, 4 * need);
        }
        CAQR_ACC(0);      // waiting for the update group (dependency)
        double P[4][4][4];
        double mytau[4] = {0.0, 0.0, 0.0, 0.0};
        {
          const gdbl* src = Y + (long)(j0 + c) * ld + row0 + 64 * w + 4 * g;
#pragma unroll
          for (int p = 0; p < 4; p++)
#pragma unroll
            for (int rb = 0; rb < 4; rb++) {
              const bool ok = (p < np) && (64 * w + 16 * rb < nr);
              d4 v = d4{0, 0, 0, 0};
              if (ok) v = *reinterpret_cast<const gd4*>(src + (long)(16 * p) * ld + 16 * rb);
#pragma unroll
              for (int e = 0; e < 4; e++) P[rb][e][p] = v[e];
            }
        }
        CAQR_ACC(1);      // panel load
        if (i == 0) {
          panel16<0, false>(P, Pt, mytau, w, lds, ptarget, step);
          CAQR_ACC(2);
          if (np > 1) panel16<1, false>(P, Pt, mytau, w, lds, ptarget, step);
          if (np > 2) panel16<2, false>(P, Pt, mytau, w, lds, ptarget, step);
          if (np > 3) panel16<3, false>(P, Pt, mytau, w, lds, ptarget, step);
        } else {
          panel16<0, true>(P, Pt, mytau, w, lds, ptarget, step);
          CAQR_ACC(2);
          if (np > 1) panel16<1, true>(P, Pt, mytau, w, lds, ptarget, step);
          if (np > 2) panel16<2, true>(P, Pt, mytau, w, lds, ptarget, step);
          if (np > 3) panel16<3, true>(P, Pt, mytau, w, lds, ptarget, step);
        }
        CAQR_ACC(3);      // sub-panels 1..3
        // hand-over: the update group has finished with the previous image
        wait_at_least(sync + 3, 4 * seq);
        CAQR_ACC(4);      // waiting for the LDS image to be free
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
          for (int rb = 0; rb < 4; rb++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int row = 64 * w + 16 * rb + 4 * g + e, col = 16 * p + c;
              double v = P[rb][e][p];
              if (i == 0) v = (row > col) ? v : ((row == col) ? 1.0 : 0.0);      // unit lower trapezoidal head
              if (row >= nr || p >= np) v = 0.0;
              V[row * 64 + (col ^ swz(4 * g + e))] = v;
            }
        if (w == 0 && g == 0) {
#pragma unroll
          for (int p = 0; p < 4; p++) tauL[16 * p + c] = mytau[p];
        }
        if (i == 0) {
          // R_kk (and zeros below it) to memory; it is the top of the later chunks and part of the result
          if (w == 0) {
            gdbl* dst = Y + (long)(j0 + c) * ld + row0 + 4 * g;
#pragma unroll
            for (int p = 0; p < 4; p++)
              if (p < np) {
#pragma unroll
                for (int rb = 0; rb < 4; rb++) {
                  d4 v;
#pragma unroll
                  for (int e = 0; e < 4; e++) v[e] = (16 * rb + 4 * g + e <= 16 * p + c) ? P[rb][e][p] : 0.0;
                  *reinterpret_cast<gd4*>(dst + (long)(16 * p) * ld + 16 * rb) = v;
                }
              }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        group_arrive_wait(sync + 0, ptarget);                 // image, taus (and R_kk) complete
        CAQR_ACC(5);      // image write
        if (w == 0 && lane == 0) __hip_atomic_store(sync + 2, seq + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        seq++;
        if (i == 0 && nch > 1) {
          const gdbl* src = Y + (long)(j0 + c) * ld + top0 + 16 * w + 4 * g;
#pragma unroll
          for (int p = 0; p < 4; p++) {
            d4 v = d4{0, 0, 0, 0};
            if (p < np) v = *reinterpret_cast<const gd4*>(src + (long)(16 * p) * ld);
#pragma unroll
            for (int e = 0; e < 4; e++) Pt[e][p] = v[e];
          }
        }
      }
      if (nch > 1) {
        gdbl* dst = Y + (long)(j0 + c) * ld + top0 + 16 * w + 4 * g;
#pragma unroll
        for (int p = 0; p < 4; p++)
          if (p < np) {
            d4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = Pt[e][p];
            *reinterpret_cast<gd4*>(dst + (long)(16 * p) * ld) = v;
          }
      }
      seqblk = seq - nch;
    }
  } else {
    // ================================================================ update group
    const int u = wave - 4;
    CAQR_T0();
    int utarget = 0;
    int seq = 0;
    for (int k = 0; k < pl.nblk; k++) {
      const int j0 = 64 * k, top0 = 64 * k;
      const int np = (pl.bw(k) + 15) >> 4;
      const int nch = pl.nch(k);
      const int ntl = pl.ntl(k);
      for (int i = 0; i < nch; i++) {
        const int row0 = 64 * k + 256 * i;
        const int nr = min(256, rows - row0);
        const int nrb = nr >> 4;
        wait_at_least(sync + 2, seq + 1);
        CAQR_ACC(0);      // waiting for the panel group
        if (ntl > 0) {
          if (i == 0) build_images<false>(u, nrb, np, V, OPS, tauL, sync + 1, utarget);
          else build_images<true>(u, nrb, np, V, OPS, tauL, sync + 1, utarget);
          CAQR_ACC(1);    // Gram blocks, T, operand images
          for (int t = u; t < ntl; t += 4) {
            const int col0 = j0 + 64 + 16 * t;
            if (i == 0) {
              if (nrb == 16) update_tile<16, false>(Y, ld, row0, col0, top0, np, V, OPS);
              else if (nrb == 12) update_tile<12, false>(Y, ld, row0, col0, top0, np, V, OPS);
              else if (nrb == 8) update_tile<8, false>(Y, ld, row0, col0, top0, np, V, OPS);
              else update_tile<4, false>(Y, ld, row0, col0, top0, np, V, OPS);
            } else {
              if (nrb == 16) update_tile<16, true>(Y, ld, row0, col0, top0, np, V, OPS);
              else if (nrb == 12) update_tile<12, true>(Y, ld, row0, col0, top0, np, V, OPS);
              else if (nrb == 8) update_tile<8, true>(Y, ld, row0, col0, top0, np, V, OPS);
              else update_tile<4, true>(Y, ld, row0, col0, top0, np, V, OPS);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          CAQR_ACC(2);    // tiles
        }
        seq++;
        if (lane == 0) __hip_atomic_fetch_add(sync + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
}

}  // namespace caqr
