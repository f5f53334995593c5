// Communication-avoiding form of the batched R-only QR (grid level): the gauge-sweep factorisation of the compress
// engine for Y_t too tall for one workgroup (BASELINE configs[2..4]; reference: the orthogonalize_right! half of
// `compress!` inside `op`, src/recursive_bp_factor.jl:127 - only R = Lf^T is kept, engine.h).
//
// The launch-per-panel Householder form (v2_kernels.h) applies full-height block reflectors: per 64 columns the trailing
// matrix is read twice and written once, the reflectors are streamed twice, and the partial products of the row chunks
// meet in memory.  Here every 64-column block is reduced by a 4-ary TREE of 256-row nodes (CAQR):
//
//   level 0   node = 256 consecutive rows from row 64 k on: plain Householder QR of its 256 x 64 piece of the block;
//             its R (64 x 64) stays in the node's first 64 rows
//   level l   node = the first 64 rows of four level l-1 nodes (four SEGMENTS of 64 rows): the same factorisation of
//             the stacked R factors; the root's R lands in rows [64 k, 64 k + 64) - where R_k belongs
//
// so the reflectors of a node touch only the node's 256 rows.  Two kernels per level:
//   k_cq_fac2 one 256-thread workgroup per node: the 64 column steps entirely in registers (rows across lanes in the MFMA
//             B-operand layout; a column is broadcast along its 16-lane row with DPP, its dot products are in-lane sums +
//             two permlane levels + one LDS exchange of the four waves), the later sub-panels updated on the matrix pipes;
//             the node's reflectors go to its slot of the per-problem scratch as an XOR-swizzled image, followed by the
//             T / cross-Gram operand images (MFMA Grams from the registers, dlarft recurrence)
//   k_cq_upd  workgroups over (tile group, node, problem): the image (148 KB) into LDS once, then every wave takes
//             256 x 16 tiles of the trailing matrix: read ONCE into registers, W0 = V^T C, W = T-recurrence, C -= V W,
//             written ONCE.  Measured alone (tools/probes/caqr_update_probe.hip): 60 % of the fp64 MFMA peak with 256
//             workgroups streaming, against ~35 % for the streamed-reflector passes.
// A level's update must precede the next level's (they share the segment rows); the factorisations of the levels only
// depend on each other (they touch the block's own 64 columns).  The tree costs 1/3 more update flops than a
// full-height reflector (the stacked triangles are treated as dense) - at more than twice the rate.
//
// Layouts (lane = 16 g + c):
//   B/D layout of a 16-row group: lane (g, c) holds rows 4g + e (e = 0..3, one d4 = 32 contiguous bytes) of column c.
//     As the B operand of k-step e the MFMA k index g stands for row 4g + e; as the accumulator D, register e of lane
//     (g, c) is MFMA row g + 4e, which therefore also stands for row 4g + e: sigma(i) = 4 (i & 3) + (i >> 2).
//   V image: V[row * 64 + (col ^ swz(row & 15))], swz(m) = (m & 3) | (m & 8) | ((m & 4) << 2)   (row = node row 0..255)
//   operand images of the 16-column triangular factors T_p (p = 0..3) and the cross Grams S_pr (r < p):
//     Timg[p][s][lane (g, c)] = T_p[4g + s][sigma(c)],  Simg[p,r][s][lane] = -S_pr[sigma(c)][4g + s]
#pragma once
#include <type_traits>

namespace cq {
using namespace wgc;

constexpr int IMG_V = 256 * 64;             // doubles
constexpr int IMG_OPS = 10 * 256;
constexpr int IMG_DOUBLES = IMG_V + IMG_OPS;          // one node's slot in the scratch: 18944 doubles = 151,552 bytes
__device__ __forceinline__ int swz(int m) { return (m & 3) | (m & 8) | ((m & 4) << 2); }
__device__ __forceinline__ int sig(int i) { return 4 * (i & 3) + (i >> 2); }

// The node `node` of level `level` of block row offset j0 (= 64 k) in a matrix of rows32 rows: first row and valid row
// count of its four 64-row segments.  Returns whether the node has work (level 0: any row; above: two children or more).
__device__ __forceinline__ bool node_segments(int rows32, int j0, int level, int node, int (&base)[4], int (&cnt)[4]) {
  long first, stride;
  if (level == 0) { first = j0 + 256L * node; stride = 64; }
  else {
    long ls = 256;
    for (int i = 1; i < level; i++) ls *= 4;
    first = j0 + 4L * node * ls; stride = ls;
  }
  int nseg = 0;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const long b = first + stride * s;
    const long n = (long)rows32 - b;
    const int m = n > 64 ? 64 : (n > 0 ? (int)n : 0);
    base[s] = m > 0 ? (int)b : j0;
    cnt[s] = m;
    nseg += (m > 0);
  }
  return level == 0 ? (cnt[0] > 0) : (nseg >= 2);
}

struct Refl { double beta, tau, scale; };
// LAPACK dlarfg from the pivot and the squared norm below it (rsq / rcp + Newton, as v2::reflector)
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

// ------------------------------------------------------------------------------------------------------------------
// Factorisation of a node: the 64 columns as four 16-column SUB-PANELS.  The column steps touch only their
// own sub-panel (VALU: a third of the work of updating all four at every step), the later sub-panels are brought up to
// date with the sub-panel's block reflector on the matrix pipes (W0 = V^T C summed over the four waves through LDS,
// W = T^T W0, C -= V W; C never leaves the registers).  A column is broadcast along its 16-lane row with DPP
// (row_newbcast - one VALU move instead of a trip through the LDS crossbar), the partial dot products of all 16
// (wave, row group) pairs meet in LDS with one barrier per column.
// TREE: the node is a stack of four upper-triangular R factors, reflector j only touches rows <= j of segments 1..3
// (the zero rows are skipped - 5/8 of the work on average).
// ------------------------------------------------------------------------------------------------------------------
// LDS of the node factorisation (68 KB: two node workgroups per CU).  The reflector image itself goes straight to the node's slot in
// HBM; the LDS keeps the operand images, the exchange areas of the column steps and ONE 16-column panel of the image
// (leading dimension 17: both MFMA operand shapes read it conflict free) for the in-register update of the later sub-panels.
constexpr int F_OPS = 0;                    // [10][256] operand images
constexpr int F_SCR = F_OPS + IMG_OPS;      // [4][256]
constexpr int L2_PART = F_SCR + 4 * 256;    // [2][4][64]
constexpr int L2_ROW = L2_PART + 512;       // [2][16]
constexpr int L2_TAU = L2_ROW + 32;         // [64]
constexpr int F_VP = L2_TAU + 64;           // [256][17] panel; after the last sub-panel: [3][4][256] cross-Gram partial sums
constexpr int VP_LD = 17;
constexpr int L2_FAC_TOTAL = F_VP + 256 * VP_LD;   // 8544 doubles = 68,352 bytes
// 512 bytes behind the image that nobody reads: the landing zone of the tile touches (touch_tile)
constexpr int TOUCH_DOUBLES = 64;
constexpr int UPD_LDS_DOUBLES = IMG_DOUBLES + TOUCH_DOUBLES;          // dynamic LDS of k_cq_upd and k_cq_updfac
constexpr int FAC_LDS_DOUBLES = L2_FAC_TOTAL;                         // dynamic LDS of k_cq_fac2
static_assert(L2_FAC_TOTAL <= IMG_DOUBLES, "the factor workgroups of k_cq_updfac live inside the update's LDS");

template <int CJ>
__device__ __forceinline__ double bcast16(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + CJ, 0xf, 0xf, false);      // v_mov_b64_dpp row_newbcast:CJ
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// sum over the four 16-lane rows of the wave, in every lane (v_permlane32_swap / v_permlane16_swap: no LDS)
__device__ __forceinline__ double reduce_rows(double s) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  {
    const unsigned lo = __double2loint(s), hi = __double2hiint(s);
    const u2v a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const u2v b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  {
    const unsigned lo = __double2loint(s), hi = __double2hiint(s);
    const u2v a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const u2v b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    s = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  return s;
}

#ifdef CQ_PROF
__device__ unsigned long long cq_prof[4][8];
#define CQ_T(slot) do { const unsigned long long t1_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) cq_prof[w][slot] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define CQ_T(slot) do {} while (0)
#endif

// The sixteen column steps of sub-panel PJ.  During the steps column cj keeps the UNSCALED vector x below its pivot (the
// update of the other columns is P -= x (scale tw)); the columns are scaled to v = x scale once, after the last step.
// MASK: some of this wave's rows are excluded (the first segment's rows up to the pivot; the zero rows of a triangle).
// The steps are written out with the column index CJ a compile-time constant (round-3 review item 2: as a run-time loop
// the sixteen broadcasts sat behind a 16-way `switch` - an indirect jump per step that the instruction prefetch cannot
// follow - and every pivot-row access was a select): the broadcast, the readlane of the pivot column and the pivot-row
// element are immediates, and the row groups a step cannot touch (above the pivot's group in the first segment of a dense
// node) are skipped at compile time.
template <int PJ, bool TREE, bool MASK, int CJ>
__device__ __forceinline__ void col_step(double (&P)[4][4][4], double (&mytau)[4], double& myscale, int w, int g, int c, ldbl* part, ldbl* rowb) {
  constexpr int par = CJ & 1, GJ = CJ >> 2, EJ = CJ & 3;
  // row groups of this wave that reflector CJ can touch
  auto live = [](int rb) constexpr { return TREE ? (rb <= PJ) : (MASK ? (rb >= PJ) : true); };
#ifdef CQ_PROF
  unsigned long long t0_ = __builtin_readcyclecounter();
#endif
  double x[4][4];
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int rb = 0; rb < 4; rb++) {
    if (!live(rb)) continue;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      double xv = bcast16<CJ>(P[rb][e][PJ]);
      if (MASK) {
        bool incl;
        if (TREE) incl = (rb < PJ || 4 * g + e <= CJ);                              // segments 1..3 of a stack of triangles
        else incl = (rb > PJ) || (4 * g + e > CJ);                                  // first segment of a dense node (rb == PJ)
        xv = incl ? xv : 0.0;
      }
      x[rb][e] = xv;
      if (e & 1) s1 += xv * P[rb][e][PJ]; else s0 += xv * P[rb][e][PJ];
    }
  }
  CQ_T(1);
  const double sw = reduce_rows(s0 + s1);
  if (g == 0) part[par * 64 + w * 16 + c] = sw;
  if (w == 0 && g == GJ) rowb[par * 16 + c] = P[PJ][EJ][PJ];
  lds_barrier();
  CQ_T(2);
  const double dt = (part[par * 64 + c] + part[par * 64 + 16 + c]) + (part[par * 64 + 32 + c] + part[par * 64 + 48 + c]);
  const double rv = rowb[par * 16 + c];
  const double ss = readlane_d(dt, CJ), alpha = readlane_d(rv, CJ);
  CQ_T(3);
  const Refl h = dlarfg(alpha, ss);
  const bool iscj = (c == CJ);
  mytau[PJ] = iscj ? h.tau : mytau[PJ];
  myscale = iscj ? h.scale : myscale;
  const double tw = (c <= CJ) ? 0.0 : h.tau * (rv + h.scale * dt);
  const double tws = -h.scale * tw;
  CQ_T(4);
#pragma unroll
  for (int rb = 0; rb < 4; rb++) {
    if (!live(rb)) continue;
#pragma unroll
    for (int e = 0; e < 4; e++) P[rb][e][PJ] += x[rb][e] * tws;
  }
  if (w == 0 && g == GJ) P[PJ][EJ][PJ] = iscj ? h.beta : (rv - tw);
  CQ_T(5);
  if constexpr (CJ + 1 < 16) col_step<PJ, TREE, MASK, CJ + 1>(P, mytau, myscale, w, g, c, part, rowb);
}
template <int PJ, bool TREE, bool MASK>
__device__ __forceinline__ void subpanel_steps_w(double (&P)[4][4][4], double (&mytau)[4], int w, ldbl* lds) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  double myscale = 0.0;
  col_step<PJ, TREE, MASK, 0>(P, mytau, myscale, w, g, c, lds + L2_PART, lds + L2_ROW);
  // v = x scale below the pivots
#pragma unroll
  for (int rb = 0; rb < 4; rb++) {
    if (TREE && rb > PJ) continue;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      bool incl = true;
      if (MASK && !TREE) incl = (rb > PJ) || (rb == PJ && 4 * g + e > c);
      P[rb][e][PJ] = incl ? P[rb][e][PJ] * myscale : P[rb][e][PJ];
    }
  }
}

// A wave of the first segment of a stack of triangles has no rows below the pivots: it only hands out the pivot rows.
template <int PJ, int CJ>
__device__ __forceinline__ void pivot_step(double (&P)[4][4][4], double (&mytau)[4], int g, int c, ldbl* part, ldbl* rowb) {
  constexpr int par = CJ & 1, GJ = CJ >> 2, EJ = CJ & 3;
  if (g == 0) part[par * 64 + c] = 0.0;
  if (g == GJ) rowb[par * 16 + c] = P[PJ][EJ][PJ];
  lds_barrier();
  const double dt = (part[par * 64 + c] + part[par * 64 + 16 + c]) + (part[par * 64 + 32 + c] + part[par * 64 + 48 + c]);
  const double rv = rowb[par * 16 + c];
  const double ss = readlane_d(dt, CJ), alpha = readlane_d(rv, CJ);
  const Refl h = dlarfg(alpha, ss);
  const bool iscj = (c == CJ);
  mytau[PJ] = iscj ? h.tau : mytau[PJ];
  const double tw = (c <= CJ) ? 0.0 : h.tau * (rv + h.scale * dt);
  if (g == GJ) P[PJ][EJ][PJ] = iscj ? h.beta : (rv - tw);
  if constexpr (CJ + 1 < 16) pivot_step<PJ, CJ + 1>(P, mytau, g, c, part, rowb);
}
template <int PJ>
__device__ __forceinline__ void subpanel_steps_pivots(double (&P)[4][4][4], double (&mytau)[4], ldbl* lds) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  pivot_step<PJ, 0>(P, mytau, g, c, lds + L2_PART, lds + L2_ROW);
}

template <int PJ, bool TREE>
__device__ __forceinline__ void subpanel_steps(double (&P)[4][4][4], double (&mytau)[4], int w, ldbl* lds) {
  if (TREE) {
    if (w == 0) subpanel_steps_pivots<PJ>(P, mytau, lds);
    else subpanel_steps_w<PJ, true, true>(P, mytau, w, lds);
  } else {
    if (w == 0) subpanel_steps_w<PJ, false, true>(P, mytau, w, lds);
    else subpanel_steps_w<PJ, false, false>(P, mytau, w, lds);
  }
}

// T of a sixteen-reflector panel from its Gram matrix (dlarft) in registers: lane c of every 16-lane row holds row c of G
// and builds row c of T; G(i2, j) of another row comes by DPP broadcast from lane i2.
//   T(i,j) = -tau_j sum_{i2=i}^{j-1} T(i,i2) G(i2,j)  (i < j),  T(j,j) = tau_j   (T(i,i2) = 0 for i2 < i)
template <int J, int I2>
__device__ __forceinline__ void t_acc(const double (&grow)[16], const double (&trow)[16], double& sacc) {
  if constexpr (I2 < J) { sacc += trow[I2] * bcast16<I2>(grow[J]); t_acc<J, I2 + 1>(grow, trow, sacc); }
}
template <int J>
__device__ __forceinline__ void t_cols(const double (&grow)[16], double (&trow)[16], double tauc, int c) {
  if constexpr (J < 16) {
    const double tj = bcast16<J>(tauc);
    double sacc = 0.0;
    t_acc<J, 0>(grow, trow, sacc);
    trow[J] = (J == c) ? tj : ((J > c) ? -tj * sacc : 0.0);
    t_cols<J + 1>(grow, trow, tauc, c);
  }
}

// After the column steps of sub-panel PJ: its reflectors into the node's image (Vg: the slot in HBM, nullptr when nobody
// will read it) and into the LDS panel, T_PJ (operand image in OPS slot PJ), and the block-reflector update of the later
// sub-panels of the block in registers.
template <int PJ, bool TREE>
__device__ __forceinline__ void subpanel_finish(double (&P)[4][4][4], const double (&mytau)[4], int w, int mycnt, int np, ldbl* lds, gdbl* Vg) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  ldbl* Vp = lds + F_VP;
  ldbl* OPS = lds + F_OPS;
  ldbl* tauL = lds + L2_TAU;
  ldbl* scr = lds + F_SCR;
  {
    d4 acc = d4{0, 0, 0, 0};
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rl = 16 * rb + 4 * g + e, col = 16 * PJ + c;
        double v = P[rb][e][PJ];
        if (w == 0) v = (rl > col) ? v : ((rl == col) ? 1.0 : 0.0);
        if (16 * rb >= mycnt) v = 0.0;
        if (Vg) Vg[(64 * w + rl) * 64 + (col ^ swz(4 * g + e))] = v;
        Vp[(64 * w + rl) * VP_LD + c] = v;
        if (!(TREE && rb > PJ)) acc = mfma(v, v, acc);
      }
#pragma unroll
    for (int e = 0; e < 4; e++) scr[w * 256 + 64 * e + lane] = acc[e];
    if (w == 0 && g == 0) tauL[16 * PJ + c] = mytau[PJ];
  }
  lds_barrier();
  if (w == 0) {
    ldbl* G = OPS + PJ * 256;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      double sgm = 0.0;
#pragma unroll
      for (int ww = 0; ww < 4; ww++) sgm += scr[ww * 256 + 64 * e + lane];
      G[(g + 4 * e) + 16 * c] = sgm;
    }
    // dlarft recurrence with row c of G in registers and the entries of the other rows taken by DPP broadcast
    double grow[16], trow[16];
#pragma unroll
    for (int j = 0; j < 16; j++) grow[j] = G[c + 16 * j];
    t_cols<0>(grow, trow, mytau[PJ], c);
    if (g == 0) {
#pragma unroll
      for (int j = 0; j < 16; j++) G[c + 16 * j] = trow[j];
    }
    double img[4];
#pragma unroll
    for (int s = 0; s < 4; s++) img[s] = G[(4 * g + s) + 16 * sig(c)];
#pragma unroll
    for (int s = 0; s < 4; s++) G[64 * s + lane] = img[s];
  }
  lds_barrier();
  if (PJ < 3 && PJ + 1 < np) {
    const int sc = sig(c);
    double aA[4][4], aC[4][4];
#pragma unroll
    for (int rb = 0; rb < 4; rb++) {
      if (TREE && rb > PJ) continue;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int m = 4 * g + e;
        aA[rb][e] = Vp[(64 * w + 16 * rb + m) * VP_LD + sc];          // own rows only: no other wave writes or reads them
        aC[rb][e] = Vp[(64 * w + 16 * rb + sc) * VP_LD + m];
      }
    }
    double timg[4];
#pragma unroll
    for (int s = 0; s < 4; s++) timg[s] = OPS[PJ * 256 + 64 * s + lane];
#pragma unroll
    for (int p = PJ + 1; p < 4; p++) {
      if (p < np) {
        d4 acc = d4{0, 0, 0, 0};
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
          if (TREE && rb > PJ) continue;
#pragma unroll
          for (int e = 0; e < 4; e++) acc = mfma(aA[rb][e], P[rb][e][p], acc);
        }
#pragma unroll
        for (int e = 0; e < 4; e++) scr[w * 256 + 64 * e + lane] = acc[e];
        lds_barrier();
        d4 t = d4{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          double sgm = 0.0;
#pragma unroll
          for (int ww = 0; ww < 4; ww++) sgm += scr[ww * 256 + 64 * e + lane];
          t[e] = sgm;
        }
        d4 o = d4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 4; s++) o = mfma(timg[s], t[s], o);
        o = -o;
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
          if (TREE && rb > PJ) continue;
          d4 cc = d4{P[rb][0][p], P[rb][1][p], P[rb][2][p], P[rb][3][p]};
#pragma unroll
          for (int s = 0; s < 4; s++) cc = mfma(aC[rb][s], o[s], cc);
#pragma unroll
          for (int e = 0; e < 4; e++) P[rb][e][p] = cc[e];
        }
        lds_barrier();
      }
    }
  }
}

// Cross Grams S_pr = V_p^T V_r (r < p) of the finished node -> operand images in OPS slots 4..9, from the registers: the
// sub-panels still sit in P (below the pivots; wave 0's triangle is masked to the unit-lower-trapezoidal head on the fly).
// Two rounds of three blocks; the partial sums of the four waves meet in the panel area, which is free by now.
//   blocks: 4 + p(p-1)/2 + r = (p,r).  G_pr[i][j] = sum_rows V[row][16p+i] V[row][16r+j]
template <bool TREE>
__device__ __forceinline__ void cross_grams(const double (&P)[4][4][4], int w, int mycnt, int np, ldbl* lds) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  ldbl* OPS = lds + F_OPS;
  ldbl* part = lds + F_VP;                      // [3][4][256]
#pragma unroll
  for (int round = 0; round < 2; round++) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int b = 4 + 3 * round + k;
      const int q = b - 4;
      const int p = (q < 1) ? 1 : ((q < 3) ? 2 : 3), r = q - p * (p - 1) / 2;
      d4 acc = d4{0, 0, 0, 0};
#pragma unroll
      for (int rb = 0; rb < 4; rb++) {
        if (TREE && rb > r) continue;             // sub-panel r of a stack of triangles is zero below row group r
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int rl = 16 * rb + 4 * g + e;
          double a = P[rb][e][p], bb = P[rb][e][r];
          if (w == 0) {
            a = (rl > 16 * p + c) ? a : ((rl == 16 * p + c) ? 1.0 : 0.0);
            bb = (rl > 16 * r + c) ? bb : ((rl == 16 * r + c) ? 1.0 : 0.0);
          }
          if (16 * rb >= mycnt || p >= np) { a = 0.0; bb = 0.0; }
          acc = mfma(a, bb, acc);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; e++) part[(k * 4 + w) * 256 + 64 * e + lane] = acc[e];
    }
    lds_barrier();
    if (w < 3) {
      const int b = 4 + 3 * round + w;
      ldbl* G = OPS + b * 256;
      // plain block G[i = g + 4e][j = c], then the image Simg[s][lane] = -S_pr[sig(c)][4g + s] in place (same wave, in order)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        double sgm = 0.0;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) sgm += part[(w * 4 + ww) * 256 + 64 * e + lane];
        G[(g + 4 * e) + 16 * c] = sgm;
      }
      double img[4];
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) img[s2] = -G[sig(c) + 16 * (4 * g + s2)];
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) G[64 * s2 + lane] = img[s2];
    }
    lds_barrier();
  }
}


// k_cq_fac2: grid (nodes of the level, problems), 256 threads, FAC_LDS_DOUBLES doubles of dynamic LDS (two workgroups per CU).
template <bool TREE>
__device__ __forceinline__ void fac2_body(const v2::QrProb& Pr, int64_t ws_off, int jb, int slot, const int (&base)[4], const int (&cnt)[4],
                                          int la, ldbl* lds) {
  (void)la;
  const int np = min(4, (Pr.kmax - jb + 15) >> 4);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  gdbl* Y = (gdbl*)Pr.Y;
  const long ld = Pr.ld;
  const int mybase = (w == 0) ? base[0] : (w == 1) ? base[1] : (w == 2) ? base[2] : base[3];
  const int mycnt = (w == 0) ? cnt[0] : (w == 1) ? cnt[1] : (w == 2) ? cnt[2] : cnt[3];
  ldbl* OPS = lds + F_OPS;
  const int cols16 = (Pr.cols + 15) & ~15;
  const bool trailing = cols16 > jb + 64;
  // the node's image: [256][64] reflectors (XOR-swizzled) + the operand images, read by the update of this level
  gdbl* slotp = (gdbl*)Pr.aux + ws_off + (long)slot * IMG_DOUBLES;
  gdbl* Vg = trailing ? slotp : nullptr;

  double P[4][4][4];
  double mytau[4] = {0.0, 0.0, 0.0, 0.0};
  {
    const gdbl* src = Y + (long)(jb + c) * ld + mybase + 4 * g;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int rb = 0; rb < 4; rb++) {
        d4 v = d4{0, 0, 0, 0};
        if (p < np && 16 * rb < mycnt) v = *reinterpret_cast<const gd4*>(src + (long)(16 * p) * ld + 16 * rb);
#pragma unroll
        for (int e = 0; e < 4; e++) P[rb][e][p] = v[e];
      }
  }
  subpanel_steps<0, TREE>(P, mytau, w, lds);
  subpanel_finish<0, TREE>(P, mytau, w, mycnt, np, lds, Vg);
  if (np > 1) { subpanel_steps<1, TREE>(P, mytau, w, lds); subpanel_finish<1, TREE>(P, mytau, w, mycnt, np, lds, Vg); }
  if (np > 2) { subpanel_steps<2, TREE>(P, mytau, w, lds); subpanel_finish<2, TREE>(P, mytau, w, mycnt, np, lds, Vg); }
  if (np > 3) { subpanel_steps<3, TREE>(P, mytau, w, lds); subpanel_finish<3, TREE>(P, mytau, w, mycnt, np, lds, Vg); }
  // R to its place (first segment: upper triangle, zeros below)
  if (w == 0) {
    gdbl* dst = Y + (long)(jb + c) * ld + mybase + 4 * g;
#pragma unroll
    for (int p = 0; p < 4; p++)
      if (p < np) {
#pragma unroll
        for (int rb = 0; rb < 4; rb++) {
          if (16 * rb < mycnt) {
            d4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (16 * rb + 4 * g + e <= 16 * p + c) ? P[rb][e][p] : 0.0;
            *reinterpret_cast<gd4*>(dst + (long)(16 * p) * ld + 16 * rb) = v;
          }
        }
      }
  }
  if (!trailing) return;
  if (np < 4) {
    // (cannot happen for rows >= cols: a block with trailing columns has four panels) - keep the image well defined
#pragma unroll
    for (int p = 0; p < 4; p++)
      if (p >= np) {
#pragma unroll
        for (int rb = 0; rb < 4; rb++)
#pragma unroll
          for (int e = 0; e < 4; e++) Vg[(64 * w + 16 * rb + 4 * g + e) * 64 + ((16 * p + c) ^ swz(4 * g + e))] = 0.0;
        for (int i = lane; i < 256; i += 64) if (w == 0) OPS[p * 256 + i] = 0.0;
      }
    lds_barrier();
  }
  cross_grams<TREE>(P, w, mycnt, np, lds);
  // the operand images -> behind the reflectors in the node's slot
  {
    gd4* dstv = reinterpret_cast<gd4*>(slotp + IMG_V);
    typedef __attribute__((address_space(3))) d4 ld4;
    const ld4* srcv = reinterpret_cast<const ld4*>(OPS);
    for (int i = tid; i < IMG_OPS / 4; i += 256) dstv[i] = srcv[i];
  }
}

__device__ __forceinline__ void fac2_kernel_body(const v2::QrProb* probs, int64_t ws_off, int jb, int level, int slot0, int la, ldbl* lds) {
  const v2::QrProb Pr = probs[blockIdx.y];
  if (jb >= Pr.kmax) return;
  const int rows32 = (Pr.rows + 31) & ~31;
  int base[4], cnt[4];
  if (!node_segments(rows32, jb, level, blockIdx.x, base, cnt)) return;
#ifdef CQ_REPS
  for (int rep = 0; rep < CQ_REPS; rep++) {            // probe builds: the same node again and again (warm instruction cache)
    if (level == 0) fac2_body<false>(Pr, ws_off, jb, slot0 + blockIdx.x, base, cnt, la, lds);
    else fac2_body<true>(Pr, ws_off, jb, slot0 + blockIdx.x, base, cnt, la, lds);
    __syncthreads();
  }
  return;
#endif
  if (level == 0) fac2_body<false>(Pr, ws_off, jb, slot0 + blockIdx.x, base, cnt, la, lds);
  else fac2_body<true>(Pr, ws_off, jb, slot0 + blockIdx.x, base, cnt, la, lds);
}
// One 256-thread workgroup per node and per CU: a lone wave per SIMD with the whole register file (256 + ~240 registers, no
// scratch).  (Round 3 also built the body at two waves per SIMD - `k_cq_fac2x2`, 256 registers, 1072 spills, 459 scratch loads
// and 334 stores of them inside the column steps - for launches with more nodes than CUs; with the column steps straight-line
// it no longer bought anything - 63.5 against 63.8 ms at 7200 x 900 x 128 - and was removed in round 4.)
__global__ void __launch_bounds__(256) k_cq_fac2(const v2::QrProb* probs, int64_t ws_off, int jb, int level, int slot0, int la) {
  extern __shared__ __attribute__((aligned(16))) double cq_lds_raw[];
  fac2_kernel_body(probs, ws_off, jb, level, slot0, la, (ldbl*)cq_lds_raw);
}

// ------------------------------------------------------------------------------------------------------------------
// One trailing tile (256 node rows in four segments x 16 columns at col0) against the node's LDS image.
// nrb: valid 16-row groups (a prefix of the node's rows).
// TREE: the node is a stack of four triangles - sub-panel p of the image is zero in the row groups (rb & 3) > p of every
// segment, those products are skipped (5/8 of the MFMAs remain).
constexpr bool tree_skip(bool tree, int rb, int p) { return tree && (rb & 3) > p; }
constexpr int pair_next(bool tree, int rb, int p) {           // the next (row group, sub-panel) = 4 rb + p after this one (64: none)
  for (int k = 4 * rb + p + 1; k < 64; k++) if (!tree_skip(tree, k >> 2, k & 3)) return k;
  return 64;
}
// Touches the tile at column col_next: one 4-byte load per 128-byte line of the wave's 16 column segments (row group
// (lane >> 4) + 4 k, column lane & 15), so that the real load one tile later finds it in a cache instead of waiting for HBM.  A wave
// of the update has no second wave on its SIMD to switch to, and a second tile buffer in registers gains nothing (upd_body).  The
// loads go straight to an LDS landing zone nobody reads (global_load_lds_dword: no destination register, so nothing the
// register allocator does can be hit by the late write) and are issued AFTER the wave has waited for its own tile, from
// inline assembly the compiler's wait-count model does not see: no instruction waits for them, a later wait at most over-waits.
__device__ __forceinline__ void touch_tile(const gdbl* Y, long ld, const int (&base)[4], int nrb, int col_next, unsigned lds_off) {
#if !defined(CQ_NO_TOUCH) && !defined(CQ_NO_GLOBAL)
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const gdbl* cp = Y + (long)(col_next + c) * ld;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int rb = g + 4 * k;
    const int rbc = rb < nrb ? rb : 0;
    const gdbl* p = cp + base[rbc >> 2] + 16 * (rbc & 3);
    unsigned m0save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0save) : "v"(p), "s"(lds_off) : "memory");
  }
#endif
}
#ifdef CQ_TRACE
// probe builds: per wave and tile, the shader clock at [0] tile requested, [1] tile arrived, [2] end of phase A, [3] of B, [4] of C
__device__ unsigned long long* cq_trace_buf;
#define CQ_TR(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
  if ((threadIdx.x & 63) == 0) cq_trace_buf[((long)(blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64 * 5 + cq_tr_tile * 5 + (k)] = t_; \
  __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ int cq_tr_dummy;
#else
#define CQ_TR(k) do {} while (0)
#endif
#ifdef CQ_TRACE
#define CQ_TR0(tile) do { const int cq_tr_tile = (tile); CQ_TR(0); } while (0)
#else
#define CQ_TR0(tile) do {} while (0)
#endif
// ------------------------------------------------------------------------------------------------------------------
// compute_tile: one trailing tile (256 node rows in four segments x 16 columns) against the node's LDS image.
//
// Written (round 4) around what tools/probes/mfma_operand_probe.hip measured on gfx950 (profiles/r04_mfma_operand_probe.txt):
//  * v_mfma_f64_16x16x4_f64 issues every 64 cycles from ONE wave per SIMD whatever the operands are: a dependent chain on one
//    accumulator, 2 - 16 accumulators taking turns, 1 - 16 distinct A and B source registers - 77.4 - 77.9 TFLOP/s chip-wide in
//    every case (the data-sheet 78.6).  (An earlier note in this file blamed "the variety of the MFMA's source registers" for
//    the 49 TFLOP/s ceiling of this kernel's instruction stream: wrong - the stand-in builds with few operand registers had
//    three of the four sub-panel products of phase A merged by the compiler and executed 65 % of the MFMAs.)
//  * the fp64 MFMA runs on the SIMD's own double-precision lanes (matrix and vector fp64 peak are the same number), and EVERY
//    VALU instruction of ANY wave of that SIMD is time taken from the MFMA stream: one v_add_u32 behind each MFMA of a chain
//    costs 11.8 cycles, each further one 4 - 5 (8 per MFMA: 104.6 instead of 64 cycles), a v_accvgpr_write + read pair 19; a
//    second wave on the SIMD does not hide them (2 waves, 1 VALU each per MFMA: 69.9 cycles per MFMA);
//  * one ds_read_b64 behind every MFMA is free, two cost 9.5 cycles.
// The round-3 tile code had 1.4 VALU instructions per MFMA (LDS addresses with > 16-bit offsets re-added before every read:
// 708 v_add_u32 per 1824 MFMAs; every tile load and store moved through VGPR <-> AGPR copies: 1432; 130 address adds) and its
// phases A / C took 1.5 - 2.2 x their MFMA issue time.  Here the MFMA stream has none:
//  * every LDS address is one of 16 registers per phase (set up at the start of the phase from two lane constants the
//    optimiser cannot see through - else it hoists them out of the tile loop and spills them, or folds the 64 KB half into
//    the immediate and re-adds it) + a 16-bit immediate;
//  * global addresses are a uniform segment pointer (scalar registers) + ONE per-lane byte offset + an immediate;
//  * the LDS operands ping-pong between two register sets by a compile-time index, one read of the next group in front of
//    every MFMA of this one (no copies, one ds_read per MFMA);
//  * the sign of phase C rides on the MFMA's negate-A bit; a full node's tile is loaded without the zero fill;
//  * nothing is kept in a VGPR across phases that a scratch reload would have to bring back (a scratch load waits for the
//    outstanding tile loads as well).
// Per tile and wave, shader cycles, 256 tiles x 64 nodes (tools/probes/cq_upd_probe.hip -DCQ_TRACE; MFMA issue 16384 / 2560 /
// 16384): phase A 17.1 K, B 3.7 K, C 18.8 K (20.2 K with the touches in it) - 90 % of the issue rate - and the tile itself:
// 13.9 K cycles from request to arrival untouched, 8.8 K touched a tile ahead, 3.9 K touched half a tile ahead (shipped).
// What is left is that wait and the image load of a workgroup (~36 K cycles): 43.8 against 40 TFLOP/s in the probe,
// -3.5 % on the factorisations (7200 x 900 x 128: 64.3 -> 62.2 ms; 5400 x 900 x 256: 94.4 -> 91.9; 16384 x 4096: 27.7 -> 27.1;
// profiles/r04_qrbench3.txt).  Tried on top of it and measured (same probe, profiles/r04_cq_upd_probe.txt):
//  * two waves per SIMD (512 threads, <= 256 registers, tiles handed out through an LDS counter): 43 - 46.6 TFLOP/s at 256
//    workgroups and 49.9 with four rounds of workgroups (256 threads: 40.2), 58.6 without global traffic - but the fused
//    update + factor launch needs the 256-thread shape (the factor nodes use 496 registers), and unfused the factorisations
//    lose more than the update gains except at >= 128 problems (7200 x 900 x 128: 61.0 against 62.0 ms);
//  * two tile buffers in the 512 registers of a lone wave, the next tile requested as soon as the current one has arrived:
//    396 registers, no spill, but the compiler's wait in front of the first MFMA of every tile is vmcnt(0) - it waits for the
//    tile just requested as well; with straight-line loop bodies the count is right (vmcnt(32)) but the allocator keeps both
//    buffers in the architectural half, spills addresses, and every scratch reload is a vmcnt(0) again; pinned into the
//    accumulation half by "+a" constraints it copies them back and forth (1312 v_accvgpr_*): no gain in any form;
//  * the next tile's row groups requested inside phase C, each behind the store that frees its registers: 239 spills.
// uniform base + this lane's 32-bit byte offset + a constant: the shape the compiler turns into `global_* v_off, s[base:base+1] offset:imm`
__device__ __forceinline__ gd4* lane_ptr(gdbl* base, unsigned lane_bytes, int doubles) {
  typedef __attribute__((address_space(1))) char gchar;
  return reinterpret_cast<gd4*>(reinterpret_cast<gchar*>(base) + lane_bytes + 8 * doubles);
}
struct TileAddr {
  const ldbl* V;                // the node's image
  unsigned loff;                // this lane's place in a tile: column lane & 15, rows 4 (lane >> 4) (bytes; the segment pointers are uniform,
                                // so that global loads and stores take them from scalar registers: 16 address registers fewer)
  const ldbl* ops;              // operand images (uniform: the lane offset is added where phase B starts - kept in a register across the
                                // phases it was spilled, and a scratch reload waits for every outstanding global load as well)
};
// phase A (TR = false): [p & 1][row group >> 3][e]; phase C (TR = true): [p & 1][row group >> 3][s].  Computed at the start of each
// phase of each tile from two lane constants the optimiser cannot see through (else the sixteen addresses are hoisted out of the
// tile loop and spilled, or folded into > 16-bit immediates and re-added before every read).
template <bool TR>
__device__ __forceinline__ void phase_addr(const ldbl* (&P)[2][2][4], const ldbl* V) {
  int g = (threadIdx.x & 63) >> 4, sc = sig(threadIdx.x & 15);
  asm volatile("" : "+v"(g), "+v"(sc));
  const int zc = swz(sc);
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int m = 4 * g + e, z = TR ? zc : swz(m);
    const int lo = TR ? sc * 64 + (m ^ (z & 15)) : m * 64 + (sc ^ (z & 15)), hi = 16 * (z >> 4);
    P[0][0][e] = V + lo + hi; P[1][0][e] = V + lo - hi; P[0][1][e] = V + lo + hi + 8192; P[1][1][e] = V + lo - hi + 8192;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < 4; k++) asm volatile("" : "+v"(P[i][j][k]));
}
constexpr int groupA_next(bool tree, int k) {          // phase A walks (p, rb) = (k >> 4, k & 15); the next group not skipped (64: none)
  for (int q = k + 1; q < 64; q++) if (!tree_skip(tree, q & 15, q >> 4)) return q;
  return 64;
}
// seg[s]: the tile's column `lane & 15`, rows base[s] + 4 (lane >> 4) of segment s; segn: the same of the wave's next tile (touched)
template <bool TREE>
__device__ __forceinline__ void compute_tile(gdbl* const (&seg)[4], int nrb, d4 (&C)[16], const TileAddr& A, int cq_tr_tile = 0, const gdbl* Yt = nullptr, long ldt = 0,
                                               const int* baset = nullptr, int col_next = 0, unsigned touch_off = 0) {
  d4 w[4];
#ifdef CQ_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  CQ_TR(1);
#endif
  // ------------------------------------------------ phase A: W0_p = V_p^T C
  {
    const ldbl* pa[2][2][4];
    phase_addr<false>(pa, A.V);
    double a[2][4];
    int par = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) a[0][e] = pa[0][0][e][0];
    d4 acc = d4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 64; k++) {
      const int p = k >> 4, rb = k & 15;
      if (tree_skip(TREE, rb, p)) continue;
      const int nk = groupA_next(TREE, k), np = nk >> 4, nr = nk & 15;
      if (rb == 0) acc = d4{0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; e++) {          // one read of the NEXT group in front of every MFMA of this one
        if (nk < 64) a[par ^ 1][e] = pa[np & 1][nr >> 3][e][16 * np + 1024 * (nr & 7)];
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma(a[par][e], C[rb][e], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (groupA_next(TREE, k) >= 64 || (groupA_next(TREE, k) >> 4) != p) w[p] = acc;
      par ^= 1;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  CQ_TR(2);
  // ------------------------------------------------ phase B: W_p = T_p^T (W0_p - sum_{r<p} S_pr W_r)
  const ldbl* ops = A.ops + (threadIdx.x & 63);
#pragma unroll
  for (int p = 0; p < 4; p++) {
    d4 t = w[p];
#pragma unroll
    for (int r = 0; r < p; r++)
#pragma unroll
      for (int s = 0; s < 4; s++) t = mfma(ops[(4 + p * (p - 1) / 2 + r) * 256 + 64 * s], w[r][s], t);
    d4 o = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) o = mfma(ops[p * 256 + 64 * s], t[s], o);
    w[p] = o;
    __builtin_amdgcn_sched_barrier(0);
  }
  CQ_TR(3);
  // the wave's next tile is touched HERE: half a tile (~17 K cycles) before it is loaded.  Touched after the first quarter of
  // phase A (round 3) a tile waited ~8.8 K cycles for its data, touched here ~3.9 K: a CU's four waves stream 256 KB through
  // its share of the 4 MB L2 per tile time, and a line touched a whole tile early is gone again when it is wanted.
  if (col_next > 0) {
    const int (&bt)[4] = *reinterpret_cast<const int (*)[4]>(baset);
    touch_tile(Yt, ldt, bt, nrb, col_next, touch_off);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ------------------------------------------------ phase C: C -= sum_p V_p W_p
  {
    const ldbl* pc[2][2][4];
    phase_addr<true>(pc, A.V);
    double a[2][4];
    int par = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) a[0][s] = pc[0][0][s][0];
#pragma unroll
    for (int rb = 0; rb < 16; rb++) {
      d4 acc = C[rb];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        if (tree_skip(TREE, rb, p)) continue;
        const int nx = pair_next(TREE, rb, p), nr = nx >> 2, np = nx & 3;
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (nx < 64) a[par ^ 1][s] = pc[np & 1][nr >> 3][s][16 * np + 1024 * (nr & 7)];
          __builtin_amdgcn_sched_barrier(0);
          acc = mfma_na(a[par][s], w[p][s], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        par ^= 1;
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef CQ_NO_GLOBAL
      C[rb] = acc; if (acc[0] == 1.2345e301) __builtin_nontemporal_store(acc, lane_ptr(seg[rb >> 2], A.loff, 16 * (rb & 3)));
#else
      if (rb < nrb) __builtin_nontemporal_store(acc, lane_ptr(seg[rb >> 2], A.loff, 16 * (rb & 3)));
#endif
    }
  }
  CQ_TR(4);
}
template <int NT>
__device__ __forceinline__ void upd_body(const v2::QrProb& Pr, int64_t ws_off, int jb, int level, int slot, int node, int tg, int tpg,
                                           int tfirst, ldbl* lds) {
  const int cols16 = (Pr.cols + 15) & ~15;
  if (jb + 64 > Pr.kmax || cols16 <= jb + 64) return;
  const int ntl = (cols16 - jb - 64) >> 4;
  const int t0 = tfirst + tg * tpg;
  if (t0 >= ntl) return;
  const int rows32 = (Pr.rows + 31) & ~31;
  int base[4], cnt[4];
  if (!node_segments(rows32, jb, level, node, base, cnt)) return;
  const int nrb = (cnt[0] + cnt[1] + cnt[2] + cnt[3]) >> 4;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  constexpr int nwave = NT >> 6;
  {
    const gdbl* src = (const gdbl*)Pr.aux + ws_off + (long)slot * IMG_DOUBLES;
    for (int ch = wave; ch < IMG_DOUBLES / 128; ch += nwave)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ch * 128 + lane * 2),
                                       (__attribute__((address_space(3))) void*)(lds + ch * 128), 16, 0, 0);
  }
  const int t1 = min(t0 + tpg, ntl);
  int t = t0 + wave;
  gdbl* seg[4];
  {
    gdbl* cp = (gdbl*)Pr.Y + (long)(jb + 64 + 16 * t) * Pr.ld;
#pragma unroll
    for (int s = 0; s < 4; s++) seg[s] = cp + base[s];
  }
  const long tstep = 16L * nwave * Pr.ld;
  TileAddr A{lds, 8u * ((unsigned)(lane & 15) * (unsigned)Pr.ld + 4u * (unsigned)(lane >> 4)), lds + IMG_V};
  auto load = [&](d4 (&C)[16]) {
    if (nrb == 16) {                    // (a full node: no zero fill - 128 VALU moves per tile otherwise)
#pragma unroll
      for (int rb = 0; rb < 16; rb++) C[rb] = __builtin_nontemporal_load(lane_ptr(seg[rb >> 2], A.loff, 16 * (rb & 3)));
    } else {
#pragma unroll
      for (int rb = 0; rb < 16; rb++) {
        C[rb] = d4{0, 0, 0, 0};
        if (rb < nrb) C[rb] = __builtin_nontemporal_load(lane_ptr(seg[rb >> 2], A.loff, 16 * (rb & 3)));
      }
    }
  };
  auto compute = [&](d4 (&C)[16], gdbl* const (&sg)[4], int trt) {
    const int cnext = NT == 256 && t + nwave < t1 ? jb + 64 + 16 * (t + nwave) : 0;          // (touches only with a lone wave per SIMD)
    if (level == 0) compute_tile<false>(sg, nrb, C, A, trt, (const gdbl*)Pr.Y, Pr.ld, base, cnext, (unsigned)(IMG_DOUBLES * 8));
    else compute_tile<true>(sg, nrb, C, A, trt, (const gdbl*)Pr.Y, Pr.ld, base, cnext, (unsigned)(IMG_DOUBLES * 8));
  };
  if constexpr (NT == 256) {
    // One wave per SIMD: static assignment, the wave's next tile touched (touch_tile) while this one is updated.  (Tried: TWO
    // tile buffers in the 512 registers of a lone wave, the next tile requested as soon as the current one has arrived - 396
    // registers, no spill, but the compiler waits with vmcnt(0) in front of the first MFMA of every tile, i.e. for the tile
    // it has just requested as well: no gain.)
    d4 C[16];
    if (t < t1) load(C);
    __syncthreads();
    int trt = 0;
    for (bool first = true; t < t1; t += nwave, first = false) {
      CQ_TR0(trt);
      if (!first) {
#pragma unroll
        for (int s = 0; s < 4; s++) seg[s] += tstep;
        load(C);
      }
      compute(C, seg, trt++);
    }
  } else {
    // Two waves per SIMD: the tiles of the group are handed out through a counter in LDS.  (Statically assigned, the second wave
    // of every SIMD fell behind the first - the older wave wins the issue port - and ran its last tiles alone.)
    typedef __attribute__((address_space(3))) int lint;
    lint* ctr = reinterpret_cast<lint*>(lds + IMG_DOUBLES);
    if (tid == 0) *ctr = nwave;
    d4 C[16];
    if (t < t1) load(C);
    __syncthreads();
    int trt = 0;
    gdbl* seg0[4];
#pragma unroll
    for (int s = 0; s < 4; s++) seg0[s] = seg[s] - 16L * wave * Pr.ld;          // tile t0
    for (bool first = true; t < t1; first = false) {
      CQ_TR0(trt);
      if (!first) {
#pragma unroll
        for (int s = 0; s < 4; s++) seg[s] = seg0[s] + 16L * (t - t0) * Pr.ld;
#ifndef CQ_NO_GLOBAL
        load(C);
#endif
      }
      compute(C, seg, trt++);
      int k = 0;
      if (lane == 0) k = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      t = t0 + __builtin_amdgcn_readfirstlane(k);
    }
  }
}

// k_cq_upd: grid (tile groups, nodes of the level, problems), IMG_DOUBLES + TOUCH_DOUBLES doubles of dynamic LDS.
// Tiles [tfirst + tg * tpg, + tpg) of the columns right of the block.  The library launches the 256-thread build (one wave per
// SIMD); the 512-thread build (two waves per SIMD, tiles handed out through an LDS counter) is kept for tools/probes/cq_upd_probe.hip.
template <int NT>
__global__ void __launch_bounds__(NT) k_cq_upd(const v2::QrProb* probs, int64_t ws_off, int jb, int level, int slot0, int tpg, int tfirst) {
  extern __shared__ __attribute__((aligned(16))) double cq_lds_raw[];
  upd_body<NT>(probs[blockIdx.z], ws_off, jb, level, slot0 + blockIdx.y, blockIdx.y, blockIdx.x, tpg, tfirst, (ldbl*)cq_lds_raw);
}

// ------------------------------------------------------------------------------------------------------------------
// k_cq_updfac: the update of level `level` AND the factorisation of level `level + 1` in one launch (256 threads,
// L2_FAC_TOTAL doubles of dynamic LDS).  The two do not touch the same data (the factorisation works on the block's own
// 64 columns and only needs the R factors the previous launch left there, the update works on the columns to the
// right), and the few factor workgroups - first in the grid, so they start at once - run on CUs of their own beside
// the update's instead of holding the whole chip for 70 us per level.  Two streams do not give this: the hand-over
// between them costs 20-40 us per event (see qr_batch).
// grid.x = nprob * nfac + nprob * nupd * ntg  (nfac / nupd: nodes of level + 1 / level, ntg: tile groups)
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cq_updfac(const v2::QrProb* probs, int nprob, int64_t ws_off, int jb, int level, int slot_u, int nupd,
                                                   int ntg, int tpg, int slot_f, int nfac) {
  extern __shared__ __attribute__((aligned(16))) double cq_lds_raw[];
  ldbl* lds = (ldbl*)cq_lds_raw;
  int b = blockIdx.x;
  if (b < nprob * nfac) {
    const int node = b % nfac;
    const v2::QrProb Pr = probs[b / nfac];
    if (jb >= Pr.kmax) return;
    const int rows32 = (Pr.rows + 31) & ~31;
    int base[4], cnt[4];
    if (!node_segments(rows32, jb, level + 1, node, base, cnt)) return;
    fac2_body<true>(Pr, ws_off, jb, slot_f + node, base, cnt, 0, lds);
    return;
  }
  b -= nprob * nfac;
  const int tg = b % ntg, node = (b / ntg) % nupd;
  upd_body<256>(probs[b / (ntg * nupd)], ws_off, jb, level, slot_u + node, node, tg, tpg, 0, lds);
}

}  // namespace cq
