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
constexpr int tree_next(bool tree, int rb, int p) {          // the next row group after rb that sub-panel p touches (16: none)
  for (int r = rb + 1; r < 16; r++) if (!tree_skip(tree, r, p)) return r;
  return 16;
}
__device__ __forceinline__ void load_tile(const gdbl* Y, long ld, const int (&base)[4], int nrb, int col0, d4 (&C)[16]) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const gdbl* cp = Y + (long)(col0 + c) * ld + 4 * g;
#pragma unroll
  for (int rb = 0; rb < 16; rb++) {
    C[rb] = d4{0, 0, 0, 0};
    if (rb < nrb) C[rb] = __builtin_nontemporal_load(reinterpret_cast<const gd4*>(cp + base[rb >> 2] + 16 * (rb & 3)));
  }
}
// Touches the tile at column col_next: one 4-byte load per 128-byte line of the wave's 16 column segments (row group
// (lane >> 4) + 4 k, column lane & 15), so that the real load one tile later hits the L2 instead of waiting for HBM.  A wave
// of the update has no second wave on its SIMD to switch to, and a second tile buffer in registers spills (k_cq_upd).  The
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
#ifdef CQ_NO_LDS
// probe operand values: CQ_NO_LDS=1 a low-activity constant (1 + k 1e-9: mostly zero mantissa bits), CQ_NO_LDS=2 a full random mantissa
__device__ __forceinline__ double CQ_NO_LDS_VALUE(unsigned k) {
#if CQ_NO_LDS == 2
  unsigned long long h = (k + 1) * 0x9E3779B97F4A7C15ULL; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ULL; h ^= h >> 32;
  return __longlong_as_double((long long)((h >> 12) | 0x3FF0000000000000ULL)) - 1.5;
#else
  return 1.0 + 1e-9 * (double)k;
#endif
}
#endif
#ifdef CQ_UPROF
__device__ unsigned long long cq_uprof[8];
#define CQ_UT(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) atomicAdd(&cq_uprof[slot], t1_ - cq_ut0); cq_ut0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CQ_UT_DECL unsigned long long cq_ut0 = __builtin_readcyclecounter()
#else
#define CQ_UT(slot) do {} while (0)
#define CQ_UT_DECL do {} while (0)
#endif
// the tile in C (load_tile) against the node's image; the updated tile is stored
// NEXT: the wave has another tile (at column col_next) after this one: it is touched (touch_tile) once this one is in registers.
// (Tried in round 4 and measured no faster: the next tile's row groups requested in phase C, each right behind the store
// that frees its registers - with the touches the tile load is not what a wave waits for; without them the last row
// groups arrive late: 117 against 108 us for 64 tiles x 64 nodes, tools/probes/cq_upd_probe.hip.)
template <bool TREE, bool NEXT>
__device__ __forceinline__ void compute_tile(gdbl* Y, long ld, const int (&base)[4], int nrb, int col0, d4 (&C)[16], const ldbl* V, const ldbl* OPS,
                                             int col_next, unsigned touch_off) {
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  gdbl* cp = Y + (long)(col0 + c) * ld + 4 * g;
  // LDS offsets: p even / odd variants absorb the bit-4 part of the swizzle
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
#ifdef CQ_NO_LDS
#ifndef CQ_NCST
#define CQ_NCST 16
#endif
  double cst[CQ_NCST];                 // the stand-in operands: CQ_NCST registers (a power of two), filled once
#pragma unroll
  for (int k = 0; k < CQ_NCST; k++) cst[k] = CQ_NO_LDS_VALUE(threadIdx.x * 16 + k);
#endif
  CQ_UT_DECL;
#ifdef CQ_UPROF
  { double sink = 0; for (int rb = 0; rb < 16; rb++) sink += C[rb][0] + C[rb][3]; if (sink == 1.2345e301) cp[0] = sink; }   // wait for the tile here
  CQ_UT(1);
#endif
  // ------------------------------------------------ phase A: W0_p = V_p^T C
  // The LDS operands are fetched one row group ahead by hand and the schedule is pinned per group: left alone, the
  // scheduler hoists dozens of ds_reads above the MFMA chain and spills the C tile.
  d4 w[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    d4 acc = d4{0, 0, 0, 0};
    // two base pointers per operand: the row-group offset (8 KB per group) then fits the 16-bit ds_read immediate;
    // with one base the compiler materialises an address register per (group, e) - 204 of them, spilled
    const ldbl* Vlo[4];
    const ldbl* Vhi[4];
#pragma unroll
    for (int e = 0; e < 4; e++) { Vlo[e] = V + ((p & 1) ? aO[e] : aE[e]); Vhi[e] = Vlo[e] + 8192; }
    double a[4], an[4];
#pragma unroll
#ifdef CQ_NO_LDS          // probe builds: the reflector operands are constants in registers (no LDS traffic at all)
    for (int e = 0; e < 4; e++) a[e] = cst[(e + 4 * p) & (CQ_NCST - 1)];
#else
    for (int e = 0; e < 4; e++) a[e] = Vlo[e][16 * p];                       // row group 0 is never skipped
#endif
#pragma unroll
    for (int rb = 0; rb < 16; rb++) {
      if (tree_skip(TREE, rb, p)) continue;
      const int nx = tree_next(TREE, rb, p);
      if (nx < 16) {
#pragma unroll
#ifdef CQ_NO_LDS
        for (int e = 0; e < 4; e++) an[e] = a[e];
#else
        for (int e = 0; e < 4; e++) an[e] = (nx < 8) ? Vlo[e][16 * p + 1024 * nx] : Vhi[e][16 * p + 1024 * (nx - 8)];
#endif
      }
#pragma unroll
      for (int e = 0; e < 4; e++) acc = mfma(a[e], C[rb][e], acc);
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] = an[e];
      __builtin_amdgcn_sched_barrier(0);
    }
    w[p] = acc;
    if (NEXT && p == 0) {                       // the tile is in registers by now (the first MFMAs waited for it)
      touch_tile(Y, ld, base, nrb, col_next, touch_off);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  CQ_UT(2);
  // ------------------------------------------------ phase B: W_p = T_p^T (W0_p - sum_{r<p} S_pr W_r)
#pragma unroll
  for (int p = 0; p < 4; p++) {
    d4 t = w[p];
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
  }
#pragma unroll
  for (int p = 0; p < 4; p++) w[p] = -w[p];
  CQ_UT(3);
  // ------------------------------------------------ phase C: C -= sum_p V_p W_p
  // (the LDS operands of row group rb + 1 are fetched while the MFMAs of row group rb run, as in phase A: fetched right in
  //  front of their MFMAs, every (row group, sub-panel) waited ~100 cycles for its four reads)
  double a[4][4], an[4][4];
  // all sixteen base pointers up front (the image is 128 KB, a ds_read immediate reaches 64 KB: even / odd sub-panel x
  // lower / upper half) - computed inside the loop they were 256 address adds per tile between the MFMAs
  const ldbl* cb[2][2][4];
#pragma unroll
  for (int s = 0; s < 4; s++) { cb[0][0][s] = V + cE[s]; cb[0][1][s] = V + cE[s] + 8192; cb[1][0][s] = V + cO[s]; cb[1][1][s] = V + cO[s] + 8192; }
  auto fetch = [&](int rb, double (&dst)[4][4]) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (tree_skip(TREE, rb, p)) continue;
#pragma unroll
#ifdef CQ_NO_LDS
      for (int s = 0; s < 4; s++) dst[p][s] = cst[(4 * p + s + 5 * rb) & (CQ_NCST - 1)];
#else
      for (int s = 0; s < 4; s++) dst[p][s] = cb[p & 1][rb >> 3][s][16 * p + 1024 * (rb & 7)];
#endif
    }
  };
  fetch(0, a);
#pragma unroll
  for (int rb = 0; rb < 16; rb++) {
    if (rb + 1 < 16) fetch(rb + 1, an);
    d4 acc = C[rb];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (tree_skip(TREE, rb, p)) continue;
#pragma unroll
      for (int s = 0; s < 4; s++) acc = mfma(a[p][s], w[p][s], acc);
    }
#ifdef CQ_NO_GLOBAL            // probe builds: the tile never leaves the registers (compute-only ceiling of the wave's instruction stream)
    C[rb] = acc;
    if (acc[0] == 1.2345e301 && rb < nrb) __builtin_nontemporal_store(acc, reinterpret_cast<gd4*>(cp + base[rb >> 2] + 16 * (rb & 3)));
#else
    if (rb < nrb) __builtin_nontemporal_store(acc, reinterpret_cast<gd4*>(cp + base[rb >> 2] + 16 * (rb & 3)));
#endif
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int s = 0; s < 4; s++) a[p][s] = an[p][s];
    __builtin_amdgcn_sched_barrier(0);
  }
  CQ_UT(4);
}

// ------------------------------------------------------------------------------------------------------------------
// k_cq_upd: grid (tile groups, nodes of the level, problems), 512 or 256 threads, IMG_DOUBLES doubles of dynamic LDS.
// Tiles [tfirst + tg * tpg, + tpg) of the columns right of the block, one per wave at a time.
// (Tried: four-wave workgroups that fetch the next tile while the current one is updated, two tile buffers in the 512
// registers of a lone wave - the allocator spills 57-206 registers around the two buffers and the kernel is 8 % slower.
// Round 4, tools/probes/cq_upd_probe.hip, 128 tiles x 64 nodes, all on one box: (i) TWO tiles per wave against the same LDS
// operands - every ds_read feeds two MFMAs, 256 + 256 registers, no spill - 37-38 against 38-40 TFLOP/s: the LDS operand
// stream is not what a wave waits for; (ii) the 256 LDS address adds of phase C hoisted (kept): no change - nor is VALU
// issue; (iii) no global traffic at all (-DCQ_NO_GLOBAL): 44 TFLOP/s = 23 us per tile and wave against 14.7 us of MFMA issue
// at 2.4 GHz, 20.8 us with half the CUs busy; the shader clock under this load is 2.26 GHz (s_memtime against s_memrealtime,
// -DCQ_UPROF), so a wave spends ~84 cycles per 64-cycle MFMA whatever stands between them; (iv) the next row group's LDS
// reads forced right behind the first MFMA of a group (sched_group_barrier; the compiler issues them after the last one, 64
// cycles before their use): same wait counts as hand analysis (lgkmcnt 6 / 5 / 4), no change; (v) eight-wave workgroups (two
// waves per SIMD, 256 registers, spills): 30-33 TFLOP/s; (vi) -DCQ_NO_GLOBAL -DCQ_NO_LDS: the same MFMA stream with its A
// operands in registers - no LDS, no global traffic, no waits.  With SIXTEEN operand registers cycling it reaches 49 (random
// mantissas) - 52 (1 + k 1e-9) TFLOP/s; with FOUR loop-invariant ones 74 - 76 (the data-sheet peak).  So the ceiling of this
// stream on this part is set by the variety of the MFMA's source registers (operand fetch from the register file), not by
// LDS, memory, waits or the clock, and the shipped kernel (40 with memory, 45 without) runs at 80 - 90 % of it.  (By number of
// distinct A registers: 2 or 4 -> 73.6, 8 -> 63.8, 16 or 32 -> 49.5 TFLOP/s.  Fewer LDS-FED registers do not help: phase C with
// its operands fetched one sub-panel ahead - 8 instead of 32 registers - 40.2 against 39.5 TFLOP/s.))
// ------------------------------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void upd_body(const v2::QrProb& Pr, int64_t ws_off, int jb, int level, int slot, int node, int tg, int tpg,
                                         int tfirst, ldbl* lds, unsigned touch_off) {
  const int cols16 = (Pr.cols + 15) & ~15;
  if (jb + 64 > Pr.kmax || cols16 <= jb + 64) return;
  const int ntl = (cols16 - jb - 64) >> 4;
  const int t0 = tfirst + tg * tpg;
  if (t0 >= ntl) return;
  const int rows32 = (Pr.rows + 31) & ~31;
  int base[4], cnt[4];
  if (!node_segments(rows32, jb, level, node, base, cnt)) return;
  const int nrb = (cnt[0] + cnt[1] + cnt[2] + cnt[3]) >> 4;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int nwave = NT >> 6;
#ifdef CQ_UPROF
  const unsigned long long cq_wg_t0 = __builtin_readcyclecounter();
  const unsigned long long cq_wg_r0 = __builtin_amdgcn_s_memrealtime();
  struct ClockNote { unsigned long long c0, r0; bool on; __device__ ~ClockNote() { if (on) { atomicAdd(&cq_uprof[5], __builtin_readcyclecounter() - c0); atomicAdd(&cq_uprof[7], __builtin_amdgcn_s_memrealtime() - r0); } } };
  ClockNote cq_note{cq_wg_t0, cq_wg_r0, threadIdx.x == 0};          // [5] shader-clock cycles, [7] 100 MHz ticks of the workgroup
#endif
  // The node's image (148 KB) -> LDS by LDS-DMA: 148 wave-instructions of 1 KiB (global_load_lds_dwordx4, no register
  // staging), all in flight at once, and the wave's FIRST tile is requested behind them before anybody waits.  (As a plain
  // copy loop the compiler serialised it - two 16-byte loads, wait, two LDS writes, 18 round trips per thread: ~12 us per
  // workgroup, as long as two tiles, during which the matrix pipes idle.)
  {
    const gdbl* src = (const gdbl*)Pr.aux + ws_off + (long)slot * IMG_DOUBLES;
    const int lane = tid & 63;
    static_assert(IMG_DOUBLES % 128 == 0, "the image is a whole number of 1 KiB pieces");
    for (int ch = wave; ch < IMG_DOUBLES / 128; ch += nwave)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ch * 128 + lane * 2),
                                       (__attribute__((address_space(3))) void*)(lds + ch * 128), 16, 0, 0);
  }
  const int t1 = min(t0 + tpg, ntl);
  gdbl* Y = (gdbl*)Pr.Y;
  d4 C[16];
  int t = t0 + wave;
  if (t < t1) load_tile(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C);
  __syncthreads();
#ifdef CQ_UPROF
  if ((tid & 63) == 0) { atomicAdd(&cq_uprof[0], __builtin_readcyclecounter() - cq_wg_t0); atomicAdd(&cq_uprof[6], 1ull); }
#endif
  if (level == 0) {
    for (bool first = true; t < t1; t += nwave, first = false) {
#ifndef CQ_NO_GLOBAL
      if (!first) load_tile(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C);
#endif
      if (t + nwave < t1) compute_tile<false, true>(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C, lds, lds + IMG_V, jb + 64 + 16 * (t + nwave), touch_off);
      else compute_tile<false, false>(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C, lds, lds + IMG_V, 0, touch_off);
    }
  } else {
    for (bool first = true; t < t1; t += nwave, first = false) {
      if (!first) load_tile(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C);
      if (t + nwave < t1) compute_tile<true, true>(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C, lds, lds + IMG_V, jb + 64 + 16 * (t + nwave), touch_off);
      else compute_tile<true, false>(Y, Pr.ld, base, nrb, jb + 64 + 16 * t, C, lds, lds + IMG_V, 0, touch_off);
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(NT) k_cq_upd(const v2::QrProb* probs, int64_t ws_off, int jb, int level, int slot0, int tpg, int tfirst) {
  extern __shared__ __attribute__((aligned(16))) double cq_lds_raw[];
  upd_body<NT>(probs[blockIdx.z], ws_off, jb, level, slot0 + blockIdx.y, blockIdx.y, blockIdx.x, tpg, tfirst, (ldbl*)cq_lds_raw, (unsigned)(IMG_DOUBLES * 8));
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
  upd_body<256>(probs[b / (ntg * nupd)], ws_off, jb, level, slot_u + node, node, tg, tpg, 0, lds, (unsigned)(IMG_DOUBLES * 8));
}

}  // namespace cq
