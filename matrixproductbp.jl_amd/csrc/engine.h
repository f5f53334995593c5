// The compress engine: one workgroup turns (train1 (x) train2, coupling table) into the truncated,
// each-matrix-normalised output train, i.e. one call of `op` in compute_prob_ys
// (reference src/recursive_bp_factor.jl:118-131: Kronecker build :120-123 + compress! :127 +
// normalize_eachmatrix! :128), or - with `mirror` - the `mpem2 |> compress!(is_orthogonal=:left) |>
// normalize_eachmatrix!` chain of onebpiter! (src/recursive_bp_factor.jl:156-157, src/mpems.jl:67-94).
//
// Algorithm (function-equivalent to the reference's two SVD sweeps, validated against the numpy oracle to
// 1e-14):
//   sweep 1 (no truncation; the reference's orthogonalize_right!(TruncThresh(0.0)) is a pure gauge change):
//       keep only the triangular factors  Lf_t  with  Lf_t Lf_t^T = X_t X_t^T,  X_t = A_t (I (x) Lf_{t+1}),
//       by an R-only blocked Householder QR of  Y_t = X_t^T.  Y_t is built from the two factor trains
//       through two small-matrix x streamed-tensor GEMMs - the a*b x a*b x p product core never exists.
//   sweep 2 (truncating; the reference's orthogonalize_left!(svd_trunc)):
//       N_t = C_{t-1} A_t (structured), M_t = N_t Lf_{t+1}; left singular vectors of M_t by QR(M_t^T)
//       + one-sided Jacobi; SVDTrunc rule on the singular values; new core = U, carry C_t = U^T N_t.
// Cores keep the ORIGINAL bond basis on the right, so Q is never formed or stored.
#if !defined(WG_THREADS)
#error "engine.h is included from kernels.h, once per workgroup-size variant"
#endif

namespace eng {

using namespace wg;

__device__ __forceinline__ int r16(int x) { return (x + 15) & ~15; }
__device__ __forceinline__ int r32(int x) { return (x + 31) & ~31; }

typedef __attribute__((address_space(1))) int32_t gint;
typedef __attribute__((address_space(3))) int lint;

// stage logical core t of a train into dst[m + a*(n + an*(y + ny*xi))]
template <class DP>
__device__ inline void stage_core(DP dst, const gdbl* base, const gint* bond, int64_t stride, int L,
                                  int t, bool mirror, int nyq) {
  const int tp = mirror ? (L - 1 - t) : t;
  const int pl = bond[tp], pr = bond[tp + 1];
  const int a = mirror ? pr : pl, an = mirror ? pl : pr;
  const gdbl* src = base + (int64_t)tp * stride;
  const int tot = a * an * nyq;
  for (int idx = threadIdx.x; idx < tot; idx += WG_THREADS) {
    int m = idx % a; int rest = idx / a; int n = rest % an; int s = rest / an;
    int mp = mirror ? n : m, np_ = mirror ? m : n;
    dst[idx] = src[mp + pl * (np_ + pr * s)];
  }
}

// E_xi[(m2 + b*y) + M2*(n2 + bn*y1)] = sum_y2 pyy[y,y1,y2,xi] * A2c[m2,n2,y2,xi]
template <class DP>
__device__ inline void build_E(DP E, DP A2c, const gdbl* pyy, int b, int bn, int ny, int ny1,
                               int ny2, int q) {
  const int M2 = b * ny, K2 = bn * ny1;
  const int tot = M2 * K2 * q;
  for (int idx = threadIdx.x; idx < tot; idx += WG_THREADS) {
    int i = idx % M2; int rest = idx / M2; int kk = rest % K2; int xi = rest / K2;
    int m2 = i % b, y = i / b, n2 = kk % bn, y1 = kk / bn;
    double s = 0.0;
    for (int y2 = 0; y2 < ny2; y2++) {
      double c = pyy[y + ny * (y1 + ny1 * (y2 + ny2 * xi))];
      if (c != 0.0) s += c * A2c[m2 + b * (n2 + bn * (y2 + ny2 * xi))];
    }
    E[idx] = s;
  }
}

__device__ inline void atomic_max_double_pos(unsigned long long* addr, double v) {
  atomicMax(addr, (unsigned long long)__double_as_longlong(v));
}

// CORES_LDS / JAC_LDS: whether the staged cores + coupling table / the Jacobi matrix live in LDS or in the slot's
// HBM scratch (decided per launch by the host, plan_cfg); compile-time so that every access has a known address space.
template <bool CORES_LDS, bool JAC_LDS, bool EXT>
__device__ void run_problem(const EngProb& P, const EngCfg& cfg, double* slot_, double* lds_, EngStats* stats) {
  typedef typename std::conditional<CORES_LDS, ldbl*, gdbl*>::type CP;
  typedef typename std::conditional<JAC_LDS, ldbl*, gdbl*>::type JP;
  const int tid = threadIdx.x;
  const int L = cfg.L;
  const bool mirror = P.mirror != 0;
  const int ny1 = P.ny1, ny2 = P.ny2, ny = P.ny, q = P.q;
  gdbl* slot = (gdbl*)slot_;
  ldbl* lds = (ldbl*)lds_;
  ldbl* ldsG = lds + cfg.lds_gemm;
  ldbl* ldsQ = lds + cfg.lds_qr;
  ldbl* misc = lds + cfg.lds_misc;              // [32 + 2*nmax]: reductions, sigma, order
  CP A1c = CORES_LDS ? (CP)(lds + cfg.lds_A1c) : (CP)(slot + cfg.off_A1c);
  CP A2c = CORES_LDS ? (CP)(lds + cfg.lds_A2c) : (CP)(slot + cfg.off_A2c);
  CP E = CORES_LDS ? (CP)(lds + cfg.lds_E) : (CP)(slot + cfg.off_E);
  JP JA = JAC_LDS ? (JP)(lds + cfg.lds_JA) : (JP)(slot + cfg.off_JA);
  lint* rdim = (lint*)(lds + cfg.lds_rdim);     // [L+1]
  ldbl* red = misc;                          // 32 doubles
  ldbl* sig = misc + 32;                     // [nmax]
  lint* ord = (lint*)(misc + 32 + cfg.nmax);   // [nmax]
  gdbl* LfS = slot + cfg.off_Lf;
  gdbl* Z = slot + cfg.off_Z;
  gdbl* Y = slot + cfg.off_Y;
  gdbl* T1 = slot + cfg.off_T1;
  gdbl* Nt = slot + cfg.off_Nt;
  gdbl* Mt = slot + cfg.off_Mt;
  gdbl* Ccur = slot + cfg.off_C0;
  gdbl* Cnew = slot + cfg.off_C1;
  const gdbl* PA1 = (const gdbl*)P.A1; const gdbl* PA2 = (const gdbl*)P.A2;
  const gint* Pb1 = (const gint*)P.bond1; const gint* Pb2 = (const gint*)P.bond2;
  const gdbl* Ppyy = (const gdbl*)P.pyy;
  gdbl* Pout = (gdbl*)P.out; gint* Pob = (gint*)P.obond;

  auto LB1 = [&](int t) { return mirror ? Pb1[L - t] : Pb1[t]; };
  auto LB2 = [&](int t) { return mirror ? Pb2[L - t] : Pb2[t]; };
  auto TP = [&](int t) { return mirror ? (L - 1 - t) : t; };
  // triangular factors: own sweep 1 (slot stack, ranks in LDS) or the batched gauge sweep's (P.lf != null)
  constexpr bool ext = EXT;        // compile time: the in-workgroup sweep 1 keeps its code generation
  const gdbl* Plf = (const gdbl*)P.lf;
  const __attribute__((address_space(1))) int64_t* Plfoff = (const __attribute__((address_space(1))) int64_t*)P.lfoff;
  const gint* Prd = (const gint*)P.rdim;
  auto RD = [&](int t) -> int { return ext ? (int)Prd[t] : (int)rdim[t]; };
  auto LFP = [&](int t) -> const gdbl* { return ext ? Plf + Plfoff[t] : (const gdbl*)(LfS + (int64_t)t * cfg.lf_stride); };
  Prof* pr = cfg.prof;
  unsigned long long plast = pr ? wall_clock64() : 0ULL;
#define PROF(ph) prof_mark(pr, plast, ph)

  // ------------------------------------------------------------------ sweep 1: triangular factors
  if (tid == 0 && !ext) { rdim[L] = 1; LfS[(int64_t)L * cfg.lf_stride] = 1.0; }
  __syncthreads();
  for (int t = ext ? 0 : L - 1; t >= 1; t--) {
    const int a = LB1(t), an = LB1(t + 1), b = LB2(t), bn = LB2(t + 1);
    const int Bm = a * b, Bn = an * bn;
    const int r1 = rdim[t + 1];
    const gdbl* Lf1 = LfS + (int64_t)(t + 1) * cfg.lf_stride;      // Lf^T: [r1 x Bn], ld r1 (rank index fastest)
    stage_core(A1c, PA1, Pb1, P.stride1, L, t, mirror, ny1 * q);
    stage_core(A2c, PA2, Pb2, P.stride2, L, t, mirror, ny2 * q);
    __syncthreads();
    build_E(E, A2c, Ppyy + (int64_t)TP(t) * P.pyy_tstride, b, bn, ny, ny1, ny2, q);
    __syncthreads();
    PROF(PH_STAGE);
    // Every tile index below has the rank index k fastest, so loads and stores are contiguous along k.
    // Y1: Z[(k,n2) ; (m1,y1,xi)] = sum_n1 A1[m1,n1,y1,xi] Lf[(n1,n2),k]      Z[j + r1*bn*i], j = k + r1*n2
    const int M1 = a * ny1 * q;
    const int64_t zld = (int64_t)r1 * bn;
    gemm_direct<false, true>(M1, r1 * bn, an, A1c,          // Z is written once and read once, 5 MB away: nontemporal
         [=](int i) { return (i % a) + a * an * (i / a); }, [=](int kk) { return a * kk; },
         Lf1, [=](int kk) { return (int64_t)r1 * kk; },
         [=](int j) { return (int64_t)(j % r1) + (int64_t)r1 * an * (j / r1); },
         Z, [=](int i) { return zld * i; }, [=](int j) { return (int64_t)j; }, false);
    PROF(PH_Y1);
    // Y2 (per xi): Y[(k,y,xi) ; (m1,m2)] = sum_(n2,y1) E_xi[(m2,y),(n2,y1)] Z[(k,n2),(m1,y1,xi)]
    const int rowsY = r1 * ny * q;
    const int ldY = r32(rowsY);
    const int cols16 = r16(Bm) + 16;
    // zero padding rows / columns
    for (int64_t idx = tid; idx < (int64_t)(ldY - rowsY) * cols16; idx += WG_THREADS) {
      int rr = rowsY + (int)(idx % (ldY - rowsY)); int64_t c = idx / (ldY - rowsY);
      Y[rr + (int64_t)ldY * c] = 0.0;
    }
    for (int64_t idx = tid; idx < (int64_t)ldY * (cols16 - Bm); idx += WG_THREADS)
      Y[(int64_t)ldY * Bm + idx] = 0.0;
    const int M2 = b * ny, K2 = bn * ny1;
    for (int xi = 0; xi < q; xi++) {
      gemm_direct<true, true>(M2, r1 * a, K2, E + (int64_t)xi * M2 * K2,
           [=](int i) { return i; }, [=](int kk) { return M2 * kk; },
           Z + zld * a * ny1 * xi,
           [=](int kk) { return (int64_t)r1 * (kk % bn) + zld * a * (kk / bn); },
           [=](int j) { return (int64_t)(j % r1) + zld * (j / r1); },
           Y + (int64_t)r1 * ny * xi,
           [=](int i) { return (int64_t)r1 * (i / b) + (int64_t)ldY * a * (i % b); },
           [=](int j) { return (int64_t)(j % r1) + (int64_t)ldY * (j / r1); }, false);
    }
    PROF(PH_Y2);
    qr_r(Y, ldY, rowsY, Bm, ldsQ, ldsG, pr, &plast, PH_QR1_PANEL, PH_QR1_TRAIL, cfg.force_generic != 0);
    const int kmax = min(rowsY, Bm);
    // scale = max |R| over the upper trapezoid; Lf^T = R / scale  (column loops: no integer division)
    const int lane_ = tid & 63, wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Four columns per wave and step, their loads issued together (a wave walking one column at a time ran this phase - 1.3 MB read
    // twice, 1.3 MB written - at a dependent round trip per 64 elements: 4.5 % of a configs[1] sweep).  Indices past a column's end
    // are clamped instead of branched on: a repeated element does not change a maximum, and the copy selects the VALUE, not the load.
    constexpr int LFC = 4;
    double mx = 0.0;
    for (int m0 = LFC * wave_; m0 < Bm; m0 += LFC * WG_WAVES) {
      const int kcap = min(kmax, m0 + LFC);
      for (int k = lane_; k < kcap; k += 64) {
        double v[LFC];
#pragma unroll
        for (int c = 0; c < LFC; c++) {
          const int m = min(m0 + c, Bm - 1);
          v[c] = Y[(int64_t)ldY * m + min(k, min(kmax, m + 1) - 1)];
        }
#pragma unroll
        for (int c = 0; c < LFC; c++) mx = fmax(mx, fabs(v[c]));
      }
    }
    mx = wg_max(mx, red);
    const double inv = (mx > 0.0 && isfinite(mx)) ? 1.0 / mx : 1.0;
    gdbl* Lf0 = LfS + (int64_t)t * cfg.lf_stride;                   // Lf^T = R: [kmax x Bm], ld kmax
    for (int m0 = LFC * wave_; m0 < Bm; m0 += LFC * WG_WAVES) {
      for (int k = lane_; k < kmax; k += 64) {
        double v[LFC];
#pragma unroll
        for (int c = 0; c < LFC; c++) {
          const int m = min(m0 + c, Bm - 1);
          v[c] = Y[(int64_t)ldY * m + min(k, min(kmax, m + 1) - 1)];
        }
#pragma unroll
        for (int c = 0; c < LFC; c++) {
          const int m = m0 + c;
          if (m < Bm) Lf0[(int64_t)kmax * m + k] = (k < min(kmax, m + 1)) ? v[c] * inv : 0.0;          // wave-uniform branch
        }
      }
    }
    if (tid == 0) rdim[t] = kmax;
    __syncthreads();
    PROF(PH_LF);
  }

  // ------------------------------------------------------------------ sweep 2: truncation
  double logc = 0.0;
  int kc = 1;
  if (tid == 0) Ccur[0] = 1.0;
  if (tid == 0) { Pob[mirror ? L : 0] = 1; Pob[mirror ? 0 : L] = 1; }
  __syncthreads();
  for (int t = 0; t < L; t++) {
    const int a = LB1(t), an = LB1(t + 1), b = LB2(t), bn = LB2(t + 1);
    const int Bn = an * bn;
    const int tp = TP(t);
    stage_core(A1c, PA1, Pb1, P.stride1, L, t, mirror, ny1 * q);
    stage_core(A2c, PA2, Pb2, P.stride2, L, t, mirror, ny2 * q);
    __syncthreads();
    build_E(E, A2c, Ppyy + (int64_t)tp * P.pyy_tstride, b, bn, ny, ny1, ny2, q);
    __syncthreads();
    PROF(PH_STAGE);
    // N1: T1[(k,m2) ; (n1,y1,xi)] = sum_m1 A1[m1,n1,y1,xi] C[k,(m1,m2)]        T1[j + kc*b*i], j = k + kc*m2
    const int MT1 = an * ny1 * q;
    const int64_t tld = (int64_t)kc * b;
    gemm_direct(MT1, kc * b, a, A1c,
         [=](int i) { return a * (i % an) + a * an * (i / an); }, [=](int kk) { return kk; },
         Ccur, [=](int kk) { return (int64_t)kc * kk; },
         [=](int j) { return (int64_t)(j % kc) + (int64_t)kc * a * (j / kc); },
         T1, [=](int i) { return tld * i; }, [=](int j) { return (int64_t)j; }, false);
    // N2 (per xi): Nt[(k,y,xi) ; (n1,n2)] = sum_(m2,y1) E_xi[(m2,y),(n2,y1)] T1[(k,m2),(n1,y1,xi)]
    const int Rr = kc * ny * q;
    const int M2 = b * ny, K2 = bn * ny1;
    for (int xi = 0; xi < q; xi++) {
      gemm_direct(bn * ny, kc * an, b * ny1, E + (int64_t)xi * M2 * K2,
           [=](int i) { return b * (i / bn) + M2 * (i % bn); },
           [=](int kk) { return (kk % b) + M2 * bn * (kk / b); },
           T1 + tld * an * ny1 * xi,
           [=](int kk) { return (int64_t)kc * (kk % b) + tld * an * (kk / b); },
           [=](int j) { return (int64_t)(j % kc) + tld * (j / kc); },
           Nt + (int64_t)kc * ny * xi,
           [=](int i) { return (int64_t)kc * (i / bn) + (int64_t)Rr * an * (i % bn); },
           [=](int j) { return (int64_t)(j % kc) + (int64_t)Rr * (j / kc); }, false);
    }
    // rescale by max-abs (the reference rescales M at every step into z)
    {
      double mx = wg_maxabs(Nt, Rr, Rr, Bn, red);
      if (!(mx == mx) || isinf(mx)) { if (tid == 0) stats->nan_flag = 1; }
      if (mx > 0.0 && isfinite(mx)) {
        const double inv = 1.0 / mx;
        for (int64_t idx = tid; idx < (int64_t)Rr * Bn; idx += WG_THREADS) Nt[idx] *= inv;
        logc += log(mx);
      }
      __syncthreads();
    }
    PROF(PH_N);
    gdbl* oc = Pout + (int64_t)tp * P.ostride;
    if (t == L - 1) {
      // last core: [kc, 1, s] = Nt[(k,s), 0]
      for (int idx = tid; idx < Rr; idx += WG_THREADS) {
        int k = idx % kc, s = idx / kc;
        // logical (m=k, n=0): physical offset  !mirror: k + kc*(0 + 1*s);  mirror: 0 + 1*(k + kc*s)
        oc[k + kc * s] = Nt[idx];
      }
      __syncthreads();
      break;
    }
    const int r1 = RD(t + 1);
    const gdbl* Lf1 = LFP(t + 1);
    // Mt^T [r1 x Rr] = Lf1^T Nt^T
    const int ldM = r32(r1);
    const int Rr16 = r16(Rr) + 16;
    for (int64_t idx = tid; idx < (int64_t)(ldM - r1) * Rr16; idx += WG_THREADS) {
      int rr = r1 + (int)(idx % (ldM - r1)); int64_t c = idx / (ldM - r1);
      Mt[rr + (int64_t)ldM * c] = 0.0;
    }
    for (int64_t idx = tid; idx < (int64_t)ldM * (Rr16 - Rr); idx += WG_THREADS) Mt[(int64_t)ldM * Rr + idx] = 0.0;
    // (operand roles: the Rr <= 96 rows of N_t are the M side - ONE block of row tiles - and the r1 columns of Lf1 the N side, so that
    //  all eight waves own output tiles: 16 columns each per 128-column chunk.  The other way round - r1 as M, round 1 - 3 - only
    //  ceil(Rr / 16) = 5 of the 8 waves of configs[1] had a tile, over five M blocks.)
    gemm(Rr, r1, Bn, Nt, [=](int i) { return (int64_t)i; }, [=](int kk) { return (int64_t)Rr * kk; },
         Lf1, [=](int kk) { return (int64_t)r1 * kk; }, [=](int j) { return j; }, false,
         Mt, [=](int i) { return (int64_t)ldM * i; }, [=](int j) { return j; }, false, ldsG);
    PROF(PH_MT);
    qr_r(Mt, ldM, r1, Rr, ldsQ, ldsG, pr, &plast, PH_QR2_PANEL, PH_QR2_TRAIL, cfg.force_generic != 0);
    const int k2 = min(r1, Rr);
    // Left singular vectors of M_t = right singular vectors of R2.  Hestenes on the columns of
    // JA = R2^T [Rr x k2] (the "L form": about half the sweeps of the R form) WITHOUT accumulating V:
    // the rotated columns are sigma_j u_j.  Columns below 1e-14 ||.||_F are numerically null (they only
    // multiply what the right environment annihilates) and become zero columns of U.
    const int ldJ = JAC_LDS ? (Rr | 1) : ((Rr + 15) & ~15);      // in HBM: columns on 128-byte lines (wg::jac_pair_hbm16)
    double fro2 = 0.0;
    for (int idx = tid; idx < k2 * Rr; idx += WG_THREADS) {
      int r = idx % Rr, c = idx / Rr;            // JA[r, c] = R2[c, r]
      double v = (r >= c) ? Mt[c + (int64_t)ldM * r] : 0.0;
      JA[r + ldJ * c] = v;
      fro2 += v * v;
    }
    fro2 = wg_sum(fro2, red);
    __syncthreads();
    const int sw = jacobi_rsv(JA, ldJ, Rr, k2, nullptr, 0, red, ord, 60);
    if (tid == 0) {
      if (sw < 0) stats->jacobi_fail = 1;
      atomicAdd(&stats->jac_sweeps, (unsigned long long)(sw < 0 ? 60 : sw));
      atomicAdd(&stats->jac_calls, 1ULL);
    }
    PROF(PH_JAC);
    // singular values = column norms; normalise the columns (null ones -> 0); order descending
    const double nul2 = 1e-28 * fro2;
    for (int c = tid; c < Rr; c += WG_THREADS) sig[c] = 0.0;
    __syncthreads();
    for (int c = tid; c < k2; c += WG_THREADS) {
      double s = 0.0;
      for (int r = 0; r < Rr; r++) { double v = JA[r + ldJ * c]; s += v * v; }
      const bool ok = s > nul2 && s > 0.0;
      const double sg = ok ? sqrt(s) : 0.0, inv = ok ? 1.0 / sqrt(s) : 0.0;
      for (int r = 0; r < Rr; r++) JA[r + ldJ * c] *= inv;
      sig[c] = sg;
    }
    __syncthreads();
    for (int c = tid; c < k2; c += WG_THREADS) {
      const double sc = sig[c];
      int rank = 0;
      for (int j = 0; j < k2; j++) { double sj = sig[j]; rank += (sj > sc) || (sj == sc && j < c); }
      ord[rank] = c;
    }
    __syncthreads();
    // truncation rule (TensorTrains SVDTrunc functors)
    const int len = min(Rr, r1);
    int kp;
    {
      double tot = 0.0;
      for (int j = 0; j < len; j++) { double s = sig[ord[j]]; tot += s * s; }
      const mpbp_trunc tr = cfg.trunc;
      int kth = len;
      if (tr.kind == MPBP_TRUNC_THRESH || tr.kind == MPBP_TRUNC_BOND_THRESH) {
        const double thr = tr.eps * sqrt(tot);
        kth = 0;
        for (int j = 0; j < len; j++) if (sig[ord[j]] > thr) kth = j + 1;
        if (kth < 1) kth = 1;
      }
      if (tr.kind == MPBP_TRUNC_THRESH) kp = kth;
      else if (tr.kind == MPBP_TRUNC_BOND_THRESH) kp = min(kth, tr.mprime);
      else kp = min(len, tr.mprime);
      if (kp < 1) kp = 1;
      if (kp > P.cap_out) { kp = P.cap_out; if (tid == 0) stats->capacity_flag = 1; }
      if (tr.kind == MPBP_TRUNC_BOND_MAX && tid == 0 && tot > 0.0) {
        double dropped = 0.0;
        for (int j = kp; j < len; j++) { double s = sig[ord[j]]; dropped += s * s; }
        atomic_max_double_pos(&stats->maxerr_bits, sqrt(dropped / tot));
      }
    }
    // new core t: logical [kc, kp, s] = V[(k + kc*s), ord[k']]
    for (int idx = tid; idx < Rr * kp; idx += WG_THREADS) {
      int row = idx % Rr, k2i = idx / Rr;
      int k = row % kc, s = row / kc;
      double v = JA[row + ldJ * ord[k2i]];
      // physical layout: !mirror [kc, kp, s];  mirror [kp, kc, s]
      int64_t off = mirror ? ((int64_t)k2i + (int64_t)kp * (k + (int64_t)kc * s))
                           : ((int64_t)k + (int64_t)kc * (k2i + (int64_t)kp * s));
      oc[off] = v;
    }
    if (tid == 0) Pob[mirror ? (L - (t + 1)) : (t + 1)] = kp;
    PROF(PH_TRUNC);
    // carry C' [kp x Bn] = U^T Nt
    gemm_direct(kp, Bn, Rr, JA, [=](int i) { return (int64_t)ldJ * ord[i]; }, [=](int kk) { return kk; },
         Nt, [=](int kk) { return kk; }, [=](int j) { return (int64_t)Rr * j; },
         Cnew, [=](int i) { return i; }, [=](int j) { return (int64_t)kp * j; }, false);
    gdbl* tmp = Ccur; Ccur = Cnew; Cnew = tmp;
    kc = kp;
    PROF(PH_CARRY);
  }

  // ------------------------------------------------------------------ normalize_eachmatrix! + z
  double logz = (P.logz1 ? *P.logz1 : 0.0) + (P.logz2 ? *P.logz2 : 0.0) - logc;
  __syncthreads();
  for (int tp = 0; tp < L; tp++) {
    const int n = Pob[tp] * Pob[tp + 1] * ny * q;
    gdbl* oc = Pout + (int64_t)tp * P.ostride;
    double mx = 0.0;
    for (int idx = tid; idx < n; idx += WG_THREADS) mx = fmax(mx, fabs(oc[idx]));
    mx = wg_max(mx, red);
    if (mx > 0.0 && isfinite(mx)) {
      const double inv = 1.0 / mx;
      for (int idx = tid; idx < n; idx += WG_THREADS) oc[idx] *= inv;
      logz -= log(mx);
    }
  }
  PROF(PH_NORM);
#undef PROF
  if (tid == 0) {
    *P.ologz = logz;
    atomicAdd(&stats->n_compress, 1ULL);
    if (!(logz == logz)) stats->nan_flag = 1;
  }
  __syncthreads();
}

}  // namespace eng
