// Grid-level ("batched, lock-step") kernels of the gauge sweep: sweep 1 of the compress engine (engine.h) for problems
// that one workgroup cannot hold - BASELINE configs[2..4]: product bonds 900 / 1600 / 4096, Y_t up to 16384 x 4096 -
// and for batches whose workgroup-per-problem form would leave the chip idle.
//
// All problems of a batch advance through the time steps together; a time step is a short sequence of launches, each
// over (problem, tile / row chunk), so that ONE problem can occupy many compute units and a kernel boundary is the only
// synchronisation (no spin waits).  Per time step t (reference: the orthogonalize_right!(TruncThresh(0.0)) half of
// compress! inside `op`, src/recursive_bp_factor.jl:127; function-equivalent R-only QR as in engine.h):
//     k_build_E, k_gemm (Y1: Z = A1 Lf), k_zero_pads, k_gemm (Y2: Y = E Z)            - Y_t = X_t^T
//     per 64-column block: up to 4 x { k_trailW/k_trailU on the next panel's tile (left-looking inside the block),
//                                       panel factorisation, k_gram, k_build_T },
//                          k_trailW / k_trailU<NP> on everything to the right          - blocked Householder QR, R only
//     k_maxabs, k_lf_write                                                             - Lf_t^T = R / max|R|
// Panel factorisation: one launch of k_fpanel (register-resident panel, wg::qr_panel_regs) while the rows below the
// diagonal fit one workgroup (<= 2048), else 17 launches of k_colstep (one Householder column per launch, the rows
// shared out to chunks of 2048; the per-column dot products are reduced through a small per-problem buffer in a
// fixed order, so results do not depend on scheduling).
//
// Every panel is 16 columns wide: rows [rows, rows32) and columns [cols, cols16 + 16) of Y are zero, a reflector whose
// column is zero below the diagonal gets tau = 0 and a zero T row/column, and the padding stays zero - so "short" last
// panels need no special cases anywhere.
#pragma once

namespace v2 {

using namespace v512::wg;

constexpr int CH = 2048;          // rows per chunk (512 threads x 4 rows for the column steps; 64 stages of 32 rows for the updates)
constexpr int GSUB = 4;           // the Gram pass splits a chunk over GSUB workgroups (one or two problems: bandwidth of more CUs)

struct QrProb {
  double* Y;                      // column-major, ld % 32 == 0; see the padding contract above
  double* aux;                    // per-problem scratch, AuxLay
  int32_t ld, rows, cols, kmax;   // kmax = min(rows, cols)
};

// offsets (doubles) into QrProb::aux - the same for every problem of a batch
struct AuxLay {
  int32_t nchunk;                 // partial slots per tile in w0 = nchunk * 4 (row sub-chunks of the in-block updates)
  int64_t T, S, tau, piv, mx, bar, part, gram, w0;
};
// Per-problem scratch = TWO copies of this layout (the look-ahead of qr_batch: block k+1's panel factorisation writes its
// T / S / partial products while block k's trailing update still reads its own).  The two fixed-size HEADERS (T .. bar,
// AUX_HDR doubles each; zeroed once per gauge sweep, `bar` re-zeroed by k_build_T after every panel) sit together at the
// front, the two bodies (part, gram, w0) behind them:   [hdr 0 | hdr 1 | body 0 | body 1],  2 * auxlay_doubles in all.
// The communication-avoiding QR keeps its node slots in [a.part, 2 * auxlay_doubles) - both bodies, neither header.
constexpr int64_t AUX_HDR = 4 * 256 + 6 * 256 + 64 + 256 + 16 + 16;
__host__ __device__ inline AuxLay make_auxlay(int nchunk, int ntile) {
  AuxLay a; a.nchunk = nchunk;
  int64_t o = 0;
  a.T = o; o += 4 * 256;
  a.S = o; o += 6 * 256;
  a.tau = o; o += 64;
  a.piv = o; o += 256;
  a.mx = o; o += 16;
  a.bar = o; o += 16;            // 17 int32 arrival counters of the cooperative column steps (+ padding)
  o += AUX_HDR;                  // the second copy's header
  a.part = o; o += (int64_t)nchunk * 256;
  a.gram = o; o += (int64_t)nchunk * GSUB * 4 * 256;
  a.w0 = o; o += (int64_t)ntile * nchunk * 4 * 4 * 256;
  (void)o;
  return a;
}
__host__ __device__ inline int64_t auxlay_doubles(int nchunk, int ntile) {     // ONE copy (header + body)
  return AUX_HDR + (int64_t)nchunk * 256 + (int64_t)nchunk * GSUB * 1024 + (int64_t)ntile * nchunk * 4096;
}
// the second copy of layout `a` (auxd = auxlay_doubles of the same nchunk, ntile)
__host__ __device__ inline AuxLay second_auxlay(AuxLay a, int64_t auxd) {
  const int64_t body = auxd - AUX_HDR;
  a.T += AUX_HDR; a.S += AUX_HDR; a.tau += AUX_HDR; a.piv += AUX_HDR; a.mx += AUX_HDR; a.bar += AUX_HDR;
  a.part += body; a.gram += body; a.w0 += body;
  return a;
}

__device__ __forceinline__ double sel16(const double (&v)[16], int j) {
  double x = 0.0;
#pragma unroll
  for (int c = 0; c < 16; c++) x = (c == j) ? v[c] : x;
  return x;
}

// Householder reflector of a column with pivot alpha and squared norm ss below it (LAPACK dlarfg; the rsq / rcp +
// Newton form of wg::qr_panel_step, so that the two panel paths agree to rounding)
__device__ __forceinline__ void larfg(double alpha, double ss, double& beta, double& tj, double& scale) {
  if (ss == 0.0) { beta = alpha; tj = 0.0; scale = 0.0; return; }
  const double n2 = alpha * alpha + ss;
  double ri = __builtin_amdgcn_rsq(n2);
  ri = ri * (1.5 - 0.5 * n2 * ri * ri);
  ri = ri * (1.5 - 0.5 * n2 * ri * ri);
  double nrm = n2 * ri;
  nrm = nrm + 0.5 * ri * (n2 - nrm * nrm);
  beta = -copysign(nrm, alpha);
  tj = 1.0 + fabs(alpha) * ri;
  const double dd = alpha - beta;
  double rd = __builtin_amdgcn_rcp(dd);
  rd = rd * (2.0 - dd * rd);
  rd = rd * (2.0 - dd * rd);
  scale = rd;
}

// ------------------------------------------------------------------------------------------------------------------
// Column step jj (0..16) of the panel at column / row jp, rows shared out to chunks of CH.  Launch jj first applies
// reflector jj-1 (its dot products were left by launch jj-1) and then takes the dot products of column jj.
// grid (nchunk, nprob), 512 threads.  pidx = index of the panel inside its 64-column block (selects the tau slot).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_colstep(const QrProb* probs, AuxLay lay, int jp, int jj, int pidx) {
  const QrProb P = probs[blockIdx.y];
  if (jp >= P.kmax) return;
  const int chunk = blockIdx.x;
  const int rows32 = (P.rows + 31) & ~31;
  const int cfirst = jp / CH, clast = (rows32 - 1) / CH;
  if (chunk < cfirst || chunk > clast) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  gdbl* Y = (gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  const long ld = P.ld;
  __shared__ double red_[8 * 16];
  ldbl* red = (ldbl*)red_;
  double Pn[4][16];
  bool rv[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int r = chunk * CH + tid + 512 * s;
    rv[s] = r < rows32 && r >= jp;
    const int rc = rv[s] ? r : jp;
#pragma unroll
    for (int c = 0; c < 16; c++) { const double v = Y[(long)(jp + c) * ld + rc]; Pn[s][c] = rv[s] ? v : 0.0; }
  }
  if (jj >= 1) {
    const int j = jj - 1;
    double tot[16], rowv[16];
#pragma unroll
    for (int c = 0; c < 16; c++) tot[c] = 0.0;
    for (int cc = cfirst; cc <= clast; cc++) {
#pragma unroll
      for (int c = 0; c < 16; c++) tot[c] += aux[lay.part + ((long)cc * 16 + j) * 16 + c];
    }
#pragma unroll
    for (int c = 0; c < 16; c++) rowv[c] = aux[lay.piv + j * 16 + c];
    const double ss = sel16(tot, j), alpha = sel16(rowv, j);
    double beta, tj, scale;
    larfg(alpha, ss, beta, tj, scale);
    double tw[16];
#pragma unroll
    for (int c = 0; c < 16; c++) tw[c] = (c > j) ? tj * (rowv[c] + scale * tot[c]) : 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int r = chunk * CH + tid + 512 * s;
      const bool below = rv[s] && r > jp + j, pivot = r == jp + j;
      const double xj = sel16(Pn[s], j);
      const double v = below ? xj * scale : (pivot ? 1.0 : 0.0);
      const double nxj = below ? v : (pivot ? beta : xj);
#pragma unroll
      for (int c = 0; c < 16; c++) Pn[s][c] = (c == j) ? nxj : Pn[s][c] - tw[c] * v;
    }
    if (chunk == cfirst && tid == 0) aux[lay.tau + pidx * 16 + j] = tj;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int r = chunk * CH + tid + 512 * s;
      if (rv[s]) {
#pragma unroll
        for (int c = 0; c < 16; c++) Y[(long)(jp + c) * ld + r] = Pn[s][c];
      }
    }
  }
  if (jj < 16) {
    double vals[16];
#pragma unroll
    for (int c = 0; c < 16; c++) vals[c] = 0.0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int r = chunk * CH + tid + 512 * s;
      const bool below = rv[s] && r > jp + jj;
      const double x = below ? sel16(Pn[s], jj) : 0.0;
#pragma unroll
      for (int c = 0; c < 16; c++) vals[c] += x * Pn[s][c];
      if (r == jp + jj) {
#pragma unroll
        for (int c = 0; c < 16; c++) aux[lay.piv + jj * 16 + c] = Pn[s][c];
      }
    }
    int idx;
    const double wsum = wave_sum16(vals, lane, idx);
    if ((lane & 3) == 0) red[wave * 16 + idx] = wsum;
    __syncthreads();
    if (tid < 16) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 8; w++) s += red[w * 16 + tid];
      aux[lay.part + ((long)chunk * 16 + jj) * 16 + tid] = s;
    }
  }
}

// 8-byte write-through store / cache-bypassing load (global_store_dwordx2 sc1 / global_load_dwordx2 sc1): the payload forms
// of the fence-free hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, "Valid forms")
__device__ __forceinline__ void st_sc1(gdbl* p, double v) {
  __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)p, (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const gdbl* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const __attribute__((address_space(1))) unsigned long long*)p,
                                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ------------------------------------------------------------------------------------------------------------------
// The 17 column steps of a panel in ONE launch, for batches small enough that every participating workgroup is
// resident at once (host: nchunk * nprob <= 3/4 of the CUs; one 512-thread workgroup of this kernel fits per CU): each workgroup keeps its 2048 x 16 slice of the panel in
// registers for the whole factorisation (one load, one store instead of 17 of each) and the column steps are separated by
// an arrival counter per (problem, column) instead of kernel boundaries.  Hand-off protocol: the fence-free form gfx950
// is measured valid for (MI355X_MICROARCH.md, inter-workgroup visibility, "Valid forms", first table row; one workgroup
// per CU): every handed-off byte (16 partial sums per workgroup, the pivot row) is stored write-through (8-byte `sc1`
// stores) -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> lane 0: relaxed agent atomic add, relaxed `sc1` polls
// with s_sleep -> the polling wave reads every handed-off byte with `sc1` loads (never this CU's L1) and passes the
// sums on through LDS -> workgroup barrier.  No release / acquire fence (they cost 1.7 us each per step: 139 -> 1xx us
// per panel).  Every spin is bounded: after ~2^21 polls the
// workgroup raises *err and leaves (the host reports MPBP_EHIP) - the grid always drains.
// Counters are zeroed by k_build_T (the launch that follows every panel).  grid (nchunk, nprob), 512 threads.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_colsteps_coop(const QrProb* probs, AuxLay lay, int jp, int pidx, int* err) {
  const QrProb P = probs[blockIdx.y];
  if (jp >= P.kmax) return;
  const int chunk = blockIdx.x;
  const int rows32 = (P.rows + 31) & ~31;
  const int cfirst = jp / CH, clast = (rows32 - 1) / CH;
  if (chunk < cfirst || chunk > clast) return;
  const int nwg = clast - cfirst + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  gdbl* Y = (gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  typedef __attribute__((address_space(1))) int gint32;
  gint32* bar = (gint32*)(P.aux + lay.bar);           // typed: global_ (never flat_) atomics and polls
  gint32* gerr = (gint32*)err;
  const long ld = P.ld;
  __shared__ double red_[8 * 16 + 32];
  ldbl* red = (ldbl*)red_;
  ldbl* ex = red + 8 * 16;            // [16 column totals | 16 pivot-row entries] of the step just handed off
  double Pn[4][16];
  bool rv[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int r = chunk * CH + tid + 512 * s;
    rv[s] = r < rows32 && r >= jp;
    const int rc = rv[s] ? r : jp;
#pragma unroll
    for (int c = 0; c < 16; c++) { const double v = Y[(long)(jp + c) * ld + rc]; Pn[s][c] = rv[s] ? v : 0.0; }
  }
  // One column step with the column index a compile-time constant (the 17 steps are written out below): the panel slice
  // is then indexed statically - no 16-way selects - and columns left of the current one are skipped (as a run-time loop
  // the arithmetic of a step took as long as its hand-off).
  auto step = [&](auto jjc) {
    constexpr int jj = decltype(jjc)::value;
    if constexpr (jj >= 1) {
      constexpr int j = jj - 1;
      double tot[16], rowv[16];          // left in LDS by wave 0 after the hand-off of step j
#pragma unroll
      for (int c = j; c < 16; c++) { tot[c] = ex[c]; rowv[c] = ex[16 + c]; }
      double beta, tj, scale;
      larfg(rowv[j], tot[j], beta, tj, scale);
      double tw[16];
#pragma unroll
      for (int c = j + 1; c < 16; c++) tw[c] = tj * (rowv[c] + scale * tot[c]);
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int r = chunk * CH + tid + 512 * s;
        const bool below = rv[s] && r > jp + j, pivot = r == jp + j;
        const double xj = Pn[s][j];
        const double v = below ? xj * scale : (pivot ? 1.0 : 0.0);
        Pn[s][j] = below ? v : (pivot ? beta : xj);
#pragma unroll
        for (int c = j + 1; c < 16; c++) Pn[s][c] -= tw[c] * v;
      }
      if (chunk == cfirst && tid == 0) aux[lay.tau + pidx * 16 + j] = tj;
    }
    if constexpr (jj < 16) {
      double vals[16];
#pragma unroll
      for (int c = 0; c < 16; c++) vals[c] = 0.0;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const int r = chunk * CH + tid + 512 * s;
        const bool below = rv[s] && r > jp + jj;
        const double x = below ? Pn[s][jj] : 0.0;
#pragma unroll
        for (int c = jj; c < 16; c++) vals[c] += x * Pn[s][c];
        if (r == jp + jj) {
#pragma unroll
          for (int c = 0; c < 16; c++) st_sc1(aux + lay.piv + jj * 16 + c, Pn[s][c]);
        }
      }
      int idx;
      const double wsum = wave_sum16(vals, lane, idx);
      if ((lane & 3) == 0) red[wave * 16 + idx] = wsum;
      __syncthreads();
      if (tid < 16) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < 8; w++) s += red[w * 16 + tid];
        st_sc1(aux + lay.part + ((long)chunk * 16 + jj) * 16 + tid, s);
      }
      // ---- arrival: every (write-through) store of this workgroup is out, then lane 0 signals and waits for the others
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (wave == 0) {
        if (lane == 0) {
          __hip_atomic_fetch_add(&bar[jj], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while (__hip_atomic_load(&bar[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
            if (++spins > (1 << 21)) { __hip_atomic_store(gerr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(4);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the loads below behind the poll
        // the polling wave reads every handed-off byte with sc1 loads (L2 / memory, never this CU's L1) and leaves the
        // column totals and the pivot row in LDS for the other waves: fixed summation order (chunks 4 apart per lane
        // group, then the groups)
        const int c = lane & 15, sub = lane >> 4;
        double s = 0.0;
        for (int cc = cfirst + sub; cc <= clast; cc += 4) s += ld_sc1(aux + lay.part + ((long)cc * 16 + jj) * 16 + c);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const double pv = ld_sc1(aux + lay.piv + jj * 16 + c);
        if (lane < 16) { ex[c] = s; ex[16 + c] = pv; }
      }
      __syncthreads();
    }
  };
#define MPBP_STEP(J) step(std::integral_constant<int, J>{});
  MPBP_STEP(0) MPBP_STEP(1) MPBP_STEP(2) MPBP_STEP(3) MPBP_STEP(4) MPBP_STEP(5) MPBP_STEP(6) MPBP_STEP(7) MPBP_STEP(8)
  MPBP_STEP(9) MPBP_STEP(10) MPBP_STEP(11) MPBP_STEP(12) MPBP_STEP(13) MPBP_STEP(14) MPBP_STEP(15) MPBP_STEP(16)
#undef MPBP_STEP
  // an arrival counter timed out somewhere in the launch (workgroups not co-resident): the panel in registers is
  // garbage - leave Y as it was; the host repeats the batch with one launch per column step (launch_engine)
  if (__hip_atomic_load(gerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int r = chunk * CH + tid + 512 * s;
    if (rv[s]) {
#pragma unroll
      for (int c = 0; c < 16; c++) Y[(long)(jp + c) * ld + r] = Pn[s][c];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Register-resident panel (rows below the diagonal <= 2048): the whole 16-column factorisation in ONE launch.
// grid (nprob), 512 threads.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_fpanel(const QrProb* probs, AuxLay lay, int jb, int pidx) {
  const QrProb P = probs[blockIdx.x];
  const int jp = jb + 16 * pidx;
  if (jp >= P.kmax) return;
  extern __shared__ __attribute__((aligned(16))) double dyn_[];
  ldbl* red = (ldbl*)dyn_;                 // [2][8*16]
  ldbl* tau = red + 2 * 8 * 16;            // [16]
  ldbl* bc = tau + 16;                     // [96]
  ldbl* Ts = bc + 96;                      // [256]
  ldbl* big = Ts + 256;                    // [8 * 256 * (pidx + 1)]
  const int tid = threadIdx.x;
  gdbl* Y = (gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  const int rows32 = (P.rows + 31) & ~31;
  if (tid < 16) tau[tid] = 0.0;
  __syncthreads();
  qr_panel_regs(Y, P.ld, rows32, jp, 16, red, tau, bc);
  // Gram of the new panel with itself and with the earlier panels of its block, T from the Gram (one pass over the rows)
  switch (pidx) {
    case 0: { const int jy[1] = {jp}; qr_gramN<1>(Y, P.ld, rows32, jp, jp, jy, big); break; }
    case 1: { const int jy[2] = {jp, jb}; qr_gramN<2>(Y, P.ld, rows32, jp, jp, jy, big); break; }
    case 2: { const int jy[3] = {jp, jb, jb + 16}; qr_gramN<3>(Y, P.ld, rows32, jp, jp, jy, big); break; }
    default: { const int jy[4] = {jp, jb, jb + 16, jb + 32}; qr_gramN<4>(Y, P.ld, rows32, jp, jp, jy, big); break; }
  }
  for (int k = 1; k <= pidx; k++)
    for (int i = tid; i < 256; i += 512) aux[lay.S + (long)(pidx * (pidx - 1) / 2 + (k - 1)) * 256 + i] = big[256 * k + i];
  qr_T_from_gram(big, tau, 16, Ts);
  for (int i = tid; i < 256; i += 512) aux[lay.T + pidx * 256 + i] = Ts[i];
  if (tid < 16) aux[lay.tau + pidx * 16 + tid] = tau[tid];
}
constexpr int FPANEL_LDS_BASE = 2 * 8 * 16 + 16 + 96 + 256;
__host__ __device__ inline size_t fpanel_lds_bytes(int pidx) { return sizeof(double) * (FPANEL_LDS_BASE + 8 * 256 * (pidx + 1)); }

// Left-looking update inside a block for problems whose rows fit one workgroup's pass (any number of rows works; the
// launch is used while the rows below the diagonal are <= CH): tile = panel NP of the block at jb, updated by panels
// 0 .. NP-1; the eight waves share the rows, partial products through LDS.  grid (nprob), 512 threads.
template <int NP>
__global__ void __launch_bounds__(512) k_inblock(const QrProb* probs, AuxLay lay, int jb) {
  const QrProb P = probs[blockIdx.x];
  const int cb0 = jb + 16 * NP;
  if (cb0 >= P.kmax) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  gdbl* Y = (gdbl*)P.Y;
  const gdbl* aux = (const gdbl*)P.aux;
  const long ld = P.ld;
  const int rows32 = (P.rows + 31) & ~31;
  constexpr int NS = NP * (NP - 1) / 2;
  __shared__ double sh_[(NP + NS + 8 * NP) * 256];
  ldbl* Tq = (ldbl*)sh_;
  ldbl* Sq = Tq + NP * 256;
  ldbl* big = Sq + NS * 256;
  for (int i = tid; i < NP * 256; i += 512) Tq[i] = aux[lay.T + i];
  for (int i = tid; i < (NP * (NP - 1) / 2) * 256; i += 512) Sq[i] = aux[lay.S + i];
  d4 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = d4{0, 0, 0, 0};
  const int nrb = (rows32 - jb) >> 4;
  for (int rb = wave; rb < nrb; rb += 8) {
    const int row0 = jb + 16 * rb + 4 * g;
    const d4 c = *reinterpret_cast<const gd4*>(Y + (long)(cb0 + l15) * ld + row0);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      d4 v = *reinterpret_cast<const gd4*>(Y + (long)(jb + 16 * p + l15) * ld + row0);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int rr = 16 * (rb - p) + 4 * g + e;
        double a = v[e];
        a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
        v[e] = (rr >= 0) ? a : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) acc[p] = mfma(v[e], c[e], acc[p]);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int r = 0; r < 4; r++) big[(wave * NP + p) * 256 + (g + 4 * r) + 16 * l15] = acc[p][r];
  __syncthreads();
  d4 w[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    d4 t = d4{0, 0, 0, 0};
#pragma unroll
    for (int w2 = 0; w2 < 8; w2++)
#pragma unroll
      for (int r = 0; r < 4; r++) t[r] += big[(w2 * NP + p) * 256 + (g + 4 * r) + 16 * l15];
#pragma unroll
    for (int r = 0; r < NP; r++) {
      if (r < p) {
        const ldbl* S = Sq + (p * (p - 1) / 2 + r) * 256;
#pragma unroll
        for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][s], t);
      }
    }
    d4 o = d4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; s++) o = mfma(Tq[p * 256 + (4 * s + g) + 16 * l15], t[s], o);
    w[p] = o;
  }
  const int jb32 = jb & ~31;
  const int nst = (rows32 - jb32) >> 5;
  for (int st = wave; st < nst; st += 8) {
    const int row = jb32 + 32 * st + 2 * l15;
    d2 v[NP][4];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        const int k = 4 * s2 + g;
        d2 x = *reinterpret_cast<const gd2*>(Y + (long)(jb + 16 * p + k) * ld + row);
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const int rp = row + e - jb - 16 * p;
          double a = x[e];
          a = (rp < 16) ? ((rp > k) ? a : ((rp == k) ? 1.0 : 0.0)) : a;
          x[e] = (rp >= 0) ? a : 0.0;
        }
        v[p][s2] = x;
      }
    d2 c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row);
#pragma unroll
    for (int e = 0; e < 2; e++) {
      d4 a4 = d4{c[0][e], c[1][e], c[2][e], c[3][e]};
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
        for (int p = 0; p < NP; p++) a4 = mfma_na(w[p][s2], v[p][s2][e], a4);
#pragma unroll
      for (int r = 0; r < 4; r++) c[r][e] = a4[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) *reinterpret_cast<gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row) = c[r];
  }
}

// Trailing update by the four panels of the block at jb for problems that have all four (np == 4): every wave takes a
// pair of 16-column tiles over ALL rows (wg::qr_trail4: phases A, B, C in registers, the V fragments of a row block
// shared by both tiles).  For batches of many problems - one wave per tile pair is only enough parallelism then.
// grid (ceil(tile pairs / 4), nprob), 256 threads.
template <int NT>
__global__ void __launch_bounds__(256, NT == 1 ? 3 : 2) k_trail4f(const QrProb* probs, AuxLay lay, int jb) {
  const QrProb P = probs[blockIdx.y];
  if (P.kmax - jb < 64) return;                       // fewer than four panels left: k_trailW / k_trailU<NP> take it
  const int c0 = jb + 64;
  const int ntile = (P.cols > c0) ? (P.cols - c0 + 15) >> 4 : 0;
  if (blockIdx.x * 4 * NT >= ntile) return;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const gdbl* aux = (const gdbl*)P.aux;
  __shared__ double ts_[10 * 256];
  ldbl* T0 = (ldbl*)ts_;
  for (int i = tid; i < 4 * 256; i += 256) T0[i] = aux[lay.T + i];
  for (int i = tid; i < 6 * 256; i += 256) T0[1024 + i] = aux[lay.S + i];
  __syncthreads();
  const ldbl* const Tq[4] = {T0, T0 + 256, T0 + 512, T0 + 768};
  const ldbl* const Sq[6] = {T0 + 1024, T0 + 1280, T0 + 1536, T0 + 1792, T0 + 2048, T0 + 2304};
  const int tile = (blockIdx.x * 4 + wave) * NT;
  if (tile >= ntile) return;
  const int rows32 = (P.rows + 31) & ~31;
  if (NT == 2 && tile + 1 < ntile) qr_trail4<2>((gdbl*)P.Y, P.ld, rows32, jb, c0 + 16 * tile, Tq, Sq);
  else qr_trail4<1>((gdbl*)P.Y, P.ld, rows32, jb, c0 + 16 * tile, Tq, Sq);
}

// ------------------------------------------------------------------------------------------------------------------
// Gram blocks of the freshly factored panel x (index pidx of the block at column jb) against itself and the earlier
// panels of its block:  G_0 = Vx^T Vx,  G_k = Vx^T V_{k-1} (k = 1..pidx), rows of a quarter chunk.  grid (nchunk * GSUB, nprob).
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_gram(const QrProb* probs, AuxLay lay, int jb, int pidx) {
  const QrProb P = probs[blockIdx.y];
  const int jx = jb + 16 * pidx;
  if (jx >= P.kmax) return;
  const int chunk = blockIdx.x / GSUB, sub = blockIdx.x % GSUB;
  const int rows32 = (P.rows + 31) & ~31;
  const int cfirst = jx / CH, clast = (rows32 - 1) / CH;
  if (chunk < cfirst || chunk > clast) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const gdbl* Y = (const gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  const long ld = P.ld;
  const int np = pidx + 1;
  __shared__ double big_[8 * 1024];
  ldbl* big = (ldbl*)big_;
  const int r0 = chunk * CH + sub * (CH / GSUB);
  const int rb0 = max(r0, jx) >> 4, rb1 = min(r0 + CH / GSUB, rows32) >> 4;        // may be empty: zeros are written
  d4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; k++) acc[k] = d4{0, 0, 0, 0};
  for (int rb = rb0 + wave; rb < rb1; rb += 8) {
    const int row0 = 16 * rb + 4 * g;
    d4 vx = *reinterpret_cast<const gd4*>(Y + (long)(jx + l15) * ld + row0);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rr = row0 + e - jx;
      double a = vx[e];
      a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
      vx[e] = (rr >= 0) ? a : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k < np) {
        const int jy = (k == 0) ? jx : jb + 16 * (k - 1);
        d4 vy = *reinterpret_cast<const gd4*>(Y + (long)(jy + l15) * ld + row0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int rr = row0 + e - jy;
          double a = vy[e];
          a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
          vy[e] = (rr >= 0) ? a : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) acc[k] = mfma(vx[e], vy[e], acc[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int r = 0; r < 4; r++) big[wave * 1024 + 256 * k + (g + 4 * r) + 16 * l15] = acc[k][r];
  __syncthreads();
  for (int idx = tid; idx < 256 * np; idx += 512) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < 8; w++) s += big[w * 1024 + idx];
    aux[lay.gram + (long)(chunk * GSUB + sub) * 1024 + idx] = s;
  }
}

// T of the panel from its Gram block and taus, cross Grams S_pr of the block; grid (nprob), 256 threads: one entry of
// the (up to four) 16x16 blocks per thread, the partial sums of the row chunks in four independent chains so that the
// loads overlap (as 64 threads with one serial chain per entry this launch took 22 us for 1 us of work)
__global__ void __launch_bounds__(256) k_build_T(const QrProb* probs, AuxLay lay, int jb, int pidx) {
  const QrProb P = probs[blockIdx.x];
  const int jx = jb + 16 * pidx;
  if (jx >= P.kmax) return;
  const int rows32 = (P.rows + 31) & ~31;
  const int cfirst = jx / CH, cl = (rows32 - 1) / CH;
  const int tid = threadIdx.x;
  gdbl* aux = (gdbl*)P.aux;
  __shared__ double g_[256 + 16 + 256];
  ldbl* G = (ldbl*)g_;
  ldbl* tau = G + 256;
  ldbl* Ts = tau + 16;
  const int np = pidx + 1;
  double sk[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (k < np) {
      const gdbl* src = aux + lay.gram + 256 * k + tid;
      int cc = cfirst * GSUB;
      const int clast = cl * GSUB + GSUB - 1;
      for (; cc + 3 <= clast; cc += 4) {
        s0 += src[(long)cc * 1024]; s1 += src[(long)(cc + 1) * 1024]; s2 += src[(long)(cc + 2) * 1024]; s3 += src[(long)(cc + 3) * 1024];
      }
      for (; cc <= clast; cc++) s0 += src[(long)cc * 1024];
    }
    sk[k] = (s0 + s1) + (s2 + s3);
  }
  G[tid] = sk[0];
#pragma unroll
  for (int k = 1; k < 4; k++)
    if (k < np) aux[lay.S + (long)(pidx * (pidx - 1) / 2 + (k - 1)) * 256 + tid] = sk[k];
  if (tid < 16) tau[tid] = aux[lay.tau + pidx * 16 + tid];
  __syncthreads();
  qr_T_from_gram(G, tau, 16, Ts);
  aux[lay.T + pidx * 256 + tid] = Ts[tid];
  if (tid < 32) ((int*)(P.aux + lay.bar))[tid] = 0;          // arrival counters of the next cooperative panel
}

// ------------------------------------------------------------------------------------------------------------------
// Block-reflector update of 16-column tiles by the NP panels of the block at column jb:
//     C <- C - sum_p V_p W_p,   W_p = T_p^T (V_p^T C - sum_{r<p} S_pr W_r)
// in two launches: k_trailW leaves the partial products V_p^T C of every (tile, row sub-chunk), k_trailU sums them
// in a fixed order, runs the W recurrence and updates its own rows.
// A workgroup of 4 waves covers tw tiles x rw row sub-chunks (tw * rw = 4): tw = 4 for the trailing matrix, rw = 4 for
// the single tile of a left-looking update inside the block.  grid (ceil(ntile / tw), nchunk, nprob), 256 threads.
// `inblock`: the tile is panel NP of the block (column jb + 16 NP); else tiles start right of the block's panels.
// ------------------------------------------------------------------------------------------------------------------
struct TrailGeom { int np, c0, ntile; };
__device__ __forceinline__ TrailGeom trail_geom(const QrProb& P, int jb, int NP, bool inblock) {
  TrailGeom G;
  const int npan = (P.kmax - jb + 15) >> 4;              // panels of this problem from column jb on (>= 1)
  if (inblock) { G.np = NP; G.c0 = jb + 16 * NP; G.ntile = (npan > NP) ? 1 : 0; return G; }
  G.np = min(NP, npan);                                  // NP = panels of the widest problem in this block (<= 4)
  G.c0 = jb + 16 * G.np;
  G.ntile = (P.cols > G.c0) ? (P.cols - G.c0 + 15) >> 4 : 0;
  return G;
}

template <int NP>
__global__ void __launch_bounds__(256) k_trailW(const QrProb* probs, AuxLay lay, int jb, int inblock, int tw, int only_short, int tf0 = 0) {
  const QrProb P = probs[blockIdx.z];
  if (jb >= P.kmax) return;
  if (only_short && P.kmax - jb >= 64) return;        // k_trail4f has taken the problems with four panels
  const TrailGeom G = trail_geom(P, jb, NP, inblock != 0);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  // tw = 0: ONE tile (the left-looking update inside a block), a quarter chunk per workgroup and a quarter of that per
  // wave (grid.x = 4): four times the workgroups of the tw = 1 form, whose waves walked 512 rows each with one problem
  // on the chip; the four waves' products are summed through LDS into the same slot layout (chunk * 4 + quarter)
  const bool fine = tw == 0;
  const int rw = fine ? 4 : 4 / tw;
  // fine form: grid.x = 4 * tiles, starting at tile tf0 (the in-block update has one tile; the look-ahead of qr_batch
  // updates the next block's four panel tiles this way, inblock = 0)
  const int tile = fine ? tf0 + (int)(blockIdx.x >> 2) : blockIdx.x * tw + (wave % tw), rsub = fine ? (int)(blockIdx.x & 3) : wave / tw;
  if (tile >= G.ntile) return;
  const int rows32 = (P.rows + 31) & ~31;
  const int chunk = blockIdx.y;
  const int sub = CH / rw;                                 // rows per sub-chunk (multiple of 32)
  const int wsub = fine ? sub / 4 : sub, woff = fine ? wave * wsub : 0;
  const int ra = max(chunk * CH + rsub * sub + woff, jb), rbnd = min(chunk * CH + rsub * sub + woff + wsub, rows32);
  const gdbl* Y = (const gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  const long ld = P.ld;
  const int cb0 = G.c0 + 16 * tile;
  d4 acc[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) acc[p] = d4{0, 0, 0, 0};
  const gdbl* ccol = Y + (long)(cb0 + l15) * ld + 4 * g;
  for (int rb = ra >> 4; rb < (rbnd >> 4); rb++) {
    const int row0 = 16 * rb + 4 * g;
    const d4 c = *reinterpret_cast<const gd4*>(ccol + 16 * rb);
#pragma unroll
    for (int p = 0; p < NP; p++) {
      if (p < G.np) {
        const int jv = jb + 16 * p;
        d4 v = *reinterpret_cast<const gd4*>(Y + (long)(jv + l15) * ld + row0);
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int rr = row0 + e - jv;
          double a = v[e];
          a = (rr < 16) ? ((rr > l15) ? a : ((rr == l15) ? 1.0 : 0.0)) : a;
          v[e] = (rr >= 0) ? a : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) acc[p] = mfma(v[e], c[e], acc[p]);
      }
    }
  }
  gdbl* w0 = aux + lay.w0 + ((long)tile * (lay.nchunk * 4) + chunk * 4 + rsub) * 1024;
  if (fine) {
    __shared__ double ws_[4 * NP * 256];
    ldbl* ws = (ldbl*)ws_;
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int r = 0; r < 4; r++) ws[(wave * NP + p) * 256 + (g + 4 * r) + 16 * l15] = acc[p][r];
    __syncthreads();
    for (int i = threadIdx.x; i < NP * 256; i += 256)
      w0[i] = (ws[i] + ws[NP * 256 + i]) + (ws[2 * NP * 256 + i] + ws[3 * NP * 256 + i]);
    return;
  }
#pragma unroll
  for (int p = 0; p < NP; p++)
#pragma unroll
    for (int r = 0; r < 4; r++) w0[256 * p + (g + 4 * r) + 16 * l15] = acc[p][r];
}

template <int NP>
__global__ void __launch_bounds__(256) k_trailU(const QrProb* probs, AuxLay lay, int jb, int inblock, int tw, int only_short, int tf0 = 0) {
  const QrProb P = probs[blockIdx.z];
  if (jb >= P.kmax) return;
  if (only_short && P.kmax - jb >= 64) return;
  const TrailGeom G = trail_geom(P, jb, NP, inblock != 0);
  if (G.ntile == 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  gdbl* Y = (gdbl*)P.Y;
  const gdbl* aux = (const gdbl*)P.aux;
  const long ld = P.ld;
  __shared__ double ts_[10 * 256];
  ldbl* Tq = (ldbl*)ts_;
  ldbl* Sq = Tq + 4 * 256;
  for (int i = tid; i < 4 * 256; i += 256) Tq[i] = aux[lay.T + i];
  for (int i = tid; i < 6 * 256; i += 256) Sq[i] = aux[lay.S + i];
  __syncthreads();
  const bool fine = tw == 0;                               // see k_trailW
  const int rw = fine ? 4 : 4 / tw;
  const int tile = fine ? tf0 + (int)(blockIdx.x >> 2) : blockIdx.x * tw + (wave % tw), rsub = fine ? (int)(blockIdx.x & 3) : wave / tw;
  if (tile >= G.ntile) return;
  const int rows32 = (P.rows + 31) & ~31;
  const int chunk = blockIdx.y;
  const int sub = CH / rw;
  const int wsub = fine ? sub / 4 : sub, woff = fine ? wave * wsub : 0;
  const int ra = max(chunk * CH + rsub * sub + woff, jb & ~31), rbnd = min(chunk * CH + rsub * sub + woff + wsub, rows32);
  if (!fine && ra >= rbnd) return;
  const int cb0 = G.c0 + 16 * tile;
  // partial products of every slot that took part: chunks from the one holding row jb, all 4 / rw... sub-chunks
  const int cfirst = jb / CH, clast = (rows32 - 1) / CH;
  // fine form: the four waves of the workgroup (same tile) share the sum over the slots - slots 4 apart per wave, the four
  // partial sums through LDS in a fixed order - instead of every wave walking all of them
  __shared__ double wsum_[4 * NP * 256];
  ldbl* wsum = (ldbl*)wsum_;
  if (fine) {
#pragma unroll
    for (int p = 0; p < NP; p++) {
      d4 t = d4{0, 0, 0, 0};
      if (p < G.np)
        for (int sl = cfirst * 4 + wave; sl <= clast * 4 + 3; sl += 4) {
          const gdbl* w0 = aux + lay.w0 + ((long)tile * (lay.nchunk * 4) + sl) * 1024 + 256 * p;      // every slot was written
#pragma unroll
          for (int r = 0; r < 4; r++) t[r] += w0[(g + 4 * r) + 16 * l15];
        }
#pragma unroll
      for (int r = 0; r < 4; r++) wsum[(wave * NP + p) * 256 + (g + 4 * r) + 16 * l15] = t[r];
    }
  }
  if (fine) __syncthreads();          // uniform: `fine` is a launch parameter, and no wave of a fine workgroup has left
  if (ra >= rbnd) return;
  d4 w[NP];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    d4 t = d4{0, 0, 0, 0};
    if (p < G.np) {
      if (fine) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int o = p * 256 + (g + 4 * r) + 16 * l15;
          t[r] = (wsum[o] + wsum[NP * 256 + o]) + (wsum[2 * NP * 256 + o] + wsum[3 * NP * 256 + o]);
        }
      } else
      for (int cc = cfirst; cc <= clast; cc++)
        for (int rs = 0; rs < rw; rs++) {
          // sub-chunks entirely above row jb or below the matrix wrote nothing: skip them exactly as k_trailW did
          const int sa = max(cc * CH + rs * sub, jb), sb = min(cc * CH + (rs + 1) * sub, rows32);
          if ((sa >> 4) >= (sb >> 4)) continue;
          const gdbl* w0 = aux + lay.w0 + ((long)tile * (lay.nchunk * 4) + cc * 4 + rs) * 1024 + 256 * p;
#pragma unroll
          for (int r = 0; r < 4; r++) t[r] += w0[(g + 4 * r) + 16 * l15];
        }
#pragma unroll
      for (int r = 0; r < NP; r++) {
        if (r < p) {
          const ldbl* S = Sq + (p * (p - 1) / 2 + r) * 256;
#pragma unroll
          for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][s], t);
        }
      }
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[p * 256 + (4 * s + g) + 16 * l15], t[s], o);
      t = o;
    }
    w[p] = t;
  }
  for (int st = ra >> 5; st < (rbnd >> 5); st++) {
    const int row = 32 * st + 2 * l15;
    d2 v[NP][4];
#pragma unroll
    for (int p = 0; p < NP; p++)
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++) {
        const int k = 4 * s2 + g;
        d2 x = d2{0, 0};
        if (p < G.np) {
          x = *reinterpret_cast<const gd2*>(Y + (long)(jb + 16 * p + k) * ld + row);
#pragma unroll
          for (int e = 0; e < 2; e++) {
            const int rp = row + e - jb - 16 * p;
            double a = x[e];
            a = (rp < 16) ? ((rp > k) ? a : ((rp == k) ? 1.0 : 0.0)) : a;
            x[e] = (rp >= 0) ? a : 0.0;
          }
        }
        v[p][s2] = x;
      }
    d2 c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = *reinterpret_cast<const gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row);
#pragma unroll
    for (int e = 0; e < 2; e++) {
      d4 a4 = d4{c[0][e], c[1][e], c[2][e], c[3][e]};
#pragma unroll
      for (int s2 = 0; s2 < 4; s2++)
#pragma unroll
        for (int p = 0; p < NP; p++) a4 = mfma_na(w[p][s2], v[p][s2][e], a4);
#pragma unroll
      for (int r = 0; r < 4; r++) c[r][e] = a4[r];
    }
#pragma unroll
    for (int r = 0; r < 4; r++) *reinterpret_cast<gd2*>(Y + (long)(cb0 + g + 4 * r) * ld + row) = c[r];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two launches with the reflector panels shared through LDS (the scheme of wg::qr_trail4_coop, split at the
// reduction over row chunks): a workgroup of 8 waves takes 16 tiles (two per wave) x one row chunk; every 32-row stage of
// the four panels is loaded once by the workgroup into a double-buffered LDS tile (column stride 34 rows: conflict free
// for both fragment shapes; heads in unit-lower-trapezoidal form), C goes HBM -> registers (one stage prefetched),
// stage barriers order LDS only.  Problems with four full panels in the block; the others take k_trailW / k_trailU<NP>.
// NT = tiles per wave: 2 (16 tiles per workgroup, one workgroup per CU) or 1 (8 tiles, two workgroups per CU: more
// workgroups for small grids, more waves to hide latency).  grid (ceil(ntile / (8 NT)), nchunk, nprob), 512 threads.
// ------------------------------------------------------------------------------------------------------------------
struct CoopStage {
  const gdbl* vsrc; int sc, sr, spanel, scol;
  __device__ __forceinline__ void init(const gdbl* Y, long ld, int jb) {
    const int tid = threadIdx.x;
    sc = tid >> 3; sr = tid & 7; spanel = sc >> 4; scol = sc & 15;
    vsrc = Y + (long)(jb + sc) * ld + 4 * sr;
  }
  // rows [32 s, 32 s + 32) (absolute); panel p's diagonal block starts at row jb + 16 p
  __device__ __forceinline__ d4 load(int s, int slast) const { return *reinterpret_cast<const gd4*>(vsrc + 32 * min(s, slast)); }
  __device__ __forceinline__ void store(ldbl* Vs, int s, int jb, d4 v) const {
    ldbl* dst = Vs + (s & 1) * QR_VS_STAGE + sc * QR_VS_LD + 4 * sr;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int rp = 32 * s + 4 * sr + e - jb - 16 * spanel;
      double a = v[e];
      a = (rp < 16) ? ((rp > scol) ? a : ((rp == scol) ? 1.0 : 0.0)) : a;
      v[e] = (rp >= 0) ? a : 0.0;
    }
    *reinterpret_cast<ld2*>(dst) = d2{v[0], v[1]};
    *reinterpret_cast<ld2*>(dst + 2) = d2{v[2], v[3]};
  }
};

template <int NT>
__global__ void __launch_bounds__(512, NT == 1 ? 4 : 2) k_trailW_coop(const QrProb* probs, AuxLay lay, int jb, int t0, int t1) {
  // tiles [t0, t1) of the trailing matrix (counted from column jb + 64): the look-ahead of qr_batch updates the next
  // blocks' panel columns first and the rest beside the next panel factorisation
  const QrProb P = probs[blockIdx.z];
  if (P.kmax - jb < 64) return;
  const int c0 = jb + 64;
  const int ntile = min(t1, (P.cols > c0) ? (P.cols - c0 + 15) >> 4 : 0);
  if (t0 + blockIdx.x * 8 * NT >= ntile) return;
  const int rows32 = (P.rows + 31) & ~31;
  const int chunk = blockIdx.y;
  const int s0 = max(chunk * CH, jb) >> 5, s1 = min((chunk + 1) * CH, rows32) >> 5;     // stages of this chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  const gdbl* Y = (const gdbl*)P.Y;
  gdbl* aux = (gdbl*)P.aux;
  const long ld = P.ld;
  __shared__ double vs_[2 * QR_VS_STAGE];
  ldbl* Vs = (ldbl*)vs_;
  const int tile = t0 + blockIdx.x * 8 * NT + NT * wave;
  const int nt = max(0, min(NT, ntile - tile));
  const int cq0 = c0 + 16 * ((nt >= 1) ? tile : 0), cq1 = (nt >= 2) ? cq0 + 16 : cq0;
  d4 w0[4][NT];
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < NT; q++) w0[p][q] = d4{0, 0, 0, 0};
  if (s0 < s1) {
    CoopStage st; st.init(Y, ld, jb);
    const gdbl* c0p = Y + (long)(cq0 + l15) * ld + 4 * g;
    const gdbl* c1p = Y + (long)(cq1 + l15) * ld + 4 * g;
    d4 vreg = st.load(s0, s1 - 1);
    st.store(Vs, s0, jb, vreg);
    d4 cc[NT][2];
    cc[0][0] = *reinterpret_cast<const gd4*>(c0p + 32 * s0); cc[0][1] = *reinterpret_cast<const gd4*>(c0p + 32 * s0 + 16);
    if (NT > 1) { cc[NT - 1][0] = *reinterpret_cast<const gd4*>(c1p + 32 * s0); cc[NT - 1][1] = *reinterpret_cast<const gd4*>(c1p + 32 * s0 + 16); }
    for (int s = s0; s < s1; s++) {
      vreg = st.load(s + 1, s1 - 1);
      const int sn = min(s + 1, s1 - 1);
      d4 cn[NT][2];
      cn[0][0] = *reinterpret_cast<const gd4*>(c0p + 32 * sn); cn[0][1] = *reinterpret_cast<const gd4*>(c0p + 32 * sn + 16);
      if (NT > 1) { cn[NT - 1][0] = *reinterpret_cast<const gd4*>(c1p + 32 * sn); cn[NT - 1][1] = *reinterpret_cast<const gd4*>(c1p + 32 * sn + 16); }
      lds_barrier();
      const ldbl* vb = Vs + (s & 1) * QR_VS_STAGE;
      if (nt > 0) {
#pragma unroll
        for (int rb = 0; rb < 2; rb++) {
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const ldbl* vp = vb + (16 * p + l15) * QR_VS_LD + 16 * rb + 4 * g;
            const d2 va = *reinterpret_cast<const ld2*>(vp), vbb = *reinterpret_cast<const ld2*>(vp + 2);
            const double v4[4] = {va[0], va[1], vbb[0], vbb[1]};
            if (NT > 1 && nt > 1) {
#pragma unroll
              for (int e = 0; e < 4; e++) { w0[p][0] = mfma(v4[e], cc[0][rb][e], w0[p][0]); w0[p][NT - 1] = mfma(v4[e], cc[NT - 1][rb][e], w0[p][NT - 1]); }
            } else {
#pragma unroll
              for (int e = 0; e < 4; e++) w0[p][0] = mfma(v4[e], cc[0][rb][e], w0[p][0]);
            }
          }
        }
      }
      st.store(Vs, s + 1, jb, vreg);
#pragma unroll
      for (int q = 0; q < NT; q++) { cc[q][0] = cn[q][0]; cc[q][1] = cn[q][1]; }
    }
  }
  // partial products of this chunk, in the slot layout k_trailU reads (row sub-chunk 0 of 4)
#pragma unroll
  for (int q = 0; q < NT; q++) {
    if (q < nt) {
      gdbl* w0o = aux + lay.w0 + ((long)(tile + q) * (lay.nchunk * 4) + chunk * 4) * 1024;
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int r = 0; r < 4; r++) w0o[256 * p + (g + 4 * r) + 16 * l15] = w0[p][q][r];
    }
  }
}

template <int NT>
__global__ void __launch_bounds__(512, NT == 1 ? 4 : 2) k_trailU_coop(const QrProb* probs, AuxLay lay, int jb, int t0, int t1) {
  const QrProb P = probs[blockIdx.z];
  if (P.kmax - jb < 64) return;
  const int c0 = jb + 64;
  const int ntile = min(t1, (P.cols > c0) ? (P.cols - c0 + 15) >> 4 : 0);
  if (t0 + blockIdx.x * 8 * NT >= ntile) return;
  const int rows32 = (P.rows + 31) & ~31;
  const int chunk = blockIdx.y;
  const int s0 = max(chunk * CH, jb) >> 5, s1 = min((chunk + 1) * CH, rows32) >> 5;
  if (s0 >= s1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  gdbl* Y = (gdbl*)P.Y;
  const gdbl* aux = (const gdbl*)P.aux;
  const long ld = P.ld;
  __shared__ double sh_[10 * 256 + 2 * QR_VS_STAGE];
  ldbl* Tq = (ldbl*)sh_;
  ldbl* Sq = Tq + 4 * 256;
  ldbl* Vs = Sq + 6 * 256;
  for (int i = tid; i < 4 * 256; i += 512) Tq[i] = aux[lay.T + i];
  for (int i = tid; i < 6 * 256; i += 512) Sq[i] = aux[lay.S + i];
  const int tile = t0 + blockIdx.x * 8 * NT + NT * wave;
  const int nt = max(0, min(NT, ntile - tile));
  const int tq0 = (nt >= 1) ? tile : 0, tq1 = (nt >= 2) ? tile + 1 : tq0;
  const int cq0 = c0 + 16 * tq0, cq1 = c0 + 16 * tq1;
  CoopStage st; st.init(Y, ld, jb);
  d4 vreg = st.load(s0, s1 - 1);
  __syncthreads();                                           // T, S are in LDS
  st.store(Vs, s0, jb, vreg);
  // W of this wave's tiles: partial products of every chunk in a fixed order, then the recurrence
  const int cfirst = jb / CH, clast = (rows32 - 1) / CH;
  d4 w[4][NT];
#pragma unroll
  for (int q = 0; q < NT; q++) {
    const int tl = q ? tq1 : tq0;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      d4 t = d4{0, 0, 0, 0};
      for (int cc = cfirst; cc <= clast; cc++) {
        const gdbl* w0 = aux + lay.w0 + ((long)tl * (lay.nchunk * 4) + cc * 4) * 1024 + 256 * p;
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] += w0[(g + 4 * r) + 16 * l15];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (r < p) {
          const ldbl* S = Sq + (p * (p - 1) / 2 + r) * 256;
#pragma unroll
          for (int s = 0; s < 4; s++) t = mfma_na(S[l15 + 16 * (4 * s + g)], w[r][q][s], t);
        }
      }
      d4 o = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) o = mfma(Tq[p * 256 + (4 * s + g) + 16 * l15], t[s], o);
      w[p][q] = o;
    }
  }
  gdbl* cp0 = Y + (long)(cq0 + g) * ld + 2 * l15;
  gdbl* cp1 = Y + (long)(cq1 + g) * ld + 2 * l15;
  d2 cc[NT][4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    cc[0][r] = *reinterpret_cast<const gd2*>(cp0 + (long)(4 * r) * ld + 32 * s0);
    if (NT > 1) cc[NT - 1][r] = *reinterpret_cast<const gd2*>(cp1 + (long)(4 * r) * ld + 32 * s0);
  }
  for (int s = s0; s < s1; s++) {
    vreg = st.load(s + 1, s1 - 1);
    const int sn = min(s + 1, s1 - 1);
    d2 cn[NT][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      cn[0][r] = *reinterpret_cast<const gd2*>(cp0 + (long)(4 * r) * ld + 32 * sn);
      if (NT > 1) cn[NT - 1][r] = *reinterpret_cast<const gd2*>(cp1 + (long)(4 * r) * ld + 32 * sn);
    }
    lds_barrier();
    const ldbl* vb = Vs + (s & 1) * QR_VS_STAGE;
    if (nt > 0) {
      d4 acc[NT][2];
#pragma unroll
      for (int q = 0; q < NT; q++)
#pragma unroll
        for (int e = 0; e < 2; e++) acc[q][e] = d4{cc[q][0][e], cc[q][1][e], cc[q][2][e], cc[q][3][e]};
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
          const d2 v = *reinterpret_cast<const ld2*>(vb + (16 * p + 4 * s2 + g) * QR_VS_LD + 2 * l15);
          if (NT > 1 && nt > 1) {
#pragma unroll
            for (int e = 0; e < 2; e++) { acc[0][e] = mfma_na(w[p][0][s2], v[e], acc[0][e]); acc[NT - 1][e] = mfma_na(w[p][NT - 1][s2], v[e], acc[NT - 1][e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 2; e++) acc[0][e] = mfma_na(w[p][0][s2], v[e], acc[0][e]);
          }
        }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        *reinterpret_cast<gd2*>(cp0 + (long)(4 * r) * ld + 32 * s) = d2{acc[0][0][r], acc[0][1][r]};
        if (NT > 1 && nt > 1) *reinterpret_cast<gd2*>(cp1 + (long)(4 * r) * ld + 32 * s) = d2{acc[NT - 1][0][r], acc[NT - 1][1][r]};
      }
    }
    st.store(Vs, s + 1, jb, vreg);
#pragma unroll
    for (int q = 0; q < NT; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) cc[q][r] = cn[q][r];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Batched contraction O = S X with separable two-level index maps (the Y1 / Y2 assembly of engine.h, every wave of
// the grid taking 16-column tiles of the output).  grid (nwg, ngemm), 512 threads.
// ------------------------------------------------------------------------------------------------------------------
struct Map2 {
  int32_t d; int64_t s0, s1;                       // f(i) = (i % d) * s0 + (i / d) * s1
  __device__ __forceinline__ long operator()(int i) const { return (long)(i % d) * s0 + (long)(i / d) * s1; }
};
struct GemmDesc {
  const double* S; const double* X; double* O;
  int32_t M, N, K, pad;
  Map2 sro, sco, xro, xco, oro, oco;
};
__global__ void __launch_bounds__(512) k_gemm(const GemmDesc* descs) {
  const GemmDesc D = descs[blockIdx.y];
  if (D.M <= 0 || D.N <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile0 = blockIdx.x * 8 + wave, tstride = gridDim.x * 8;
  if (blockIdx.x * 8 * 16 >= D.N) return;           // whole workgroup beyond the last tile
  gemm_direct<false, false>(D.M, D.N, D.K, (const gdbl*)D.S, D.sro, D.sco, (const gdbl*)D.X, D.xro, D.xco, (gdbl*)D.O,
                            D.oro, D.oco, false, tile0, tstride);
}

// E_xi[(m2 + b*y) + M2*(n2 + bn*y1)] = sum_y2 pyy[y,y1,y2,xi] A2[m2,n2,y2,xi]   (engine.h build_E); grid (nwg, nprob)
struct EDesc {
  const double* A2; const double* pyy; double* E;
  int32_t b, bn, ny, ny1, ny2, q;
};
__global__ void __launch_bounds__(256) k_build_E(const EDesc* descs) {
  const EDesc D = descs[blockIdx.y];
  const int M2 = D.b * D.ny, K2 = D.bn * D.ny1;
  const long tot = (long)M2 * K2 * D.q;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % M2); const long rest = idx / M2; const int kk = (int)(rest % K2); const int xi = (int)(rest / K2);
    const int m2 = i % D.b, y = i / D.b, n2 = kk % D.bn, y1 = kk / D.bn;
    double s = 0.0;
    for (int y2 = 0; y2 < D.ny2; y2++) {
      const double c = D.pyy[y + D.ny * (y1 + D.ny1 * (y2 + D.ny2 * xi))];
      if (c != 0.0) s += c * D.A2[m2 + (long)D.b * (n2 + (long)D.bn * (y2 + D.ny2 * xi))];
    }
    D.E[idx] = s;
  }
}

// zero the padding of Y for this step (rows [rows, rows32) of columns [0, cols), columns [cols, cols16 + 16) over rows32)
// and the max-abs slot; grid (nwg, nprob), 256 threads
__global__ void __launch_bounds__(256) k_zero_pads(const QrProb* probs, AuxLay lay) {
  const QrProb P = probs[blockIdx.y];
  gdbl* Y = (gdbl*)P.Y;
  const long ld = P.ld;
  const int rows32 = (P.rows + 31) & ~31, cols16 = ((P.cols + 15) & ~15) + 16;
  const int padr = rows32 - P.rows;
  const long gtid = (long)blockIdx.x * 256 + threadIdx.x, gsz = (long)gridDim.x * 256;
  if (gtid == 0) *(unsigned long long*)(P.aux + lay.mx) = 0ULL;
  if (padr > 0)
    for (long idx = gtid; idx < (long)padr * P.cols; idx += gsz) Y[(long)(idx / padr) * ld + P.rows + (idx % padr)] = 0.0;
  const long nz = (long)rows32 * (cols16 - P.cols);
  for (long idx = gtid; idx < nz; idx += gsz) Y[(long)(P.cols + idx / rows32) * ld + (idx % rows32)] = 0.0;
}

// max |R| over the upper trapezoid; grid (nwg, nprob), 256 threads
__global__ void __launch_bounds__(256) k_maxabs(const QrProb* probs, AuxLay lay) {
  const QrProb P = probs[blockIdx.y];
  const gdbl* Y = (const gdbl*)P.Y;
  const long ld = P.ld;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  double mx = 0.0;
  for (int m = gw; m < P.cols; m += nw) {
    const int kend = min(P.kmax, m + 1);
    for (int k = lane; k < kend; k += 64) mx = fmax(mx, fabs(Y[(long)m * ld + k]));
  }
  mx = wave_max(mx);
  if (lane == 0 && mx > 0.0) atomicMax((unsigned long long*)(P.aux + lay.mx), (unsigned long long)__double_as_longlong(mx));
}

// Lf_t^T = R / max|R|  as [kmax x cols], ld kmax (rank index fastest); grid (nwg, nprob), 256 threads
struct LfDesc { double* Lf; };
__global__ void __launch_bounds__(256) k_lf_write(const QrProb* probs, const LfDesc* lfd, AuxLay lay) {
  const QrProb P = probs[blockIdx.y];
  const gdbl* Y = (const gdbl*)P.Y;
  gdbl* Lf = (gdbl*)lfd[blockIdx.y].Lf;
  const long ld = P.ld;
  const double mx = P.aux[lay.mx];
  const double inv = (mx > 0.0 && isfinite(mx)) ? 1.0 / mx : 1.0;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  const int kmax = P.kmax;
  for (int m = gw; m < P.cols; m += nw) {
    const gdbl* yc = Y + (long)m * ld;
    gdbl* lc = Lf + (long)kmax * m;
    const int kend = min(kmax, m + 1);
    for (int k = lane; k < kend; k += 64) lc[k] = yc[k] * inv;
    for (int k = kend + lane; k < kmax; k += 64) lc[k] = 0.0;
  }
}

// ==================================================================================================================
// The truncating sweep (sweep 2 of engine.h: the orthogonalize_left!(svd_trunc) half of compress!) in grid-level form,
// for the SVDTrunc rules whose kept rank follows from the dimensions (TruncBond, TruncBondMax), so that the host can
// plan every time step.  Per step: k_build_E, k_gemm (N1), k_gemm (N2), k_absmax + k_scale (the max-abs rescale that the
// reference folds into z), k_zero_pads + k_gemm (M_t^T = Lf^T N_t^T), the batched QR above, k_svd_trunc (one-sided
// Jacobi on the transposed triangular factor, truncation, new core), k_gemm (carry C_t = U^T N_t).
// ==================================================================================================================
struct ScaleDesc { double* A; int64_t n; double* mx; double* logc; double* mx_next; };   // mx_next: the slot of the next step, zeroed here

__global__ void __launch_bounds__(256) k_absmax(const ScaleDesc* descs) {
  const ScaleDesc D = descs[blockIdx.y];
  const gdbl* A = (const gdbl*)D.A;
  double mx = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < D.n; i += (int64_t)gridDim.x * 256) mx = fmax(mx, fabs(A[i]));
  mx = wave_max(mx);
  // NaN compares false everywhere: flag it through the bit pattern of +inf so that k_scale sees a non-finite maximum
  if ((threadIdx.x & 63) == 0) {
    if (!(mx == mx)) mx = __builtin_inf();
    if (mx > 0.0) atomicMax((unsigned long long*)D.mx, (unsigned long long)__double_as_longlong(mx));
  }
}
__global__ void __launch_bounds__(256) k_scale(const ScaleDesc* descs, EngStats* stats) {
  const ScaleDesc D = descs[blockIdx.y];
  gdbl* A = (gdbl*)D.A;
  const double mx = *D.mx;
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.mx_next = 0.0;
  if (!(mx == mx) || isinf(mx)) { if (blockIdx.x == 0 && threadIdx.x == 0) stats->nan_flag = 1; return; }
  if (!(mx > 0.0)) return;
  const double inv = 1.0 / mx;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < D.n; i += (int64_t)gridDim.x * 256) A[i] *= inv;
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.logc += log(mx);
}

// leading dimension of the Jacobi matrix in HBM: columns start on 128-byte lines (wg::jac_pair_hbm16)
__host__ __device__ inline int jac_ld(int m) { return (m + 15) & ~15; }
struct SvdDesc {
  const double* Mt; double* JA; double* U; double* core; int32_t* obond;   // obond: the output train's bond table
  int32_t ldM, r1, Rr, kc, kp, t, L, kind, mprime, cap_out;
  double* scal;    // per-problem scalars of the multi-launch Jacobi: [3] ||R||_F^2, [4] worst cos^2 of the sweep (bits), [5] sweeps, [6] done, [7] active columns
  int32_t* act;    // block Jacobi: the columns still in the tournament (numerically null ones leave it, as in wg::jacobi_rsv); may be null (k_jac_round)
};
__device__ __forceinline__ void atomic_max_pos(unsigned long long* addr, double v) { atomicMax(addr, (unsigned long long)__double_as_longlong(v)); }

// grid (nprob), 512 threads; dynamic LDS: 32 + Rr doubles + Rr ints
// mode 0: everything in one launch (Jacobi inside the workgroup); 1: build JA only (the Jacobi then runs as k_jac_round /
// k_jac_check launches over the grid); 2: everything after the Jacobi
__global__ void __launch_bounds__(512) k_svd_trunc(const SvdDesc* descs, EngStats* stats, int mode) {
  const SvdDesc D = descs[blockIdx.x];
  extern __shared__ __attribute__((aligned(16))) double dyn_[];
  ldbl* red = (ldbl*)dyn_;
  ldbl* sig = red + 32;
  typedef __attribute__((address_space(3))) int lint_t;
  lint_t* ord = (lint_t*)(sig + D.Rr);
  const int tid = threadIdx.x;
  const int Rr = D.Rr, r1 = D.r1, kc = D.kc;
  const int k2 = min(r1, Rr);
  const int ldJ = jac_ld(Rr);
  const gdbl* Mt = (const gdbl*)D.Mt;
  gdbl* JA = (gdbl*)D.JA;
  double fro2 = 0.0;
  if (mode != 2) {
    for (int idx = tid; idx < k2 * Rr; idx += 512) {
      const int r = idx % Rr, c = idx / Rr;            // JA[r, c] = R2[c, r]
      const double v = (r >= c) ? Mt[c + (int64_t)D.ldM * r] : 0.0;
      JA[r + (int64_t)ldJ * c] = v;
      fro2 += v * v;
    }
    fro2 = wg_sum(fro2, red);
    __syncthreads();
  }
  if (mode == 1) {
    if (tid == 0) { D.scal[3] = fro2; D.scal[4] = 0.0; D.scal[5] = 0.0; D.scal[6] = (k2 < 2) ? 1.0 : 0.0; D.scal[7] = (double)k2; }
    if (D.act) for (int c = tid; c < k2; c += 512) D.act[c] = c;
    return;
  }
  int sw;
  if (mode == 0) sw = jacobi_rsv(JA, ldJ, Rr, k2, nullptr, 0, red, ord, 60);
  else { fro2 = D.scal[3]; sw = (D.scal[6] != 0.0) ? (int)D.scal[5] : -1; }
  if (tid == 0) {
    if (sw < 0) stats->jacobi_fail = 1;
    atomicAdd(&stats->jac_sweeps, (unsigned long long)(sw < 0 ? 60 : sw));
    atomicAdd(&stats->jac_calls, 1ULL);
  }
  const double nul2 = 1e-28 * fro2;
  for (int c = tid; c < Rr; c += 512) sig[c] = 0.0;
  __syncthreads();
  for (int c = tid; c < k2; c += 512) {
    double s = 0.0;
    for (int r = 0; r < Rr; r++) { const double v = JA[r + (int64_t)ldJ * c]; s += v * v; }
    const bool ok = s > nul2 && s > 0.0;
    const double sg = ok ? sqrt(s) : 0.0, inv = ok ? 1.0 / sqrt(s) : 0.0;
    for (int r = 0; r < Rr; r++) JA[r + (int64_t)ldJ * c] *= inv;
    sig[c] = sg;
  }
  __syncthreads();
  for (int c = tid; c < k2; c += 512) {
    const double sc = sig[c];
    int rank = 0;
    for (int j = 0; j < k2; j++) { const double sj = sig[j]; rank += (sj > sc) || (sj == sc && j < c); }
    ord[rank] = c;
  }
  __syncthreads();
  const int len = min(Rr, r1);
  int kp = min(len, D.mprime);
  if (kp < 1) kp = 1;
  if (kp > D.cap_out) { kp = D.cap_out; if (tid == 0) stats->capacity_flag = 1; }
  // kp == D.kp by construction (the host planned the buffers with the same rule)
  if (D.kind == MPBP_TRUNC_BOND_MAX && tid == 0) {
    double tot = 0.0, dropped = 0.0;
    for (int j = 0; j < len; j++) { const double s = sig[ord[j]]; tot += s * s; if (j >= kp) dropped += s * s; }
    if (tot > 0.0) atomic_max_pos(&stats->maxerr_bits, sqrt(dropped / tot));
  }
  gdbl* oc = (gdbl*)D.core;
  gdbl* U = (gdbl*)D.U;
  for (int idx = tid; idx < Rr * kp; idx += 512) {
    const int row = idx % Rr, k2i = idx / Rr;
    const int k = row % kc, s = row / kc;
    const double v = JA[row + (int64_t)ldJ * ord[k2i]];
    oc[(int64_t)k + (int64_t)kc * (k2i + (int64_t)kp * s)] = v;       // core t: [kc, kp, s]
    U[idx] = v;                                                        // sorted left singular vectors, ld Rr
  }
  if (tid == 0) {
    D.obond[D.t + 1] = kp;
    if (D.t == 0) { D.obond[0] = 1; D.obond[D.L] = 1; }
  }
}

// One round of the round-robin tournament of the one-sided Jacobi (wg::jacobi_rsv's rotations, thresholds and pairing)
// over the grid, for factors too large for one workgroup to rotate quickly (configs[2]: 660 x 660): 32 lanes per column
// pair, the two columns in registers between the dot products and the rotation (<= 1024 rows).  A kernel boundary
// separates the rounds; k_jac_check closes a sweep (convergence test of jacobi_rsv: the largest cos^2 met < 1e-16).
// grid (ceil(pairs / 16), nprob), 512 threads.
__global__ void __launch_bounds__(512) k_jac_round(const SvdDesc* descs, int round) {
  const SvdDesc D = descs[blockIdx.y];
  if (D.scal[6] != 0.0) return;                      // converged
  const int m = D.Rr, n = min(D.r1, D.Rr);
  const int ne = (n + 1) & ~1;
  if (round >= ne - 1) return;
  const int pi = blockIdx.x * 16 + (threadIdx.x >> 5), sub = threadIdx.x & 31;
  if (pi >= ne / 2) return;
  int p, q;
  if (pi == 0) { p = ne - 1; q = round; }
  else {
    p = round + pi; p -= (p >= ne - 1) ? (ne - 1) : 0;
    q = round + (ne - 1) - pi; q -= (q >= ne - 1) ? (ne - 1) : 0;
  }
  if (p > q) { const int t_ = p; p = q; q = t_; }
  if (q >= n) return;                                // the dummy player of an odd tournament
  const int ldJ = jac_ld(m);
  gdbl* ap = (gdbl*)D.JA + (int64_t)ldJ * p;
  gdbl* aq = (gdbl*)D.JA + (int64_t)ldJ * q;
  double x[32], y[32];
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int r = sub + 32 * i;
    const bool ok = r < m;
    const int rc = ok ? r : sub;
    const double xv = ap[rc], yv = aq[rc];
    x[i] = ok ? xv : 0.0; y[i] = ok ? yv : 0.0;
  }
  double al = 0, be = 0, ga = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) { al += x[i] * x[i]; be += y[i] * y[i]; ga += x[i] * y[i]; }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) { al += __shfl_xor(al, o, 32); be += __shfl_xor(be, o, 32); ga += __shfl_xor(ga, o, 32); }
  const double nul = 1e-28 * D.scal[3], tol = 1e-15;
  if (!(ga * ga > (tol * tol) * (al * be) && al > nul && be > nul)) return;
  double c, s;
  jac_cs(al, be, ga, c, s);
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const int r = sub + 32 * i;
    if (r < m) { ap[r] = c * x[i] - s * y[i]; aq[r] = s * x[i] + c * y[i]; }
  }
  if (sub == 0) atomicMax((unsigned long long*)(D.scal + 4), (unsigned long long)__double_as_longlong(ga * ga / (al * be)));
}
__global__ void k_jac_check(const SvdDesc* descs, int nprob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nprob) return;
  const SvdDesc D = descs[i];
  if (D.scal[6] != 0.0) return;
  D.scal[5] += 1.0;
  if (D.scal[4] < 1e-16) D.scal[6] = 1.0;
  D.scal[4] = 0.0;
}

// ------------------------------------------------------------------------------------------------------------------
// Two-level (block) one-sided Jacobi: the columns of the factor are cut into blocks of `nb`; a launch is one round of the
// round-robin tournament of the BLOCKS, every workgroup takes one block pair, holds its 2 nb columns in LDS (odd leading
// dimension), runs ONE sweep over all pairs of those columns there - same rotation, thresholds and null test as
// wg::jacobi_rsv - and writes them back.  The one-workgroup form keeps a 160 x 160 factor in HBM and touches every column
// two to three times per round through the CU's L1 (~3 us per round, 159 rounds per sweep, ONE CU per problem whatever the
// batch); here a round is ~0.6 us of LDS traffic, a sweep is (blocks - 1) launches, and a problem occupies blocks / 2 CUs.
// Pairs inside a block meet in every round of the block tournament (more rotations per sweep, fewer sweeps).
// k_jac_check closes a sweep as for k_jac_round.  grid (block pairs, nprob), 512 threads, dynamic LDS (m | 1) * 2 nb + 32 doubles.
// ------------------------------------------------------------------------------------------------------------------
template <int LP>     // lanes per column pair: 8 or 16 (one DPP row)
__device__ __forceinline__ double jac_lds_pair(ldbl* ap, ldbl* aq, int sub, int m, double tol, double nul) {
  constexpr int JC = 12;                     // rows per lane held in registers at a time
  const int rpl = (m + LP - 1) / LP;
  double al = 0, be = 0, ga = 0;
  double x[JC], y[JC];
  auto loadc = [&](int i0) {
#pragma unroll
    for (int i = 0; i < JC; i++) {
      const int r = sub + (i0 + i) * LP;
      const bool ok = r < m;
      const int rc = ok ? r : sub;
      const double xv = ap[rc], yv = aq[rc];
      x[i] = ok ? xv : 0.0; y[i] = ok ? yv : 0.0;
    }
  };
  for (int i0 = 0; i0 < rpl; i0 += JC) {
    loadc(i0);
#pragma unroll
    for (int i = 0; i < JC; i++) { al += x[i] * x[i]; be += y[i] * y[i]; ga += x[i] * y[i]; }
  }
  if (LP == 16) { al += dpp64<0x140>(al); be += dpp64<0x140>(be); ga += dpp64<0x140>(ga); }      // row_mirror: lanes i, 15 - i
  al += dpp64<0x141>(al); be += dpp64<0x141>(be); ga += dpp64<0x141>(ga);
  al += dpp64<0x4E>(al); be += dpp64<0x4E>(be); ga += dpp64<0x4E>(ga);
  al += dpp64<0xB1>(al); be += dpp64<0xB1>(be); ga += dpp64<0xB1>(ga);
  if (!(ga * ga > (tol * tol) * (al * be) && al > nul && be > nul)) return 0.0;
  double c, s;
  jac_cs(al, be, ga, c, s);
  for (int i0 = 0; i0 < rpl; i0 += JC) {
    if (rpl > JC) loadc(i0);
#pragma unroll
    for (int i = 0; i < JC; i++) {
      const int r = sub + (i0 + i) * LP;
      if (r < m) { ap[r] = c * x[i] - s * y[i]; aq[r] = s * x[i] + c * y[i]; }
    }
  }
  return ga * ga / (al * be);
}
// one sweep over all pairs of the nc columns of an LDS-resident m x nc matrix (512 threads); the largest cos^2 met, per lane
template <int LP>
__device__ __forceinline__ double jac_lds_sweep(ldbl* A, int lda, int m, int nc, double tol, double nul) {
  const int tid = threadIdx.x, sub = tid & (LP - 1), grp = tid / LP;
  const int ne = (nc + 1) & ~1, npairs = ne / 2;
  double worst = 0.0;
  for (int round = 0; round < ne - 1; round++) {
    for (int pb = 0; pb < npairs; pb += 512 / LP) {
      const int pi = pb + grp;
      if (pi < npairs) {
        int p, q;
        if (pi == 0) { p = ne - 1; q = round; }
        else {
          p = round + pi; p -= (p >= ne - 1) ? (ne - 1) : 0;
          q = round + (ne - 1) - pi; q -= (q >= ne - 1) ? (ne - 1) : 0;
        }
        if (p > q) { const int t_ = p; p = q; q = t_; }
        if (q < nc) worst = fmax(worst, jac_lds_pair<LP>(A + (long)lda * p, A + (long)lda * q, sub, m, tol, nul));
      }
    }
    __syncthreads();
  }
  return worst;
}
__global__ void __launch_bounds__(512) k_jac_block(const SvdDesc* descs, int round, int nb) {
  const SvdDesc D = descs[blockIdx.y];
  if (D.scal[6] != 0.0) return;                      // converged
  const int m = D.Rr, n = (int)D.scal[7];            // the active columns, through D.act
  if (n < 2) return;
  const int nblk = (n + nb - 1) / nb, ne = (nblk + 1) & ~1;
  const int pi = blockIdx.x;
  if (pi >= ne / 2) return;
  int bp, bq;
  if (ne == 2) { if (round > 0) return; bp = 0; bq = 1; }
  else {
    if (round >= ne - 1) return;
    if (pi == 0) { bp = ne - 1; bq = round; }
    else {
      bp = round + pi; bp -= (bp >= ne - 1) ? (ne - 1) : 0;
      bq = round + (ne - 1) - pi; bq -= (bq >= ne - 1) ? (ne - 1) : 0;
    }
    if (bp > bq) { const int t_ = bp; bp = bq; bq = t_; }
  }
  // a lone block (nblk == 1) still needs its inner sweep; the dummy player of an odd tournament sits out with its partner
  // EXCEPT that the partner's inner pairs are covered by its other rounds
  const bool lone = nblk == 1;
  if (!lone && bq >= nblk) return;
  extern __shared__ __attribute__((aligned(16))) double jb_lds_[];
  ldbl* A = (ldbl*)jb_lds_;
  const int lda = m | 1;
  const int ldJ = jac_ld(m);
  gdbl* JA = (gdbl*)D.JA;
  const int c0p = bp * nb, ncp = min(nb, n - c0p);
  const int c0q = lone ? 0 : bq * nb, ncq = lone ? 0 : min(nb, n - c0q);
  const int nc = ncp + ncq;
  const int tid = threadIdx.x;
  // columns -> LDS (two rows per access: the HBM columns start on 128-byte lines, m is even on this path or the tail is scalar)
  const int32_t* act = D.act;
  for (int idx = tid; idx < nc * m; idx += 512) {
    const int c = idx / m, r = idx - c * m;
    const int gc = act[c < ncp ? c0p + c : c0q + (c - ncp)];
    A[r + (long)lda * c] = JA[r + (int64_t)ldJ * gc];
  }
  __syncthreads();
  const double nul = 1e-28 * D.scal[3], tol = 1e-15;
  double worst;
  if ((nc + 1) / 2 * 16 <= 512) worst = jac_lds_sweep<16>(A, lda, m, nc, tol, nul);
  else worst = jac_lds_sweep<8>(A, lda, m, nc, tol, nul);
  for (int idx = tid; idx < nc * m; idx += 512) {
    const int c = idx / m, r = idx - c * m;
    const int gc = act[c < ncp ? c0p + c : c0q + (c - ncp)];
    JA[r + (int64_t)ldJ * gc] = A[r + (long)lda * c];
  }
  worst = wave_max(worst);
  if ((tid & 63) == 0 && worst > 0.0) atomicMax((unsigned long long*)(D.scal + 4), (unsigned long long)__double_as_longlong(worst));
}

// Closes a sweep of the block Jacobi: the convergence test of wg::jacobi_rsv (the largest cos^2 met < 1e-16), then its
// deflation - a column whose squared norm is below 1e-28 ||A||_F^2 is numerically null, never rotates again and leaves
// the tournament (the factors here have singular values over twenty decades: most columns are gone after two sweeps, and
// the later sweeps run over a fraction of the blocks).  grid (nprob), 512 threads.
__global__ void __launch_bounds__(512) k_jac_deflate(const SvdDesc* descs) {
  const SvdDesc D = descs[blockIdx.x];
  if (D.scal[6] != 0.0) return;
  __shared__ int keep_[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = D.Rr, n = (int)D.scal[7];
  const bool conv = D.scal[4] < 1e-16;
  if (!conv) {
    const int ldJ = jac_ld(m);
    const gdbl* JA = (const gdbl*)D.JA;
    const double nul = 1e-28 * D.scal[3];
    for (int i = wave; i < n; i += 8) {                         // a wave per column
      const gdbl* ac = JA + (int64_t)ldJ * D.act[i];
      double s2 = 0.0;
      for (int r = lane; r < m; r += 64) s2 += ac[r] * ac[r];
      s2 = wave_sum(s2);
      if (lane == 0) keep_[i] = (s2 > nul) ? 1 : 0;
    }
  }
  __syncthreads();
  if (tid == 0) {
    D.scal[5] += 1.0;
    if (conv) D.scal[6] = 1.0;
    else {
      int w = 0;
      for (int i = 0; i < n; i++) if (keep_[i]) D.act[w++] = D.act[i];
      D.scal[7] = (double)w;
      if (w < 2) D.scal[6] = 1.0;
    }
    D.scal[4] = 0.0;
  }
}
// pending[0] = problems of the batch whose block Jacobi has not converged, pending[1] = the most active columns among them
__global__ void k_jac_pending2(const SvdDesc* descs, int nprob, int* pending) {
  __shared__ int cnt, mx;
  if (threadIdx.x == 0) { cnt = 0; mx = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < nprob; i += blockDim.x)
    if (descs[i].scal[6] == 0.0) { atomicAdd(&cnt, 1); atomicMax(&mx, (int)descs[i].scal[7]); }
  __syncthreads();
  if (threadIdx.x == 0) { pending[0] = cnt; pending[1] = mx; }
}

// *pending = number of problems whose Jacobi has not converged yet (the host polls it every few sweeps); one workgroup
__global__ void k_jac_pending(const SvdDesc* descs, int nprob, int* pending) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < nprob; i += blockDim.x) mine += descs[i].scal[6] == 0.0 ? 1 : 0;
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) *pending = cnt;
}

// last core of the truncating sweep: [kc, 1, s] = N_t[(k,s), 0]; grid (nprob)
struct LastDesc { const double* Nt; double* core; int32_t* obond; int32_t Rr, L; };
__global__ void __launch_bounds__(256) k_lastcore(const LastDesc* descs) {
  const LastDesc D = descs[blockIdx.x];
  for (int idx = threadIdx.x; idx < D.Rr; idx += 256) D.core[idx] = D.Nt[idx];
  if (threadIdx.x == 0) { D.obond[0] = 1; D.obond[D.L] = 1; }
}

// normalize_eachmatrix! of the output train + its log z (engine.h tail); grid (nprob), 256 threads
struct NormDesc {
  double* out; const int32_t* obond; int64_t ostride; const double* logz1; const double* logz2; const double* logc; double* ologz;
  int32_t sphys, L;
};
__global__ void __launch_bounds__(256) k_normalize_out(const NormDesc* descs, EngStats* stats) {
  const NormDesc D = descs[blockIdx.x];
  __shared__ double red_[8];
  ldbl* red = (ldbl*)red_;
  const int tid = threadIdx.x;
  double logz = (D.logz1 ? *D.logz1 : 0.0) + (D.logz2 ? *D.logz2 : 0.0) - *D.logc;
  for (int tp = 0; tp < D.L; tp++) {
    const int n = D.obond[tp] * D.obond[tp + 1] * D.sphys;
    gdbl* oc = (gdbl*)D.out + (int64_t)tp * D.ostride;
    double mx = 0.0;
    for (int idx = tid; idx < n; idx += 256) mx = fmax(mx, fabs(oc[idx]));
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (mx > 0.0 && isfinite(mx)) {
      const double inv = 1.0 / mx;
      for (int idx = tid; idx < n; idx += 256) oc[idx] *= inv;
      logz -= log(mx);
    }
  }
  if (tid == 0) {
    *D.ologz = logz;
    atomicAdd(&stats->n_compress, 1ULL);
    if (!(logz == logz)) stats->nan_flag = 1;
  }
}

// bond tables of both operands of every problem -> one contiguous buffer [nprob][2][L+1]; Lf_L = 1; grid (nprob)
struct BondSrc { const int32_t* b1; const int32_t* b2; };
__global__ void __launch_bounds__(64) k_gather_bonds(const BondSrc* src, int32_t* out, int L) {
  const BondSrc S = src[blockIdx.x];
  int32_t* o = out + (long)blockIdx.x * 2 * (L + 1);
  for (int t = threadIdx.x; t <= L; t += 64) { o[t] = S.b1[t]; o[L + 1 + t] = S.b2[t]; }
}
struct SetOne { double* p; };
__global__ void k_set_one(const SetOne* s, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) *s[i].p = 1.0;
}

}  // namespace v2
