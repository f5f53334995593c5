// Workgroup-cooperative fp64 building blocks for gfx950 (CDNA4): one 512-thread workgroup (8 wave64)
// owns one dense problem.  All matrices are column-major.
//
//   wg_gemm      O = S * X with MFMA v_mfma_f64_16x16x4_f64, operands addressed through separable
//                (row-offset + column-offset) index maps so that tensor contractions over composite
//                indices need no explicit transposes/reshapes in HBM; tiles staged through LDS.
//   wg_qr_r      in-place blocked Householder QR keeping only R (panel width 16; trailing update
//                V, T, C all through MFMA straight from global/L2 with 128-B-line aligned fragments).
//   wg_jacobi_rsv one-sided (Hestenes) Jacobi on a small matrix: right singular vectors + values.
//
// f64 MFMA fragment maps (cdna_hip_programming.md section 3, "f64 MFMA does NOT use these maps"):
//   A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15];
//   C/D: reg r of lane l is C[row = (l>>4) + 4r][col = l&15].
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define WG_THREADS 512
#define WG_WAVES 8

namespace wg {

constexpr int GM_MB = 96;    // rows per M block (6 MFMA row tiles)
constexpr int GM_NT = 128;   // columns per N chunk (one 16-col tile per wave)
constexpr int GM_KC = 16;    // K per LDS chunk
constexpr int GM_LDA = 112;  // LDS leading dims: == 16 (mod 32) doubles so that the two 16-lane runs of a
constexpr int GM_LDB = 144;  // ds_read_b64 half-wave land on disjoint banks
constexpr int GM_LDS_DOUBLES = GM_KC * (GM_LDA + GM_LDB);   // 4096 doubles = 32 KiB

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// O(i,j) (+)= sum_k S(i,k) X(k,j),  i<M, j<N, k<K.
//   S(i,k) = Sp[sro(i) + sco(k)],  X(k,j) = Xp[xro(k) + xco(j)],  O(i,j) = Op[oro(i) + oco(j)].
// `xkfast`: X is contiguous along k (else along j) - only picks the coalesced staging pattern.
// `lds` must provide GM_LDS_DOUBLES doubles.  Ends with a __syncthreads().
template <class SRO, class SCO, class XRO, class XCO, class ORO, class OCO>
__device__ void gemm(int M, int N, int K, const double* Sp, SRO sro, SCO sco, const double* Xp, XRO xro,
                     XCO xco, bool xkfast, double* Op, ORO oro, OCO oco, bool accumulate, double* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  double* As = lds;
  double* Bs = lds + GM_KC * GM_LDA;
  for (int mb = 0; mb < M; mb += GM_MB) {
    const int mrows = min(GM_MB, M - mb);
    const int ntm = (mrows + 15) >> 4;
    // staging map for S: i = tid & 127 (only < 96 used), kk = (tid >> 7) + 4e
    const int si = tid & 127;
    const bool s_act = si < ntm * 16;
    const bool s_in = si < mrows;
    const long s_ro = s_in ? (long)sro(mb + si) : 0;
    for (int n0 = 0; n0 < N; n0 += GM_NT) {
      const int ncols = min(GM_NT, N - n0);
      // staging map for X
      long x_co[4];
      bool x_in[4];
      if (xkfast) {      // kk = tid & 15, j = (tid >> 4) + 32e
#pragma unroll
        for (int e = 0; e < 4; e++) {
          int j = (tid >> 4) + 32 * e;
          x_in[e] = j < ncols;
          x_co[e] = x_in[e] ? (long)xco(n0 + j) : 0;
        }
      } else {           // j = tid & 127, kk = (tid >> 7) + 4e
        int j = tid & 127;
        x_in[0] = j < ncols;
        x_co[0] = x_in[0] ? (long)xco(n0 + j) : 0;
      }
      d4 acc[6];
#pragma unroll
      for (int t = 0; t < 6; t++) acc[t] = d4{0, 0, 0, 0};
      const bool w_act = wave * 16 < ncols;
      for (int k0 = 0; k0 < K; k0 += GM_KC) {
        // ---- stage S tile: As[kk][i]
        if (s_act) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            int kk = (tid >> 7) + 4 * e;
            double v = 0.0;
            if (s_in && k0 + kk < K) v = Sp[s_ro + (long)sco(k0 + kk)];
            As[kk * GM_LDA + si] = v;
          }
        }
        // ---- stage X tile: Bs[kk][j]
        if (xkfast) {
          int kk = tid & 15;
          bool kin = k0 + kk < K;
          long r = kin ? (long)xro(k0 + kk) : 0;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            int j = (tid >> 4) + 32 * e;
            double v = 0.0;
            if (kin && x_in[e]) v = Xp[r + x_co[e]];
            Bs[kk * GM_LDB + j] = v;
          }
        } else {
          int j = tid & 127;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            int kk = (tid >> 7) + 4 * e;
            double v = 0.0;
            if (x_in[0] && k0 + kk < K) v = Xp[(long)xro(k0 + kk) + x_co[0]];
            Bs[kk * GM_LDB + j] = v;
          }
        }
        __syncthreads();
        if (w_act) {
#pragma unroll
          for (int ks = 0; ks < GM_KC / 4; ks++) {
            const int krow = ks * 4 + g;
            const double b = Bs[krow * GM_LDB + wave * 16 + l15];
#pragma unroll
            for (int t = 0; t < 6; t++) {
              if (t < ntm) {
                const double a = As[krow * GM_LDA + t * 16 + l15];
                acc[t] = mfma(a, b, acc[t]);
              }
            }
          }
        }
        __syncthreads();
      }
      // ---- write out
      if (w_act) {
        const int col = wave * 16 + l15;
        if (col < ncols) {
          const long oc = (long)oco(n0 + col);
#pragma unroll
          for (int t = 0; t < 6; t++) {
            if (t < ntm) {
#pragma unroll
              for (int r = 0; r < 4; r++) {
                int row = t * 16 + g + 4 * r;
                if (row < mrows) {
                  double* p = Op + (long)oro(mb + row) + oc;
                  double v = acc[t][r];
                  if (accumulate) v += *p;
                  *p = v;
                }
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// workgroup reductions through LDS (red must hold WG_WAVES*NV doubles)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// all threads receive the sum; red: >= WG_WAVES doubles; contains 2 barriers
__device__ __forceinline__ double wg_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0;
#pragma unroll
  for (int w = 0; w < WG_WAVES; w++) s += red[w];
  return s;
}
__device__ __forceinline__ double wg_max(double v, double* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < WG_WAVES; w++) s = fmax(s, red[w]);
  return s;
}

// max |x| over a strided 2D region (rows x cols, leading dim ld); all threads get it
__device__ inline double wg_maxabs(const double* A, long ld, int rows, int cols, double* red) {
  double m = 0.0;
  const long tot = (long)rows * cols;
  for (long idx = threadIdx.x; idx < tot; idx += WG_THREADS) {
    int r = (int)(idx % rows);
    long c = idx / rows;
    m = fmax(m, fabs(A[r + ld * c]));
  }
  return wg_max(m, red);
}

// ---------------------------------------------------------------------------------------------
// Blocked Householder QR, R only, in place.
//   Y: column-major, leading dim ld (multiple of 16, >= rows rounded up to 16); the padding rows
//      [rows, ld16) and padding columns [cols, cols16) MUST be zero on entry (cols16 = cols up to 16).
//   On exit the upper trapezoid Y[0:min(rows,cols), 0:cols] holds R (everything below is garbage).
//   lds: QR_LDS_DOUBLES doubles.
// Panel factorisation is LAPACK dgeqr2/dlarft (level 2, in global memory, workgroup reductions);
// the trailing update C -= V (T^T (V^T C)) runs per 16-column tile inside one wave, entirely in MFMA
// registers: W0 = V^T C (k-permuted 32-B row fragments, full 128-B lines), W = T^T W0 (accumulator
// fed back as B operand), C^T -= W^T V^T (so that every C access is a full line).
// ---------------------------------------------------------------------------------------------
constexpr int QR_NB = 16;
constexpr int QR_LDS_DOUBLES = WG_WAVES * 16 + 16 * 16 + 16 + 32;

__device__ void qr_r(double* Y, long ld, int rows, int cols, double* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  double* red = lds;                    // [WG_WAVES*16]
  double* Ts = lds + WG_WAVES * 16;     // [16*16] T, column-major Ts[i + 16*j]
  double* tau = Ts + 256;               // [16]
  double* bc = tau + 16;                // broadcast scratch [32]
  const int kmax = min(rows, cols);
  const int rows16 = (rows + 15) & ~15;
  const int ctiles = (cols + 15) >> 4;

  for (int j0 = 0; j0 < kmax; j0 += QR_NB) {
    const int nb = min(QR_NB, kmax - j0);
    // zero T
    if (tid < 256) Ts[tid] = 0.0;
    if (tid < 16) tau[tid] = 0.0;
    __syncthreads();
    // ------------------------------------------------------------------ panel factorisation
    for (int jj = 0; jj < nb; jj++) {
      const int col = j0 + jj;
      double* x = Y + (long)col * ld;
      // norm of x[col+1:rows]
      double ss = 0.0;
      for (int r = col + 1 + tid; r < rows; r += WG_THREADS) { double v = x[r]; ss += v * v; }
      ss = wg_sum(ss, red);
      const double alpha = x[col];
      double beta, tj, scale;
      if (ss == 0.0) { beta = alpha; tj = 0.0; scale = 0.0; }
      else {
        beta = -copysign(sqrt(alpha * alpha + ss), alpha);
        tj = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      __syncthreads();   // everyone has read alpha
      // scale v in place, v[col] implicitly 1; store beta on the diagonal
      for (int r = col + 1 + tid; r < rows; r += WG_THREADS) x[r] *= scale;
      if (tid == 0) { x[col] = beta; tau[jj] = tj; }
      __syncthreads();
      // dots of v with the other 15 panel columns: c>jj -> w_c (apply), c<jj -> z_c (T factor)
      double part[QR_NB];
#pragma unroll
      for (int c = 0; c < QR_NB; c++) part[c] = 0.0;
      for (int r = col + 1 + tid; r < rows; r += WG_THREADS) {
        const double v = x[r];
#pragma unroll
        for (int c = 0; c < QR_NB; c++)
          if (c != jj && c < nb) part[c] += v * Y[(long)(j0 + c) * ld + r];
      }
#pragma unroll
      for (int c = 0; c < QR_NB; c++) part[c] = wave_sum(part[c]);
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < QR_NB; c++) red[wave * 16 + c] = part[c];
      }
      __syncthreads();
      if (tid < 16) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < WG_WAVES; w++) s += red[w * 16 + tid];
        // add the row `col` term (v[col] = 1)
        if (tid < nb && tid != jj) s += Y[(long)(j0 + tid) * ld + col];
        bc[tid] = s;
      }
      __syncthreads();
      // apply H to the remaining panel columns
      for (int c = jj + 1; c < nb; c++) {
        const double w = tj * bc[c];
        double* yc = Y + (long)(j0 + c) * ld;
        for (int r = col + 1 + tid; r < rows; r += WG_THREADS) yc[r] -= w * x[r];
        if (tid == 0) yc[col] -= w;
      }
      // T(0:jj, jj) = -tau * T(0:jj,0:jj) * z(0:jj)
      if (tid < jj) {
        double s = 0.0;
        for (int i2 = tid; i2 < jj; i2++) s += Ts[tid + 16 * i2] * bc[i2];
        Ts[tid + 16 * jj] = -tj * s;
      }
      if (tid == 0) Ts[jj + 16 * jj] = tj;
      __syncthreads();
    }
    // ------------------------------------------------------------------ trailing update
    const int ct0 = (j0 + nb + 15) >> 4;         // first full column tile right of the panel ...
    // ... but columns j0+nb .. 16*ct0 (same tile as the panel when nb < 16) also need the update:
    // nb < 16 only happens for the last panel (kmax reached); then remaining columns [j0+nb, cols)
    // may start inside the panel's own tile.  Handle generally: tiles start at column j0+nb.
    const int cstart = j0 + nb;
    const int ntl = (cols - cstart + 15) / 16;   // tiles of 16 columns starting at cstart (may be 0)
    for (int tl = wave; tl < ntl; tl += WG_WAVES) {
      const int cb = cstart + tl * 16;           // first column of this tile
      // columns beyond `cols` inside the tile are padding (zero) as long as cb+15 < cols16 + ... ;
      // guard explicitly: lane's column valid?
      const bool cval_row = (cb + l15) < cols;   // for W0 (column index = l15)
      // ---- phase A: W0[i][c] = sum_r V[r][i] C[r][c]
      d4 w0 = d4{0, 0, 0, 0};
      for (int rb = j0; rb < rows16; rb += 16) {
        const int rr = rb + 4 * g;               // this lane's 4 consecutive rows
        d4 v = *reinterpret_cast<const d4*>(Y + (long)(j0 + l15) * ld + rr);
        d4 c = cval_row ? *reinterpret_cast<const d4*>(Y + (long)(cb + l15) * ld + rr) : d4{0, 0, 0, 0};
        if (rb == j0) {   // unit lower-trapezoidal head of V
#pragma unroll
          for (int e = 0; e < 4; e++) {
            int rho = 4 * g + e;
            v[e] = (rho > l15) ? v[e] : ((rho == l15) ? 1.0 : 0.0);
          }
        }
        if (l15 >= nb) v = d4{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) w0 = mfma(v[e], c[e], w0);
      }
      // ---- phase B: W = T^T W0   (A[i'][k] = T[k][i'], B k-step s = reg s of W0)
      d4 w = d4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < 4; s++) w = mfma(Ts[(4 * s + g) + 16 * l15], w0[s], w);
      // ---- phase C: C^T[c][r] -= sum_k W[k][c] V[r][k]
      for (int rb = j0; rb < rows16; rb += 16) {
        d4 acc;
        const int row = rb + l15;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int cc = cb + g + 4 * r;
          acc[r] = (cc < cols) ? Y[(long)cc * ld + row] : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int k = 4 * s + g;               // reflector index
          double v = Y[(long)(j0 + k) * ld + row];
          if (rb == j0) v = (l15 > k) ? v : ((l15 == k) ? 1.0 : 0.0);
          if (k >= nb) v = 0.0;
          acc = mfma(-w[s], v, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int cc = cb + g + 4 * r;
          if (cc < cols && row < rows) Y[(long)cc * ld + row] = acc[r];
        }
      }
    }
    (void)ct0; (void)ctiles;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// One-sided Jacobi (Hestenes): A (m x n, ld lda) -> A V = U Sigma; V (n x n, ld ldv) accumulated from I.
// Column norms of the final A are the singular values (unsorted).  8 lanes per column pair.
// Returns number of sweeps used (all threads), or -1 if not converged in maxsweeps.
// n is padded to even internally via a virtual zero column (skipped).  `red`: >= 16 doubles.
// A and V may live in LDS or global memory (generic pointers).
// ---------------------------------------------------------------------------------------------
__device__ int jacobi_rsv(double* A, int lda, int m, int n, double* V, int ldv, double* red, int maxsweeps) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < n * n; idx += WG_THREADS) {
    int r = idx % n, c = idx / n;
    V[r + (long)ldv * c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  if (n < 2) return 0;
  // columns whose norm falls below 1e-14 ||A||_F are numerically null (they only carry the rounding noise
  // of the preceding QR); rotating them against anything never converges and never matters
  double fro2 = 0.0;
  for (int idx = tid; idx < m * n; idx += WG_THREADS) { double v = A[(idx % m) + (long)lda * (idx / m)]; fro2 += v * v; }
  fro2 = wg_sum(fro2, red);
  const double nul = 1e-28 * fro2;
  const int ne = (n + 1) & ~1;          // even number of "players"
  const int npairs = ne / 2;
  const int sub = tid & 7;              // lane inside the 8-lane pair group
  const int grp = tid >> 3;             // 64 groups per pass
  const double tol = 1e-15;
  int sweep = 0;
  bool conv = false;
  for (; sweep < maxsweeps; sweep++) {
    int rotated = 0;
    for (int round = 0; round < ne - 1; round++) {
      for (int pb = 0; pb < npairs; pb += WG_THREADS / 8) {
        const int pi = pb + grp;
        if (pi < npairs) {
          // round-robin tournament: player ne-1 fixed, others rotate
          int p, q;
          if (pi == 0) { p = ne - 1; q = round % (ne - 1); }
          else {
            p = (round + pi) % (ne - 1);
            q = (round + (ne - 1) - pi) % (ne - 1);
          }
          if (p > q) { int t_ = p; p = q; q = t_; }
          if (q < n) {
            double* ap = A + (long)lda * p;
            double* aq = A + (long)lda * q;
            double al = 0, be = 0, ga = 0;
            for (int r = sub; r < m; r += 8) { double x = ap[r], y = aq[r]; al += x * x; be += y * y; ga += x * y; }
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {
              al += __shfl_xor(al, o, 64); be += __shfl_xor(be, o, 64); ga += __shfl_xor(ga, o, 64);
            }
            const double lim = tol * sqrt(al * be);
            if (fabs(ga) > lim && al > nul && be > nul) {
              rotated = 1;
              const double zeta = (be - al) / (2.0 * ga);
              const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
              const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
              for (int r = sub; r < m; r += 8) { double x = ap[r], y = aq[r]; ap[r] = c * x - s * y; aq[r] = s * x + c * y; }
              double* vp = V + (long)ldv * p;
              double* vq = V + (long)ldv * q;
              for (int r = sub; r < n; r += 8) { double x = vp[r], y = vq[r]; vp[r] = c * x - s * y; vq[r] = s * x + c * y; }
            }
          }
        }
      }
      __syncthreads();
    }
    // converged if nobody rotated in this sweep
    double any = wg_max((double)rotated, red);
    if (any == 0.0) { conv = true; break; }
  }
  __syncthreads();
  return conv ? sweep + 1 : -1;
}

}  // namespace wg
