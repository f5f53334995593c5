// Device kernels of libmpbp_hip: the compress engine launcher and the HBM-bound scan kernels
// (Pxy application, MPEM3 -> explicit MPEM2 embedding, environment scans for normalize!/marginals,
// pair beliefs).  gfx950 only.
#pragma once
#include "wg_common.h"
#include "engine_types.h"

// ------------------------------------------------------------------------------------------------
// The building blocks and the engine are compiled twice: `v512` - one 512-thread workgroup per CU for the
// MFMA-heavy cavity products - and `v64` - single-wave workgroups, several per CU, for the small problems
// (message finalisation, products with a bond-1 operand, low bond dimensions), whose cost is latency:
// a single wave has no workgroup barrier to wait at and many of them hide each other's latency.
// Engine launcher: persistent workgroups pull problems (sorted by decreasing cost) from a counter.
// ------------------------------------------------------------------------------------------------
#define MPBP_ENGINE_KERNEL(LB)                                                                                    \
  __global__ void LB eng_kernel(const EngProb* probs, int nprob, int* counter, EngCfg cfg, double* scratch,       \
                                EngStats* stats) {                                                                \
    extern __shared__ __attribute__((aligned(16))) double lds[];                                                  \
    __shared__ int s_p;                                                                                           \
    double* slot = scratch + (int64_t)blockIdx.x * cfg.slot_doubles;                                              \
    for (;;) {                                                                                                    \
      if (threadIdx.x == 0) s_p = atomicAdd(counter, 1);                                                          \
      __syncthreads();                                                                                            \
      const int p = s_p;                                                                                          \
      __syncthreads();                                                                                            \
      if (p >= nprob) break;                                                                                      \
      if (probs[p].lf == nullptr) {                                                                               \
        if (cfg.lds_A1c >= 0) {                                                                                   \
          if (cfg.lds_JA >= 0) eng::run_problem<true, true, false>(probs[p], cfg, slot, lds, stats);              \
          else eng::run_problem<true, false, false>(probs[p], cfg, slot, lds, stats);                             \
        } else {                                                                                                  \
          if (cfg.lds_JA >= 0) eng::run_problem<false, true, false>(probs[p], cfg, slot, lds, stats);             \
          else eng::run_problem<false, false, false>(probs[p], cfg, slot, lds, stats);                            \
        }                                                                                                         \
      } else {   /* triangular factors from the batched gauge sweep: sweep 2 only */                              \
        if (cfg.lds_A1c >= 0) {                                                                                   \
          if (cfg.lds_JA >= 0) eng::run_problem<true, true, true>(probs[p], cfg, slot, lds, stats);               \
          else eng::run_problem<true, false, true>(probs[p], cfg, slot, lds, stats);                              \
        } else {                                                                                                  \
          if (cfg.lds_JA >= 0) eng::run_problem<false, true, true>(probs[p], cfg, slot, lds, stats);              \
          else eng::run_problem<false, false, true>(probs[p], cfg, slot, lds, stats);                             \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }                                                                                                               \
  /* self-test / microbenchmark: Hestenes Jacobi on a pseudo-random m x n matrix held in LDS */                   \
  __global__ void LB jac_bench_kernel(int m, int n, int* sweeps) {                                                \
    extern __shared__ __attribute__((aligned(16))) double lds[];                                                  \
    const int lda = m | 1;                                                                                        \
    double* A = lds + 64 + n;                                                                                     \
    unsigned long long st = 88172645463325252ULL + 977ULL * blockIdx.x;                                           \
    for (int idx = threadIdx.x; idx < m * n; idx += WG_THREADS) {                                                 \
      unsigned long long z = st + 0x9E3779B97F4A7C15ULL * (unsigned long long)(idx + 1);                          \
      z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;           \
      const int r = idx % m, c = idx / m;                                                                         \
      A[r + lda * c] = (r >= c) ? ((double)(z % 2000001ULL) / 1e6 - 1.0) / (1.0 + c) : 0.0;                       \
    }                                                                                                             \
    __syncthreads();                                                                                              \
    int sw = wg::jacobi_rsv((ldbl*)A, lda, m, n, nullptr, 0, (ldbl*)lds,                                              \
                            (__attribute__((address_space(3))) int*)(lds + 64), 60);                 \
    if (threadIdx.x == 0) sweeps[blockIdx.x] = sw;                                                                \
  }

#define WG_THREADS 512
#define WG_WAVES 8
namespace v512 {
#include "wg_blocks.h"
#include "engine.h"
MPBP_ENGINE_KERNEL(__launch_bounds__(512))
}  // namespace v512
#undef WG_THREADS
#undef WG_WAVES

#define WG_THREADS 64
#define WG_WAVES 1
namespace v64 {
#include "wg_blocks.h"
#include "engine.h"
MPBP_ENGINE_KERNEL(__launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))))
}  // namespace v64
#undef WG_THREADS
#undef WG_WAVES

// everything below (self-tests) uses the 512-thread variant
#define WG_THREADS 512
#define WG_WAVES 8
namespace wg = v512::wg;

// ------------------------------------------------------------------------------------------------
// prep: B_k[t][m,n,y,xi] = sum_xk Pxy[t][y,xk,xi] mu_{k->i}[t][m,n,xk,xi]
//   (reference src/recursive_bp_factor.jl:108-115; psi_ik is folded into the table by the host)
// ------------------------------------------------------------------------------------------------
struct PrepProb {
  const double* msg; const int32_t* mbond; int64_t mstride;
  const double* tab; int64_t tab_tstride;    // tab[t*tstride + y + ny1*(xk + q*xi)]
  double* out; int32_t* obond; int64_t ostride; double* ologz;
  int32_t ny1, q;
};

__global__ void prep_kernel(const PrepProb* probs, int L) {
  const PrepProb P = probs[blockIdx.y];
  const int t = blockIdx.x;
  const int bl = P.mbond[t], br = P.mbond[t + 1];
  if (threadIdx.x == 0) {
    P.obond[t] = bl;
    if (t == L - 1) { P.obond[L] = br; *P.ologz = 0.0; }
  }
  const double* A = P.msg + (int64_t)t * P.mstride;
  const double* tab = P.tab + (int64_t)t * P.tab_tstride;
  double* O = P.out + (int64_t)t * P.ostride;
  const int bb = bl * br, q = P.q, ny1 = P.ny1;
  const int tot = bb * ny1 * q;
  for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    int mn = idx % bb; int rest = idx / bb; int y = rest % ny1; int xi = rest / ny1;
    double s = 0.0;
    for (int xk = 0; xk < q; xk++) s += tab[y + ny1 * (xk + q * xi)] * A[mn + bb * (xk + q * xi)];
    O[idx] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// kron: the un-truncated product of incoming messages that the generic (exhaustive-trace) update sums over
//   A[t][(a_1..a_n),(a_1'..a_n'), y = (x_1..x_n), xi] = prod_k mu_{k->i}[t][a_k, a_k', x_k, xi]
//   (reference src/bp_core.jl:36-38 / :74-76: `kron` over the neighbours k != j inside the loop over x_{\partial i \ j};
//   the psi factors and the transition probability are folded into W by the host, so that ctilde_kernel - which computes
//   B[t][m,n,xi,xj,x'] = sum_y W[t][x',xi,xj,y] A[t][m,n,y,xi] - finishes f_bp exactly as it finishes _f_bp_partial).
//   First listed neighbour fastest in all three groups (a gauge choice for the bonds, a convention for y shared with W).
// ------------------------------------------------------------------------------------------------
constexpr int KRON_MAXK = 8;
struct KronProb {
  const double* msg[KRON_MAXK]; const int32_t* mbond[KRON_MAXK]; int64_t mstride;
  double* out; int32_t* obond; int64_t ostride; double* ologz;
  int32_t nk, q;
};

__global__ void kron_kernel(const KronProb* probs, int L) {
  const KronProb P = probs[blockIdx.y];
  const int t = blockIdx.x;
  const int q = P.q, nk = P.nk;
  int bl[KRON_MAXK], br[KRON_MAXK];               // fixed trip counts + predicates: the arrays stay in registers
  int64_t Bl = 1, Br = 1, ny = 1;
#pragma unroll
  for (int k = 0; k < KRON_MAXK; k++) {
    bl[k] = 1; br[k] = 1;
    if (k < nk) { bl[k] = P.mbond[k][t]; br[k] = P.mbond[k][t + 1]; Bl *= bl[k]; Br *= br[k]; ny *= q; }
  }
  if (threadIdx.x == 0) {
    P.obond[t] = (int32_t)Bl;
    if (t == L - 1) { P.obond[L] = (int32_t)Br; *P.ologz = 0.0; }       // messages are stored normalised (z = 1)
  }
  double* O = P.out + (int64_t)t * P.ostride;
  const int64_t tot = Bl * Br * ny * q;
  for (int64_t idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    int64_t r = idx;
    int64_t m = r % Bl; r /= Bl;
    int64_t n = r % Br; r /= Br;
    int64_t y = r % ny; const int xi = (int)(r / ny);
    double v = 1.0;
#pragma unroll
    for (int k = 0; k < KRON_MAXK; k++) {
      if (k < nk) {
        const int a = (int)(m % bl[k]); m /= bl[k];
        const int b = (int)(n % br[k]); n /= br[k];
        const int xk = (int)(y % q); y /= q;
        v *= P.msg[k][(int64_t)t * P.mstride + a + (int64_t)bl[k] * (b + (int64_t)br[k] * (xk + q * xi))];
      }
    }
    O[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// ctilde: apply the transition table and embed the MPEM3 as an explicit MPEM2 of doubled bond
//   B[t][m,n,xi,xj,x'] = sum_y W[t][x',xi,xj,y] A[t][m,n,y,xi]        (src/recursive_bp_factor.jl:79-84)
//   Ct[t][(m,a),(n,x'),(xi,xj)] = delta(a,xi) B[t][m,n,xi,xj,x']       (exact restatement of the index
//   move that mpem2's SVD sweep performs, src/mpems.jl:67-94; first core has no `a`, last no `x'`)
// ------------------------------------------------------------------------------------------------
struct CtProb {
  const double* in; const int32_t* ibond; int64_t istride; const double* ilogz; int32_t ny;
  const double* W;      // W[t*(q*q*qj*ny) + x' + q*(xi + q*(xj + qj*y))]
  int32_t q, qj;
  double* out; int32_t* obond; int64_t ostride; double* ologz;
  int32_t periodic;     // chains periodic in time: one more bond factor q carries c = x_i^1 to the last site, whose factor
                        // depends on x' = x^{T+2} = x^1:  Ct[t][(m,a,c),(n,x',c'),(xi,xj)] = delta(a,xi) delta(c,c') B[t][...],
                        // first core: delta(c',xi) B, last core: delta(a,xi) B[..., x' = c]
};

__global__ void ctilde_kernel(const CtProb* probs, int L) {
  const CtProb P = probs[blockIdx.y];
  const int t = blockIdx.x;
  const int q = P.q, qj = P.qj, ny = P.ny;
  const int bl = P.ibond[t], br = P.ibond[t + 1];
  const int mem = P.periodic ? q : 1;             // size of the carried x_i^1 index
  const int cl = (t == 0) ? 1 : bl * q * mem, cr = (t == L - 1) ? 1 : br * q * mem;
  if (threadIdx.x == 0) {
    P.obond[t] = cl;
    if (t == L - 1) { P.obond[L] = 1; *P.ologz = P.ilogz ? *P.ilogz : 0.0; }
  }
  const double* A = P.in + (int64_t)t * P.istride;
  const double* W = P.W + (int64_t)t * q * q * qj * ny;
  double* O = P.out + (int64_t)t * P.ostride;
  const bool first = t == 0, last = t == L - 1;
  const int tot = cl * cr * q * qj;
  for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    int row = idx % cl; int rest = idx / cl; int col = rest % cr; rest /= cr; int xi = rest % q; int xj = rest / q;
    const int m = row % bl; const int ra = row / bl; const int aa = ra % q, cc = ra / q;     // (m, a, c): m fastest
    const int n = col % br; const int rx = col / br; int xp = rx % q; const int cp = rx / q;   // (n, x', c')
    bool keep = first || aa == xi;
    if (P.periodic) {
      if (first) keep = cp == xi;                 // the memory index is created from x_i^1
      else if (!last) keep = keep && cc == cp;    // and carried unchanged
      if (last) xp = cc;                          // closed on the last site: x' = x_i^1
    }
    double v = 0.0;
    if (keep) {
      for (int y = 0; y < ny; y++)
        v += W[xp + q * (xi + q * (xj + qj * y))] * A[m + bl * (n + br * (y + ny * xi))];
    }
    O[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// compose: train of  new(x) + c * old(x)  by block-diagonal direct sum (TensorTrains `_compose`, used by the
// damping branch of set_msg!, reference src/recursive_bp_factor.jl:172-173); both inputs have z = 1.
// ------------------------------------------------------------------------------------------------
struct ComposeProb {
  const double* an; const int32_t* bn; int64_t nstride;     // new message
  const double* ao; const int32_t* bo; int64_t ostride;     // old message (message slab)
  double* out; int32_t* obond; int64_t outstride; double* ologz;
  double c; int32_t p;
};

__global__ void compose_kernel(const ComposeProb* probs, int L) {
  const ComposeProb P = probs[blockIdx.y];
  const int t = blockIdx.x;
  const int nl = P.bn[t], nr = P.bn[t + 1], ol = P.bo[t], orr = P.bo[t + 1];
  const int cl = (t == 0) ? 1 : nl + ol, cr = (t == L - 1) ? 1 : nr + orr;
  if (threadIdx.x == 0) {
    P.obond[t] = cl;
    if (t == L - 1) { P.obond[L] = 1; *P.ologz = 0.0; }
  }
  const double* A = P.an + (int64_t)t * P.nstride;
  const double* B = P.ao + (int64_t)t * P.ostride;
  double* O = P.out + (int64_t)t * P.outstride;
  const int tot = cl * cr * P.p;
  for (int idx = threadIdx.x; idx < tot; idx += blockDim.x) {
    int m = idx % cl; int rest = idx / cl; int n = rest % cr; int s = rest / cr;
    double v = 0.0;
    if (t == 0) {
      v = (n < nr) ? A[(int64_t)nl * (n + (int64_t)nr * s)] : P.c * B[(int64_t)ol * ((n - nr) + (int64_t)orr * s)];
    } else if (t == L - 1) {
      v = (m < nl) ? A[m + (int64_t)nl * nr * s] : B[(m - nl) + (int64_t)ol * orr * s];
    } else {
      if (m < nl && n < nr) v = A[m + (int64_t)nl * (n + (int64_t)nr * s)];
      else if (m >= nl && n >= nr) v = B[(m - nl) + (int64_t)ol * ((n - nr) + (int64_t)orr * s)];
    }
    O[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// env: environment scans of an explicit train: log sum_x prod_t A_t(x_t) (normalize!, normalization),
// per-site marginals, and the rescaled copy that becomes the stored message.
// ------------------------------------------------------------------------------------------------
struct EnvProb {
  const double* in; const int32_t* ibond; int64_t istride; const double* ilogz; int32_t p;
  double* dst; int32_t* dbond; int64_t dstride;   // normalised copy (sum = 1, z = 1) or null
  double* marg;                                    // [p x L] marginals or null
  double* logz_out;                                // log normalisation (incl. z) or null
  double* rvec;                                    // scratch [L+1][bmax]
  int32_t bmax;
};

__global__ void env_kernel(const EnvProb* probs, int L) {
  const EnvProb P = probs[blockIdx.x];
  __shared__ double sh[256 + 1024 + 64];
  __shared__ double s_scale;
  double* lv = sh;          // [<=256] left vector
  double* tmp = sh + 256;   // [<=1024]
  double* red = sh + 1280;  // [64]
  const int tid = threadIdx.x;
  const int p = P.p;
  // backward pass
  if (tid == 0) P.rvec[(int64_t)L * P.bmax] = 1.0;
  __syncthreads();
  double logP = 0.0;
  for (int t = L - 1; t >= 0; t--) {
    const int bl = P.ibond[t], br = P.ibond[t + 1];
    const double* A = P.in + (int64_t)t * P.istride;
    const double* rn = P.rvec + (int64_t)(t + 1) * P.bmax;
    double* rt = P.rvec + (int64_t)t * P.bmax;
    double mx = 0.0;
    for (int m = tid; m < bl; m += blockDim.x) {
      double s = 0.0;
      for (int n = 0; n < br; n++) {
        double a = 0.0;
        for (int x = 0; x < p; x++) a += A[m + bl * (n + br * x)];
        s += a * rn[n];
      }
      tmp[m] = s;
      mx = fmax(mx, fabs(s));
    }
    // block max
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      double m2 = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); w++) m2 = fmax(m2, red[w]);
      s_scale = m2;
    }
    __syncthreads();
    const double sc = s_scale;
    const double inv = (sc > 0.0 && isfinite(sc)) ? 1.0 / sc : 1.0;
    if (sc > 0.0 && isfinite(sc)) logP += log(sc);
    for (int m = tid; m < bl; m += blockDim.x) rt[m] = tmp[m] * inv;
    __syncthreads();
  }
  {
    double r0 = P.rvec[0];
    logP += log(fabs(r0));     // r0 is +-1 after rescaling (or 0 -> -inf)
  }
  const double logz_in = P.ilogz ? *P.ilogz : 0.0;
  if (tid == 0 && P.logz_out) *P.logz_out = logP - logz_in;
  // normalised copy
  if (P.dst) {
    const double f = exp(-logP / L);
    for (int t = 0; t < L; t++) {
      const int n = P.ibond[t] * P.ibond[t + 1] * p;
      const double* A = P.in + (int64_t)t * P.istride;
      double* D = P.dst + (int64_t)t * P.dstride;
      for (int idx = tid; idx < n; idx += blockDim.x) D[idx] = A[idx] * f;
    }
    for (int t = tid; t <= L; t += blockDim.x) P.dbond[t] = P.ibond[t];
  }
  // forward pass: marginals
  if (P.marg) {
    if (tid == 0) lv[0] = 1.0;
    __syncthreads();
    for (int t = 0; t < L; t++) {
      const int bl = P.ibond[t], br = P.ibond[t + 1];
      const double* A = P.in + (int64_t)t * P.istride;
      const double* rn = P.rvec + (int64_t)(t + 1) * P.bmax;
      // val_x = sum_{m,n} l[m] A[m,n,x] r[n]   (threads over (n,x))
      for (int idx = tid; idx < br * p; idx += blockDim.x) {
        int n = idx % br, x = idx / br;
        double s = 0.0;
        for (int m = 0; m < bl; m++) s += lv[m] * A[m + bl * (n + br * x)];
        tmp[idx] = s;      // [n + br*x] = (l A_x)[n]
      }
      __syncthreads();
      if (tid < p) {
        double v = 0.0;
        for (int n = 0; n < br; n++) v += tmp[n + br * tid] * rn[n];
        red[tid] = v;
      }
      __syncthreads();
      if (tid < p) {
        double tot = 0.0;
        for (int x = 0; x < p; x++) tot += red[x];
        P.marg[tid + p * t] = red[tid] / tot;
      }
      // l <- l * sum_x A_x, rescaled
      double mx = 0.0;
      double mine = 0.0;
      if (tid < br) {
        for (int x = 0; x < p; x++) mine += tmp[tid + br * x];
        mx = fabs(mine);
      }
      for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
      __syncthreads();
      if ((tid & 63) == 0) red[16 + (tid >> 6)] = mx;
      __syncthreads();
      double m2 = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); w++) m2 = fmax(m2, red[16 + w]);
      const double inv = (m2 > 0.0 && isfinite(m2)) ? 1.0 / m2 : 1.0;
      if (tid < br) lv[tid] = mine * inv;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// pair beliefs: environments of the Kronecker train mu_ij (x) mu_ji * psi without forming it
//   (reference src/bp_core.jl:95-109, src/mpbp.jl:218-235).  One workgroup per directed edge.
//   l[a,b] (x) -> l'[a',b'] = sum_{xi,xj} psi[xi,xj] (Aij_{xi,xj}^T l Aji_{xj,xi})
// ------------------------------------------------------------------------------------------------
struct PairProb {
  const double* aij; const int32_t* bij; const double* aji; const int32_t* bji; int64_t stride;
  const double* psi;       // psi[t*q*q + xi + q*xj]
  double* out;             // [q x q x L]
  double* logz;            // log z_ij
  double* scratch;         // [(L+1) * cap*cap] right environments + 2*cap*cap work
  int32_t q, cap;
};

__device__ inline void pair_apply_left(const double* l, const double* Aij, const double* Aji, int a, int a2, int b,
                                       int b2, int xi, int xj, int q, double* tmpm, double* outm, double w,
                                       bool accumulate) {
  // tmpm[a', bq] = sum_a Aij[a,a',xi,xj] l[a,bq] ; outm[a',b'] (+)= w * sum_bq tmpm[a',bq] Aji[bq,b',xj,xi]
  const double* X = Aij + (int64_t)a * a2 * (xi + q * xj);
  const double* Yj = Aji + (int64_t)b * b2 * (xj + q * xi);
  for (int idx = threadIdx.x; idx < a2 * b; idx += blockDim.x) {
    int ap = idx % a2, bq = idx / a2;
    double s = 0.0;
    for (int aa = 0; aa < a; aa++) s += X[aa + a * ap] * l[aa + a * bq];
    tmpm[idx] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < a2 * b2; idx += blockDim.x) {
    int ap = idx % a2, bp = idx / a2;
    double s = 0.0;
    for (int bq = 0; bq < b; bq++) s += tmpm[ap + a2 * bq] * Yj[bq + b * bp];
    outm[idx] = (accumulate ? outm[idx] : 0.0) + w * s;
  }
  __syncthreads();
}

__device__ inline void pair_apply_right(const double* r, const double* Aij, const double* Aji, int a, int a2, int b,
                                        int b2, int xi, int xj, int q, double* tmpm, double* outm, double w,
                                        bool accumulate) {
  // tmpm[a, b'] = sum_a' Aij[a,a',xi,xj] r[a',b'] ; outm[a,b] (+)= w * sum_b' tmpm[a,b'] Aji[b,b',xj,xi]
  const double* X = Aij + (int64_t)a * a2 * (xi + q * xj);
  const double* Yj = Aji + (int64_t)b * b2 * (xj + q * xi);
  for (int idx = threadIdx.x; idx < a * b2; idx += blockDim.x) {
    int aa = idx % a, bp = idx / a;
    double s = 0.0;
    for (int ap = 0; ap < a2; ap++) s += X[aa + a * ap] * r[ap + a2 * bp];
    tmpm[idx] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < a * b; idx += blockDim.x) {
    int aa = idx % a, bq = idx / a;
    double s = 0.0;
    for (int bp = 0; bp < b2; bp++) s += tmpm[aa + a * bp] * Yj[bq + b * bp];
    outm[idx] = (accumulate ? outm[idx] : 0.0) + w * s;
  }
  __syncthreads();
}

__device__ inline double block_maxabs_scale(double* v, int n, double* red) {
  double mx = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) mx = fmax(mx, fabs(v[i]));
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  double m2 = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) m2 = fmax(m2, red[w]);
  if (m2 > 0.0 && isfinite(m2)) {
    const double inv = 1.0 / m2;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v[i] *= inv;
  }
  __syncthreads();
  return m2;
}

__global__ void pair_kernel(const PairProb* probs, int L) {
  const PairProb P = probs[blockIdx.x];
  __shared__ double red[16];
  const int q = P.q, cc = P.cap * P.cap;
  double* R = P.scratch;                         // [(L+1)][cc]
  double* work = P.scratch + (int64_t)(L + 1) * cc;   // tmp [cc], lcur [cc], lnew [cc]
  double* tmpm = work, *lcur = work + cc, *lnew = work + 2 * cc;
  // right environments
  if (threadIdx.x == 0) R[(int64_t)L * cc] = 1.0;
  __syncthreads();
  for (int t = L - 1; t >= 0; t--) {
    const int a = P.bij[t], a2 = P.bij[t + 1], b = P.bji[t], b2 = P.bji[t + 1];
    const double* Aij = P.aij + (int64_t)t * P.stride;
    const double* Aji = P.aji + (int64_t)t * P.stride;
    double* rt = R + (int64_t)t * cc;
    const double* rn = R + (int64_t)(t + 1) * cc;
    bool first = true;
    for (int xi = 0; xi < q; xi++)
      for (int xj = 0; xj < q; xj++) {
        pair_apply_right(rn, Aij, Aji, a, a2, b, b2, xi, xj, q, tmpm, rt, P.psi[(int64_t)t * q * q + xi + q * xj], !first);
        first = false;
      }
    block_maxabs_scale(rt, a * b, red);
  }
  // forward: marginals and log z
  if (threadIdx.x == 0) lcur[0] = 1.0;
  __syncthreads();
  double logz = 0.0;
  for (int t = 0; t < L; t++) {
    const int a = P.bij[t], a2 = P.bij[t + 1], b = P.bji[t], b2 = P.bji[t + 1];
    const double* Aij = P.aij + (int64_t)t * P.stride;
    const double* Aji = P.aji + (int64_t)t * P.stride;
    const double* rn = R + (int64_t)(t + 1) * cc;
    double vals[16];
    double tot = 0.0;
    for (int xi = 0; xi < q; xi++)
      for (int xj = 0; xj < q; xj++) {
        // lx = l applied with (xi,xj) only, then <lx, r>
        pair_apply_left(lcur, Aij, Aji, a, a2, b, b2, xi, xj, q, tmpm, lnew, P.psi[(int64_t)t * q * q + xi + q * xj], false);
        double s = 0.0;
        for (int i = threadIdx.x; i < a2 * b2; i += blockDim.x) s += lnew[i] * rn[i];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        double v = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) v += red[w];
        __syncthreads();
        if (xi + q * xj < 16) vals[xi + q * xj] = v;
        tot += v;
      }
    if (threadIdx.x == 0)
      for (int s = 0; s < q * q && s < 16; s++) P.out[s + q * q * t] = vals[s] / tot;
    // advance l with the summed core
    bool first = true;
    for (int xi = 0; xi < q; xi++)
      for (int xj = 0; xj < q; xj++) {
        pair_apply_left(lcur, Aij, Aji, a, a2, b, b2, xi, xj, q, tmpm, lnew, P.psi[(int64_t)t * q * q + xi + q * xj], !first);
        first = false;
      }
    double m2 = block_maxabs_scale(lnew, a2 * b2, red);
    if (m2 > 0.0 && isfinite(m2)) logz += log(m2);
    double* sw = lcur; lcur = lnew; lnew = sw;
  }
  if (threadIdx.x == 0) *P.logz = logz + log(fabs(lcur[0]));
}

// ------------------------------------------------------------------------------------------------
// Two-time marginals of the belief trains: p_i(x^t, x^u), t < u <= t + maxdist (TensorTrains `twovar_marginals`,
// used by autocorrelations / autocovariances / alternate_marginals, reference src/mpbp.jl:245-286).
//   tv_env_kernel: left and right environments of the x-summed cores (rescaled by max-abs), one workgroup per node
//   tv_kernel:     one workgroup per (start time t, node): mid = l_{t-1} b[t][:,:,x], then for u = t+1 ..:
//                  p[x,y] = mid[x,:] b[u][:,:,y] r_{u+1},  mid <- mid (sum_y b[u][:,:,y]) rescaled
// ------------------------------------------------------------------------------------------------
struct TvProb {
  const double* cores; const int32_t* bond; int64_t stride;   // belief train: core t at cores + t*stride, [b_t, b_{t+1}, q]
  double* lenv; double* renv;                                 // [(L+1) * bmax] each: lenv[t] before core t, renv[t] after core t-1
  double* out;                                                // [L][L][q*q] (device), zero where undefined
  int32_t q, bmax;
};

__global__ void __launch_bounds__(256) tv_env_kernel(const TvProb* probs, int L) {
  const TvProb P = probs[blockIdx.x];
  __shared__ double red[8];
  const int tid = threadIdx.x, q = P.q;
  auto wgmax = [&](double v) {
    v = wg::wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  };
  // left: lenv[0] = [1]; lenv[t+1][n] = sum_m lenv[t][m] S_t[m,n],  S_t = sum_x b[t][:,:,x]
  if (tid == 0) { P.lenv[0] = 1.0; P.renv[(int64_t)L * P.bmax] = 1.0; }
  __syncthreads();
  for (int t = 0; t < L; t++) {
    const int a = P.bond[t], an = P.bond[t + 1];
    const double* c = P.cores + (int64_t)t * P.stride;
    const double* lv = P.lenv + (int64_t)t * P.bmax;
    double* lo = P.lenv + (int64_t)(t + 1) * P.bmax;
    double mx = 0.0, mine = 0.0;
    if (tid < an) {
      double s = 0.0;
      for (int x = 0; x < q; x++)
        for (int m = 0; m < a; m++) s += lv[m] * c[m + (int64_t)a * (tid + (int64_t)an * x)];
      mine = s; mx = fabs(s);
    }
    mx = wgmax(mx);
    if (tid < an) lo[tid] = (mx > 0.0 && isfinite(mx)) ? mine / mx : mine;
    __syncthreads();
  }
  for (int t = L - 1; t >= 0; t--) {
    const int a = P.bond[t], an = P.bond[t + 1];
    const double* c = P.cores + (int64_t)t * P.stride;
    const double* rv = P.renv + (int64_t)(t + 1) * P.bmax;
    double* ro = P.renv + (int64_t)t * P.bmax;
    double mx = 0.0, mine = 0.0;
    if (tid < a) {
      double s = 0.0;
      for (int x = 0; x < q; x++)
        for (int n = 0; n < an; n++) s += c[tid + (int64_t)a * (n + (int64_t)an * x)] * rv[n];
      mine = s; mx = fabs(s);
    }
    mx = wgmax(mx);
    if (tid < a) ro[tid] = (mx > 0.0 && isfinite(mx)) ? mine / mx : mine;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) tv_kernel(const TvProb* probs, int L, int maxdist) {
  const TvProb P = probs[blockIdx.y];
  const int t = blockIdx.x, tid = threadIdx.x, q = P.q;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* mid = lds;                    // [q][bmax]
  double* nmid = lds + q * P.bmax;      // [q][bmax]
  double* pacc = nmid + q * P.bmax;     // [4 waves][16]
  double* red = pacc + 64;              // [8]
  double* out = P.out + (int64_t)t * L * q * q;
  {
    const int a = P.bond[t], an = P.bond[t + 1];
    const double* c = P.cores + (int64_t)t * P.stride;
    const double* lv = P.lenv + (int64_t)t * P.bmax;
    for (int idx = tid; idx < q * an; idx += 256) {
      const int n = idx % an, x = idx / an;
      double s = 0.0;
      for (int m = 0; m < a; m++) s += lv[m] * c[m + (int64_t)a * (n + (int64_t)an * x)];
      mid[x * P.bmax + n] = s;
    }
  }
  __syncthreads();
  const int uend = min(L, t + maxdist + 1);
  for (int u = t + 1; u < uend; u++) {
    const int a = P.bond[u], an = P.bond[u + 1];
    const double* c = P.cores + (int64_t)u * P.stride;
    const double* rv = P.renv + (int64_t)(u + 1) * P.bmax;
    double pl[16];
#pragma unroll
    for (int k = 0; k < 16; k++) pl[k] = 0.0;
    double mx = 0.0;
    for (int n = tid; n < an; n += 256) {
      const double r = rv[n];
      for (int x = 0; x < q; x++) {
        double tot = 0.0;
        for (int y = 0; y < q; y++) {
          const double* col = c + (int64_t)a * (n + (int64_t)an * y);
          double s = 0.0;
          for (int m = 0; m < a; m++) s += mid[x * P.bmax + m] * col[m];
          tot += s;
#pragma unroll
          for (int k = 0; k < 16; k++) pl[k] += (k == x + q * y) ? s * r : 0.0;
        }
        nmid[x * P.bmax + n] = tot;
        mx = fmax(mx, fabs(tot));
      }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) pl[k] = wg::wave_sum(pl[k]);
    mx = wg::wave_max(mx);
    if ((tid & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) pacc[(tid >> 6) * 16 + k] = pl[k];
      red[tid >> 6] = mx;
    }
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (tid < q * q) {
      double tot = 0.0;
      for (int k = 0; k < q * q; k++) tot += pacc[k] + pacc[16 + k] + pacc[32 + k] + pacc[48 + k];
      const double v = pacc[tid] + pacc[16 + tid] + pacc[32 + tid] + pacc[48 + tid];
      out[(int64_t)u * q * q + tid] = v / tot;
    }
    const double inv = (mx > 0.0 && isfinite(mx)) ? 1.0 / mx : 1.0;
    for (int idx = tid; idx < q * an; idx += 256) { const int n = idx % an, x = idx / an; mid[x * P.bmax + n] = nmid[x * P.bmax + n] * inv; }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// self-test kernels (building blocks against host references; used by tests/)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(WG_THREADS) st_gemm_kernel(int M, int N, int K, const double* A, const double* B, double* C) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  wg::gemm(M, N, K, (const gdbl*)A, [=](int i) { return i; }, [=](int k) { return (int64_t)M * k; },
           (const gdbl*)B, [=](int k) { return k; }, [=](int j) { return (int64_t)K * j; }, true,
           (gdbl*)C, [=](int i) { return i; }, [=](int j) { return (int64_t)M * j; }, false, (ldbl*)lds);
}
__global__ void __launch_bounds__(WG_THREADS) st_qr_kernel(double* Y, int ld, int rows, int cols, wg::Prof* pr) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  unsigned long long last = pr ? wall_clock64() : 0ULL;
  wg::qr_r((gdbl*)Y + (int64_t)blockIdx.x * ld * (((cols + 15) & ~15) + 16), ld, rows, cols, (ldbl*)lds,
           (ldbl*)lds + wg::QR_LDS_DOUBLES,
           pr, &last, -1, -1);
}
__global__ void __launch_bounds__(WG_THREADS) st_svd_kernel(double* A, int m, int n, double* V, double* sigma, int* sweeps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  int sw = wg::jacobi_rsv((gdbl*)A, m, m, n, V, n, (ldbl*)lds, (__attribute__((address_space(3))) int*)(lds + 32), 60);
  for (int c = threadIdx.x; c < n; c += WG_THREADS) {
    double s = 0.0;
    for (int r = 0; r < m; r++) s += A[r + (int64_t)m * c] * A[r + (int64_t)m * c];
    sigma[c] = sqrt(s);
  }
  if (threadIdx.x == 0) *sweeps = sw;
}
