// Problem / configuration records of the compress engine (shared by all workgroup-size variants).
#pragma once
#include "wg_common.h"
#include "../../include/mpbp_hip.h"


struct EngProb {
  const double* A1; const int32_t* bond1; int64_t stride1; int32_t ny1;   // cores [m,n,y1,xi]
  const double* A2; const int32_t* bond2; int64_t stride2; int32_t ny2;   // cores [m,n,y2,xi]
  const double* logz1; const double* logz2;                               // may be null (= 0)
  const double* pyy; int64_t pyy_tstride;   // pyy[tp*tstride + y + ny*(y1 + ny1*(y2 + ny2*xi))]
  int32_t ny, q, mirror, cap_out;
  double* out; int32_t* obond; int64_t ostride; double* ologz;            // output cores [m,n,y,xi]
  // triangular factors of sweep 1 supplied by the batched gauge sweep (v2_engine.hip): Lf_t^T at lf + lfoff[t] as
  // [rdim[t] x a_t b_t], ld rdim[t]; null = the workgroup runs sweep 1 itself into its slot
  const double* lf; const int64_t* lfoff; const int32_t* rdim;
};

struct EngCfg {
  int32_t L;
  int32_t Bmax;        // max product bond  (cap1*cap2)
  int32_t nmax;        // max rows of M_t   (cap_out*ny*q)
  // per-slot global scratch layout (offsets in doubles)
  int64_t off_Lf, lf_stride;   // (L+1) triangular factors, lf_stride doubles each
  int64_t off_Z, off_Y, off_C0, off_C1, off_T1, off_Nt, off_Mt, off_JA, off_JV, off_A1c, off_A2c, off_E;
  int64_t slot_doubles;
  // LDS layout (offsets in doubles from the dynamic LDS base); negative => use the global copy
  int32_t lds_gemm, lds_qr, lds_misc, lds_A1c, lds_A2c, lds_E, lds_JA, lds_JV, lds_rdim;
  mpbp_trunc trunc;
  wgc::Prof* prof;     // optional phase timers (null = off)
  int32_t force_generic;  // debug: take the large-problem code paths (global-memory QR panel)
};

struct EngStats {
  unsigned long long maxerr_bits;
  unsigned long long n_compress;
  int32_t nan_flag, capacity_flag, jacobi_fail;
  unsigned long long jac_sweeps, jac_calls;
};

