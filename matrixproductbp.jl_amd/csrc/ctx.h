// Host-side context of libmpbp_hip (shared by the translation units of the library).
#pragma once
#include "wg_common.h"
#include "engine_types.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

extern thread_local std::string g_create_error;

// On failure the call is abandoned - but host vectors of the caller may still be the source of an asynchronous copy in
// flight on the context's stream, so the stream is drained before the early return lets them die (round-3 advisor).
#define HIPCHK(ctx, call)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      if ((ctx)->stream) (void)hipStreamSynchronize((ctx)->stream);                               \
      return (ctx)->fail(MPBP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                             \
  } while (0)

struct DevTrain {       // a tensor train resident in HBM
  double* cores = nullptr; int32_t* bonds = nullptr; double* logz = nullptr;
  int64_t stride = 0;   // doubles between cores
  int cap = 0, ny = 0, d = 0, level = 0;
};

struct NodeFactor {
  bool set = false; int deg = 0, nt = 1;
  std::vector<int> ny;
  std::vector<double> prob_y, prob_xy, prob_yy, prob_y0;
  std::vector<int64_t> yy_off;   // offset of block (d1,d2) inside one time block of prob_yy
  int64_t yy_tblock = 0;
  // generic (non-recursive) factor, mpbp_set_generic_factor: the dense transition table
  //   gen_w[t][x' + q (x_i + q (x_1 + q (x_2 + ... )))] = w_i^t(x' | x_1 .. x_deg, x_i)     (neighbours in position order)
  bool generic = false;
  std::vector<double> gen_w;
};

// bump allocator over one device arena, regrown on demand between sweeps
struct Arena {
  char* base = nullptr; size_t cap = 0, used = 0, want = 0;
  void reset() { used = 0; want = 0; }
  void* take(size_t bytes) {
    size_t a = (bytes + 255) & ~size_t(255);
    want += a;
    if (used + a > cap) { return nullptr; }
    void* p = base + used; used += a; return p;
  }
};

struct mpbp_ctx {
  int N = 0, E = 0, T = 0, L = 0, q = 0, cap = 0, device = 0, nslots = 0;
  bool periodic = false;          // time-periodic chains (mpbp_desc::periodic)
  int ct_factor() const { return periodic ? q * q : q; }   // bond factor of the MPEM3 -> MPEM2 embedding
  std::vector<int> nbr_ptr, in_edge, out_edge, slot_of_edge;
  // heterogeneous nstates(bp, i) (reference src/mpbp.jl: q per node): node i uses the first qnode[i] <= q states, the others
  // are padding that carries exactly zero weight (mpbp_set_node_states); edge_src / edge_dst: end nodes of every edge
  std::vector<int> qnode, edge_src, edge_dst;
  bool hetero_q = false;
  std::vector<NodeFactor> fac;
  std::vector<double> phi, psi;           // host copies (ABI layouts)
  bool own_cores = false, own_bonds = false, own_stream = false;
  double* d_cores = nullptr; int32_t* d_bonds = nullptr;
  int64_t core_stride = 0, slot_doubles = 0;
  hipStream_t stream = nullptr;
  // persistent outputs
  double* d_beliefs = nullptr;    // [q][L][N]
  double* d_btrain = nullptr; int32_t* d_bbond = nullptr;   // normalised belief trains (MPEM1, bond <= q*cap), optional
  int64_t bt_stride = 0, bt_slot = 0;
  double* d_logz_node = nullptr;  // [N]
  double* d_logz_pos = nullptr;   // [nnz]  log z_{i->j} per neighbour position
  std::vector<double> h_logz_node, h_logz_pos, h_f;
  EngStats* d_stats = nullptr; int* d_counter = nullptr; wgc::Prof* d_prof = nullptr;
  double* d_one = nullptr; int32_t* d_ones = nullptr; double* d_ident = nullptr; int ident_n = 0;
  // tables
  bool tables_dirty = true;
  double* d_tab = nullptr; size_t tab_doubles = 0;
  std::vector<int64_t> pxy_off;    // per neighbour position
  std::vector<int64_t> pxy_tstride;
  std::vector<int64_t> wmsg_off;   // per neighbour position
  std::vector<int64_t> wbel_off;   // per node
  std::vector<int64_t> init_off;   // per node: [ny0][q]
  std::vector<int64_t> pyy_base;   // per node: offset of prob_yy blob
  Arena arena, scratch, v2arena;   // work trains of a sweep / engine slots + launch records / batched gauge sweep
  int num_cu = 256;
  bool dev_held = false;        // v2_device_acquire done (released in mpbp_destroy)
  bool no_coop_panel = false;   // set after a cooperative panel launch timed out: the column steps then run as separate launches
  int profiling = 0;            // 0 off, 1 HIP-event timing of the cavity launches, 2 also the in-kernel phase timers
  std::string err;
  mpbp_stats last{};

  int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    err = buf; return code;
  }
  int nnz() const { return nbr_ptr[N]; }
  double* slot_cores(int e) const { return d_cores + (int64_t)slot_of_edge[e] * slot_doubles; }
  int32_t* slot_bonds(int e) const { return d_bonds + (int64_t)slot_of_edge[e] * (L + 1); }
  // incoming messages of a pass: the live slab, or the snapshot of a split Jacobi sweep (mpbp_sweep).  The snapshot is
  // compact: it holds the in-edges of the listed nodes only, snap_index[e] = position of edge e in it (or -1).
  double* snap_cores = nullptr; int32_t* snap_bonds = nullptr;
  std::vector<int32_t> snap_index;
  const double* read_slot_cores(int e) const {
    return snap_cores ? snap_cores + (int64_t)snap_index[e] * slot_doubles : d_cores + (int64_t)slot_of_edge[e] * slot_doubles;
  }
  const int32_t* read_slot_bonds(int e) const {
    return snap_bonds ? snap_bonds + (int64_t)snap_index[e] * (L + 1) : d_bonds + (int64_t)slot_of_edge[e] * (L + 1);
  }
};

// grows an arena (contents are NOT preserved); used between launches only
inline int ensure_arena(mpbp_ctx* c, Arena& a, size_t bytes) {
  if (a.cap >= bytes) return MPBP_OK;
  if (a.base) { hipFree(a.base); a.base = nullptr; a.cap = 0; }
  size_t want = bytes + (bytes >> 3) + (1 << 20);
  hipError_t e = hipMalloc((void**)&a.base, want);
  if (e != hipSuccess) {
    want = bytes + 4096;
    e = hipMalloc((void**)&a.base, want);
  }
  if (e != hipSuccess) return c->fail(MPBP_ENOMEM, "hipMalloc(%zu MiB work arena) failed: %s", want >> 20, hipGetErrorString(e));
  a.cap = want;
  return MPBP_OK;
}
