// Interface between the two translation units of libmpbp_hip: the batched gauge sweep (v2_engine.hip) and the sweep
// driver / workgroup-per-problem engine (mpbp_hip.hip).
#pragma once
#include "ctx.h"

// Sweep 1 (triangular factors Lf_t of the product trains) of probs[0 .. n) as one lock-step batch of grid-level
// launches on c->stream.  Processes the longest prefix whose buffers fit device memory (*n_done, >= 1 on success) and
// sets probs[i].lf / lfoff / rdim (pointers into c->v2arena, valid until the next call).  Problems must not be
// mirrored.  Asynchronous: the caller may enqueue sweep 2 on the same stream right away.
// `hb`: host copy of the operands' bond tables, [n][2][L+1] (v2_gather_bonds).
// `trunc2` non-null: also run the truncating sweep (sweep 2) in grid-level form - only for rules whose kept rank follows
// from the dimensions (MPBP_TRUNC_BOND / BOND_MAX); *did_sweep2 tells the caller that the outputs are complete.
int v2_gauge_sweep(mpbp_ctx* c, EngProb* probs, int n, const int32_t* hb, const mpbp_trunc* trunc2, int* n_done, int* did_sweep2);

// Bond tables (bond1, bond2; L+1 entries each) of probs[0 .. n) -> host, [n][2][L+1].  Synchronises the stream.
int v2_gather_bonds(mpbp_ctx* c, const EngProb* probs, int n, std::vector<int32_t>& hb);

// Per-device shared state of the batched QR (the look-ahead's two CU-masked streams + six events): every context holds a
// reference from mpbp_create to mpbp_destroy; the last release destroys them (nothing is left for static destructors).
void v2_device_acquire(int dev);
void v2_device_release(int dev);
