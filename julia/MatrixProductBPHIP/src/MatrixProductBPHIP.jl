# MatrixProductBPHIP — Julia side of the drop-in: `iterate!(bp, HIPBackend(bp; max_bond))` runs the sweep loop of
# MatrixProductBP.jl (src/mpbp.jl:185-198) with `onebpiter!` (src/recursive_bp_factor.jl:146-165) executed by
# libmpbp_hip.so on an MI355X, and writes `bp.μ`, `bp.b`, `bp.f` back so that every observable of the reference
# (`beliefs`, `pair_beliefs`, `bethe_free_energy`, `autocorrelations`, `CB_BP`) works unchanged.
#
# STATUS: written against include/mpbp_hip.h; the build image has no Julia, so this file has not been executed.
# The Python mirror (matrixproductbp.jl_amd/mpbp.py) binds the same entry points and IS what the parity tests run.
# Dependency UUIDs in Project.toml are the ones the reference's own Project.toml lists.
module MatrixProductBPHIP

using MatrixProductBP, TensorTrains, IndexedGraphs, SparseArrays
import MatrixProductBP: MPBP, RecursiveBPFactor, nstates, prob_y, prob_xy, prob_yy, prob_y0, getT, CB_BP

export HIPBackend, push_messages!, pull_messages!, pull_beliefs!

const LIB = get(ENV, "MPBP_HIP_LIB", "libmpbp_hip.so")

struct Trunc                       # mpbp_trunc
    kind::Int32; mprime::Int32; eps::Float64
end
abi(t::TruncThresh)     = Trunc(0, 0, t.ε)
abi(t::TruncBond)       = Trunc(1, t.mprime, 0.0)
abi(t::TruncBondMax)    = Trunc(2, t.mprime, 0.0)
abi(t::TruncBondThresh) = Trunc(3, t.mprime, t.ε)

struct Desc                        # mpbp_desc, same field order as the header
    n_nodes::Int32; n_edges::Int32; T::Int32; q::Int32
    nbr_ptr::Ptr{Int32}; in_edge::Ptr{Int32}; out_edge::Ptr{Int32}
    max_bond::Int32; device::Int32
    slot_of_edge::Ptr{Int32}; n_slots::Int32
    ext_cores::Ptr{Cvoid}; ext_bonds::Ptr{Cvoid}; stream::Ptr{Cvoid}
    periodic::Int32                # 1 for `periodic_mpbp` (MPBP{...,<:PeriodicMPEM2,...}, src/mpbp.jl:113-114)
end

mutable struct Stats               # mpbp_stats
    maxerr::Float64; n_compress::Int64; nan_flag::Int32; capacity_flag::Int32; jacobi_not_converged::Int32
    ms_total::Float32; ms_orth::Float32; n_orth_launches::Int32; jacobi_sweeps::Int64; jacobi_calls::Int64
    Stats() = new(0.0, 0, 0, 0, 0, 0f0, 0f0, 0, 0, 0)
end

mutable struct HIPBackend
    h::Ptr{Cvoid}
    N::Int; E::Int; T::Int; q::Int
    stats::Stats
end

lasterr(h) = unsafe_string(ccall((:mpbp_last_error, LIB), Cstring, (Ptr{Cvoid},), h))
check(rc, h=C_NULL) = rc == 0 ? nothing : error("libmpbp_hip: " * lasterr(h))

"Neighbour tables of `g` in the library's convention (0-based; position p of node i ↔ p-th in/out edge)."
function neighbour_tables(g::IndexedBiDiGraph)
    X = g.X
    nbr_ptr  = Int32.(X.colptr .- 1)
    in_edge  = Int32.(collect(0:nnz(X)-1))      # idx of the p-th in-edge of i is p (column-major order of A)
    out_edge = Int32.(nonzeros(X) .- 1)          # g.X.nzval maps in-edge position → idx of the reverse edge
    nbr_ptr, in_edge, out_edge
end
# InfiniteRegularGraph (src/infinite_graph.jl:8-20): one node, k aliases of the single edge
neighbour_tables(g::MatrixProductBP.InfiniteRegularGraph) = (Int32[0, g.k], zeros(Int32, g.k), zeros(Int32, g.k))

"Dense tables of the RecursiveBPFactor interface (src/recursive_bp_factor.jl:11-27) for one node."
function factor_tables(w::RecursiveBPFactor, q::Int, deg::Int)
    ny  = Int32[nstates(w, l) for l in 0:deg]
    py  = Float64[prob_y(w, xn, x, y, deg) for xn in 1:q, x in 1:q, y in 1:ny[deg+1]]
    pxy = Float64[prob_xy(w, y, xk, xi, k) for y in 1:ny[2], xk in 1:q, xi in 1:q, k in 1:deg]
    pyy = Float64[]
    for d1 in 0:deg, d2 in 0:deg-d1
        append!(pyy, vec(Float64[prob_yy(w, y, y1, y2, xi, d1, d2)
                                 for y in 1:ny[d1+d2+1], y1 in 1:ny[d1+1], y2 in 1:ny[d2+1], xi in 1:q]))
    end
    py0 = Float64[prob_y0(w, y, xi) for y in 1:ny[1], xi in 1:q]
    ny, py, pxy, pyy, py0
end

"Dense transition table of a generic `BPFactor` (src/bp_core.jl:1-10) for `mpbp_set_generic_factor`:
 `w[x', x, x_1, ..., x_deg]`, first index fastest."
function generic_table(w::BPFactor, q::Int, deg::Int)
    tab = zeros(Float64, q, q, ntuple(_ -> q, deg)...)
    for xs in Iterators.product(ntuple(_ -> 1:q, deg)...), x in 1:q, xn in 1:q
        tab[xn, x, xs...] = w(xn, collect(xs), x)
    end
    vec(tab)
end

function HIPBackend(bp::MPBP; max_bond::Integer, device::Integer=0)
    # nstates may differ from node to node (src/mpbp.jl:22-26): the device works with q = the largest and treats the states
    # beyond nstates(bp, i) as padding of zero weight (mpbp_set_node_states); factor tables are evaluated for all q states
    # (a factor that throws on a state it does not know has to be wrapped), ϕ / ψ are zero-padded
    g = bp.g; N = nv(g); E = ne(g); T = getT(bp)
    qn = Int32[nstates(b) for b in bp.b]; q = Int(maximum(qn))
    nbr_ptr, in_edge, out_edge = neighbour_tables(g)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve nbr_ptr in_edge out_edge begin
        d = Desc(N, E, T, q, pointer(nbr_ptr), pointer(in_edge), pointer(out_edge), max_bond, device,
                 C_NULL, 0, C_NULL, C_NULL, C_NULL, Int32(MatrixProductBP.is_periodic(bp)))
        check(ccall((:mpbp_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{Desc}), h, d))
    end
    be = HIPBackend(h[], N, E, T, q, Stats())
    finalizer(b -> ccall((:mpbp_destroy, LIB), Cvoid, (Ptr{Cvoid},), b.h), be)
    any(qn .!= q) && check(ccall((:mpbp_set_node_states, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), be.h, qn), be.h)
    for i in 1:N
        deg = Int(nbr_ptr[i+1] - nbr_ptr[i])
        same = all(w == bp.w[i][1] for w in bp.w[i])
        ts = same ? (1:1) : (1:T+1)
        if !(eltype(bp.w[i]) <: RecursiveBPFactor)      # generic factor: exhaustive-trace update (src/mpbp.jl:117-154)
            tab = reduce(vcat, [generic_table(bp.w[i][t], q, deg) for t in ts])
            check(ccall((:mpbp_set_generic_factor, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Int32, Ptr{Float64}),
                        be.h, i - 1, deg, length(ts), tab), be.h)
            continue
        end
        tabs = [factor_tables(bp.w[i][t], q, deg) for t in ts]
        ny = tabs[1][1]
        cat(k) = reduce(vcat, [vec(tb[k]) for tb in tabs])
        check(ccall((:mpbp_set_factor, LIB), Cint,
                    (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    be.h, i - 1, deg, ny, length(ts), cat(2), cat(3), cat(4), cat(5)), be.h)
    end
    ends = [(src(e), dst(e)) for e in edges(g)]                  # edge ids in the order of idx(e)
    ϕ = Float64[x <= qn[i] ? bp.ϕ[i][t][x] : 0.0 for x in 1:q, t in 1:T+1, i in 1:N]
    ψ = Float64[(xi <= qn[ends[e][1]] && xj <= qn[ends[e][2]]) ? bp.ψ[e][t][xi, xj] : 0.0 for xi in 1:q, xj in 1:q, t in 1:T+1, e in 1:E]
    check(ccall((:mpbp_set_phi, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), be.h, ϕ), be.h)
    check(ccall((:mpbp_set_psi, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), be.h, ψ), be.h)
    push_messages!(be, bp)
    be
end

"Upload `bp.μ` (must be normalised, as it is after `set_msg!`); z of each message is folded into its first core."
function push_messages!(be::HIPBackend, bp::MPBP)
    T = be.T
    bonds = Int32[]; offsets = Int64[]; data = Float64[]
    for e in 1:be.E
        A = bp.μ[e]
        push!(offsets, length(data))
        for t in 1:T+1
            push!(bonds, size(A[t], 1))
            c = t == 1 ? A[t] ./ float(A.z) : A[t]
            append!(data, vec(c))
        end
        push!(bonds, 1)
    end
    check(ccall((:mpbp_set_messages, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int64}, Ptr{Float64}),
                be.h, bonds, offsets, data), be.h)
end

"Download the messages into `bp.μ` (z = 1)."
function pull_messages!(bp::MPBP, be::HIPBackend)
    T = be.T; q = be.q
    bonds = zeros(Int32, be.E * (T + 2))
    check(ccall((:mpbp_get_bonds, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), be.h, bonds), be.h)
    B = reshape(bonds, T + 2, be.E)
    sizes = [sum(Int(B[t, e]) * Int(B[t+1, e]) * q * q for t in 1:T+1) for e in 1:be.E]
    offsets = Int64.(cumsum([0; sizes[1:end-1]]))
    data = zeros(Float64, sum(sizes))
    check(ccall((:mpbp_get_messages, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}), be.h, offsets, data), be.h)
    for e in 1:be.E
        o = offsets[e]
        cores = map(1:T+1) do t
            n = Int(B[t, e]) * Int(B[t+1, e]) * q * q
            c = reshape(data[o+1:o+n], Int(B[t, e]), Int(B[t+1, e]), q, q); o += n; c
        end
        bp.μ[e] = MPEM2(cores)
    end
    bp
end

"Download `bp.b` (normalised MPEM1 trains) and `bp.f`."
function pull_beliefs!(bp::MPBP, be::HIPBackend)
    T = be.T; q = be.q
    for i in 1:be.N
        bonds = zeros(Int32, T + 2)
        check(ccall((:mpbp_get_belief_train, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Float64}, Int64),
                    be.h, i - 1, bonds, C_NULL, 0), be.h)
        n = sum(Int(bonds[t]) * Int(bonds[t+1]) * q for t in 1:T+1)
        data = zeros(Float64, n)
        check(ccall((:mpbp_get_belief_train, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}, Ptr{Float64}, Int64),
                    be.h, i - 1, bonds, data, n), be.h)
        o = 0
        bp.b[i] = MPEM1(map(1:T+1) do t
            m = Int(bonds[t]) * Int(bonds[t+1]) * q
            c = reshape(data[o+1:o+m], Int(bonds[t]), Int(bonds[t+1]), q); o += m; c
        end)
    end
    check(ccall((:mpbp_free_energy, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), be.h, bp.f), be.h)
    bp
end

"""
    iterate!(bp, be::HIPBackend; maxiter, svd_trunc, cb, tol, nodes, damp, schedule=:jacobi)

Device version of `iterate!` (src/mpbp.jl:185-198).  `schedule=:jacobi` updates all `nodes` from the messages of
the previous sweep in one call; `:sequential` calls the library once per node in the given order, which is the
reference's single-thread order exactly.  The callback contract is unchanged.
"""
function MatrixProductBP.iterate!(bp::MPBP, be::HIPBackend; maxiter::Integer=5, svd_trunc=TruncBond(be.q),
        showprogress=false, cb=CB_BP(bp; showprogress), tol=1e-10, nodes=collect(vertices(bp.g)), damp=0.0,
        schedule::Symbol=:jacobi, pull_messages::Bool=true)
    tr = abi(svd_trunc)
    sweep(ns) = check(ccall((:mpbp_sweep, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32, Trunc, Float64, Ref{Stats}),
                            be.h, Int32.(ns .- 1), length(ns), tr, damp, be.stats), be.h)
    for it in 1:maxiter
        if schedule == :jacobi
            sweep(nodes)
        else
            foreach(i -> sweep([i]), nodes)
        end
        be.stats.nan_flag != 0 && @error "NaN in tensor train"
        be.stats.jacobi_not_converged != 0 && @warn "a Jacobi SVD hit its sweep limit in this update"
        be.stats.capacity_flag != 0 && @warn "a bond hit max_bond: results differ from the reference"
        svd_trunc isa TruncBondMax && (svd_trunc.maxerr[] = max(svd_trunc.maxerr[], be.stats.maxerr))
        pull_beliefs!(bp, be)
        Δ = cb(bp, it, svd_trunc)
        if Δ < tol
            pull_messages && pull_messages!(bp, be)
            return it, cb
        end
    end
    pull_messages && pull_messages!(bp, be)
    return maxiter, cb
end

"""
    twovar_marginals_device(bp, be; sites, maxdist) -> Array{Float64,5}  # [x_t, x_u, u, t, k]

Two-time marginals of the beliefs of `sites` computed on the device (`mpbp_twovar_marginals`; replaces the
`twovar_marginals(bp.b[i]; maxdist)` scan behind `autocorrelations` / `autocovariances`, src/mpbp.jl:245-255).
Entries with `u <= t` or `u > t + maxdist` are zero.
"""
function twovar_marginals_device(bp::MPBP, be::HIPBackend; sites=collect(vertices(bp.g)), maxdist::Integer=getT(bp) + 1)
    L = getT(bp) + 1; q = be.q
    out = zeros(Float64, q, q, L, L, length(sites))
    check(ccall((:mpbp_twovar_marginals, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}, Int32, Int32, Ptr{Float64}),
                be.h, Int32.(sites .- 1), length(sites), maxdist, out), be.h)
    out
end

"""
    allgather_slots!(be, comm, rank, world, slots_per_rank)

Exchange step of the multi-GPU path: one in-place RCCL all-gather of the message slab and one of the bond table
(`mpbp_allgather_slots`; `comm` is the `ncclComm_t` of the host's RCCL binding).  Makes the messages written by every
rank's `mpbp_sweep` visible on all ranks - the `bp.μ[idx(e)] = μj` of src/recursive_bp_factor.jl:177 across GPUs.
"""
allgather_slots!(be::HIPBackend, comm::Ptr{Cvoid}, rank::Integer, world::Integer, S::Integer) =
    check(ccall((:mpbp_allgather_slots, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Int32), be.h, comm, rank, world, S), be.h)

end # module
