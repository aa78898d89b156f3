# SDPLRPlusHIP.jl — the reference-side binding of libsdplr_hip.so (MI355X / gfx950 device backend).
#
# Drop this file next to src/SDPLRPlus.jl and `include("SDPLRPlusHIP.jl")` after the other includes (it uses
# SDPData, SolverVars, SolverAuxiliary, SolverStats, BurerMonteiroConfig, b_vector, C_matrix, rank_update!,
# printintermediate, copy2y_λ!).  It adds a second aux type, `HIPAux`, with
#   * methods of the plug-in operator set that src/lowrankopt.jl:57-135 overloads for its own model type —
#     side_dimension, 𝒜! (one- and two-argument), 𝒜t_preprocess!, 𝒜t! (both orientations) — plus f!, g!, fg!,
#     dual_obj, approx_mineigval_lanczos, SDP_S_eigval and DIMACS_errors, so that every generic caller in
#     src/coreop.jl / src/linesearch.jl works on a HIPAux, and
#   * a method of `_sdplr(data, var, aux::HIPAux, stats, config)` (src/sdplr.jl:140-449) whose inner `while`
#     (:190-278) is ONE call — the loop runs device-driven behind sdplr_hip_inner_loop — and
#   * `sdplr_hip(C, As, b, r; kwargs...)`: `sdplr` (src/sdplr.jl:91-138) with the HIPAux in place of aux.
# Arrays cross the boundary in the reference's own memory layout: `Rt` is r×n column-major, index vectors are the
# 1-based Int64 vectors of SolverAuxiliary (index_base = 1).  Every ccall below is checked against
# include/sdplr_hip.h by tests/test_julia_binding.py (symbol, argument count, C types); the file itself could not be
# executed in the build image (no Julia there).

const LIBSDPLR_HIP = get(ENV, "LIBSDPLR_HIP", "libsdplr_hip.so")

# ---- slot ids of include/sdplr_hip.h ---------------------------------------------------------------------------
const HIP_F_RT = Int32(0)
const HIP_F_GT = Int32(1)
const HIP_F_DIRT = Int32(2)
const HIP_F_SCRATCH = Int32(300)
const HIP_V_LAMBDA = Int32(0)
const HIP_V_LAMBDA_UB = Int32(1)
const HIP_V_B = Int32(2)
const HIP_V_Y = Int32(3)
const HIP_V_PV_RAW = Int32(4)
const HIP_V_PV_LB = Int32(5)
const HIP_V_PV = Int32(6)
const HIP_V_SCRATCH = Int32(14)
const HIP_S_SIGMA = Int32(0)
const HIP_S_OBJ = Int32(1)

"status → exception, with the library's message (same failure mode as the reference's `error(...)`, src/linesearch.jl:60-62)"
function hip_check(rc::Int32, h::Ptr{Cvoid})
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:sdplr_hip_last_error, LIBSDPLR_HIP), Cstring, (Ptr{Cvoid},), h))
    error(msg)
end

"""
    HIPAux(data, aux, r, numlbfgsvecs)

The device twin of `SolverAuxiliary` + the device-resident `SolverVars`/`LBFGSHistory` (src/structs.jl:194-294,
src/lbfgs.jl:21-47): an opaque handle.  The host `aux` it was built from is kept for `rank_update!` and sizes.
"""
mutable struct HIPAux
    handle::Ptr{Cvoid}
    n::Int
    m::Int
    host::Any
    # HIPAux(data, r, numlbfgsvecs): preprocess_sparsecons (src/preprocess.jl:24-169) runs INSIDE the library — the sparse
    # matrices go over as concatenated COO triplets in findnz order (sdplr_hip_set_sparse_coo), no SolverAuxiliary is built
    function HIPAux(data, r::Integer, numlbfgsvecs::Integer)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        n, m = Int(data.n), Int(data.m)
        hip_check(ccall((:sdplr_hip_create, LIBSDPLR_HIP), Int32,
                        (Int64, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}), n, m, r, numlbfgsvecs, h), C_NULL)
        hd = h[]
        ent_ptr, I, J, V, gids = Int64[1], Int64[], Int64[], Float64[], Int64[]
        lowrank = Tuple{Any,Int}[]
        for (gid, A) in Iterators.flatten((enumerate(data.As), ((m + 1, data.C),)))      # src/structs.jl:303-334
            if A isa SymLowRankMatrix
                push!(lowrank, (A, gid))
                continue
            end
            i, j, v = findnz(A isa Diagonal ? sparse(A) : A)
            append!(I, i); append!(J, j); append!(V, v)
            push!(ent_ptr, length(I) + 1)
            push!(gids, gid)
        end
        if !isempty(gids)
            hip_check(ccall((:sdplr_hip_set_sparse_coo, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
                hd, 1, length(gids), ent_ptr, I, J, V, gids), hd)
        end
        for (A, gid) in lowrank
            d = collect(Float64, diag(A.D))
            B = Matrix{Float64}(A.B)
            hip_check(ccall((:sdplr_hip_add_symlowrank, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}), hd, 1, gid, size(B, 2), B, d), hd)
        end
        hip_check(ccall((:sdplr_hip_finalize, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), hd), hd)
        obj = new(hd, n, m, nothing)
        finalizer(x -> ccall((:sdplr_hip_destroy, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), x.handle), obj)
        return obj
    end
    function HIPAux(data, aux, r::Integer, numlbfgsvecs::Integer)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        n = size(aux.sparse_S, 1)
        m = length(b_vector(data))
        hip_check(ccall((:sdplr_hip_create, LIBSDPLR_HIP), Int32,
                        (Int64, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}), n, m, r, numlbfgsvecs, h), C_NULL)
        hd = h[]
        if aux.n_sparse_matrices > 0
            hip_check(ccall((:sdplr_hip_set_sparse, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64},
                 Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                hd, 1, aux.n_sparse_matrices, aux.triu_agg_sparse_A_matptr, aux.triu_agg_sparse_A_nzind,
                aux.triu_agg_sparse_A_nzval_one, aux.triu_agg_sparse_A_nzval_two, aux.sparse_As_global_inds,
                length(aux.triu_sparse_S.rowval), aux.triu_sparse_S.colptr, aux.triu_sparse_S.rowval,
                length(aux.sparse_S.rowval), aux.sparse_S.colptr, aux.sparse_S.rowval,
                aux.agg_sparse_A_mappedto_triu), hd)
        end
        for (A, gid) in zip(aux.symlowrank_As, aux.symlowrank_As_global_inds)
            d = collect(Float64, diag(A.D))
            B = Matrix{Float64}(A.B)
            hip_check(ccall((:sdplr_hip_add_symlowrank, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}), hd, 1, gid, size(B, 2), B, d), hd)
        end
        hip_check(ccall((:sdplr_hip_finalize, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), hd), hd)
        obj = new(hd, n, m, aux)
        finalizer(x -> ccall((:sdplr_hip_destroy, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), x.handle), obj)
        return obj
    end
end

side_dimension(aux::HIPAux) = aux.n                                       # src/structs.jl:363

# ---- state transfer -------------------------------------------------------------------------------------------
hip_set_factor!(aux::HIPAux, slot::Int32, A::Matrix{Float64}) = hip_check(
    ccall((:sdplr_hip_set_factor, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), aux.handle, slot, A), aux.handle)
hip_get_factor!(A::Matrix{Float64}, aux::HIPAux, slot::Int32) = hip_check(
    ccall((:sdplr_hip_get_factor, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), aux.handle, slot, A), aux.handle)
hip_set_vec!(aux::HIPAux, slot::Int32, v::Vector{Float64}) = hip_check(
    ccall((:sdplr_hip_set_vec, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64), aux.handle, slot, v, length(v)), aux.handle)
hip_get_vec!(v::Vector{Float64}, aux::HIPAux, slot::Int32) = hip_check(
    ccall((:sdplr_hip_get_vec, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64), aux.handle, slot, v, length(v)), aux.handle)
hip_set_scalar!(aux::HIPAux, slot::Int32, x::Float64) = hip_check(
    ccall((:sdplr_hip_set_scalar, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Float64), aux.handle, slot, x), aux.handle)
function hip_get_scalar(aux::HIPAux, slot::Int32)
    x = Ref(0.0)
    hip_check(ccall((:sdplr_hip_get_scalar, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), aux.handle, slot, x), aux.handle)
    return x[]
end

"upload the state `SolverVars(data, r, config)` created on the host (src/structs.jl:225-263); the handle's L-BFGS history starts cleared (lbfgs_init, src/lbfgs.jl:35-47)"
function upload!(aux::HIPAux, data, var::SolverVars)
    hip_set_factor!(aux, HIP_F_RT, var.Rt)
    hip_set_vec!(aux, HIP_V_LAMBDA, var.λ)
    hip_set_vec!(aux, HIP_V_LAMBDA_UB, var.λ_ub)
    hip_set_vec!(aux, HIP_V_B, collect(Float64, b_vector(data)))
    hip_set_vec!(aux, HIP_V_PV_LB, var.primal_vio_lb)
    hip_set_scalar!(aux, HIP_S_SIGMA, var.σ[])
    return nothing
end

"bring the host `var` up to date with the device (for the result Dict, DIMACS_errors, user callbacks)"
function download!(var::SolverVars, aux::HIPAux)
    hip_get_factor!(var.Rt, aux, HIP_F_RT)
    hip_get_factor!(var.Gt, aux, HIP_F_GT)
    hip_get_vec!(var.λ, aux, HIP_V_LAMBDA)
    hip_get_vec!(var.y, aux, HIP_V_Y)
    hip_get_vec!(var.primal_vio_raw, aux, HIP_V_PV_RAW)
    hip_get_vec!(var.primal_vio, aux, HIP_V_PV)
    var.obj[] = hip_get_scalar(aux, HIP_S_OBJ)
    var.σ[] = hip_get_scalar(aux, HIP_S_SIGMA)
    return var
end

# ---- the plug-in operator set (cf. src/lowrankopt.jl:57-135) ----------------------------------------------------
"𝒜!(out, aux, Ut) — src/coreop.jl:36-49, on the caller's own matrix through the scratch slots"
function 𝒜!(out::Vector{Float64}, aux::HIPAux, Ut::Matrix{Float64})
    hip_set_factor!(aux, HIP_F_SCRATCH, Ut)
    hip_check(ccall((:sdplr_hip_A, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Int32, Int32),
                    aux.handle, HIP_F_SCRATCH, Int32(-1), HIP_V_SCRATCH), aux.handle)
    hip_get_vec!(out, aux, HIP_V_SCRATCH)
    return out
end

"𝒜!(out, aux, Ut, Vt) — src/coreop.jl:54-70: 𝒜((UVᵀ + VUᵀ)/2); the caller doubles it (src/linesearch.jl:13)"
function 𝒜!(out::Vector{Float64}, aux::HIPAux, Ut::Matrix{Float64}, Vt::Matrix{Float64})
    hip_set_factor!(aux, HIP_F_SCRATCH, Ut)
    hip_set_factor!(aux, HIP_F_SCRATCH + Int32(1), Vt)
    hip_check(ccall((:sdplr_hip_A, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Int32, Int32),
                    aux.handle, HIP_F_SCRATCH, HIP_F_SCRATCH + Int32(1), HIP_V_SCRATCH), aux.handle)
    hip_get_vec!(out, aux, HIP_V_SCRATCH)
    return out
end

"𝒜t_preprocess!(var, aux) — src/coreop.jl:248-258: S.nzval from var.y"
function 𝒜t_preprocess!(var::SolverVars, aux::HIPAux)
    hip_set_vec!(aux, HIP_V_Y, var.y)
    hip_check(ccall((:sdplr_hip_At_preprocess, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), aux.handle), aux.handle)
    return nothing
end

"𝒜t!(y, x, aux, var) — src/coreop.jl:260-279: y = x·S + Σ coeff·x·B·D·Bᵀ on r×n matrices"
function 𝒜t!(y::Matrix{Float64}, x::Matrix{Float64}, aux::HIPAux, var::SolverVars)
    hip_set_factor!(aux, HIP_F_SCRATCH, x)
    hip_check(ccall((:sdplr_hip_At_left, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Int32),
                    aux.handle, HIP_F_SCRATCH + Int32(1), HIP_F_SCRATCH), aux.handle)
    hip_get_factor!(y, aux, HIP_F_SCRATCH + Int32(1))
    return y
end

"𝒜t!(y, aux, x, var) — src/coreop.jl:281-300: y = S·x + Σ coeff·B·D·Bᵀ·x on an n-vector or n×k matrix"
function 𝒜t!(y::StridedVecOrMat{Float64}, aux::HIPAux, x::StridedVecOrMat{Float64}, var::SolverVars)
    hip_check(ccall((:sdplr_hip_At_right, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                    aux.handle, x, y, size(x, 2)), aux.handle)
    return y
end

"f!(data, var, aux) — src/coreop.jl:11-31; the device state is the argument, the host `var` receives the scalars"
function f!(data, var::SolverVars, aux::HIPAux)
    L = Ref(0.0)
    hip_check(ccall((:sdplr_hip_f, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Ptr{Float64}), aux.handle, L), aux.handle)
    var.obj[] = hip_get_scalar(aux, HIP_S_OBJ)
    return L[]
end

"g!(var, aux) — src/coreop.jl:305-317"
function g!(var::SolverVars, aux::HIPAux)
    hip_check(ccall((:sdplr_hip_g, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), aux.handle), aux.handle)
    return nothing
end

"fg!(data, var, aux, normC, normb, config) — src/coreop.jl:323-349"
function fg!(data, var::SolverVars, aux::HIPAux, normC, normb, config)
    L, g, p = Ref(0.0), Ref(0.0), Ref(0.0)
    hip_check(ccall((:sdplr_hip_fg, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    aux.handle, normC, normb, config.gtol_mode == :relative, config.ptol_mode == :relative, L, g, p), aux.handle)
    return L[], g[], p[]
end

"the inner `while` of _sdplr — src/sdplr.jl:190-278 — as one call → (ℒ, ‖grad‖, ‖pv‖, α, iterations, exit_reason)"
function inner_loop!(aux::HIPAux, normC, normb, config, use_armijo::Bool, cur_gtol, budget::Integer, time_left, L, g, p)
    Lr, gr, pr, ar, it, why = Ref(Float64(L)), Ref(Float64(g)), Ref(Float64(p)), Ref(0.0), Ref{Int64}(0), Ref{Int32}(0)
    hip_check(ccall((:sdplr_hip_inner_loop, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Int32, Float64, Float64, Int64, Float64,
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
                    aux.handle, normC, normb, config.gtol_mode == :relative, config.ptol_mode == :relative, use_armijo,
                    cur_gtol, config.fprec * eps(), budget, time_left, Lr, gr, pr, ar, it, why), aux.handle)
    return Lr[], gr[], pr[], ar[], Int(it[]), Int(why[])
end

"""
    major_iteration!(aux, normC, normb, config, use_armijo, update_λ, σ, cur_gtol, budget, time_left)

The device work between two host decisions of `_sdplr` as ONE call: [λ update, src/sdplr.jl:358-362] → `var.σ[] = σ` →
`lbfgs_clear!` (:384) → `fg!` (:389) → the inner `while` (:190-278) on what `fg!` returned.  On small instances the
library runs all of it as one kernel launch.
"""
function major_iteration!(aux::HIPAux, normC, normb, config, use_armijo::Bool, update_λ::Bool, σ, cur_gtol,
                          budget::Integer, time_left)
    Lr, gr, pr, ar, it, why = Ref(0.0), Ref(0.0), Ref(0.0), Ref(0.0), Ref{Int64}(0), Ref{Int32}(0)
    hip_check(ccall((:sdplr_hip_major_iteration, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Int32, Int32, Float64, Float64, Float64, Int64, Float64,
                     Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
                    aux.handle, normC, normb, config.gtol_mode == :relative, config.ptol_mode == :relative, use_armijo,
                    update_λ, σ, cur_gtol, config.fprec * eps(), budget, time_left, Lr, gr, pr, ar, it, why), aux.handle)
    return Lr[], gr[], pr[], ar[], Int(it[]), Int(why[])
end

"primes the library's pools (HIP streams, events, pinned staging) for `n` handles alive at once — optional, before a batch"
hip_warmup(n::Integer) = hip_check(ccall((:sdplr_hip_warmup, LIBSDPLR_HIP), Int32, (Int32,), n), C_NULL)
# what the library's pools cache and no handle uses goes back to the HIP runtime (a process that shares the GPU)
hip_trim_pools() = hip_check(ccall((:sdplr_hip_trim_pools, LIBSDPLR_HIP), Int32, ()), C_NULL)
# one rank = one GPU: sticky — every later call of this process, from any task / thread, is bound to the device
hip_set_device(dev::Integer) = hip_check(ccall((:sdplr_hip_set_device, LIBSDPLR_HIP), Int32, (Int32,), dev), C_NULL)
# `update_lambda` of sdplr_hip_major_iteration / HIPMajorItem: continue a loop that ran out of its iteration budget
const SDPLR_MAJOR_RESUME = Int32(2)

"approx_mineigval_lanczos(var, aux, q) — src/coreop.jl:461-514 (the start vector replaces the internal randn of :473)"
function approx_mineigval_lanczos(var::SolverVars, aux::HIPAux, q::Integer)
    v0 = randn(side_dimension(aux))
    ev = Ref(0.0)
    hip_check(ccall((:sdplr_hip_approx_mineigval_lanczos, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}), aux.handle, q, v0, ev), aux.handle)
    return ev[]
end

"SDP_S_eigval(var, aux, nevs, preprocessed; which, ncv, tol, maxiter) — src/coreop.jl:351-374, solver on the device"
function SDP_S_eigval(var::SolverVars, aux::HIPAux, nevs::Integer, preprocessed::Bool=false;
                      which::Symbol=:SA, ncv::Integer=min(100, side_dimension(aux)), tol::Real=0.0, maxiter::Integer=1000000)
    preprocessed || 𝒜t_preprocess!(var, aux)
    evs = zeros(nevs)
    mv, nc = Ref{Int64}(0), Ref{Int64}(0)
    dt = @elapsed hip_check(ccall((:sdplr_hip_S_eigval, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Int64, Int32, Int64, Float64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}),
                    aux.handle, nevs, which == :SA ? 0 : 1, ncv, tol, min(maxiter, 100000), C_NULL, evs, mv, nc), aux.handle)
    nc[] < nevs && @warn "SDP_S_eigval: $(nc[]) of $nevs eigenvalues converged."
    return sort(evs), dt
end

"dual_obj(data, var, aux, trace_bound, iter; highprecision) — src/coreop.jl:376-415"
function dual_obj(data, var::SolverVars, aux::HIPAux, trace_bound, iter::Integer; highprecision::Bool=false)
    b = b_vector(data)
    m = length(b)
    if highprecision                                                       # :389-400
        g!(var, aux)                                                       # leaves y = copy2y_λ_sub_pvio!, S current (:384-385)
        hip_get_vec!(var.y, aux, HIP_V_Y)
        evs, _ = SDP_S_eigval(var, aux, 1, true; which=:SA, ncv=min(100, side_dimension(aux)), tol=1e-6, maxiter=1000000)
        return -dot(view(var.y, 1:m), b) + trace_bound * min(evs[1], 0.0), evs[1]
    end
    v0 = randn(side_dimension(aux))
    d, e = Ref(0.0), Ref(0.0)
    hip_check(ccall((:sdplr_hip_dual_obj, LIBSDPLR_HIP), Int32,
                    (Ptr{Cvoid}, Float64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                    aux.handle, trace_bound, iter, v0, d, e), aux.handle)
    return d[], e[]
end

"DIMACS_errors(data, var, aux) — src/coreop.jl:426-453; `var` must be current (download!)"
function DIMACS_errors(data, var::SolverVars, aux::HIPAux)
    b = b_vector(data)
    m = length(b)
    err1 = norm(view(var.primal_vio_raw, 1:m), 2) / (1.0 + norm(b, 2))
    @. var.y[1:m] = -var.λ                                                 # copy2y_λ! (:238-246)
    var.y[m + 1] = 1.0
    𝒜t_preprocess!(var, aux)                                               # (:436)
    evs, _ = SDP_S_eigval(var, aux, 1, true; which=:SA, ncv=min(100, side_dimension(aux)), maxiter=1000000)
    err4 = max(0.0, -evs[1]) / (1.0 + norm(C_matrix(data), 2))
    λb = dot(var.λ, b)
    err5 = (var.obj[] - λb) / (1.0 + abs(var.obj[]) + abs(λb))
    hip_check(ccall((:sdplr_hip_At_left, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Int32),
                    aux.handle, HIP_F_SCRATCH, HIP_F_RT), aux.handle)      # Rt·S (:449) …
    xz = Ref(0.0)
    hip_check(ccall((:sdplr_hip_factor_dot, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Float64}),
                    aux.handle, HIP_F_RT, HIP_F_SCRATCH, xz), aux.handle)  # … and its dot with Rt, on the device
    err6 = xz[] / (1.0 + abs(var.obj[]) + abs(λb))
    return [err1, 0.0, 0.0, err4, err5, err6]
end

# ---- the solve loop -------------------------------------------------------------------------------------------
"""
    _sdplr(data, var, aux::HIPAux, stats, config)

src/sdplr.jl:140-449 for the device backend: the same major-iteration schedule (tolerances, σ/λ updates, duality-gap
test, rank doubling, result Dict), with the inner L-BFGS loop (:190-278) run by the device in one call, the λ update
(:358-362) and lbfgs_clear! (:384) as entry points, and `var` refreshed from the device at the end.
"""
function _sdplr(data, var::SolverVars{Ti,Tv}, aux::HIPAux, stats::SolverStats{Tv},
                config::BurerMonteiroConfig{Ti,Tv}) where {Ti<:Integer,Tv}
    h = aux.handle
    m = length(var.λ)
    stats.starttime[] = time()
    Rt0, λ0 = copy(var.Rt), copy(var.λ)
    normb = norm(b_vector(data), 2)
    normC = norm(C_matrix(data), 2)
    upload!(aux, data, var)

    σ = var.σ[]
    cur_gtol = max(1.0 / σ, config.gtol)                                   # :165-169
    cur_ptol = max(1.0 / σ^0.1, config.ptol)
    𝓛, gnorm, pnorm = NaN, NaN, NaN     # the fg! of :170 rides the first major_iteration! (see `pending` below)
    iter, majoriter, localiter = 0, 0, 0
    use_armijo = data.has_inequalities                                     # :176
    stall_left = config.rankupd_tol
    min_gap, best_dual = 1e20, -1e20
    best_λ = copy(var.λ)
    obj = NaN
    report(li) = printintermediate(config.dataset, majoriter, li, iter, 𝓛, obj, σ, cur_gtol, cur_ptol, gnorm, pnorm,
                                   min_gap, best_dual)

    # The tail of a major iteration — λ update or σ increase (:358-369), lbfgs_clear! (:384), fg! (:389) — has no host
    # decision in it, nor has the while loop it feeds: the four travel as ONE call (major_iteration!); `pending` holds a
    # tail that has not been sent yet: (update_λ, σ).  The fg! of :170 and the first pass of the loop are such a call too:
    # no λ update, σ as it stands, and lbfgs_clear! on the fresh history of lbfgs_init (:163) changes nothing.
    pending = (false, σ)
    for _ in 1:config.maxmajoriter                                         # :185
        majoriter += 1
        localiter = 0
        budget = max(config.maxiter + 1 - iter, 1)
        time_left = max(config.maxtime - (time() - stats.starttime[]), 1e-9)
        if pending !== nothing
            𝓛, gnorm, pnorm, _, localiter, _ = major_iteration!(aux, normC, normb, config, use_armijo, pending[1],
                                                                pending[2], cur_gtol, budget, time_left)
            pending = nothing
            iter += localiter
        elseif gnorm > cur_gtol                                            # the while of :190-278, device-driven
            𝓛, gnorm, pnorm, _, localiter, _ = inner_loop!(aux, normC, normb, config, use_armijo, cur_gtol, budget,
                                                           time_left, 𝓛, gnorm, pnorm)
            iter += localiter
        end
        obj = hip_get_scalar(aux, HIP_S_OBJ)
        report(localiter)                                                  # :281-296
        if time() - stats.starttime[] > config.maxtime                     # :298-301
            @warn "Time limit exceeded. Stop optimizing."
            break
        end
        if iter > config.maxiter                                           # :303-306
            @warn "Iteration limit exceeded. Stop optimizing."
            break
        end

        grow_rank = false
        if pnorm <= cur_ptol                                               # :310
            dual_dt = @elapsed begin
                dual_value, _ = dual_obj(data, var, aux, config.prior_trace_bound, iter;
                                         highprecision=config.eigval_highprecision)
            end
            if dual_value > best_dual                                      # :324-327
                hip_get_vec!(var.y, aux, HIP_V_Y)
                best_λ = -copy(var.y)
                best_dual = dual_value
            end
            gap = config.objtol_mode == :relative ? (obj - best_dual) / min(abs(obj), abs(best_dual)) : obj - best_dual
            stats.dual_time[] += dual_dt
            @show obj best_dual gap
            if pnorm <= config.ptol                                        # :335-357
                config.objtol == Inf && break
                if gap <= config.objtol
                    min_gap = min(min_gap, gap)
                    break
                end
                stall_left = (min_gap - gap < config.objtol) ? stall_left - 1 : config.rankupd_tol
                min_gap = min(min_gap, gap)
                grow_rank = stall_left == 0
            end
            update_λ = true                                                # :358-362
            cur_ptol /= σ^0.9                                              # :363-364
            cur_gtol /= σ
        else
            update_λ = false
            σ *= config.σfac                                               # :366-369
            cur_ptol, cur_gtol = 1 / σ^0.1, 1 / σ
        end
        fuse_tail = !grow_rank && majoriter < config.maxmajoriter
        if !fuse_tail
            if update_λ
                hip_check(ccall((:sdplr_hip_update_lambda, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), h), h)
            else
                hip_set_scalar!(aux, HIP_S_SIGMA, σ)
            end
        end

        if grow_rank                                                       # :373-382
            var = rank_update!(data, var, config)                          # a fresh host SolverVars at the doubled rank
            hip_check(ccall((:sdplr_hip_reset_rank, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int64), h, var.r[]), h)
            upload!(aux, data, var)
            σ = var.σ[]
            cur_ptol, cur_gtol = 1 / σ^0.1, 1 / σ
            min_gap, best_dual = 1e20, -1e20
            stall_left = config.rankupd_tol
            @info "rank doubled, newrank is $(var.r[])."
        elseif !fuse_tail
            hip_check(ccall((:sdplr_hip_lbfgs_clear, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), h), h)     # :384
        end
        cur_ptol = max(cur_ptol, config.ptol)                              # :387-389
        cur_gtol = max(cur_gtol, config.gtol)
        if fuse_tail
            pending = (update_λ, σ)                                        # sent with the next pass of the while loop
        else
            𝓛, gnorm, pnorm = fg!(data, var, aux, normC, normb, config)
        end
        majoriter == config.maxmajoriter && @warn "Major iteration limit exceeded. Stop optimizing."
    end

    𝓛, gnorm, pnorm = fg!(data, var, aux, normC, normb, config)            # :396
    download!(var, aux)
    obj = var.obj[]
    report(-1)
    stats.endtime[] = time()
    totaltime = stats.endtime[] - stats.starttime[]
    stats.primal_time[] = totaltime - stats.dual_time[]
    DIMACS_errs = zeros(6)
    stats.DIMACS_time[] = @elapsed begin                                   # :419-425
        if config.eval_DIMACS_errs
            DIMACS_errs = DIMACS_errors(data, var, aux)
        end
    end
    return Dict([                                                          # :426-448
        "Rt" => var.Rt, "lambda" => best_λ, "Rt0" => Rt0, "lambda0" => λ0, "sigma" => var.σ[],
        "grad_norm" => gnorm, "primal_vio" => pnorm, "obj" => var.obj[], "max_dual_value" => best_dual,
        "min_duality_gap" => min_gap, "totaltime" => totaltime, "dual_time" => stats.dual_time[],
        "primaltime" => stats.primal_time[], "iter" => iter, "majoriter" => majoriter, "DIMACS_errs" => DIMACS_errs,
        "ptol" => config.ptol, "objtol" => config.objtol, "fprec" => config.fprec,
        "rankupd_tol" => config.rankupd_tol, "r" => size(var.Rt, 1),
    ])
end

"""
    sdplr_hip(C, As, b, r; constraint_types=nothing, config=BurerMonteiroConfig{Int,Float64}(), kwargs...)

`sdplr` (src/sdplr.jl:91-138) on the MI355X backend: same arguments, same result Dict.
"""
function sdplr_hip(C::AbstractMatrix{Tv}, As::Vector, b::Vector{Tv}, r::Ti;
                   constraint_types::Union{Nothing,AbstractVector{Bool}}=nothing,
                   config::BurerMonteiroConfig{Ti,Tv}=BurerMonteiroConfig{Ti,Tv}(), kwargs...) where {Ti<:Integer,Tv}
    for (key, value) in kwargs                                             # :102-108
        if hasfield(BurerMonteiroConfig, Symbol(key))
            setfield!(config, Symbol(key), value)
        else
            @error "Unrecognized keyword argument $key"
        end
    end
    config.printlevel > 0 && printheading(1)
    preprocess_dt = @elapsed begin
        data = constraint_types === nothing ? SDPData(C, As, b) : SDPData(C, As, b, constraint_types)
        var = SolverVars(data, r, config)
        aux = HIPAux(data, SolverAuxiliary(data), r, config.numlbfgsvecs)
        stats = SolverStats{Tv}()
    end
    ans = _sdplr(data, var, aux, stats, config)
    ans["preprocess_time"] = preprocess_dt
    ans["totaltime"] += preprocess_dt
    config.printlevel > 0 && printheading(0)
    return ans
end
