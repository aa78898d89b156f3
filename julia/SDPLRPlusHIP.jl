# SDPLRPlusHIP.jl — the reference-side binding for libsdplr_hip.so (UNTESTED: no Julia in the build image).
#
# Drop this file next to src/SDPLRPlus.jl and `include("SDPLRPlusHIP.jl")` after the other includes.
# It adds a second method of `_sdplr` for a `HIPAux` argument; everything above `_sdplr` (`sdplr`,
# SDPData, BurerMonteiroConfig, result Dict) is unchanged.  All arrays cross the boundary in the
# reference's own memory layout: `Rt` is r×n column-major, indices are the 1-based Int64 vectors of
# SolverAuxiliary (index_base = 1).

const LIBSDPLR_HIP = get(ENV, "LIBSDPLR_HIP", "libsdplr_hip.so")

mutable struct HIPAux
    handle::Ptr{Cvoid}
    n::Int
    function HIPAux(data::SDPData{Ti,Tv}, aux::SolverAuxiliary{Ti,Tv}, r::Int, numlbfgsvecs::Int) where {Ti,Tv}
        h = Ref{Ptr{Cvoid}}(C_NULL)
        n = size(aux.sparse_S, 1)
        check(ccall((:sdplr_hip_create, LIBSDPLR_HIP), Int32,
                    (Int64, Int64, Int64, Int64, Ptr{Ptr{Cvoid}}), n, data.m, r, numlbfgsvecs, h), C_NULL)
        hd = h[]
        if aux.n_sparse_matrices > 0
            GC.@preserve aux check(ccall((:sdplr_hip_set_sparse, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64},
                 Int64, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                hd, 1, aux.n_sparse_matrices, aux.triu_agg_sparse_A_matptr, aux.triu_agg_sparse_A_nzind,
                aux.triu_agg_sparse_A_nzval_one, aux.triu_agg_sparse_A_nzval_two, aux.sparse_As_global_inds,
                length(aux.triu_sparse_S.rowval), aux.triu_sparse_S.colptr, aux.triu_sparse_S.rowval,
                length(aux.sparse_S.rowval), aux.sparse_S.colptr, aux.sparse_S.rowval,
                aux.agg_sparse_A_mappedto_triu), hd)
        end
        for (A, gid) in zip(aux.symlowrank_As, aux.symlowrank_As_global_inds)
            d = collect(diag(A.D))
            GC.@preserve A d check(ccall((:sdplr_hip_add_symlowrank, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Float64}, Ptr{Float64}),
                hd, 1, gid, size(A.B, 2), A.B, d), hd)
        end
        check(ccall((:sdplr_hip_finalize, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), hd), hd)
        obj = new(hd, n)
        finalizer(x -> ccall((:sdplr_hip_destroy, LIBSDPLR_HIP), Int32, (Ptr{Cvoid},), x.handle), obj)
        return obj
    end
end

function check(rc::Int32, h)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:sdplr_hip_last_error, LIBSDPLR_HIP), Cstring, (Ptr{Cvoid},), h))
    error(msg)          # same failure mode as the reference's error(...) (src/linesearch.jl:60-62)
end

side_dimension(aux::HIPAux) = aux.n                                       # src/structs.jl:363

# slot ids of include/sdplr_hip.h
const F_RT, V_LAMBDA, V_LAMBDA_UB, V_B, V_PV_LB, S_SIGMA = 0, 0, 1, 2, 5, 0

"upload the state SolverVars(data, r, config) created on the host (src/structs.jl:225-263)"
function upload!(aux::HIPAux, data, var::SolverVars)
    h = aux.handle
    GC.@preserve var data begin
        check(ccall((:sdplr_hip_set_factor, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}), h, F_RT, var.Rt), h)
        for (slot, v) in ((V_LAMBDA, var.λ), (V_LAMBDA_UB, var.λ_ub), (V_B, b_vector(data)), (V_PV_LB, var.primal_vio_lb))
            check(ccall((:sdplr_hip_set_vec, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Int64), h, slot, v, length(v)), h)
        end
    end
    check(ccall((:sdplr_hip_set_scalar, LIBSDPLR_HIP), Int32, (Ptr{Cvoid}, Int32, Float64), h, S_SIGMA, var.σ[]), h)
end

"fg! — src/coreop.jl:323-349"
function fg!(data, var::SolverVars, aux::HIPAux, normC, normb, config)
    L, g, p = Ref(0.0), Ref(0.0), Ref(0.0)
    check(ccall((:sdplr_hip_fg, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                aux.handle, normC, normb, config.gtol_mode == :relative, config.ptol_mode == :relative, L, g, p), aux.handle)
    return L[], g[], p[]
end

"the inner while loop of _sdplr — src/sdplr.jl:190-278 — as one call; returns (ℒ, ‖grad‖, ‖pv‖, α, iterations, exit_reason)"
function inner_loop!(aux::HIPAux, normC, normb, config, use_armijo, cur_gtol, budget, time_left, L, g, p)
    Lr, gr, pr, ar, it, why = Ref(L), Ref(g), Ref(p), Ref(0.0), Ref{Int64}(0), Ref{Int32}(0)
    check(ccall((:sdplr_hip_inner_loop, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Float64, Float64, Int32, Int32, Int32, Float64, Float64, Int64, Float64,
                 Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int32}),
                aux.handle, normC, normb, config.gtol_mode == :relative, config.ptol_mode == :relative, use_armijo,
                cur_gtol, config.fprec * eps(), budget, time_left, Lr, gr, pr, ar, it, why), aux.handle)
    return Lr[], gr[], pr[], ar[], it[], why[]
end

"dual_obj — src/coreop.jl:376-415 (v0 replaces the internal randn of :473)"
function dual_obj(data, var::SolverVars, aux::HIPAux, trace_bound, iter; highprecision=false)
    highprecision && error("eigval_highprecision is not offloaded")
    v0 = randn(side_dimension(aux)); d, e = Ref(0.0), Ref(0.0)
    GC.@preserve v0 check(ccall((:sdplr_hip_dual_obj, LIBSDPLR_HIP), Int32,
                (Ptr{Cvoid}, Float64, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                aux.handle, trace_bound, iter, v0, d, e), aux.handle)
    return d[], e[]
end

# `_sdplr(data, var, aux::HIPAux, stats, config)` is src/sdplr.jl:140-449 with
#   * the `while` block :190-278 replaced by `inner_loop!` (iter += iterations done),
#   * :358-362 by `ccall(:sdplr_hip_update_lambda …)`, :384 by `ccall(:sdplr_hip_lbfgs_clear …)`,
#   * var.σ[] writes mirrored with `sdplr_hip_set_scalar(h, S_SIGMA, σ)`,
#   * rank_update! (:373-382) followed by `sdplr_hip_reset_rank(h, newr)` + `upload!`,
#   * `var.Rt`, `var.λ`, `var.obj[]` read back with sdplr_hip_get_factor / get_vec / get_scalar for the result Dict.
# sdplrplus.jl_amd/sdplr.py is that function, line for line, in Python.
