# kkt_solver_hip.jl -- Julia glue for the device-resident KKT-system level of libonephase_kkt.so.
# UNTESTED IN THIS REPOSITORY (no Julia in the build environment); mirrors symmetric.jl / schur.jl of the
# reference and implements the four methods every abstract_KKT_system_solver must define
# (src/kkt_system_solver/kkt_system_solver.jl:13-17).  See INTEGRATION.md.

mutable struct HIP_KKT_solver <: abstract_KKT_system_solver
    ls_solver::abstract_linear_system_solver    # unused (the library owns the factorisation); kept for field parity
    factor_it::Class_iterate
    delta_x_vec::Array{Float64,1}
    delta_s_vec::Array{Float64,1}
    rhs::System_rhs
    dir::Class_point
    kkt_err_norm::Class_kkt_error
    rhs_norm::Float64
    pars::Class_parameters
    schur_diag::Array{Float64,1}
    ready::Symbol
    Q::SparseMatrixCSC{Float64,Int64}           # left empty: the assembled matrix lives in HBM
    handle::Ptr{Cvoid}
    kind::Cint                                   # 0 = :schur, 1 = :symmetric, 2 = :clever_symmetric, 3 = :schur_direct
    pattern_set::Bool
    current_it::Class_iterate                    # iterate of the last kkt_associate_rhs! (schur.jl:34-45)
    reduct_factors::Class_reduction_factors
    resident_rhs::Any                            # the System_rhs object whose triple okkt_kkt_system_rhs left in HBM (nothing: none)

    function HIP_KKT_solver(kind::Symbol, opts::Union{Nothing,OkktOpts}=nothing)      # opts: okkt_opts_from_pars(pars.kkt), linear_solver_hip.jl
        this = new()
        this.ready = :not_ready
        this.kind = Dict(:schur => 0, :symmetric => 1, :clever_symmetric => 2, :schur_direct => 3)[kind]
        this.pattern_set = false
        this.resident_rhs = nothing
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = opts === nothing ?
             ccall((:okkt_kkt_create, OKKT_LIB), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Cint), h, C_NULL, this.kind) :
             ccall((:okkt_kkt_create, OKKT_LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{OkktOpts}, Cint), h, Ref(opts), this.kind)
        rc == 0 || error("okkt_kkt_create failed with code $rc")
        this.handle = h[]
        finalizer(s -> ccall((:okkt_kkt_destroy, OKKT_LIB), Cint, (Ptr{Cvoid},), s.handle), this)
        return this
    end
end

struct OkktKktPars      # okkt_kkt_pars of include/okkt.h
    delta_start::Float64
    delta_min::Float64
    delta_max::Float64
    delta_inc::Float64
    delta_dec::Float64
    delta_zero::Float64
    ItRefine_Num::Int32
    max_it::Int32
end

function kkt_hip_check(k::HIP_KKT_solver, what::String, rc)
    rc < 0 && error("$what failed ($rc): " * unsafe_string(ccall((:okkt_kkt_last_error, OKKT_LIB), Cstring, (Ptr{Cvoid},), k.handle)))
    return rc
end

# initialize!(::Clever_Symmetric_KKT_solver, it) (clever_symmetric.jl:53-61): the parallel-row grouping is computed
# once, from the Jacobian of the initial iterate
function initialize!(k::HIP_KKT_solver, intial_it::Class_iterate)
    k.dir = zero_point(dim(intial_it), ncon(intial_it))
    if k.kind == 2
        H = get_lag_hess(intial_it); J = get_jac(intial_it)
        kkt_hip_check(k, "okkt_kkt_set_structure", ccall((:okkt_kkt_set_structure, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Cint),
            k.handle, dim(intial_it), ncon(intial_it), H.colptr, H.rowval, J.colptr, J.rowval, 1))
        k.pattern_set = true
        m_new = Ref{Int64}(0)
        kkt_hip_check(k, "okkt_kkt_compute_indicies", ccall((:okkt_kkt_compute_indicies, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ref{Int64}), k.handle, J.nzval, m_new))
    end
end

function form_system!(k::HIP_KKT_solver, iter::Class_iterate, timer::class_advanced_timer)
    start_advanced_timer(timer, "HIP/form_system")
    H = get_lag_hess(iter); J = get_jac(iter)
    n = dim(iter); m = ncon(iter)
    if k.kind == 2     # kkt_system_rescale, clever_symmetric.jl:377-385
        mode = Dict(:none => 0, :u_only => 1, :u_and_x => 2)[k.pars.kkt.kkt_system_rescale]
        kkt_hip_check(k, "okkt_kkt_set_rescale", ccall((:okkt_kkt_set_rescale, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Cint, Float64, Float64), k.handle, mode, iter.point.mu, LinearAlgebra.norm(iter.point.x, Inf)))
    end
    if !k.pattern_set
        kkt_hip_check(k, "okkt_kkt_set_structure", ccall((:okkt_kkt_set_structure, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Cint),
            k.handle, n, m, H.colptr, H.rowval, J.colptr, J.rowval, 1))
        k.pattern_set = true
    end
    kkt_hip_check(k, "okkt_kkt_form_system", ccall((:okkt_kkt_form_system, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        k.handle, H.nzval, J.nzval, get_s(iter), get_y(iter)))
    k.schur_diag = zeros(n)
    ccall((:okkt_kkt_get_schur_diag, OKKT_LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), k.handle, k.schur_diag)
    k.factor_it = iter
    k.ready = :system_formed
    pause_advanced_timer(timer, "HIP/form_system")
end

function update_delta_vecs!(k::HIP_KKT_solver, delta_x_vec::Array{Float64,1}, delta_s_vec::Array{Float64,1}, timer::class_advanced_timer)
    k.delta_x_vec = delta_x_vec
    k.delta_s_vec = delta_s_vec
    sum(abs.(delta_s_vec)) > 0.0 && error("Not implemented")
    k.ready = :delta_updated
end

# factor!(kkt_solver, timer) -> factor_implementation!: a COMPLETE factorisation whatever the inertia flag, because the
# failed-step branch of one_phase.jl:231-242 computes a direction from it even when the flag is 0.  The delta loop, which
# throws failed attempts away, goes through ipopt_strategy! below (trial factorisations that may stop early).
function factor_implementation!(k::HIP_KKT_solver, timer::class_advanced_timer)
    inert = Ref(OkktInertia(0, 0, 0, 0))
    delta = length(k.delta_x_vec) > 0 ? k.delta_x_vec[1] : 0.0
    return Int(kkt_hip_check(k, "okkt_kkt_factor", ccall((:okkt_kkt_factor, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Float64, Ref{OkktInertia}), k.handle, delta, inert)))
end

# ipopt_strategy! (delta_strategy.jl:37-114) for this solver type: the whole loop behind one ccall; (status, num_fac, delta)
# are those of the reference's loop
function ipopt_strategy!(iter::Class_iterate, k::HIP_KKT_solver, pars::Class_parameters, timer::class_advanced_timer)
    num_fac = Ref{Int32}(0); delta = Ref{Float64}(0.0)
    p = OkktKktPars(pars.delta.start, pars.delta.min, pars.delta.max, pars.delta.inc, pars.delta.dec, pars.delta.zero, Int32(pars.kkt.ItRefine_Num), Int32(500))
    rc = kkt_hip_check(k, "okkt_kkt_ipopt_strategy", ccall((:okkt_kkt_ipopt_strategy, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Float64, Ref{OkktKktPars}, Ref{Int32}, Ref{Float64}), k.handle, get_delta(iter), p, num_fac, delta))
    k.delta_x_vec = delta[] * ones(dim(iter)); k.delta_s_vec = zeros(ncon(iter))
    k.ready = :factored
    # delta_strategy.jl:94-98: a failed attempt on a diagonally dominant x-block prints a warning; the library ran the scan on the
    # device after every failed attempt and counted
    nwarn = Ref{Int32}(0)
    if ccall((:okkt_kkt_diag_dom_warnings, OKKT_LIB), Cint, (Ptr{Cvoid}, Ref{Int32}), k.handle, nwarn) == 0
        for _ in 1:nwarn[]
            println("WARNING: Inertia calculation incorrect")
            @warn("Inertia calculation incorrect")
        end
    end
    # the caller reads old_delta = get_delta(iter) and then calls set_delta(iter, new_delta) itself (one_phase.jl:203-206)
    return rc == 1 ? :success : :failure, Int(num_fac[]), delta[]
end

# kkt_associate_rhs! (schur.jl:34-45, symmetric.jl: same body): System_rhs(iter, reduct_factors) evaluated on the device from
# the iterate's cached gradient / constraint values; the triple stays in HBM for compute_direction! and is also copied
# back, because the IPM reads kkt_solver.rhs (e.g. the merit function).  It also tells the library which iterate is
# current_it, which Schur_KKT_solver_direct reads (schur_direct.jl:35-37).
function kkt_associate_rhs!(k::HIP_KKT_solver, iter::Class_iterate, reduct_factors::Class_reduction_factors, timer::class_advanced_timer)
    start_advanced_timer(timer, "KKT/rhs")
    n = dim(iter); m = ncon(iter)
    rD = zeros(n); rP = zeros(m); rC = zeros(m)
    J = get_jac(iter)
    Jcur = iter === k.factor_it ? C_NULL : pointer(J.nzval)     # NULL: the Jacobian values of form_system! are current
    GC.@preserve J begin
        kkt_hip_check(k, "okkt_kkt_system_rhs", ccall((:okkt_kkt_system_rhs, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64,
             Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            k.handle, Jcur, get_grad(iter), get_cons(iter), get_s(iter), get_y(iter), get_mu(iter), iter.a_norm_penalty_par,
            reduct_factors.P, reduct_factors.D, reduct_factors.mu, rD, rP, rC))
    end
    k.rhs = System_rhs(rD, rP, rC)
    k.resident_rhs = k.rhs     # identity, not a flag: reference code that assigns k.rhs afterwards (compute_eigenvector!,
                               # kkt_system_solver.jl:205-228) must get ITS rhs solved, not the stale device copy
    k.dir.mu = -(1.0 - reduct_factors.mu) * get_mu(iter)
    k.dir.primal_scale = -(1.0 - reduct_factors.P) * iter.point.primal_scale
    k.reduct_factors = reduct_factors
    k.current_it = iter
    pause_advanced_timer(timer, "KKT/rhs")
end

function compute_direction_implementation!(k::HIP_KKT_solver, timer::class_advanced_timer)
    n = dim(k.factor_it); m = ncon(k.factor_it)
    dx = zeros(n); dy = zeros(m); ds = zeros(m)
    err = zeros(6)
    # resident rhs (k.rhs is still the object kkt_associate_rhs! created): three NULLs, nothing crosses PCIe on the way in;
    # a rhs the caller replaced on the host is uploaded
    rhs = k.rhs
    resident = rhs === k.resident_rhs
    GC.@preserve rhs begin
        r = resident ? (C_NULL, C_NULL, C_NULL) : (pointer(rhs.dual_r), pointer(rhs.primal_r), pointer(rhs.comp_r))
        kkt_hip_check(k, "okkt_kkt_compute_direction", ccall((:okkt_kkt_compute_direction, OKKT_LIB), Cint,
            (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
            k.handle, r[1], r[2], r[3], Int32(k.pars.kkt.ItRefine_Num), dx, dy, ds, err))
    end
    k.dir.x = dx; k.dir.y = dy; k.dir.s = ds
    check_for_nan(k.dir)
    k.kkt_err_norm = Class_kkt_error(err[1], err[2], err[3], err[4], err[5], err[6])
end


# Several directions of the same factor in one pass over L: the probe of the aggressive step (Reduct_affine, take_step.jl:2-3) and the
# candidates of take_step2! (take_step.jl:34-66) are right-hand sides of one factorised system.  Returns [(dir, kkt_err_norm)];
# kkt_associate_rhs!(k, iter, ...) must have told the library the current iterate.
function compute_directions!(k::HIP_KKT_solver, etas::Vector{Class_reduction_factors})
    # the same guards as compute_direction_implementation!: the BigFloat refinement of the Schur solvers ends in a MethodError in
    # the reference (schur.jl:167, eval.jl:232), and every returned direction goes through check_for_nan (IPM_tools.jl:32-49).
    # k.dir and k.kkt_err_norm are NOT updated: the caller picks one of the candidates.
    if k.pars.kkt.ItRefine_BigFloat && (k.kind == 0 || k.kind == 3)       # :schur, :schur_direct
        error("MethodError: no method matching hess_product(::Class_iterate, ::Array{BigFloat,1}) (schur.jl:167 with pars.kkt.ItRefine_BigFloat = true)")
    end
    n = dim(k.factor_it); m = ncon(k.factor_it); q = length(etas)
    e = zeros(3 * q)
    for i in 1:q
        e[3i - 2] = etas[i].P; e[3i - 1] = etas[i].D; e[3i] = etas[i].mu
    end
    dx = zeros(n * q); dy = zeros(m * q); ds = zeros(m * q); err = zeros(6 * q)
    kkt_hip_check(k, "okkt_kkt_compute_directions", ccall((:okkt_kkt_compute_directions, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        k.handle, Int32(q), e, Int32(k.pars.kkt.ItRefine_Num), dx, dy, ds, err))
    out = Vector{Tuple{Class_point, Class_kkt_error}}()
    for i in 1:q
        d = Class_point(); d.x = dx[(i - 1) * n + 1:i * n]; d.y = dy[(i - 1) * m + 1:i * m]; d.s = ds[(i - 1) * m + 1:i * m]
        d.mu = -(1.0 - etas[i].mu) * get_mu(k.current_it)
        d.primal_scale = -(1.0 - etas[i].P) * k.current_it.point.primal_scale
        check_for_nan(d)
        r = err[6 * (i - 1) + 1:6 * i]
        push!(out, (d, Class_kkt_error(r[1], r[2], r[3], r[4], r[5], r[6])))
    end
    return out
end
