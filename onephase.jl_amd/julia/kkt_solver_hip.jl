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
    kind::Cint                                   # 0 = :schur, 1 = :symmetric, 2 = :clever_symmetric
    pattern_set::Bool

    function HIP_KKT_solver(kind::Symbol)
        this = new()
        this.ready = :not_ready
        this.kind = kind == :schur ? 0 : (kind == :symmetric ? 1 : 2)
        this.pattern_set = false
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:okkt_kkt_create, OKKT_LIB), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}, Cint), h, C_NULL, this.kind)
        rc == 0 || error("okkt_kkt_create failed with code $rc")
        this.handle = h[]
        finalizer(s -> ccall((:okkt_kkt_destroy, OKKT_LIB), Cint, (Ptr{Cvoid},), s.handle), this)
        return this
    end
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

function factor_implementation!(k::HIP_KKT_solver, timer::class_advanced_timer)
    inert = Ref(OkktInertia(0, 0, 0, 0))
    delta = length(k.delta_x_vec) > 0 ? k.delta_x_vec[1] : 0.0
    return Int(kkt_hip_check(k, "okkt_kkt_factor", ccall((:okkt_kkt_factor, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Float64, Ref{OkktInertia}), k.handle, delta, inert)))
end

function compute_direction_implementation!(k::HIP_KKT_solver, timer::class_advanced_timer)
    n = dim(k.factor_it); m = ncon(k.factor_it)
    dx = zeros(n); dy = zeros(m); ds = zeros(m)
    err = zeros(6)
    kkt_hip_check(k, "okkt_kkt_compute_direction", ccall((:okkt_kkt_compute_direction, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        k.handle, k.rhs.dual_r, k.rhs.primal_r, k.rhs.comp_r, Int32(k.pars.kkt.ItRefine_Num), dx, dy, ds, err))
    k.dir.x = dx; k.dir.y = dy; k.dir.s = ds
    check_for_nan(k.dir)
    k.kkt_err_norm = Class_kkt_error(err[1], err[2], err[3], err[4], err[5], err[6])
end
