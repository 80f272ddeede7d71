# linear_solver_hip.jl -- Julia glue for libonephase_kkt.so (C ABI: include/okkt.h).
#
# UNTESTED IN THIS REPOSITORY: neither Julia nor the reference's dependencies are available in the
# build environment.  The file mirrors the reference's own plug-in pattern
# (src/linear_system_solvers/hsl.jl: a `mutable struct X <: abstract_linear_system_solver` in a file
# that `loadHSL`-style code `include`s at run time) and is what a maintainer of OnePhase.jl would
# drop into src/linear_system_solvers/ -- see INTEGRATION.md for the three edits around it.
#
# Every ccall below binds exactly one entry point of include/okkt.h; pointers are to Julia-owned
# arrays that are GC.@preserve'd for the duration of the (blocking) call.

const OKKT_LIB = get(ENV, "ONEPHASE_KKT_LIB", "libonephase_kkt")

struct OkktInertia
    pos::Int64
    neg::Int64
    zero::Int64
    nonfinite::Int64
end

# okkt_opts of include/okkt.h, field for field (an isbits struct has C layout: six Int32, four Float64, four Int32 = 72 bytes)
struct OkktOpts
    device::Int32
    host_symbolic_only::Int32
    ordering::Int32
    relax_always::Int32
    relax_small::Int32
    relax_mid::Int32
    relax_small_frac::Float64
    relax_mid_frac::Float64
    relax_any_frac::Float64
    inertia_tol::Float64
    small_front_max::Int32
    panel_nb::Int32
    early_exit::Int32
    reserved::Int32
end

function okkt_default_opts()
    o = Ref(OkktOpts(0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0))
    ccall((:okkt_default_opts, OKKT_LIB), Cint, (Ref{OkktOpts},), o)
    return o[]
end

# The back-end's own knobs ride in pars.kkt like the reference's `ma97_u` (parameters.jl:13,25 -> linear_solver_HSL(..., pars.kkt.ma97_u),
# kkt_system_solver.jl:247): the fields hip_device, hip_ordering, hip_relax_always / _small / _mid, hip_relax_small_frac / _mid_frac /
# _any_frac, hip_inertia_tol of Class_kkt_solver_options (INTEGRATION.md, edit 4), set through the existing plumbing, e.g.
# "kkt!hip_ordering" => 3 (create_pars_JuMP, JuMPinterface.jl:570-586).  -1 / 0.0 keep the library's default.
function okkt_opts_from_pars(kkt)
    d = okkt_default_opts()
    pick(v, dflt) = v > 0 ? v : dflt
    return OkktOpts(kkt.hip_device >= 0 ? Int32(kkt.hip_device) : d.device, d.host_symbolic_only,
                    kkt.hip_ordering != 0 ? Int32(kkt.hip_ordering) : d.ordering,
                    Int32(pick(kkt.hip_relax_always, d.relax_always)), Int32(pick(kkt.hip_relax_small, d.relax_small)), Int32(pick(kkt.hip_relax_mid, d.relax_mid)),
                    pick(kkt.hip_relax_small_frac, d.relax_small_frac), pick(kkt.hip_relax_mid_frac, d.relax_mid_frac), pick(kkt.hip_relax_any_frac, d.relax_any_frac),
                    kkt.hip_inertia_tol > 0 ? kkt.hip_inertia_tol : d.inertia_tol,
                    d.small_front_max, d.panel_nb, d.early_exit, d.reserved)
end

mutable struct linear_solver_HIP <: abstract_linear_system_solver
    handle::Ptr{Cvoid}
    sym::Symbol            # :definite (Cholesky semantics) or :symmetric (LDL', inertia from sign(D))
    safe_mode::Bool
    recycle::Bool
    inertia::OkktInertia
    opts::Union{Nothing,OkktOpts}      # nothing: okkt_create(NULL) = the library's defaults

    function linear_solver_HIP(sym::Symbol, safe_mode::Bool, recycle::Bool, opts::Union{Nothing,OkktOpts}=nothing)
        this = new()
        this.handle = C_NULL
        this.sym = sym
        this.safe_mode = safe_mode
        this.recycle = recycle
        this.inertia = OkktInertia(0, 0, 0, 0)
        this.opts = opts
        return this
    end
end

function okkt_error(solver::linear_solver_HIP, what::String, rc)
    msg = solver.handle == C_NULL ? "" : unsafe_string(ccall((:okkt_last_error, OKKT_LIB), Cstring, (Ptr{Cvoid},), solver.handle))
    error("$what failed with code $rc: $msg")
end

function initialize!(solver::linear_solver_HIP)
    if solver.handle == C_NULL
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = solver.opts === nothing ?
             ccall((:okkt_create, OKKT_LIB), Cint, (Ref{Ptr{Cvoid}}, Ptr{Cvoid}), h, C_NULL) :             # NULL = default options
             ccall((:okkt_create, OKKT_LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{OkktOpts}), h, Ref(solver.opts))  # pars.kkt.hip_* (okkt_opts_from_pars)
        rc == 0 || error("okkt_create failed with code $rc (no HIP device? the KKT path has no CPU fallback)")
        solver.handle = h[]
        # Early exit stays OFF at this level: ls_factor! cannot know whether its caller will solve with a factorisation
        # whose flag is 0 -- the refactorisation after a failed step does (one_phase.jl:231-242 -> take_step2!).  A caller
        # that discards failed factors (a delta loop of its own) may switch it on around those calls:
        #   ccall((:okkt_set_early_exit, OKKT_LIB), Cint, (Ptr{Cvoid}, Cint), solver.handle, 1)
        finalizer(finalize!, solver)
    end
end

function finalize!(solver::linear_solver_HIP)
    if solver.handle != C_NULL
        ccall((:okkt_destroy, OKKT_LIB), Cint, (Ptr{Cvoid},), solver.handle)
        solver.handle = C_NULL
    end
end

function ls_factor!(solver::linear_solver_HIP, SparseMatrix::SparseMatrixCSC{Float64,Int64}, n::Int64, m::Int64, timer::class_advanced_timer)
    start_advanced_timer(timer, "HIP/factorize")
    A = SparseMatrix
    dim = size(A, 1)
    rc = 0
    inert = Ref(OkktInertia(0, 0, 0, 0))
    GC.@preserve A begin
        # pattern analysis is cached by pattern hash inside the library: free when only values changed
        rc = ccall((:okkt_analyze, OKKT_LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Cint),
                   solver.handle, dim, A.colptr, A.rowval, 1)
        rc == 0 || okkt_error(solver, "okkt_analyze", rc)
        kind = solver.sym == :definite ? 0 : (solver.sym == :symmetric ? 1 : error("this.options.sym = " * string(solver.sym) * " not supported"))
        rc = ccall((:okkt_factor, OKKT_LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int64, Int64, Cint, Ref{OkktInertia}),
                   solver.handle, A.nzval, n, m, kind, inert)
    end
    pause_advanced_timer(timer, "HIP/factorize")
    rc < 0 && okkt_error(solver, "okkt_factor", rc)
    solver.inertia = inert[]
    return Int(rc)       # 1: inertia correct, 0: not (same contract as linear_solver_JULIA / linear_solver_HSL)
end

function ls_solve!(solver::linear_solver_HIP, my_rhs::Array{Float64,1}, my_sol::Array{Float64,1}, timer::class_advanced_timer)
    start_advanced_timer(timer, "HIP/ls_solve")
    GC.@preserve my_rhs my_sol begin
        rc = ccall((:okkt_solve, OKKT_LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64),
                   solver.handle, my_rhs, my_sol, 1)
        rc == 0 || okkt_error(solver, "okkt_solve", rc)
    end
    pause_advanced_timer(timer, "HIP/ls_solve")
end

function ls_solve(solver::linear_solver_HIP, my_rhs::AbstractArray, timer::class_advanced_timer)
    rhs = Vector{Float64}(my_rhs)      # SparseVector rhs is densified, as in julia.jl:105-113
    sol = zeros(length(rhs))
    ls_solve!(solver, rhs, sol, timer)
    return sol
end
