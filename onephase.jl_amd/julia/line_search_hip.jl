# Julia glue (description of the reference-side binding; cannot be executed in this image -- no julia): the step-side
# functions of simple_ls on the device, for iterates whose KKT solver is a HIP_KKT_solver (kkt_solver_hip.jl).
#
# A maintainer would include this file after src/line_search/line_search.jl and dispatch on the solver type where
# simple_ls (line_search.jl:36-199) calls the host versions:
#   line_search.jl:40-41   lb_s_predict + simple_max_step            -> hip_max_step_primal
#   move.jl:15-17          all(new_it.point.s .>= lb_s(it,dir,pars)) -> hip_s_bound_ok
#   line_search.jl:84-86   dual_bounds + lb_y + simple_max_step      -> hip_dual_step_range
#   stable_ls.jl:18, filter_ls.jl:28, kkt_ls.jl:16  merit_function_predicted_reduction(iter, dir, 1.0)
#                                                                    -> hip_merit_function_predicted_reduction
#   move.jl:100-112        the dual_ls == 1 / 3 least-squares step   -> hip_dual_step
# `iter` must be kkt.factor_it and `dir` kkt.dir (the direction the device still holds); after scale_direction
# (line_search.jl:10-19) or a correction, hip_set_direction makes the new direction resident.

function hip_set_direction(k::HIP_KKT_solver, dir::Class_point)
    kkt_hip_check(k, "okkt_kkt_set_direction", ccall((:okkt_kkt_set_direction, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), k.handle, dir.x, dir.y, dir.s))
    k.dir = dir
end

function hip_max_step_primal(k::HIP_KKT_solver, iter::Class_iterate, pars::Class_parameters)
    step = Ref(0.0); nx = Ref(0.0)
    kkt_hip_check(k, "okkt_kkt_max_step_primal", ccall((:okkt_kkt_max_step_primal, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Float64, Ref{Float64}, Ref{Float64}),
        k.handle, iter.frac_bd_predict, pars.ls.fraction_to_boundary_predict_exp, step, nx))
    return step[]
end

function hip_s_bound_ok(k::HIP_KKT_solver, iter::Class_iterate, new_s::Array{Float64,1}, pars::Class_parameters)
    ok = Ref(Int32(0))
    kkt_hip_check(k, "okkt_kkt_s_bound_ok", ccall((:okkt_kkt_s_bound_ok, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Int32}),
        k.handle, new_s, iter.frac_bd, pars.ls.fraction_to_boundary_predict_exp, ok))
    return ok[] == 1
end

function hip_dual_step_range(k::HIP_KKT_solver, iter::Class_iterate, candidate::Class_iterate, pars::Class_parameters)
    lb = Ref(0.0); ub = Ref(0.0)
    kkt_hip_check(k, "okkt_kkt_dual_step_range", ccall((:okkt_kkt_dual_step_range, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Ptr{Float64}, Ref{Float64}, Ref{Float64}),
        k.handle, candidate.point.s, candidate.point.y, candidate.point.mu, pars.ls.comp_feas, iter.frac_bd, lb, ub))
    return lb[], ub[]
end

function hip_merit_function_predicted_reduction(k::HIP_KKT_solver, iter::Class_iterate, dir::Class_point, step_size::Float64)
    out = zeros(4)   # phi reduction, norm(comp(iter), Inf), norm(comp_predicted, Inf), merit reduction
    kkt_hip_check(k, "okkt_kkt_predicted_reduction", ccall((:okkt_kkt_predicted_reduction, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Float64, Float64, Float64, Float64, Ptr{Float64}),
        k.handle, get_grad(iter), iter.point.mu, dir.mu, iter.a_norm_penalty_par, step_size, out))
    return out[4]
end

function hip_dual_step(k::HIP_KKT_solver, new_it::Class_iterate, step_size_P::Float64, lb::Float64, ub::Float64, pars::Class_parameters)
    scale = dual_scale(new_it, pars)     # move.jl:100-101: scale_D = scale_mu = dual_scale(new_it, pars)
    out = Ref(0.0)
    kkt_hip_check(k, "okkt_kkt_dual_step", ccall((:okkt_kkt_dual_step, OKKT_LIB), Cint,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Cint,
         Float64, Float64, Ref{Float64}),
        k.handle, get_jac(new_it).nzval, get_grad(new_it), new_it.point.s, new_it.point.y, new_it.point.mu,
        new_it.a_norm_penalty_par, step_size_P, lb, ub, Cint(pars.ls.dual_ls), scale, scale, out))
    return out[]
end
