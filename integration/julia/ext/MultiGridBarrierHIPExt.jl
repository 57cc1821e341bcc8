# MultiGridBarrierHIPExt.jl -- package extension binding libmgbhip.so (include/mgbhip.h) under
# MultiGridBarrier.jl's `Device` dispatch, written against the reference's own extension point:
#
#   mgb_solve(prob; device)                         src/mgb.jl:798-842
#     prob = native_to_device(device, prob)         src/mgb.jl:805, src/device.jl:40
#     mgb_driver(prob.M, prob.f, prob.g, prob.Q; …) src/mgb.jl:831   <- dispatches on typeof(prob.M)
#     device_to_native(device, sol)                 src/mgb.jl:841
#
# `native_to_device(HIPDevice, prob)` returns an `MGBProblem` whose `M` is a `HIPImage` (the two opaque
# handles of the (main, feasibility) pair plus the host metadata `mgb_driver` needs); the `mgb_driver`
# method for `HIPImage` below restates the reference's orchestration (feasibility probe, phase I with box
# escalation, `_matched_t` hand-off, main ramp; src/mgb.jl:332-584) around `mgbhip_mgb_core`.  It is the
# Julia twin of multigridbarrier.jl_amd/solve.py, which is the version this repository can execute and test
# (Julia is not installed in the build container or on the GPU box).
#
# One line is needed in the package itself (next to `struct CUDADevice <: Device end`, src/device.jl:32):
#     struct HIPDevice <: Device end;  export HIPDevice
# and in Project.toml:  [extensions] MultiGridBarrierHIPExt = "Libdl"  (any always-present trigger).
module MultiGridBarrierHIPExt

using MultiGridBarrier, SparseArrays, LinearAlgebra
import MultiGridBarrier: Device, HIPDevice, native_to_device, device_to_native, mgb_cleanup, mgb_driver,
                         MGBProblem, MGBSOL, AMG, Convex, BlockDiag, BlockColumn, MGBConvergenceFailure,
                         default_device!, newton, solve, symmetric, mgb_all_isfinite, divide_and_conquer, NoFinalize,
                         stopping_exact, stopping_inexact, linesearch_backtracking

const libmgbhip = get(ENV, "MGBHIP_LIB", "libmgbhip.so")

# ---- mirrors of the C structs (include/mgbhip.h); field order and types are the header's -------------
const MAX_PIECES, MAX_IDX, MAX_ND, MAX_NU, MAX_OPS = 4, 4, 10, 4, 8
const KIND_EP, KIND_LINEAR = Int32(1), Int32(2)          # MGBHIP_KIND_EP / MGBHIP_KIND_LINEAR

struct CPiece
    kind::Int32; ni::Int32; nc::Int32; idx::NTuple{MAX_IDX,Int32}
    A::Ptr{Float64}; b::Ptr{Float64}; p::Ptr{Float64}; mu::Ptr{Float64}
    p_const::Float64; mu_const::Float64; select::Ptr{Float64}
end
const NOPIECE = CPiece(0, 0, 0, (0, 0, 0, 0), C_NULL, C_NULL, C_NULL, C_NULL, 0.0, 0.0, C_NULL)
struct CCone; npieces::Int32; pieces::NTuple{MAX_PIECES,CPiece}; feasibility::Int32; NC::Int32; end
struct CCSR;  rows::Int64; cols::Int64; rowptr::Ptr{Int32}; colidx::Ptr{Int32}; values::Ptr{Float64}; end
struct CDesc
    p::Int32; N::Int64; nu::Int32; nD::Int32; n_ops::Int32
    ops::NTuple{MAX_OPS,Ptr{Float64}}; D_state::NTuple{MAX_ND,Int32}; D_op::NTuple{MAX_ND,Int32}
    w::Ptr{Float64}; L::Int32; R::Ptr{CCSR}; cone::CCone; barrier_weights::Ptr{Float64}
    x::Ptr{Float64}; dim::Int32
end
mutable struct COptions
    tol::Float64; t::Float64; kappa::Float64; maxit::Int32; max_newton::Int32
    ls_beta::Float64; ls_c1::Float64; line_search::Int32; stop_lambda_tol::Float64; stop_theta::Float64
    finalize::Int32; finalize_theta::Float64; early_stop::Int32
    stopping_criterion::Ptr{Cvoid}; early_stop_fn::Ptr{Cvoid}; user::Ptr{Cvoid}
    COptions() = new()
end
mutable struct CCoreResult
    k::Int32; L::Int32; failure_code::Int32; t_final::Float64; t_elapsed::Float64; solve_seconds::Float64
    newton_iterations::Int64; f0_evals::Int64; f1_evals::Int64; f2_evals::Int64; factorizations::Int64
    cap_steps::Int32; its::Ptr{Int64}; ts::Ptr{Float64}; kappas::Ptr{Float64}; times::Ptr{Float64}; c_dot_Dz::Ptr{Float64}
    CCoreResult() = new()
end

lasterr() = unsafe_string(@ccall libmgbhip.mgbhip_last_error()::Cstring)
check(rc) = rc == 0 || error("libmgbhip status $rc: $(lasterr())")

# ---- the device image --------------------------------------------------------------------------------
mutable struct HIPImage{T}
    ctx::Ptr{Cvoid}; main::Ptr{Cvoid}; feas::Ptr{Cvoid}       # feas is created on first use, sharing operators
    M::Tuple                                                    # the CPU (main, feasibility) AMG pair
    Q::Convex{T}
    keep::Vector{Any}                                           # arrays the descriptors pointed into (upload is synchronous)
end

function csr0(R::SparseMatrixCSC{Float64,Int})                 # CSC of R' == CSR of R; 0-based Int32
    Rt = SparseMatrixCSC{Float64,Int}(sparse(R'))
    (Int32.(Rt.colptr .- 1), Int32.(Rt.rowval .- 1), copy(Rt.nzval))
end

# Convex -> pieces.  EP: functor struct with .idx (src/convex_euclidian_power.jl:71-80), args (A, b, p, mu)
# (:447-453); linear: closures capturing idx (src/convex_linear.jl:119-223), args (A, b); piecewise: a
# PiecewiseBarrierF0 holding the pieces' f0 and the Val ranges into args, args[1] = select grid
# (src/convex_piecewise.jl:15-29, :143-166).
resolve_idx(idx, n) = idx isa Colon ? collect(1:n) : collect(Int, idx)
function piece_of(f0::MultiGridBarrier.EuclidianPowerBarrier{NZ}, args) where {NZ}
    (kind = KIND_EP, idx = resolve_idx(f0.idx, NZ), nc = NZ, A = args[1], b = args[2], p = args[3], mu = args[4])
end
function piece_of(f0, args)                                     # convex_linear closure: `idx` is a captured field
    A, b = args[1], args[2]
    nc = size(b, 2); ni = size(A, 2) ÷ nc
    (kind = KIND_LINEAR, idx = resolve_idx(getfield(f0, :idx), ni), nc = nc, A = A, b = b, p = nothing, mu = nothing)
end
function pieces_of(Q::Convex)
    f0 = Q.barrier[1]
    if f0 isa MultiGridBarrier.PiecewiseBarrierF0
        valof(::Val{k}) where {k} = k
        rng(k) = valof(f0.arg_ranges_val[k][1]):valof(f0.arg_ranges_val[k][2])
        return Q.args[1], [piece_of(f0.barrier_f0s[k], Q.args[rng(k)]) for k in eachindex(f0.barrier_f0s)]
    end
    return nothing, [piece_of(f0, Q.args)]
end

is_identity_block(d::Array{Float64,3}) = all(e -> view(d, :, :, e) == I, axes(d, 3))

function build_desc(M::AMG, Q::Convex; feasibility::Bool, NC::Int, keep::Vector{Any})
    col(a) = (c = Matrix{Float64}(a); push!(keep, c); pointer(c))          # n x K column-major == the reference's Q.args layout
    vec0(a) = (c = Vector{Float64}(a); push!(keep, c); pointer(c))
    blk = M.D_fine[1].active_block
    p, N = size(blk.data, 1), size(blk.data, 3)
    ops = Ptr{Float64}[]; opkey = Any[]; D_state = Int32[]; D_op = Int32[]
    for Dk in M.D_fine                                                     # BlockColumn: (state, operator) rows of D
        i = findfirst(o -> o === Dk.active_block, opkey)
        if i === nothing
            push!(opkey, Dk.active_block)
            d = Dk.active_block.data
            push!(ops, is_identity_block(d) ? Ptr{Float64}(C_NULL) : (push!(keep, d); pointer(d)))
            i = length(opkey)
        end
        push!(D_state, Int32(Dk.active_col - 1)); push!(D_op, Int32(i - 1))
    end
    csr = map(M.R_fine) do R
        rp, ci, vv = csr0(SparseMatrixCSC{Float64,Int}(R)); push!(keep, rp, ci, vv)
        CCSR(size(R, 1), size(R, 2), pointer(rp), pointer(ci), pointer(vv))
    end
    push!(keep, csr)
    select, pcs = pieces_of(Q)
    length(pcs) <= MAX_PIECES || error("HIPDevice: more than $MAX_PIECES convex pieces")
    cp = ntuple(MAX_PIECES) do k
        k > length(pcs) && return NOPIECE
        q = pcs[k]
        length(q.idx) <= MAX_IDX && q.nc <= MAX_IDX || error("HIPDevice: functor family size exceeds MGBHIP_MAX_IDX")
        idx = ntuple(j -> j <= length(q.idx) ? Int32(q.idx[j] - 1) : Int32(0), MAX_IDX)
        constp = q.p !== nothing && allequal(q.p) && allequal(q.mu)
        CPiece(q.kind, length(q.idx), q.nc, idx, col(q.A), col(q.b),
               (q.p === nothing || constp) ? C_NULL : vec0(q.p), (q.p === nothing || constp) ? C_NULL : vec0(q.mu),
               constp ? Float64(first(q.p)) : 0.0, constp ? Float64(first(q.mu)) : 0.0,
               select === nothing ? C_NULL : vec0(select[:, k]))
    end
    pad(v, z, n) = ntuple(j -> j <= length(v) ? v[j] : z, n)
    x = Matrix{Float64}(M.x); push!(keep, x)
    CDesc(p, N, M.D_fine[1].nu, length(M.D_fine), length(ops), pad(ops, Ptr{Float64}(C_NULL), MAX_OPS),
          pad(D_state, Int32(0), MAX_ND), pad(D_op, Int32(0), MAX_ND), vec0(M.w), length(csr), pointer(csr),
          CCone(length(pcs), cp, feasibility, NC), C_NULL, pointer(x), size(x, 2))
end

function create_problem(ctx, M, Q; feasibility=false, NC=0, share=C_NULL)
    keep = Any[]
    desc = build_desc(M, Q; feasibility, NC, keep)
    h = Ref{Ptr{Cvoid}}()
    GC.@preserve keep check(@ccall libmgbhip.mgbhip_problem_create(ctx::Ptr{Cvoid}, Ref(desc)::Ptr{CDesc}, share::Ptr{Cvoid},
                                                                  h::Ptr{Ptr{Cvoid}})::Cint)
    h[]                                                        # every array was copied to the device before the call returned
end

function native_to_device(::Type{HIPDevice}, prob::MGBProblem{T}) where {T}
    T === Float64 || error("HIPDevice computes in Float64")
    ctx = Ref{Ptr{Cvoid}}()
    check(@ccall libmgbhip.mgbhip_create(ctx::Ptr{Ptr{Cvoid}}, 0::Cint, C_NULL::Ptr{Cvoid})::Cint)
    img = HIPImage{T}(ctx[], create_problem(ctx[], prob.M[1], prob.Q), C_NULL, prob.M, prob.Q, Any[])
    finalizer(mgb_cleanup, img)                                  # safety net only: mgb_driver releases the image itself
    MGBProblem{T}(img, prob.f, prob.g, prob.Q, prob.geometry)   # f, g stay host arrays: mgb_core takes host pointers
end
device_to_native(::Type{HIPDevice}, sol::MGBSOL) = sol          # solutions come back as host Arrays already
function mgb_cleanup(img::HIPImage)                               # plans + factorizations die with the handles
    img.feas != C_NULL && (@ccall libmgbhip.mgbhip_problem_destroy(img.feas::Ptr{Cvoid})::Cint; img.feas = C_NULL)
    img.main != C_NULL && (@ccall libmgbhip.mgbhip_problem_destroy(img.main::Ptr{Cvoid})::Cint; img.main = C_NULL)
    img.ctx != C_NULL && (@ccall libmgbhip.mgbhip_destroy(img.ctx::Ptr{Cvoid})::Cint; img.ctx = C_NULL)
    nothing
end
mgb_cleanup(::Type{HIPDevice}) = nothing                          # no process-global caches (src/device.jl:92)
# No `mgb_cleanup(::MGBSOL)` method here: the solution holds host arrays only, and the device image -- plans,
# factorizations, ~3 GB of HBM at L = 9 -- is released deterministically at the end of `mgb_driver` below, on the
# success path and on the throw path alike (the reference flushes its caches at the same two points, src/mgb.jl:837-840).

feasibility_handle!(img::HIPImage) = img.feas != C_NULL ? img.feas :
    (img.feas = create_problem(img.ctx, img.M[2], img.Q; feasibility=true, NC=length(img.M[1].D_fine) + 1, share=img.main))

# ---- mgb_core through the C ABI -----------------------------------------------------------------------
# User callables cross the C ABI as plain C function pointers + a `user` pointer to this state object (no closure
# trampolines: works on every platform).  An exception raised by the callable is kept, the library is told to stop
# (converged / early stop = true), and hip_mgb_core rethrows it -- in the reference it would propagate out of newton.
mutable struct CallbackState
    stop::Any            # stopping_criterion(ymin, ynext, gmin, gnext, n, ndecmin, ndec) or nothing
    early::Any           # early_stop(z) / early_stop(z, t) or nothing
    zlen::Int
    err::Any
end
function stop_thunk(ymin::Cdouble, ynext::Cdouble, gmin::Cdouble, gn::Cdouble, ndecmin::Cdouble, ndec::Cdouble,
                    user::Ptr{Cvoid})::Cint
    st = unsafe_pointer_to_objref(user)::CallbackState
    st.err === nothing || return Cint(1)
    try
        # gnext arrives as its norm (the vector stays on the device): a one-element vector keeps `norm(gnext)` right
        return Cint(st.stop(ymin, ynext, gmin, [gn], nothing, ndecmin, ndec) ? 1 : 0)
    catch e
        st.err = e
        return Cint(1)
    end
end
function early_thunk(z::Ptr{Cdouble}, t::Cdouble, user::Ptr{Cvoid})::Cint
    st = unsafe_pointer_to_objref(user)::CallbackState
    st.err === nothing || return Cint(1)
    st.early === nothing && return Cint(0)
    try
        zz = copy(unsafe_wrap(Array, z, st.zlen))
        return Cint((applicable(st.early, zz, t) ? st.early(zz, t) : st.early(zz)) ? 1 : 0)
    catch e
        st.err = e
        return Cint(1)
    end
end

function options(h, n; tol, t, kappa, maxit, max_newton, early_stop::Int, finalize::Bool, state::Union{Nothing,CallbackState}=nothing)
    o = COptions()
    @ccall libmgbhip.mgbhip_default_options(o::Ref{COptions}, n::Int64)::Cvoid
    tol === nothing || (o.tol = tol); o.t = t
    kappa === nothing || (o.kappa = kappa); maxit === nothing || (o.maxit = maxit)
    max_newton === nothing || (o.max_newton = max_newton)
    o.finalize = finalize ? 1 : 0; o.early_stop = early_stop
    if state !== nothing
        o.user = pointer_from_objref(state)
        state.stop === nothing || (o.stopping_criterion =
            @cfunction(stop_thunk, Cint, (Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Cdouble, Ptr{Cvoid})))
        # installed whenever a callable is present: a raising stopping rule must be able to end the t-ramp too
        o.early_stop_fn = @cfunction(early_thunk, Cint, (Ptr{Cdouble}, Cdouble, Ptr{Cvoid}))
    end
    o
end

function hip_mgb_core(h::Ptr{Cvoid}, L::Int, z::Vector{Float64}, c::Matrix{Float64}, o::COptions;
                      state::Union{Nothing,CallbackState}=nothing)
    cap = 4096
    its = zeros(Int64, L, cap); ts = zeros(cap); kap = zeros(cap); tms = zeros(cap); cdz = zeros(cap)
    r = CCoreResult(); r.cap_steps = cap
    GC.@preserve its ts kap tms cdz z c state begin
        r.its = pointer(its); r.ts = pointer(ts); r.kappas = pointer(kap); r.times = pointer(tms); r.c_dot_Dz = pointer(cdz)
        rc = @ccall libmgbhip.mgbhip_mgb_core(h::Ptr{Cvoid}, z::Ptr{Float64}, c::Ptr{Float64}, o::Ref{COptions},
                                              r::Ref{CCoreResult})::Cint
    end
    state !== nothing && state.err !== nothing && throw(state.err)  # a user callable raised: propagate, like the reference
    if rc == 5                                                       # MGBHIP_ERR_CONVERGENCE
        code = r.failure_code == 2 ? :iteration_limit : :stall
        msg = (r.k == 1 && code === :stall && r.t_final == o.t) ?
            "Initial centering failed in mgb_solve at t=$(o.t), tol=$(o.tol), maxit=$(o.maxit)." :
            "Convergence failure in mgb_solve at t=$(r.t_final), k=$(r.k), tol=$(o.tol), maxit=$(o.maxit)."
        throw(MGBConvergenceFailure(msg, code))
    end
    check(rc)
    k = Int(r.k)
    (; z, its = its[:, 1:k], ts = ts[1:k], kappas = kap[1:k], times = tms[1:k], c_dot_Dz = cdz[1:k],
       t_begin = 0.0, t_end = r.t_elapsed, t_elapsed = r.t_elapsed, c)
end

node_barrier(h, z, n, nD) = (F = zeros(n); Dz = zeros(n, nD);
    check(@ccall libmgbhip.mgbhip_node_barrier(h::Ptr{Cvoid}, z::Ptr{Float64}, F::Ptr{Float64}, Dz::Ptr{Float64})::Cint); (F, Dz))
node_slack(h, z, n) = (s = zeros(n); check(@ccall libmgbhip.mgbhip_node_slack(h::Ptr{Cvoid}, z::Ptr{Float64}, s::Ptr{Float64})::Cint); s)

# ---- device vectors + the reference's generic `newton` on them (custom line_search / finalize closures) -----------
# `newton` (src/newton.jl:227-287) needs from its vector type: +, -, scalar *, dot, norm, mgb_all_isfinite; from its
# matrix type: symmetric(H) and solve(H, g).  HIPVec wraps an mgbhip_vec, HIPHessian is a token for "the H that
# mgbhip_f2_d left assembled at this level".  User line searches receive HIPVecs and the closures F0 / F1.
mutable struct HIPVec
    h::Ptr{Cvoid}
    ctx::Ptr{Cvoid}
    n::Int
    function HIPVec(ctx::Ptr{Cvoid}, n::Integer)
        r = Ref{Ptr{Cvoid}}()
        check(@ccall libmgbhip.mgbhip_vec_alloc(ctx::Ptr{Cvoid}, Int64(n)::Int64, r::Ptr{Ptr{Cvoid}})::Cint)
        v = new(r[], ctx, Int(n))
        finalizer(free!, v)
        v
    end
end
free!(v::HIPVec) = (v.h != C_NULL && (@ccall libmgbhip.mgbhip_vec_free(v.h::Ptr{Cvoid})::Cint; v.h = C_NULL); nothing)
function HIPVec(ctx::Ptr{Cvoid}, a::AbstractVector{Float64})
    v = HIPVec(ctx, length(a)); b = Vector{Float64}(a)
    GC.@preserve b check(@ccall libmgbhip.mgbhip_vec_upload(v.h::Ptr{Cvoid}, b::Ptr{Float64}, Int64(length(b))::Int64)::Cint)
    v
end
Base.length(v::HIPVec) = v.n
Base.Array(v::HIPVec) = (a = zeros(v.n); check(@ccall libmgbhip.mgbhip_vec_download(v.h::Ptr{Cvoid}, a::Ptr{Float64}, Int64(v.n)::Int64)::Cint); a)
Base.copy(v::HIPVec) = (w = HIPVec(v.ctx, v.n); check(@ccall libmgbhip.mgbhip_vec_copy(w.h::Ptr{Cvoid}, v.h::Ptr{Cvoid})::Cint); w)
axpy!(a::Real, x::HIPVec, y::HIPVec) = (check(@ccall libmgbhip.mgbhip_vec_axpy(Float64(a)::Float64, x.h::Ptr{Cvoid}, y.h::Ptr{Cvoid})::Cint); y)
Base.:+(a::HIPVec, b::HIPVec) = axpy!(1.0, b, copy(a))
Base.:-(a::HIPVec, b::HIPVec) = axpy!(-1.0, b, copy(a))
Base.:*(s::Real, v::HIPVec) = (w = copy(v); check(@ccall libmgbhip.mgbhip_vec_scale(Float64(s)::Float64, w.h::Ptr{Cvoid})::Cint); w)
Base.:*(v::HIPVec, s::Real) = s * v
Base.:-(v::HIPVec) = -1.0 * v
LinearAlgebra.dot(a::HIPVec, b::HIPVec) = (r = Ref{Float64}(); check(@ccall libmgbhip.mgbhip_vec_dot(a.h::Ptr{Cvoid}, b.h::Ptr{Cvoid}, r::Ptr{Float64})::Cint); r[])
LinearAlgebra.norm(a::HIPVec) = (r = Ref{Float64}(); check(@ccall libmgbhip.mgbhip_vec_norm(a.h::Ptr{Cvoid}, r::Ptr{Float64})::Cint); r[])
mgb_all_isfinite(a::HIPVec) = (r = Ref{Int32}(); check(@ccall libmgbhip.mgbhip_vec_isfinite(a.h::Ptr{Cvoid}, r::Ptr{Int32})::Cint); r[] != 0)

struct HIPHessian
    prob::Ptr{Cvoid}
    level::Int32
    ctx::Ptr{Cvoid}
end
symmetric(H::HIPHessian) = H
function solve(H::HIPHessian, g::HIPVec)
    x = HIPVec(H.ctx, g.n)
    check(@ccall libmgbhip.mgbhip_solve_d(H.prob::Ptr{Cvoid}, H.level::Int32, g.h::Ptr{Cvoid}, x.h::Ptr{Cvoid})::Cint)
    x
end

# mgb_step + mgb_core (src/mgb.jl:16-183) on device vectors, for the keyword arguments the resident ramp cannot take.
function generic_mgb_core(img::HIPImage{T}, h::Ptr{Cvoid}, z0::Vector{T}, c0::Matrix{T}; tol, t, kappa, maxit, max_newton,
                          finalize, stopping_criterion, early_stop, line_search, printlog) where {T}
    M1 = img.M[1]
    n, L = length(M1.w), length(M1.R_fine)
    tol = something(tol, sqrt(eps(T))); kappa = T(something(kappa, 10)); maxit = something(maxit, 10000)
    max_newton = something(max_newton, Int(ceil(log2(-log2(eps(T)))) + 2))
    sc = something(stopping_criterion, stopping_inexact(T(0.25) / sqrt(T(n)), T(0.9)))
    ls = something(line_search, linesearch_backtracking(T))
    fin = finalize === true ? stopping_exact(T(0.9)) : (finalize === false ? NoFinalize() : finalize)
    stop_early = something(early_stop, z -> false)
    estop(z, tt) = (zz = Array(z); applicable(stop_early, zz, tt) ? stop_early(zz, tt) : stop_early(zz))
    z = HIPVec(img.ctx, z0)
    sizes = [Int(@ccall libmgbhip.mgbhip_level_size(h::Ptr{Cvoid}, Int32(J - 1)::Int32)::Int64) for J in 1:L]
    function step(cvec::HIPVec, finalize_now, initial_step)
        its = zeros(Int, L)
        function eta(j, J, crit, mi)
            zJ = copy(z)                                                 # the closures capture a snapshot (src/mgb.jl:48)
            lev = Int32(J - 1)
            F0(s) = (r = Ref{Float64}(); check(@ccall libmgbhip.mgbhip_f0_d(h::Ptr{Cvoid}, lev::Int32, s.h::Ptr{Cvoid},
                     cvec.h::Ptr{Cvoid}, zJ.h::Ptr{Cvoid}, r::Ptr{Float64})::Cint); r[])
            F1(s) = (gv = HIPVec(img.ctx, sizes[J]); check(@ccall libmgbhip.mgbhip_f1_d(h::Ptr{Cvoid}, lev::Int32, s.h::Ptr{Cvoid},
                     cvec.h::Ptr{Cvoid}, zJ.h::Ptr{Cvoid}, gv.h::Ptr{Cvoid})::Cint); gv)
            F2(s) = (check(@ccall libmgbhip.mgbhip_f2_d(h::Ptr{Cvoid}, lev::Int32, s.h::Ptr{Cvoid}, cvec.h::Ptr{Cvoid},
                     zJ.h::Ptr{Cvoid})::Cint); HIPHessian(h, lev, img.ctx))
            SOL = newton(HIPHessian, T, F0, F1, F2, HIPVec(img.ctx, sizes[J]); maxit=mi, stopping_criterion=crit,
                         line_search=ls, printlog)
            its[J] += SOL.k
            SOL.converged && check(@ccall libmgbhip.mgbhip_prolong_add(h::Ptr{Cvoid}, lev::Int32, SOL.x.h::Ptr{Cvoid}, z.h::Ptr{Cvoid})::Cint)
            SOL.converged
        end
        mn(j, J) = (initial_step && J - j == 1) ? maxit : max_newton
        zsave = copy(z)
        converged = divide_and_conquer((j, J) -> eta(j, J, sc, mn(j, J)), 0, L)
        if finalize_now && !(fin isa NoFinalize)
            converged = eta(L - 1, L, fin, maxit) && converged
        end
        converged || (z = zsave)                                       # z = SOL.z only on success (src/mgb.jl:150-157)
        (; its, converged)
    end
    cdev(tt) = HIPVec(img.ctx, vec(tt .* c0))
    target = 1 / tol; kappa0 = kappa; t = T(t)
    itsall = Vector{Vector{Int}}(); ts = T[]; kappas = T[]
    t_begin = time()
    S = step(cdev(t), t >= target, true)
    S.converged || throw(MGBConvergenceFailure("Initial centering failed in mgb_solve at t=$t, tol=$tol, maxit=$maxit.", :stall))
    push!(itsall, S.its); push!(ts, t); push!(kappas, kappa)
    k = 1
    while t < target && kappa > 1 && k < maxit && !estop(z, t)
        k += 1; acc = zeros(Int, L)
        while kappa > 1
            t1 = kappa * t
            S = step(cdev(t1), t1 >= target, false); acc .+= S.its
            if S.converged
                maximum(S.its) <= max_newton * 0.5 && (kappa = min(kappa0, kappa^2))
                t = t1; break
            end
            kappa = sqrt(kappa)
        end
        push!(itsall, acc); push!(ts, t); push!(kappas, kappa)
    end
    (t >= target || estop(z, t)) || throw(MGBConvergenceFailure(
        "Convergence failure in mgb_solve at t=$t, k=$k, kappa=$kappa, tol=$tol, maxit=$maxit.", kappa <= 1 ? :stall : :iteration_limit))
    t_end = time()
    (; z = Array(z), its = reduce(hcat, itsall), ts, kappas, times = T[], c_dot_Dz = T[], t_begin, t_end,
       t_elapsed = t_end - t_begin, c = c0)
end

# ---- mgb_driver for the device image: src/mgb.jl:332-584 restated around the C entry points ----------
function mgb_driver(img::HIPImage{T}, f::Matrix{T}, g::Matrix{T}, Q::Convex{T}; keep_image::Bool=false, kw...) where {T}
    try
        return hip_mgb_driver(img, f, g, Q; kw...)
    finally
        # The reference flushes its plan / factorization caches after every solve, on the success and on the throw
        # path (src/mgb.jl:837-840).  Here they live in the image: release it now instead of waiting for the GC
        # (`keep_image=true` keeps it resident for a caller that re-solves, e.g. a time-stepping loop).
        keep_image || mgb_cleanup(img)
    end
end

function hip_mgb_driver(img::HIPImage{T}, f::Matrix{T}, g::Matrix{T}, Q::Convex{T};
                        t=T(0.1), t_feasibility=t, feasibility_Rmax=one(T) / sqrt(eps(T)), progress=x -> nothing,
                        printlog=(args...) -> nothing, tol=nothing, kappa=nothing, maxit=nothing, max_newton=nothing,
                        finalize=true, barrier_nodes=nothing, stopping_criterion=nothing, early_stop=nothing,
                        line_search=nothing, rest...) where {T}
    isempty(rest) || error("HIPDevice: unsupported keyword(s) $(keys(rest))")
    # mgb_driver releases the image in its `finally` (mgb_cleanup nulls ctx / main / feas): a second solve on the same
    # MGBProblem would hand C_NULL handles to libmgbhip
    (img.ctx == C_NULL || img.main == C_NULL) &&
        error("HIPDevice: this device image was already released by a previous solve; pass keep_image=true to " *
              "mgb_solve to reuse it, or call native_to_device(HIPDevice, prob) again")
    # `finalize`: true / false / NoFinalize() / a stopping_exact-style closure is not representable in mgbhip_options
    # beyond on/off + theta; closures other than the default run on the device-vector path below
    generic = line_search !== nothing || !(finalize isa Bool)
    state = (stopping_criterion === nothing && early_stop === nothing) ? nothing :
            CallbackState(stopping_criterion, early_stop, size(g, 2) * size(g, 1), nothing)
    M1 = img.M[1]
    m, nD, ncomp = length(M1.w), length(M1.D_fine), size(g, 2)
    L = length(M1.R_fine)
    # main barrier collocation weights (src/convex.jl:279-304): default = nodes with non-zero quadrature weight
    nz = barrier_nodes === nothing ? (M1.w .!= 0) : barrier_nodes
    bw = nz isa Colon || all(nz) ? C_NULL : (v = Float64.(nz) ./ count(nz); push!(img.keep, v); pointer(v))
    z2 = vec(copy(g))                                             # vcat of the columns (src/mgb.jl:409)
    SOL_feasibility = nothing
    F, _ = node_barrier(img.main, z2, m, nD)
    if !all(isfinite, F)                                          # infeasible start: phase I (src/mgb.jl:421-572)
        feas = feasibility_handle!(img)
        sl = node_slack(img.main, z2, m)
        z1 = vcat(z2, 2 .* max.(sl, one(T)))
        b = 2 * max(one(T), maximum(z1[ncomp*m+1:end]))
        c1 = zeros(T, m, nD + 1 + ncomp); c1[:, nD+1] .= one(T)
        Rbox = max(T(10), 10 * maximum(abs, z2)); Rmax = max(T(feasibility_Rmax), Rbox)
        slack_of(z) = @view z[ncomp*m+1:(ncomp+1)*m]
        while true
            printlog("mgb_driver: feasibility phase with bounding box R=", Rbox)
            check(@ccall libmgbhip.mgbhip_problem_set_box(feas::Ptr{Cvoid}, b::Float64, Rbox::Float64)::Cint)
            failure = nothing
            try
                fstate = stopping_criterion === nothing ? nothing : CallbackState(stopping_criterion, nothing, length(z1), nothing)
                o = options(feas, m; tol, t=t_feasibility, kappa, maxit, max_newton, early_stop=1, finalize=(finalize === true), state=fstate)
                SOL_feasibility = hip_mgb_core(feas, L, copy(z1), c1, o; state=fstate)
            catch e
                e isa InterruptException && rethrow(); failure = e      # each round is a probe (src/mgb.jl:505-515)
            end
            if failure === nothing
                zf = SOL_feasibility.z
                maximum(slack_of(zf)) < 0 && break
                vmax = maximum(abs, @view zf[1:ncomp*m])
                vmax <= Rbox / 2 && throw(MGBConvergenceFailure(
                    "The problem appears to be infeasible: the feasibility subproblem converged strictly inside the bounding box " *
                    "(max |nodal value| ~ $vmax <= R/2 with R = $Rbox) with positive constraint violation.", :infeasible))
            end
            10 * Rbox > Rmax && throw(MGBConvergenceFailure(
                "Could not find a strictly feasible point with nodal values bounded by R = $Rbox (cap feasibility_Rmax ~ $Rmax).",
                :feasibility_Rmax))
            Rbox *= 10
        end
        z2 = SOL_feasibility.z[1:ncomp*m]
        check(@ccall libmgbhip.mgbhip_problem_set_barrier_weights(img.main::Ptr{Cvoid}, bw::Ptr{Float64})::Cint)
        tm = Ref{Float64}(t)                                        # _matched_t (src/mgb.jl:307-330)
        check(@ccall libmgbhip.mgbhip_matched_t(img.main::Ptr{Cvoid}, z2::Ptr{Float64}, f::Ptr{Float64}, Float64(t)::Float64,
                                                tm::Ptr{Float64})::Cint)
        t = min(t, tm[])
    end
    check(@ccall libmgbhip.mgbhip_problem_set_barrier_weights(img.main::Ptr{Cvoid}, bw::Ptr{Float64})::Cint)
    SOL_main = if generic
        # a user line search (or finalize closure) needs the vectors: the reference's own `newton` runs on device
        # vectors through the fine-grained entry points (INTEGRATION.md section 2b)
        generic_mgb_core(img, img.main, z2, f; tol, t, kappa, maxit, max_newton, finalize, stopping_criterion, early_stop,
                         line_search, printlog)
    else
        o = options(img.main, m; tol, t, kappa, maxit, max_newton, early_stop=0, finalize, state)
        hip_mgb_core(img.main, L, z2, f, o; state)
    end
    progress(1.0)
    (; z = reshape(SOL_main.z, m, ncomp), SOL_feasibility, SOL_main)
end
# `mgb_solve` calls mgb_driver(prob.M, prob.f, prob.g, prob.Q; …) (src/mgb.jl:831): prob.M is the HIPImage.

function __init__()                                                 # like the CUDA extension's __init__ (:26-30)
    ctx = Ref{Ptr{Cvoid}}()
    try
        if (@ccall libmgbhip.mgbhip_create(ctx::Ptr{Ptr{Cvoid}}, 0::Cint, C_NULL::Ptr{Cvoid})::Cint) == 0
            @ccall libmgbhip.mgbhip_destroy(ctx[]::Ptr{Cvoid})::Cint
            default_device!(HIPDevice)
        end
    catch                                                           # library absent: CPUDevice stays the default
    end
end

end # module
