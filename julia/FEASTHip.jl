# FEASTHip.jl -- the binding a FEASTSolver.jl maintainer would add to call libfeasthip.so.
# NOT executed in this repository's CI: there is no Julia runtime in the build container
# or on the GPU box (SURVEY section 0).  It mirrors, call for call, the ctypes binding in
# feastsolver.jl_amd/api.py, which IS tested.  Style follows the reference's only ccall,
# src/lapack.jl:79-89 (caller-allocated outputs, status code checked after the call).
#
# Two ways in, both leave src/feast.jl and src/nlfeast.jl untouched:
#   fine   feast!(X, A, contour; store=true, factorizer=FEASTHip.hiplu)       (left_divider stays ldiv!)
#          -> the reference's own loop; F = hiplu(ZmA), ldiv!(temp, F, R), finalize!(F) reach the GPU
#   coarse FEASTHip.feast!/gen_feast!/nlfeast!(...) with the reference's signatures -> the whole loop in the library
module FEASTHip

using LinearAlgebra, SparseArrays
import LinearAlgebra: ldiv!
import FEASTSolver                                        # the package this plugs into
import FEASTSolver: finalize!                             # src/utils.jl:173 `finalize!(x::Any) = nothing`: EXTENDED below, so that
                                                          # `foreach(finalize!, facts)` (src/feast.jl:75,151) and linsolve!
                                                          # (src/utils.jl:178) release the device memory of a HipLU
using FEASTSolver: Contour, CircularContour, RectangularContour, circular_contour_trapezoidal

const lib = get(ENV, "FEASTHIP_LIB", "libfeasthip.so")
const F64, C128, I64, C64 = Cint(0), Cint(1), Cint(2), Cint(3)

struct FeastHipError <: Exception
    code::Int
    msg::String
end

function check(h::Ptr{Cvoid}, code::Integer)
    code >= 0 && return code
    msg = unsafe_string(ccall((:feasthip_last_error, lib), Cstring, (Ptr{Cvoid},), h))
    code == -2 && error("Incorrect dimensions of A, must be square")     # src/feast.jl:13-14
    code == -3 && error("Incorrect dimensions of X, must match A")       # src/feast.jl:15-16
    throw(FeastHipError(code, msg))
end

# =====================================================================================================================
# fine-grained seam: factorizer / left_divider / finalize!   (src/utils.jl:173-179)
# =====================================================================================================================
# A concrete subtype of Factorization{ComplexF64}: `facts = Array{typeof(facts1)}(undef, nodes)` (src/feast.jl:31,107) is then
# a Vector{HipLU}, and `facts = Array{Factorization}(undef, nodes)` (src/nlfeast.jl:18) accepts it as well.
mutable struct HipLU <: Factorization{ComplexF64}
    handle::Ptr{Cvoid}
    n::Int
    single::Bool                                          # factorised from a ComplexF32 matrix (mixed_prec): solves in complex64
end
Base.size(F::HipLU) = (F.n, F.n)
Base.size(F::HipLU, i::Integer) = i <= 2 ? F.n : 1

function _wrap(F::Ref{Ptr{Cvoid}}, n::Integer, code::Integer, single::Bool)
    code < 0 && check(C_NULL, code)
    obj = HipLU(F[], n, single)
    finalizer(finalize!, obj)                             # a factor object dropped without finalize! still frees its HBM at GC time
    obj
end

"factorizer(C): drop-in for `lu` in feast!(...; factorizer=hiplu).  C is copied during the call (the reference re-uses ZmA
for the next node, src/feast.jl:29-36).  Callable from Threads.@threads (src/feast.jl:34-37, src/nlfeast.jl:19-26)."
function hiplu(C::Matrix{ComplexF64}; device::Integer=0)
    F = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(C, 1)
    code = GC.@preserve C ccall((:feasthip_lu_dense, lib), Cint, (Cint, Ptr{ComplexF64}, Int64, Ref{Ptr{Cvoid}}), device, C, n, F)
    _wrap(F, n, code, false)
end

"factorizer(ZmA::Matrix{ComplexF32}) of the mixed_prec path (src/feast.jl:19-36)"
function hiplu(C::Matrix{ComplexF32}; device::Integer=0)
    F = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(C, 1)
    code = GC.@preserve C ccall((:feasthip_lu_dense_c64, lib), Cint, (Cint, Ptr{ComplexF32}, Int64, Ref{Ptr{Cvoid}}), device, C, n, F)
    _wrap(F, n, code, true)
end

function hiplu(C::SparseMatrixCSC{Tv,Int64}; device::Integer=0) where {Tv<:Union{ComplexF64,ComplexF32}}
    F = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(C, 1)
    code = GC.@preserve C ccall((:feasthip_lu_csc, lib), Cint,
        (Cint, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Cint, Cint, Ref{Ptr{Cvoid}}),
        device, n, C.colptr, C.rowval, C.nzval, Tv === ComplexF64 ? C128 : C64, 1, F)      # index_base = 1: fields passed zero-copy
    _wrap(F, n, code, Tv === ComplexF32)
end
hiplu(C::AbstractMatrix; device::Integer=0) = hiplu(issparse(C) ? SparseMatrixCSC{ComplexF64,Int64}(C) : Matrix{ComplexF64}(C); device=device)

"left_divider(Y, F, R): Y .= C \\ R, R untouched   src/feast.jl:62, src/nlfeast.jl:41,49"
function ldiv!(Y::Matrix{ComplexF64}, F::HipLU, R::Matrix{ComplexF64})
    code = GC.@preserve Y R ccall((:feasthip_ldiv, lib), Cint,
        (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Ptr{ComplexF64}, Int64, Cint),
        F.handle, Y, stride(Y, 2), R, stride(R, 2), size(R, 2))
    code < 0 && throw(FeastHipError(code, "feasthip_ldiv"))
    Y
end

"left_divider(temp::Matrix{ComplexF32}, F, R) of the mixed_prec path: `temp` is ComplexF32 (src/feast.jl:22,62); the library
returns complex128 (solved in complex64 when F came from a ComplexF32 matrix), narrowed into `temp` here."
function ldiv!(Y::Matrix{ComplexF32}, F::HipLU, R::Matrix{ComplexF64})
    Yd = Matrix{ComplexF64}(undef, size(Y))
    ldiv!(Yd, F, R)
    Y .= Yd
    Y
end
ldiv!(Y::AbstractMatrix, F::HipLU, R::AbstractMatrix) = (Y .= ldiv!(Matrix{ComplexF64}(undef, size(R)), F, Matrix{ComplexF64}(R)); Y)

"left_divider(Y, F', R): Y .= C' \\ R over the SAME factor object -- forward with U', backward with L' (feasthip_ldiv_adjoint).  The
reference factorises the adjoint matrices a second time (`lfacts`, src/feast.jl:185-197) and solves with them at :241; with
`factorizer=hiplu` a caller can keep ONE factor per node and write `ldiv!(temp, facts[i]', Rl)`."
function ldiv!(Y::Matrix{ComplexF64}, Fa::Adjoint{<:Any,HipLU}, R::Matrix{ComplexF64})
    F = parent(Fa)
    code = GC.@preserve Y R ccall((:feasthip_ldiv_adjoint, lib), Cint,
        (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Ptr{ComplexF64}, Int64, Cint),
        F.handle, Y, stride(Y, 2), R, stride(R, 2), size(R, 2))
    code < 0 && throw(FeastHipError(code, "feasthip_ldiv_adjoint"))
    Y
end

"FEASTSolver.finalize!(F::HipLU): the overload of src/utils.jl:173 that releases the device memory (src/feast.jl:75,151)"
function finalize!(F::HipLU)
    F.handle == C_NULL || ccall((:feasthip_fact_free, lib), Cint, (Ptr{Cvoid},), F.handle)
    F.handle = C_NULL
    nothing
end

"trim_memory(): hand the library's cache of released device blocks back to the driver"
trim_memory() = (ccall((:feasthip_trim_memory, lib), Cint, ()); nothing)

# =====================================================================================================================
# coarse seam: the whole loops behind the reference's signatures
# =====================================================================================================================
eltag(::Type{Float64}) = F64
eltag(::Type{ComplexF64}) = C128
eltag(::Type{Int64}) = I64

function with_handle(f::Function, device::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:feasthip_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device))
    try
        return f(h[])
    finally
        ccall((:feasthip_destroy, lib), Cint, (Ptr{Cvoid},), h[])
    end
end

# what the C ABI takes: SparseMatrixCSC{Float64|ComplexF64|Int64, Int64}, Matrix of the same element types, or `nothing` (= I).
# Everything else the reference accepts as AbstractMatrix (Diagonal: test/runtests.jl:21; Int matrices from diagm: :16; other
# index or element types) is converted once here.
_native(M::Nothing, sparse::Bool) = nothing
_native(M::SparseMatrixCSC{T,Int64}, sparse::Bool) where {T<:Union{Float64,ComplexF64,Int64}} = sparse ? M : Matrix(M)
_native(M::Matrix{T}, sparse::Bool) where {T<:Union{Float64,ComplexF64,Int64}} = sparse ? sparse_of(M) : M
_native(M::AbstractMatrix, sparse::Bool) = (T = eltype(M) <: Real ? Float64 : ComplexF64;
                                            sparse ? SparseMatrixCSC{T,Int64}(SparseArrays.sparse(M)) : Matrix{T}(M))
sparse_of(M::Matrix{T}) where {T} = SparseMatrixCSC{T,Int64}(SparseArrays.sparse(M))

# T(z) = Σ_j scales[j] z^(j-1) mats[j]; `nothing` stands for the identity (NULL pointer).  The problem is sparse as soon as ONE
# matrix is sparse (or Diagonal next to nothing else dense): mixed sparse/dense pencils go through the CSC path.
function set_terms!(h::Ptr{Cvoid}, N::Integer, mats_in::Vector, scales::Vector{ComplexF64})
    sp = any(M -> M isa AbstractSparseMatrix, mats_in)
    mats = Any[_native(M, sp) for M in mats_in]
    nt = length(mats)
    GC.@preserve mats begin
        if sp
            colptr = [M === nothing ? Ptr{Int64}(C_NULL) : pointer(M.colptr) for M in mats]
            rowval = [M === nothing ? Ptr{Int64}(C_NULL) : pointer(M.rowval) for M in mats]
            nzval = [M === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(M.nzval)) for M in mats]
            types = Cint[M === nothing ? C128 : eltag(eltype(M)) for M in mats]
            check(h, ccall((:feasthip_set_csc, lib), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}, Cint),
                h, N, nt, colptr, rowval, nzval, types, scales, 1))
        else                                                  # dense column-major matrices
            ptrs = [M === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(M)) for M in mats]
            types = Cint[M === nothing ? C128 : eltag(eltype(M)) for M in mats]
            check(h, ccall((:feasthip_set_dense, lib), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}),
                h, N, nt, ptrs, types, scales))
        end
    end
end

# contour.nodes / contour.weights + the region of in_contour (src/contour.jl:88-100)
function set_contour!(h::Ptr{Cvoid}, contour::Contour)
    nodes, weights = ComplexF64.(contour.nodes), ComplexF64.(contour.weights)
    kind, region = if contour isa CircularContour
        Cint(0), Float64[real(contour.c), imag(contour.c), contour.r, 0.0]
    elseif contour isa RectangularContour
        Cint(1), Float64[real(contour.bottom_left), imag(contour.bottom_left), real(contour.top_right), imag(contour.top_right)]
    else
        error("FEASTHip: in_contour is defined for CircularContour and RectangularContour only (src/contour.jl:88-100)")
    end
    check(h, ccall((:feasthip_set_contour, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint, Ptr{Float64}),
        h, length(nodes), nodes, weights, kind, region))
end

# X in/out: the library wants a Matrix{ComplexF64}; anything else is staged and copied back (the `!` convention)
_stage(X::Matrix{ComplexF64}) = X
_stage(X::AbstractMatrix) = Matrix{ComplexF64}(X)
_unstage!(X::Matrix{ComplexF64}, Xw::Matrix{ComplexF64}) = X
_unstage!(X::AbstractMatrix, Xw::Matrix{ComplexF64}) = (eltype(X) <: Complex && (X .= Xw); X)

# the plug-in kwargs are honoured, never ignored: with the defaults the loop runs inside the library; anything else means the
# caller wants its own solver in the reference's loop, so the call is forwarded to FEASTSolver unchanged (with hiplu that is
# the fine seam above)
_default_plugins(factorizer, left_divider) = (factorizer === lu || factorizer === hiplu) && left_divider === ldiv!

"""
    set_lazy_conversion(h, on::Bool)

Factorisations that serve one contour step (drivers of the phase calls in the `store=false` regime of `src/feast.jl:64-65`) keep the
update rows of their factors raw (`feasthip_set_lazy_conversion`); the whole-loop calls do this by themselves for `store=false`.
"""
set_lazy_conversion(h::Ptr{Cvoid}, on::Bool) = check(h, ccall((:feasthip_set_lazy_conversion, lib), Cint, (Ptr{Cvoid}, Cint), h, on ? 1 : 0))

function _linear_run(X::AbstractMatrix, A::AbstractMatrix, B, contour::Contour, iter, ϵ, store, mixed_prec, device, conjugate_fold)
    N, m₀ = size(X)
    size(A, 1) != size(A, 2) && error("Incorrect dimensions of A, must be square")     # src/feast.jl:13-14, 93-94
    size(A, 1) != N && error("Incorrect dimensions of X, must match A")                # :15-16, 95-96
    Xw = _stage(X)
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        set_terms!(h, N, Any[A, B], ComplexF64[1.0, -1.0])                             # T(z) = A - z B  (B = nothing: I)
        set_contour!(h, contour)
        conjugate_fold || check(h, ccall((:feasthip_set_conjugate_fold, lib), Cint, (Ptr{Cvoid}, Cint), h, 0))
        mixed_prec && check(h, ccall((:feasthip_set_precision, lib), Cint, (Ptr{Cvoid}, Cint), h, 1))      # src/feast.jl:19-25
        GC.@preserve Xw check(h, ccall((:feasthip_feast_run, lib), Cint,
            (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Cint, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
            h, Xw, N, m₀, iter, ϵ, store ? 1 : 0, Λ, res, inside, nit))
    end
    _unstage!(X, Xw)
    in_c = inside .!= 0
    any(in_c) || println("no eigenvalues found in contour!")                           # src/feast.jl:78, 154
    Λ[in_c], Xw[:, in_c], res[in_c]
end

"feast!(X, A, contour; iter, ϵ, debug, store, mixed_prec, factorizer, left_divider)   src/feast.jl:10-11"
function feast!(X::AbstractMatrix, A::AbstractMatrix, contour::Contour;
                iter::Integer=10, ϵ=1e-12, debug=false, store=false, mixed_prec=false, factorizer=lu, left_divider=ldiv!,
                device::Integer=0, conjugate_fold::Bool=true)
    _default_plugins(factorizer, left_divider) ||
        return FEASTSolver.feast!(X, A, contour; iter=iter, ϵ=ϵ, debug=debug, store=store, mixed_prec=mixed_prec,
                                  factorizer=factorizer, left_divider=left_divider)
    _linear_run(X, A, nothing, contour, iter, ϵ, store, mixed_prec, device, conjugate_fold)
end

"feast!(X, A; nodes, iter, c, r, ϵ, debug, store, mixed_prec, factorizer, left_divider)   src/feast.jl:3-8"
function feast!(X::AbstractMatrix, A::AbstractMatrix;
                nodes::Integer=8, iter::Integer=10, c=complex(0.0, 0.0), r=1.0, ϵ=1e-12,
                debug=false, store=false, mixed_prec=false, factorizer=lu, left_divider=ldiv!, kw...)
    contour = circular_contour_trapezoidal(c, r, nodes)
    feast!(X, A, contour; iter=iter, debug=debug, ϵ=ϵ, store=store, mixed_prec=mixed_prec, factorizer=factorizer, left_divider=left_divider, kw...)
end

"gen_feast!(X, A, B, contour; iter, debug, store, ϵ, factorizer, left_divider)   src/feast.jl:89-90.  A, B: any AbstractMatrix
the reference takes -- Matrix, SparseMatrixCSC, Diagonal (test/runtests.jl:21), mixed."
function gen_feast!(X::AbstractMatrix, A::AbstractMatrix, B::AbstractMatrix, contour::Contour;
                    iter::Integer=10, debug=false, store=false, ϵ=1e-12, factorizer=lu, left_divider=ldiv!,
                    device::Integer=0, conjugate_fold::Bool=true)
    _default_plugins(factorizer, left_divider) ||
        return FEASTSolver.gen_feast!(X, A, B, contour; iter=iter, debug=debug, store=store, ϵ=ϵ, factorizer=factorizer, left_divider=left_divider)
    _linear_run(X, A, B, contour, iter, ϵ, store, false, device, conjugate_fold)
end

"gen_feast!(X, A, B; nodes, iter, c, r, debug, store, ϵ, factorizer, left_divider)   src/feast.jl:82-87"
function gen_feast!(X::AbstractMatrix, A::AbstractMatrix, B::AbstractMatrix;
                    nodes::Integer=8, iter::Integer=10, c=complex(0.0, 0.0), r=1.0,
                    debug=false, store=false, ϵ=1e-12, factorizer=lu, left_divider=ldiv!, kw...)
    contour = circular_contour_trapezoidal(c, r, nodes)
    gen_feast!(X, A, B, contour; iter=iter, debug=debug, store=store, ϵ=ϵ, factorizer=factorizer, left_divider=left_divider, kw...)
end

# ---- nlfeast! -------------------------------------------------------------------------------------------------------
"Polynomial T(z) = Σ_j z^j coeffs[j+1] as a callable, so that `nlfeast!(T, X, nodes, iter)` keeps the reference's argument list
(src/nlfeast.jl:2; test/butterfly.jl:60-62 builds exactly this closure) while the library sees the coefficient matrices."
struct PolynomialNEP{M<:AbstractMatrix} <: Function
    coeffs::Vector{M}
end
(T::PolynomialNEP)(z) = sum(z^(j - 1) * T.coeffs[j] for j in eachindex(T.coeffs))

"nlfeast!(T, X, nodes, iter; c, r, debug, ϵ, store, spurious, factorizer, left_divider)   src/nlfeast.jl:2-4: all m₀ pairs returned.
T::PolynomialNEP runs inside the library; any other closure is forwarded to the reference's loop with the GPU factorizer
(fine seam), which is the generic-T path."
function nlfeast!(T::PolynomialNEP, X::AbstractMatrix{ComplexF64}, nodes::Integer, iter::Integer;
                  c=complex(0.0, 0.0), r=1.0, debug=false, ϵ=10e-12, store=true, spurious=1e-5, factorizer=lu, left_divider=ldiv!,
                  device::Integer=0)
    _default_plugins(factorizer, left_divider) ||
        return FEASTSolver.nlfeast!(T, X, nodes, iter; c=c, r=r, debug=debug, ϵ=ϵ, store=store, spurious=spurious,
                                    factorizer=factorizer, left_divider=left_divider)
    N, m₀ = size(X)
    Xw = _stage(X)
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        set_terms!(h, N, Any[M for M in T.coeffs], ones(ComplexF64, length(T.coeffs)))
        set_contour!(h, circular_contour_trapezoidal(c, r, nodes))                    # src/nlfeast.jl:8
        GC.@preserve Xw check(h, ccall((:feasthip_nlfeast_run, lib), Cint,
            (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
            h, Xw, N, m₀, iter, ϵ, spurious, Λ, res, inside, nit))
    end
    _unstage!(X, Xw)
    Λ, Xw, res
end
nlfeast!(T::Function, X::AbstractMatrix{ComplexF64}, nodes::Integer, iter::Integer; factorizer=hiplu, left_divider=ldiv!, kw...) =
    FEASTSolver.nlfeast!(T, X, nodes, iter; factorizer=factorizer, left_divider=left_divider, kw...)

# nlfeast_moments!(T, X, nodes, iter; c, r, ϵ, moments, spurious)  (src/nlfeast.jl:173-318) for a polynomial T:
# returns (Λ, Y, res) for all moments*m₀ candidates sorted by residual, X overwritten with the m₀ best (normalised).
function nlfeast_moments!(T::PolynomialNEP, X::AbstractMatrix{ComplexF64}, nodes::Integer, iter::Integer;
                          c=complex(0.0, 0.0), r=1.0, debug=false, ϵ=10e-12, moments::Integer=2, store=true, spurious=1e-5, device::Integer=0)
    N, m₀ = size(X)
    M = moments * m₀
    Xw = _stage(X)
    Λ, res, inside, Y, nit = zeros(ComplexF64, M), zeros(M), zeros(UInt8, M), zeros(ComplexF64, N, M), Ref{Cint}(0)
    with_handle(device) do h
        set_terms!(h, N, Any[A for A in T.coeffs], ones(ComplexF64, length(T.coeffs)))
        set_contour!(h, circular_contour_trapezoidal(c, r, nodes))                    # src/nlfeast.jl:178
        GC.@preserve Xw check(h, ccall((:feasthip_nlfeast_moments_run, lib), Cint,
            (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Cint, Float64, Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8},
             Ptr{ComplexF64}, Ref{Cint}),
            h, Xw, N, m₀, moments, iter, ϵ, spurious, Λ, res, inside, Y, nit))
    end
    _unstage!(X, Xw)
    Λ, Y, res
end

# ---- dual_gen_feast!, contour_estimate_eig (SURVEY 8f rank 3) -------------------------------------------------------------
"dual_gen_feast!(Xr, Xl, A, B, contour; iter, debug, store, ϵ, factorizer, left_divider)   src/feast.jl:165-257: right and left
eigenvectors.  K factorisations of A - z_k B; the left systems (A - z_k B)' y = r (the reference's `lfacts`, :185-197, :241) are
adjoint sweeps over the same factors.  Returns (Λ[inside], Xr[:, inside], Xl[:, inside], resr[inside]) like the reference (:256)."
function dual_gen_feast!(Xr::AbstractMatrix, Xl::AbstractMatrix, A::AbstractMatrix, B, contour::Contour;
                         iter::Integer=10, debug=false, store=false, ϵ=1e-12, factorizer=lu, left_divider=ldiv!, device::Integer=0)
    _default_plugins(factorizer, left_divider) ||
        return FEASTSolver.dual_gen_feast!(Xr, Xl, A, B, contour; iter=iter, debug=debug, store=store, ϵ=ϵ, factorizer=factorizer, left_divider=left_divider)
    N, m₀ = size(Xl)
    size(A, 1) != size(A, 2) && error("Incorrect dimensions of A, must be square")     # src/feast.jl:169-170
    size(A, 1) != N && error("Incorrect dimensions of X, must match A")                # :171-172
    Bm = B isa UniformScaling ? nothing : B
    Xrw, Xlw = _stage(Xr), _stage(Xl)
    z = ComplexF64.(contour.nodes)
    K = length(z)
    table = zeros(ComplexF64, 4, K)                                                    # row-major K x 4 for the C side: (1, z_k, 0, 0)
    table[1, :] .= 1.0; table[2, :] .= z
    Λ, resr, resl, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        set_terms!(h, N, Any[A, Bm, copy(A'), Bm === nothing ? nothing : copy(Bm')], ComplexF64[1.0, -1.0, 1.0, -1.0])
        set_contour!(h, contour)
        check(h, ccall((:feasthip_set_node_coefficients, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Cint, Cint), h, table, K, 4))
        GC.@preserve Xrw Xlw check(h, ccall((:feasthip_dual_feast_run, lib), Cint,
            (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Cint, Ptr{ComplexF64}, Ptr{Float64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
            h, Xrw, Xlw, N, m₀, iter, ϵ, store ? 1 : 0, Λ, resr, resl, inside, nit))
    end
    _unstage!(Xr, Xrw); _unstage!(Xl, Xlw)
    in_c = inside .!= 0
    any(in_c) || println("no eigenvalues found in contour!")                           # src/feast.jl:255
    Λ[in_c], Xrw[:, in_c], Xlw[:, in_c], resr[in_c]
end
dual_gen_feast!(Xr::AbstractMatrix, Xl::AbstractMatrix, A::AbstractMatrix, B::AbstractMatrix;
                nodes::Integer=8, iter::Integer=10, c=complex(0.0, 0.0), r=1.0, kw...) =                    # src/feast.jl:158-163
    dual_gen_feast!(Xr, Xl, A, B, circular_contour_trapezoidal(c, r, nodes); iter=iter, kw...)

"contour_estimate_eig(A, contour, B=I; samples, ...)   src/stochastic.jl:2-33: stochastic estimate of the number of eigenvalues inside
the contour, est = real(Σ_k w_k tr(X'(z_k B - A)⁻¹X)) / samples with a Gaussian block X drawn here like the reference (:14-15)."
function contour_estimate_eig(A::AbstractMatrix, contour::Contour, B=I; samples::Integer=min(100, size(A, 1)), ϵ=1e-12, debug=false,
                              mixed_prec=false, factorizer=lu, left_divider=ldiv!, device::Integer=0)
    _default_plugins(factorizer, left_divider) ||
        return FEASTSolver.contour_estimate_eig(A, contour, B; samples=samples, ϵ=ϵ, debug=debug, mixed_prec=mixed_prec, factorizer=factorizer, left_divider=left_divider)
    N = size(A, 1)
    X = randn(ComplexF64, N, samples)
    est = Ref{Float64}(0.0)
    with_handle(device) do h
        set_terms!(h, N, Any[A, B isa UniformScaling ? nothing : B], ComplexF64[1.0, -1.0])
        set_contour!(h, contour)
        mixed_prec && check(h, ccall((:feasthip_set_precision, lib), Cint, (Ptr{Cvoid}, Cint), h, 1))
        GC.@preserve X check(h, ccall((:feasthip_estimate_count, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Ref{Float64}), h, X, N, samples, est))
    end
    est[]
end

# ---- nlfeast_moments! for an opaque closure T (test/gun.jl:36, test/fiber.jl:33, string.jl:33) -----------------------------
"nlfeast_moments!(T, X, nodes, iter; c, r, ϵ, moments, spurious)   src/nlfeast.jl:173-318 for ANY closure z -> matrix: T(z_k) of every
node goes to the device once (one term per node, identity coefficient table), the loop runs here with the phase calls and the
residual block R[:,i] = T(λ_i) y_i, res_i = ‖R_i‖ / ‖T(λ_i)‖_F is formed on the host exactly like update_R_moments!
(src/utils.jl:118-134).  Returns (Λ, Y, res) for all moments·m₀ candidates, sorted by residual."
function nlfeast_moments!(T::Function, X::AbstractMatrix{ComplexF64}, nodes::Integer, iter::Integer;
                          c=complex(0.0, 0.0), r=1.0, debug=false, ϵ=10e-12, moments::Integer=2, store=true, spurious=1e-5, device::Integer=0)
    N, m₀ = size(X)
    M = moments * m₀
    contour = circular_contour_trapezoidal(c, r, nodes)
    Xw = _stage(X)
    Λ, res, inside = zeros(ComplexF64, M), zeros(M), zeros(UInt8, M)
    Y, R = zeros(ComplexF64, N, M), zeros(ComplexF64, N, M)
    with_handle(device) do h
        set_terms!(h, N, Any[T(z) for z in contour.nodes], ones(ComplexF64, nodes))
        set_contour!(h, contour)
        eye = Matrix{ComplexF64}(I, nodes, nodes)
        check(h, ccall((:feasthip_set_node_coefficients, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Cint, Cint), h, eye, nodes, nodes))
        GC.@preserve Xw check(h, ccall((:feasthip_nlm_begin, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint), h, Xw, N, m₀, moments))
        for nit in 0:iter                                                               # src/nlfeast.jl:200
            check(h, ccall((:feasthip_nlm_contour_step, lib), Cint, (Ptr{Cvoid}, Cint), h, nit == 0 ? 1 : 0))
            check(h, ccall((:feasthip_nlm_candidates, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}), h, Λ, Y))
            resin = zeros(M)
            for i in 1:M                                                                # update_R_moments!  src/utils.jl:118-123
                isfinite(real(Λ[i])) || (resin[i] = Inf; continue)
                Tl = T(Λ[i])
                R[:, i] = Tl * Y[:, i]
                resin[i] = norm(R[:, i]) / norm(Tl)
            end
            check(h, ccall((:feasthip_nlm_finish, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}),
                           h, R, resin, Λ, res, inside))
            in_c = inside .!= 0
            any(in_c) && maximum(res[in_c]) < ϵ && break                                # src/nlfeast.jl:299-302
        end
        check(h, ccall((:feasthip_nlm_get_y, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), h, Y))
        check(h, ccall((:feasthip_nl_finalize_x, lib), Cint, (Ptr{Cvoid},), h))
        check(h, ccall((:feasthip_get_x, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), h, Xw))
    end
    _unstage!(X, Xw)
    Λ, Y, res
end

# ---- multi-GPU: one process per GPU, the all-reduce inside the library (include/feasthip.h: feasthip_comm_*) -------------------
"128 bytes (RCCL's ncclUniqueId) drawn on rank 0; hand them to the other ranks by whatever the host program uses (Distributed, MPI.jl)."
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(C_NULL, ccall((:feasthip_comm_unique_id, lib), Cint, (Ptr{UInt8},), id))
    id
end

"gen_feast! with the node loop (src/feast.jl:134-148) sharded over `world` processes, one GPU each: every rank passes the same X, A, B,
contour and the id of rank 0 and receives the same result.  Rank g owns every world-th unit with its factorisations; the moment Q is
summed over the ranks by RCCL inside feasthip_contour_step; the Rayleigh-Ritz step (:117-127) runs redundantly on every rank."
function gen_feast_distributed!(X::AbstractMatrix, A::AbstractMatrix, B, contour::Contour, id::Vector{UInt8}, rank::Integer, world::Integer;
                                iter::Integer=10, ϵ=1e-12, store=true, device::Integer=0)
    N, m₀ = size(X)
    Xw = _stage(X)
    Λ, res, inside = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀)
    with_handle(device) do h
        set_terms!(h, N, Any[A, B isa UniformScaling ? nothing : B], ComplexF64[1.0, -1.0])
        set_contour!(h, contour)
        check(h, ccall((:feasthip_set_node_shard, lib), Cint, (Ptr{Cvoid}, Cint, Cint), h, rank, world))
        check(h, ccall((:feasthip_comm_init, lib), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), h, id, rank, world))
        store && check(h, ccall((:feasthip_factorize, lib), Cint, (Ptr{Cvoid},), h))
        GC.@preserve Xw check(h, ccall((:feasthip_set_x, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint), h, Xw, N, m₀))
        for nit in 0:iter                                                               # src/feast.jl:116
            check(h, ccall((:feasthip_rr_step, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}), h, Λ, res, inside))
            in_c = inside .!= 0
            any(in_c) && maximum(res[in_c]) < ϵ && break                                # :129-132
            nit < iter && check(h, ccall((:feasthip_contour_step, lib), Cint, (Ptr{Cvoid}, Cint), h, store ? 0 : 1))   # :133-148 + the sum over the ranks
        end
        check(h, ccall((:feasthip_get_x, lib), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}), h, Xw))
    end
    _unstage!(X, Xw)
    in_c = inside .!= 0
    Λ[in_c], Xw[:, in_c], res[in_c]
end

end # module
