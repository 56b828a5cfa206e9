# FEASTHip.jl -- the binding a FEASTSolver.jl maintainer would add to call libfeasthip.so.
# NOT executed in this repository's CI: there is no Julia runtime in the build container
# or on the GPU box (SURVEY section 0).  It mirrors, call for call, the ctypes binding in
# feastsolver.jl_amd/api.py, which IS tested.  Style follows the reference's only ccall,
# src/lapack.jl:79-89 (caller-allocated outputs, status code checked after the call).
module FEASTHip

using LinearAlgebra, SparseArrays
import LinearAlgebra: ldiv!

const lib = get(ENV, "FEASTHIP_LIB", "libfeasthip.so")
const F64, C128, I64 = Cint(0), Cint(1), Cint(2)

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

# ---- fine-grained seam: factorizer / left_divider / finalize!  (src/utils.jl:173-179) ----
mutable struct HipLU <: Factorization{ComplexF64}
    handle::Ptr{Cvoid}
    n::Int
end

"factorizer(C): drop-in for `lu` in feast!(...; factorizer=hiplu)   src/feast.jl:29-36"
function hiplu(C::Matrix{ComplexF64}; device::Integer=0)
    F = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(C, 1)
    code = GC.@preserve C ccall((:feasthip_lu_dense, lib), Cint, (Cint, Ptr{ComplexF64}, Int64, Ref{Ptr{Cvoid}}),
                               device, C, n, F)
    code < 0 && check(C_NULL, code)
    HipLU(F[], n)
end

function hiplu(C::SparseMatrixCSC{ComplexF64,Int64}; device::Integer=0)
    F = Ref{Ptr{Cvoid}}(C_NULL)
    n = size(C, 1)
    code = GC.@preserve C ccall((:feasthip_lu_csc, lib), Cint,
        (Cint, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{ComplexF64}, Cint, Cint, Ref{Ptr{Cvoid}}),
        device, n, C.colptr, C.rowval, C.nzval, C128, 1, F)      # index_base = 1: fields passed zero-copy
    code < 0 && check(C_NULL, code)
    HipLU(F[], n)
end

"left_divider(Y, F, R): Y .= C \\ R, R untouched   src/feast.jl:62"
function ldiv!(Y::Matrix{ComplexF64}, F::HipLU, R::Matrix{ComplexF64})
    code = GC.@preserve Y R ccall((:feasthip_ldiv, lib), Cint,
        (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Ptr{ComplexF64}, Int64, Cint),
        F.handle, Y, stride(Y, 2), R, stride(R, 2), size(R, 2))
    code < 0 && throw(FeastHipError(code, "feasthip_ldiv"))
    Y
end

"finalize!(F): overload of src/utils.jl:173 releasing the device memory"
function finalize!(F::HipLU)
    F.handle == C_NULL || ccall((:feasthip_fact_free, lib), Cint, (Ptr{Cvoid},), F.handle)
    F.handle = C_NULL
    nothing
end

"trim_memory(): hand the library's cache of released device blocks back to the driver (a destroyed handle leaves its large
buffers for the next solver call of the process)"
trim_memory() = (ccall((:feasthip_trim_memory, lib), Cint, ()); nothing)

# ---- coarse seam: the whole loop of gen_feast! (src/feast.jl:89-156) in one call -------------
eltag(::Type{Float64}) = F64
eltag(::Type{ComplexF64}) = C128
eltag(::Type{Int64}) = I64

function gen_feast!(X::Matrix{ComplexF64}, A::SparseMatrixCSC{TA,Int64}, B::SparseMatrixCSC{TB,Int64}, contour;
                    iter::Integer=10, debug=false, store=false, ϵ=1e-12, device::Integer=0, conjugate_fold::Bool=true) where {TA,TB}
    N, m₀ = size(X)
    size(A, 1) != size(A, 2) && error("Incorrect dimensions of A, must be square")
    size(A, 1) != N && error("Incorrect dimensions of X, must match A")
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:feasthip_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device))
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    try
        colptr = [pointer(A.colptr), pointer(B.colptr)]
        rowval = [pointer(A.rowval), pointer(B.rowval)]
        nzval = [Ptr{Cvoid}(pointer(A.nzval)), Ptr{Cvoid}(pointer(B.nzval))]
        types = Cint[eltag(TA), eltag(TB)]
        scale = ComplexF64[1.0, -1.0]                     # T(z) = A - z B
        GC.@preserve A B X begin
            check(h[], ccall((:feasthip_set_csc, lib), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}, Cint),
                h[], N, 2, colptr, rowval, nzval, types, scale, 1))
            nodes, weights = ComplexF64.(contour.nodes), ComplexF64.(contour.weights)
            kind, region = contour isa CircularContour ? (Cint(0), Float64[real(contour.c), imag(contour.c), contour.r, 0.0]) :
                (Cint(1), Float64[real(contour.bottom_left), imag(contour.bottom_left), real(contour.top_right), imag(contour.top_right)])
            check(h[], ccall((:feasthip_set_contour, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint, Ptr{Float64}),
                h[], length(nodes), nodes, weights, kind, region))
            # real A, B + mirror-symmetric contour: one factorisation per conjugate node pair (default); false = one per node
            conjugate_fold || check(h[], ccall((:feasthip_set_conjugate_fold, lib), Cint, (Ptr{Cvoid}, Cint), h[], 0))
            check(h[], ccall((:feasthip_feast_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Cint, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
                h[], X, N, m₀, iter, ϵ, store ? 1 : 0, Λ, res, inside, nit))
        end
    finally
        ccall((:feasthip_destroy, lib), Cint, (Ptr{Cvoid},), h[])
    end
    in_c = inside .!= 0
    any(in_c) || println("no eigenvalues found in contour!")           # src/feast.jl:78
    Λ[in_c], X[:, in_c], res[in_c]
end

# ---- the other drivers of the hot path, same pattern: create -> terms -> contour -> *_run -> destroy ----------------
function with_handle(f::Function, device::Integer)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:feasthip_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device))
    try
        return f(h[])
    finally
        ccall((:feasthip_destroy, lib), Cint, (Ptr{Cvoid},), h[])
    end
end

# T(z) = Σ_j scales[j] z^(j-1) mats[j]; `nothing` stands for the identity (feasthip_set_csc: NULL pointer)
function set_terms!(h::Ptr{Cvoid}, N::Integer, mats::Vector, scales::Vector{ComplexF64})
    nt = length(mats)
    if all(M -> M === nothing || M isa SparseMatrixCSC, mats)
        colptr = [M === nothing ? Ptr{Int64}(C_NULL) : pointer(M.colptr) for M in mats]
        rowval = [M === nothing ? Ptr{Int64}(C_NULL) : pointer(M.rowval) for M in mats]
        nzval = [M === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(M.nzval)) for M in mats]
        types = Cint[M === nothing ? C128 : eltag(eltype(M)) for M in mats]
        check(h, ccall((:feasthip_set_csc, lib), Cint,
            (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}, Cint),
            h, N, nt, colptr, rowval, nzval, types, scales, 1))
    else                                                  # dense column-major matrices (src/feast.jl with Matrix arguments)
        ptrs = [M === nothing ? Ptr{Cvoid}(C_NULL) : Ptr{Cvoid}(pointer(M)) for M in mats]
        types = Cint[M === nothing ? C128 : eltag(eltype(M)) for M in mats]
        check(h, ccall((:feasthip_set_dense, lib), Cint,
            (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}),
            h, N, nt, ptrs, types, scales))
    end
end

function set_contour!(h::Ptr{Cvoid}, nodes::Vector{ComplexF64}, weights::Vector{ComplexF64}, kind::Cint, region::Vector{Float64})
    check(h, ccall((:feasthip_set_contour, lib), Cint,
        (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint, Ptr{Float64}),
        h, length(nodes), nodes, weights, kind, region))
end

function circle(c, r, nodes::Integer)                     # circular_contour_trapezoidal, src/contour.jl:26-31
    θ = LinRange(π / nodes, 2π - π / nodes, nodes)
    ComplexF64.(r .* exp.(θ .* im) .+ c), ComplexF64.(r .* exp.(θ .* im) ./ nodes), Cint(0), Float64[real(c), imag(c), r, 0.0]
end

"feast!(X, A; nodes, iter, c, r, ϵ, store, mixed_prec)   src/feast.jl:3-80 (T(z) = A - z I), dense or sparse A"
function feast!(X::Matrix{ComplexF64}, A::AbstractMatrix; nodes::Integer=8, iter::Integer=10, c=complex(0.0, 0.0), r=1.0,
                ϵ=1e-12, debug=false, store=false, mixed_prec=false, device::Integer=0)
    N, m₀ = size(X)
    size(A, 1) != size(A, 2) && error("Incorrect dimensions of A, must be square")
    size(A, 1) != N && error("Incorrect dimensions of X, must match A")
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        GC.@preserve A X begin
            set_terms!(h, N, Any[A, nothing], ComplexF64[1.0, -1.0])
            set_contour!(h, circle(c, r, nodes)...)
            mixed_prec && check(h, ccall((:feasthip_set_precision, lib), Cint, (Ptr{Cvoid}, Cint), h, 1))   # src/feast.jl:19-25
            check(h, ccall((:feasthip_feast_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Cint, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
                h, X, N, m₀, iter, ϵ, store ? 1 : 0, Λ, res, inside, nit))
        end
    end
    in_c = inside .!= 0
    any(in_c) || println("no eigenvalues found in contour!")           # src/feast.jl:78
    Λ[in_c], X[:, in_c], res[in_c]
end

"gen_feast!(X, A, B; nodes, iter, c, r, ...) for dense A, B   src/feast.jl:82-156"
function gen_feast!(X::Matrix{ComplexF64}, A::Matrix, B::Matrix; nodes::Integer=8, iter::Integer=10, c=complex(0.0, 0.0), r=1.0,
                    ϵ=1e-12, debug=false, store=false, device::Integer=0)
    N, m₀ = size(X)
    size(A, 1) != size(A, 2) && error("Incorrect dimensions of A, must be square")
    size(A, 1) != N && error("Incorrect dimensions of X, must match A")
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        GC.@preserve A B X begin
            set_terms!(h, N, Any[A, B], ComplexF64[1.0, -1.0])
            set_contour!(h, circle(c, r, nodes)...)
            check(h, ccall((:feasthip_feast_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Cint, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
                h, X, N, m₀, iter, ϵ, store ? 1 : 0, Λ, res, inside, nit))
        end
    end
    in_c = inside .!= 0
    any(in_c) || println("no eigenvalues found in contour!")
    Λ[in_c], X[:, in_c], res[in_c]
end

"nlfeast!(T, X, nodes, iter; c, r, ϵ, spurious) for a polynomial T(z) = Σ_j z^j coeffs[j+1]   src/nlfeast.jl:2-84: all m₀ pairs are returned"
function nlfeast!(coeffs::Vector{<:AbstractMatrix}, X::Matrix{ComplexF64}, nodes::Integer, iter::Integer;
                  c=complex(0.0, 0.0), r=1.0, debug=false, ϵ=10e-12, store=true, spurious=1e-5, device::Integer=0)
    N, m₀ = size(X)
    Λ, res, inside, nit = zeros(ComplexF64, m₀), zeros(m₀), zeros(UInt8, m₀), Ref{Cint}(0)
    with_handle(device) do h
        GC.@preserve coeffs X begin
            set_terms!(h, N, Any[M for M in coeffs], ones(ComplexF64, length(coeffs)))
            set_contour!(h, circle(c, r, nodes)...)
            check(h, ccall((:feasthip_nlfeast_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Float64, Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8}, Ref{Cint}),
                h, X, N, m₀, iter, ϵ, spurious, Λ, res, inside, nit))
        end
    end
    Λ, X, res
end

# nlfeast_moments!(T, X, nodes, iter; c, r, ϵ, moments, spurious)  (src/nlfeast.jl:173-318) for a polynomial
# T(z) = Σ_j z^j coeffs[j+1] with sparse coefficients (test/polynomial.jl:5-20): returns (Λ, Y, res) for all
# moments*m₀ candidates sorted by residual, X overwritten with the m₀ best (normalised).
function nlfeast_moments!(coeffs::Vector{SparseMatrixCSC{ComplexF64,Int64}}, X::Matrix{ComplexF64}, nodes::Integer, iter::Integer;
                          c=complex(0.0, 0.0), r=1.0, ϵ=10e-12, moments::Integer=2, spurious=1e-5, device::Integer=0)
    N, m₀ = size(X)
    M = moments * m₀
    h = Ref{Ptr{Cvoid}}(C_NULL)
    check(C_NULL, ccall((:feasthip_create, lib), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device))
    Λ, res, inside, Y, nit = zeros(ComplexF64, M), zeros(M), zeros(UInt8, M), zeros(ComplexF64, N, M), Ref{Cint}(0)
    try
        nt = length(coeffs)
        colptr = [pointer(A.colptr) for A in coeffs]
        rowval = [pointer(A.rowval) for A in coeffs]
        nzval = [Ptr{Cvoid}(pointer(A.nzval)) for A in coeffs]
        GC.@preserve coeffs X begin
            check(h[], ccall((:feasthip_set_csc, lib), Cint,
                (Ptr{Cvoid}, Int64, Cint, Ptr{Ptr{Int64}}, Ptr{Ptr{Int64}}, Ptr{Ptr{Cvoid}}, Ptr{Cint}, Ptr{ComplexF64}, Cint),
                h[], N, nt, colptr, rowval, nzval, fill(Cint(C128), nt), ones(ComplexF64, nt), 1))
            θ = LinRange(π / nodes, 2π - π / nodes, nodes)                              # src/nlfeast.jl:178
            z, w = ComplexF64.(r .* exp.(θ .* im) .+ c), ComplexF64.(r .* exp.(θ .* im) ./ nodes)
            check(h[], ccall((:feasthip_set_contour, lib), Cint,
                (Ptr{Cvoid}, Cint, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint, Ptr{Float64}),
                h[], nodes, z, w, 0, Float64[real(c), imag(c), r, 0.0]))
            check(h[], ccall((:feasthip_nlfeast_moments_run, lib), Cint,
                (Ptr{Cvoid}, Ptr{ComplexF64}, Int64, Cint, Cint, Cint, Float64, Float64, Ptr{ComplexF64}, Ptr{Float64}, Ptr{UInt8},
                 Ptr{ComplexF64}, Ref{Cint}),
                h[], X, N, m₀, moments, iter, ϵ, spurious, Λ, res, inside, Y, nit))
        end
    finally
        ccall((:feasthip_destroy, lib), Cint, (Ptr{Cvoid},), h[])
    end
    Λ, Y, res
end

end # module
