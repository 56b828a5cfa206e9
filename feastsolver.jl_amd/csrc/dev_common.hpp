// Shared host/device definitions for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>

namespace feasthip {

// Function attributes (dynamic LDS above 64 KB) are per device: true the first time the calling thread's current device asks.
inline bool first_on_this_device(std::atomic<uint64_t>& seen) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return !(seen.fetch_or(bit) & bit);
}

typedef double2 cz;                                             // complex128 (re, im)
typedef float2 cf;                                              // complex64, mixed-precision solve path
typedef double d4 __attribute__((ext_vector_type(4)));         // f64 MFMA accumulator fragment

// One supernodal front.  Its L panel is (p+u) x p column-major (ld = p+u) and
// holds L11\U11 in the top p rows and L21 below; U12 is p x u column-major.
struct FrontD {
    int c0, p, u, parent;
    int64_t lp, f12, cb;      // element offsets: factor arena (lp, f12), CB arena of depth&1 (cb)
    int64_t sptr;             // offset into struct_idx / rel
    int64_t crow;             // row offset into the solve-contribution arena of depth&1
    int64_t child_ptr;
    int nchild, depth;
    int nb, wslot;            // nb: diagonal super-block size of the solve-ready form of this front (128, 256 or 512), see kernels.hpp;
                              // wslot: position of the front within its tree level (scratch slot of the pivot-light group kernels)
    int64_t sk_off;           // row offset of this front's split-K partial sums in the scratch (backward U12 product), or -1
    int64_t skf_off;          // ... of the forward L21 product (part q of contribution row i: skf_off + q u + i), or -1
    int64_t cmap;             // offset into the contribution-row pull map (u x nchild ints), or -1 without children
    int64_t ebp;              // as a child: offset into the extend-add break points (first contribution row that lands at or
                              // beyond parent row 64 b, b = 0 .. ceil(ld_parent / 64)), or -1 for a root
    int64_t pmap;             // offset into DevSym::pmap, the pull map of the pivot rows (p x nchild ints), or -1 without children
};

struct Tile { int f, a, b, c; };
// Work split of one launch_apply launch over the 8 XCDs (workgroup b runs on XCD b % 8): XCD x owns the tiles [off[x], off[x + 1]) of
// the launch's list for every node and column group; the grid is padded to cmax = the largest part (the spare workgroups exit at once).
// The parts are cut by the planner (Engine::build_plans) for equal estimated time, not for equal tile counts.  perm >= 0: the parts index
// DevSym::xperm + perm (tile numbers of the launch, dealt to the XCDs) instead of the list itself, whose order -- longest tiles first --
// is the one a single XCD sweeping a whole node wants.
struct XcdParts { int off[9]; int cmax; int perm; };

__host__ __device__ inline cz cz_make(double x, double y) { cz r; r.x = x; r.y = y; return r; }
__host__ __device__ inline cz cz_mul(cz a, cz b) { return cz_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__host__ __device__ inline cz cz_add(cz a, cz b) { return cz_make(a.x + b.x, a.y + b.y); }
__host__ __device__ inline cz cz_sub(cz a, cz b) { return cz_make(a.x - b.x, a.y - b.y); }
__host__ __device__ inline cz cz_conj(cz a) { return cz_make(a.x, -a.y); }
// a -= b*c
__host__ __device__ inline void cz_msub(cz& a, cz b, cz c) {
    a.x -= b.x * c.x - b.y * c.y;
    a.y -= b.x * c.y + b.y * c.x;
}
__host__ __device__ inline void cz_madd(cz& a, cz b, cz c) {
    a.x += b.x * c.x - b.y * c.y;
    a.y += b.x * c.y + b.y * c.x;
}
__host__ __device__ inline cz cz_inv(cz a) {
    // Smith's algorithm (overflow-safe reciprocal)
    if (fabs(a.x) >= fabs(a.y)) {
        double r = a.y / a.x, d = a.x + a.y * r;
        return cz_make(1.0 / d, -r / d);
    }
    double r = a.x / a.y, d = a.x * r + a.y;
    return cz_make(r / d, -1.0 / d);
}
__host__ __device__ inline double cz_abs1(cz a) { return fabs(a.x) + fabs(a.y); }

}  // namespace feasthip
