// Micro-benchmark of the factor-slab read patterns of k_apply (no arithmetic): how fast can a
// column-major slab be streamed with (a) the MFMA-fragment pattern (16 rows x 4 columns per wave
// instruction = 4 x 256 B segments) and (b) one column x 64 rows per wave instruction (1 KiB)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void slab_read(const double2* __restrict__ A, long lda, int K, double* out) {
    int tile = blockIdx.x;                       // 64 rows per tile
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double2* base = A + (long)tile * 64;
    double s = 0;
    if (MODE == 0) {
        int row = wave * 16 + li;
        for (int kc = 0; kc < K; kc += 32) {
            double2 v[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) v[ks] = base[row + (long)(kc + ks * 4 + lk) * lda];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) s += v[ks].x + v[ks].y;
        }
    } else {
        for (int kc = 0; kc < K; kc += 32) {
            double2 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = base[lane + (long)(kc + wave + 4 * q) * lda];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q].x + v[q].y;
        }
    }
    if (s == 1.2345) out[0] = s;
}

int main() {
    long rows = 64 * 4096, K = 1024, lda = rows;          // 4096 tiles x 1024 columns = 4.3 GB
    double2* A; double* out;
    CK(hipMalloc(&A, rows * K * 16)); CK(hipMalloc(&out, 64)); CK(hipMemset(A, 0, rows * K * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) slab_read<0><<<4096, 256>>>(A, lda, (int)K, out); else slab_read<1><<<4096, 256>>>(A, lda, (int)K, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("pattern %s: %.3f ms  %.2f TB/s\n", mode == 0 ? "mfma-fragment 4x256B" : "column 1KiB        ", ms, rows * K * 16 / ms / 1e9);
        }
    return 0;
}
