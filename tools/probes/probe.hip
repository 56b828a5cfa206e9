// Hardware probe for gfx950: checks the f64 MFMA fragment layout the kernels
// assume and measures the rates DESIGN.md prices the roofline against
// (f64 MFMA, f64 VALU FMA, f32 MFMA, HBM copy).  Stand-alone: hipcc probe.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__global__ void mfma_layout(const double* A, const double* B, double* D) {
    // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major
    int l = threadIdx.x;
    double a = A[(l & 15) * 4 + (l >> 4)];
    double b = B[(l >> 4) * 16 + (l & 15)];
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

__global__ void mfma_f64_rate(double* out, int iters) {
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void mfma_f32_rate(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
    }
    f4 s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void valu_f64_rate(double* out, int iters) {
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double m = 1.0000001, a = 1e-9;
    for (int i = 0; i < iters; ++i) {
        x0 = fma(x0, m, a); x1 = fma(x1, m, a); x2 = fma(x2, m, a); x3 = fma(x3, m, a);
        x4 = fma(x4, m, a); x5 = fma(x5, m, a); x6 = fma(x6, m, a); x7 = fma(x7, m, a);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

__global__ void copy16(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

__global__ void read16(const double2* __restrict__ in, double* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    double s = 0;
    for (; i < n; i += stride) { double2 v = in[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}

template <class F> static float time_ms(F f, int reps) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz mem %.1f GB gcn %s\n", p.name, p.multiProcessorCount, p.clockRate, p.totalGlobalMem / 1e9, p.gcnArchName);
    // layout
    std::vector<double> A(64), B(64), D(256), Dr(256, 0.0);
    for (int i = 0; i < 64; ++i) { A[i] = (i * 7 % 13) - 6 + 0.25 * i; B[i] = (i * 5 % 11) - 5 - 0.125 * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) Dr[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    double *dA, *dB, *dD; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    mfma_layout<<<1, 64>>>(dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost));
    double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(D[i] - Dr[i]));
    printf("mfma_f64_16x16x4 layout max err %.3e  (%s)\n", err, err < 1e-12 ? "LAYOUT OK" : "LAYOUT MISMATCH");

    int ncu = p.multiProcessorCount;
    double* out; CK(hipMalloc(&out, sizeof(double) * ncu * 8 * 1024));
    for (int wpc : {4, 8, 16}) {   // waves per CU
        int blocks = ncu * wpc / 4, iters = 4096;
        float ms = time_ms([&] { mfma_f64_rate<<<blocks, 256>>>(out, iters); }, 5);
        double fl = (double)blocks * 4 * iters * 4 * 2048.0;
        printf("f64 mfma  waves/CU %2d: %.2f ms  %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
        ms = time_ms([&] { mfma_f32_rate<<<blocks, 256>>>((float*)out, iters); }, 5);
        printf("f32 mfma  waves/CU %2d: %.2f ms  %.1f TFLOP/s\n", wpc, ms, fl / ms / 1e9);
        ms = time_ms([&] { valu_f64_rate<<<blocks, 256>>>(out, iters * 8); }, 5);
        double fv = (double)blocks * 256 * iters * 8 * 8 * 2.0;
        printf("f64 valu  waves/CU %2d: %.2f ms  %.1f TFLOP/s\n", wpc, ms, fv / ms / 1e9);
    }
    {   // sustained f64 MFMA rate (DVFS): ~0.5 s of back-to-back launches
        int blocks = ncu * 4, iters = 16384;
        for (int rep = 0; rep < 4; ++rep) {
            float ms = time_ms([&] { mfma_f64_rate<<<blocks, 256>>>(out, iters); }, 20);
            double fl = (double)blocks * 4 * iters * 4 * 2048.0;
            printf("f64 mfma sustained rep %d: %.2f ms/launch  %.1f TFLOP/s\n", rep, ms, fl / ms / 1e9);
        }
    }
    size_t n = (size_t)1 << 27;  // 2 GiB of double2
    double2 *in, *o2; CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&o2, n * 16));
    CK(hipMemset(in, 1, n * 16)); CK(hipMemset(o2, 0, n * 16));
    float ms = time_ms([&] { copy16<<<ncu * 8, 256>>>(in, o2, n); }, 5);
    printf("copy 2x%.1f GB: %.3f ms  %.2f TB/s (read+write)\n", n * 16 / 1e9, ms, 2.0 * n * 16 / ms / 1e9);
    ms = time_ms([&] { read16<<<ncu * 8, 256>>>(in, out, n); }, 5);
    printf("read %.1f GB: %.3f ms  %.2f TB/s\n", n * 16 / 1e9, ms, 1.0 * n * 16 / ms / 1e9);
    // launch overhead
    ms = time_ms([&] { for (int i = 0; i < 100; ++i) read16<<<256, 64>>>(in, out, 64); }, 5);
    printf("100 tiny launches: %.3f ms (%.2f us each)\n", ms, ms * 10);
    return 0;
}
