// Can gfx950 stream HBM and run the f64 matrix cores at the same time?  Each wave streams 8 KiB per
// iteration (8 x 16 B per lane, software pipelined one iteration ahead) and issues M f64 MFMAs per
// iteration whose A operand is the data loaded one iteration earlier.  Reports TB/s and TFLOP/s per M.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

template <int M, bool LOADS>
__global__ __launch_bounds__(256, 2) void mix(const double2* __restrict__ in, double* out, size_t per_wave_iters, size_t wave_stride) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double2* p = in + wave * wave_stride + lane;        // wave-contiguous 1 KiB segments
    double2 cur[8], nxt[8];
    d4 acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = (d4){0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = p[j * 64];
    double b = p[512 * 64 + 3].x + 1.0;
    for (size_t it = 1; it <= per_wave_iters; ++it) {
        if (LOADS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) nxt[j] = p[(it * 8 + j) * 64];
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double a = (m & 1) ? cur[(m >> 1) & 7].y : cur[(m >> 1) & 7].x;
            acc[m % 12] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[m % 12], 0, 0, 0);
        }
        if (LOADS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j].x = -cur[j].x;
        }
    }
    d4 s = acc[0];
#pragma unroll
    for (int q = 1; q < 12; ++q) s += acc[q];
    double r = s[0] + s[1] + s[2] + s[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += cur[j].x + cur[j].y;
    if (r == 12345.6789) out[0] = r;
}

template <int M, bool LOADS>
void run(const double2* in, double* out, int ncu, size_t bytes_total) {
    int blocks = ncu * 2 * 4;                         // 8 workgroups per CU in the grid, 2 resident
    size_t waves = (size_t)blocks * 4;
    size_t per_wave = bytes_total / waves / 8192;     // iterations of 8 KiB
    size_t wave_stride = (per_wave + 1) * 512;        // in double2 units
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mix<M, LOADS>), dim3(blocks), dim3(256), 0, 0, in, out, per_wave, wave_stride);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mix<M, LOADS>), dim3(blocks), dim3(256), 0, 0, in, out, per_wave, wave_stride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    double by = (double)waves * per_wave * 8192.0, fl = (double)waves * per_wave * M * 2048.0;
    printf("M=%3d loads=%d: %.3f ms  %.2f TB/s  %.1f TFLOP/s (f64 mfma)\n", M, (int)LOADS, ms, LOADS ? by / ms / 1e9 : 0.0, fl / ms / 1e9);
}

int main(int argc, char** argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int ncu = p.multiProcessorCount;
    size_t bytes = (size_t)8 << 30;
    double2* in; double* out;
    CK(hipMalloc(&in, bytes + (64 << 20))); CK(hipMalloc(&out, 4096));
    if (argc > 1) {      // random data: the matrix cores and the memory system then toggle like the real kernel's (the clock depends on it)
        size_t n64 = (size_t)(8 << 20);
        std::vector<double> h(n64);
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < n64; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
        for (size_t off = 0; off < bytes + (64 << 20); off += n64 * 8) CK(hipMemcpy((char*)in + off, h.data(), std::min(n64 * 8, bytes + (64 << 20) - off), hipMemcpyHostToDevice));
        printf("random data\n");
    } else CK(hipMemset(in, 0, bytes + (64 << 20)));
    run<0, true>(in, out, ncu, bytes);
    run<12, true>(in, out, ncu, bytes);
    run<24, true>(in, out, ncu, bytes);
    run<36, true>(in, out, ncu, bytes);
    run<48, true>(in, out, ncu, bytes);
    run<72, true>(in, out, ncu, bytes);
    run<96, true>(in, out, ncu, bytes);
    run<48, false>(in, out, ncu, bytes);
    run<96, false>(in, out, ncu, bytes);
    return 0;
}
