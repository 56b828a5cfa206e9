// Does data a kernel has just written come back from the memory-side cache (256 MB) faster than from HBM?
// kernel W writes S bytes, kernel R reads them (sum); R's rate against S.  Also read-after-read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_write(double2* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_double2((double)i, 1.0);
}
__global__ void k_read(const double2* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345) out[0] = s;
}
int main() {
    double2* buf; double* out;
    const size_t maxb = (size_t)4 << 30;
    hipMalloc(&buf, maxb); hipMalloc(&out, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
        size_t n = mb * 1024 * 1024 / 16;
        float best_rw = 1e9, best_rr = 1e9, best_w = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            // flush: touch 1 GB elsewhere
            k_write<<<2048, 256>>>(buf + ((size_t)3 << 30) / 16, ((size_t)1 << 30) / 16);
            hipEventRecord(e0); k_write<<<2048, 256>>>(buf, n); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); best_w = t < best_w ? t : best_w;
            hipEventRecord(e0); k_read<<<2048, 256>>>(buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1); best_rw = t < best_rw ? t : best_rw;
            hipEventRecord(e0); k_read<<<2048, 256>>>(buf, n, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t, e0, e1); best_rr = t < best_rr ? t : best_rr;
        }
        printf("%5zu MB: write %7.1f GB/s   read after write %7.1f GB/s   read after read %7.1f GB/s\n", mb, mb / 1024.0 / (best_w * 1e-3), mb / 1024.0 / (best_rw * 1e-3), mb / 1024.0 / (best_rr * 1e-3));
    }
    return 0;
}
