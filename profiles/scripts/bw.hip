#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void rd(const double2 *p, double *o, size_t n) {
    double s = 0; size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) { double2 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st]; s += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y; }
    if (s == 1.2345) o[0] = s;
}
__global__ void wr(double2 *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_double2(1.0, 2.0);
}
__global__ void cp(const double2 *p, double2 *q, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i + 3 * st < n; i += 4 * st) { double2 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st]; q[i] = a; q[i + st] = b; q[i + 2 * st] = c; q[i + 3 * st] = d; }
}
int main() {
    for (size_t mb : {64, 512, 2048}) {
        size_t bytes = mb << 20, n = bytes / 16;
        double2 *a, *b; double *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
        hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](const char *name, int kind, double traffic) {
            float best = 1e9;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(rd, dim3(256 * 8), dim3(256), 0, 0, a, o, n);
                if (kind == 1) hipLaunchKernelGGL(wr, dim3(256 * 8), dim3(256), 0, 0, b, n);
                if (kind == 2) hipLaunchKernelGGL(cp, dim3(256 * 8), dim3(256), 0, 0, a, b, n);
                if (kind == 3) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%5zu MB %-8s %8.1f us  %6.2f TB/s (bytes moved)\n", mb, name, best * 1e3, traffic / (best * 1e-3) * 1e-12);
        };
        run("read", 0, (double)bytes); run("write", 1, (double)bytes); run("copy", 2, 2.0 * bytes); run("memcpy", 3, 2.0 * bytes);
        hipFree(a); hipFree(b); hipFree(o);
    }
}
