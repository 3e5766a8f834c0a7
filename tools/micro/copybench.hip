// Streaming-copy probe: 8 B/lane vs 16 B/lane, read:write = 1:1 and 3.5:1 (the stage kernel's mix), sizes around the
// state buffers of the bench.  hipcc --offload-arch=gfx950 -O3 tools/micro/copybench.hip -o copybench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void copy8(const double *a, double *b, size_t n)
{
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n) b[i] = a[i];
}
__global__ void copy16(const double2 *a, double2 *b, size_t n2)
{
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i < n2) b[i] = a[i];
}
// planes: read R planes of n doubles, write W planes (one lane per "cell", like the stage kernel's own-cell accesses)
template <int R, int W, typename T>
__global__ void planes(const T *a, T *b, size_t n, size_t stride)
{
    const size_t i = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (i >= n) return;
    T v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = a[(size_t)r*stride + i];
#pragma unroll
    for (int w = 0; w < W; w++) {
        T s = v[w % R];
        if (w + W < R) { s.x += v[w + W].x; }
        b[(size_t)w*stride + i] = s;
    }
}
struct d1 { double x; };

template <typename F>
float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms/reps;
}

int main()
{
    const size_t cells[] = {1000000, 4000000};
    for (size_t nc : cells) {
        const size_t stride = (nc + 255)/256*256;
        const size_t n = 9*stride;                       // one state buffer
        double *a, *b;
        hipMalloc(&a, 18*stride*sizeof(double));
        hipMalloc(&b, 9*stride*sizeof(double));
        hipMemset(a, 0, 18*stride*sizeof(double));
        hipMemset(b, 0, 9*stride*sizeof(double));
        float t8 = timeit([&] { hipLaunchKernelGGL(copy8, dim3((n + 255)/256), dim3(256), 0, 0, a, b, n); }, 50);
        float t16 = timeit([&] { hipLaunchKernelGGL(copy16, dim3((n/2 + 255)/256), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, n/2); }, 50);
        const double gb = 2.0*n*8/1e9;
        printf("cells %zu  copy 1:1  8B/lane %.2f us %.0f GB/s   16B/lane %.2f us %.0f GB/s\n", nc, 1e3*t8, gb/(t8*1e-3), 1e3*t16, gb/(t16*1e-3));
        // 18 planes read, 9 written, one lane per cell (8 B) or per cell pair (16 B)
        float p8 = timeit([&] { hipLaunchKernelGGL((planes<18, 9, d1>), dim3((nc + 255)/256), dim3(256), 0, 0, (const d1 *)a, (d1 *)b, nc, stride); }, 50);
        float p16 = timeit([&] { hipLaunchKernelGGL((planes<18, 9, double2>), dim3((nc/2 + 255)/256), dim3(256), 0, 0, (const double2 *)a, (double2 *)b, nc/2, stride/2); }, 50);
        const double gbp = 27.0*nc*8/1e9;
        printf("cells %zu  planes 18r:9w  8B/lane %.2f us %.0f GB/s   16B/lane %.2f us %.0f GB/s\n", nc, 1e3*p8, gbp/(p8*1e-3), 1e3*p16, gbp/(p16*1e-3));
        hipFree(a); hipFree(b);
    }
    return 0;
}
