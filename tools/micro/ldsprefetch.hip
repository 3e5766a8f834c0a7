// Model of the stage kernel's memory behaviour: per cell 18 plane reads + 9 plane writes (8 B each), NFMA dependent FP64
// FMAs per lane, 3 waves per SIMD (12 one-wave workgroups per CU, forced through the LDS allocation).
//   A: one workgroup per 64 cells, direct loads (what the stage kernel does)
//   C: persistent waves looping over tiles, direct loads
//   B: persistent waves, the NEXT tile's 18 planes are fetched into LDS by buffer_load_dwordx4 ... lds while the current
//      tile is computed (no VGPRs hold data in flight)
// Result on MI355X (GB/s of the 27 planes, 1 M / 4 M cells): no compute A 7950 / 5610, C 7730 / 4940, B 7720 / 5260; with 1080
// FP64 VALU instructions per lane A 7430 / 5840, C 6680 / 5040, B 6370 / 5000  =>  at 3 waves/SIMD a streaming kernel hides
// ~1000 VALU instructions completely with plain loads, and neither persistence nor LDS prefetch adds anything: what the
// stage kernel loses against this model (40 vs 29 us at 1 M, 210 vs 148 us at 4 M) is its dependent index -> gather phase.
// CAVEAT: variant B is a timing model only - the final check shows that the LDS image of the second plane group (LDS
// offsets >= 4 KB) is not what the read-back expects; it issues the same loads, which is all the timing needs.
// hipcc --offload-arch=gfx950 -O3 tools/micro/ldsprefetch.hip -o ldsprefetch && LDSPREFETCH_TIMING=1 ./ldsprefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define LDS_PER_WAVE 13312      // bytes: 160 KB / 12 waves per CU -> 3 waves per SIMD

__device__ __forceinline__ rsrc_t mk(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xffffffff, 0x00020000); }
__device__ __forceinline__ double ld(rsrc_t r, unsigned v, unsigned s) { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0)); }
__device__ __forceinline__ void st(rsrc_t r, unsigned v, unsigned s, double x) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), r, v, s, 0); }

template <int NFMA>
__device__ __forceinline__ void compute(const double v[18], double o[9])
{
#pragma unroll
    for (int i = 0; i < 9; i++) o[i] = v[i] + v[9 + i];
    // dependent chains: 9 accumulators, NFMA/9 rounds
#pragma unroll
    for (int j = 0; j < NFMA/9; j++) {
#pragma unroll
        for (int i = 0; i < 9; i++) o[i] = fma(o[i], 1.0000001, v[(i + j) % 18]*1e-9);
    }
}

template <int NFMA>
__global__ __launch_bounds__(64) void kernA(const double *in, double *out, int n, unsigned S8)
{
    __shared__ char pad[LDS_PER_WAVE];
    const int k = blockIdx.x*64 + threadIdx.x;
    if (k >= n) return;
    if (n < 0) pad[threadIdx.x] = 1;       // keep the allocation
    const rsrc_t ri0 = mk(in), ri1 = mk(reinterpret_cast<const char *>(in) + 9ull*S8), ro = mk(out);
    double v[18], o[9];
#pragma unroll
    for (int p = 0; p < 9; p++) { v[p] = ld(ri0, k*8u, p*S8); v[9 + p] = ld(ri1, k*8u, p*S8); }
    compute<NFMA>(v, o);
#pragma unroll
    for (int p = 0; p < 9; p++) st(ro, k*8u, p*S8, o[p]);
    if (n < 0) out[0] = pad[1];
}

template <int NFMA>
__global__ __launch_bounds__(64) void kernC(const double *in, double *out, int n, unsigned S8, int ntiles)
{
    __shared__ char pad[LDS_PER_WAVE];
    if (n < 0) pad[threadIdx.x] = 1;
    const rsrc_t ri0 = mk(in), ri1 = mk(reinterpret_cast<const char *>(in) + 9ull*S8), ro = mk(out);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int k = t*64 + threadIdx.x;
        if (k >= n) break;
        double v[18], o[9];
#pragma unroll
        for (int p = 0; p < 9; p++) { v[p] = ld(ri0, k*8u, p*S8); v[9 + p] = ld(ri1, k*8u, p*S8); }
        compute<NFMA>(v, o);
#pragma unroll
        for (int p = 0; p < 9; p++) st(ro, k*8u, p*S8, o[p]);
    }
    if (n < 0) out[0] = pad[1];
}

// LDS image of a tile: plane p of the wave's 64 cells at byte offset p*512.  One dwordx4 LDS load moves 64 lanes x 16 B =
// two planes: lanes 0..31 fetch plane 2j (cells 2l, 2l+1), lanes 32..63 plane 2j+1.
__device__ __forceinline__ void prefetch(rsrc_t ri0, rsrc_t ri1, double *lds, int tile, unsigned S8)
{
    const unsigned lane = threadIdx.x;
    const unsigned voff = (unsigned)tile*512u + (lane & 31u)*16u + (lane >> 5)*S8;
#pragma unroll
    for (int j = 0; j < 4; j++)        // planes 0..7 of group 0
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ri0, (__attribute__((address_space(3))) void *)(lds + 128*j), 16, voff, 2u*j*S8, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; j++)        // planes 9..16 = group 1 planes 0..7
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ri1, (__attribute__((address_space(3))) void *)(lds + 128*(4 + j)), 16, voff, 2u*j*S8, 0, 0);
    // planes 8 and 17: lanes 0..31 plane 8 of group 0, lanes 32..63 plane 8 of group 1 (= + 9 planes from group 0)
    const unsigned voff2 = (unsigned)tile*512u + (lane & 31u)*16u + (lane >> 5)*9u*S8;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ri0, (__attribute__((address_space(3))) void *)(lds + 128*8), 16, voff2, 8u*S8, 0, 0);
}

template <int NFMA>
__global__ __launch_bounds__(64) void kernB(const double *in, double *out, int n, unsigned S8, int ntiles)
{
    __shared__ double lds[LDS_PER_WAVE/8];
    const rsrc_t ri0 = mk(in), ri1 = mk(reinterpret_cast<const char *>(in) + 9ull*S8), ro = mk(out);
    int t = blockIdx.x;
    if (t >= ntiles) return;
    prefetch(ri0, ri1, lds, t, S8);
    for (; t < ntiles; t += gridDim.x) {
        const int k = t*64 + threadIdx.x;
        double v[18], o[9];
        // LDS order: [g0 p0][g0 p1]...[g0 p7][g1 p0]...[g1 p7][g0 p8][g1 p8]
#pragma unroll
        for (int p = 0; p < 8; p++) { v[p] = lds[64*p + threadIdx.x]; v[9 + p] = lds[64*(8 + p) + threadIdx.x]; }
        v[8] = lds[64*16 + threadIdx.x];
        v[17] = lds[64*17 + threadIdx.x];
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the reads are done before the next DMA may overwrite the tile
        const int tn = t + gridDim.x;
        if (tn < ntiles) prefetch(ri0, ri1, lds, tn, S8);
        compute<NFMA>(v, o);
        if (k < n) {
#pragma unroll
            for (int p = 0; p < 9; p++) st(ro, k*8u, p*S8, o[p]);
        }
    }
}

template <typename F>
float timeit(F f, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms/reps;
}

template <int NFMA>
void run(size_t nc)
{
    const size_t stride = (nc + 255)/256*256;
    double *a, *b, *c;
    hipMalloc(&a, 18*stride*sizeof(double));
    hipMalloc(&b, 9*stride*sizeof(double));
    hipMalloc(&c, 9*stride*sizeof(double));
    hipMemset(a, 0, 18*stride*sizeof(double));
    const unsigned S8 = (unsigned)(stride*8);
    const int ntiles = (int)((nc + 63)/64);
    const int pgrid = 256*12;
    // warm the clocks
    for (int i = 0; i < 300; i++) hipLaunchKernelGGL(kernA<NFMA>, dim3(ntiles), dim3(64), 0, 0, a, b, (int)nc, S8);
    hipDeviceSynchronize();
    float ta = timeit([&] { hipLaunchKernelGGL(kernA<NFMA>, dim3(ntiles), dim3(64), 0, 0, a, b, (int)nc, S8); }, 40);
    float tc = timeit([&] { hipLaunchKernelGGL(kernC<NFMA>, dim3(pgrid), dim3(64), 0, 0, a, b, (int)nc, S8, ntiles); }, 40);
    float tb = timeit([&] { hipLaunchKernelGGL(kernB<NFMA>, dim3(pgrid), dim3(64), 0, 0, a, c, (int)nc, S8, ntiles); }, 40);
    // check B against A on a random-ish input
    const double gb = 27.0*nc*8/1e9;
    printf("cells %zu NFMA %d: A %.1f us %.0f GB/s | C(persistent) %.1f us %.0f GB/s | B(LDS prefetch) %.1f us %.0f GB/s\n",
           nc, NFMA, 1e3*ta, gb/(ta*1e-3), 1e3*tc, gb/(tc*1e-3), 1e3*tb, gb/(tb*1e-3));
    hipFree(a); hipFree(b); hipFree(c);
}

int main()
{
    if (getenv("LDSPREFETCH_TIMING")) {
        run<0>(1000000); run<0>(4000000);
        run<180>(1000000); run<180>(4000000);
        run<360>(1000000); run<360>(4000000);
        run<540>(1000000); run<540>(4000000);
    }
    // correctness of the LDS image: B == A for a non-trivial input
    {
        const size_t nc = 100000, stride = (nc + 255)/256*256;
        double *a, *b, *c;
        hipMalloc(&a, 18*stride*8); hipMalloc(&b, 9*stride*8); hipMalloc(&c, 9*stride*8);
        double *h = (double *)malloc(18*stride*8);
        for (size_t i = 0; i < 18*stride; i++) h[i] = (double)(i % 1000003)*1e-3;
        hipMemcpy(a, h, 18*stride*8, hipMemcpyHostToDevice);
        hipMemset(b, 0, 9*stride*8); hipMemset(c, 0, 9*stride*8);
        const unsigned S8 = (unsigned)(stride*8);
        const int ntiles = (int)((nc + 63)/64);
        hipLaunchKernelGGL(kernA<0>, dim3(ntiles), dim3(64), 0, 0, a, b, (int)nc, S8);
        hipLaunchKernelGGL(kernB<0>, dim3(256*12), dim3(64), 0, 0, a, c, (int)nc, S8, ntiles);
        double *hb = (double *)malloc(9*stride*8), *hc = (double *)malloc(9*stride*8);
        hipMemcpy(hb, b, 9*stride*8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 9*stride*8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t p = 0; p < 9; p++) for (size_t i = 0; i < nc; i++) if (hb[p*stride + i] != hc[p*stride + i]) {
            if (bad < 6 || (bad % 100000) == 0) printf("  plane %zu cell %zu: A %.6f B %.6f\n", p, i, hb[p*stride + i], hc[p*stride + i]);
            bad++;
        }
        printf("LDS-prefetch result check: %zu mismatches\n", bad);
    }
    return 0;
}
