// Microbenchmark: what does the MI355X memory system give a radix scatter of 16-byte records into
// S private streams per workgroup, as a function of the burst length b (records written to one
// stream back to back by adjacent lanes)?  hipcc --offload-arch=gfx950 -O3 tools/scatter_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct alignas(16) Rec { unsigned long long a, b; };

template <bool READ>
__global__ __launch_bounds__(512) void k_scatter(const Rec *__restrict__ in, Rec *__restrict__ out, int S, int b,
                                                 long per_wg, int shuffle) {
    const long wg = blockIdx.x;
    Rec *base = out + wg * per_wg;
    const Rec *src = in + wg * per_wg;
    const long slen = per_wg / S;                 // records per stream
    const int groups = 512 / b;                   // lane groups per iteration
    const long iters = per_wg / 512;
    for (long t = 0; t < iters; t++) {
        const long gi = t * groups + threadIdx.x / b;       // global group visit index
        long s = gi % S;
        const long visit = gi / S;
        if (shuffle) s = (s * 167 + 13) % S;                  // de-correlate neighbouring groups' streams
        const long pos = visit * b + threadIdx.x % b;
        Rec r;
        if (READ) r = src[t * 512 + threadIdx.x]; else { r.a = (unsigned long long)t; r.b = threadIdx.x; }
        if (pos < slen) base[s * slen + pos] = r;
    }
}

int main(int argc, char **argv) {
    const long total = 640L << 20;                // 640 Mi records = 10.7 GB
    int S = argc > 1 ? atoi(argv[1]) : 512;
    Rec *in, *out;
    CK(hipMalloc(&in, total * sizeof(Rec)));
    CK(hipMalloc(&out, total * sizeof(Rec)));
    CK(hipMemset(in, 1, total * sizeof(Rec)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {256, 512, 1024}) {
        const long per_wg = total / wgs;
        for (int read = 0; read < 2; read++)
            for (int b : {1, 2, 4, 8, 16, 32, 64}) {
                float best = 1e9;
                for (int rep = 0; rep < 3; rep++) {
                    CK(hipEventRecord(e0));
                    if (read) hipLaunchKernelGGL(k_scatter<true>, dim3(wgs), dim3(512), 0, 0, in, out, S, b, per_wg, 1);
                    else hipLaunchKernelGGL(k_scatter<false>, dim3(wgs), dim3(512), 0, 0, in, out, S, b, per_wg, 1);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("S=%d wgs=%d read=%d burst=%2d recs (%4d B): %7.3f ms  write %.2f TB/s\n", S, wgs, read, b, b * 16, best,
                       total * 16.0 / best / 1e9);
            }
    }
    return 0;
}
