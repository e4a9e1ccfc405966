// Microbenchmark: LDS atomic / read / write throughput on gfx950 with every CU busy.
// hipcc --offload-arch=gfx950 -O3 tools/lds_bench.hip -o gpurun_out/lds_bench && ./gpurun_out/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NT = 512, ITER = 2048, SLOTS = 4096;

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(NT) void k(uint32_t *out, int conflict) {
    __shared__ unsigned long long t64[SLOTS];
    __shared__ uint32_t t32[SLOTS];
    for (int i = threadIdx.x; i < SLOTS; i += NT) { t64[i] = ~0ULL; t32[i] = 0; }
    __syncthreads();
    uint32_t s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1;
    uint32_t acc = 0;
    for (int it = 0; it < ITER; it++) {
        uint32_t r = lcg(s);
        uint32_t slot = conflict ? (r & 7) : (r & (SLOTS - 1));       // conflict: 8 hot addresses
        if (MODE == 0) atomicAdd(&t32[slot], 1u);                                   // no return
        if (MODE == 1) acc += atomicAdd(&t32[slot], 1u);                            // returning
        if (MODE == 2) acc += (uint32_t)atomicCAS(&t64[slot], ~0ULL, (unsigned long long)slot);   // 64-bit CAS rtn
        if (MODE == 3) acc += t32[slot];                                            // plain read b32
        if (MODE == 4) acc += (uint32_t)t64[slot];                                  // plain read b64
        if (MODE == 5) t32[slot] = r;                                               // plain write b32
        if (MODE == 6) t64[slot] = r;                                               // plain write b64
        if (MODE == 7) { uint64_t m = __ballot(r & 1); acc += (uint32_t)__popcll(m); }      // ballot only
        if (MODE == 8) acc += __shfl(r, (int)(r & 63), 64);                         // bpermute
        if (MODE == 9) acc += atomicAdd((unsigned long long *)&t64[slot], 1ULL) & 1;       // 64-bit add rtn
    }
    if (acc == 0x12345678u) out[0] = acc;
    if (MODE == 0 || MODE == 5 || MODE == 6) { __syncthreads(); if (t32[threadIdx.x] + (uint32_t)t64[threadIdx.x] == 0x12345678u) out[1] = 1; }
}

template <int MODE>
int run(const char *name, uint32_t *d, int conflict) {
    const int blocks = 256 * 4;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NT), 0, 0, d, conflict);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(NT), 0, 0, d, conflict);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * NT * ITER;
    // per CU: lane-ops per ns -> per clock at ~2.1 GHz
    double per_cu_per_ns = ops / 256.0 / (ms * 1e6);
    printf("%-28s conflict=%d  %8.3f ms  %7.2f lane-ops/ns/CU  (%.1f cycles per wave-instr @2.1GHz)\n", name, conflict, ms,
           per_cu_per_ns, 64.0 / per_cu_per_ns * 2.1);
    return 0;
}

int main() {
    uint32_t *d; CHECK(hipMalloc(&d, 64));
    for (int c = 0; c < 2; c++) {
        run<0>("ds_add_u32 (no return)", d, c);
        run<1>("ds_add_rtn_u32", d, c);
        run<2>("ds_cmpst_rtn_b64", d, c);
        run<9>("ds_add_rtn_u64", d, c);
        run<3>("ds_read_b32", d, c);
        run<4>("ds_read_b64", d, c);
        run<5>("ds_write_b32", d, c);
        run<6>("ds_write_b64", d, c);
    }
    run<7>("ballot+popc", d, 0);
    run<8>("ds_bpermute (shfl)", d, 0);
    return 0;
}
