// Micro-benchmark: issue cost of the integer VALU instructions the leaf / level-1 kernels are made of, on gfx950, with every
// CU busy at the leaf kernel's occupancy (12 waves per workgroup, two workgroups per CU = 6 waves per SIMD) and at one wave
// per SIMD.  Output: cycles per wave-instruction per SIMD (s_memtime ticks of the whole kernel / instructions per wave /
// waves per SIMD) -- the number that turns "84 VALU wave-instructions per k-mer slot" into a time floor.
// hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o gpurun_out/valu_bench && ./gpurun_out/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 512, UNR = 16;     // ITER x UNR x (ops per body) instructions per thread

template <int MODE>
__global__ void k(uint32_t *out, uint32_t seed, unsigned long long *clk) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = b * 31u + 7u, d = c ^ (a >> 3);
    uint64_t x = ((uint64_t)a << 32) | b, y = ((uint64_t)c << 32) | d;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            // four independent chains (a, b, c, d) so that the issue rate, not a dependency, is measured
            if (MODE == 0) { a += b; b += c; c += d; d += a; }                                            // v_add_u32
            // (round 4: xor and multiply chains are linear maps the compiler folded away -- 0.006 ms, "0.08 cycles" in
            // profiles/r03_valu_bench.txt; as volatile inline assembly they are the instructions themselves)
            if (MODE == 1) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(b) : "v"(c));
                             asm volatile("v_xor_b32 %0, %0, %1" : "+v"(c) : "v"(d)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(a)); }   // v_xor_b32
            if (MODE == 2) { a = __builtin_amdgcn_alignbit(a, b, 7); b = __builtin_amdgcn_alignbit(b, c, 9);
                             c = __builtin_amdgcn_alignbit(c, d, 11); d = __builtin_amdgcn_alignbit(d, a, 13); }   // v_alignbit_b32
            if (MODE == 3) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(d | 1u)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b) : "v"(d | 1u));
                             asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d | 1u)); }                  // v_mul_lo_u32 (three chains, one odd multiplier)
            if (MODE == 11) { asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a) : "v"(d)); asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(b) : "v"(d));
                              asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(c) : "v"(d)); }                 // v_mad_u32_u24 (level 1's minimiser order)
            if (MODE == 4) { x = (x << 3) ^ y; y = (y >> 5) ^ x; x = (x << 7) ^ y; y = (y >> 9) ^ x; }      // 64-bit shifts + xors
            if (MODE == 5) { a = __brev(a) ^ b; b = __brev(b) ^ c; c = __brev(c) ^ d; d = __brev(d) ^ a; }  // v_bfrev + xor
            if (MODE == 6) { a = a < b ? a : b + 1; b = b < c ? b : c + 1; c = c < d ? c : d + 1; d = d < a ? d : a + 1; }   // min-like: cmp + cndmask / v_min
            if (MODE == 7) { a += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x111, 0xf, 0xf, false);
                             b += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xf, 0xf, false);
                             c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x114, 0xf, 0xf, false);
                             d += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0x118, 0xf, 0xf, false); }        // DPP adds
            if (MODE == 8) { a = __popc(a) + b; b = __popc(b) + c; c = __popc(c) + d; d = __popc(d) + a; }  // v_bcnt
            if (MODE == 9) { x = x * 0x9E3779B97F4A7C15ull + y; y = y * 0xC2B2AE3D27D4EB4Full + x; }        // 64-bit multiplies (mul_lo + mul_hi + mads)
            if (MODE == 10) { uint64_t m = __ballot(a & 1); a += (uint32_t)m; uint64_t n = __ballot(b & 2); b += (uint32_t)n;
                              uint64_t o = __ballot(c & 4); c += (uint32_t)o; uint64_t p = __ballot(d & 8); d += (uint32_t)p; }   // v_cmp + s_mov -> v
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((a ^ b ^ c ^ d ^ (uint32_t)x ^ (uint32_t)y) == 0x12345678u) out[0] = a;
    if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int MODE>
static int run(const char *name, int ops_per_body, int threads, int wg_per_cu, int ncu, uint32_t *d_out, unsigned long long *d_clk) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * wg_per_cu), dim3(threads), 0, 0, d_out, 1u, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(ncu * wg_per_cu), dim3(threads), 0, 0, d_out, 2u, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk = 0; CHECK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    const double instr_per_wave = (double)ITER * UNR * ops_per_body;
    const double waves_per_simd = (double)threads / 64 * wg_per_cu / 4;
    // s_memtime ticks at 100 MHz on this part; the kernel time in shader cycles from the event time at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s %4d thr x %d wg/CU: %.3f ms  -> %.2f shader cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, threads, wg_per_cu, ms,
           cyc / (instr_per_wave * waves_per_simd));
    return 0;
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.name, ncu, p.clockRate);
    uint32_t *d_out; unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_clk, 8));
    for (int occ = 0; occ < 2; occ++) {
        const int thr = occ == 0 ? 768 : 256, wg = occ == 0 ? 2 : 1;
        run<0>("v_add_u32", 4, thr, wg, ncu, d_out, d_clk);
        run<1>("v_xor_b32", 4, thr, wg, ncu, d_out, d_clk);
        run<2>("v_alignbit_b32", 4, thr, wg, ncu, d_out, d_clk);
        run<3>("v_mul_lo_u32", 3, thr, wg, ncu, d_out, d_clk);
        run<11>("v_mad_u32_u24", 3, thr, wg, ncu, d_out, d_clk);
        run<4>("64-bit shift + xor (x4)", 8, thr, wg, ncu, d_out, d_clk);
        run<5>("v_bfrev + xor", 8, thr, wg, ncu, d_out, d_clk);
        run<6>("compare-select / min", 8, thr, wg, ncu, d_out, d_clk);
        run<7>("DPP row_shr add", 4, thr, wg, ncu, d_out, d_clk);
        run<8>("v_bcnt + add", 4, thr, wg, ncu, d_out, d_clk);
        run<9>("64-bit multiply-add (x2)", 2, thr, wg, ncu, d_out, d_clk);
        run<10>("ballot + add (x4)", 8, thr, wg, ncu, d_out, d_clk);
    }
    return 0;
}
