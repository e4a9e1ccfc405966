// rfx_device.h -- device-side helpers (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RFX_WAVE 64

namespace rfxd {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (RFX_WAVE - 1); }

// ~((~0L) << 2*bases), bases <= 31
__device__ __host__ __forceinline__ uint64_t low_mask(int bases) { return ~((~0ULL) << (2 * bases)); }

// Bijective 64-bit mix (murmur3 finalizer): top bits pick radix buckets / the owning GPU.
__device__ __host__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

// Bijective, one 64-bit multiply: the product's high half is well mixed, folding it into the
// low half mixes that too.  Its top OWNER_BITS pick the owning GPU, the bits below feed the
// local radix levels and the leaf hash table (rfx_kmer.hip).  Result correctness never depends
// on hash quality (bucket sizes are exact histograms); only load balance does.
constexpr int OWNER_BITS = 6;
__device__ __host__ __forceinline__ uint64_t kmer_hash(uint64_t x) {
    uint64_t h = x * 0x9E3779B97F4A7C15ULL;
    return h ^ (h >> 32);
}

// swap the two bits of every base pair
// (on the two 32-bit halves: a pair never straddles them, and 32-bit shifts issue at full rate where the 64-bit ones do not)
__device__ __forceinline__ uint32_t pair_swap32(uint32_t x) {
    return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u);
}
__device__ __forceinline__ uint64_t pair_swap(uint64_t x) {
    return ((uint64_t)pair_swap32((uint32_t)(x >> 32)) << 32) | pair_swap32((uint32_t)x);
}

// reverse complement of a right-aligned k-mer (k <= 32): complement = ~, reversal of base
// order = bit reversal + swap inside each pair.  Equals the k-iteration loop of
// KmerReverseComplement.call (P/ReflexivMain.java:2916-2923).
__device__ __forceinline__ uint64_t revcomp(uint64_t kmer, int k) {
    uint64_t r = pair_swap(__brevll(~kmer));       // complement-reverse of all 32 pairs
    return r >> (64 - 2 * k);
}

// k-mer starting at base p of a packed read (32 bases per word, first base in the top
// pair).  w points at the read's first word.
__device__ __forceinline__ uint64_t kmer_at(const uint64_t *__restrict__ w, int p, int k) {
    int wi = p >> 5, sh = 2 * (p & 31);
    uint64_t x = w[wi] << sh;
    if (sh + 2 * k > 64) x |= w[wi + 1] >> (64 - sh);     // sh > 0 here
    return x >> (64 - 2 * k);
}

// canonical form: ReverseComplementKmerBinaryExtraction.call keeps fwd when
// fwd < rc else rc (P/ReflexivMain.java:3051-3055).
__device__ __forceinline__ uint64_t canonical(uint64_t fwd, int k) {
    uint64_t rc = revcomp(fwd, k);
    return fwd < rc ? fwd : rc;
}

// Java: Long.SIZE/2 - (Long.numberOfLeadingZeros(w)/2 + 1); nlz(0) = 64
__device__ __forceinline__ int sentinel_len(uint64_t w) {
    int nlz = w ? __clzll(w) : 64;
    return 32 - (nlz / 2 + 1);
}

// base q of an extension stored in the reference's word layout
__device__ __forceinline__ unsigned ext_base(const uint64_t *__restrict__ w, int f, int64_t q) {
    if (q < f) return (unsigned)((w[0] >> (2 * (f - 1 - (int)q))) & 3);
    int64_t r = q - f;
    int64_t wi = 1 + r / 31;
    int j = (int)(r % 31);
    return (unsigned)((w[wi] >> (2 * (30 - j))) & 3);
}

__device__ __forceinline__ unsigned key_base(uint64_t key, int sub, int i) {
    return (unsigned)((key >> (2 * (sub - 1 - i))) & 3);
}

// ---- (k-1)-mer keys of KW words (k > 32: P/ReflexivDSMain64.java, 31 bases per word, the last word the
// remaining (k-2)%31+1 bases right-aligned, U/DefaultParam.java:93-94; KW = 1 is the single long of k <= 32).
// Stored AoS, KW consecutive words per record, as the reference's Row(long[]).
template <int KW> struct alignas(KW % 2 == 0 ? 16 : 8) KeyW { uint64_t w[KW]; };

template <int KW> __device__ __forceinline__ bool key_eq(const KeyW<KW> &a, const KeyW<KW> &b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < KW; i++) e = e && (a.w[i] == b.w[i]);
    return e;
}

// base i (0 = first) of a key of `sub` bases
template <int KW> __device__ __forceinline__ unsigned key_base_w(const KeyW<KW> &key, int sub, int i) {
    if (KW == 1) return (unsigned)((key.w[0] >> (2 * (sub - 1 - i))) & 3);
    const int wi = i / 31, j = i - 31 * wi;
    const int nb = wi < KW - 1 ? 31 : sub - 31 * (KW - 1);
    uint64_t x = key.w[0];
#pragma unroll
    for (int q = 1; q < KW; q++) if (wi == q) x = key.w[q];
    return (unsigned)((x >> (2 * (nb - 1 - j))) & 3);
}

// key from a base function f(t), t = 0 .. sub-1
template <int KW, class F> __device__ __forceinline__ KeyW<KW> build_key(int sub, F f) {
    KeyW<KW> key;
    int t = 0;
#pragma unroll
    for (int w = 0; w < KW; w++) {
        const int nb = w < KW - 1 ? 31 : sub - 31 * (KW - 1);
        uint64_t x = 0;
        for (int j = 0; j < nb; j++) x = (x << 2) | (uint64_t)f(t++);
        key.w[w] = x;
    }
    return key;
}

__device__ __host__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// block-wide exclusive scan of one uint32 per thread (blockDim.x multiple of 64, <= 1024)
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds_wave_sums,
                                                         uint32_t *total) {
    int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o, 64);
        if (lane >= o) x += y;
    }
    if (lane == 63) lds_wave_sums[wave] = x;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < nw; i++) {
        uint32_t s = lds_wave_sums[i];
        if (i < wave) base += s;
        tot += s;
    }
    __syncthreads();
    if (total) *total = tot;
    return base + x - v;
}

}  // namespace rfxd
