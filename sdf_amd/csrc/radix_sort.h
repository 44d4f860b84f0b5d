// radix_sort.h — hand-written LDS radix sort of (u32 key, u32 value) pairs + the exclusive scan it uses.
// Part of sdfhip.hip (one translation unit).
#pragma once

// =========================================================================================
// LDS radix sort of (u32 key, u32 value) pairs, 8 bits per pass, stable.
//   k_rs_hist   : per-tile digit histogram (LDS atomics) -> hist[digit][tile]
//   scan        : exclusive scan of hist (digit-major), hierarchical
//   k_rs_scatter: per 64-key wave chunk, ballot "match" masks give each key its stable rank
//                 among equal digits; chunk bases from per-wave running digit counts in LDS
// =========================================================================================
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 (12 / 16 / 20 / 24 items: 125M-key scatter 0.56 / 0.50 / 0.49 / 0.57 ms)
constexpr int RS_CHUNKS = RS_TILE / 64;         // 64

// element (k, t) of a tile = index base + k * 256 + t, so a wave reads 64 consecutive keys.
// A block takes RS_HT consecutive tiles: with one 12 KB tile per block the launch of 41k blocks (125M keys)
// was what bounded the kernel (0.18 ms for 0.5 GB).
constexpr int RS_HT = 4;
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n,
                                                        int shift, uint32_t* __restrict__ hist,
                                                        uint32_t ntiles) {
    __shared__ uint32_t h[RS_HT][256];
#pragma unroll
    for (int t = 0; t < RS_HT; ++t) h[t][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * RS_HT;
    uint32_t key[RS_HT][RS_ITEMS];
#pragma unroll
    for (int t = 0; t < RS_HT; ++t)
#pragma unroll
        for (int k = 0; k < RS_ITEMS; ++k) {  // all loads in flight before the first LDS atomic
            const uint32_t i = (tile0 + t) * RS_TILE + k * RS_THREADS + threadIdx.x;
            key[t][k] = keys[i < n ? i : n - 1];
        }
#pragma unroll
    for (int t = 0; t < RS_HT; ++t)
#pragma unroll
        for (int k = 0; k < RS_ITEMS; ++k) {
            const uint32_t i = (tile0 + t) * RS_TILE + k * RS_THREADS + threadIdx.x;
            if (i < n) atomicAdd(&h[t][(key[t][k] >> shift) & 255u], 1u);
        }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < RS_HT; ++t)
        if (tile0 + t < ntiles) hist[threadIdx.x * ntiles + tile0 + t] = h[t][threadIdx.x];
}

// block-level exclusive scan of 2048 entries per block; block totals to sums[]
__global__ __launch_bounds__(256) void k_scan_block(uint32_t* __restrict__ data, uint32_t n,
                                                    uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const uint32_t base = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t v[8], t = 0;
    for (int k = 0; k < 8; ++k) {
        v[k] = (base + k < n) ? data[base + k] : 0u;
        t += v[k];
    }
    // inclusive scan of t across the block
    uint32_t x = t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    uint32_t run = woff + x - t;
    for (int k = 0; k < 8; ++k) {
        if (base + k < n) data[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 255 && sums) sums[blockIdx.x] = woff + x;
}
__global__ void k_scan_add(uint32_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ sums) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] += sums[i / 2048u];
}

// in-place exclusive scan of n u32 (n <= 2048^3); s1/s2 hold ceil(n/2048) and ceil(n/2048^2) sums
static int exclusive_scan_u32(uint32_t* data, uint32_t n, uint32_t* s1, uint32_t* s2, hipStream_t s) {
    if (n == 0) return 0;
    const uint32_t n1 = (n + 2047) / 2048, n2 = (n1 + 2047) / 2048;
    k_scan_block<<<n1, 256, 0, s>>>(data, n, s1);
    if (n1 > 1) {
        k_scan_block<<<n2, 256, 0, s>>>(s1, n1, s2);
        if (n2 > 1) {
            k_scan_block<<<1, 256, 0, s>>>(s2, n2, nullptr);
            k_scan_add<<<(n1 + 255) / 256, 256, 0, s>>>(s1, n1, s2);
        }
        k_scan_add<<<(n + 255) / 256, 256, 0, s>>>(data, n, s1);
    }
    return 0;
}

// Stable scatter of one tile.  Wave w owns a quarter of the tile's elements as RS_ITEMS chunks of 64
// consecutive keys.  Ranks: a ballot "match" gives each key its rank among the equal digits of its chunk, and a
// per-wave running count per digit (LDS, [wave][digit]; a wave's chunks go through it in order, so no atomics and
// no zero-filled chunk x digit matrix as in the first version: 230 -> 110 LDS operations per thread) gives the
// chunk's base.  A scan over the 4 x 256 wave totals then places every (digit, wave) run, the tile is permuted
// into digit order inside LDS, and consecutive lanes write consecutive global addresses (runs of tile/256
// elements per digit instead of single keys).
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n, int shift,
    const uint32_t* __restrict__ hist, uint32_t ntiles, int vals_iota) {
    __shared__ uint32_t skey[RS_TILE];
    __shared__ uint32_t sval[RS_TILE];
    __shared__ uint32_t wcnt[4][256];          // per wave: running count of each digit, then the run's first slot
    __shared__ uint32_t bstart[256];           // first local slot of each digit
    __shared__ uint32_t gbase[256];            // global base of each digit for this tile
    __shared__ uint32_t wsum[4];
    // Blocks go round-robin over the 8 XCDs, each with its own L2.  A tile writes runs of ~12 elements
    // (48 B) per digit, and the next tile's run follows it in memory: with XCD x taking the x-th eighth of
    // the tiles, neighbouring runs are written through the same L2 at about the same time and leave it
    // as whole lines.  (grid = 8 * ceil(ntiles / 8) blocks.)
    const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;  // block-uniform
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) wcnt[k][threadIdx.x] = 0;
    const uint32_t base = tile * RS_TILE;
    const uint32_t wbase = base + (uint32_t)w * (RS_ITEMS * 64);
    uint32_t key[RS_ITEMS], val[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {  // all loads in flight before the first LDS round trip
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : 0xFFFFFFFFu;
        val[k] = valid ? (vals_iota ? i : vals_in[i]) : 0u;
    }
    __syncthreads();  // wcnt is zero
    uint16_t rank[RS_ITEMS];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        const bool valid = i < n;
        const uint32_t dg = (key[k] >> shift) & 255u;
        unsigned long long m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bl = __ballot((dg >> b) & 1u);
            m &= ((dg >> b) & 1u) ? bl : ~bl;
        }
        const uint32_t r = (uint32_t)__popcll(m & lt);
        const uint32_t before = wcnt[w][dg];  // (this wave's earlier chunks; its own writes, in program order)
        rank[k] = (uint16_t)(before + r);
        if (valid && r == 0) wcnt[w][dg] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();  // (the next chunk reads what other lanes of this wave wrote)
    }
    __syncthreads();
    {   // thread = digit: the four wave totals -> first slot of each (digit, wave) run; block scan of the digit totals
        const uint32_t dg = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t t = wcnt[k][dg];
            wcnt[k][dg] = run;
            run += t;
        }
        uint32_t x = run;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        bstart[dg] = woff + x - run;
        gbase[dg] = hist[dg * ntiles + tile];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t i = wbase + k * 64 + lane;
        if (i < n) {
            const uint32_t dg = (key[k] >> shift) & 255u;
            const uint32_t slot = bstart[dg] + wcnt[w][dg] + rank[k];
            skey[slot] = key[k];
            sval[slot] = val[k];
        }
    }
    __syncthreads();
    const uint32_t tile_n = (n - base) < (uint32_t)RS_TILE ? (n - base) : (uint32_t)RS_TILE;
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const uint32_t slot = k * RS_THREADS + threadIdx.x;
        if (slot < tile_n) {
            const uint32_t kk = skey[slot];
            const uint32_t dg = (kk >> shift) & 255u;
            const uint32_t dst = gbase[dg] + (slot - bstart[dg]);
            if (keys_out) keys_out[dst] = kk;  // (the last pass of a sort whose keys nobody reads)
            vals_out[dst] = sval[slot];
        }
    }
}

struct RadixSorter {
    uint32_t* keys_alt = nullptr;
    uint32_t* vals_alt = nullptr;
    uint32_t* hist = nullptr;
    uint32_t* sums1 = nullptr;
    uint32_t* sums2 = nullptr;
    uint32_t cap = 0;

    int reserve(uint32_t n) {
        if (n <= cap) return 0;
        release();
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const uint32_t nh = 256u * ntiles;
        const uint32_t n1 = (nh + 2047) / 2048, n2 = (n1 + 2047) / 2048;
        HIPC(hipMalloc(&keys_alt, sizeof(uint32_t) * (size_t)n));
        HIPC(hipMalloc(&vals_alt, sizeof(uint32_t) * (size_t)n));
        HIPC(hipMalloc(&hist, sizeof(uint32_t) * (size_t)nh));
        HIPC(hipMalloc(&sums1, sizeof(uint32_t) * (size_t)(n1 + 1)));
        HIPC(hipMalloc(&sums2, sizeof(uint32_t) * (size_t)(n2 + 1)));
        cap = n;
        return 0;
    }
    void release() {
        void* ps[] = {keys_alt, vals_alt, hist, sums1, sums2};
        for (void* p : ps)
            if (p) (void)hipFree(p);
        keys_alt = vals_alt = hist = sums1 = sums2 = nullptr;
        cap = 0;
    }
    // sorts (keys, vals) ascending by the low `bits` bits of keys (8 bits per pass).  The result
    // is in *keys_out / *vals_out: the caller's buffers after an even number of passes, the
    // sorter's alternates after an odd number.  iota = 1: input values are taken as 0..n-1.
    // keys_wanted = false: only the permutation is needed; the last pass then writes no keys.
    // hist0_done: the caller's key kernel has filled `hist` for the first pass (k_query_morton does).
    int sort(uint32_t* keys, uint32_t* vals, uint32_t n, int bits, int iota, hipStream_t s,
             uint32_t** keys_out, uint32_t** vals_out, bool keys_wanted = true, bool hist0_done = false) {
        if (keys_out) *keys_out = keys;
        if (vals_out) *vals_out = vals;
        if (n == 0) return 0;
        if (int rc = reserve(n)) return rc;
        const int passes = (bits + 7) / 8;
        const uint32_t ntiles = (n + RS_TILE - 1) / RS_TILE;
        const uint32_t nh = 256u * ntiles;
        const uint32_t n1 = (nh + 2047) / 2048, n2 = (n1 + 2047) / 2048;
        if (n2 > 2048) return fail(SDFHIP_E_INVALID, "radix sort: too many keys (%u)", n);
        uint32_t *ki = keys, *vi = vals, *ko = keys_alt, *vo = vals_alt;
        for (int p = 0; p < passes; ++p) {
            const int shift = 8 * p;
            if (p > 0 || !hist0_done)
                k_rs_hist<<<(ntiles + RS_HT - 1) / RS_HT, RS_THREADS, 0, s>>>(ki, n, shift, hist, ntiles);
            exclusive_scan_u32(hist, nh, sums1, sums2, s);
            k_rs_scatter<<<8u * ((ntiles + 7u) / 8u), RS_THREADS, 0, s>>>(ki, vi, (p == passes - 1 && !keys_wanted) ? nullptr : ko, vo, n,
                                                       shift, hist, ntiles, (p == 0 && iota) ? 1 : 0);
            uint32_t* t = ki;
            ki = ko, ko = t;
            t = vi;
            vi = vo, vo = t;
        }
        if (keys_out) *keys_out = ki;
        if (vals_out) *vals_out = vi;
        HIPC(hipGetLastError());
        return 0;
    }
};
