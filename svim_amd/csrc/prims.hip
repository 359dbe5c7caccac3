// prims.hip - device-wide sort / scan used between the stages: hand-written like the stages themselves (rounds 1-2 borrowed rocPRIM here).
//
// svx_sort_pairs_u64: stable LSD radix sort of (64-bit key, 32-bit value) pairs over the key bits [begin_bit, end_bit), 8 bits per pass.
//   A pass is three launches over tiles of 2048 pairs (256 threads x 8) - no workgroup ever waits for another one:
//     k_radix_hist     digit histogram of every tile (LDS atomics) -> cnt[digit][tile], and the digit totals of the pass
//     k_radix_offsets  one workgroup per digit: start of the digit (sum of the totals below it) + running sum over the tiles -> off[digit][tile]
//     k_radix_scatter  rank of every pair among the pairs of its digit inside its tile, kept STABLE: a wave takes its 512 pairs in 8 rounds of 64
//                      consecutive ones; in a round the lanes with the same digit find each other with 8 ballots (one per digit bit), their rank is
//                      the population count of the peers below them, the running count per digit lives in a per-wave LDS table (one lane per digit
//                      group updates it: no atomics); digit counts of the four waves are prefix-summed by the 256 threads = 256 digits
//   Up to RADIX_ONE (16384) pairs everything runs in ONE launch of one workgroup of 1024 threads (all passes; small calls are latency-bound), and
//   passes over a digit that is the same in every key are skipped there.  (A one-launch-per-pass form with decoupled look-back between the tiles was
//   built and measured: 24 us per pass of 0.74 M pairs against 28 us for the three launches - not worth workgroups that spin on each other.  Taking the next pass's histogram inside the scatter with one global atomic per pair: 0.4 ms per
//   pass - the zero bytes of narrow keys send thousands of atomics to one address.)
// svx_exclusive_scan_*: scan.hpp.
#include "common.hpp"
#include "scan.hpp"
#include "hostcopy.hpp"
#include <algorithm>
#include <vector>

#define RADIX_T 256
#define RADIX_ITEMS 8
#define RADIX_TILE (RADIX_T * RADIX_ITEMS)
#define RADIX_ONE_T 1024                       /* the one-workgroup form: 16 waves, 8192 pairs per trip */
#define RADIX_ONE (2 * RADIX_ONE_T * RADIX_ITEMS)

__device__ __forceinline__ unsigned radix_digit(uint64_t key, int shift, unsigned mask) { return (unsigned)(key >> shift) & mask; }

// ranks of the tile's pairs (see above).  keys[r] of a lane = pair  tile_lo + wave * 512 + r * 64 + lane  (RADIX_SENTINEL digit 256 = beyond n).
// wh: [4][256] running counts per wave (zeroed here), afterwards wh[w][d] = number of pairs with digit d in wave w.  rank[r] = position of the pair
// among the pairs of its digit in its WAVE, in input order.
__device__ __forceinline__ void radix_rank_tile(const unsigned (&dig)[RADIX_ITEMS], unsigned (*wh_)[256], unsigned (&rank)[RADIX_ITEMS]) {
    const int w = (int)(threadIdx.x >> 6), lane = lane_id();
    volatile unsigned* wh = wh_[w];                                 // the wave's own table: lanes hand values to each other through it from round to round (LDS
                                                                   // operations of a wave execute in order; volatile + the wave barrier keep the compiler from caching)
    for (int d = lane; d < 256; d += 64) wh[d] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        const unsigned d = dig[r];
        const bool live = d < 256u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned below = (unsigned)__popcll(peers & lanemask_lt()), cnt = (unsigned)__popcll(peers);
        const int leader = peers ? __ffsll((long long)peers) - 1 : 0;
        unsigned old = 0;
        if (live && lane == leader) { old = wh[d]; wh[d] = old + cnt; }
        old = (unsigned)__shfl((int)old, leader, 64);
        rank[r] = old + below;
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(RADIX_T) void k_radix_hist(const uint64_t* keys, long long n, int shift, unsigned mask, long long tiles, unsigned* cnt, unsigned* digit_total) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long lo = (long long)blockIdx.x * RADIX_TILE;
#pragma unroll
    for (int k = 0; k < RADIX_ITEMS; k++) {
        const long long i = lo + (long long)k * RADIX_T + threadIdx.x;
        if (i < n) atomicAdd(&h[radix_digit(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    const unsigned v = h[threadIdx.x];
    cnt[(long long)threadIdx.x * tiles + blockIdx.x] = v;
    if (v) atomicAdd(digit_total + threadIdx.x, v);
}

// workgroup d: off[d][t] = (pairs with a smaller digit) + (pairs with digit d in the tiles before t)
__global__ __launch_bounds__(256) void k_radix_offsets(const unsigned* cnt, const unsigned* digit_total, long long tiles, unsigned* off) {
    __shared__ unsigned sh[256 / 64 + 1];
    const int d = (int)blockIdx.x;
    unsigned total;
    unsigned carry;
    {
        const unsigned mine = (int)threadIdx.x < d ? digit_total[threadIdx.x] : 0u;
        (void)scan_block_excl<unsigned>(mine, sh, &total);
        carry = total;
    }
    const unsigned* c = cnt + (long long)d * tiles;
    unsigned* o = off + (long long)d * tiles;
    for (long long lo = 0; lo < tiles; lo += 256) {
        const long long t = lo + threadIdx.x;
        const unsigned mine = t < tiles ? c[t] : 0u;
        const unsigned ex = scan_block_excl<unsigned>(mine, sh, &total);
        if (t < tiles) o[t] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(RADIX_T) void k_radix_scatter(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, long long n, int shift,
                                                           unsigned mask, long long tiles, const unsigned* off) {
    __shared__ unsigned wh[RADIX_T / 64][256];
    __shared__ unsigned base[RADIX_T / 64][256];
    const int w = (int)(threadIdx.x >> 6), lane = lane_id();
    const long long lo = (long long)blockIdx.x * RADIX_TILE + (long long)w * (64 * RADIX_ITEMS);
    uint64_t key[RADIX_ITEMS]; uint32_t val[RADIX_ITEMS]; unsigned dig[RADIX_ITEMS], rank[RADIX_ITEMS];
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        const long long i = lo + (long long)r * 64 + lane;
        key[r] = i < n ? keys_in[i] : 0ull;
        val[r] = i < n ? vals_in[i] : 0u;
        dig[r] = i < n ? radix_digit(key[r], shift, mask) : 256u;
    }
    radix_rank_tile(dig, wh, rank);
    __syncthreads();
    {
        // thread d: where the pairs of digit d of wave 0, 1, 2, 3 go
        const int d = (int)threadIdx.x;
        unsigned at = off[(long long)d * tiles + blockIdx.x];
#pragma unroll
        for (int k = 0; k < RADIX_T / 64; k++) { base[k][d] = at; at += wh[k][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RADIX_ITEMS; r++) {
        if (dig[r] < 256u) {
            const unsigned p = base[w][dig[r]] + rank[r];
            keys_out[p] = key[r];
            vals_out[p] = val[r];
        }
    }
}

// n <= RADIX_ONE: every pass in one workgroup of 1024 threads, ping-pong between the scratch pair and the caller's output (the last pass lands in the output).
// A pass whose digit is the same for every pair (zero bytes of narrow keys) moves nothing and is skipped.
__global__ __launch_bounds__(RADIX_ONE_T) void k_radix_one(const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, uint64_t* ktmp, uint32_t* vtmp,
                                                           long long n, int begin_bit, int end_bit) {
    __shared__ unsigned wh[RADIX_ONE_T / 64][256];
    __shared__ unsigned base[RADIX_ONE_T / 64][256];
    __shared__ unsigned tot[256];
    __shared__ unsigned sh[RADIX_ONE_T / 64 + 1];
    __shared__ unsigned n_used;
    const int w = (int)(threadIdx.x >> 6), lane = lane_id();
    const int passes = (end_bit - begin_bit + 7) / 8;
    const long long trip = (long long)RADIX_ONE_T * RADIX_ITEMS;
    const uint64_t* ks = keys_in; const uint32_t* vs = vals_in;
    uint64_t* kd = keys_out; uint32_t* vd = vals_out;             // destination of the next pass that moves something; the other buffer of the ping-pong is the scratch pair
    for (int p = 0; p < passes; p++) {
        const int shift = begin_bit + 8 * p;
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const unsigned mask = (1u << bits) - 1u;
        if (threadIdx.x < 256) tot[threadIdx.x] = 0;
        if (threadIdx.x == 0) n_used = 0;
        __syncthreads();
        for (long long i = threadIdx.x; i < n; i += RADIX_ONE_T) atomicAdd(&tot[radix_digit(ks[i], shift, mask)], 1u);
        __syncthreads();
        if (threadIdx.x < 256 && tot[threadIdx.x]) atomicAdd(&n_used, 1u);
        unsigned total;
        const unsigned start = scan_block_excl<unsigned, RADIX_ONE_T>(threadIdx.x < 256 ? tot[threadIdx.x] : 0u, sh, &total);
        __syncthreads();
        if (n_used <= 1u) continue;                                // (uniform: every thread reads the same LDS word after the barrier)
        if (threadIdx.x < 256) tot[threadIdx.x] = start;          // from here on: where the next pair of digit d goes (advanced trip by trip)
        __syncthreads();
        for (long long t0 = 0; t0 < n; t0 += trip) {
            const long long lo = t0 + (long long)w * (64 * RADIX_ITEMS);
            uint64_t key[RADIX_ITEMS]; uint32_t val[RADIX_ITEMS]; unsigned dig[RADIX_ITEMS], rank[RADIX_ITEMS];
#pragma unroll
            for (int r = 0; r < RADIX_ITEMS; r++) {
                const long long i = lo + (long long)r * 64 + lane;
                key[r] = i < n ? ks[i] : 0ull;
                val[r] = i < n ? vs[i] : 0u;
                dig[r] = i < n ? radix_digit(key[r], shift, mask) : 256u;
            }
            radix_rank_tile(dig, wh, rank);
            __syncthreads();
            if (threadIdx.x < 256) {
                const int d = (int)threadIdx.x;
                unsigned at = tot[d];
#pragma unroll
                for (int k = 0; k < RADIX_ONE_T / 64; k++) { base[k][d] = at; at += wh[k][d]; }
                tot[d] = at;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RADIX_ITEMS; r++) {
                if (dig[r] < 256u) {
                    const unsigned q = base[w][dig[r]] + rank[r];
                    kd[q] = key[r];
                    vd[q] = val[r];
                }
            }
            __syncthreads();
        }
        __threadfence();                                           // the next pass reads what this one wrote (same workgroup, other threads)
        __syncthreads();
        ks = kd; vs = vd;
        kd = kd == keys_out ? ktmp : keys_out; vd = vd == vals_out ? vtmp : vals_out;
    }
    if (ks != keys_out)                                            // the data sits in the input or in the scratch pair: the caller wants it in its output
        for (long long i = threadIdx.x; i < n; i += RADIX_ONE_T) { keys_out[i] = ks[i]; vals_out[i] = vs[i]; }
}

int svx_sort_pairs_u64(svx_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                       int64_t n, int begin_bit, int end_bit) {
    if (n <= 0) return SVX_OK;
    if (n >= (1ll << 32) || begin_bit < 0 || end_bit > 64 || begin_bit >= end_bit) return svx_fail(SVX_E_ARG, "svx_sort_pairs_u64: bad arguments", __FILE__, __LINE__, hipSuccess);
    hipStream_t st = c->stream;
    const int passes = (end_bit - begin_bit + 7) / 8;
    const long long tiles = (n + RADIX_TILE - 1) / RADIX_TILE;
    // scratch: the second buffer of the ping-pong, and for the tiled path cnt / off [256][tiles] + the digit totals of every pass
    const size_t pair_bytes = ((size_t)n * 8 + 255) / 256 * 256, val_bytes = ((size_t)n * 4 + 255) / 256 * 256;
    const size_t table = ((size_t)256 * (size_t)tiles * 4 + 255) / 256 * 256, totals = (size_t)passes * 256 * 4;
    SVXCHK(c->sort_tmp.reserve(pair_bytes + val_bytes + 2 * table + totals + 256));
    uint8_t* base = c->sort_tmp.as<uint8_t>();
    uint64_t* ktmp = reinterpret_cast<uint64_t*>(base);
    uint32_t* vtmp = reinterpret_cast<uint32_t*>(base + pair_bytes);
    if (n <= RADIX_ONE) {
        k_radix_one<<<1, RADIX_ONE_T, 0, st>>>(keys_in, vals_in, keys_out, vals_out, ktmp, vtmp, n, begin_bit, end_bit);
        HIPCHK(hipGetLastError());
        return SVX_OK;
    }
    unsigned* cnt = reinterpret_cast<unsigned*>(base + pair_bytes + val_bytes);
    unsigned* off = reinterpret_cast<unsigned*>(base + pair_bytes + val_bytes + table);
    unsigned* digit_total = reinterpret_cast<unsigned*>(base + pair_bytes + val_bytes + 2 * table);
    HIPCHK(hipMemsetAsync(digit_total, 0, totals, st));
    const uint64_t* ks = keys_in; const uint32_t* vs = vals_in;
    for (int p = 0; p < passes; p++) {
        const int shift = begin_bit + 8 * p;
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const unsigned mask = (1u << bits) - 1u;
        uint64_t* kd = ((passes - 1 - p) & 1) ? ktmp : keys_out;           // the last pass lands in the caller's output
        uint32_t* vd = ((passes - 1 - p) & 1) ? vtmp : vals_out;
        k_radix_hist<<<(unsigned)tiles, RADIX_T, 0, st>>>(ks, n, shift, mask, tiles, cnt, digit_total + 256 * p);
        k_radix_offsets<<<256, 256, 0, st>>>(cnt, digit_total + 256 * p, tiles, off);
        k_radix_scatter<<<(unsigned)tiles, RADIX_T, 0, st>>>(ks, vs, kd, vd, n, shift, mask, tiles, off);
        ks = kd; vs = vd;
    }
    HIPCHK(hipGetLastError());
    return SVX_OK;
}

int svx_exclusive_scan_i64(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n) {
    return svx_exclusive_scan<int64_t, int64_t>(in, out, n, c->stream, c->scan_tmp);
}
// the same on another stream with its own temporary storage (concurrent with the main stream's primitives)
int svx_exclusive_scan_i64_on(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n, hipStream_t stream, DevBuf& tmp) {
    (void)c;
    return svx_exclusive_scan<int64_t, int64_t>(in, out, n, stream, tmp);
}
int svx_exclusive_scan_i32_to_i64(svx_ctx* c, const int32_t* in, int64_t* out, int64_t n) {
    return svx_exclusive_scan<int32_t, int64_t>(in, out, n, c->stream, c->scan_tmp);
}

// self-test of the two primitives on pseudo-random data of the given shape, checked on the host (tests/test_gpu_parity.py): 0 = identical
extern "C" int svx_selftest_prims(svx_ctx* c, int64_t n, int32_t begin_bit, int32_t end_bit, uint64_t seed) {
    if (!c || n < 0) return svx_fail(SVX_E_ARG, "bad argument", __FILE__, __LINE__, hipSuccess);
    HIPCHK(hipSetDevice(c->device));
    std::vector<uint64_t> k((size_t)n), k2((size_t)n);
    std::vector<uint32_t> v((size_t)n), v2((size_t)n);
    std::vector<int64_t> s((size_t)n + 1), s2((size_t)n + 1);
    uint64_t x = seed | 1ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (int64_t i = 0; i < n; i++) {
        const uint64_t r = rnd();
        k[(size_t)i] = (r & 1) ? r : (r >> 40);                     // a mix of wide and narrow keys: many equal digits, many equal keys
        if ((r & 6) == 0) k[(size_t)i] &= 0xffull << begin_bit;
        if (seed & 0x100) k[(size_t)i] &= 0xffff00ull;               // narrow keys everywhere: whole passes see one digit only
        v[(size_t)i] = (uint32_t)i;
        s[(size_t)i] = (int64_t)(rnd() % 1000) - 100;
    }
    s[(size_t)n] = 0;
    DevBuf dk, dk2, dv, dv2, ds, ds2;
    auto done = [&](int r) { dk.release(); dk2.release(); dv.release(); dv2.release(); ds.release(); ds2.release(); return r; };
    auto on_device = [&]() -> int {                                // (HIPCHK / SVXCHK return from here; the buffers are released by the caller below)
        SVXCHK(dk.reserve((size_t)n * 8 + 8)); SVXCHK(dk2.reserve((size_t)n * 8 + 8)); SVXCHK(dv.reserve((size_t)n * 4 + 8)); SVXCHK(dv2.reserve((size_t)n * 4 + 8));
        SVXCHK(ds.reserve((size_t)(n + 1) * 8)); SVXCHK(ds2.reserve((size_t)(n + 1) * 8));
        if (n) {
            // std::vectors of up to 26 MB that die with this call: through the library's pinned bounce buffers like every other pageable array (hostcopy.hpp) -
            // a direct hipMemcpy would page-lock them in place and leave a registration of freed memory behind
            SVXCHK(svx_h2d(dk.p, k.data(), (size_t)n * 8, c->stream));
            SVXCHK(svx_h2d(dv.p, v.data(), (size_t)n * 4, c->stream));
        }
        SVXCHK(svx_h2d(ds.p, s.data(), (size_t)(n + 1) * 8, c->stream));
        SVXCHK(svx_sort_pairs_u64(c, dk.as<uint64_t>(), dk2.as<uint64_t>(), dv.as<uint32_t>(), dv2.as<uint32_t>(), n, begin_bit, end_bit));
        SVXCHK(svx_exclusive_scan_i64(c, ds.as<int64_t>(), ds2.as<int64_t>(), n + 1));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (n) {
            SVXCHK(svx_d2h(k2.data(), dk2.p, (size_t)n * 8, c->stream));
            SVXCHK(svx_d2h(v2.data(), dv2.p, (size_t)n * 4, c->stream));
        }
        SVXCHK(svx_d2h(s2.data(), ds2.p, (size_t)(n + 1) * 8, c->stream));
        return SVX_OK;
    };
    const int rc = on_device();
    if (rc != SVX_OK) return done(rc);
    const uint64_t m = (end_bit - begin_bit >= 64 ? ~0ull : ((1ull << (end_bit - begin_bit)) - 1ull)) << begin_bit;
    std::vector<uint32_t> order((size_t)n);
    for (int64_t i = 0; i < n; i++) order[(size_t)i] = (uint32_t)i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (k[a] & m) < (k[b] & m); });
    for (int64_t i = 0; i < n; i++)
        if (v2[(size_t)i] != order[(size_t)i] || k2[(size_t)i] != k[order[(size_t)i]]) return done(svx_fail(SVX_E_STATE, "radix sort differs from std::stable_sort", __FILE__, __LINE__, hipSuccess));
    int64_t run = 0;
    for (int64_t i = 0; i <= n; i++) {
        if (s2[(size_t)i] != run) return done(svx_fail(SVX_E_STATE, "exclusive scan differs from the serial sum", __FILE__, __LINE__, hipSuccess));
        run += s[(size_t)i];
    }
    return done(SVX_OK);
}

// loads this translation unit's code object (HIP does it lazily, at the first launch): called by svx_ctx_create so that the first COLLECT / CLUSTER call
// of a context does not pay for it
void svx_preload_prims() { hipFuncAttributes a; (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_radix_offsets)); (void)hipGetLastError(); }
