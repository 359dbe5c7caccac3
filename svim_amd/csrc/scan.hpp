// scan.hpp - device-wide exclusive prefix sum, hand-written (used by prims.hip and bamdev.hip).
//   n <= SCAN_ONE:  ONE launch, one workgroup walks the tiles and carries the running total (small calls are latency, not bandwidth)
//   otherwise:      three launches - the sum of every tile of 2048 elements, the scan of those sums (one workgroup), the tiles rescanned with their offsets.
//                   The input is read twice; at the sizes of this path (a few million elements) that is tens of microseconds.
// in == out is allowed (a tile is read completely before it is written; the tile sums are taken before anything is written).
#pragma once
#include "common.hpp"

#define SCAN_T 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_T * SCAN_ITEMS)
#define SCAN_ONE (4 * SCAN_TILE)         /* = one trip of the 1024-thread form */

template <class T> __device__ __forceinline__ T scan_wave_incl(T v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T t = __shfl_up(v, o, 64);
        if (lane_id() >= o) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a workgroup of NT threads; *total = the sum.  sh: NT / 64 + 1 entries of LDS
template <class T, int NT = SCAN_T> __device__ __forceinline__ T scan_block_excl(T v, T* sh, T* total) {
    const int w = (int)(threadIdx.x >> 6);
    const T incl = scan_wave_incl(v);
    __syncthreads();                                              // (sh may still be read from the tile before)
    if (lane_id() == 63) sh[w] = incl;
    __syncthreads();
    T base = 0, sum = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) { const T x = sh[k]; if (k < w) base += x; sum += x; }
    *total = sum;
    return base + incl - v;
}

// one tile: thread t owns elements t * SCAN_ITEMS .. + SCAN_ITEMS - 1 (blocked: a thread's run is contiguous, its sum is one value of the workgroup scan)
template <class Tin, class T> __device__ __forceinline__ T scan_tile_load(const Tin* in, long long lo, long long n, T (&x)[SCAN_ITEMS]) {
    T s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const long long i = lo + (long long)threadIdx.x * SCAN_ITEMS + k;
        x[k] = i < n ? (T)in[i] : (T)0;
        s += x[k];
    }
    return s;
}
template <class T> __device__ __forceinline__ void scan_tile_store(T* out, long long lo, long long n, const T (&x)[SCAN_ITEMS], T start) {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const long long i = lo + (long long)threadIdx.x * SCAN_ITEMS + k;
        if (i < n) out[i] = start;
        start += x[k];
    }
}

// (1024 threads: 8192 elements per trip.  Up to 2048 elements - the tile sums of a scan over four million, every small scan of a CLUSTER call - the workgroup has
// 256 threads (round 6): a workgroup of sixteen waves needs a CU with four free wave slots on EVERY SIMD and waits for one while a full-chip kernel of a side
// stream is being dispatched (the packed haplotype store beside the partition phase: 0.04-0.4 ms per scan, profiles/r06_step_timeline_all_kernels.txt))
#define SCAN_ONE_T 1024
template <class Tin, class T, int NT> __global__ __launch_bounds__(NT) void k_scan_one(const Tin* in, T* out, long long n) {
    __shared__ T sh[NT / 64 + 1];
    T carry = 0;
    for (long long lo = 0; lo < n; lo += NT * SCAN_ITEMS) {
        T x[SCAN_ITEMS], total;
        const T mine = scan_tile_load<Tin, T>(in, lo, n, x);
        const T ex = scan_block_excl<T, NT>(mine, sh, &total);
        scan_tile_store<T>(out, lo, n, x, carry + ex);
        carry += total;
    }
}
template <class Tin, class T> __global__ __launch_bounds__(SCAN_T) void k_scan_tile_sums(const Tin* in, long long n, T* sums) {
    __shared__ T sh[SCAN_T / 64 + 1];
    T x[SCAN_ITEMS], total;
    const T mine = scan_tile_load<Tin, T>(in, (long long)blockIdx.x * SCAN_TILE, n, x);
    (void)scan_block_excl<T>(mine, sh, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
template <class Tin, class T> __global__ __launch_bounds__(SCAN_T) void k_scan_tiles(const Tin* in, T* out, long long n, const T* tile_start) {
    __shared__ T sh[SCAN_T / 64 + 1];
    T x[SCAN_ITEMS], total;
    const long long lo = (long long)blockIdx.x * SCAN_TILE;
    const T mine = scan_tile_load<Tin, T>(in, lo, n, x);
    const T ex = scan_block_excl<T>(mine, sh, &total);
    scan_tile_store<T>(out, lo, n, x, tile_start[blockIdx.x] + ex);
}

// out[i] = in[0] + ... + in[i - 1] for i in [0, n) on `stream`; tmp: scratch for the tile sums (grown as needed)
template <class Tin, class T> static int svx_exclusive_scan(const Tin* in, T* out, long long n, hipStream_t stream, DevBuf& tmp) {
    if (n <= 0) return SVX_OK;
    if (n <= SCAN_ONE) {
        if (n <= SCAN_TILE) k_scan_one<Tin, T, SCAN_T><<<1, SCAN_T, 0, stream>>>(in, out, n);
        else k_scan_one<Tin, T, SCAN_ONE_T><<<1, SCAN_ONE_T, 0, stream>>>(in, out, n);
        HIPCHK(hipGetLastError());
        return SVX_OK;
    }
    const long long tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    SVXCHK(tmp.reserve((size_t)tiles * sizeof(T) + 64));
    T* sums = tmp.as<T>();
    k_scan_tile_sums<Tin, T><<<(unsigned)tiles, SCAN_T, 0, stream>>>(in, n, sums);
    if (tiles <= SCAN_TILE) k_scan_one<T, T, SCAN_T><<<1, SCAN_T, 0, stream>>>(sums, sums, tiles);
    else k_scan_one<T, T, SCAN_ONE_T><<<1, SCAN_ONE_T, 0, stream>>>(sums, sums, tiles);
    k_scan_tiles<Tin, T><<<(unsigned)tiles, SCAN_T, 0, stream>>>(in, out, n, sums);
    HIPCHK(hipGetLastError());
    return SVX_OK;
}
