// vmm_probe.hip - do hipMemcpy / hipMemset reach the right bytes of a hipMemMap'ped buffer when handed a pointer INSIDE the mapping?  (guard-mode allocator, api.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_copy(const uint32_t* a, uint32_t* b, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) b[i] = a[i]; }
__global__ void k_fill(uint32_t* a, size_t n, uint32_t s) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = (uint32_t)i * 3u + s; }
int main(int argc, char** argv) {
    const size_t lead = argc > 1 ? (size_t)atoll(argv[1]) : 1;        // granules of unmapped space in front of the mapping
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum)); CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu; %zu lead granules\n", gmin, grec, lead);
    const size_t gran = gmin, bytes = 100000, want = (bytes + 255) & ~(size_t)255, map_len = (want + gran - 1) / gran * gran, n = bytes / 4;
    void* va = nullptr; CK(hipMemAddressReserve(&va, map_len + (lead + 1) * gran, gran, nullptr, 0));
    char* map_at = (char*)va + lead * gran;
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, map_len, &prop, 0)); CK(hipMemMap(map_at, map_len, 0, h, 0));
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(map_at, map_len, &acc, 1));
    uint32_t* p = (uint32_t*)(map_at + (map_len - want));
    uint32_t* plain = nullptr; CK(hipMalloc(&plain, want));
    std::vector<uint32_t> hostv(n), back(n);
    for (size_t i = 0; i < n; i++) hostv[i] = (uint32_t)i * 7u + 1u;
    auto same = [&](const char* what, const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) { size_t bad = 0; for (size_t i = 0; i < n; i++) bad += a[i] != b[i]; printf("%-58s %s (%zu of %zu words differ)\n", what, bad ? "WRONG" : "ok", bad, n); };
    // 1. H2D into the interior pointer, read back through a kernel + plain memory
    CK(hipMemcpy(p, hostv.data(), bytes, hipMemcpyHostToDevice)); k_copy<<<(unsigned)((n + 255) / 256), 256>>>(p, plain, n); CK(hipMemcpy(back.data(), plain, bytes, hipMemcpyDeviceToHost)); same("hipMemcpy H2D -> interior pointer (seen by a kernel)", hostv, back);
    // 2. kernel fills, D2H straight from the interior pointer
    k_fill<<<(unsigned)((n + 255) / 256), 256>>>(p, n, 5u); CK(hipDeviceSynchronize()); CK(hipMemcpy(back.data(), p, bytes, hipMemcpyDeviceToHost));
    { std::vector<uint32_t> exp(n); for (size_t i = 0; i < n; i++) exp[i] = (uint32_t)i * 3u + 5u; same("hipMemcpy D2H <- interior pointer (filled by a kernel)", exp, back); }
    // 3. memset on the interior pointer
    CK(hipMemset(p, 0xAB, bytes)); k_copy<<<(unsigned)((n + 255) / 256), 256>>>(p, plain, n); CK(hipMemcpy(back.data(), plain, bytes, hipMemcpyDeviceToHost));
    { std::vector<uint32_t> exp(n, 0xABABABABu); same("hipMemset on the interior pointer (seen by a kernel)", exp, back); }
    // 4. D2D async both ways
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    k_fill<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(plain, n, 9u); CK(hipMemcpyAsync(p, plain, bytes, hipMemcpyDeviceToDevice, st)); CK(hipMemsetAsync(plain, 0, bytes, st));
    k_copy<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, plain, n); CK(hipMemcpyAsync(back.data(), plain, bytes, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
    { std::vector<uint32_t> exp(n); for (size_t i = 0; i < n; i++) exp[i] = (uint32_t)i * 3u + 9u; same("hipMemcpyAsync D2D plain -> interior pointer", exp, back); }
    // 5. the same four at the START of the mapping
    uint32_t* q = (uint32_t*)map_at;
    CK(hipMemcpy(q, hostv.data(), bytes, hipMemcpyHostToDevice)); k_copy<<<(unsigned)((n + 255) / 256), 256>>>(q, plain, n); CK(hipMemcpy(back.data(), plain, bytes, hipMemcpyDeviceToHost)); same("hipMemcpy H2D -> start of the mapping", hostv, back);
    // 6. small copies (other paths of the runtime): H2D then D2H of k bytes at odd offsets inside the buffer, sync and async, + small memsets
    for (size_t k : {4, 8, 24, 64, 1000, 4096, 16384, 65536}) {
        for (int async = 0; async < 2; async++) {
            const size_t off = 4 * 37;
            std::vector<uint8_t> a(k), b(k, 0);
            for (size_t i = 0; i < k; i++) a[i] = (uint8_t)(i * 13 + k + async);
            uint8_t* dp = (uint8_t*)p + off;
            if (async) { CK(hipMemcpyAsync(dp, a.data(), k, hipMemcpyHostToDevice, st)); CK(hipMemcpyAsync(b.data(), dp, k, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st)); }
            else { CK(hipMemcpy(dp, a.data(), k, hipMemcpyHostToDevice)); CK(hipMemcpy(b.data(), dp, k, hipMemcpyDeviceToHost)); }
            size_t bad = 0; for (size_t i = 0; i < k; i++) bad += a[i] != b[i];
            // and what a kernel sees there
            k_copy<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, plain, n); CK(hipStreamSynchronize(st));
            std::vector<uint8_t> c2(k); CK(hipMemcpy(c2.data(), (uint8_t*)plain + off, k, hipMemcpyDeviceToHost));
            size_t bad2 = 0; for (size_t i = 0; i < k; i++) bad2 += a[i] != c2[i];
            printf("%s H2D+D2H of %6zu bytes at +%zu: round trip %s, seen by a kernel %s\n", async ? "async" : "sync ", k, off, bad ? "WRONG" : "ok", bad2 ? "WRONG" : "ok");
        }
        CK(hipMemsetAsync((uint8_t*)p + 512, 0x5A, k, st)); k_copy<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(p, plain, n); CK(hipStreamSynchronize(st));
        std::vector<uint8_t> c3(k + 8); CK(hipMemcpy(c3.data(), (uint8_t*)plain + 512 - 4, k + 8, hipMemcpyDeviceToHost));
        size_t bad3 = 0; for (size_t i = 0; i < k; i++) bad3 += c3[i + 4] != 0x5A;
        printf("      memsetAsync of %6zu bytes: %s (neighbours %s)\n", k, bad3 ? "WRONG" : "ok", (c3[3] == 0x5A || c3[k + 4] == 0x5A) ? "TOUCHED" : "untouched");
    }
    printf("final sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
    return 0;
}
