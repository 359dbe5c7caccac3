// What the BAM file -> HBM path of the device reader is made of, measured apart: (1) page cache (memory-mapped file) -> page-locked buffer with 1..16 copy threads,
// (2) page-locked buffer -> HBM by the copy engine (hipMemcpyAsync), one and two streams, (3) a kernel reading the page-locked buffer in place (16 B per lane).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/h2d_probe.bin tools/micro/h2d_probe.hip -lpthread      Run: h2d_probe.bin [GB = 2]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <fcntl.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_read(const uint4* p, size_t n, uint4* out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 2.0) * (1ull << 30);
    const char* path = "/tmp/h2d_probe.dat";
    {   // a file in the page cache
        int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
        std::vector<char> buf(64 << 20);
        for (size_t i = 0; i < buf.size(); i++) buf[i] = (char)(i * 2654435761u >> 13);
        for (size_t done = 0; done < bytes; done += buf.size()) if (write(fd, buf.data(), buf.size()) < 0) { perror("write"); return 2; }
        close(fd);
    }
    int fd = open(path, O_RDONLY);
    const char* map = (const char*)mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    if (map == MAP_FAILED) { perror("mmap"); return 2; }
    char* pin; CHK(hipHostMalloc((void**)&pin, bytes, hipHostMallocDefault));
    memset(pin, 1, bytes);
    char* dev; CHK(hipMalloc((void**)&dev, bytes));
    for (int T : {1, 2, 4, 8, 12, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([=]() { const size_t lo = bytes * t / T, hi = bytes * (t + 1) / T; memcpy(pin + lo, map + lo, hi - lo); });
            for (auto& x : th) x.join();
            const double dt = now() - t0; if (dt < best) best = dt;
        }
        printf("page cache -> page-locked, %2d threads: %.1f GB/s\n", T, bytes / best / 1e9);
    }
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CHK(hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s1)); CHK(hipStreamSynchronize(s1));
        const double one = now() - t0;
        t0 = now();
        CHK(hipMemcpyAsync(dev, pin, bytes / 2, hipMemcpyHostToDevice, s1)); CHK(hipMemcpyAsync(dev + bytes / 2, pin + bytes / 2, bytes / 2, hipMemcpyHostToDevice, s2));
        CHK(hipStreamSynchronize(s1)); CHK(hipStreamSynchronize(s2));
        const double two = now() - t0;
        printf("page-locked -> HBM (copy engine): one stream %.1f GB/s, two streams %.1f GB/s\n", bytes / one / 1e9, bytes / two / 1e9);
    }
    void* dp; CHK(hipHostGetDevicePointer(&dp, pin, 0));
    uint4* out; CHK(hipMalloc((void**)&out, 64));
    for (int blocks : {256, 1024, 4096}) {
        hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipEventRecord(e0, s1)); k_read<<<blocks, 256, 0, s1>>>((const uint4*)dp, bytes / 16, out); CHK(hipEventRecord(e1, s1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("kernel reading page-locked memory in place, %d blocks of 256: %.1f GB/s\n", blocks, bytes / ms / 1e6);
    }
    {   // copy threads and the copy engine at once, in 64 MiB pieces (what the reader does)
        const size_t piece = 64ull << 20; const int T = 8;
        const double t0 = now();
        for (size_t off = 0; off < bytes; off += piece) {
            const size_t n = bytes - off < piece ? bytes - off : piece;
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([=]() { const size_t lo = n * t / T, hi = n * (t + 1) / T; memcpy(pin + off + lo, map + off + lo, hi - lo); });
            for (auto& x : th) x.join();
            CHK(hipMemcpyAsync(dev + off, pin + off, n, hipMemcpyHostToDevice, s1));
        }
        CHK(hipStreamSynchronize(s1));
        printf("page cache -> page-locked (8 threads) -> HBM, pipelined in 64 MiB pieces: %.1f GB/s\n", bytes / (now() - t0) / 1e9);
    }
    unlink(path);
    return 0;
}
