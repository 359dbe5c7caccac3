// malloc_cost.hip - what a context's first call pays for its device buffers: is hipMalloc priced per call or per byte?
// (bench.py first_step_ms: 123 hipMalloc calls, 3.0 GB, 16 ms with SVX_ALLOC_STATS=1.)   hipcc --offload-arch=gfx950 -O2 malloc_cost.hip -o malloc_cost.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(char* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * 4096 < n; i += (size_t)gridDim.x * blockDim.x) p[i * 4096] = 1; }
static void run(int calls, size_t each, bool do_touch) {
    std::vector<void*> p((size_t)calls);
    const double t0 = now();
    for (int i = 0; i < calls; i++) if (hipMalloc(&p[(size_t)i], each) != hipSuccess) { printf("hipMalloc failed\n"); return; }
    const double t1 = now();
    if (do_touch) { for (int i = 0; i < calls; i++) touch<<<1024, 256>>>((char*)p[(size_t)i], each); (void)hipDeviceSynchronize(); }
    const double t2 = now();
    for (int i = 0; i < calls; i++) (void)hipFree(p[(size_t)i]);
    const double t3 = now();
    printf("%4d x %8.1f MB: hipMalloc %7.2f ms (%6.1f us per call, %5.2f us per MB)%s, hipFree %6.2f ms\n", calls, each / 1e6, 1e3 * (t1 - t0), 1e6 * (t1 - t0) / calls,
           1e6 * (t1 - t0) / (calls * (each / 1e6)), do_touch ? "" : "", 1e3 * (t3 - t2));
    if (do_touch) printf("       first touch of every page: %.2f ms\n", 1e3 * (t2 - t1));
}
int main() {
    (void)hipFree(nullptr);
    for (int rep = 0; rep < 2; rep++) {
        printf("pass %d\n", rep);
        run(123, (size_t)24 << 20, rep == 0);
        run(1, (size_t)3000 << 20, rep == 0);
        run(12, (size_t)250 << 20, false);
        run(123, (size_t)1 << 20, false);
        run(1000, (size_t)64 << 10, false);
    }
    return 0;
}
