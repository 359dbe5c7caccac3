// hostcopy.hpp - every copy between PAGEABLE host memory (the caller's arrays, numpy buffers, std::vectors of the library) and the device goes through
// page-locked bounce buffers that the library owns for the life of the process (api.hip).
//
// Why (round 5, the "illegal memory access" of GPUTEST_r04): handed a pageable pointer and a megabyte or more, the HIP runtime page-locks the CALLER's pages in
// place (a KFD userptr registration of that address range) and keeps such registrations in per-queue caches beyond the copy; hipHostRegister does the same thing
// on request.  A registration that outlives the memory it described - the array was freed, the address range unmapped and later handed out again by the allocator -
// describes physical pages that are gone: the next copy the runtime resolves against it makes the GPU fault, in whatever innocent call happens to use that
// address (a fresh numpy array in the failing test).  The library cannot see or flush those caches, so it never lets the runtime near memory it does not own:
// the runtime only ever sees hipHostMalloc memory (allocated once, never unmapped) and device pointers.  (Round 5 kept the direct copies behind SVX_COPY_DIRECT=1
// for A/B runs; the one run with it hung and nothing ever tested it: removed in round 6.)
//
// A HostCopy object batches the copies of one call on one stream:
//     HostCopy hc(stream);
//     SVXCHK(hc.h2d(dev, host, bytes));     // returns when `host` has been read (it may be reused / freed); the device copy is ordered on the stream.
//                                           // `host` inside memory from svx_host_alloc: no bounce, the copy engine reads it in place - `host` has been read after finish()
//     SVXCHK(hc.d2h(host, dev, bytes));     // enqueued; `host` holds the data after finish()
//     SVXCHK(hc.out(dst, dev, bytes));      // dst may be host OR device memory (the hipMemcpyDefault of before): looked up once
//     SVXCHK(hc.finish());                  // waits for the stream where device -> host copies are pending, then hands the bytes out
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <vector>

struct BounceSlot;
struct HostCopy {
    explicit HostCopy(hipStream_t st) : st_(st) {}
    ~HostCopy();
    HostCopy(const HostCopy&) = delete; HostCopy& operator=(const HostCopy&) = delete;
    int h2d(void* dev_dst, const void* host_src, size_t bytes);
    int d2h(void* host_dst, const void* dev_src, size_t bytes);
    int out(void* dst_host_or_device, const void* dev_src, size_t bytes);
    int finish();
private:
    struct Pending { BounceSlot* slot; void* dst; size_t bytes; };
    int drain(size_t keep_big, size_t keep_all);
    hipStream_t st_;
    std::vector<Pending> pend_;
    size_t pend_big_ = 0;
    bool direct_pending_ = false;          // a copy straight out of the library's own page-locked memory (svx_host_alloc) is in flight
};
bool svx_host_owned(const void* p, size_t bytes);
bool svx_is_device_pointer(const void* p);
// one-shot forms (a HostCopy of one copy + finish)
int svx_h2d(void* dev_dst, const void* host_src, size_t bytes, hipStream_t st);
int svx_d2h(void* host_dst, const void* dev_src, size_t bytes, hipStream_t st);
