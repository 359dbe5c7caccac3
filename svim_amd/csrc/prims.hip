// prims.hip - device-wide sort / scan used between the hand-written stages.
// Round-1 scaffolding: these two generic primitives come from rocPRIM (AMD's own gfx950-tuned library); they
// account for a negligible share of the path's time (profiles/), every stage kernel is hand-written.
#include "common.hpp"
#include <rocprim/rocprim.hpp>

int svx_sort_pairs_u64(svx_ctx* c, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                       int64_t n, int begin_bit, int end_bit) {
    if (n <= 0) return SVX_OK;
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)begin_bit,
                                     (unsigned)end_bit, c->stream));
    SVXCHK(c->sort_tmp.reserve(bytes));
    HIPCHK(rocprim::radix_sort_pairs(c->sort_tmp.p, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, (unsigned)begin_bit,
                                     (unsigned)end_bit, c->stream));
    return SVX_OK;
}

int svx_exclusive_scan_i64(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n) {
    if (n <= 0) return SVX_OK;
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    SVXCHK(c->sort_tmp.reserve(bytes));
    HIPCHK(rocprim::exclusive_scan(c->sort_tmp.p, bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    return SVX_OK;
}

// the same on another stream with its own temporary storage (concurrent with the main stream's primitives)
int svx_exclusive_scan_i64_on(svx_ctx* c, const int64_t* in, int64_t* out, int64_t n, hipStream_t stream, DevBuf& tmp) {
    (void)c;
    if (n <= 0) return SVX_OK;
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), stream));
    SVXCHK(tmp.reserve(bytes));
    HIPCHK(rocprim::exclusive_scan(tmp.p, bytes, in, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), stream));
    return SVX_OK;
}

struct I32ToI64 {
    __host__ __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

int svx_exclusive_scan_i32_to_i64(svx_ctx* c, const int32_t* in, int64_t* out, int64_t n) {
    if (n <= 0) return SVX_OK;
    auto it = rocprim::make_transform_iterator(in, I32ToI64());
    size_t bytes = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, bytes, it, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    SVXCHK(c->sort_tmp.reserve(bytes));
    HIPCHK(rocprim::exclusive_scan(c->sort_tmp.p, bytes, it, out, (int64_t)0, (size_t)n, rocprim::plus<int64_t>(), c->stream));
    return SVX_OK;
}
