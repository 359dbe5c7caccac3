// devdec.hpp - internal interface between the host side of the BAM reader (bamio.cpp, plain C++) and its device-resident decode path (bamdev.hip):
// BGZF blocks -> inflated stream in HBM -> record table -> svx_batch arrays on the device.  No HIP types here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/svx.h"

struct DevDecBlock { const uint8_t* comp; uint32_t clen, isize, crc; };   // one BGZF block: raw DEFLATE payload in the memory-mapped file, inflated size, CRC32 of the inflated bytes

struct svx_devdec;
// names_blob: the reference names NUL-separated in header order (SA tags name contigs)
int  devdec_create(int device, int n_threads /* CPUs the host's share of the inflate may use */, int32_t n_ref, const int32_t* ref_len, const char* names_blob, const int32_t* contig_rank, svx_devdec** out);
void devdec_destroy(svx_devdec* d);
// Inflate `n` blocks into chunk slot `slot` (0..2) behind the unconsumed tail of slot `carry_slot` (-1: none), skip `skip_bytes` at the start of the
// stream (the BAM header, first chunk only), find every complete record and decode all of them.  final_chunk: nothing follows (a partial record at the
// end is an error).  min_mapq: primaries below it get no segment rows (src/svim/SVIM_COLLECT.py:143-161).
// mode 1 = query-name-sorted input (src/svim/SVIM_COLLECT.py:96-129): no SA tag is expanded, and unless final_chunk the last read's group of the chunk is left to the
// next load (groups never straddle loads).
int  devdec_load(svx_devdec* d, int slot, const DevDecBlock* blocks, size_t n, int carry_slot, uint64_t skip_bytes, bool final_chunk, int min_mapq, int mode);
// records decoded in the slot; n_valid: those before the first record whose reference id is negative or above tid_limit (tid_limit -2: all)
int  devdec_count(svx_devdec* d, int slot, int32_t tid_limit, int64_t* n_rec, int64_t* n_valid);
// device-resident svx_batch over records [first, first + count) of the slot (arrays stay valid until the slot is loaded again)
// mode 1: *count grows to the end of the read group it ends in; flags carry SVX_FLAG_SKIP, the segment table holds the good supplementary records of every analysed
// read and the emission slots follow the reference's per-read order (bamio.cpp svx_bam_read_batch does the same on the host)
int  devdec_batch(svx_devdec* d, int slot, int64_t first, int64_t* count, int mode, int min_mapq, svx_batch* out);
// read names interned so far, id order (host copy; grows with every load)
const std::vector<std::string>& devdec_names(svx_devdec* d);
// timing / accounting of the loads so far
struct DevDecStats { double t_stage = 0, t_inflate_wait = 0, t_discover = 0, t_decode = 0, t_names = 0; int64_t blocks = 0, gpu_blocks = 0, cpu_blocks = 0, bytes = 0, records = 0, fallbacks = 0; double inflate_kernel_ms = 0; };
void devdec_stats(svx_devdec* d, DevDecStats* out);
void devdec_reset_names(svx_devdec* d);
