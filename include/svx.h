/*
 * svx.h - C ABI of the MI355X-native COLLECT+CLUSTER path (libsvx.so).
 *
 * The reference (eldariont/svim v2.0.0) has no FFI seam: its hot path is the Python call boundary
 *   analyze_alignment_file_coordsorted / _querysorted   src/svim/SVIM_COLLECT.py:132 / :96
 *   analyze_alignment_indel / analyze_cigar_indel        src/svim/SVIM_intra.py:33 / :8
 *   analyze_read_segments                                src/svim/SVIM_inter.py:24
 *   cluster_sv_signatures -> partition_and_cluster       src/svim/SVIM_CLUSTER.py:7, SVIM_clustering.py:375
 * The entry points below are what a ctypes binding behind those Python functions binds
 * (INTEGRATION.md shows the stub).  Plain pointers and sizes only; every array is Structure-of-Arrays.
 *
 * Conventions
 *   - return value: 0 = ok, negative = SVX_E_* (no exceptions cross the boundary)
 *   - pointers in svx_batch / svx_sig_view / svx_genome may be HOST or DEVICE memory; the `on_device`
 *     member says which (device pointers let a caller keep everything resident in HBM)
 *   - results stay resident in the context (HBM) until fetched with svx_*_fetch into caller-allocated
 *     host arrays sized from svx_*_count
 *   - HOST arrays handed to the library are pageable memory as far as it is concerned: it never registers them with the GPU (no
 *     hipHostRegister, no copy call of the runtime ever sees their address) - every host <-> device copy goes through page-locked
 *     buffers the library owns (csrc/hostcopy.hip).  An input array may be changed or freed as soon as the call returns; an output
 *     array holds its data when the call returns
 *   - the exception a caller can ask for: an input array that lies in memory from svx_host_alloc (page-locked, owned by the library, mapped
 *     until the process ends) is read by the copy engine in place - no bounce pass (what a numpy array costs: one memcpy of its bytes)
 *   - one context per GPU / per process; a context is not re-entrant
 */
#ifndef SVX_H
#define SVX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SVX_OK               0
#define SVX_E_NODEVICE     (-1)   /* no HIP device / extension cannot run: the product path never falls back to CPU */
#define SVX_E_HIP          (-2)   /* a HIP runtime call failed (svx_last_error() has the text) */
#define SVX_E_ARG          (-3)
#define SVX_E_CAPACITY     (-4)   /* an internal fixed-capacity buffer overflowed */
#define SVX_E_STATE        (-5)   /* call order violated (e.g. cluster before collect/set_signatures) */

/* signature types, in the order CLUSTER processes them (src/svim/SVIM_CLUSTER.py:19-24) */
enum { SVX_DEL = 0, SVX_INS = 1, SVX_INV = 2, SVX_DUP_TAN = 3, SVX_BND = 4, SVX_DUP_INT = 5, SVX_NTYPES = 6 };
/* sig.src */
enum { SVX_SRC_CIGAR = 0, SVX_SRC_SUPPL = 1 };
/* sig.aux for INV (src/svim/SVIM_inter.py:159-198) */
enum { SVX_LEFT_FWD = 0, SVX_LEFT_REV = 1, SVX_RIGHT_FWD = 2, SVX_RIGHT_REV = 3, SVX_DIR_ALL = 4 };
/* sig.aux for BND: bit0 = direction1 is 'rev', bit1 = direction2 is 'rev'; for DUP_TAN: bit0 = fully_covered */

/* host-set marker in svx_batch.flag: record is to be ignored (query-sorted mode: read without exactly one
 * good primary, src/svim/SVIM_COLLECT.py:108) */
#define SVX_FLAG_SKIP 0x8000u
/* host-set marker: this record's segment rows were rebuilt from its SA tag, so they are void when the
 * record has hard-clipped bases (get_cigar_stats()[0][5] > 0, src/svim/SVIM_COLLECT.py:47) */
#define SVX_FLAG_SA   0x4000u

/* options read on the path (src/svim/SVIM_input_parsing.py:279-371) */
typedef struct svx_params {
    int32_t min_mapq;                      /* 20     */
    int32_t min_sv_size;                   /* 40     */
    int32_t max_sv_size;                   /* 100000 */
    int32_t segment_gap_tolerance;         /* 10     */
    int32_t segment_overlap_tolerance;     /* 5      */
    int32_t all_bnds;                      /* 0      */
    int64_t partition_max_distance;        /* 1000   */
    double  position_distance_normalizer;  /* 900    */
    double  edit_distance_normalizer;      /* 1.0    */
    double  cluster_max_distance;          /* 0.5    */
} svx_params;

/* One batch of BAM records in file order (SoA).  Replaces the per-record pysam accessors used at
 * src/svim/SVIM_COLLECT.py:143-161 and the SA-tag re-materialisation at :44-93 (the host parses the SA
 * string into the segment table; the device derives every coordinate from the packed CIGARs). */
typedef struct svx_batch {
    int32_t on_device;
    int64_t n_rec;
    const uint16_t* flag;        /* [n_rec] SAM FLAG (+ SVX_FLAG_SKIP) */
    const int32_t*  tid;         /* [n_rec] reference_id */
    const int32_t*  pos;         /* [n_rec] 0-based reference_start */
    const uint8_t*  mapq;        /* [n_rec] */
    const int32_t*  lseq;        /* [n_rec] l_qseq of the stored SEQ (0 = none) */
    const int32_t*  read_id;     /* [n_rec] interned query_name */
    const uint32_t* order;       /* [n_rec] emission slot of this record's CIGAR indels */
    const uint32_t* seg_order;   /* [n_rec] emission slot of this read's split-alignment signatures */
    const uint64_t* cigar_off;   /* [n_rec+1] */
    const uint32_t* cigar;       /* BAM packed: len<<4 | op */
    const uint64_t* seq_off;     /* [n_rec+1] byte offsets into seq */
    const uint8_t*  seq;         /* 4-bit packed bases, BAM layout (high nibble first) */
    const uint32_t* seg_off;     /* [n_rec+1] rows of the segment table per record (other alignments of the read) */
    int64_t n_seg;
    const int32_t*  seg_tid;     /* [n_seg] */
    const int32_t*  seg_pos;     /* [n_seg] 0-based */
    const uint8_t*  seg_rev;     /* [n_seg] 1 = reverse strand */
    const uint8_t*  seg_mapq;    /* [n_seg] (SA mapq > 255 already mapped to 0, src/svim/SVIM_COLLECT.py:81-84) */
    const int32_t*  seg_lseq;    /* [n_seg] l_qseq of the segment record (SA rebuild: the primary's) */
    const uint64_t* seg_cigar_off; /* [n_seg+1] */
    const uint32_t* seg_cigar;
    int32_t n_contig;
    const int32_t*  contig_rank; /* [n_contig] rank of each contig NAME in Python str order */
    /* optional sparse SEQ (NULL: every record's whole SEQ sits at seq_off): COLLECT reads a record's bases only for reported insertions,
     * so a reader may keep just those ranges (svx_bam_set_seq_filter).  Record i then owns ranges seq_rng_off[i] .. seq_rng_off[i+1]-1,
     * range r holds bases seq_rng_q0[r] (even) .. +seq_rng_len[r]-1 of the read, packed from byte seq_rng_byte[r] of seq on. */
    const uint32_t* seq_rng_off; /* [n_rec+1] */
    const int32_t*  seq_rng_q0;
    const int32_t*  seq_rng_len;
    const uint64_t* seq_rng_byte;
    int64_t n_seq_rng;
} svx_batch;

/* Signature table (SoA).  Mirrors the fields of the six Signature classes (src/svim/SVSignature.py:3-233). */
typedef struct svx_sig_view {
    int32_t on_device;
    int64_t n;
    uint64_t* key;       /* emission order key: slot<<32 | phase<<30 | ordinal (ascending = reference list order) */
    uint8_t*  type;      /* SVX_DEL .. SVX_DUP_INT */
    uint8_t*  src;       /* SVX_SRC_* */
    uint8_t*  aux;
    int32_t*  contig;    /* contig / contig1 / source contig (tid) */
    int32_t*  start;     /* start / pos1 */
    int32_t*  end;       /* end   / pos1+1 for BND */
    int32_t*  contig2;   /* BND contig2, DUP_INT destination contig, else -1 */
    int32_t*  pos2;      /* BND pos2, DUP_INT pos, DUP_TAN copies, else 0 */
    int32_t*  read_id;
    int64_t*  seq_off;   /* [n+1] INS: offsets into seq; other types have empty ranges */
    uint8_t*  seq;       /* inserted bases, one 4-bit code ("=ACMGRSVTWYHKDBN") per byte */
} svx_sig_view;

/* Reference genome, one 4-bit code per byte, upper-cased (pysam.FastaFile.fetch(...).upper() at
 * src/svim/SVIM_clustering.py:37-43). */
typedef struct svx_genome {
    int32_t on_device;
    int32_t n_contig;
    const int64_t* off;      /* [n_contig+1] */
    const uint8_t* codes;
} svx_genome;

/* Consolidated clusters (src/svim/SVSignature.py:236-311, SVIM_clustering.py:214-303), grouped by type
 * in SVX_* order; unilocal types already sorted by (contig name, (start+end)/2) as at :381. */
typedef struct svx_cluster_view {
    int64_t n;                 /* capacity on input to fetch, count on output */
    int64_t type_count[SVX_NTYPES];
    uint8_t* type;
    int32_t* contig;  int32_t* start;  int32_t* end;      /* unilocal / source */
    int32_t* contig2; int32_t* start2; int32_t* end2;     /* destination (bilocal types) */
    uint8_t* aux;              /* BND: direction bits */
    double*  score;
    double*  std_span;         /* NaN = None */
    double*  std_pos;
    int32_t* size;
    int64_t* member_off;       /* [n+1] */
    int32_t* members;          /* [n_members] indices into the clustered signature table, cluster-major */
    int64_t  n_members;
} svx_cluster_view;

typedef struct svx_ctx svx_ctx;

/* timing / traffic counters of the last collect / cluster call (seconds measured with HIP events on the
 * context's stream) */
typedef struct svx_stats {
    double  t_collect_ms, t_cluster_ms;
    double  t_cigar_scan_ms, t_segments_ms, t_sort_ms, t_partition_ms, t_edit_ms, t_linkage_ms, t_gather_ms;
    int64_t n_rec_used, n_ops, n_seg, n_seg_ops, n_sig, n_bnd_side, n_ins_bases;
    int64_t n_partitions, n_large_partitions, n_pairs, n_edit_pairs, n_edit_cells, n_clusters, n_hap_bytes;
    /* edit-distance work actually executed by the last svx_cluster: 32-bit word-columns (32 DP cells each) the waves issued (lock-step
     * and padding included), the ones the pairs needed, the issued ones of retry rounds (>= 1) and of the band kernels; the band
     * speculation fraction the call's pilot chose */
    int64_t n_edit_wordcols_issued, n_edit_wordcols_useful, n_edit_wordcols_retry, n_edit_wordcols_band;
    double  edit_guess;
} svx_stats;

/* ---- lifecycle ------------------------------------------------------------------------------- */
int  svx_ctx_create(int device_ordinal, svx_ctx** out);     /* SVX_E_NODEVICE when no GPU is present */
void svx_ctx_destroy(svx_ctx* ctx);
const char* svx_last_error(void);
int  svx_version(void);
int  svx_get_stats(svx_ctx* ctx, svx_stats* out);
void* svx_stream(svx_ctx* ctx);                              /* hipStream_t the kernels run on */
int  svx_memcpy_d2h(void* host_dst, const void* device_src, uint64_t bytes);   /* inspection of device-resident results (tests) */
int  svx_memcpy_h2d(void* device_dst, const void* host_src, uint64_t bytes);
void* svx_dev_alloc(uint64_t bytes);                                              /* library-owned device memory (tests, stress tools); NULL on failure */
void svx_dev_free(void* p);
/* page-locked host memory for the caller's batch arrays (svx_batch with on_device = 0): arrays built here are uploaded without the bounce pass.  The reference
   builds its per-read Python objects from pysam (src/svim/SVIM_COLLECT.py:133-167); the batcher that replaces that loop (svim_amd/batch.py) fills these arrays.
   svx_host_free hands the block back to the library (it is reused by a later svx_host_alloc, never unmapped); NULL on failure */
void* svx_host_alloc(uint64_t bytes);
void svx_host_free(void* p);
int  svx_device_synchronize(void);                                               /* SVX_OK iff no kernel / copy of the process has faulted */
/* self-test of the library's own radix sort (64-bit keys + 32-bit values, key bits [begin_bit, end_bit), stable) and exclusive scan on n pseudo-random
   elements, checked on the host against std::stable_sort / a serial sum: 0 = identical (tests) */
int  svx_selftest_prims(svx_ctx* ctx, int64_t n, int32_t begin_bit, int32_t end_bit, uint64_t seed);

/* ---- COLLECT: replaces analyze_alignment_file_* (src/svim/SVIM_COLLECT.py:96-167) -------------- */
int  svx_collect(svx_ctx* ctx, const svx_batch* batch, const svx_params* p);
/* One input file usually arrives as several record batches (svx_bam_read_batch).  With accumulation on (mode 1; it also clears what was
 * accumulated before) every svx_collect APPENDS its two lists to the lists resident in the context, so that svx_collect_count / _fetch and
 * svx_cluster(source 0 / 1) see the whole file - the loop of src/svim/SVIM_COLLECT.py:132-167 over all records - without a table ever
 * leaving HBM.  The emission keys of a batch are shifted by slot_base << 32: pass the number of emission slots of all earlier batches
 * (2 per record is always enough).  mode 0: back to one batch per call; what was accumulated stays resident as "the result of the last
 * COLLECT" (count / fetch / svx_cluster source 0 and 1 keep seeing the whole file). */
int  svx_collect_accumulate(svx_ctx* ctx, int mode);
int  svx_collect_set_slot_base(svx_ctx* ctx, uint64_t slot_base);
int  svx_collect_count(svx_ctx* ctx, int64_t* n_sig, int64_t* n_seq_bytes, int64_t* n_bnd_side);
/* which: 0 = sv_signatures, 1 = translocation_signatures_all_bnds (second list of the reference's tuple) */
int  svx_collect_fetch(svx_ctx* ctx, int which, svx_sig_view* host_out);

/* ---- CLUSTER: replaces cluster_sv_signatures (src/svim/SVIM_CLUSTER.py:7-26) -------------------- */
int  svx_set_genome(svx_ctx* ctx, const svx_genome* g);      /* FastaFile(options.genome), SVIM_clustering.py:377 */
/* source: 0 = signatures resident from the last svx_collect, 1 = its all_bnds side list,
 *         2 = the table passed in `sigs` (host or device memory) */
int  svx_cluster(svx_ctx* ctx, int source, const svx_sig_view* sigs, int32_t n_contig,
                 const int32_t* contig_rank_host, const svx_params* p);
int  svx_cluster_count(svx_ctx* ctx, int64_t* n_clusters, int64_t* n_members);
int  svx_cluster_fetch(svx_ctx* ctx, svx_cluster_view* out);  /* destination arrays: host or device memory */
/* partitions of the last svx_cluster as form_partitions (src/svim/SVIM_clustering.py:17-29) makes them: sorted_index[n_sig] = signature indices in partition
 * order, part_start[n_part + 1] = where each partition begins in it.  Inspection hook (tests compare it with the reference's partitions); NULL arrays: counts only */
int  svx_cluster_partitions_fetch(svx_ctx* c, int64_t* n_sig, int64_t* n_part, uint32_t* sorted_index, int64_t* part_start);

/* multi-GPU, contig-sharded ranks: each rank clusters ONLY the signatures of the contigs it owns - every partition of
 * src/svim/SVIM_clustering.py:17-29 is local to one rank.  What still couples the ranks is the random.sample word stream, which a signature type's
 * > 100-member partitions consume in global sorted order without re-seeding (src/svim/SVIM_clustering.py:129-134).  When ranks own consecutive ranges
 * of the name-sorted contig list that order is rank-major: rank r continues each type's stream where the partitions of ranks 0..r-1 stop.
 * svx_cluster finds those positions itself, with ALL-GATHERS ONLY (no rank waits for another rank's sampling): (1) the sizes of everybody's large
 * partitions (4 B each); (2) every rank builds, concurrently, its transfer table "stream position before my partitions -> position after them" for a
 * 6-sigma window around the start it expects from (1) (a few thousand 8 B entries per type), the tables are all-gathered and composed.  Only when a
 * partition beyond 1045 members exists somewhere (random.sample's set method: its consumption depends on the values drawn) or a start leaves its
 * window do the ranks fall back to publishing exact end positions rank after rank (world all-gathers of 128 B).
 * The transport is injected: `fn` must all-gather `bytes` bytes of host memory per rank into `recv` (rank-major, world * bytes) and return 0 - e.g. one
 * torch.distributed.all_gather_into_tensor over RCCL.  Every rank must call svx_cluster once per step (a rank without signatures takes part with an
 * empty table); a rank that fails sends a poison header so that the others fail instead of hanging (svx_cluster_abort_ranks: the same for a failure
 * before the call).  fn == NULL or world <= 1: single rank, every stream starts at 0. */
typedef int (*svx_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes);
int  svx_cluster_set_ranks(svx_ctx* ctx, int rank, int world, svx_allgather_fn fn, void* user);
int  svx_cluster_abort_ranks(svx_ctx* ctx);
/* where each type's stream started / stopped on this rank in the last svx_cluster (32-bit words after seed(1524), SVX_* type order) */
int  svx_cluster_stream_positions(svx_ctx* ctx, int64_t* start /* [SVX_NTYPES] */, int64_t* end /* [SVX_NTYPES] */);

/* ---- GENOTYPE: replaces the per-candidate BAM re-fetch of genotype() (src/svim/SVIM_genotyping.py:34-93) --------
 * by an interval join over the alignment records, resident in HBM.  Records are in file order of a coordinate-sorted BAM
 * (tid, pos non-decreasing); AlignmentFile.fetch(contig, start, stop) of the reference (:48) becomes "records of that contig with
 * pos < stop and end_or_pos1 > start, in order" (htslib's overlap rule; end_or_pos1 = reference_end, or pos + 1 for a record
 * without reference span). */
typedef struct svx_aln_index {
    int64_t n;                      /* records */
    int32_t n_contig;
    int32_t reserved;
    const int64_t* contig_first;    /* [n_contig+1] first record of every contig */
    const int64_t* contig_len;      /* [n_contig]   bam.get_reference_length */
    const int32_t* pos;             /* reference_start */
    const int32_t* end;             /* reference_end (pos when the record has no reference span) */
    const uint16_t* flag;
    const uint8_t*  mapq;
    const int32_t* name_id;         /* query_name interned: records of one read share the id */
} svx_aln_index;
int  svx_set_alignment_index(svx_ctx* ctx, const svx_aln_index* host_index);
/* mode 0 = DEL / INV candidates (locus = source start..end), 1 = INS / DUP_INT (locus = destination start, end == start).
 * member_names: interned read names of the candidate's members, SORTED within every candidate; out_ref_reads[i] =
 * len(reads_supporting_reference) of candidate i (:52-77: first 500 eligible alignments around the locus, distinct names). */
int  svx_genotype(svx_ctx* ctx, int32_t mode, int64_t n_cand, const int32_t* cand_tid, const int32_t* cand_start, const int32_t* cand_end,
                  const int64_t* member_off /* [n_cand+1] */, const int32_t* member_names, int32_t min_mapq, int32_t* out_ref_reads);

/* ---- single-function entry points kept importable by the reference's API ------------------------ */
/* analyze_cigar_indel (src/svim/SVIM_intra.py:8-30) on one packed CIGAR; out arrays sized n_ops */
int  svx_cigar_indel(svx_ctx* ctx, const uint32_t* cigar_host, int64_t n_ops, int32_t min_length,
                     int64_t* out_pos_ref, int64_t* out_pos_read, int32_t* out_len, uint8_t* out_is_del,
                     int64_t* out_n);
/* edlib.align(a,b)["editDistance"] (src/svim/SVIM_clustering.py:45) for n pairs of code strings */
int  svx_edit_distance(svx_ctx* ctx, int64_t n_pairs, const uint8_t* codes_host, const int64_t* a_off,
                       const int64_t* b_off /* [n_pairs+1] each, a and b ranges in codes */, int32_t* out_dist);
/* scipy linkage(method='average') + fcluster(criterion='distance') (SVIM_clustering.py:170-171) for a
 * batch of condensed matrices; labels out (1-based) */
int  svx_linkage_fcluster(svx_ctx* ctx, int64_t n_problems, const int32_t* n_host, const int64_t* d_off,
                          const double* d_host, double cutoff, const int64_t* label_off, int32_t* labels_out);

/* test / debug hook: span_position_distance (src/svim/SVIM_clustering.py:47-96) of n_pairs pairs (a[k], b[k]) of a HOST signature table
 * through the device code the clustering itself runs - FP64 operation order and haplotype edit distances included (svx_set_genome
 * first when insertions are among them) */
int  svx_pair_distances(svx_ctx* ctx, const svx_sig_view* host_sigs, int64_t n_pairs, const int64_t* a, const int64_t* b, const svx_params* p,
                        double* out);

/* ---- BGZF inflate on the GPU (first piece of a device-resident BAM front-end) ----------------------
 * Replaces zlib's inflate() as htslib runs it under pysam.AlignmentFile (src/svim/SVIM_COLLECT.py:132-137): every BGZF block is an independent
 * raw DEFLATE stream; one wavefront inflates one block (svim_amd/csrc/inflate_core.hpp, bit-identical to zlib).  The caller packs the payloads
 * (the bytes between the block header and the CRC32/ISIZE trailer) into the pinned staging buffer at 8-byte aligned offsets. */
typedef struct svx_inflater svx_inflater;
int   svx_inflater_create(int device, svx_inflater** out);
void  svx_inflater_destroy(svx_inflater* f);
/* three slots (0..2), each with its own stream, device buffers and pinned staging buffer: while one sub-batch is inflated and copied back, the
 * caller packs the next.  enqueue = H2D + inflate + copy of the inflated range to `out` (host, or device when out_on_device), asynchronous;
 * wait = its completion (error if a block was not a sound DEFLATE stream of ISIZE bytes).  run = enqueue + wait on slot 0. */
void* svx_inflater_staging(svx_inflater* f, int slot, uint64_t bytes);   /* NULL (SVX_E_STATE) while the slot is busy and the buffer would have to grow: wait first */
int   svx_inflater_enqueue(svx_inflater* f, int slot, int64_t n, const uint64_t* in_off, const uint32_t* clen, const uint32_t* isize,
                           const uint64_t* out_at, uint64_t staged_bytes, uint8_t* out, uint64_t out_bytes, int out_on_device);
int   svx_inflater_wait(svx_inflater* f, int slot, float* kernel_ms /* may be NULL */);
int   svx_inflater_run(svx_inflater* f, int64_t n, const uint64_t* in_off, const uint32_t* clen, const uint32_t* isize, const uint64_t* out_at,
                       uint64_t staged_bytes, uint8_t* out, uint64_t out_bytes, int out_on_device, float* kernel_ms /* may be NULL */);

/* ---- native BAM front-end (host side) --------------------------------------------------
 * Replaces pysam.AlignmentFile(bam).fetch(until_eof=True) + the per-record accessors + the SA-tag string handling of
 * src/svim/SVIM_COLLECT.py:8-41,44-85,133 for BAM inputs: multi-threaded BGZF inflate, records decoded straight into
 * a record batch whose host arrays are owned by the handle and stay valid until the next read or close.  mode 0 = coordinate-sorted rules
 * (:132-167), 1 = query-name-sorted rules (:96-129; a read's group is never split across batches). */
typedef struct svx_bam svx_bam;
int  svx_bam_open(const char* path, int n_threads /* 0 = auto */, svx_bam** out);
void svx_bam_close(svx_bam* h);
int  svx_bam_header(svx_bam* h, int32_t* n_ref, const char** names_nul_separated, const int32_t** lengths, const char** sort_order);
int  svx_bam_read_batch(svx_bam* h, int64_t max_records, int mode, int min_mapq, svx_batch* out, int64_t* n_out);
/* The arrays of a batch stay valid until the SECOND next svx_bam_read_batch (two sets alternate): the caller can upload / collect batch i
 * while another thread already reads batch i+1.
 * svx_bam_set_seq_filter(h, min_ins_len > 0): coordinate mode keeps only the SEQ ranges COLLECT can read - insertions of at least
 * min_ins_len bases (pass params.min_sv_size) and the whole SEQ of records with an SA tag - and describes them in svx_batch.seq_rng_*. */
int  svx_bam_set_seq_filter(svx_bam* h, int min_ins_len);
/* back to the first record, keeping buffers, worker threads and the interned read names (a second pass = the steady state of a long file) */
int  svx_bam_rewind(svx_bam* h);
/* contig-sharded ranks: continue at BGZF virtual offset `voff` (the .bai gives the first record of every contig) and report end-of-file at
 * the first record whose reference id exceeds last_tid or that is unplaced (-2: no limit) */
int  svx_bam_seek(svx_bam* h, uint64_t voff, int32_t last_tid);
/* BGZF inflate shared between the GPU (svx_inflater on `device`; < 0: off) and the host's cores: of every chunk of blocks the GPU takes sub-batches
 * from the front while the worker threads take blocks from the back - whoever is faster inflates more.  stats: blocks inflated by either so far. */
int  svx_bam_set_gpu_inflate(svx_bam* h, int device);
int  svx_bam_gpu_inflate_stats(svx_bam* h, int64_t* gpu_blocks, int64_t* cpu_blocks, double* gpu_kernel_ms);
int  svx_bam_read_names(svx_bam* h, int64_t* n_names, const char** nul_separated, int64_t* blob_len);
/* Device-resident front-end (either sort order): the compressed file slice is the only thing that crosses PCIe.  Every chunk of BGZF blocks (8 GB of inflated
 * data) is inflated by the GPU into HBM (one wavefront per block), the record boundaries are found there (BGZF blocks are the restart points of the
 * block_size chain: a speculative record start per block, verified by linking the chains), and fixed fields, CIGAR (CG tag included), the SA tag -> segment
 * table and the read names (interned by 2 x 64-bit hashes) are decoded by kernels.  svx_bam_read_batch then returns an svx_batch whose pointers are DEVICE
 * memory (on_device = 1; seq points into the inflated stream - no bases are copied).  Lifetime: the arrays that are views of the chunk (tid, pos, mapq, lseq,
 * read_id, cigar_off, cigar, seq_off, seq and - mode 0 - flag and the seg_* table) stay valid until the THIRD next chunk is loaded; the arrays made per batch
 * (order, seg_order and - mode 1, query-name order - flag and the seg_* table built from the read's supplementary records) alternate between two sets per chunk
 * and stay valid until the SECOND next svx_bam_read_batch, like the batches of the host reader.  Every
 * inflated block is checked against the CRC32 of its BGZF trailer on the device (as htslib's bgzf_read_block does; environment SVX_BAM_VERIFY_CRC=0: off);
 * a damaged block fails the svx_bam_read_batch that would have handed out its records.  device < 0: back to the host reader. */
int  svx_bam_set_device_decode(svx_bam* h, int device);

#ifdef __cplusplus
}
#endif
#endif
