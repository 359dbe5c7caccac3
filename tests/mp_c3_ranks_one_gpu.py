"""Worker of tests/test_gpu_workloads.py::test_c3_whole_genome_contig_sharded_ranks_on_one_gpu (run under torch.distributed.run, all ranks on cuda:0, backend gloo -
RCCL refuses several ranks on one device): BASELINE.json configs[3] as far as one GPU allows.  A whole-genome-like BAM (svim_amd/workloads.py profile c3: 47 contigs
whose name order interleaves the header order) is read contig-sharded - every rank its own contig runs through the .bai with the DEVICE-RESIDENT reader -,
collected, clustered with svx_cluster's rank exchange over the process group, gathered on rank 0 (harness.collect_cluster_bam_sharded: foreign BND / DUP_INT rows
travel with their read names).  Rank 0 compares the merged result with a single-rank run over the whole file and with the oracle.
Prints one line: C3_RANKS_OK <clusters> <foreign rows> <ranks that own contigs> - or the first difference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _compare(res, full):
    """rank 0: StepResult (device tensors) vs a single-process ClusterTable; members through their position in emission order"""
    from svim_amd._abi import CLU_DTYPES
    got = res.to_host()
    if got.n != full.n or list(got.type_count) != list(full.type_count):
        return "n %d/%d type_count %r/%r" % (got.n, full.n, got.type_count, full.type_count)
    for k in CLU_DTYPES:
        a, b = getattr(got, k), getattr(full, k)[:full.n]
        same = (np.isnan(a) & np.isnan(b)) | (np.abs(a - b) <= 1e-12 * np.maximum(1.0, np.abs(b))) if a.dtype == np.float64 else a == b
        if not same.all():
            i = int(np.nonzero(~same)[0][0])
            return "%s[%d]: %r != %r" % (k, i, a[i], b[i])
    if not np.array_equal(got.member_off, full.member_off[:full.n + 1]):
        return "member_off differs"
    keys = res.sig_cols["key"].cpu().numpy()
    pos = np.empty(keys.size, dtype=np.int64)
    pos[np.argsort(keys, kind="stable")] = np.arange(keys.size)
    if not np.array_equal(pos[got.members], full.members[:full.n_members]):
        return "member positions differ"
    return "ok"


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    bam_path, scale, seed = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    torch.cuda.set_device(0)
    dev = "cuda:0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from svim_amd import _abi, _lib, harness, multigpu as MG, workloads
    from svim_amd.batch import contig_ranks
    o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                              partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5,
                              all_bnds=False)
    p = _abi.Params.from_options(o)
    prof = workloads.profile("c3", scale)
    b, genome, g_off, meta = workloads.make_batch_full(prof, seed=seed, device=dev)          # every rank: the same batch (for the genome)
    refs = [c[0] for c in prof["contigs"]]
    lens = [int(x) for x in (g_off[1:] - g_off[:-1]).tolist()]
    if rank == 0:
        harness.write_bam_from_device_batch(bam_path, b, refs, lens, index=True, slab_bytes=64 << 20, threads=4)
    dist.barrier()
    eng = _lib.Engine(0)
    eng.set_genome(g_off, genome, on_device=True)
    ad = MG.SvxAdapter(eng, dev)
    res, pipe, names = harness.collect_cluster_bam_sharded(bam_path, o, eng, ad, rank, world, threads=2, batch_records=4000)
    device_reader = bool(pipe.device_decode)
    n_regions = len(pipe.region_slots)
    pipe.close()
    owner = MG.assign_contigs(refs, lens, world)
    verdict = "ok"
    if rank == 0:
        from oracle import oracle as om
        from helpers import granted_cpus
        # (a) one rank, the whole file, same reader
        e1 = _lib.Engine(0)
        e1.set_genome(g_off, genome, on_device=True)
        p1 = harness.BamPipeline(bam_path, o, e1, threads=2, batch_records=4000)
        p1.run()
        p1.cluster()
        one = e1.fetch_clusters()
        one_sig = e1.fetch_signatures(0)
        one_names = p1.bam.read_names()
        p1.close()
        e1.close()
        # (b) the oracle on the records of the file
        orc = om.Oracle()
        orc.set_threads(granted_cpus())
        orc.set_genome(g_off.cpu().numpy().astype(np.int64), genome.cpu().numpy())
        hb = b.slice_records(0, b.n_rec)
        osig, _ = orc.collect(hb, p)
        full = orc.cluster(p, contig_ranks(refs), source=0)
        if one.first_difference(full, rtol=1e-12) is not None:
            verdict = "single rank vs oracle: %s" % (one.first_difference(full, rtol=1e-12),)
        else:
            verdict = _compare(res, full)
        if verdict == "ok":
            # every gathered signature resolves to its read name on rank 0; in emission order they are the single-rank run's
            keys = res.sig_cols["key"].cpu().numpy()
            order = np.argsort(keys, kind="stable")
            got = [res.read_name(int(i)) for i in res.sig_cols["read_id"].cpu().numpy()[order]]
            want = [one_names[int(i)] for i in one_sig.read_id[:one_sig.n]]
            if got != want:
                verdict = "read names of the gathered signatures differ"
        cross = np.isin(osig.type[:osig.n], (_abi.SVX_BND, _abi.SVX_DUP_INT)) & (osig.contig2[:osig.n] >= 0)
        n_foreign = int((owner[osig.contig[:osig.n][cross]] != owner[osig.contig2[:osig.n][cross]]).sum())          # signatures whose two contigs belong to different ranks
        print(("C3_RANKS_OK %d %d %d %d" % (full.n, n_foreign, len(set(owner.tolist())), int(device_reader))) if verdict == "ok" else "C3_RANKS_FAIL " + verdict, flush=True)
    # diagnostics for a failure: what every rank collected (signatures by type, inserted bases) next to what the oracle's list says it should have
    n_own, n_seq = ad.collect_counts()
    by_type = [0] * 6
    if n_own:
        cols, _, _ = ad.fetch_signatures(with_seq=False)
        by_type = torch.bincount(cols["type"].long(), minlength=6).tolist()
    stats = [None] * world
    dist.all_gather_object(stats, (n_regions, device_reader, by_type, int(n_seq)))
    if rank == 0 and verdict != "ok":
        hb_tid = hb.arrays["tid"]
        rec_of_slot = {int(o) >> 1: i for i, o in enumerate(hb.arrays["order"].tolist())}
        want = [[0] * 6 for _ in range(world)]
        want_seq = [0] * world
        for i in range(osig.n):
            rec = rec_of_slot.get(int(osig.key[i] >> np.uint64(33)))
            r = int(owner[hb_tid[rec]]) if rec is not None else -1
            if r >= 0:
                want[r][int(osig.type[i])] += 1
                want_seq[r] += int(osig.seq_off[i + 1] - osig.seq_off[i])
        for r in range(world):
            print("C3_DIAG rank %d collected %r seq %d | oracle says %r seq %d" % (r, stats[r][2], stats[r][3], want[r], want_seq[r]), flush=True)
    if rank == 0:
        print("C3_REGIONS " + " ".join(str(s[0]) for s in stats), flush=True)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
