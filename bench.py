#!/usr/bin/env python3
"""bench.py - COLLECT+CLUSTER throughput on MI355X (BASELINE.json metric: aligned reads/sec through
COLLECT+CLUSTER; SV signatures/sec clustered).

    python bench.py --gpus 1 --steps 3 --warmup 1                          # configs[1], the configuration the metric is quoted on
    python bench.py --workload c2                                          # configs[2] stand-in: HiFi profile, full SV-type set
    python bench.py --workload c4 --partition-max-distance 20000           # configs[4] stand-in: 60x CLR profile, large partitions
    python bench.py --bam reads.bam --fasta ref.fa                         # a real coordinate-sorted BAM (configs[2]-[4] proper)
    python bench.py --scaling strong [--workload c3 --scale 0.2]           # ONE whole-genome batch sharded by contig ownership (configs[3]'s shape): total work fixed,
                                                                           # N=1 is the whole batch; under torchrun with --gpus N: the strong-scaling line + fabric accounting
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (svx_collect + svx_cluster through the C ABI) over one record batch that is
already resident in HBM.  Default workload: BASELINE.json configs[1] - 1M synthetic ONT reads (N50 20 kb), one 250 Mb
contig, planted DEL/INS/INV (svim_amd/devsynth.py); c2 / c4: svim_amd/workloads.py.  Weak scaling: every rank owns its own
batch of that size (its own contigs); see svim_amd/multigpu.py for what crosses xGMI.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# VALU issue peak of the edit-distance kernels: 1024 SIMDs x 2.4 GHz, one wave64 instruction per 2 cycles (MI355X_MICROARCH.md)
VALU_PEAK_WAVE_INSTR_PER_S = 1024 * 2.4e9 / 2
# the Myers/Hyyro column update: VALU instructions per 32-bit word-column (svim_amd/csrc/myers_column.hpp): 9 full-rate instructions (2 issue cycles per
# wave64) + 3 v_addc_co of the carry chains (4) = 30 issue cycles by the rates of the hardware; the same classes MEASURE 2.3 / 4.4 cycles
# (profiles/r06_valu_banks.txt) = 33.9, and the update alone runs at 33.2 (profiles/r06_column_shift.txt).  Round 5: 12.4 instructions, 38 cycles (two v_alignbit).
INSTR_PER_WORDCOL = 12.0
CYCLES_PER_WORDCOL = 30.0
CYCLES_PER_WORDCOL_MEASURED_RATES = 33.9


def options(pmd=1000):
    return types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10,
                                 segment_overlap_tolerance=5, partition_max_distance=pmd, position_distance_normalizer=900,
                                 edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)


def reference_python_figure():
    """what the reference ITSELF (pure Python, src/svim/SVIM_COLLECT.py + SVIM_CLUSTER.py) needed for configs[0] at its stated size, recorded when
    tests/golden/make_golden.py ran it in the build container (the reference cannot travel to the GPU box): reads/s over COLLECT + CLUSTER"""
    try:
        import gzip
        with gzip.open(os.path.join(REPO, "tests", "golden", "g_c1_full.json.gz"), "rt") as fh:
            g = json.load(fh)
        t = g.get("reference_seconds") or {}
        tc, tk = float(t.get("collect", 0)), float(t.get("cluster", 0))
        if tc + tk <= 0:
            return None
        return {"reads_per_s": g["n_records"] / (tc + tk), "records": g["n_records"], "collect_s": tc, "cluster_s": tk,
                "workload": "configs[0] at its stated size (10 000 records, 250 Mb contig, triangular(100, 20000, 15000) read lengths, DEL/INS only), CPython 3.10, "
                            "1 core of the BUILD container, pysam / edlib stubbed (tests/golden/make_golden.py)"}
    except (OSError, ValueError, KeyError):
        return None


def reference_python_c1_shape():
    """the reference itself on a configs[1]-SHAPED sample (tests/golden/g_c1_bench_sample.json.gz: 20 000 reads + their supplementary records at the bench workload's
    densities, DEL / INS / INV), timed in the build container by tests/golden/make_golden.py c1_bench_sample"""
    try:
        import gzip
        with gzip.open(os.path.join(REPO, "tests", "golden", "g_c1_bench_sample.json.gz"), "rt") as fh:
            g = json.load(fh)
        t = g["reference_seconds"]
        return {"reads_per_s": g["n_reads"] / (t["collect"] + t["cluster"]), "reads": g["n_reads"], "records": g["n_records"], "collect_s": t["collect"], "cluster_s": t["cluster"],
                "collect_only_reads_per_s": g["n_reads"] / t["collect"],
                "workload": "configs[1]'s shape at a fiftieth of its size: 20 000 synthetic ONT reads (N50 20 kb) + 4 794 supplementary records on a 5 Mb contig, the bench "
                            "workload's densities of DEL / INS / INV sites; CPython 3.10, 1 core of the BUILD container, pysam / edlib stubbed (the edlib stand-in is a "
                            "pure-Python bit-vector Levenshtein: CLUSTER, 357 of the 367 s, is an upper bound - with the C edlib it would be a small fraction of that)"}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(batch, g_off, genome, params, eng=None, budget_s=20.0, parity_records=100_000):
    """The oracle (single-threaded C restatement of the reference algorithm, kind 'port') on a bounded, contiguous
    slice of the same batch (contiguous in coordinate order = full local coverage, so per-partition work is
    representative)."""
    from oracle import oracle as om
    orc = om.Oracle()
    orc.set_genome(g_off.cpu().numpy().astype(np.int64), genome.cpu().numpy())
    n = min(batch.n_rec, 4000)
    best = None
    for _ in range(4):
        hb = batch.slice_records(0, n)
        t0 = time.perf_counter()
        sig, _ = orc.collect(hb, params)
        t1 = time.perf_counter()
        ct = orc.cluster(params, hb.contig_rank, source=0)
        t2 = time.perf_counter()
        st = orc.stats()
        best = dict(n_rec=n, used=st["n_rec_used"], n_sig=sig.n, t_collect=t1 - t0, t_cluster=t2 - t1, n_clusters=ct.n,
                    edit_cells=st["n_edit_cells"], ops=st["n_ops"])
        if (t2 - t0) >= budget_s / 4 or n >= batch.n_rec:
            break
        n = min(batch.n_rec, n * 4)
    t = best["t_collect"] + best["t_cluster"]
    parity = None
    if eng is not None:
        # the bench line carries its own parity evidence: a slice of >= 100 k records through the HIP path (host batch this time) and through the oracle -
        # the oracle's pair distances on every granted CPU here (svo_set_threads: the checker, not the timed baseline above, which stays on ONE core)
        npar = min(batch.n_rec, max(parity_records, best["n_rec"]))
        hb = batch.slice_records(0, npar)
        t0 = time.perf_counter()
        orc.set_threads(_granted_cpus())
        try:
            sig, _ = orc.collect(hb, params)
            ct = orc.cluster(params, hb.contig_rank, source=0)
        finally:
            orc.set_threads(1)
        t_or = time.perf_counter() - t0
        gs, _ = eng.collect(hb, params)
        gc = eng.cluster(params, hb.contig_rank, source=0)
        parity = {"records": npar, "signatures": int(sig.n), "clusters": int(ct.n), "signatures_identical": gs.first_difference(sig) is None,
                  "clusters_identical": gc.first_difference(ct, rtol=1e-12) is None, "oracle_seconds_on_%d_threads" % _granted_cpus(): t_or}
    ref_py = reference_python_figure()
    return {"parity_vs_gpu_on_sample": parity, "value": best["used"] / t, "unit": "reads/s", "cores": 1, "kind": "port",
            "reference_python_reads_per_s": ref_py["reads_per_s"] if ref_py else None, "reference_python": ref_py,
            "reference_python_on_configs1_shape": reference_python_c1_shape(),
            "what": "oracle/svx_oracle.c: single-threaded C restatement of the reference's algorithm (a STRONGER baseline than the "
                    "reference's Python loops: tests/golden/g_c1.json.gz records 1.99 s + 4.75 s of reference Python for 10 k records; "
                    "the reference itself cannot travel to the GPU box)",
            "sample": "first %d records of the same batch (coordinate order): %d reads used, %d signatures, %d clusters; "
                      "collect %.2f s + cluster %.2f s on 1 host core (%d visible, %d granted by the cgroup CPU quota)" % (
                          best["n_rec"], best["used"], best["n_sig"], best["n_clusters"], best["t_collect"], best["t_cluster"],
                          os.cpu_count(), _granted_cpus()),
            "signatures_per_s": best["n_sig"] / max(best["t_cluster"], 1e-9),
            "cigar_ops_per_s": best["ops"] / max(best["t_collect"], 1e-9),
            "edit_cells_per_s": best["edit_cells"] / max(best["t_cluster"], 1e-9)}


def _granted_cpus():
    from svim_amd.harness import effective_cpus
    return effective_cpus()


def load_profile_json(name):
    try:
        with open(os.path.join(REPO, "profiles", name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def emit_strong(args, rank, world, local_rank, dev, use_dist, dist, opts, p, emit):
    out = measure_strong(args, rank, world, local_rank, dev, use_dist, dist, opts, p, args.steps, args.warmup)
    if use_dist:
        from svim_amd import multigpu as MG
        MG.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)


def measure_strong(args, rank, world, local_rank, dev, use_dist, dist, opts, p, steps, warmup, scale_default=0.2):
    """--scaling strong: BASELINE.json configs[3]'s shape - one coordinate-sorted whole-genome batch (svim_amd/workloads.py, the same on every rank: same seed),
    every rank collects the records of the coordinate WINDOWS it owns (multigpu.assign_windows, round 6: world - 1 cuts in (contig name, coordinate) order that
    balance the records - a histogram of their start coordinates per Mb, what a BAM index gives - and cut long contigs; the step moves every cut into a corridor
    no partition can straddle; the record runs are views of the resident batch) and the ranks cluster with svx_cluster's rank exchange; rank 0 gathers cluster
    rows + member lists (signature columns stay on their ranks: gather_signatures=False).  Total work does not depend on N and N=1 is the plain single-GPU step
    over the whole batch, so value(N) / value(1) IS the strong-scaling speedup.  The line carries what crossed the fabric in the last step (multigpu.WIRE).
    Returns the line (rank 0) / None; the process group stays up."""
    import torch
    from svim_amd import _lib, multigpu as MG, workloads
    name = args.workload if args.workload != "c1" else "c3"
    scale = args.scale if (args.scale != 1.0 or name != "c3") else scale_default      # c3 at full scale is a 30x human genome (4.6 M reads): the default is a fifth of it (~1 M reads)
    t0 = time.perf_counter()
    prof = workloads.profile(name, scale)
    batch, genome, g_off, meta = workloads.make_batch_full(prof, seed=3, device=dev)            # NOT seed + rank: every rank holds the same batch
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    refs = [c[0] for c in prof["contigs"]]
    lens = [int(x) for x in (g_off[1:] - g_off[:-1]).tolist()]
    n_contig = len(refs)
    order = sorted(range(n_contig), key=lambda i: refs[i])
    crank = np.zeros(n_contig, dtype=np.int32)
    crank[order] = np.arange(n_contig, dtype=np.int32)
    # proposed windows: equal numbers of RECORDS per rank, from the records' start coordinates per Mb (the batch is sorted by reference id, position)
    tid = batch.t["tid"][:batch.n_rec].to(torch.int64)
    pos = batch.t["pos"][:batch.n_rec].to(torch.int64)
    BIN = 1 << 20
    dens = []
    for k in range(n_contig):
        sel = pos[tid == k]
        nb = max(1, -(-lens[k] // BIN))
        dens.append(torch.bincount(sel // BIN, minlength=nb)[:nb].cpu().numpy().astype(np.float64) if sel.numel() else np.zeros(nb))
    windows = MG.assign_windows(refs, lens, world, weights=dens, bin_size=BIN)
    # this rank's record runs: maximal runs of consecutive records whose start lies in one of its windows
    mine = windows.owner_of_positions(tid.clamp_min(0), pos) == rank
    edge = torch.nonzero(mine[1:] != mine[:-1]).flatten() + 1
    bounds = [0] + [int(x) for x in edge.tolist()] + [batch.n_rec]
    runs = [(a, b) for a, b in zip(bounds[:-1], bounds[1:]) if b > a and bool(mine[a].item())]
    views = [batch.view_records(lo, hi) for lo, hi in runs]
    structs = [v.struct() for v in views]
    eng = _lib.Engine(local_rank)
    eng.set_genome(g_off, genome, on_device=True)
    adapter = MG.SvxAdapter(eng, dev)
    gid = np.arange(n_contig, dtype=np.int64)
    last = {}

    def step():
        used = sigs = ops = 0
        scan_ms = 0.0
        eng.accumulate(True)
        for b in structs:
            eng.set_slot_base(0)                       # the views keep the batch's own emission slots (2 per record in file order): already global
            eng.collect(b, p, fetch=False)
            st = eng.stats()
            used += st["n_rec_used"]; ops += st["n_ops"]; scan_ms += st["t_cigar_scan_ms"]
        eng.accumulate(False)
        sigs = eng.collect_counts()[0]
        if use_dist:
            MG.wire_reset()
            last["res"] = MG.cluster_step(adapter, p, rank, world, gid, crank, windows, key_base=0, read_base=0, gather_signatures=False)
        else:
            eng.cluster(p, crank, source=0, fetch=False)
        last.update(used=used, sigs=sigs, ops=ops, scan_ms=scan_ms, cluster_ms=eng.stats()["t_cluster_ms"], clusters=eng.stats()["n_clusters"])

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            MG.barrier()
        torch.cuda.synchronize()

    barrier()
    f0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_step_ms = 1e3 * (time.perf_counter() - f0)
    for _ in range(max(0, warmup - 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine = torch.tensor([last["used"], last["sigs"], last["ops"], len(runs), sum(hi - lo for lo, hi in runs), int(1e3 * last["scan_ms"]), int(1e3 * last["cluster_ms"]),
                             last["clusters"]], dtype=torch.int64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[int(x) for x in r.tolist()] for r in allr]
        wire = dict(MG.WIRE)
        backend = dist.get_backend()
        world_seen = dist.get_world_size()
    else:
        per_rank = [[last["used"], last["sigs"], last["ops"], len(runs), sum(hi - lo for lo, hi in runs), int(1e3 * last["scan_ms"]), int(1e3 * last["cluster_ms"]), last["clusters"]]]
        wire, backend, world_seen = None, None, 1
    eng.close()
    if rank != 0:
        return None
    tot_used = sum(r[0] for r in per_rank)
    res = last.get("res")
    cuts_used = getattr(res, "windows", None) or windows
    sig_rows = [r[1] for r in per_rank]
    out = {
        "metric": "aligned reads/sec through COLLECT+CLUSTER", "value": tot_used * steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32/i32 CIGAR + u8 bases, int64 positions, f64 distances", "data": "synthetic",
        "config": {"workload": "configs[3] stand-in (%s profile, svim_amd/workloads.py): ONE whole-genome batch of %d records on %d contigs (%.0f Mb), the same on every rank, "
                               "sharded by coordinate-window ownership; --scaling strong --scale %g (the driver's default line is configs[1], weak; with --gpus N > 1 it "
                               "carries this measurement as its strong_scaling block)" % (
                                   name, meta["n_records"], n_contig, meta["genome_bases"] / 1e6, scale),
                   "records_total": meta["n_records"], "cigar_ops_total": meta["n_ops"],
                   "parallelism": "1 process/GPU; coordinate windows sharded over ranks (consecutive ranges of the (contig name, coordinate) order, balanced by records, cuts "
                                  "inside contigs moved into corridors no partition straddles), every partition local; over the fabric per step: the merged signature stretches "
                                  "around the cuts, foreign rows (reads across a cut, BND / DUP_INT), the rank exchange of svx_cluster (all-gathers), cluster rows + member lists to rank 0"},
        "ownership": {"cuts": [[refs[int(c)], int(x)] for c, x in zip(cuts_used.cut_contig, cuts_used.cut_pos)],
                      "signatures_collected_per_rank": sig_rows,
                      "max_over_mean_signatures": (max(sig_rows) / (sum(sig_rows) / float(len(sig_rows)))) if sum(sig_rows) else None,
                      "max_over_mean_records": max(r[4] for r in per_rank) / (sum(r[4] for r in per_rank) / float(len(per_rank)))},
        "first_step_ms": first_step_ms, "synth_seconds": t_gen,
        "counts": {"reads_used": tot_used, "signatures": sum(r[1] for r in per_rank), "cigar_ops": sum(r[2] for r in per_rank),
                   "clusters_gathered": (res.n if res is not None else last["clusters"])},
        "per_rank": [dict(zip(("reads_used", "signatures", "cigar_ops", "contig_runs", "records", "k_cigar_scan_us", "cluster_us", "clusters"), r)) for r in per_rank],
        "multi_gpu": None if not use_dist else {
            "backend": backend + (" (= RCCL)" if backend == "nccl" else ""), "world_size_seen_by_the_process_group": world_seen,
            "fabric_last_step_rank0": {"collectives": wire["collectives"], "payload_bytes": wire["bytes"],
                                       "seconds_between_device_synchronises": wire["seconds"] if os.environ.get("SVX_WIRE_STATS") == "1" else None,
                                       "by_kind": {k: {"calls": v[0], "bytes": v[1], "seconds": v[2] if os.environ.get("SVX_WIRE_STATS") == "1" else None}
                                                   for k, v in wire["by_kind"].items()}},
            "note": "payload bytes as the wire sees them (padded slots x world for all-gathers, all ranks' padded slots for the gathers to rank 0); SVX_WIRE_STATS=1 times "
                    "every collective between two device synchronises (perturbs the step: off by default)"},
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=("c1", "c2", "c3", "c4"), default="c1")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="strong: ONE whole-genome batch (c3 profile unless c2 / c4 is named), the same on every rank, sharded by contig ownership - total work "
                         "is fixed and N=1 is the whole batch; weak (default, the driver's contract): every rank its own batch")
    ap.add_argument("--scale", type=float, default=1.0, help="c2 / c4: scale of the contig lengths (reads and sites follow)")
    ap.add_argument("--partition-max-distance", type=int, default=1000)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--n50", type=int, default=20000)
    ap.add_argument("--contig-len", type=int, default=250_000_000)
    ap.add_argument("--sites", type=int, default=None)
    ap.add_argument("--bam", default=None, help="coordinate-sorted BAM instead of a synthetic workload (needs --fasta)")
    ap.add_argument("--fasta", default=None)
    ap.add_argument("--foreign-frac", type=float, default=0.0, help="--gpus N, c2 / c4: this fraction of the cross-contig split-read segments points at a contig of "
                    "ANOTHER rank, so that BND signatures owned by other ranks (foreign rows) cross the fabric in the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--end-to-end-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--e2e-records-qual", type=int, default=300_000, help="end_to_end: records of the sample BAM with base qualities (the headline file)")
    ap.add_argument("--e2e-records", type=int, default=180_000, help="end_to_end: records of the sample BAM without qualities (side figure)")
    ap.add_argument("--e2e-chunk-mb", type=int, default=2048, help="end_to_end: inflated MB per chunk of the device reader on the headline file")
    ap.add_argument("--resident", type=float, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    # a GPU fault must end the run, not write a multi-GB GPU core dump first
    os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")
    try:
        import resource
        resource.setrlimit(resource.RLIMIT_CORE, (0, 0))
    except Exception:
        pass

    # stdout carries the ONE JSON line and nothing else: everything libraries write to fd 1 (RCCL prints a version banner there) goes to stderr
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if os.environ.get("SVX_BENCH_ONE_GPU") == "1":        # tests: every rank on cuda:0 (with SVX_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if args.end_to_end_child:
        from svim_amd import devsynth, harness
        batch, genome, meta = devsynth.make_batch(n_reads=args.reads, n50=args.n50, contig_len=args.contig_len, n_sites=args.sites, seed=2, device=dev)
        g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device=dev)
        emit(harness.end_to_end_sample(batch, g_off, genome, options(1000), device=local_rank, resident_reads_per_s=args.resident, n_records=args.e2e_records,
                                       n_records_qual=args.e2e_records_qual, chunk_mb=args.e2e_chunk_mb))
        return
    dist = None
    # SVX_BENCH_FORCE_DIST=1 exercises the multi-GPU code path (RCCL exchange + sharded clustering) with any world size
    use_dist = world > 1 or os.environ.get("SVX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # SVX_BENCH_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL refuses two ranks on the same device)
        dist.init_process_group(os.environ.get("SVX_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    from svim_amd import _abi, _lib, devsynth
    opts = options(args.partition_max_distance)
    p = _abi.Params.from_options(opts)
    if args.bam:
        from svim_amd import harness
        out = harness.run_bam(args.bam, args.fasta, opts, rank=rank, world=world, device=local_rank, steps=args.steps, warmup=args.warmup)
        if use_dist:
            from svim_amd import multigpu
            multigpu.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit(out)
        return
    if args.scaling == "strong" or args.workload == "c3":
        emit_strong(args, rank, world, local_rank, dev, use_dist, dist, opts, p, emit)
        return
    t0 = time.perf_counter()
    if args.workload == "c1":
        batch, genome, meta = devsynth.make_batch(n_reads=args.reads, n50=args.n50, contig_len=args.contig_len, n_sites=args.sites,
                                                  seed=2 + rank, device=dev)
        g_off = torch.tensor([0, genome.numel()], dtype=torch.int64, device=dev)
        label = "configs[1]: %d synthetic ONT reads per GPU (N50 %d), single %d Mb contig, DEL/INS/INV" % (
            args.reads, args.n50, args.contig_len // 1_000_000)
    else:
        from svim_amd import workloads
        prof = workloads.profile(args.workload, args.scale)
        batch, genome, g_off, meta = workloads.make_batch_full(prof, seed=3 + rank, device=dev)
        label = {"c2": "configs[2] stand-in: PacBio-HiFi profile (%d reads of 17.5 kb, %d contigs, %.0f Mb), full SV-type set incl. BND, "
                       "DUP_TAN, DUP_INT and split-read INS (edit-distance path)",
                 "c4": "configs[4] stand-in: 60x PacBio-CLR profile (%d reads, %d contig, %.0f Mb), dense sites, "
                       "--partition_max_distance %d" % (meta["n_reads"], meta["n_contig"], meta["genome_bases"] / 1e6, args.partition_max_distance)
                 }[args.workload]
        if args.workload == "c2":
            label = label % (meta["n_reads"], meta["n_contig"], meta["genome_bases"] / 1e6)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    eng = _lib.Engine(local_rank)
    rank_arr = batch.t["contig_rank"].cpu().numpy().astype(np.int32)
    if use_dist:
        # Contig-sharded layout (svim_amd/multigpu.py): rank r's contigs are "g<r>_<name>" - global contig id = r * n_local + local id,
        # name order is rank-major.  The batch carries GLOBAL contig ids; every rank holds the reference sequence of ITS contigs only.
        from svim_amd import multigpu as MG
        n_local = len(rank_arr)
        n_global = n_local * world
        names = ["g%03d_%s" % (r, nm) for r in range(world) for nm in (getattr(batch, "references", None) or ["chr1"])]
        order = sorted(range(n_global), key=lambda i: names[i])
        crank_global = np.zeros(n_global, dtype=np.int32)
        crank_global[order] = np.arange(n_global, dtype=np.int32)
        owner = np.repeat(np.arange(world, dtype=np.int32), n_local)
        base = rank * n_local
        batch.t["tid"] = batch.t["tid"] + base
        n_foreign_planted = 0
        if args.foreign_frac > 0 and world > 1 and batch.n_seg:
            # a cross-contig segment row (the partner of a BND) moves to the same-named contig of another rank: the signature then belongs to
            # whichever rank owns its canonical first end (src/svim/SVSignature.py:194) - a foreign row for one of the two
            gen = torch.Generator(device=dev)
            gen.manual_seed(77 + rank)
            counts = (batch.t["seg_off"][1:] - batch.t["seg_off"][:-1]).to(torch.int64)
            prim_tid = torch.repeat_interleave(batch.t["tid"].to(torch.int64), counts)
            st_local = batch.t["seg_tid"][:batch.n_seg].to(torch.int64)
            cross = (st_local != prim_tid) & (torch.rand(batch.n_seg, generator=gen, device=dev) < args.foreign_frac)
            other = (rank + 1 + torch.randint(0, world - 1, (batch.n_seg,), generator=gen, device=dev)) % world
            batch.t["seg_tid"][:batch.n_seg] = torch.where(cross, other * n_local + st_local - base, st_local - base).to(batch.t["seg_tid"].dtype)
            n_foreign_planted = int(cross.sum().item())
        batch.t["seg_tid"] = batch.t["seg_tid"] + base
        batch.t["contig_rank"] = torch.as_tensor(crank_global, device=dev)
        batch.n_contig = n_global
        rank_arr = crank_global
        g_off_global = torch.zeros(n_global + 1, dtype=torch.int64, device=dev)
        g_off_global[base:base + n_local + 1] = g_off
        g_off_global[base + n_local + 1:] = g_off[-1]
        eng.set_genome(g_off_global, genome, on_device=True)
        sizes = MG._all_gather_counts([batch.n_rec, int(batch.t["read_id"].max().item()) + 1], dev)
        key_base = 2 * sum(c[0] for c in sizes[:rank])
        read_base = sum(c[1] for c in sizes[:rank])
        adapter = MG.SvxAdapter(eng, dev)
        gid = np.arange(n_global, dtype=np.int64)
    else:
        eng.set_genome(g_off, genome, on_device=True)
    bstruct = batch.struct()

    last = {}

    def step():
        eng.collect(bstruct, p, fetch=False)
        if use_dist:
            last["res"] = MG.cluster_step(adapter, p, rank, world, gid, crank_global, owner, key_base=key_base, read_base=read_base)
            return
        eng.cluster(p, rank_arr, source=0, fetch=False)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            MG.barrier()
        torch.cuda.synchronize()

    # the very first pass of a fresh context: includes every device allocation; the library keeps NO tuning state between calls
    barrier()
    f0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first_step_ms = 1e3 * (time.perf_counter() - f0)
    for _ in range(max(0, args.warmup - 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(args.steps):
        step()
        s = eng.stats()
        for k, v in s.items():
            acc[k] = acc.get(k, 0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    st = eng.stats()
    avg = {k: acc[k] / args.steps for k in acc}                       # per-step averages over the timed region (HIP events inside libsvx)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([st["n_rec_used"], st["n_sig"], st["n_ops"]], dtype=torch.int64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_used, tot_sig, tot_ops = (int(x) for x in cnt.tolist())
        # With more than one rank the weak-scaling value is linear by construction; the line therefore also carries a STRONG-scaling measurement (one whole-genome
        # batch, configs[3]'s shape at a tenth of its size, sharded by coordinate windows) - measure_strong at N ranks; the same block at N=1 is the reference point
        strong_block = None
        try:
            sb = measure_strong(args, rank, world, local_rank, dev, True, dist, opts, p, max(2, min(args.steps, 3)), 1, scale_default=0.1)
            if rank == 0:
                strong_block = {k: sb[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "scaling", "config", "counts", "ownership", "multi_gpu")}
                strong_block["how_to_read"] = "value(N) / value(1) of THIS block is the strong-scaling speedup (same total work at every N); python bench.py --scaling strong --scale 0.1 gives value(1)"
        except Exception as e:                                        # the headline line stands whatever happens here
            strong_block = {"error": repr(e)}
        MG.barrier()
        dist.destroy_process_group()                                  # all ranks together, before rank 0 goes on alone
    else:
        strong_block = None
        tot_used, tot_sig, tot_ops = st["n_rec_used"], st["n_sig"], st["n_ops"]
    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    reads_per_s = tot_used * args.steps / elapsed
    # ---- roofline of the HBM-bound kernel (k_cigar_scan), algorithmic bytes per SURVEY.md section 8(d) ----
    n_ins = st["n_ins_bases"]
    scan_bytes = 32 * st["n_rec_used"] + 4 * st["n_ops"] + 16 * st["n_seg"] + 4 * st["n_seg_ops"] + 32 * st["n_sig"]
    bytes_collect = scan_bytes + n_ins // 2
    bytes_cluster = 36 * st["n_sig"] + 48 * st["n_clusters"] + st["n_hap_bytes"]
    scan_s = avg["t_cigar_scan_ms"] * 1e-3
    achieved = scan_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
    # HBM traffic of that kernel from the PMC counters (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction): measured
    # for the default workload and committed under profiles/ with the commit it was measured at; null for any other workload
    traffic = None
    tj = load_profile_json("traffic_k_cigar_scan.json")
    if tj and tj.get("workload_cigar_ops") == meta["n_ops"]:
        traffic = tj.get("traffic_bytes_per_launch")
    roofline_scan = {"kernel": "k_cigar_scan", "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": (tj or {}).get("source"),
                "algorithmic_bytes_per_launch": scan_bytes, "kernel_ms": avg["t_cigar_scan_ms"],
                "whole_path": {"algorithmic_bytes_per_step": bytes_collect + bytes_cluster,
                               "achieved_GBps": (bytes_collect + bytes_cluster) / (ms_per_step * 1e-3) / 1e9,
                               "frac_of_hbm": (bytes_collect + bytes_cluster) / (ms_per_step * 1e-3) / 8e12,
                               "note": "the path as a whole is integer-VALU bound in the edit-distance kernels: see roofline"}}
    # ---- roofline of the dominant kernels (k_edit_bands + k_edit_fulls): integer VALU issue ----
    edit_s = avg["t_edit_ms"] * 1e-3
    wc_issued, wc_useful = st.get("n_edit_wordcols_issued", 0), st.get("n_edit_wordcols_useful", 0)
    roofline_edit = None
    pj = load_profile_json("pmc_edit_kernels.json")
    if edit_s > 0 and wc_issued:
        instr = wc_issued / 64.0 * INSTR_PER_WORDCOL
        roofline_edit = {
            "kernel": "k_edit_bands<P> + k_edit_fulls<P> (all rounds; the window also holds pack/prep/sort/pilot): the dominant kernels of the step", "bound": "valu",
            "share_of_step": avg["t_edit_ms"] / ms_per_step, "traffic": None,
            "seconds": edit_s, "word_columns_executed": wc_issued, "word_columns_useful": wc_useful,
            "word_columns_retry_rounds": st.get("n_edit_wordcols_retry"), "retry_fraction": st.get("n_edit_wordcols_retry", 0) / wc_issued,
            "cells_executed": wc_issued * 32, "cells_full_matrix": st["n_edit_cells"],
            "wave_valu_instr_model": instr, "instr_per_word_column_model": INSTR_PER_WORDCOL,
            "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave64 VALU instructions/s (1024 SIMD x 2.4 GHz / 2)",
            "achieved": instr / edit_s, "frac": instr / edit_s / VALU_PEAK_WAVE_INSTR_PER_S,
            "frac_issue_cycles": (wc_issued / 64.0 * CYCLES_PER_WORDCOL) / (edit_s * 1024 * 2.4e9),
            "frac_useful_work_only": (wc_useful / 64.0 * CYCLES_PER_WORDCOL) / (edit_s * 1024 * 2.4e9),
            "frac_issue_cycles_at_measured_rates": (wc_issued / 64.0 * CYCLES_PER_WORDCOL_MEASURED_RATES) / (edit_s * 1024 * 2.4e9),
            "note": "frac counts 2 cycles per instruction for the 12 instructions of the update; 3 of them (v_addc_co of the three carry chains) are half rate: "
                    "frac_issue_cycles prices the 30 issue cycles per word-column, frac_issue_cycles_at_measured_rates the 33.9 the instruction classes measure "
                    "(2.3 / 4.4 cycles, profiles/r06_valu_banks.txt).  A step that skips word-columns (narrowing windows) lowers the time and the numerator alike: "
                    "frac says how well the issued work runs, word_columns_executed against cells_full_matrix how much of it was avoided",
            "band_speculation_fraction": st.get("edit_guess"),
            "gcups_executed": wc_issued * 32 / edit_s / 1e9, "gcups_full_matrix_equivalent": st["n_edit_cells"] / edit_s / 1e9,
        }
    if roofline_edit is not None and pj and pj.get("workload_cigar_ops") == meta["n_ops"]:
        # the same kernels under rocprofv3 --pmc (profiles/, measured at the commit named inside): instructions actually issued
        wc_ref = pj.get("word_columns_executed_at_measurement") or wc_issued       # the counters belong to the commit they were measured at, and so does its word-column count
        roofline_edit["pmc"] = {"wave_valu_instr": pj["wave_valu_instr_per_step"], "source": pj["source"], "commit": pj["measured_at_commit"],
                                "word_columns_executed_at_that_commit": wc_ref,
                                "valu_instr_per_word_column": pj["wave_valu_instr_per_step"] / (wc_ref / 64.0),
                                "frac": pj["wave_valu_instr_per_step"] / edit_s / VALU_PEAK_WAVE_INSTR_PER_S}
    kernels = {
        "k_cigar_scan_ms": avg["t_cigar_scan_ms"], "k_segments_ms": avg["t_segments_ms"], "collect_order_ms": avg["t_sort_ms"],
        "collect_gather_ms": avg["t_gather_ms"], "collect_total_ms": avg["t_collect_ms"],
        "cluster_partition_sample_ms": avg["t_partition_ms"], "cluster_edit_distance_ms": avg["t_edit_ms"],
        "cluster_linkage_ms": avg["t_linkage_ms"], "cluster_total_ms": avg["t_cluster_ms"],
        "edit_pairs": st["n_edit_pairs"], "edit_cells_full_matrix": st["n_edit_cells"], "pair_distances": st["n_pairs"],
        "bytes_collect_model": bytes_collect, "bytes_cluster_model": bytes_cluster,
        "edit_wordcols_issued": wc_issued, "edit_wordcols_useful": wc_useful,
        "edit_wordcols_retry_rounds": st.get("n_edit_wordcols_retry"), "edit_wordcols_band_kernels": st.get("n_edit_wordcols_band"),
        "edit_guess": st.get("edit_guess"),
    }
    cfg = {"workload": label, "records_per_gpu": meta["n_records"], "cigar_ops_per_gpu": meta["n_ops"], "planted_sites": meta["n_sites"],
           "parallelism": "1 process/GPU; contigs sharded over ranks, every partition local; per step over xGMI: foreign signatures (%s), "
                          "the random.sample stream positions by all-gathers only (sizes of the large partitions + per-rank transfer tables, no "
                          "rank waits for another rank's sampling), final gather of clusters + members + fixed-width signature columns to rank 0 "
                          "(svim_amd/multigpu.py)" % ("--foreign-frac %g" % args.foreign_frac if use_dist and args.foreign_frac > 0 else "none here"),
           "options": "SVIM alignment-mode defaults" + ("" if args.partition_max_distance == 1000 else ", partition_max_distance %d" % args.partition_max_distance)}
    if "reads_by_layout" in meta:
        cfg["reads_by_layout"] = meta["reads_by_layout"]
    out = {
        "metric": "aligned reads/sec through COLLECT+CLUSTER", "value": reads_per_s, "unit": "reads/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "scaling_note": "--gpus N is WEAK scaling: every rank generates its own batch of the same shape on its own contigs (seed = 2 + rank for c1, 3 + rank for "
                        "c2 / c4), so value = N x per-rank records / max-over-ranks time and only rank 0's batch is the N=1 batch; the ranks meet in the rank "
                        "exchange of svx_cluster and in the final gather (no hardware curve has been measured by the builder: one GPU per box)",
        "vs_baseline": None, "dtype": "u32/i32 CIGAR + u8 bases, int64 positions, f64 distances", "data": "synthetic",
        "config": cfg,
        "signatures_per_s": tot_sig * args.steps / elapsed,
        "signatures_per_s_cluster_only": st["n_sig"] / (avg["t_cluster_ms"] * 1e-3) if avg["t_cluster_ms"] > 0 else None,
        "first_step_ms": first_step_ms,
        "first_step_note": "first pass of a fresh context (every device allocation included); libsvx keeps no tuning state between calls - "
                           "the band speculation is chosen inside each call from a sample of its own pairs",
        "counts": {"reads_used": tot_used, "signatures": tot_sig, "cigar_ops": tot_ops, "partitions": st["n_partitions"],
                   "large_partitions": st["n_large_partitions"], "clusters": st["n_clusters"], "ins_bases": n_ins},
        # `roofline` describes the dominant kernels of the step (edit distance: integer VALU); the HBM-bound streaming kernel of COLLECT is `roofline_scan`
        "roofline": roofline_edit if roofline_edit is not None else roofline_scan, "roofline_scan": roofline_scan, "roofline_edit": roofline_edit,
        "kernels": kernels, "synth_seconds": t_gen,
    }
    if use_dist:
        out["strong_scaling"] = strong_block
        res = last.get("res")
        out["multi_gpu"] = {"foreign_segments_planted_rank0": n_foreign_planted, "signatures_per_rank": list(res.sig_counts) if res is not None else None,
                            "clusters_gathered": res.n if res is not None else None, "stream_end_rank0": res.chain_end if res is not None else None,
                            "rank_exchange_allgathers_last_step": getattr(getattr(adapter, "transport", None), "calls", 0)}
    if world == 1 and not use_dist:
        fetch = getattr(eng, "fetch_clusters", None)
        if fetch is not None:
            ct = eng.fetch_clusters()
            out["counts"]["clusters_by_type"] = dict(zip(_abi.TYPE_NAMES, [int(x) for x in ct.type_count]))
    if not args.no_end_to_end and world == 1 and not use_dist and args.workload == "c1":
        # The end-to-end sample (BAM file -> reader with GPU-assisted inflate -> pipeline) runs in a process of its own: whatever happens there,
        # the headline number above stands.  The child regenerates the same batch from its seed.
        import subprocess
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--end-to-end-child", "--reads", str(args.reads), "--n50", str(args.n50),
                   "--contig-len", str(args.contig_len), "--resident", repr(reads_per_s), "--e2e-records", str(args.e2e_records), "--e2e-records-qual", str(args.e2e_records_qual),
                   "--e2e-chunk-mb", str(args.e2e_chunk_mb)] + (["--sites", str(args.sites)] if args.sites else [])
            child = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, LOCAL_RANK=str(local_rank)))
            lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
            out["end_to_end"] = json.loads(lines[-1]) if child.returncode == 0 and lines else {"error": "exit %d: %s" % (child.returncode, child.stderr[-400:])}
        except Exception as e:
            out["end_to_end"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1 and not use_dist:           # the CPU baseline is a rank-0, N=1 measurement
        out["cpu_baseline"] = cpu_baseline(batch, g_off, genome, p, eng)
        out["speedup_vs_cpu_port"] = reads_per_s / world / out["cpu_baseline"]["value"]
        e2e = (out.get("end_to_end") or {}).get("bam_file_reads_per_s")
        if e2e:                                                            # the like-for-like pair: both sides start from the alignment records of a file
            out["speedup_vs_cpu_port_from_bam_file"] = e2e / out["cpu_baseline"]["value"]
            out["speedup_note"] = ("speedup_vs_cpu_port divides the HBM-resident rate by the port's rate on host arrays; speedup_vs_cpu_port_from_bam_file divides the "
                                   "BAM-file rate (inflate, record decode, COLLECT, CLUSTER) by the same port rate, which does not even include reading a file")
    emit(out)


if __name__ == "__main__":
    main()
