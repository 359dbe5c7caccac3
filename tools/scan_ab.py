"""k_cigar_scan alone on the configs[1] batch: t_cigar_scan_ms of repeated COLLECT passes for every library variant / environment setting given.
Usage: python tools/scan_ab.py name=libpath[,ENV=VAL...] ..."""
import os, sys, types, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from svim_amd import _abi, _lib, devsynth
    o = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, segment_gap_tolerance=10, segment_overlap_tolerance=5,
                              partition_max_distance=1000, position_distance_normalizer=900, edit_distance_normalizer=1.0, cluster_max_distance=0.5, all_bnds=False)
    p = _abi.Params.from_options(o)
    b, genome, meta = devsynth.make_batch(n_reads=1000000, contig_len=250_000_000, seed=2, device="cuda:0")
    bs = b.struct()
    eng = _lib.Engine(0)
    ts = []
    for _ in range(6):
        eng.collect(bs, p, fetch=False)
        ts.append(eng.stats()["t_cigar_scan_ms"])
    print(json.dumps({"scan_ms": [round(t, 3) for t in ts], "n_sig": eng.stats()["n_sig"], "n_rec": int(b.n_rec), "n_seg": int(b.n_seg),
                      "recs_with_segments": int(((b.t["seg_off"][1:] > b.t["seg_off"][:-1]) & ((b.t["flag"].long() & 2048) == 0)).sum().item())}))
    sys.exit(0)
for spec in sys.argv[1:]:
    name, rest = spec.split("=", 1)
    parts = rest.split(",")
    env = dict(os.environ)
    if parts[0]:
        env["SVX_LIB"] = os.path.abspath(parts[0])
    for kv in parts[1:]:
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(name, line[-1] if line else ("FAILED " + out.stderr[-300:]))
