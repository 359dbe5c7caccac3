#!/usr/bin/env python3
"""Turns the PMC summaries of tools/gpu_profiles.sh (gpurun_out/<tag>_pmc_*.csv, copied to profiles/) into the two small JSON files bench.py reads:
   profiles/traffic_k_cigar_scan.json   HBM traffic of k_cigar_scan per launch (FETCH_SIZE + WRITE_SIZE passes, gfx950 correction)
   profiles/pmc_edit_kernels.json       SQ counters of k_edit_bands + k_edit_fulls per bench step
Usage: python tools/make_profile_json.py <tag> <commit> <steps in the PMC runs> <cigar ops of the workload> [word-columns the edit kernels issued per step at that commit]"""
import csv
import json
import os
import sys

tag, commit, steps, n_ops = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
wc_at_commit = int(float(sys.argv[5])) if len(sys.argv) > 5 else None        # roofline_edit.word_columns_executed of the bench line at that commit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def rows(name):
    with open(os.path.join(root, name)) as fh:
        return list(csv.DictReader(fh))


def per_dispatch(name, kernel, counter):
    for r in rows(name):
        if r["Kernel"].strip('"') == kernel and r["Counter"] == counter:
            return float(r["Total"]) / int(r["Dispatches"]), int(r["Dispatches"])
    return None, 0


fetch_kb, nf = per_dispatch("%s_pmc_FETCH_SIZE.csv" % tag, "k_cigar_scan", "FETCH_SIZE")
write_kb, nw = per_dispatch("%s_pmc_WRITE_SIZE.csv" % tag, "k_cigar_scan", "WRITE_SIZE")
traffic = {"kernel": "k_cigar_scan", "measured_at_commit": commit,
           "source": "profiles/%s_pmc_FETCH_SIZE.csv + profiles/%s_pmc_WRITE_SIZE.csv (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, two separate passes of "
                     "`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end`, %d dispatches each)" % (tag, tag, nf),
           "workload_cigar_ops": n_ops, "fetch_size_kb": fetch_kb, "write_size_kb": write_kb,
           "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE uncalibrated, taken as is (KB = 1024 B)",
           "traffic_bytes_per_launch": int(2 * fetch_kb * 1024 + (write_kb or 0) * 1024)}
# round 6: k_scan_prepare (the item table of the scan; it sits inside the HIP-event bracket of the bench line's k_cigar_scan_ms) - its own, smaller traffic beside the scan's
pf, _ = per_dispatch("%s_pmc_FETCH_SIZE.csv" % tag, "k_scan_prepare", "FETCH_SIZE")
pw, _ = per_dispatch("%s_pmc_WRITE_SIZE.csv" % tag, "k_scan_prepare", "WRITE_SIZE")
if pf is not None:
    traffic["k_scan_prepare"] = {"fetch_size_kb": pf, "write_size_kb": pw, "traffic_bytes_per_launch": int(2 * pf * 1024 + (pw or 0) * 1024),
                                 "note": "FETCH_SIZE doubled like the scan's (an upper bound for its narrower loads)"}
with open(os.path.join(root, "traffic_k_cigar_scan.json"), "w") as fh:
    json.dump(traffic, fh, indent=2)
    fh.write("\n")

edit = {"measured_at_commit": commit, "bench_steps_in_run": steps, "workload_cigar_ops": n_ops,
        "source": "profiles/%s_pmc_SQ_INSTS_VALU_SQ_WAVE_CY.csv + profiles/%s_pmc_SQ_ACTIVE_INST_VALU_SQ_I.csv (rocprofv3 --kernel-trace --pmc ..., totals over the run "
                  "divided by its %d bench steps)" % (tag, tag, steps), "per_step": {}}
# every instantiation of the two fused launches (k_edit_bands<P, QM>: P bit planes, QM = widest window the kernel is built for; k_edit_fulls<P>)
kerns = set()
for name in ("%s_pmc_SQ_INSTS_VALU_SQ_WAVE_CY.csv" % tag, "%s_pmc_SQ_ACTIVE_INST_VALU_SQ_I.csv" % tag):
    for r in rows(name):
        k = r["Kernel"].strip('"')
        if "k_edit_bands<" in k or "k_edit_fulls<" in k:
            kerns.add(k)
for kern in sorted(kerns):
    d = {}
    for name, ctrs in (("%s_pmc_SQ_INSTS_VALU_SQ_WAVE_CY.csv" % tag, ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")),
                       ("%s_pmc_SQ_ACTIVE_INST_VALU_SQ_I.csv" % tag, ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"))):
        for r in rows(name):
            if r["Kernel"].strip('"') == kern and r["Counter"] in ctrs:
                d[r["Counter"]] = float(r["Total"]) / steps
    edit["per_step"][kern.replace("void ", "")] = d
tot = sum(v.get("SQ_INSTS_VALU", 0) for v in edit["per_step"].values())
edit["wave_valu_instr_per_step"] = tot
if wc_at_commit:
    edit["word_columns_executed_at_measurement"] = wc_at_commit
edit["note"] = "GRBM_GUI_ACTIVE sums the 8 XCDs; band and full-matrix launches overlap in time, so their active cycles do not add up"
with open(os.path.join(root, "pmc_edit_kernels.json"), "w") as fh:
    json.dump(edit, fh, indent=2)
    fh.write("\n")
print(json.dumps(traffic["traffic_bytes_per_launch"]), tot)
