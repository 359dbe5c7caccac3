"""Where the cycles of k_bgzf_inflate go on the sample BAM: needs a library built with -DINF_PROFILE
(tools/build_variants.sh prof="-DINF_PROFILE"; SVX_LIB=svim_amd/variants/libsvx_prof.so python tools/inflate_profile.py [n_records])."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                        # noqa: E402
from svim_amd import devsynth, harness                    # noqa: E402
from svim_amd import _lib                                 # noqa: E402
from svim_amd._lib import Inflater, bgzf_blocks           # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
qual = 7 if len(sys.argv) > 2 and sys.argv[2] == "qual" else None           # "qual": the sample carries base qualities (literal-heavy blocks)
b, genome, meta = devsynth.make_batch(n_reads=max(n, 1000), n50=20000, contig_len=max(3_000_000, 250 * n), seed=2, device="cuda:0")
hb = b.slice_records(0, min(n, b.n_rec))
path = "/tmp/bgzf_prof.bam"
nrec, raw = harness.write_bam_from_batch(path, hb, ["chr1"], [int(genome.numel())], qual_seed=qual)
print("sample: %d records%s" % (nrec, ", WITH base qualities (random Phred values)" if qual else ", QUAL absent (0xff)"))
blocks = bgzf_blocks(path)
out_bytes = sum(s for _, s in blocks)
lib = _lib.lib()
fn = lib.svx_inflate_profile
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
f = Inflater(0)
f.inflate(blocks)
fn(None, 1)
f.inflate(blocks)
ms = f.kernel_ms
arr = (ctypes.c_ulonglong * 16)()
fn(arr, 1)
v = np.array(list(arr), dtype=np.float64)
names = ["window gather", "literal chain + write", "token decode (lanes)", "token chain", "emission, one byte per lane", "emission, match by match", "serial token",
         "block header + tables", "input refill + ring flush"]
cyc = v[:9]
print("%d blocks, %.1f MB inflated, kernel %.2f ms = %.1f GB/s" % (len(blocks), out_bytes / 1e6, ms, out_bytes / ms / 1e6))
print("cycles per block-wave: %.0f (sum over the slots); share per slot:" % (cyc.sum() / len(blocks)))
for nm, c in zip(names, cyc):
    print("  %-32s %5.1f %%   %9.0f cycles per block" % (nm, 100 * c / cyc.sum(), c / len(blocks)))
steps, stepsB, serial, seq, lits, toks, nbytes = v[9], v[10], v[11], v[12], v[13], v[14], v[15]
print("per block: %.0f steps (%.0f with whole-token decode, %.0f of those emitted match by match), %.0f serial tokens, %.0f literals from the literal chain, %.0f tokens from the token chain, %.0f bytes" % (
    steps / len(blocks), stepsB / len(blocks), seq / len(blocks), serial / len(blocks), lits / len(blocks), toks / len(blocks), nbytes / len(blocks)))
print("cycles per step %.0f; bytes per step %.1f" % (cyc.sum() / max(1, steps), nbytes / max(1, steps)))
f.close()
