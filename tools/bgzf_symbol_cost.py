"""Cost per DEFLATE symbol of the GPU inflater: 2048 identical blocks (two waves per SIMD: latency, not throughput) of (a) literals only,
(b) maximal matches only, (c) short matches at long distances, (d) stored.  Usage: python tools/bgzf_symbol_cost.py"""
import os, sys, zlib, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svim_amd._lib import Inflater

rng = random.Random(1)
noise = bytes(rng.getrandbits(8) for _ in range(60000))
def deflate(raw, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
    return co.compress(raw) + co.flush()
pat = bytes(rng.getrandbits(8) for _ in range(3000))
short = bytearray(pat)
while len(short) < 60000:                     # 8-byte pieces copied from up to 3000 bytes back, separated by one fresh literal
    o = rng.randrange(len(short) - 2990, len(short) - 8)
    short += short[o:o + 8] + bytes([rng.getrandbits(8)])
skew = bytes(rng.choice(b"AAAACCGT!#5") for _ in range(60000))
cases = [("literals only (Huffman-only, 60000 symbols)", deflate(skew, 6, zlib.Z_HUFFMAN_ONLY), 60000, 60000),
         ("maximal matches (zeros: ~233 symbols)", deflate(bytes(60000)), 60000, 60000 // 258 + 2),
         ("8-byte matches + 1 literal (~13300 symbols)", deflate(bytes(short[:60000]), 9), 60000, 2 * (60000 // 9)),
         ("stored", deflate(noise, 0), 60000, 1)]
f = Inflater(0)
for name, comp, isize, nsym in cases:
    blocks = [(comp, isize)] * 2048
    ms = []
    for _ in range(3):
        got = f.inflate(blocks); ms.append(f.kernel_ms)
    assert got[:isize].tobytes() == zlib.decompress(comp, -15)
    t = min(ms)
    print("%-48s %6d B compressed: %.2f ms per block-wave -> %.0f ns per symbol, %.1f MB/s per wave" % (name, len(comp), t, 1e6 * t / nsym, isize / t / 1e3))
f.close()
