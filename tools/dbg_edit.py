import random, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from svim_amd import _lib, synth
from oracle import oracle as om
eng = _lib.Engine(0); orc = om.Oracle()
rng = random.Random(3)
def mk(la, lb, sim):
    a = synth.random_seq(rng, la)
    if sim:
        b = list(a)
        for _ in range(max(1, la // 15)):
            p = rng.randrange(len(b)); r = rng.random()
            if r < 0.4: b[p] = rng.choice("ACGTN")
            elif r < 0.7: del b[p]
            else: b.insert(p, rng.choice("ACGT"))
        b = "".join(b)
    else:
        b = synth.random_seq(rng, lb)
    return a, b
cases = [(5000, 300, False), (1, 5000, False), (300, 5000, False), (4000, 4100, False), (64, 700, False), (200, 210, True), (2000, 2000, True), (600, 600, False)]
pairs = [mk(*c) for c in cases]
exp = [orc.edit_distance(a, b) for a, b in pairs]
print("batch :", eng.edit_distances(pairs))
print("single:", [eng.edit_distances([p])[0] for p in pairs])
print("expect:", exp)
