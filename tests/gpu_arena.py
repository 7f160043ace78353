"""Ad-hoc: the pair-list arena of a block set with corpus.spines, default and tiny (python tests/gpu_arena.py [ext_blocks])"""
import sys
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus, shard
from tests import oracle_py
nlzm_amd.init(0)
nlzm_amd.set_option("stage_report", 1)
data = corpus.spines(3_000_000, corpus.SEED + 26)
k, hb = 4, 20
want = [oracle_py.compress(data[slice(*shard.block_range(data.size, k, i))], hb) for i in range(k)]
print("single stream of block 0:", nlzm_amd.compress(data[:750_000], hb) == want[0], flush=True)
for ext in [int(a) for a in sys.argv[1:]] or [-1, 1]:
    nlzm_amd.set_option("block_ext_blocks", ext)
    got = nlzm_amd.compress_blocks(data, k, hb)
    print("ext", ext, [g == w for g, w in zip(got, want)], "redo", nlzm_amd.counter("block_redo_streams"), flush=True)
