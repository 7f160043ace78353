"""Ad-hoc GPU probe (not a test): one stream run as a block set of ONE (the block-mode path overlaps the pre-pass and the frame coder of consecutive launches; the
single-stream path does not).  python tests/gpu_one_as_block.py [mb=300] [window=28]"""
import hashlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, '.')
import nlzm_amd
from nlzm_amd import corpus
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
hb = int(sys.argv[2]) if len(sys.argv) > 2 else 28
nlzm_amd.init(0)
data = corpus.syn_text(int(mb * 1e6))
nlzm_amd.set_option("batch_chunks", 8)
t = time.time(); one = nlzm_amd.compress(data, hb); t_one = time.time() - t
print(f"single stream: {t_one:.2f} s wall, {nlzm_amd.timing()}", flush=True)
for k, v in (("block_parser_helper", 1), ("block_worker_threads", 128), ("block_hot_waves", 2), ("block_batch_chunks", 8)):
    nlzm_amd.set_option(k, v)
for rep in range(2):
    t = time.time(); blk = nlzm_amd.compress_blocks(data, 1, hb); t_blk = time.time() - t
    print(f"block set of one (run {rep}): {t_blk:.2f} s wall; same bytes: {blk[0] == one}", flush=True)
print(len(one), hashlib.sha256(one).hexdigest()[:16])
