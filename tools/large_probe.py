import os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "kaldi-decoder_amd", "python")]
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl = bench.WORKLOADS["hlg-large"]
t = time.time(); g = synth.hlg_graph(**wl["graph"]); print("graph %.1f s" % (time.time() - t), flush=True)
B = int(sys.argv[1]); T, V = 250, 500
lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
G = cabi.Graph(g); dev = torch.from_numpy(np.stack(lps)).cuda()
ptrs = [dev[i].data_ptr() for i in range(B)]
bt = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **wl["opts"])
for it in range(3):
    t = time.time()
    bt.decode_device(ptrs, [T] * B, V)
    print("call %d: %.1f ms kernel %.1f ms" % (it, (time.time() - t) * 1e3, bt.last_kernel_ms()), bt.capacity(), flush=True)
st = bt.stats().sum(0)
print("per frame: tokens in %.0f, arcs %.0f, links %.0f, max tokens %d, spill frames %d" % (st[1] / st[0], (st[3] + st[6]) / st[0], st[10] / st[0], bt.stats()[:, 8].max(), st[9]))
