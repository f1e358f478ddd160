import os, sys
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path[:0]=[R, R+"/kaldi-decoder_amd/python"]
import numpy as np, torch
from kaldi_decoder import cabi, synth
import bench
wl=bench.WORKLOADS["hlg-lattice"]; g=synth.hlg_graph(**wl["graph"]); G=cabi.Graph(g)
for utt in (0,1,2):
    lp=synth.make_logprobs(250,500,utt,peak=12.0,lexicon=g["lexicon"]); dev=torch.from_numpy(lp).cuda()
    for pi in (25, 10, 10**6):
        lb=cabi.LatticeBatch(G,max_batch=1,max_frames=250,**{**wl["opts"],"prune_interval":pi})
        lb.decode_device([dev.data_ptr()],[250],500)
        c=lb.capacity(); st=lb.stats()[0]
        print(utt, pi, 'tokens_used',c['tokens_used'],'links_used',c['links_used'],'tokens_out',st[5],'links',st[10],'max',st[8], 'kernel %.2f'%lb.last_kernel_ms())
