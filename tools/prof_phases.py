import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'kaldi-decoder_amd', 'python'))
import numpy as np, torch
from kaldi_decoder import cabi, synth
def run(kind):
    if kind=='hl':
        g=synth.hl_graph(); opts=dict(beam=16.0,max_active=3000); B=64
    else:
        g=synth.hlg_graph(V=500,n_words=20000,n_hist=20000,avg_succ=50,seed=2); opts=dict(beam=15.0); B=128
    T=250
    lps=[synth.make_logprobs(T,500,i,peak=12.0,lexicon=g['lexicon']) for i in range(B)]
    G=cabi.Graph(g); dev=torch.from_numpy(np.stack(lps)).cuda()
    bt=cabi.Batch(G,max_batch=B,max_frames=T,**opts)
    for _ in range(2):
        bt.init_decoding(); bt.advance_device([dev[i].data_ptr() for i in range(B)],[T]*B,500)
    cyc=bt.phase_cycles(); st=bt.stats()
    names=['0 frame start/ll row','1 GetCutoff','2 prepass','3 emit items+small','4 emit big loop','5 prune+seed',
           '6 eps round setup','7 eps items+small','8 eps big loop','9 resolve R1 + count','10 scan+min+alloc',
           '11 C1 write tokens','12 R2 + stats','13 clear/loop','14 spill-mode frames','15 aborted LDS attempts']
    tot=cyc.sum(0); allc=tot.sum()
    print(kind,'kernel_ms %.2f'%bt.last_kernel_ms(),'spill_frames',st[:,9].sum(),'cycles/utt-frame %.0f'%(allc/(B*T)))
    for n,v in zip(names,tot): print('  %-26s %5.1f%%  %8.0f cyc/utt-frame'%(n,100*v/allc,v/(B*T)))
    tcyc=cyc.sum(1); print('  per-utt Mcyc: min %.1f med %.1f max %.1f'%(tcyc.min()/1e6,np.median(tcyc)/1e6,tcyc.max()/1e6))
run('hl'); run('hlg')
