"""tools/tapc_probe.py -- CPU probe behind csrc/spconv_tapc.hip: MFMA passes issued / useful on the 4D levels for 16-row tiles (active (group, tap)
slots) against per-tap compaction of the rows of an R-row block into dense groups of 16 (oracle tables of a quarter-size S0 window)."""
import sys, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.synth import make_window
from oracle import ref_ops as R
w = make_window(seed=0, n_scans=10, n_az=472)
c0,k0,_ = R.me_quantize(w[:, [0,1,2,4]], [0.1]*4)
for lvl in (1,2,3):
    c,k,_ = R.me_stride_down(c0,k0,lvl)
    offs = R.me_kernel_offsets([3,3,3,3],[1<<lvl]*3+[1])
    nbr = R.me_nbr(c,k,offs)
    n = len(c)
    pres = nbr >= 0    # (81, n)
    pairs = pres.sum()
    ng = n//16
    act = pres[:, :ng*16].reshape(81, ng, 16).any(axis=2)
    print("level",lvl,"rows",n,"valid taps/row %.2f"%(pairs/n),"active slots/16-group %.2f"%act.sum(0).mean(), "-> MFMA waste x%.2f"%(act.sum()*16/pres[:, :ng*16].sum()))
    for Rr in (32,64,128,256,512):
        nb = n//Rr
        cnt = pres[:, :nb*Rr].reshape(81, nb, Rr).sum(axis=2)   # (81, nb)
        grp = (cnt+15)//16
        print("   R=%d: all taps: MFMA groups x16 / pairs = %.3f ; groups/tap-visit %.2f; nonempty taps/block %.1f"%(Rr, grp.sum()*16/cnt.sum(), grp.sum()/(cnt>0).sum(), (cnt>0).sum(0).mean()))
        # tap classes k%4 each separately compacted over the same R rows: same thing (classes partition taps). 
