import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from insmos_amd.synth import make_window
from oracle import ref_ops as R
w = make_window(seed=0, n_scans=10, n_az=472)
c0,k0,_ = R.me_quantize(w[:, [0,1,2,4]], [0.1]*4)
for lvl in (2,1):
    c,k,_ = R.me_stride_down(c0,k0,lvl)
    offs = R.me_kernel_offsets([3,3,3,3],[1<<lvl]*3+[1])
    nbr = R.me_nbr(c,k,offs)
    n = len(c); ng = n//16
    act = (nbr[:, :ng*16].reshape(81, ng, 16) >= 0).any(axis=2)   # (81, groups)
    print("level",lvl,"groups",ng,"active slots/group",act.sum(0).mean())
    for JT in (1,2):
      for NW in (4,8):
        for PH in (1,2,4,8,81):
            G = JT*NW
            nwg = ng//G
            a = act[:, :nwg*G].reshape(81, nwg, NW, JT).sum(axis=3)   # items per wave per tap
            # phases
            nph = (81+PH-1)//PH
            pad = nph*PH-81
            a2 = np.concatenate([a, np.zeros((pad,)+a.shape[1:], a.dtype)],0).reshape(nph, PH, nwg, NW).sum(axis=1)  # (nph,nwg,NW)
            tot = a2.sum()
            cost = a2.max(axis=2).sum()*NW
            print(f"  JT={JT} waves={NW} phase={PH:2d} taps: utilisation {tot/cost:.3f}")
