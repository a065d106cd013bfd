import os, sys, cProfile, pstats, io, time
sys.path.insert(0, "/root/repo")
import torch, bench
from insmos_amd import params as P
from insmos_amd.models import InsMOSNet
cfg = P.default_cfg(); sd = P.random_state_dict(cfg, seed=0)
pts = torch.from_numpy(bench.load_window(0, 1886)).cuda()
model = InsMOSNet(cfg, state_dict=sd).cuda(0).eval()
bench.calibrate_head(model, pts, 1500)
eng = model.model.engine
for _ in range(3): eng.forward_window(pts)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20): eng.forward_window(pts)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(22); print(s.getvalue()[:6000])
