import sys, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import raider_amd as R
from raider_amd.synthetic import synthetic_cube, scene_grid
dev = torch.device('cuda')
ctx = R.Context.default(); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
c = synthetic_cube(300, 300, 80, seed=0)
tot = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet_total']).to(dev), torch.from_numpy(c['hydro_total']).to(dev), order='zyx')
x2, y2, _, _ = scene_grid(1000, 1000)
xt, yt = torch.from_numpy(x2).to(dev), torch.from_numpy(y2).to(dev)
zt = torch.from_numpy(c['zs'][:40].copy()).to(dev)
ow = torch.empty((40, 1000, 1000), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
for _ in range(4):
    tot.build_cube(xt, yt, zt, out=(ow, oh))
torch.cuda.synchronize()
