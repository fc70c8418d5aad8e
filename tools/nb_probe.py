"""Development-only: the level-0 neighbour build / query / cell order of the synthetic sphere through the device-lens entry,
N times per variant of MVK_NB_DBG (run under rocprofv3 --kernel-trace; tools/trace_by_grid.py lists the launches)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mvkpconv
ops, syn = mvkpconv.sub("ops"), mvkpconv.sub("synthetic")
dev = torch.device("cuda:0")
staged = syn.stage_spheres([syn.raw_sphere(seed=0)], dev, None)
p = (staged['points'][0] - staged['center'][0]).contiguous()
n = p.shape[0]
cap = 20480
pp = torch.zeros(cap, 3, device=dev); pp[:n] = p
lens = torch.tensor([n], dtype=torch.int32, device=dev)
status = torch.zeros(4, dtype=torch.int32, device=dev)
out = torch.empty(cap, 58, dtype=torch.int32, device=dev)
order = torch.empty(cap, dtype=torch.int32, device=dev)
for dbg in sys.argv[1:] or ["0"]:      # (MVK_NB_DBG: a development switch that existed while the cell order was taken apart)
    os.environ["MVK_NB_DBG"] = dbg
    for _ in range(10):
        ops.radius_neighbors_dev(pp, pp, lens, lens, 0.1, out, cap, status)
        ops.neighbors_cell_order(cap, cap, 1, order, lens)
    torch.cuda.synchronize()
    print("dbg", dbg, "done")
