"""Worker of test_sync_batchnorm_two_ranks_equal_one_rank_with_two_spheres (tests/test_model_gpu.py): one rank of a
two-rank gloo group, both on cuda:0, each with ONE sphere of the batch the parent ran stacked on a single rank.
usage: python _syncbn_worker.py RANK WORLD PORT WORKDIR"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, work = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mvkpconv
    syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    meta = torch.load(os.path.join(work, "meta.pt"), weights_only=False)
    torch.manual_seed(0)
    np.random.seed(0)
    cfg = syn.make_config(meta["variant"])
    sph = [syn.raw_sphere(seed=meta["seeds"][rank], radius=meta["radius"], density=meta["density"])]
    views = [syn.sphere_views(s, nv=3, h=60, w=80) for s in sph]
    staged = syn.stage_spheres(sph, dev, views)
    rots = [r[rank:rank + 1] for r in meta["rotations"]]
    batch, lens = syn.build_batch(cfg, staged, meta["limits"], torch.int64, rotations=rots)
    net = syn.build_model(cfg, dev)
    net.load_state_dict(meta["state"])
    net.train()
    for m in net.net_2d._modules.values():
        m.train(False)
    ops.set_sync_batchnorm(True)
    out = net(batch, cfg)
    # the stacked run averages the cross entropy over ALL points: weight this rank's mean by its share of them
    n_all = torch.tensor([float(lens[0])], device=dev)
    dist.all_reduce(n_all)
    loss = net.loss(out, batch.labels) * (lens[0] / n_all.item())
    loss.backward()
    total = loss.detach().clone()
    dist.all_reduce(total)
    grads = {}
    for n, p in net.named_parameters():
        if p.grad is not None:
            g = p.grad.detach().clone()
            dist.all_reduce(g)                       # sum over ranks = gradient of the all-points mean
            grads[n] = g.cpu()
    torch.save({"logits": out.detach().cpu(), "loss": total.cpu(), "grads": grads if rank == 0 else None, "n": lens[0],
                "running_mean": net.encoder_blocks[3].batch_norm_conv.batch_norm.running_mean.cpu()},
               os.path.join(work, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
