"""Multi-process soak of multigpu.DistStageTrainer on the bench inputs (gloo rehearsal on one GPU, or RCCL on a node):
N iterations of one stage, then: losses finite, and the working ranks' generator / discriminator replicas bit-identical.
usage: HPVG_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/soak_dist.py [stage] [iters]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from hp_vae_gan_amd import multigpu

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 5
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("HPVG_DIST_BACKEND", "nccl")
ndev = torch.cuda.device_count()
local = int(os.environ.get("LOCAL_RANK", rank))
dev = torch.device("cuda", local if backend == "nccl" else local % max(ndev, 1))
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
os.environ["HPVG_VAE_ON_RANK0"] = "0"
runner = multigpu.build_bench_runner(bench.video_opt, [stage], dev, rank, world)
out = None
for i in range(iters):
    runner.timed_stage(0)
out = runner.last.get(stage) or {}
torch.cuda.synchronize()
vals = {k: float(v) for k, v in out.items()}
assert all(math.isfinite(v) for v in vals.values()), (rank, vals)
# replica check over the working ranks: max - min of every parameter over the ranks must be exactly 0
s, trainer, real, rz = runner.items[0]
if trainer is not None and getattr(trainer, "is_gan", False):
    nwork = trainer.nwork
    flat = torch.cat([p.detach().reshape(-1) for p in list(trainer.netG.parameters()) + list(trainer.netD.parameters())])
    hi, lo = flat.clone(), flat.clone()
    if rank >= nwork:                      # idle ranks hold stale replicas: neutral elements
        hi.fill_(-float("inf")); lo.fill_(float("inf"))
    h, l = (hi.cpu(), lo.cpu()) if backend == "gloo" else (hi, lo)
    dist.all_reduce(h, op=dist.ReduceOp.MAX)
    dist.all_reduce(l, op=dist.ReduceOp.MIN)
    spread = float((h - l).abs().max())
    if rank == 0:
        print("stage", stage, "world", world, "mode", "oct" if trainer.nh == 2 else ("quad" if trainer.quad else "pair"), "iters", iters,
              {k: round(v, 4) for k, v in vals.items()}, "replica spread", spread)
    assert spread == 0.0, spread
dist.barrier()
dist.destroy_process_group()
