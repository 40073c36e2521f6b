"""End-to-end soak of the product path the way the reference's main loop drives it (train_video.py:396-417): ONE generator
grown stage by stage (init_next_stage), a fresh StageTrainer + discriminator per stage (D warm-started from the previous
stage's checkpoint), train() with its eager->hipGraph switch, checkpoint save / load round trip between stages, on a
synthetic clip served by SingleVideoDataset.  Checks: finite losses and parameters at every stage, checkpoints load back
bit-exactly.
usage: python tools/e2e_soak.py [last_stage] [niter]"""
import math
import os
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from hp_vae_gan_amd import checkpoint, datasets, train as hp_train, utils as hu
from hp_vae_gan_amd.modules import networks_3d

last_stage = int(sys.argv[1]) if len(sys.argv) > 1 else 5
niter = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(0)
opt = bench.video_opt(dev, niter=niter)
hu.adjust_scales2image(opt.img_size, opt)
opt.stop_scale_time = opt.stop_scale
opt.Noise_Amps = []
rng = np.random.default_rng(0)
# the 13-frame 256x144 clip of BASELINE configs[2] (smooth synthetic content: low-pass noise), --data-rep 2 as SURVEY 8d notes
base = rng.standard_normal((13, 18, 32, 3))
frames = np.clip(np.kron(base, np.ones((1, 8, 8, 1))) * 60 + 128, 0, 255).astype(np.uint8)
opt.frames, opt.data_rep, opt.hflip, opt.max_frames, opt.start_frame = frames, 2, False, 13, 0
ds = datasets.SingleVideoDataset(opt)
netG = networks_3d.GeneratorHPVAEGAN(opt).to(dev)
with tempfile.TemporaryDirectory() as ckdir:
    for s in range(last_stage + 1):
        opt.scale_idx = s
        if s > 0:
            netG.init_next_stage()
            netG.to(dev)
        opt.fps, opt.td, opt.fps_index = hu.get_fps_td_by_index(s, opt)
        ds.generate_frames(s)
        items = [ds[i] for i in range(opt.batch_size)]                      # what DataLoader(batch_size=2) would collate
        data = [tuple(torch.stack([it[j] for it in items]) for j in range(2))] if s > 0 else [torch.stack(items)]
        netD = None
        if opt.vae_levels < s + 1:
            netD = networks_3d.WDiscriminator3D(opt).to(dev)
            if os.path.exists(os.path.join(ckdir, "netD_%d.pth" % (s - 1))):
                checkpoint.warm_start_discriminator(netD, ckdir, s, map_location=dev)
        tr = hp_train.train(opt, netG, data, netD=netD)
        torch.cuda.synchronize()
        sc = {k: float(v) for k, v in tr.last.items() if torch.is_tensor(v) and v.numel() == 1}
        sc["clip_norm"] = float(tr.last["clip_info"][1])
        assert all(math.isfinite(v) for v in sc.values()), (s, sc)
        assert all(bool(torch.isfinite(v.float()).all()) for v in netG.state_dict().values()), s
        checkpoint.save_stage(ckdir, opt, tr)
        back = torch.load(os.path.join(ckdir, "netG.pth"), weights_only=True)
        assert back["scale"] == s and all(torch.equal(v, netG.state_dict()[k].cpu()) for k, v in back["state_dict"].items())
        print("stage", s, "iters", tr.iteration, "graph" if getattr(tr, "_graph", None) is not None else "eager",
              {k: round(v, 4) for k, v in sc.items()}, "amps", [round(float(a), 4) for a in opt.Noise_Amps])
print("e2e ok")
