"""40 eager iterations of train.BaselineStageTrainer for each baseline generator / critic combination on synthetic inputs:\nlosses stay finite and the reconstruction loss falls.  usage: python tools/soak_baselines.py"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from hp_vae_gan_amd import train as hp_train, utils as hu
from hp_vae_gan_amd.modules import networks_3d
dev = torch.device("cuda", 0)
for gen, crit in (("GeneratorSG", "WDiscriminator3D"), ("GeneratorCSG", "WDiscriminator3D"), ("GeneratorSG", "WDiscriminatorBaselines")):
    torch.manual_seed(0)
    opt = bench.video_opt(dev, Dsteps=1, Gsteps=1, alpha=10.0, train_depth=1, nfc=32)
    hu.adjust_scales2image(opt.img_size, opt); opt.stop_scale_time = opt.stop_scale
    opt.scale_idx = 2; opt.Noise_Amps = [1, 0.05]
    opt.discriminator = crit
    netG = getattr(networks_3d, gen)(opt)
    for _ in range(2): netG.init_next_stage()
    netG.to(dev)
    netD = getattr(networks_3d, crit)(opt).to(dev)
    shape = hu.images.level_shape_3d(2, opt)
    real = (torch.rand(2, 3, *shape, device=dev) * 2 - 1)
    tr = hp_train.BaselineStageTrainer(opt, netG, netD)
    hist = []
    for i in range(40):
        out = tr.step(real)
        if i % 10 == 9:
            torch.cuda.synchronize()
            sc = {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
            assert all(math.isfinite(v) for v in sc.values()), (gen, crit, i, sc)
            hist.append(round(sc["rec_loss"], 4))
    print(gen, crit, "rec_loss every 10:", hist)
