"""Checkpoint dictionaries in the reference's format (train_video.py:246-258, utils/saver.py:39-45) and its resume
rule (train_video.py:399-412, 47-52).  Files are plain torch.save dicts:

  netG.pth        {'scale', 'state_dict', 'optimizer', 'noise_amps'}
  netD_<s>.pth    {'scale', 'state_dict', 'optimizer'}
  Noise_Amps.pth  {'data': [...]}

state_dict keys are identical to the reference's and 'optimizer' is torch.optim.Adam's own
state_dict layout (optim.FlatAdam.state_dict: per-parameter step / exp_avg / exp_avg_sq on the CPU, parameters numbered
group by group), so checkpoints interchange in both directions (the reference never reloads the optimizer state:
train_video.py:399-412 restores weights and noise amplitudes only).  Only tensors / python scalars are written; load with
weights_only=True.

hipGraph note (train.train): the switch to graph replay after two eager iterations runs ONE more real iteration on the
batch at hand as the capture warm-up (it counts towards niter), and from then on the gradient penalty's alpha comes from
the device generator instead of the CPU generator - same distribution, a different stream than the reference's."""
import os

import torch


def stage_checkpoint(opt, trainer):
    """The three dictionaries the reference saves at the end of a stage."""
    netG, netD = trainer.netG, trainer.netD
    out = {
        'Noise_Amps.pth': {'data': list(opt.Noise_Amps)},
        'netG.pth': {'scale': opt.scale_idx, 'state_dict': {k: v.detach().cpu() for k, v in netG.state_dict().items()},
                     'optimizer': trainer.optimizerG.state_dict(), 'noise_amps': list(opt.Noise_Amps)},
    }
    if netD is not None:
        out['netD_{}.pth'.format(opt.scale_idx)] = {
            'scale': opt.scale_idx, 'state_dict': {k: v.detach().cpu() for k, v in netD.state_dict().items()},
            'optimizer': trainer.optimizerD.state_dict()}
    return out


def save_stage(directory, opt, trainer):
    os.makedirs(directory, exist_ok=True)
    for name, obj in stage_checkpoint(opt, trainer).items():
        torch.save(obj, os.path.join(directory, name))


def resume_generator(netG, directory, map_location='cpu'):
    """--netG resume (train_video.py:399-410): grow the body to the saved scale, load weights and noise amplitudes.
    Returns (scale, noise_amps).  As in the reference the optimizer state is not restored."""
    ckpt = torch.load(os.path.join(directory, 'netG.pth'), map_location=map_location, weights_only=True)
    for _ in range(ckpt['scale']):
        netG.init_next_stage()
    netG.load_state_dict(ckpt['state_dict'])
    amps = torch.load(os.path.join(directory, 'Noise_Amps.pth'), map_location=map_location, weights_only=True)['data']
    return ckpt['scale'], list(amps)


def warm_start_discriminator(netD, directory, scale_idx, map_location='cpu'):
    """D of stage s starts from netD_{s-1}.pth (train_video.py:47-52)."""
    path = os.path.join(directory, 'netD_{}.pth'.format(scale_idx - 1))
    netD.load_state_dict(torch.load(path, map_location=map_location, weights_only=True)['state_dict'])
    return netD
