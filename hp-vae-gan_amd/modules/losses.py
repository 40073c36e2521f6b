"""Loss terms of the train step - mirror of the reference's modules/losses.py."""
from .. import ops

__all__ = ['kl_criterion', 'kl_bern_criterion', 'mse_loss', 'wgan_mean']


def kl_criterion(mu, logvar):
    """KL(N(mu, exp(logvar)) || N(0, 1)) averaged over ALL elements (reference: modules/losses.py:7-9)."""
    return ops.KL.apply(mu, logvar)


def mse_loss(a, b):
    """nn.MSELoss() as configured at train_video.py:355 (`opt.rec_loss`)."""
    return ops.MSE.apply(a, b)


def wgan_mean(x, sign=1.0):
    """sign * x.mean(): the critic terms errD_real / errD_fake / errG (train_video.py:170,178,194)."""
    return ops.MeanScaled.apply(x, float(sign))


def kl_bern_criterion(x):
    """KL(Bernoulli(x) || Bernoulli(0.5)) averaged over all elements (reference: modules/losses.py:12-14)."""
    return ops.KLBern.apply(x)
