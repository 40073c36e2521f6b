"""WGAN-GP gradient penalty - mirror of the reference's modules/utils.py."""
import torch

from .. import ops


def calc_gradient_penalty(netD, real_data, fake_data, LAMBDA, device, alpha=None):
    """lambda * mean_{b,voxel}((||dD/dx_hat[b,:,voxel]||_2 - 1)^2) with x_hat = alpha*real + (1-alpha)*fake
    (reference: modules/utils.py:4-19).  `alpha` is one scalar for the whole batch drawn from the CPU generator,
    as in the reference; tests inject it.  x_hat is a fresh leaf, so nothing flows back into `fake_data`.
    The double backward runs through ops.Conv / ConvBwdData / ConvBwdWeight / LReLUMaskMul."""
    if alpha is None:
        alpha = torch.rand(1, 1)
    if not alpha.is_cuda:
        alpha = ops.host_floats_to_device([float(alpha.reshape(-1)[0])], real_data.device)  # no host stall (see ops)
    alpha = alpha.reshape(1).to(device=real_data.device, dtype=torch.float32)
    interpolates = ops.lerp(real_data, fake_data, alpha)
    interpolates.requires_grad_(True)
    disc_interpolates = netD(interpolates)
    ones = torch.ones_like(disc_interpolates)
    with ops.inputs_only():  # d/dx_hat only: no weight / bias gradients in this pass (they come from the double backward)
        gradients = torch.autograd.grad(outputs=disc_interpolates, inputs=interpolates, grad_outputs=ones,
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
    return ops.GradPenalty.apply(gradients, float(LAMBDA))
