"""KL terms of the ELBO as small torch functions (reference modules/losses.py:8-48), for callers that
want them on host tensors; the training path computes them inside libsgvae.so."""
import torch


def kl(mu, log_var):
    log_var = torch.clamp(log_var, min=-30, max=30)
    return torch.mean(0.5 * torch.sum(mu ** 2 + torch.exp(log_var) - log_var - 1, dim=[1]), dim=0)


def kl_2(delta_mu, delta_log_var, mu, log_var):
    log_var = torch.clamp(log_var, min=-30, max=30)
    delta_log_var = torch.clamp(delta_log_var, min=-30, max=30)
    var = torch.exp(log_var) + 1e-8
    loss = 0.5 * torch.sum(torch.exp(delta_log_var) / var + (mu - delta_mu) ** 2 / var - delta_log_var + log_var - 1,
                           dim=[1, 2])
    return torch.mean(loss, dim=0)
