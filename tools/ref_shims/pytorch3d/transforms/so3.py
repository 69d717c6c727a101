import torch


def hat(v):
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    o = torch.zeros_like(x)
    return torch.stack([torch.stack([o, -z, y], -1),
                        torch.stack([z, o, -x], -1),
                        torch.stack([-y, x, o], -1)], -2)


def so3_exp_map(log_rot, eps=1e-4):
    nrms = (log_rot * log_rot).sum(1)
    theta = torch.clamp(nrms, eps).sqrt()
    inv = 1.0 / theta
    fac1 = inv * theta.sin()
    fac2 = inv * inv * (1.0 - theta.cos())
    k = hat(log_rot)
    k2 = torch.bmm(k, k)
    eye = torch.eye(3, dtype=log_rot.dtype, device=log_rot.device)[None]
    return fac1[:, None, None] * k + fac2[:, None, None] * k2 + eye
