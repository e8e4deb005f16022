"""Free energies on the engine (reference ``imdbn/utils/energy_utils.py:19-56``).

``rbm_free_energy`` is the function the reference's evaluation tooling calls; here it is one engine call
(``imdbn_rbm_free_energy``).  ``class_free_energies`` evaluates F([z, e_k]) for every label k by stacking the K
one-hot completions of each row into one [B*K, V] batch -- the engine streams W once for the whole stack -- instead of
the reference's dense [B, K, H] broadcast.  The tracing / plotting helpers of that module are observability and are
out of scope.
"""
from __future__ import annotations

import torch


@torch.no_grad()
def rbm_free_energy(rbm, v: torch.Tensor) -> torch.Tensor:
    """F(v) = -v.b - sum_j softplus(c_j + (vW)_j); v: [B, V] in [0, 1] (may be mean-field); returns [B]."""
    return rbm.free_energy(v)


@torch.no_grad()
def class_free_energies(joint_rbm, z_img_top: torch.Tensor, K: int, Dz: int) -> torch.Tensor:
    """F_k(z) = F([z, e_k]) for k = 0..K-1; z_img_top: [B, Dz] -> [B, K]."""
    z = z_img_top.to(joint_rbm.W.device).float()
    B = z.size(0)
    v = torch.zeros(B, K, Dz + K, device=z.device)
    v[:, :, :Dz] = z.unsqueeze(1)
    v[:, :, Dz:] = torch.eye(K, device=z.device).unsqueeze(0)
    return joint_rbm.free_energy(v.view(B * K, Dz + K)).view(B, K)
