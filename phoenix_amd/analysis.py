"""Row f3 of SURVEY.md section 8: the batched analysis callers of the hot path.

`gene_influence_scores` is the loop of find_gene_influences.py:64-77 (per gene: integrate a random batch of initial
states, overwrite that gene's column with a second random draw, integrate again, score = mean |difference| over all
OTHER genes and all projected time points after the first).  Every solve keeps the reference's batch semantics (one
shared adaptive step size per `odeint` call of the `[n_inputs, 1, N]` batch, 10 interpolated outputs on a float64
grid), but the two solves of `genes_per_launch` genes go to the engine together (`odeint_calls`: one batch group and
one step controller per call, as many calls per launch as the device holds) and the scores stay on the device until
the end (the reference calls `.item()` inside the loop).
`DataHandler.calculate_trajectory` (phoenix_amd/data.py) is the other caller of this row.
"""
import numpy as np
import torch

from .odeint import odeint_calls


def influence_score(unpert_out, pert_out, this_gene):
    """find_gene_influences.py:74-75: mean over outputs 1.., samples and every gene but `this_gene`."""
    diff = (unpert_out[1:] - pert_out[1:]).abs()
    n_other = diff.shape[-1] - 1
    total = diff.sum() - diff[..., this_gene].sum()
    return total / (diff[..., 0].numel() * n_other)


def gene_influence_scores(odenet, dim, method, n_random_inputs_per_gene=60, time_pts_to_project=None, device="cuda",
                          genes=None, draws=None, genes_per_launch=8):
    """Returns a float32 numpy array with one score per gene in `genes` (default: all `dim` genes, in order).
    `draws(gene) -> (this_init [n,1,dim], this_pert_col [n])` replaces the reference's two `torch.rand` calls
    (find_gene_influences.py:69,71); by default they are drawn on `device` in the same order."""
    if time_pts_to_project is None:
        time_pts_to_project = torch.from_numpy(np.arange(0, 1, 0.1))              # :65 (float64 grid)
    t = time_pts_to_project.to(device)
    genes = list(range(dim)) if genes is None else list(genes)
    scores = torch.zeros(len(genes), dtype=torch.float32, device=device)
    with torch.no_grad():
        for k0 in range(0, len(genes), genes_per_launch):
            batch = genes[k0:k0 + genes_per_launch]
            inits = []
            for this_gene in batch:
                if draws is None:
                    this_init = 1 * (torch.rand(n_random_inputs_per_gene, 1, dim, device=device) - 0.5)
                    this_pert_col = 1 * (torch.rand(n_random_inputs_per_gene, device=device) - 0.5)
                else:
                    this_init, this_pert_col = (x.to(device) for x in draws(this_gene))
                pert_init = this_init.clone()
                pert_init[:, 0, this_gene] = this_pert_col
                inits += [this_init, pert_init]
            out = odeint_calls(odenet, torch.stack(inits), t, method=method)   # [2k, T, n, 1, dim]
            for j, this_gene in enumerate(batch):
                scores[k0 + j] = influence_score(out[2 * j], out[2 * j + 1], this_gene)
    return scores.cpu().numpy()
