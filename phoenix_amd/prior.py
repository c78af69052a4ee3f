"""Prior-matrix handling of the reference drivers.

`read_prior_matrix` mirrors train_insilico.py:64-73 (dense CSV, or 1-based `(i, j, v)` triplets when
`sparse=True`) but keeps the matrix sparse; `prior_targets` builds `prior_grad = X @ P`
(train_insilico.py:207-211) with a CSC SpMM on the GPU instead of a dense N x N product."""
import ctypes as C

import numpy as np
import torch

from . import _lib, engine


class PriorMatrix:
    """N x N prior in CSC form on the device (rows ascending inside each column)."""

    def __init__(self, rows, cols, vals, num_genes, device):
        rows = np.asarray(rows, np.int64)
        cols = np.asarray(cols, np.int64)
        vals = np.asarray(vals, np.float32)
        keep = vals != 0
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        # torch.sparse_coo_tensor(...).to_dense() sums duplicate coordinates (train_insilico.py:71-72)
        order = np.lexsort((rows, cols))
        rows, cols, vals = rows[order], cols[order], vals[order]
        if len(rows):
            key = cols * num_genes + rows
            uniq, first = np.unique(key, return_index=True)
            vals = np.add.reduceat(vals, first).astype(np.float32)
            rows, cols = rows[first], cols[first]
        self.N = int(num_genes)
        self.nnz = int(len(vals))
        colptr = np.zeros(self.N + 1, np.int32)
        np.add.at(colptr, cols + 1, 1)
        self.colptr = torch.from_numpy(np.cumsum(colptr).astype(np.int32)).to(device)
        self.rowidx = torch.from_numpy(rows.astype(np.int32)).to(device)
        self.vals = torch.from_numpy(vals).to(device)
        self.device = torch.device(device)
        self._build_sell(rows, cols, vals, np.cumsum(colptr))

    def _build_sell(self, rows, cols, vals, colptr):
        """sliced-ELL copy for phx_prior_targets_sell: slices of 64 columns, entry i of column 64 s + l at
        sptr[s] + 64 i + l (rows ascending inside a column, padding row 0 / value 0)"""
        N = self.N
        ns = (N + 63) // 64
        lens = np.zeros(ns * 64, np.int64)
        lens[:N] = np.diff(colptr)
        width = lens.reshape(ns, 64).max(axis=1)
        sptr = np.concatenate([[0], np.cumsum(width * 64)])
        ridx = np.zeros(int(sptr[-1]), np.int32)
        v = np.zeros(int(sptr[-1]), np.float32)
        if len(vals):
            i_in_col = np.arange(len(vals)) - colptr[cols]          # position inside its column (rows are ascending)
            dst = sptr[cols // 64] + i_in_col * 64 + (cols % 64)
            ridx[dst] = rows
            v[dst] = vals
        self.sell_ptr = torch.from_numpy(sptr[:-1].astype(np.int64)).to(self.device)
        self.sell_width = torch.from_numpy(width.astype(np.int32)).to(self.device)
        self.sell_rows = torch.from_numpy(ridx).to(self.device)
        self.sell_vals = torch.from_numpy(v).to(self.device)

    def abs(self):
        """`torch.abs(prior_mat)` of the real-data drivers (train_breast.py:264-265, train_yeast.py:263)"""
        out = object.__new__(PriorMatrix)
        out.__dict__.update(self.__dict__)
        out.vals = self.vals.abs()
        out.sell_vals = self.sell_vals.abs()
        return out

    def to_dense(self):
        d = torch.zeros(self.N, self.N, dtype=torch.float32)
        cp = self.colptr.cpu().numpy()
        cols = np.repeat(np.arange(self.N), np.diff(cp))
        d[self.rowidx.cpu().long(), torch.from_numpy(cols).long()] = self.vals.cpu()
        return d


def read_prior_matrix(prior_mat_file_loc, sparse=False, num_genes=11165, device="cuda"):
    """train_insilico.py:64-73.  sparse=False: dense CSV (N rows of N values); sparse=True: rows of 1-based
    `i, j, value` triplets.  Returns a PriorMatrix (never a dense N x N tensor)."""
    mat = np.genfromtxt(prior_mat_file_loc, delimiter=",")
    if not sparse:
        mat = np.atleast_2d(mat).astype(np.float32)
        r, c = np.nonzero(mat)
        return PriorMatrix(r, c, mat[r, c], mat.shape[0], device)
    mat = np.atleast_2d(mat)
    return PriorMatrix(mat[:, 0].astype(int) - 1, mat[:, 1].astype(int) - 1, mat[:, 2], num_genes, device)


def prior_targets(batch_for_prior, prior):
    """prior_grad = torch.matmul(batch_for_prior, prior_mat) (train_insilico.py:209-210) for X [K,1,N] or [K,N]."""
    X = batch_for_prior
    engine._require_gpu(X, "batch_for_prior")
    if X.dtype != torch.float32:
        raise TypeError("batch_for_prior must be float32")
    N = prior.N
    if X.shape[-1] != N:
        raise ValueError("last dimension %d != number of genes %d" % (X.shape[-1], N))
    X2 = X.detach().reshape(-1, N).contiguous()
    out = torch.empty_like(X2)
    lib = _lib.load()
    rc = lib.phx_prior_targets_sell(engine._p(prior.sell_ptr), engine._p(prior.sell_width), engine._p(prior.sell_rows),
                                    engine._p(prior.sell_vals), engine._p(X2), engine._p(out), X2.shape[0], N,
                                    engine._stream_ptr())
    if rc == 4:      # a row of X does not fit LDS (N > ~40 000): the plain CSC kernel
        rc = lib.phx_prior_targets(engine._p(prior.colptr), engine._p(prior.rowidx), engine._p(prior.vals),
                                   engine._p(X2), engine._p(out), X2.shape[0], N, engine._stream_ptr())
    engine._check_call(rc)
    return out.reshape(X.shape)
