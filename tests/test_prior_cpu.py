"""Row f1: prior-matrix parsing against the golden captured from the reference's
read_prior_matrix (CPU part; the SpMM itself is checked on the GPU in test_gpu_parity.py)."""
import os
import tempfile

import numpy as np

from conftest import load_golden, sub


def _write_triplets(trip):
    fh = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False)
    for a, b, v in trip:
        fh.write("%d,%d,%g\n" % (a, b, v))
    fh.close()
    return fh.name


def test_triplet_format_matches_reference_dense():
    from phoenix_amd.prior import read_prior_matrix
    g = sub(load_golden("g8_prior"), "trip/")
    path = _write_triplets(g["triplets"])
    try:
        P = read_prior_matrix(path, sparse=True, num_genes=40, device="cpu")
    finally:
        os.unlink(path)
    assert np.array_equal(P.to_dense().numpy(), g["dense"])          # duplicates summed, 1-based indices
    assert np.array_equal(P.abs().to_dense().numpy(), np.abs(g["dense"]))


def test_dense_format_matches_reference():
    from phoenix_amd.prior import read_prior_matrix
    g = sub(load_golden("g8_prior"), "g350/")
    n = 350
    dense = np.zeros((n, n), np.float32)
    dense[g["rows"], g["cols"]] = g["vals"]
    fh = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False)
    np.savetxt(fh, dense, delimiter=",", fmt="%g")
    fh.close()
    try:
        P = read_prior_matrix(fh.name, sparse=False, device="cpu")
    finally:
        os.unlink(fh.name)
    assert P.N == n and P.nnz == len(g["vals"]) == 550               # 550 nnz of 122 500
    assert np.array_equal(P.to_dense().numpy(), dense)
