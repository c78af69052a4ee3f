"""Row f2 of SURVEY.md section 8: the data path either side of the hot loop.

`readcsv` / `writecsv` keep the reference's wire format (csvreader.py:13-73) and `DataHandler` keeps the reference's
class surface (datahandler.py:19-419: constructor, `fromcsv`, `get_batch`, validation sets, `get_true_mu_set_*`,
`get_times`, `calculate_trajectory`) and -- this is the parity contract the tests check against golden G9 -- the exact
sequence of `np.random` draws, so a run seeded like the reference's sees the same noise, the same train/validation
split and the same batches in the same order.

What is different is where the data lives.  The reference keeps a Python list of per-trajectory tensors and assembles
every batch with per-sample `append` + `torch.stack`; here all time points of all trajectories sit in ONE packed
device tensor `[total_points, 1, N]` (288 GB of HBM holds every shipped dataset thousands of times over), batches are
one `index_select` on the device from host-computed row numbers, and `calculate_trajectory` integrates all plotted
samples in one engine launch instead of one `odeint` per sample.  Plotting (`compare_train_val_plot`) is out of scope.
"""
import csv
from math import ceil

import numpy as np
import torch


def expression_maker(val):
    return val                                                   # csvreader.py:9-11 (identity in the reference)


def readcsv(fp, device, noise_to_add, scale_expression):
    """csvreader.py:13-56.  First row `dim, ntraj`; then per trajectory `dim` gene rows and one time row; empty cells
    are NaN.  One `np.random.normal(0, noise_to_add)` draw per gene cell in file order (drawn even for zero noise, as
    the reference does, so the global numpy stream stays in step)."""
    data_np, data_pt, data_np_0noise, data_pt_0noise, t_np, t_pt = [], [], [], [], [], []
    with open(fp, "r") as f:
        rows = [np.array([float(s) if s != "" else np.nan for s in r], dtype=np.float64)
                for r in csv.reader(f, delimiter=",")]
    dim, ntraj = int(rows[0][0]), int(rows[0][1])
    data = rows[1:]
    for traj in range(ntraj):
        base = traj * (dim + 1)
        length = len(data[base])
        traj_data = np.zeros((length, 1, dim), dtype=np.float32)
        traj_data_0noise = np.zeros((length, 1, dim), dtype=np.float32)
        for d in range(dim):
            row = data[base + d]
            noisy = row + np.random.normal(0, noise_to_add, size=row.shape[0])   # same stream as per-cell draws
            traj_data[:, 0, d] = scale_expression * expression_maker(noisy)
            traj_data_0noise[:, 0, d] = scale_expression * expression_maker(row)
        trow = data[base + dim]
        t_np.append(np.array(trow))
        t_pt.append(torch.tensor(trow.tolist()).to(device))                       # float32, like torch.tensor(list)
        data_np.append(traj_data)
        data_np_0noise.append(traj_data_0noise)
        data_pt.append(torch.from_numpy(traj_data).to(device))
        data_pt_0noise.append(torch.from_numpy(traj_data_0noise).to(device))
    return data_np, data_pt, t_np, t_pt, dim, ntraj, data_np_0noise, data_pt_0noise


def writecsv(fp, dim, ntraj, data_np, t_np):
    """csvreader.py:58-73."""
    with open(fp, "w") as f:
        writer = csv.writer(f, delimiter=",", lineterminator="\n")
        writer.writerow(np.array([dim, ntraj]))
        for i in range(ntraj):
            for j in range(dim):
                writer.writerow(data_np[i][:, :, j].flatten())
            writer.writerow(t_np[i])


class _Packed:
    """All time points of all trajectories in one device tensor; `views[i]` aliases trajectory i."""

    def __init__(self, tensors, device):
        self.offsets = np.concatenate([[0], np.cumsum([int(x.shape[0]) for x in tensors])]).astype(np.int64)
        self.flat = torch.cat([x.to(device) for x in tensors], dim=0).contiguous()
        self.views = [self.flat[self.offsets[i]:self.offsets[i + 1]] for i in range(len(tensors))]

    def take(self, rows):
        idx = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int64), device=self.flat.device)
        return self.flat.index_select(0, idx.reshape(-1)).reshape(tuple(idx.shape) + tuple(self.flat.shape[1:]))


class DataHandler:
    """datahandler.py:19-57 (same constructor arguments and attribute names)."""

    def __init__(self, data_np, data_pt, time_np, time_pt, dim, ntraj, val_split, device, normalize, batch_type,
                 batch_time, batch_time_frac, data_np_0noise, data_pt_0noise, img_save_dir):
        self.device = device
        self.data_np = data_np
        self.data_np_0noise = data_np_0noise
        self.time_np = time_np
        self.dim = dim
        self.ntraj = ntraj
        self.batch_type = batch_type
        self.batch_time = batch_time
        self.batch_npoints = int(ceil(batch_time_frac * batch_time))
        self._data = _Packed(data_pt, device)
        self._data0 = _Packed(data_pt_0noise, device)
        self._time = _Packed(time_pt, device)
        self.data_pt, self.data_pt_0noise, self.time_pt = self._data.views, self._data0.views, self._time.views
        if normalize:
            self._normalize()
        self.val_split = val_split
        self.epoch_done = False
        self.img_save_dir = img_save_dir
        self.num_trajs_to_plot = 7
        self._calc_datasize()
        if batch_type == "single":
            self._split_data_single(val_split)
            self._create_validation_set_single()
        elif batch_type == "trajectory":
            self._split_data_traj(val_split)
            self._create_validation_set_traj()
        elif batch_type == "batch_time":
            self._split_data_single(val_split)                   # _split_data_time is the same code (:206-214)
            self._create_validation_set_time()
        else:
            print("Invalid batch type: '{}'".format(batch_type))
            raise ValueError

    @classmethod
    def fromcsv(cls, fp, device, val_split, normalize=False, batch_type="single", batch_time=1, batch_time_frac=1.0,
                noise=0, img_save_dir="", scale_expression=1, **_ignored):
        """datahandler.py:59-63.  Extra keywords some reference callers pass (`log_scale`, `init_bias_y`,
        find_gene_influences.py:53-54) are accepted and unused, as they are not part of this data path."""
        r = readcsv(fp, device, noise_to_add=noise, scale_expression=scale_expression)
        data_np, data_pt, t_np, t_pt, dim, ntraj, data_np_0noise, data_pt_0noise = r
        return cls(data_np, data_pt, t_np, t_pt, dim, ntraj, val_split, device, normalize, batch_type, batch_time,
                   batch_time_frac, data_np_0noise, data_pt_0noise, img_save_dir)

    def saveascsv(self, fp):
        writecsv(fp, self.dim, self.ntraj, self.data_np, self.time_np)

    def _normalize(self):
        """datahandler.py:75-82: divide the noisy data by the largest per-trajectory max |value|.  The reference's
        `max(abs(data)) > max_val` test is False for a trajectory holding a NaN cell, so those do not take part."""
        per_traj = torch.stack([v.abs().max() for v in self._data.views]).cpu().numpy()
        ok = per_traj[~np.isnan(per_traj)]
        max_val = np.float32(ok.max()) if ok.size and ok.max() > 0 else np.float32(0)
        # a tensor divisor: torch turns division by a host scalar into a multiply by its reciprocal on the GPU
        self._data.flat.div_(torch.tensor(max_val, device=self._data.flat.device))
        for i in range(self.ntraj):
            self.data_np[i] = self.data_np[i] / max_val

    # ------------------------------------------------------------------ index bookkeeping (host)
    def _calc_datasize(self):
        """datahandler.py:216-245: `indx` = every (trajectory, time index) that has a successor."""
        shrink = self.batch_time if self.batch_type == "batch_time" else 0
        self.datasize = 0
        self.indx = []
        for row_indx, row in enumerate(self.time_np):
            rowsize = row.size - shrink
            self.datasize += rowsize
            self.indx.extend((row_indx, x) for x in range(max(rowsize - 1, 0)))

    def _split_data_single(self, val_split):
        """datahandler.py:182-193."""
        self.n_val = int((self.datasize - self.ntraj) * val_split)
        all_indx = np.arange(len(self.indx))
        val_indx = np.random.choice(all_indx, size=self.n_val, replace=False)
        train_indx = np.setdiff1d(all_indx, val_indx, assume_unique=True)
        self.val_set_indx = [self.indx[x] for x in val_indx]
        self.train_set_original = [self.indx[x] for x in train_indx]
        self.train_data_length = len(self.train_set_original)

    def _split_data_traj(self, val_split):
        """datahandler.py:196-203."""
        self.n_val = int(round(self.ntraj * val_split))
        self.val_set_indx = np.random.choice(np.arange(self.ntraj), size=self.n_val, replace=False)
        self.train_set_original = np.setdiff1d(np.arange(self.ntraj), self.val_set_indx)
        self.train_data_length = len(self.train_set_original)

    def reset_epoch(self):
        self.train_set = self.train_set_original.copy()
        self.epoch_done = False

    def _rows(self, pairs, shift=0):
        p = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        return self._data.offsets[p[:, 0]] + p[:, 1] + shift

    def _pairs(self, packed, pairs):
        """(point, successor, [t_i, t_i+1]) of every (trajectory, index) pair: three device gathers."""
        r = self._rows(pairs)
        return packed.take(r), packed.take(r + 1), self._time.take(np.stack([r, r + 1], 1))

    # ------------------------------------------------------------------ batches
    def get_batch(self, batch_size):
        if self.batch_type == "single":
            return self._get_batch_single(batch_size)
        elif self.batch_type == "trajectory":
            return self._get_batch_traj(batch_size)
        elif self.batch_type == "batch_time":
            return self._get_batch_time(batch_size)

    def _draw(self, batch_size):
        n = len(self.train_set)
        if n > batch_size:
            indx = np.random.choice(n, batch_size, replace=False)
        else:
            indx = np.arange(n)
            self.epoch_done = True
        return np.sort(indx)[::-1]

    def _get_batch_single(self, batch_size):
        """datahandler.py:95-120 -> batch [B,1,N], t [B,2], target [B,1,N] on the device."""
        indx = self._draw(batch_size)
        batch, target, t = self._pairs(self._data, [self.train_set[x] for x in indx])
        for i in indx:
            self.train_set.pop(i)
        return batch, t, target

    def _get_batch_traj(self, batch_size):
        """datahandler.py:122-149: one whole trajectory as consecutive pairs; pairs with a NaN time are dropped."""
        n = len(self.train_set)
        if n > 1:
            i = np.random.choice(n, replace=False)
            indx = self.train_set[i]
            self.train_set = np.delete(self.train_set, i)
        else:
            indx = self.train_set[0]
            self.epoch_done = True
        tt = self.time_np[indx]
        keep = [(indx, i) for i in range(tt.size - 1) if not (np.isnan(tt[i]) or np.isnan(tt[i + 1]))]
        if not keep:
            z = self._data.flat[:0]
            return z, self._time.flat[:0].reshape(0, 2), z
        batch, target, t = self._pairs(self._data, keep)
        return batch, t, target

    def _get_batch_time(self, batch_size):
        """datahandler.py:151-180."""
        indx = self._draw(batch_size)
        rows = []
        for x in indx:
            i = self.train_set[x]
            sub = np.random.choice(np.arange(start=i[1], stop=i[1] + self.batch_time), size=self.batch_npoints,
                                   replace=False)
            sub = np.sort(sub)
            rows.append(self._data.offsets[i[0]] + np.append(sub, sub[-1] + 1))
        for i in indx:
            self.train_set.pop(i)
        rows = np.stack(rows)
        batch = self._data.take(rows[:, :-1]).squeeze()
        target = self._data.take(rows[:, 1:] + 1).squeeze()      # data[ii + 1] for ii in sub_indx[1:]  (:171)
        t = self._time.take(rows)
        return batch, t, target

    # ------------------------------------------------------------------ validation sets
    def _create_validation_set_single(self):
        """datahandler.py:342-354."""
        self.val_data, self.val_target, self.val_t = [], [], []
        if self.val_set_indx:
            self.val_data, self.val_target, self.val_t = self._pairs(self._data, self.val_set_indx)

    def _create_validation_set_traj(self):
        """datahandler.py:356-368 (needs equal-length validation trajectories, like the reference's stack)."""
        self.val_data, self.val_target, self.val_t = [], [], []
        if self.val_set_indx.any():
            self.val_data = torch.stack([self.data_pt[i][0:-1] for i in self.val_set_indx], dim=0)
            self.val_target = torch.stack([self.data_pt[i][1::] for i in self.val_set_indx], dim=0)
            self.val_t = torch.stack([self.time_pt[i] for i in self.val_set_indx], dim=0)

    def _create_validation_set_time(self):
        """datahandler.py:372-384."""
        self.val_data, self.val_target, self.val_t = [], [], []
        if self.val_set_indx:
            r = self._rows(self.val_set_indx)[:, None] + np.arange(self.batch_time + 1)[None, :]
            self.val_data = self._data.take(r[:, :-1]).squeeze()
            self.val_target = self._data.take(r[:, 1:]).squeeze()
            self.val_t = self._time.take(r)

    def get_validation_set(self):
        return self.val_data, self.val_t, self.val_target, self.n_val

    # ------------------------------------------------------------------ noise-free views
    def get_mu0(self):
        return [x[0] for x in self.data_pt_0noise]

    def get_mu1(self):
        return [x[0] for x in self.data_pt_0noise]

    def get_true_mu_set_pairwise(self, val_only=False, batch_type="trajectory"):
        """datahandler.py:258-288: all noise-free consecutive pairs (optionally of the validation set only), minus
        pairs with a NaN time."""
        if batch_type == "trajectory":
            vs = set(int(v) for v in np.asarray(self.val_set_indx).reshape(-1)) if val_only else None
            all_indx = [p for p in self.indx if vs is None or p[0] in vs]
        elif batch_type == "single":
            all_indx = self.val_set_indx if val_only else list(self.indx)
        keep = [p for p in all_indx
                if not (np.isnan(self.time_np[p[0]][p[1]]) or np.isnan(self.time_np[p[0]][p[1] + 1]))]
        if not keep:
            z = self._data0.flat[:0]
            return z, self._time.flat[:0].reshape(0, 2), z
        mean_data, mean_target, mean_t = self._pairs(self._data0, keep)
        return mean_data, mean_t, mean_target

    def get_true_mu_set_init_val_based(self, val_only=False):
        """datahandler.py:290-304."""
        t = torch.stack([self.time_pt[i] for i in self.val_set_indx])
        batch = torch.stack([self.data_pt[i][0] for i in self.val_set_indx])
        target = torch.stack([self.data_pt[i][1::] for i in self.val_set_indx])
        return batch, t, target

    def get_times(self):
        return torch.stack(self.time_pt)

    # ------------------------------------------------------------------ row f3: dense trajectories for plotting
    def calculate_trajectory(self, odenet, method, num_val_trajs, fixed_traj_idx=None):
        """datahandler.py:310-340: integrate the noise-free initial state of the chosen samples over
        `np.arange(0, 15, 0.05)` (300 outputs, float64 grid).  The reference runs one `odeint` per sample; here all
        samples go through ONE engine launch with per-sample step control (identical numerics per sample).  Returns
        the reference's triple: list of CPU tensors `[300, 1, N]`, the sample indices, the time grid."""
        from .odeint import odeint_per_sample
        extrap_time_points = np.arange(0, 15, 0.05)
        mu1 = self.get_mu1()
        if self.val_split == 1:
            if fixed_traj_idx is None:
                samples = sorted(np.random.choice(self.val_set_indx, num_val_trajs, replace=False))
            else:
                samples = fixed_traj_idx
        elif num_val_trajs > 0:
            samples = sorted(np.random.choice(self.val_set_indx, num_val_trajs, replace=False)) + \
                sorted(np.random.choice(self.train_set_original, self.num_trajs_to_plot - num_val_trajs, replace=False))
        elif self.batch_type == "single":
            owners = list(set(x[0] for x in self.train_set_original))
            try:
                samples = sorted(np.random.choice(owners, self.num_trajs_to_plot, replace=False))
            except ValueError:
                samples = sorted(owners)
        else:
            samples = sorted(np.random.choice(self.train_set_original, self.num_trajs_to_plot, replace=False))
        y0 = torch.stack([mu1[j] for j in samples])                                   # [S, 1, N]
        tt = torch.from_numpy(extrap_time_points).to(y0.device).expand(len(samples), -1).contiguous()
        with torch.no_grad():
            y = odeint_per_sample(odenet, y0, tt, method=method, adjoint=False)      # [300, S, 1, N]
        y = y.cpu()
        return [y[:, s] for s in range(len(samples))], samples, extrap_time_points
