"""Data in / out of the hot path: latent batches in, ``.pkl`` sample files out.

The reference feeds tf.data TFRecords (input_pipeline.py:113-235); TensorFlow is not on the target
image and BASELINE's workloads are synthetic, so this module provides
  * ``SyntheticLatents``   clip(0.25*N(0,1), -1, 1) batches with the dataset attributes the
                           reference loops rely on (.examples, .min, .max)   [BASELINE.md section 2]
  * ``ArrayLatents``       the same interface over a NumPy array / .npy / .pkl of latents
  * ``normalize_dataset`` / ``slice_transform`` / ``inverse_data_transform`` (input_pipeline.py:
    36-48,78-110) and ``save`` / ``load`` (utils/data_utils.py:30-41, pickle protocol 4)
  * ``open_dataset``       the reference's TFRecord shards (tfrecord.py: no TensorFlow) or .npy / .pkl arrays,
                           with the reference's slice -> batch -> per-split min/max (+ cache pickles) -> normalise order
"""
from __future__ import annotations

import glob
import os
import pickle
from typing import Iterator, Optional, Sequence

import numpy as np
import torch


def save(obj, path: str) -> None:
    """utils/data_utils.py:30-35."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(obj, f, protocol=4)


def load(path: str):
    """utils/data_utils.py:38-41."""
    with open(path, "rb") as f:
        return pickle.load(f)


def normalize_dataset(batch, data_min, data_max):
    """input_pipeline.py:36-40."""
    batch = (batch - data_min) / (data_max - data_min)
    return 2.0 * batch - 1.0


def slice_transform(batch, slice_idx=None, dim_weights=None):
    """input_pipeline.py:43-48 (NumPy gather instead of tf.gather)."""
    if dim_weights is not None:
        batch = batch * dim_weights
    if slice_idx is not None:
        batch = np.take(batch, slice_idx, axis=-1)
    return batch


def data_transform(batch, pca=None):
    """input_pipeline.py:51-75 ('vae' problem): the optional PCA projection, rows flattened for >2-D examples."""
    if pca is not None:
        batch = np.asarray(batch)
        if batch.ndim > 2:
            shape = batch.shape
            batch = pca.transform(batch.reshape(shape[0], -1)).reshape(*shape)
        else:
            batch = pca.transform(batch)
    return batch


def inverse_data_transform(batch, normalize=True, pca=None, data_min=0.0, data_max=1.0, slice_idx=None,
                           dim_weights=None, out_channels=512):
    """input_pipeline.py:78-110.  The non-selected latent dims are filled with an unseeded
    np.random.randn exactly like the reference (:102-105)."""
    batch = np.asarray(batch)
    if normalize:
        batch = (batch + 1.0) / 2.0
        batch = (data_max - data_min) * batch + data_min
    if pca is not None:
        batch = pca.inverse_transform(batch)
    if slice_idx is not None:
        transformed = np.random.randn(*batch.shape[:-1], out_channels)
        transformed[..., slice_idx] = batch
        batch = transformed
    if dim_weights is not None:
        batch = batch / dim_weights
    return batch


class ArrayLatents:
    """Batched view of an in-memory latent array with the attributes train_ncsn.py reads from its
    tf.data datasets: ``examples`` = batches per epoch (:359,381), ``min`` / ``max`` (:424-431)."""

    def __init__(self, array: np.ndarray, batch_size: int, data_min: float = -1.0, data_max: float = 1.0,
                 drop_remainder: bool = True, device: Optional[str] = None, rank: int = 0, world_size: int = 1,
                 shuffle: bool = False, seed: int = 0):
        """``shuffle``: a fresh seeded permutation of the WHOLE split every epoch (the reference reshuffles its files and an
        8*batch buffer per epoch, utils/data_utils.py:159-183); with world_size > 1 every rank takes its own slice of the same
        global permutation, so the ranks stay disjoint: the split then stays in (pinned) HOST memory and only this rank's
        1 / world_size of the epoch is gathered and copied to the device, once per epoch (no per-step copies, 1 / world_size of the
        HBM footprint).  Unshuffled: one contiguous shard."""
        array = np.ascontiguousarray(array, dtype=np.float32)
        self.shuffle, self.seed, self.epoch = bool(shuffle), int(seed), 0
        self.rank, self.world_size = int(rank), int(world_size)
        per = len(array) // world_size
        if world_size > 1 and not shuffle:      # disjoint contiguous shard per rank
            array = array[rank * per:(rank + 1) * per]
        self.array = torch.from_numpy(array)
        self.device = device
        self.host_resident = bool(shuffle) and world_size > 1 and device is not None
        self._pinned_shard = None
        if self.host_resident:
            # the whole split stays in PAGEABLE host memory on every rank; only this rank's per-epoch shard (per rows) is
            # staged through one pinned buffer, so pinned memory does not grow with the world size and the H2D copy of the
            # gathered shard is a real asynchronous DMA (a gather result is pageable: its copy would be staged and blocking)
            try:
                self._pinned_shard = torch.empty((per, *self.array.shape[1:]), dtype=torch.float32).pin_memory()
            except Exception:            # no accelerator runtime (CPU-only tests): pageable memory works the same
                self._pinned_shard = None
        elif device is not None:
            self.array = self.array.to(device)   # resident in HBM: no per-step H2D copy
        self.batch_size = batch_size
        self.per_rank = per
        self.examples = per // batch_size if drop_remainder else -(-per // batch_size)
        self.min, self.max = data_min, data_max

    @property
    def sample_shape(self):
        return tuple(self.array.shape[1:])

    def epoch_order(self, epoch: int) -> torch.Tensor:
        """This rank's example indices for ``epoch``: slice [rank*per, (rank+1)*per) of randperm(N; seed, epoch)."""
        g = torch.Generator().manual_seed((self.seed * 1000003 + epoch) & 0x7FFFFFFFFFFFFFFF)
        perm = torch.randperm(len(self.array), generator=g)
        return perm[self.rank * self.per_rank:(self.rank + 1) * self.per_rank]

    def set_epoch(self, epoch: int) -> None:
        """The training loop's epoch (a resumed run continues with the permutation of ITS epoch, not epoch 0's)."""
        self.epoch = int(epoch)

    def __iter__(self) -> Iterator[torch.Tensor]:
        if not self.shuffle:
            for i in range(self.examples):
                yield self.array[i * self.batch_size:(i + 1) * self.batch_size]
            return
        idx = self.epoch_order(self.epoch)
        self.epoch += 1
        if self.host_resident:           # this rank's share of the epoch: one gather on the host, one H2D copy
            if self._pinned_shard is not None:
                torch.index_select(self.array, 0, idx, out=self._pinned_shard)
                shard = self._pinned_shard.to(self.device, non_blocking=True)
                torch.cuda.current_stream(self.device).synchronize()     # the buffer is rewritten at the next epoch boundary
            else:
                shard = self.array.index_select(0, idx).to(self.device)
            for i in range(self.examples):
                yield shard[i * self.batch_size:(i + 1) * self.batch_size]
            return
        idx = idx.to(self.array.device)
        for i in range(self.examples):
            yield self.array.index_select(0, idx[i * self.batch_size:(i + 1) * self.batch_size])

    def __len__(self):
        return self.examples

    def take_examples(self, n: int) -> np.ndarray:
        return self.array[:n].cpu().numpy()


class SyntheticLatents(ArrayLatents):
    """x0 = clip(0.25*N(0,1), -1, 1), torch CPU generator seed 1234 (+rank) -- BASELINE.md section 2."""

    def __init__(self, sample_shape: Sequence[int], num_examples: int, batch_size: int, seed: int = 1234,
                 device: Optional[str] = None, rank: int = 0, world_size: int = 1):
        g = torch.Generator().manual_seed(seed + rank)
        per = num_examples // world_size
        x = torch.clamp(0.25 * torch.randn(per, *sample_shape, generator=g), -1.0, 1.0)
        super().__init__(x.numpy(), batch_size, -1.0, 1.0, True, device)


def compute_dataset_min_max(batched: np.ndarray, ds_split: str = "train", cache: bool = False,
                            cache_dir: Optional[str] = None, config: str = ""):
    """utils/data_utils.py:128-156: min / max over the batched (drop_remainder) split, read from / written to
    ``{cache_dir}/cache/{split}_{config}_{min,max}.pkl`` (pickled float32 scalars) like the reference."""
    if cache_dir is not None:
        min_p = os.path.join(cache_dir, f"cache/{ds_split}_{config}_min.pkl")
        max_p = os.path.join(cache_dir, f"cache/{ds_split}_{config}_max.pkl")
        if os.path.exists(min_p) and os.path.exists(max_p):
            return load(min_p), load(max_p)
    ds_min, ds_max = np.float32(batched.min()), np.float32(batched.max())
    if cache and cache_dir is not None:
        try:
            save(ds_min, min_p)
            save(ds_max, max_p)
        except OSError:                      # read-only dataset directory: the cache is an optimisation only
            pass
    return ds_min, ds_max


def _config_name(*ckpts: str) -> str:
    """input_pipeline.py:186-188: concatenated basenames (sans extension) of the pca / slice / dim-weights files."""
    return "".join((c or "").split("/")[-1].split(".")[0] for c in ckpts)


def open_dataset(path: str, batch_size: int, sample_shape: Sequence[int], device=None, rank=0, world_size=1,
                 normalize=True, slice_idx=None, dim_weights=None, data_shape: Optional[Sequence[int]] = None,
                 pca_ckpt: str = "", slice_ckpt: str = "", dim_weights_ckpt: str = "", cache: bool = True,
                 shuffle: bool = True, seed: int = 0):
    """``--dataset``: the reference's ``{train,eval}-*.tfrecord`` shards (input_pipeline.py:126-139; read without
    TensorFlow, tfrecord.py) or {train,eval}.npy / .pkl arrays of raw latents (N, *data_shape).

    Order of operations as in get_dataset (:113-235): PCA (``--pca_ckpt``, a pickled object with .transform /
    .inverse_transform) -> slice / weight transform -> batch with drop_remainder -> per-split min / max (cached under
    ``{dataset}/cache``) -> each split normalised with ITS OWN min / max (:189-208).  Records are read in sorted file
    order; the training split is then reshuffled every epoch with a permutation seeded by (seed, epoch) (the reference
    shuffles files and an 8*batch buffer with an unseeded tf.data shuffle); the eval split stays ordered."""
    from . import tfrecord
    path = os.path.expanduser(path)
    raw_shape = tuple(int(v) for v in (data_shape if data_shape is not None else sample_shape))
    pca = load(os.path.expanduser(pca_ckpt)) if pca_ckpt else None              # input_pipeline.py:144
    out = []
    for split in ("train", "eval"):
        arr = None
        if glob.glob(os.path.join(path, f"{split}-*.tfrecord")):
            arr = tfrecord.read_latents(os.path.join(path, f"{split}-*.tfrecord"), raw_shape)
        else:
            for ext in (".npy", ".pkl"):
                f = os.path.join(path, split + ext)
                if os.path.exists(f):
                    arr = np.load(f) if ext == ".npy" else load(f)
                    break
        if arr is None:
            raise FileNotFoundError(f"{path}: neither {split}-*.tfrecord nor {split}.npy|.pkl found (or pass --synthetic)")
        arr = np.asarray(data_transform(np.asarray(arr, np.float32), pca), np.float32)       # :159-166
        arr = slice_transform(arr, slice_idx, dim_weights)
        out.append(arr[:(len(arr) // batch_size) * batch_size])                 # batch(drop_remainder=True)
    config = _config_name(pca_ckpt, slice_ckpt, dim_weights_ckpt)
    sets = []
    for split, arr in zip(("train", "eval"), out):
        dmin, dmax = 0.0, 1.0
        if normalize:
            dmin, dmax = compute_dataset_min_max(arr, split, cache, path if cache else None, config)
            arr = normalize_dataset(arr, dmin, dmax)
        is_train = split == "train"
        sets.append(ArrayLatents(arr.reshape(len(arr), *sample_shape), batch_size, float(dmin), float(dmax), True, device,
                                 rank if is_train else 0, world_size if is_train else 1,
                                 shuffle=shuffle and is_train, seed=seed))
    return sets[0], sets[1]
