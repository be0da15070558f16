"""Reader for the reference's capture data sets (data_provider_pytorch.py:12-79).

File format (one `.npz` per split, names `train*.npz` / `valid*.npz` / `test*.npz`):
    z_p, z_c   [N, H, W, 1]  projector- / camera-frame depth; either float, or the uint16 codes
                             of the original recording, depth = code / 50000 + 0.3   (:16)
    pose_p2c   [N, 4, 4]     projector -> camera pose
The reference wraps this in a torch DataLoader with four worker processes; here a rank reads only
its own share (samples rank, rank + world, ...; the same per-rank batch on every rank, remainder
dropped like the reference's drop_last), shuffles it with a seeded generator per epoch, stages each
batch in pinned memory and hands out device tensors uploaded on a side stream one batch ahead.
"""
from __future__ import annotations

import glob
import os

import numpy as np
import torch


def decode_depth(a: np.ndarray) -> np.ndarray:
    """uint16 codes -> metres (value / 50000 + 0.3); float input is passed through as float32."""
    if a.dtype == np.uint16:
        return a.astype(np.float32) / np.float32(50000.0) + np.float32(0.3)
    return a.astype(np.float32, copy=False)


def find_split(root_path: str, mode: str = "train") -> str:
    """First file matching the split's prefix, as read_data does (:53-65)."""
    prefix = {"train": "train", "valid": "valid"}.get(mode, "test")
    hits = sorted(glob.glob(os.path.join(root_path, prefix + "*.npz")))
    if not hits:
        raise FileNotFoundError(f"No {mode} data files found in {root_path}")
    return hits[0]


class CaptureShard:
    """One rank's share of a split.  Iterating yields dicts of device tensors
    {'z_p': [B,H,W,1], 'z_c': [B,H,W,1], 'pose_p2c': [B,4,4]} (fp32), `len()` batches per epoch."""

    def __init__(self, npz_path: str, batch: int, rank: int = 0, world: int = 1, device=None, shuffle: bool = True,
                 seed: int = 0):
        if batch <= 0 or not (0 <= rank < world):
            raise ValueError(f"batch {batch}, rank {rank}, world {world}")
        with np.load(npz_path) as data:                      # numpy's own loader: arrays only, nothing is unpickled
            n = int(data["pose_p2c"].shape[0])
            per_rank = (n // world // batch) * batch         # same count on every rank
            self.index = np.arange(rank, n, world)[:per_rank]
            self.z_p = decode_depth(data["z_p"][self.index])
            self.z_c = decode_depth(data["z_c"][self.index])
            self.pose = data["pose_p2c"][self.index].astype(np.float32)
        if self.z_p.ndim == 3:
            self.z_p, self.z_c = self.z_p[..., None], self.z_c[..., None]
        self.batch, self.shuffle, self.seed, self.epoch = batch, shuffle, seed + 7919 * rank, 0
        self.device = torch.device(device) if device is not None else None
        self._stream = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == "cuda" else None

    def __len__(self):
        return len(self.index) // self.batch

    def _host_batch(self, sel):
        out = {"z_p": torch.from_numpy(self.z_p[sel]), "z_c": torch.from_numpy(self.z_c[sel]),
               "pose_p2c": torch.from_numpy(self.pose[sel])}
        if self._stream is not None:
            out = {k: v.pin_memory() for k, v in out.items()}
        return out

    def _upload(self, host):
        if self.device is None:
            return host
        if self._stream is None:
            return {k: v.to(self.device) for k, v in host.items()}
        with torch.cuda.stream(self._stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}
        return dev

    def __iter__(self):
        order = np.arange(len(self.index))
        if self.shuffle:
            np.random.RandomState(self.seed + self.epoch).shuffle(order)
        self.epoch += 1
        nb = len(self)
        nxt = self._upload(self._host_batch(np.sort(order[:self.batch]))) if nb else None
        for i in range(nb):
            cur = nxt
            if self._stream is not None:                     # the batch about to be used must have landed
                torch.cuda.current_stream(self.device).wait_stream(self._stream)
                for v in cur.values():
                    v.record_stream(torch.cuda.current_stream(self.device))
            if i + 1 < nb:
                nxt = self._upload(self._host_batch(np.sort(order[(i + 1) * self.batch:(i + 2) * self.batch])))
            yield cur


def read_data(root_path: str, batchsize: int, mode: str = "train", rank: int = 0, world: int = 1, device=None,
              seed: int = 0) -> CaptureShard:
    """Counterpart of data_provider_pytorch.read_data (:40-79): shuffled for 'train', sequential otherwise,
    incomplete last batch dropped; `batchsize` is per rank."""
    return CaptureShard(find_split(root_path, mode), batchsize, rank, world, device, shuffle=(mode == "train"), seed=seed)


def write_npz(path: str, z_p: np.ndarray, z_c: np.ndarray, pose_p2c: np.ndarray, as_uint16: bool = False) -> None:
    """Writes a split in the format above (used to turn synthetic scenes into data sets and by the tests)."""
    if as_uint16:
        enc = lambda z: np.clip(np.rint((z.astype(np.float64) - 0.3) * 50000.0), 0, 65535).astype(np.uint16)  # noqa: E731
        z_p, z_c = enc(z_p), enc(z_c)
    np.savez(path, z_p=z_p, z_c=z_c, pose_p2c=pose_p2c.astype(np.float32))
