"""Reader for the reference's capture data sets (data_provider_pytorch.py:12-79).

File format (one `.npz` per split, names `train*.npz` / `valid*.npz` / `test*.npz`):
    z_p, z_c   [N, H, W, 1]  projector- / camera-frame depth in metres
    pose_p2c   [N, 4, 4]     projector -> camera pose
Like the reference (:32-36) the stored values are handed out as they are, cast to float32: its
docstring (:14) says the uint16 codes of the original recording were turned into metres
(code / 50000 + 0.3) when the .npz was written.  For files that still hold the raw uint16 codes
pass decode_uint16=True (a deliberate extension; the reference would return the codes as floats).

The reference wraps this in a torch DataLoader with four worker processes; here a rank reads only
its own share (samples rank, rank + world, ...; the same per-rank batch on every rank, remainder
dropped like the reference's drop_last), shuffles it with a seeded generator per epoch, stages each
batch in pinned memory and hands out device tensors uploaded on a side stream one batch ahead.
Arrays stored uncompressed in the .npz (np.savez) are memory-mapped and converted batch by batch, so
host memory holds the batches in flight, not the split.
"""
from __future__ import annotations

import glob
import os
import zipfile

import numpy as np
import torch


def decode_depth(a: np.ndarray, decode_uint16: bool = False) -> np.ndarray:
    """Stored depth -> float32.  Default: the values as stored (what the reference's loader returns);
    decode_uint16: uint16 codes of the original recording -> metres (code / 50000 + 0.3)."""
    if decode_uint16 and a.dtype == np.uint16:
        return a.astype(np.float32) / np.float32(50000.0) + np.float32(0.3)
    return a.astype(np.float32, copy=False)


def open_member(npz_path: str, name: str) -> np.ndarray:
    """Array `name` of an .npz: memory-mapped when the archive stores it uncompressed, loaded otherwise.
    Only numpy's own array format is parsed; nothing is unpickled."""
    with zipfile.ZipFile(npz_path) as zf:
        info = zf.getinfo(name + ".npy")
        if info.compress_type == zipfile.ZIP_STORED:
            with open(npz_path, "rb") as f:
                f.seek(info.header_offset)
                hdr = f.read(30)                                   # local file header: name and extra lengths at 26, 28
                n_name, n_extra = int.from_bytes(hdr[26:28], "little"), int.from_bytes(hdr[28:30], "little")
                f.seek(info.header_offset + 30 + n_name + n_extra)
                version = np.lib.format.read_magic(f)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                         else np.lib.format.read_array_header_2_0(f))
                if not dtype.hasobject:
                    return np.memmap(npz_path, dtype=dtype, mode="r", offset=f.tell(), shape=shape,
                                     order="F" if fortran else "C")
    with np.load(npz_path) as data:
        return data[name]


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
                 seed: int = 0, decode_uint16: bool = False):
        if batch <= 0 or not (0 <= rank < world):
            raise ValueError(f"batch {batch}, rank {rank}, world {world}")
        self._zp, self._zc = open_member(npz_path, "z_p"), open_member(npz_path, "z_c")
        self.pose = np.asarray(open_member(npz_path, "pose_p2c")).astype(np.float32)
        n = int(self.pose.shape[0])
        per_rank = (n // world // batch) * batch             # same count on every rank
        self.index = np.arange(rank, n, world)[:per_rank]
        self.decode_uint16 = bool(decode_uint16)
        self.batch, self.shuffle, self.seed, self.epoch = batch, shuffle, seed + 7919 * rank, 0
        self.device = torch.device(device) if device is not None else None
        self._stream = torch.cuda.Stream(self.device) if self.device is not None and self.device.type == "cuda" else None

    def __len__(self):
        return len(self.index) // self.batch

    def _depth(self, store, rows):
        z = decode_depth(np.asarray(store[rows]), self.decode_uint16)
        return np.ascontiguousarray(z[..., None] if z.ndim == 3 else z)

    def _host_batch(self, sel):
        rows = self.index[sel]                               # sel indexes this rank's share, ascending
        out = {"z_p": torch.from_numpy(self._depth(self._zp, rows)), "z_c": torch.from_numpy(self._depth(self._zc, rows)),
               "pose_p2c": torch.from_numpy(self.pose[rows])}
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
              seed: int = 0, decode_uint16: bool = False) -> CaptureShard:
    """Counterpart of data_provider_pytorch.read_data (:40-79): shuffled for 'train', sequential otherwise,
    incomplete last batch dropped; `batchsize` is per rank."""
    return CaptureShard(find_split(root_path, mode), batchsize, rank, world, device, shuffle=(mode == "train"), seed=seed,
                        decode_uint16=decode_uint16)


def write_npz(path: str, z_p: np.ndarray, z_c: np.ndarray, pose_p2c: np.ndarray, as_uint16: bool = False) -> None:
    """Writes a split in the format above (used to turn synthetic scenes into data sets and by the tests)."""
    if as_uint16:
        enc = lambda z: np.clip(np.rint((z.astype(np.float64) - 0.3) * 50000.0), 0, 65535).astype(np.uint16)  # noqa: E731
        z_p, z_c = enc(z_p), enc(z_c)
    np.savez(path, z_p=z_p, z_c=z_c, pose_p2c=pose_p2c.astype(np.float32))
