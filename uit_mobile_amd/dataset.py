"""Input side of the path: where the clips of a batch come from.

The reference's evaluation reads int16 clips out of HDF5 files through DataLoader worker processes, converts each to float32
(`x / 32768`, dataset.py:40-46) and zero-pads a batch to its longest clip (`pad` / `sequential_pad`, dataset.py:183-204).  Here the
clips stay int16 until the front-end kernel (the scale is fused into its load), the batch is assembled directly in pinned host memory
by the producer threads of `streaming.PinnedBatchFeeder`, and the H2D copy of batch i+1 runs under the kernels of batch i:

    store  = ClipStore.from_hdf5_table(df)            # the reference's label table (filename, hdf5path); needs h5py
    store  = ClipStore.from_npy_dir("clips/")         # or a directory of int16 / float .npy clips
    feeder = store.feeder(batch_size=8192, clip_samples=16000, producers=8)
    for (first, count, names), probs in zip(store.batch_index(8192), HostClipStreamer(model, 8192, 16000).run(feeder)):
        use(probs[:count], names, store.lengths(first))

Nothing here touches the device; the HDF5 backend is imported lazily (h5py is not part of this image, the .npy backend is what the
tests run).
"""
import threading
from pathlib import Path

import numpy as np

from .streaming import PinnedBatchFeeder


def to_int16(clip):
    """What the kernel's int16 input means: sample / 32768.  Integer clips pass through (the reference's stored format,
    dump_audio_to_hdf5.py); float clips in [-1, 1) are rounded to the int16 grid (exact for clips that were int16 to begin with)."""
    a = np.asarray(clip)
    if a.dtype == np.int16:
        return a
    if np.issubdtype(a.dtype, np.integer):
        return np.clip(a, -32768, 32767).astype(np.int16)
    return np.clip(np.rint(a.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


class ClipStore:
    """An ordered list of clips behind a `read(i) -> 1-D array` function that must be safe to call from several threads."""

    def __init__(self, names, read):
        self.names = list(names)
        self._read = read
        self._lengths = {}

    def __len__(self):
        return len(self.names)

    def read(self, i):
        return to_int16(self._read(i)).reshape(-1)

    # ---- backends
    @classmethod
    def from_arrays(cls, clips, names=None):
        clips = list(clips)
        return cls(names or [str(i) for i in range(len(clips))], lambda i: clips[i])

    @classmethod
    def from_npy_dir(cls, path, pattern="*.npy"):
        files = sorted(Path(path).glob(pattern))
        if not files:
            raise FileNotFoundError(f"no {pattern} under {path}")
        return cls([f.stem for f in files], lambda i: np.load(files[i], mmap_mode="r"))

    @classmethod
    def from_hdf5_table(cls, table):
        """`table`: rows with `filename` and `hdf5path` (a pandas DataFrame like the reference's label files, or a list of pairs).
        One dataset per clip, keyed by file name (dataset.py:40-43).  Every reader thread keeps its own file handles."""
        try:
            import h5py
        except ImportError as e:                              # pragma: no cover (h5py is absent from the build image)
            raise ImportError("ClipStore.from_hdf5_table needs h5py; convert the clips to .npy and use from_npy_dir instead") from e
        rows = [(r["filename"], r["hdf5path"]) for _, r in table.iterrows()] if hasattr(table, "iterrows") else [tuple(r) for r in table]
        local = threading.local()

        def read(i):                                          # pragma: no cover
            name, path = rows[i]
            files = local.__dict__.setdefault("files", {})
            if path not in files:
                files[path] = h5py.File(path, "r")
            return files[path][str(name)][:]
        return cls([r[0] for r in rows], read)

    # ---- batching
    def batch_index(self, batch_size):
        """[(first clip, clips in the batch, their names)] for every batch the feeder will hand out, in order."""
        return [(first, min(batch_size, len(self) - first), self.names[first:first + batch_size]) for first in range(0, len(self), batch_size)]

    def lengths(self, first):
        """Sample counts of the clips of the batch that starts at clip `first`, as read (before truncation / padding): available once
        that batch has been filled -- `sequential_pad` returns the same vector (dataset.py:200-204)."""
        return self._lengths[first]

    def fill_fn(self, batch_size):
        """The `fill(i, out)` of PinnedBatchFeeder: clip j of batch i into row j, truncated / zero right-padded to the row (the
        reference's collate pads to the batch maximum; the row length is the model's clip length), rows behind the last clip zeroed.
        The clips' own lengths are kept per batch (`sequential_pad` returns them too)."""
        def fill(i, out):
            first = i * batch_size
            count = max(0, min(batch_size, len(self) - first))
            lengths = np.zeros(count, np.int64)
            for j in range(count):
                clip = self.read(first + j)
                lengths[j] = len(clip)
                PinnedBatchFeeder.collate(out[j], clip)
            out[count:] = 0
            self._lengths[first] = lengths
        return fill

    def feeder(self, batch_size, clip_samples, producers=4, pin=True, autostart=True):
        n_batches = (len(self) + batch_size - 1) // batch_size
        return PinnedBatchFeeder(self.fill_fn(batch_size), n_batches, batch_size, clip_samples, producers=producers, pin=pin,
                                 autostart=autostart)
