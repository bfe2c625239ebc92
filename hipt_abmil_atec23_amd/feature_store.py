"""Feature store either side of the hot path.

Writer = what ``compute_w_loader`` + ``save_hdf5`` + the ``.pt`` export do in the reference
(``extract_features_fp.py:159-171,248-255``, ``utils/file_utils.py:16-35``): per slide, ``features [n, 192]``
fp32 and ``coords [n, 2]`` appended batch by batch, then ``pt_files/{slide}.pt`` = ``torch.save(features)``.
Reader = the bag loader of ``datasets/dataset_generic.py:505-528``: ``torch.load`` of that file and, when the slide
has more than ``max_patches_per_slide`` rows, ``np.random.choice(n, max)`` (WITH replacement, as the reference)
rows of it.

The ``h5_files/{slide}.h5`` twin (datasets ``features`` / ``coords``, chunks ``(1, .)``, resizable first axis) is
written only when ``h5py`` is importable; this image has none, and nothing on the GPU path reads it.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Tuple

import numpy as np
import torch


class FeatureWriter:
    """Accumulates one slide's region features / coordinates; ``close()`` writes the files."""

    def __init__(self, feat_dir: str, slide_id: str, write_h5: Optional[bool] = None):
        self.feat_dir, self.slide_id = feat_dir, slide_id
        self._feats, self._coords = [], []
        if write_h5 is None:
            try:
                import h5py  # noqa: F401
                write_h5 = True
            except ImportError:
                write_h5 = False
        self.write_h5 = write_h5

    def append(self, features, coords) -> None:
        f = features.detach().float().cpu() if torch.is_tensor(features) else torch.as_tensor(np.asarray(features), dtype=torch.float32)
        c = coords.detach().cpu() if torch.is_tensor(coords) else torch.as_tensor(np.asarray(coords))
        if f.dim() != 2 or c.dim() != 2 or f.shape[0] != c.shape[0]:
            raise ValueError(f"features {tuple(f.shape)} / coords {tuple(c.shape)}: expected [n, d] and [n, 2]")
        self._feats.append(f)
        self._coords.append(c.to(torch.int64))

    def __len__(self) -> int:
        return sum(f.shape[0] for f in self._feats)

    def close(self) -> str:
        if not self._feats:
            raise ValueError(f"slide {self.slide_id}: nothing was appended")
        feats, coords = torch.cat(self._feats, 0), torch.cat(self._coords, 0)
        os.makedirs(os.path.join(self.feat_dir, "pt_files"), exist_ok=True)
        pt = os.path.join(self.feat_dir, "pt_files", self.slide_id + ".pt")
        torch.save(feats, pt)  # extract_features_fp.py:255: the tensor itself, nothing else
        if self.write_h5:
            import h5py
            os.makedirs(os.path.join(self.feat_dir, "h5_files"), exist_ok=True)
            with h5py.File(os.path.join(self.feat_dir, "h5_files", self.slide_id + ".h5"), "w") as fh:
                for key, val in (("features", feats.numpy()), ("coords", coords.numpy())):
                    fh.create_dataset(key, data=val, maxshape=(None,) + val.shape[1:], chunks=(1,) + val.shape[1:])
        return pt


def extract_slide(model, batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], feat_dir: str, slide_id: str, coalesce: int = 8) -> str:
    """The loop of ``compute_w_loader`` (extract_features_fp.py:159-171): ``batches`` yields ``(regions, coords)``;
    regions are whatever ``model`` takes — here ``[R, 3, W, H]`` float or raw ``uint8`` (planar or interleaved), R >= 1.

    The reference's loader yields ONE region per batch (``batch_size = 1``, extract_features_fp.py:128), and one region per
    call leaves the GPU a third idle (tile quantisation + launch latency of the small kernels: 175 regions/s against 285 at
    8+ regions per call).  Consecutive loader batches of the same shape and type are therefore gathered until ``coalesce``
    regions are at hand and go through ``model`` in ONE call; features and coordinates are appended in loader order, and a
    region's features do not depend on what else is in the call (rows are independent through every kernel), so the saved
    files are the ones the one-by-one loop writes -- bit for bit: only regions whose patch count is a multiple of 16 (4096 x 4096:
    256) are gathered, because both calls then take the same kernels (tested); other sizes go one loader batch per call.
    ``coalesce <= 1`` restores the one-by-one loop for everything.  Peak memory of a gathered call: the gathered batch plus one
    loader batch."""
    w = FeatureWriter(feat_dir, slide_id)
    held: list = []  # (regions, coords) waiting for company

    def flush():
        if not held:
            return
        counts, coords = [r.shape[0] for r, _ in held], [c for _, c in held]
        if len(held) == 1:
            regions = held[0][0]
        else:
            # gathered in ONE buffer, each loader batch released as soon as it is copied: the peak is the gathered batch plus one
            # loader batch (8 fp32 4096 x 4096 regions: 1.6 GB + 0.2 GB), not two copies of everything
            regions = torch.empty((sum(counts),) + tuple(held[0][0].shape[1:]), dtype=held[0][0].dtype, device=held[0][0].device)
            o = 0
            for i in range(len(held)):
                r = held[i][0]
                regions[o:o + r.shape[0]].copy_(r)
                o += r.shape[0]
                held[i] = None
                del r
        held.clear()
        feats = model(regions)
        del regions
        o = 0
        for n, c in zip(counts, coords):  # one append per loader batch, as the reference's loop does
            w.append(feats[o:o + n], c)
            o += n

    def gathers_bit_exactly(regions) -> bool:
        # whole 16-row fragments in every call (patch count a multiple of 16): the gathered call and the one-by-one call take the
        # same kernels and write the same bits; other region sizes are NOT gathered (they would agree to the bf16 bar only)
        if regions.dim() != 4:
            return False
        hw = regions.shape[2:] if regions.shape[1] == 3 else regions.shape[1:3]  # planar [R,3,W,H] or interleaved [R,W,H,3]
        return ((hw[0] // 256) * (hw[1] // 256)) % 16 == 0

    with torch.no_grad():
        for regions, coords in batches:
            if held and (regions.shape[1:] != held[0][0].shape[1:] or regions.dtype != held[0][0].dtype or regions.device != held[0][0].device):
                flush()  # a different shape / type cannot share a call
            held.append((regions, coords))
            if coalesce <= 1 or not gathers_bit_exactly(regions) or sum(r.shape[0] for r, _ in held) >= coalesce:
                flush()
        flush()
    return w.close()


def load_bag(data_dir: str, slide_id: str, max_patches_per_slide: Optional[int] = None, rng=None) -> torch.Tensor:
    """``datasets/dataset_generic.py:512-520``: the slide's ``[n, d]`` features, sub-sampled WITH replacement to
    ``max_patches_per_slide`` rows when it has more (``rng``: a ``numpy.random.Generator`` / ``RandomState`` for
    reproducible draws; default ``np.random`` as the reference)."""
    path = os.path.join(data_dir, "pt_files", f"{slide_id}.pt")
    try:
        features = torch.load(path)
    except Exception as e:  # the reference asserts with the slide name
        raise AssertionError(f"Error caused by slide {slide_id}") from e
    if max_patches_per_slide is not None and max_patches_per_slide < len(features):
        idx = (rng if rng is not None else np.random).choice(len(features), max_patches_per_slide)
        features = features[torch.as_tensor(np.asarray(idx), dtype=torch.int64)]
    return features
