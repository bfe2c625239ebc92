"""Feature store either side of the hot path.

Writer = what ``compute_w_loader`` + ``save_hdf5`` + the ``.pt`` export do in the reference
(``extract_features_fp.py:159-171,248-255``, ``utils/file_utils.py:16-35``): per slide, ``features [n, 192]``
fp32 and ``coords [n, 2]`` appended batch by batch, then ``pt_files/{slide}.pt`` = ``torch.save(features)``.
Reader = the bag loader of ``datasets/dataset_generic.py:505-528``: ``torch.load`` of that file and, when the slide
has more than ``max_patches_per_slide`` rows, ``np.random.choice(n, max)`` (WITH replacement, as the reference)
rows of it.

``coords`` are integer data and pass through bit for bit: whatever integer dtype the loader hands over
(the reference's h5 ``coords`` dataset: int64, or int32 from older patch files) is what is stored.  The reference keeps them
in ``h5_files/{slide}.h5`` beside the features; ``h5py`` is not part of every host (this image has none), so the coordinates
are ALWAYS written to a sidecar ``coords_files/{slide}.npy`` (plain ``numpy.save`` of the ``[n, 2]`` array, same row order as
``pt_files/{slide}.pt``; ``load_coords`` reads it back) and the ``.h5`` twin (datasets ``features`` / ``coords``, chunks
``(1, .)``, resizable first axis: ``utils/file_utils.py:16-35``) in addition where ``h5py`` is importable.  The directory is a
sibling of ``pt_files/`` because the reference lists that directory to find finished slides
(``extract_features_fp.py:231-238``) and loads bags from it by name: nothing else may live there.
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
        c = coords.detach().cpu().numpy() if torch.is_tensor(coords) else np.asarray(coords)
        if f.dim() != 2 or c.ndim != 2 or f.shape[0] != c.shape[0]:
            raise ValueError(f"features {tuple(f.shape)} / coords {tuple(c.shape)}: expected [n, d] and [n, 2]")
        if c.dtype.kind not in "iu":
            raise TypeError(f"slide {self.slide_id}: coords must be integers (got {c.dtype}); they are stored bit for bit, never rounded")
        if self._coords and (c.dtype != self._coords[0].dtype or c.shape[1:] != self._coords[0].shape[1:]):
            # (the reference's resizable h5 dataset keeps the dtype of the first batch and would cast silently: refuse instead)
            raise TypeError(f"slide {self.slide_id}: coords {c.dtype}{list(c.shape[1:])} after {self._coords[0].dtype}{list(self._coords[0].shape[1:])}")
        self._feats.append(f)
        self._coords.append(np.array(c, copy=True))  # (the caller may reuse its buffer)

    def __len__(self) -> int:
        return sum(f.shape[0] for f in self._feats)

    def close(self) -> str:
        if not self._feats:
            raise ValueError(f"slide {self.slide_id}: nothing was appended")
        feats, coords = torch.cat(self._feats, 0), np.concatenate(self._coords, 0)
        for d in ("pt_files", "coords_files"):
            os.makedirs(os.path.join(self.feat_dir, d), exist_ok=True)
        # coordinates first: a slide counts as finished when its .pt exists (the auto-skip looks at pt_files only), so the .pt is
        # written last and both files go through a rename -- a run killed in between leaves no half-written finished slide
        cpath = coords_path(self.feat_dir, self.slide_id)
        with open(cpath + ".tmp", "wb") as fh:
            np.save(fh, coords, allow_pickle=False)
        os.replace(cpath + ".tmp", cpath)
        if self.write_h5:
            import h5py
            os.makedirs(os.path.join(self.feat_dir, "h5_files"), exist_ok=True)
            with h5py.File(os.path.join(self.feat_dir, "h5_files", self.slide_id + ".h5"), "w") as fh:
                for key, val in (("features", feats.numpy()), ("coords", coords)):
                    fh.create_dataset(key, data=val, maxshape=(None,) + val.shape[1:], chunks=(1,) + val.shape[1:])
        pt = os.path.join(self.feat_dir, "pt_files", self.slide_id + ".pt")
        torch.save(feats, pt + ".tmp")  # extract_features_fp.py:255: the tensor itself, nothing else
        os.replace(pt + ".tmp", pt)
        return pt


def coords_path(feat_dir: str, slide_id: str) -> str:
    return os.path.join(feat_dir, "coords_files", slide_id + ".npy")


def load_coords(feat_dir: str, slide_id: str) -> np.ndarray:
    """The slide's ``coords [n, 2]`` exactly as they were appended (dtype and bits); row i belongs to row i of
    ``pt_files/{slide}.pt``.  Reads the sidecar, or the reference's ``h5_files/{slide}.h5`` where only that exists."""
    p = coords_path(feat_dir, slide_id)
    if os.path.isfile(p):
        return np.load(p, allow_pickle=False)
    h5 = os.path.join(feat_dir, "h5_files", slide_id + ".h5")
    if os.path.isfile(h5):
        try:
            import h5py
        except ImportError as e:
            raise FileNotFoundError(f"{p} is missing and {h5} needs h5py, which this host lacks") from e
        with h5py.File(h5, "r") as fh:
            return fh["coords"][:]
    raise FileNotFoundError(f"no coordinates for slide {slide_id} under {feat_dir} (coords_files/ or h5_files/)")


class _HostFeed:
    """The H2D hop of the driver loop (extract_features_fp.py:162-166: ``batch = batch.to(device, non_blocking=True)`` from whatever memory
    the loader hands over; HIPT_4K/hipt_4k.py:69).  Loader batches that live in HOST memory are copied into one of two device gather
    buffers on a COPY stream while the previous gathered call computes: pinned sources (``DataLoader(pin_memory=True)``) straight from
    where they lie, pageable ones through a pinned staging buffer of the same shape (one host memcpy, then the same asynchronous copy).
    Events order the three parties: a gather buffer is refilled only after the call that read it has finished (``free``), a call starts only
    after its copies have landed (``ready``), a staging buffer is rewritten only after its last copy out has completed.  No extra device
    copy: the gather buffer IS the tensor the model call reads."""

    SLOTS = 2

    def __init__(self, device: torch.device):
        self.device = device
        # HIGH-PRIORITY streams: HIP multiplexes a process's streams over a few hardware queues (four by default), and a copy whose stream shares
        # an in-order hardware queue with a compute stream starts only behind the ~26 ms of kernels already enqueued there -- measured in
        # bench.py's process (14 streams alive): 33.3 ms per gathered call against 27.3 for the same loop in a fresh process.  Priority
        # streams get hardware queues of their own.
        self.copy_stream = torch.cuda.Stream(device=device, priority=-1)
        self.buf = [None] * self.SLOTS       # device gather buffers [cap, ...]
        self.stage = [None] * self.SLOTS     # pinned staging twins (allocated only when a pageable batch arrives)
        self.free = [None] * self.SLOTS      # event: the call that read buf[s] has finished
        self.staged = [None] * self.SLOTS    # event: the last copy out of stage[s] has completed
        self.slot = 0
        self.fill = 0
        self.read_stream = torch.cuda.Stream(device=device, priority=-1)  # features come back on their own stream, behind THEIR call only
        self._hbuf = None

    def read_back(self, feats: torch.Tensor, done: "torch.cuda.Event") -> torch.Tensor:
        """features of a finished call as a host tensor, without waiting for anything enqueued after that call (a `.cpu()` on the compute
        stream would wait for the NEXT call too, and the host could not feed the one after it meanwhile)"""
        if self._hbuf is None or self._hbuf.shape[1:] != feats.shape[1:] or self._hbuf.shape[0] < feats.shape[0] or self._hbuf.dtype != feats.dtype:
            self._hbuf = torch.empty((max(feats.shape[0], 8),) + tuple(feats.shape[1:]), dtype=feats.dtype, pin_memory=True)
        self.read_stream.wait_event(done)
        with torch.cuda.stream(self.read_stream):
            self._hbuf[:feats.shape[0]].copy_(feats, non_blocking=True)
        feats.record_stream(self.read_stream)
        self.read_stream.synchronize()
        return self._hbuf[:feats.shape[0]].clone()

    def begin(self, cap: int, like: torch.Tensor):
        """start gathering up to `cap` regions shaped like `like` ([r, ...]) into the current slot"""
        s = self.slot
        shape = (cap,) + tuple(like.shape[1:])
        if self.buf[s] is None or self.buf[s].shape != shape or self.buf[s].dtype != like.dtype:
            self.buf[s] = torch.empty(shape, dtype=like.dtype, device=self.device)
            self.stage[s] = None
        if self.free[s] is not None:
            self.copy_stream.wait_event(self.free[s])  # the compute that read this buffer two calls ago
        self.fill = 0

    def add(self, r: torch.Tensor):
        s, o, n = self.slot, self.fill, r.shape[0]
        dst = self.buf[s][o:o + n]
        src = r
        if not r.is_pinned():
            if self.stage[s] is None:
                self.stage[s] = torch.empty(self.buf[s].shape, dtype=self.buf[s].dtype, pin_memory=True)
            if self.staged[s] is not None and o == 0:
                self.staged[s].synchronize()  # (long complete: its call has been launched and usually finished)
            src = self.stage[s][o:o + n]
            src.copy_(r)  # host memcpy: pageable -> pinned
        with torch.cuda.stream(self.copy_stream):
            dst.copy_(src, non_blocking=True)
        self.fill = o + n

    def capacity(self) -> int:
        return 0 if self.buf[self.slot] is None else self.buf[self.slot].shape[0]

    def finish(self):
        """(the gathered regions as a device tensor the CURRENT stream may read, its slot); switches to the other slot"""
        s = self.slot
        ev = torch.cuda.Event()
        ev.record(self.copy_stream)
        self.staged[s] = ev
        torch.cuda.current_stream(self.device).wait_event(ev)
        out = self.buf[s][:self.fill]
        self.slot = (s + 1) % self.SLOTS
        return out, s

    def release(self, s: int):
        """the call that read slot s has been enqueued on the current stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.free[s] = ev


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
    loader batch.

    **Host batches** (round 6; the reference's loader yields CPU tensors and moves each to the device inside the loop,
    extract_features_fp.py:162-166): when ``model`` is a ``HIPT_4K`` on a HIP device and a loader batch is not there, the batch
    is copied host -> device AS IT ARRIVES, on a copy stream, into one of two gather buffers while the previous call computes (``_HostFeed``),
    and the features of call k are read back (on a stream of their own) only after call k + 1 has been enqueued -- the link, the GPU and the host loop overlap.  Same
    gathered tensor, same kernels, same bits as with resident batches."""
    w = FeatureWriter(feat_dir, slide_id)
    held: list = []  # (regions or None when already staged into the host feed's gather buffer, coords, count) waiting for company
    shape_key = [None]
    ncalls = [0]
    m256 = getattr(model, "model256", None)  # HIPT_4K: where its first-level ViT's weights live NOW (.to() may have moved it since construction)
    dev = getattr(m256, "weight_device", None) if m256 is not None else None
    dev = torch.device(dev) if dev is not None else None
    feed = [None]        # _HostFeed, made when the first host batch for a HIP model arrives
    pending: list = []   # [(features on the device, counts, coords)]: read back one call late

    def append_call(feats, counts, coords, done=None):
        if done is not None:
            feats = feed[0].read_back(feats, done)
        o = 0
        for n, c in zip(counts, coords):  # one append per loader batch, as the reference's loop does
            w.append(feats[o:o + n], c)
            o += n

    def drain():
        while pending:
            append_call(*pending.pop(0))

    def flush():
        if not held:
            return
        counts, coords = [n for _, _, n in held], [c for _, c, _ in held]
        staged = held[0][0] is None  # host batches: already on their way into the open gather buffer (stage())
        slot = None
        if staged:
            regions, slot = feed[0].finish()
        elif len(held) == 1:
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
        shape_key[0] = None
        ncalls[0] += 1
        feats = model(regions)
        if slot is not None:
            feed[0].release(slot)
        del regions
        if staged:
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(dev))
            pending.append((feats, counts, coords, done))  # read back after the NEXT call is enqueued: the GPU never waits for the host
            while len(pending) > 1:
                append_call(*pending.pop(0))
        else:
            drain()
            append_call(feats, counts, coords)

    def stage(regions) -> bool:
        """a HOST batch for a model on a HIP device: its copy starts NOW, into the open gather buffer (opened here if none is)"""
        if not (dev is not None and dev.type == "cuda" and regions.device.type == "cpu"):
            return False
        if feed[0] is None:
            feed[0] = _HostFeed(dev)
        f, n = feed[0], regions.shape[0]
        if not held:
            # capacity: what a gathered call reaches with loader batches of this size (the batch that crosses `coalesce` is not split), so that
            # ragged tails re-use the buffer (a later, larger batch that would overrun it starts a new call); a shape that is not gathered: this batch alone
            f.begin(coalesce + n - 1 if (coalesce > 1 and gathers_bit_exactly(regions)) else n, regions)
        f.add(regions)
        return True

    def gathers_bit_exactly(regions) -> bool:
        # whole 16-row fragments in every call (patch count a multiple of 16): the gathered call and the one-by-one call take the
        # same kernels and write the same bits; other region sizes are NOT gathered (they would agree to the bf16 bar only)
        if regions.dim() != 4:
            return False
        hw = regions.shape[2:] if regions.shape[1] == 3 else regions.shape[1:3]  # planar [R,3,W,H] or interleaved [R,W,H,3]
        return ((hw[0] // 256) * (hw[1] // 256)) % 16 == 0

    with torch.no_grad():
        for regions, coords in batches:
            key = (tuple(regions.shape[1:]), regions.dtype, regions.device)
            if held and key != shape_key[0]:
                flush()  # a different shape / type cannot share a call
            if held and held[0][0] is None and feed[0].fill + regions.shape[0] > feed[0].capacity():
                flush()  # (a host batch that would overrun the open gather buffer: a loader batch larger than `coalesce` behind smaller ones)
            shape_key[0] = key
            held.append((None if stage(regions) else regions, coords, regions.shape[0]))
            # (host batches: the FIRST call of a slide is a short one -- a quarter of `coalesce` -- so that the GPU starts behind 2 regions' copies
            #  instead of 8; nothing hides the first call's copy, every later one lands under the call before it.  A region's bits do not depend
            #  on the call it shares.)
            want = max(1, coalesce // 4) if (held[0][0] is None and ncalls[0] == 0) else coalesce
            if coalesce <= 1 or not gathers_bit_exactly(regions) or sum(n for _, _, n in held) >= want:
                flush()
        flush()
        drain()
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
