"""Slide-level driver: the per-slide loop of ``extract_features_fp.py`` followed by the slide's aggregation, sharded
over one process per GPU (BASELINE.json configs[4]; SURVEY.md §8d "config 5", §8e).

Reference shape of the loop (``extract_features_fp.py:223-255``): for every slide of the CSV, skip it when
``pt_files/{slide}.pt`` exists, else run ``compute_w_loader`` -- batches of 4096x4096 regions through ``HIPT_4K`` ->
``features [n, 192]`` + ``coords [n, 2]`` -> feature files.  ``main.py`` / ``eval.py`` later load one bag per slide and
run ``CLAM_SB`` on it (``utils/core_utils.py:409``, ``utils/eval_utils.py:134``).  The reference's "multi-GPU" is a
``nn.DataParallel`` wrap over a batch of one (``:217-218``): ineffective.  Here:

    slide i  ->  rank i mod G        (``distributed.shard_slides``; weights replicated, no data-path collective)
    per slide:  regions -> HIPT_4K (R regions per call) -> features [n, 192] (-> feature store, optional)
                -> CLAM_SB -> logits [C], A_raw [n]
    end:        ONE all-gather of every slide's logits and ragged A_raw (``distributed.gather_slide_outputs``)

The same function runs on one GPU (world 1: the gather degenerates to a local copy), under ``torch.distributed`` with the
``nccl`` (= RCCL) backend on the GPU node, and with ``gloo`` on the CPU for the plumbing tests (any callable may stand in
for the two models there: the driver itself holds no device code).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import distributed as D


@dataclass
class SlideSpec:
    """One whole-slide image as the driver sees it: an id, how many 4096x4096 regions its tissue mask produced
    (``patches/{slide}.h5`` ``coords``, wsi_core/WholeSlideImage.py:484-496) and where they sit."""
    slide_id: str
    n_regions: int
    seed: int = 0
    region_px: int = 4096
    grid_cols: int = 128  # regions per row of the slide's coordinate grid (coords are pass-through integers)

    def coords(self, idx: Sequence[int]) -> torch.Tensor:
        """[len(idx), 2] int64 top-left corners (x, y), the layout of the reference's ``coords`` dataset."""
        i = torch.as_tensor(list(idx), dtype=torch.int64)
        return torch.stack([(i % self.grid_cols) * self.region_px, (i // self.grid_cols) * self.region_px], dim=1)


def synthetic_slides(n_slides: int, n_regions: int, region_px: int = 4096, ragged: bool = True, seed: int = 0) -> List[SlideSpec]:
    """``n_slides`` synthetic slides of about ``n_regions`` regions each (ragged by up to -12 % when ``ragged``: real
    slides differ in tissue area, which is what makes the A_raw gather ragged)."""
    out = []
    for i in range(n_slides):
        n = n_regions - ((i * 2654435761 + seed) % max(1, n_regions // 8)) if ragged and n_regions >= 8 else n_regions
        out.append(SlideSpec(f"slide_{i:03d}", int(n), seed=seed * 1000 + i, region_px=region_px))
    return out


def hashed_regions(spec: SlideSpec, idx: Sequence[int], device, uint8: bool = False) -> torch.Tensor:
    """Deterministic pixels of regions ``idx`` of a slide, generated on ``device`` ([len, 3, px, px] fp32 in [-1, 1), or the
    same as raw uint8 RGB): the synthetic stand-in for ``wsi.read_region`` + ``eval_transforms`` (dataset_h5.py:194-207)."""
    from . import synth
    px = spec.region_px
    x = torch.stack([synth.hash_uniform_torch((3, px, px), 7919 * spec.seed + int(i) + 1, device=device) for i in idx])
    if uint8:
        x = ((x * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8)
    return x


def sample_indices(n_regions: int, sample: Optional[int]) -> List[int]:
    """The regions of a slide that are actually extracted: all of them, or ``sample`` evenly spaced ones."""
    if sample is None or sample >= n_regions:
        return list(range(n_regions))
    return [(j * n_regions) // sample for j in range(sample)]


@dataclass
class SlideRun:
    """What ``process_slides`` returns (identical on every rank except the ``local_*`` fields)."""
    logits: torch.Tensor                      # [n_slides, C]
    a_raw: List[torch.Tensor]                 # n_slides x [n_i]
    local_slides: List[int] = field(default_factory=list)
    local_regions: int = 0                    # regions this rank pushed through HIPT_4K
    local_features: Dict[int, torch.Tensor] = field(default_factory=dict)   # slide -> [n_s, d] (kept when keep_features)
    seconds: float = 0.0                      # wall time of this rank's loop + gather (no barrier inside)
    local_seconds: float = 0.0                # ... of its own slides alone (device work finished, before the gather): the load-imbalance figure


def process_slides(model: Callable, clam: Callable, slides: Sequence[SlideSpec], rank: int = 0, world: int = 1, *,
                   device=None, regions_per_call: int = 8, sample_regions: Optional[int] = None,
                   region_source: Optional[Callable] = None, expand_bag: bool = False, feat_dir: Optional[str] = None,
                   keep_features: bool = False, skip_existing: bool = False) -> SlideRun:
    """Run this rank's share of ``slides`` and gather every slide's outputs.

    ``model(regions [R, 3, px, px]) -> features [R, d]`` (``HIPT_4K``), ``clam(bag [n, d]) -> (logits, Y_prob, Y_hat,
    A_raw, dict)`` (``CLAM_SB``).  ``region_source(spec, idx) -> regions`` supplies pixels (default: ``hashed_regions`` on
    ``device``; the bench passes views of a resident pool so that nothing is generated inside its timed region).
    ``sample_regions``: extract only that many evenly spaced regions per slide (a full synthetic job is 64 x 8 192 x 3.15
    TFLOP: SURVEY.md §8d lets the harness time a stated sub-sample); with ``expand_bag`` the slide's bag is then tiled
    back to ``n_regions`` rows so that CLAM_SB and the gather carry their true sizes.  ``feat_dir``: also write
    ``pt_files/{slide}.pt`` through the feature store (and, with ``skip_existing``, skip slides already there, the
    reference's crude resume, extract_features_fp.py:231-238: such a slide's bag is loaded back instead)."""
    import os

    from .feature_store import FeatureWriter, load_bag
    if device is None:
        device = torch.device("cpu")
    src = region_source or (lambda spec, idx: hashed_regions(spec, idx, device))
    mine = D.shard_slides(len(slides), rank, world)
    t0 = time.perf_counter()
    run = SlideRun(logits=torch.empty(0), a_raw=[], local_slides=mine)
    lg, ar = [], []
    with torch.no_grad():
        for sid in mine:
            spec = slides[sid]
            done = feat_dir is not None and skip_existing and os.path.isfile(os.path.join(feat_dir, "pt_files", spec.slide_id + ".pt"))
            if done:
                feats = load_bag(feat_dir, spec.slide_id).to(device)
            else:
                idx = sample_indices(spec.n_regions, sample_regions)
                writer = FeatureWriter(feat_dir, spec.slide_id) if feat_dir is not None else None
                parts = []
                for b0 in range(0, len(idx), regions_per_call):
                    sub = idx[b0:b0 + regions_per_call]
                    f = model(src(spec, sub))
                    parts.append(f)
                    if writer is not None:
                        writer.append(f, spec.coords(sub))
                    run.local_regions += len(sub)
                feats = torch.cat(parts, dim=0)
                if writer is not None:
                    writer.close()
            if keep_features:
                run.local_features[sid] = feats
            bag = feats
            if expand_bag and feats.shape[0] < spec.n_regions:
                reps = -(-spec.n_regions // feats.shape[0])
                bag = feats.repeat(reps, 1)[:spec.n_regions].contiguous()
            logits, _, _, a_raw, _ = clam(bag)
            lg.append(logits.reshape(-1))
            ar.append(a_raw.reshape(-1))
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)  # (this rank's share is DONE here: what min / max over the ranks compare)
    run.local_seconds = time.perf_counter() - t0
    run.logits, run.a_raw = D.gather_slide_outputs(mine, lg, ar, len(slides), device=device)
    run.seconds = time.perf_counter() - t0
    return run
