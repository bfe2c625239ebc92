"""Multi-GPU harness: one process per GPU, slides sharded round-robin, ONE all-gather (+ a 3-integer all-reduce for its shape) at the end.

The reference has no distributed code (an unused ``import torch.distributed`` and two ineffective
``nn.DataParallel`` wraps, extract_features_fp.py:217-218).  Regions are independent through
HIPT_4K and slides are independent through CLAM_SB, so the data path needs no
collective; the only exchange is collecting the per-slide outputs (logits [C] and the attention
logits A_raw [n_i], n_i varying per slide) on every rank.  Over RCCL/xGMI that is a single
latency-bound all-gather of a padded buffer — no all-reduce, no ring, no gradient traffic.

Works with any initialised ``torch.distributed`` backend (``nccl`` = RCCL on the GPU node,
``gloo`` in the CPU tests) and degrades to a no-op for a single process.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None, single_rank_group: bool | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the process group for WORLD_SIZE > 1.
    ONE rank gets NO group by default (schedulers and wrappers export RANK / WORLD_SIZE / MASTER_PORT to single-process jobs too: a library
    must not answer that with an RCCL communicator and a TCP rendezvous).  ``single_rank_group=True`` (or ``HIPT_SINGLE_RANK_GROUP=1``) opts in,
    when a launcher set the environment up (RANK + WORLD_SIZE + MASTER_PORT present): `python -m torch.distributed.run --nproc-per-node 1
    bench.py` then carries its gather over a real 1-rank RCCL communicator -- how the collective path is exercised on a one-GPU box.
    Without a group the gather is a local copy.  RCCL needs dmabuf IPC on this pool, hence HSA_ENABLE_IPC_MODE_LEGACY=0."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if single_rank_group is None:
        single_rank_group = os.environ.get("HIPT_SINGLE_RANK_GROUP", "0") not in ("", "0")
    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
    if (world > 1 or (launched and single_rank_group)) and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def grouped() -> bool:
    """Is a process group alive?  (Then every exchange below is a real collective, whatever the world size.)"""
    return dist.is_available() and dist.is_initialized()


def group_info() -> dict:
    """{"backend": "nccl" | "gloo", "ranks": world size} of the live process group, {} without one.  "nccl" IS RCCL on ROCm."""
    return {"backend": str(dist.get_backend()), "ranks": int(dist.get_world_size())} if grouped() else {}


def world() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_slides(n_slides: int, rank: int, world_size: int) -> List[int]:
    """Slide i is processed by rank i mod G (BASELINE config 5)."""
    return list(range(rank, n_slides, world_size))


def owner_of(slide: int, world_size: int) -> int:
    return slide % world_size


def gather_slide_outputs(slide_ids: Sequence[int], logits: Sequence[torch.Tensor], a_raw: Sequence[torch.Tensor],
                         n_slides: int, device=None):
    """All-gather the per-slide outputs of every rank.

    ``logits[j]`` is [C] (or [1,C]) and ``a_raw[j]`` is [n_j] (or [1,n_j]) for local slide
    ``slide_ids[j]``.  Returns ``(all_logits [n_slides, C], all_a_raw: list of [n_i] tensors)``
    ordered by global slide id, identical on every rank.  Two collectives in all: one 3-integer
    all-reduce (MAX) that agrees on the block shape (slides per rank, longest bag, classes), then ONE
    all-gather that carries everything -- each rank contributes a [S, 4 + C + n_max] fp32 block whose
    first four words are NOT floats but the int64 slide id and the int64 length n_i as raw bits
    (two 32-bit words each: an id or a length above 2**24 would not survive a float), followed by
    the logits and the zero-padded A_raw."""
    if device is None:
        device = logits[0].device if len(logits) else torch.device("cpu")
    C = int(logits[0].numel()) if len(logits) else 0
    s_local = len(slide_ids)
    n_local_max = max([int(a.numel()) for a in a_raw], default=0)
    ws = world()
    meta = torch.tensor([s_local, n_local_max, C], dtype=torch.int64, device=device)
    if grouped():  # (a 1-rank group too: the same two collectives, over a one-rank communicator)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    S, n_max, C = (int(v) for v in meta.tolist())
    HDR = 4  # int64 id | int64 length, bit-cast into four fp32 slots
    # The block is assembled with a constant number of device operations whatever the slide count (round 6: it used to be three scalar
    # writes per slide): the header from ONE host tensor, the logits from one stack, the ragged A_raw rows from one pad.
    block = torch.zeros((S, HDR + C + n_max), dtype=torch.float32, device=device)
    hdr_h = torch.full((S, 2), -1, dtype=torch.int64)  # id -1: empty slot
    for j, sid in enumerate(slide_ids):
        hdr_h[j, 0] = int(sid)
        hdr_h[j, 1] = int(a_raw[j].numel())
    block[:, :HDR] = hdr_h.to(device).view(torch.float32)  # raw bits: the collective moves bytes, nothing interprets these words as floats
    if s_local:
        if C:
            block[:s_local, HDR:HDR + C] = torch.stack([l.reshape(-1).float() for l in logits])
        if n_local_max:
            pad = torch.nn.utils.rnn.pad_sequence([a.reshape(-1).float() for a in a_raw], batch_first=True)
            block[:s_local, HDR + C:HDR + C + pad.shape[1]] = pad
    if grouped():
        out = torch.empty((ws * S, block.shape[1]), dtype=torch.float32, device=device)  # concatenated along dim 0
        dist.all_gather_into_tensor(out, block)
    else:
        out = block
    all_logits = torch.zeros((n_slides, C), dtype=torch.float32, device=device)
    all_a: List[torch.Tensor] = [torch.empty(0, device=device) for _ in range(n_slides)]
    meta_i = out[:, :HDR].contiguous().view(torch.int64).cpu()  # ONE read-back: ids and lengths of every row
    ids, lens = meta_i[:, 0].tolist(), meta_i[:, 1].tolist()
    rows = [r for r, sid in enumerate(ids) if sid >= 0]
    if rows:
        sel = torch.tensor(rows, dtype=torch.int64, device=device)
        all_logits[torch.tensor([ids[r] for r in rows], dtype=torch.int64, device=device)] = out[sel, HDR:HDR + C]
    for r in rows:  # (views of the gathered buffer: no launch per slide)
        all_a[ids[r]] = out[r, HDR + C:HDR + C + lens[r]]
    return all_logits, all_a
