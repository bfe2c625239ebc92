"""N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Exercises slide sharding
(slide i -> rank i mod G) and the one all-gather of per-slide logits / ragged attention logits."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slide(i):
    """Deterministic fake per-slide outputs: logits [2], A_raw [n_i] with ragged n_i."""
    n = 5 + 7 * (i % 4)
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(2, generator=g), torch.randn(n, generator=g)


def _worker(rank, world, port, n_slides, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from hipt_abmil_atec23_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo")
    mine = D.shard_slides(n_slides, r, w)
    lg, ar = zip(*[_slide(i) for i in mine]) if mine else ((), ())
    all_logits, all_a = D.gather_slide_outputs(mine, list(lg), list(ar), n_slides, device=torch.device("cpu"))
    ok = True
    for i in range(n_slides):
        l, a = _slide(i)
        ok &= torch.equal(all_logits[i], l) and torch.equal(all_a[i], a)
    q.put((rank, mine, bool(ok)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("n_slides", [7, 2, 1])
def test_gather_slide_outputs_world2(n_slides):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_slides, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res[0][1] == list(range(0, n_slides, 2)) and res[1][1] == list(range(1, n_slides, 2))  # slide i -> rank i mod G
    assert all(r[2] for r in res)


def test_single_process_degenerates_to_noop():
    from hipt_abmil_atec23_amd import distributed as D
    assert D.world() == 1 and D.shard_slides(5, 0, 1) == [0, 1, 2, 3, 4] and D.owner_of(11, 8) == 3
    lg, ar = zip(*[_slide(i) for i in range(3)])
    all_logits, all_a = D.gather_slide_outputs([0, 1, 2], list(lg), list(ar), 3)
    assert all(torch.equal(all_logits[i], lg[i]) and torch.equal(all_a[i], ar[i]) for i in range(3))


def test_gather_carries_ids_and_lengths_as_integers():
    """Slide ids and bag lengths travel as int64 bits inside the gathered block: a bag of 2**24 + 1 rows keeps its last row
    (a float32 length would have lost it)."""
    from hipt_abmil_atec23_amd import distributed as D
    n = 2 ** 24 + 1
    a = torch.zeros(n)
    a[-1] = 7.0
    all_logits, all_a = D.gather_slide_outputs([1, 0], [torch.tensor([1.0, 2.0]), torch.tensor([3.0, 4.0])], [a, torch.arange(3.0)], 2)
    assert all_a[1].numel() == n and float(all_a[1][-1]) == 7.0 and all_a[0].tolist() == [0.0, 1.0, 2.0]
    assert all_logits.tolist() == [[3.0, 4.0], [1.0, 2.0]]


def _one_rank_worker(port, opt_in, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.pop("HIPT_SINGLE_RANK_GROUP", None)
    from hipt_abmil_atec23_amd import distributed as D
    r, w, _ = D.init_from_env(backend="gloo", single_rank_group=True if opt_in == "arg" else None) if opt_in != "env" else (None, None, None)
    if opt_in == "env":
        os.environ["HIPT_SINGLE_RANK_GROUP"] = "1"
        r, w, _ = D.init_from_env(backend="gloo")
    q.put((r, w, D.grouped(), D.group_info()))
    if D.grouped():
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("opt_in,grouped", [("no", False), ("arg", True), ("env", True)])
def test_one_rank_group_is_opt_in(opt_in, grouped):
    """a launcher-style environment with WORLD_SIZE=1 must NOT make a library user a process group (and a TCP rendezvous) unless
    asked: init_from_env(single_rank_group=True) -- what bench.py passes -- or HIPT_SINGLE_RANK_GROUP=1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(_free_port(), opt_in, q))
    p.start()
    r, w, g, info = q.get(timeout=90)
    p.join(timeout=30)
    assert p.exitcode == 0 and (r, w) == (0, 1) and g == grouped
    assert info == ({"backend": "gloo", "ranks": 1} if grouped else {})
