"""CPU tier: the N > 1 path (sequence partition + final angle all-gather) with world_size 2 on
gloo.  The solver injected here is the oracle -- this test is about the sharding / collective
plumbing, which is identical on RCCL."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import PKG_PARENT, ROOT, leg_arrays, load_golden


def _worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from oracle import c_oracle
    from seqikpy_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_100.npz"))
    legs = [str(l) for l in z["legs"]]
    # 5 sequences (odd on purpose: ranks get 3 and 2) of 12 frames, 6 legs
    pose = np.stack([np.stack([z[f"{l}_pose"][o:o + 12] for l in legs]) for o in (0, 10, 20, 40, 70)])

    def solve(p):
        out = np.zeros(p.shape[:3] + (7,))
        for s in range(p.shape[0]):
            for li, l in enumerate(legs):
                out[s, li] = c_oracle.seq_leg(p[s, li], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"],
                                              want_fk=False)["angles"]
        return out

    gathered = sharding.solve_sharded(pose, solve)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), gathered)
    if rank == 0:
        np.save(os.path.join(out_dir, "single.npy"), solve(pose))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_balanced_and_complete():
    from seqikpy_amd.sharding import partition
    for n in (0, 1, 5, 8, 1000003):
        for w in (1, 2, 3, 8):
            spans = [partition(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_solve_equals_single_process(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    single = np.load(tmp_path / "single.npy")
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), single)


def _pipeline_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from seqikpy_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [torch.zeros(4, 3, dtype=torch.float64) for _ in range(2)]
    pipe = sharding.GatherPipeline(dist, world, rank, bufs[0])
    seen = []
    for step in range(5):
        b = step % 2
        pipe.wait_buffer(b)
        bufs[b].fill_(100.0 * step + rank)  # "solve" of this step
        pipe.submit(b, bufs[b])
        if rank == 0 and step >= 1:
            pb = (step - 1) % 2
            pipe.wait_buffer(pb)
            seen.append([float(t[0, 0]) for t in pipe.recv[pb]])
    pipe.drain()
    if rank == 0:
        seen.append([float(t[0, 0]) for t in pipe.recv[4 % 2]])
        np.save(os.path.join(out_dir, "seen.npy"), np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gather_pipeline_double_buffering(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pipeline_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    seen = np.load(tmp_path / "seen.npy")
    assert np.array_equal(seen, np.array([[100.0 * s, 100.0 * s + 1] for s in range(5)]))
