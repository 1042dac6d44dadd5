"""CPU tier: the N > 1 path (sequence partition + final angle all-gather) with world_size 2 on
gloo.  The solver injected here is the oracle -- this test is about the sharding / collective
plumbing, which is identical on RCCL."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import PKG_PARENT, ROOT, load_golden


def free_port():
    """A rendezvous port nobody is listening on right now (fixed, pid-derived ports collided with sockets of earlier
    tests now and then)."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


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
    port = free_port()
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
@pytest.mark.parametrize("world", [2, 4, 8])   # 8: the size the driver's scaling run ends with
def test_gather_pipeline_double_buffering(tmp_path, world):
    """Rank 0 posts one receive per peer and step (grouped point-to-point), the peers one send each; buffers
    alternate; what rank 0 sees for step s is every rank's block of step s."""
    port = free_port()
    mp.spawn(_pipeline_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    seen = np.load(tmp_path / "seen.npy")
    assert np.array_equal(seen, np.array([[100.0 * s + r for r in range(world)] for s in range(5)]))


class OracleSlab:
    """Stand-in for frame_sharding.DeviceSlab on the CPU: the model of the library's chunked call (tests/chunk_model.py,
    built on the C oracle) for every chain of the slab, with the same frame_lead / chunk_resume semantics."""

    def __init__(self, pose_slab, legs, chunk, halo, tol, lead, want_fk, affine=None, device=-1):
        import torch
        from chunk_model import ChunkedChain
        from oracle import c_oracle
        self.torch = torch
        z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
        self.S, self.L, self.n = pose_slab.shape[:3]
        self.lead, self.repaired = lead, 0
        self.serial = lead == 0 and chunk >= self.n      # (what pick_frame_chunks does with a one-chunk call)
        par = [leg if isinstance(leg, tuple) else (z[f"{leg}_seg"], z[f"{leg}_bounds"], z[f"{leg}_seeds"]) for leg in legs]
        self.chains = [[ChunkedChain(c_oracle, pose_slab[s, li], *par[li], chunk, halo, tol=max(tol, 0.0), lead=lead)
                        for li in range(len(legs))] for s in range(self.S)]
        self.oracle = c_oracle

    def speculate(self):
        for row in self.chains:
            for m in row:
                if self.serial:
                    r = self.oracle.seq_leg(m.pose, *m.par)
                    m.angles[:], m.fk[:] = r["angles"], r["fk"]
                else:
                    m.speculate()
                    m.settle()

    def resume(self, left):
        for s, row in enumerate(self.chains):
            for li, m in enumerate(row):
                m.settle(init=left[s, li].numpy().copy(), resume=True)
                self.repaired += int(m.stats[3:7].sum())

    # the lockstep pieces (frame_sharding.DeviceSlab.speculate_only / inc_flags / one_round / sweep_only)
    def speculate_only(self):
        for row in self.chains:
            for m in row:
                if self.serial:
                    r = self.oracle.seq_leg(m.pose, *m.par)
                    m.angles[:], m.fk[:] = r["angles"], r["fk"]
                else:
                    m.speculate_only()

    def inc_flags(self, left=None):
        K = self.chains[0][0].K
        out = np.zeros((self.S, self.L, K), bool)
        if not self.serial:
            for s, row in enumerate(self.chains):
                for li, m in enumerate(row):
                    out[s, li] = m.inc_flags(left[s, li].numpy().copy() if left is not None else None)
        return self.torch.from_numpy(out)

    def one_round(self, left=None, left_blocked=None):
        if self.serial:
            return
        for s, row in enumerate(self.chains):
            for li, m in enumerate(row):
                self.repaired += m.one_round(left[s, li].numpy().copy() if left is not None else None,
                                             bool(left_blocked[s, li]) if left_blocked is not None else False)

    def sweep_only(self, left=None):
        if self.serial:
            return
        for s, row in enumerate(self.chains):
            for li, m in enumerate(row):
                self.repaired += m.sweep_only(left[s, li].numpy().copy() if left is not None else None)

    def _stack(self, get):
        return self.torch.from_numpy(np.stack([np.stack([get(m) for m in row]) for row in self.chains]))

    def end_state(self):
        return self._stack(lambda m: m.angles[-1])

    def angles(self):
        return self._stack(lambda m: m.angles[self.lead:])

    def fk(self):
        return self._stack(lambda m: m.fk[self.lead:])


def _frame_shard_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from seqikpy_amd import frame_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = ["RF", "LM", "RH"]
    n_frames = int(os.environ.get("SEQIK_TEST_FRAMES", "610"))
    pose = np.stack([z[f"{l}_pose"][:n_frames] for l in legs])[None]
    res = {}
    for name, tol, lockstep in (("spec", 1e-6, True), ("exact", 0.0, True), ("old", 1e-6, False)):
        st = {}
        out = frame_sharding.solve_frame_sharded(pose, legs, chunk=50, halo=8, tol=tol, stats=st, slab_factory=OracleSlab, lockstep=lockstep)
        res[name + "_angles"], res[name + "_fk"] = out["angles"], out["fk"]
        res[name + "_rounds"], res[name + "_slab"] = st["boundary_rounds"], np.array(st["slab"])
        res[name + "_resumes"] = st["resume_calls"]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_one_recording_sharded_by_frame_over_ranks(tmp_path, oracle, world):
    """610 frames x 3 legs cut into slabs of whole 50-frame chunks over 2 / 3 / 8 ranks (the model of the library's chunked
    call stands in for the GPU).  LOCKSTEP protocol (the default since round 6): every slab speculates, last frames and
    inconsistency flags are exchanged, and the ranks run the repair rounds and the sweep of ONE chunked call together.  The
    result is the ONE-rank chunked result bit for bit; with tol = 0 every run-in is refused, three rounds repair three chunks
    and the sweep walks the rest slab by slab: the serial walk, bit for bit; every rank ends up with the whole recording.
    The round-2 protocol (`lockstep=False`: every slab settles itself, then its boundary in a resume call) gives the same
    bits on this well-posed recording."""
    from chunk_model import chunked_oracle
    port = free_port()
    mp.spawn(_frame_shard_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = ["RF", "LM", "RH"]
    par = {l: (z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs}
    one = [chunked_oracle(oracle, z[f"{l}_pose"][:610], *par[l], 50, 8) for l in legs]
    serial = [oracle.seq_leg(z[f"{l}_pose"][:610], *par[l]) for l in legs]
    slabs = []
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        for li in range(len(legs)):
            assert np.array_equal(got["spec_angles"][0, li], one[li]["angles"]) and np.array_equal(got["spec_fk"][0, li], one[li]["fk"])
            assert np.array_equal(got["exact_angles"][0, li], serial[li]["angles"]) and np.array_equal(got["exact_fk"][0, li], serial[li]["fk"])
            assert np.array_equal(got["old_angles"][0, li], one[li]["angles"]) and np.array_equal(got["old_fk"][0, li], one[li]["fk"])
        assert int(got["spec_rounds"]) == 0 and int(got["spec_resumes"]) == 0      # nothing inconsistent anywhere: two small all-gathers
        assert int(got["exact_rounds"]) == 3 and int(got["exact_resumes"]) >= 3     # three lockstep rounds, then the sweep
        assert int(got["old_rounds"]) == 0 and int(got["old_resumes"]) == (1 if r > 0 else 0)
        slabs.append(tuple(got["spec_slab"]))
    assert slabs[0][0] == 0 and slabs[-1][1] == 610 and all(slabs[i][1] == slabs[i + 1][0] for i in range(world - 1))


# placements of the anipose LF recording (start frame, chunk, halo) for which a rank boundary falls next to / into the
# kinematic-singularity episode of frames 284-301 (tests/conftest.py::LF_DEGENERATE) and chunks of it are repaired after the exchange
LF_EDGE_CASES = [(108, 16, 4), (156, 16, 4), (124, 4, 4), (176, 4, 4), (132, 8, 8), (164, 8, 2)]


def _lf_edge_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from seqikpy_amd import frame_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    za = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    par = (za["LF_seg"], za["LF_bounds"], za["LF_seeds"])
    res = {}
    for start, chunk, halo in LF_EDGE_CASES:
        pose = za["LF_pose"][start:start + 320][None, None]
        for name, lockstep in (("a", True), ("o", False)):
            st = {}
            out = frame_sharding.solve_frame_sharded(pose, [par], chunk=chunk, halo=halo, tol=1e-6, stats=st, slab_factory=OracleSlab,
                                                     lockstep=lockstep)
            res[f"{name}_{start}_{chunk}_{halo}"] = out["angles"][0, 0]
            res[f"{name}r_{start}_{chunk}_{halo}"] = np.array([st["boundary_rounds"], st["resume_calls"], st["chunks_repaired_after_exchange"]])
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_frame_sharding_is_independent_of_the_number_of_ranks(tmp_path, oracle):
    """Round-5 review, item 6: is the frame-sharded result independent of the number of ranks?  Since round 6 YES, by
    construction: the ranks run the speculative pass, the repair rounds and the sweep of ONE chunked call together (LOCKSTEP:
    `SeqikOptions.chunk_resume` = 3 / 4 / 5, `FrameShardedRecording._solve_lockstep`), exchanging the slabs' last frames and
    "my last chunk is inconsistent" in between -- the same sequence of solves one GPU runs for the whole recording.  Asserted
    where it is hard: rank boundaries swept across the anipose LF kinematic-singularity episode (frames 284-301), where chunks
    fail their verification and repairs cascade, at world 2 / 3 / 8 and four chunk geometries: == the one-rank chunked call bit
    for bit, every placement.  The real loop over a gloo process group (2 and 3 ranks) == the one-process re-enactment used
    for the sweep.
    The round-2 protocol (`lockstep=False`: every slab settles itself first, its boundary afterwards) is NOT independent of N
    there -- a chunk behind a boundary is verified after the exchange instead of in round 1 and can be re-solved from a
    predecessor state that differs in the last bits: three placements in four still give identical bits, the others stay
    within 1e-7 rad of the one-rank call outside the episode and differ by more only on the episode's own frames.  Kept as an
    option (one exchange and one library call fewer when a repair IS needed) and characterised here."""
    from chunk_model import chunked_oracle, lockstep_sharded_oracle, sharded_chunked_oracle
    from conftest import LF_DEGENERATE
    za = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    par = (za["LF_seg"], za["LF_bounds"], za["LF_seeds"])
    for world in (2, 3):
        out_dir = tmp_path / f"w{world}"
        out_dir.mkdir()
        mp.spawn(_lf_edge_worker, args=(world, free_port(), str(out_dir)), nprocs=world, join=True)
        for start, chunk, halo in LF_EDGE_CASES:
            pose = za["LF_pose"][start:start + 320]
            one = chunked_oracle(oracle, pose, *par, chunk, halo)
            old = sharded_chunked_oracle(oracle, pose, *par, chunk, halo, world)
            for r in range(world):
                got = np.load(out_dir / f"rank{r}.npz")
                assert np.array_equal(got[f"a_{start}_{chunk}_{halo}"], one["angles"]), (world, start, chunk, halo, r)
                assert np.array_equal(got[f"o_{start}_{chunk}_{halo}"], old["angles"]), (world, start, chunk, halo, r)
    identical, different, rounds_seen = 0, [], set()
    for world in (2, 3, 8):
        n = 320 if world < 8 else 640
        for start, chunk, halo in LF_EDGE_CASES + [(s, c, h) for c, h in ((16, 4), (4, 4)) for s in range(100, 300, 24)]:
            pose = za["LF_pose"][start:start + n]
            one = chunked_oracle(oracle, pose, *par, chunk, halo)
            lock = lockstep_sharded_oracle(oracle, pose, *par, chunk, halo, world)
            assert np.array_equal(lock["angles"], one["angles"]) and np.array_equal(lock["fk"], one["fk"]), (world, start, chunk, halo)
            rounds_seen.add((lock["rounds"], lock["swept"] > 0))
            sh = sharded_chunked_oracle(oracle, pose, *par, chunk, halo, world)         # the round-2 protocol
            d = np.abs(sh["angles"] - one["angles"]).max(1)
            outside = np.ones(n, bool)
            outside[max(LF_DEGENERATE[0] - start, 0):max(LF_DEGENERATE[1] - start, 0)] = False
            assert d[outside].max() <= 1e-7, (world, start, chunk, halo, float(d[outside].max()))
            if d.max() == 0:
                identical += 1
            else:
                different.append((world, start, chunk, halo, float(d.max())))
                assert sh["repaired_after_exchange"] > 0, (world, start, chunk, halo)
    assert identical > 3 * len(different) > 0, (identical, different)
    assert any(r >= 2 for r, _ in rounds_seen), rounds_seen      # the sweep of placements does exercise cascades over several rounds


@pytest.mark.timeout(600)
def test_frame_sharding_with_more_ranks_than_chunks(tmp_path, oracle, monkeypatch):
    """130 frames in chunks of 50 = 3 chunks over 4 ranks: one rank owns nothing, its right neighbour takes the state of
    the nearest rank that does; every rank still ends up with the whole result == the one-rank chunked result."""
    from chunk_model import chunked_oracle
    monkeypatch.setenv("SEQIK_TEST_FRAMES", "130")
    mp.spawn(_frame_shard_worker, args=(4, free_port(), str(tmp_path)), nprocs=4, join=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = ["RF", "LM", "RH"]
    one = [chunked_oracle(oracle, z[f"{l}_pose"][:130], z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"], 50, 8) for l in legs]
    slabs = []
    for r in range(4):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert got["spec_angles"].shape == (1, 3, 130, 7)
        for li in range(3):
            assert np.array_equal(got["spec_angles"][0, li], one[li]["angles"]) and np.array_equal(got["spec_fk"][0, li], one[li]["fk"])
        slabs.append(tuple(int(v) for v in got["spec_slab"]))
    assert sorted(b - a for a, b in slabs) == [0, 30, 50, 50] and slabs[0][0] == 0


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("kind", ["rccl point-to-point", "peer writes"])
def test_gather_pipeline_on_rccl_single_rank(hiplib, kind):
    """The bench's per-step gather pipeline under a real "nccl" (= RCCL) process group, as far as one GPU allows:
    one rank (so no peer to receive from: the root's own block is the submitted buffer), high-priority
    communicator stream, solver launches on three streams, buffers handed round -- every gathered block is the
    solver's output of that step.  (This test found the stream-ordered workspace pool handing one launch's
    hand-off frames to another stream's launch; see `Workspace` in csrc/seqik_hip.hip.)"""
    import torch
    import torch.distributed as dist
    from conftest import load_golden
    from seqikpy_amd import peer_gather, sharding
    if hiplib.load().seqik_device_count() < 1:
        pytest.fail("GPU tier needs a GPU")
    z = load_golden("df3d_100")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    T = 25
    pose = np.stack([np.stack([z[f"{l}_pose"][k * T:(k + 1) * T] for l in legs]) for k in range(4)])  # (4, 6, T, 5, 3)
    want = hiplib.solve_seq(pose, params, want_fk=False)["angles"]
    opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    port = free_port()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0), pg_options=opts)
    try:
        d_pose = [torch.from_numpy(np.ascontiguousarray(pose[k:k + 1])).cuda() for k in range(4)]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(2)]
        d_ang = [torch.zeros((1, 6, T, 7), dtype=torch.float64, device="cuda") for _ in range(3)]
        if kind == "peer writes":   # no peer to write, but the flag all-reduces and their chaining run on RCCL
            pipe = peer_gather.PeerWriteGather(dist, 1, 0, d_ang[0], n_buffers=3)
            assert pipe.ok and pipe.probe()[0]
        else:
            pipe = sharding.GatherPipeline(dist, 1, 0, d_ang[0], n_buffers=3)
        got = {}
        for i in range(8):
            b, k = i % 3, i % 4
            with torch.cuda.stream(streams[b]):
                pipe.wait_buffer(b)
                if i >= 3:
                    got[i - 3] = pipe.recv[b][0].cpu().numpy().copy()  # ordered after the gather on this stream
                hiplib.solve_seq_device(d_pose[k].data_ptr(), 1, 6, T, params, d_ang[b].data_ptr(), 0,
                                        stream=streams[b].cuda_stream)
                pipe.submit(b, d_ang[b])
        pipe.drain()
        torch.cuda.synchronize()
        for i in range(5, 8):
            got[i] = pipe.recv[i % 3][0].cpu().numpy().copy()
        for i in range(8):
            assert np.array_equal(got[i], want[i % 4:i % 4 + 1]), i
        if kind == "peer writes":
            pipe.close()
    finally:
        dist.destroy_process_group()


def _peer_gather_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from seqikpy_amd import peer_gather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)   # every rank on the one GPU of the box: the mapping / copy path is the same
    n_buf = 3
    bufs = [torch.zeros(5, 7, 11, dtype=torch.float64, device="cuda") for _ in range(n_buf)]
    pipe, how = peer_gather.make_gather(dist, world, rank, bufs[0], n_buffers=n_buf, prefer="peer")
    assert isinstance(pipe, peer_gather.PeerWriteGather), how
    seen = []
    side = torch.cuda.Stream()
    for step in range(7):
        b = step % n_buf
        with torch.cuda.stream(side):          # "solve" of this step on a side stream, as the bench does
            pipe.wait_buffer(b)
            bufs[b].fill_(100.0 * step + rank)
            pipe.submit(b, bufs[b])
            if rank == 0 and step >= 1:
                pb = (step - 1) % n_buf
                pipe.wait_buffer(pb)
                seen.append([float(t.min().item()) for t in pipe.recv[pb]] + [float(t.max().item()) for t in pipe.recv[pb]])
    pipe.drain()
    if rank == 0:
        last = pipe.recv[6 % n_buf]
        seen.append([float(t.min().item()) for t in last] + [float(t.max().item()) for t in last])
        np.save(os.path.join(out_dir, "seen.npy"), np.array(seen))
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_peer_write_gather_between_processes(tmp_path):
    """Three processes on the box's GPU: rank 0 exports its receive buffers, ranks 1-2 map them and push their
    block per step with seqik_peer_copy; the 8-byte all-reduce (gloo here, RCCL on a node) is the completion flag.
    What rank 0 sees for step s is every rank's block of step s, whole (min == max == 100 s + rank)."""
    world = 3
    port = free_port()
    mp.spawn(_peer_gather_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    seen = np.load(tmp_path / "seen.npy")
    want = np.array([[100.0 * s + r for r in range(world)] * 2 for s in range(7)])
    assert np.array_equal(seen, want)


def _strong_worker(rank, world, port, out_dir):
    """bench.py --scaling strong, dry run: the fixed problem's slice per rank (sharding.rank_share), solved (oracle as
    the solver), gathered to rank 0; the pieces must tile the fixed problem exactly."""
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import c_oracle
    from seqikpy_amd import data, sharding, synthetic, utils
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    legs = data.LEGS[:2]
    body = utils.calculate_body_size(data.TEMPLATE_NMF_LOCOMOTION, legs)
    S_total, T = 7, 6
    pose = synthetic.synthetic_pose(S_total, T, legs, data.BOUNDS_LOCOMOTION, body, data.TEMPLATE_NMF_LOCOMOTION,
                                    variant="smooth", seed=synthetic.SEED_BASE)  # identical on every rank
    lo, hi, total = sharding.rank_share(S_total, world, rank, "strong")
    par = [c_oracle.leg_params(l, data.BOUNDS_LOCOMOTION, body, data.INITIAL_ANGLES_LOCOMOTION) for l in legs]

    def solve(p):
        out = np.zeros(p.shape[:3] + (7,))
        for s in range(p.shape[0]):
            for li in range(len(legs)):
                out[s, li] = c_oracle.seq_leg(p[s, li], *par[li], want_fk=False)["angles"]
        return out

    mine = torch.from_numpy(solve(pose[lo:hi]))
    counts = [sharding.rank_share(S_total, world, r, "strong") for r in range(world)]
    allr = sharding.all_gather_rows(mine, [b - a for a, b, _ in counts])
    units = torch.tensor([(hi - lo) * len(legs) * T])
    dist.all_reduce(units)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), allr.numpy())
        np.save(os.path.join(out_dir, "single.npy"), solve(pose))
        np.save(os.path.join(out_dir, "units.npy"), np.array([int(units.item()), total * len(legs) * T]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_strong_scaling_shares_tile_the_fixed_problem(tmp_path, world):
    from seqikpy_amd.sharding import rank_share
    assert [rank_share(10, 4, r, "weak") for r in range(4)] == [(0, 10, 40), (10, 20, 40), (20, 30, 40), (30, 40, 40)]
    assert [rank_share(10, 4, r, "strong")[:2] for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    with pytest.raises(ValueError):
        rank_share(10, 2, 0, "sideways")
    port = free_port()
    mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert np.array_equal(np.load(tmp_path / "gathered.npy"), np.load(tmp_path / "single.npy"))
    u = np.load(tmp_path / "units.npy")
    assert u[0] == u[1]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_strong_scaling_two_ranks_on_one_gpu(tmp_path):
    """The real bench.py under torch.distributed.run with two ranks sharing the box's GPU (gloo as the process-group
    backend): --scaling strong splits the fixed problem, the line says so."""
    import json
    import subprocess
    env = dict(os.environ, SEQIK_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4",
           "--warmup", "1", "--scaling", "strong", "--frames", "8192", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert len(lines[0]) < 4096 and b["scaling"] == "strong" and b["n_gpus"] == 2
    assert "FIXED problem" in b["config"]["workload"] and b["multi_gpu"]["ranks_seen_n"] == 2 and b["multi_gpu"]["rccl_ranks"] == 2
    assert abs(b["value"] - 8192 * 6 * b["steps"] / (b["ms_per_step"] * 1e-3 * b["steps"])) < 1e-5 * b["value"]


@pytest.mark.timeout(300)
def test_bench_launcher_propagates_rank_failure(tmp_path):
    """`python bench.py --gpus 2` without a rank environment becomes the launcher (bench.launch_ranks_if_needed): it
    starts torch.distributed.run as a child and exits with the child's code.  Here there is no GPU, so every rank
    fails: the launcher must come back non-zero and print no JSON line (never a half result)."""
    import subprocess
    if __import__("torch").cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_launches_its_own_ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SEQIK_BENCH_TIMEOUT"] = "240"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--frames", "128",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert "torch.distributed.run" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks(tmp_path):
    """The driver's command form, `python3 bench.py --gpus 2 ...` (no torchrun): the process launches its two ranks itself
    (sharing the box's one GPU, so the process group falls back to gloo), relays ONE compact JSON line (< 4 KB), rc 0.  The
    DEFAULT N > 1 run is what a first real 8-GPU execution needs and nothing else (round-5 review, item 3): provisional headline ->
    calibrated, verified headline -> n1_reference -> ranks_seen; the HEADLINE is config 3 literally, the fixed problem split over
    the ranks, gathered by grouped point-to-point transfers of the process group (RCCL on a real job)."""
    import json
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEQIK_BENCH_BACKEND", "SEQIK_GATHER")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    detail = tmp_path / "detail.json"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
                        "--frames", "8256", "--detail-path", str(detail)], env=env, capture_output=True, text=True, timeout=800)
    took = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(lines[0]) < 4096 and took < 240
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["roofline"]["frac"] > 0 and b["roofline"]["avg_launch_ms"] > 0
    assert "FIXED problem" in b["config"]["workload"] and "provisional" not in b["config"]
    assert b["config"]["gather"] == "grouped RCCL point-to-point" and b["config"]["env"]["GPU_MAX_HW_QUEUES"] == "22"
    assert abs(b["value"] - 8256 * 6 / (b["ms_per_step"] * 1e-3)) < 1e-5 * b["value"]       # the fixed problem / time
    m = b["multi_gpu"]
    assert set(m) == {"backend", "ranks_seen_n", "rccl_ranks", "devices_distinct", "efficiency_vs_n1", "speedup_vs_n1", "n1_value",
                      "n1_ms_per_step", "gather", "rank_ms_per_step_min_max", "legs"}
    assert m["ranks_seen_n"] == 2 and m["rccl_ranks"] == 2 and m["legs"] == [] and m["n1_value"] > 0
    assert abs(m["efficiency_vs_n1"] - b["value"] / (2 * m["n1_value"])) < 1e-4 * m["efficiency_vs_n1"]
    full = json.loads(detail.read_text())
    # 129 sequences: the shares are UNEVEN (65 + 64), as those of the real problem are (15 625 = 8 x 1 953 + 1); equal, padded blocks
    assert full["config"]["sequences_per_gpu"] == 65 and full["config"]["leg_frames_per_step_all_ranks"] == 8256 * 6
    fm = full["multi_gpu"]
    assert sorted(r_["rank"] for r_ in fm["ranks_seen"]) == [0, 1] and fm["rank_ms_per_step"]["min"] <= fm["rank_ms_per_step"]["max"]
    assert fm["n1_reference"]["leg_frames_per_step"] == 8256 * 6 and len(full["config"]["depth_calibration"]["candidates"]) >= 1
    assert set(fm) == {"backend", "ranks_seen", "rccl_ranks", "devices_distinct", "rank_ms_per_step", "n1_reference", "speedup_vs_n1",
                       "efficiency_vs_n1"}


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_four_ranks_on_one_gpu_rehearsal(tmp_path):
    """The N > 1 path at the largest rank count the GPU box allows beside the test process (the pool's process guard admits 6
    processes on a card; the 8-rank host logic runs in the CPU tier on gloo).  Four ranks share the GPU.  (i) the DEFAULT run:
    exactly the fields of the compact line, < 4 KB, in under a minute once the ranks are up; (ii) `--legs all`: both gathers, the
    weak leg, the truly frame-sharded recording and config 5 as named legs of the full record."""
    import json
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEQIK_BENCH_BACKEND", "SEQIK_GATHER")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["SEQIK_BENCH_CONFIG5_FRAMES"] = "256000"
    detail = tmp_path / "detail.json"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "6", "--warmup", "1", "--frames", "32064",
           "--detail-path", str(detail)]
    t0 = time.time()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    took = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(lines[0]) < 4096 and took < 120, took          # (process start-up of four ranks + torch import included)
    b = json.loads(lines[0])
    m = b["multi_gpu"]
    assert set(b) == {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                      "dtype", "data", "config", "roofline", "verified", "multi_gpu", "detail"}
    assert b["n_gpus"] == 4 and m["ranks_seen_n"] == 4 and m["rccl_ranks"] == 4 and m["legs"] == []
    assert b["scaling"] == "strong" and "FIXED problem" in b["config"]["workload"] and "4 ranks" in b["config"]["workload"]
    assert abs(b["value"] - 32064 * 6 * b["steps"] / (b["ms_per_step"] * 1e-3 * b["steps"])) < 1e-5 * b["value"]
    assert m["efficiency_vs_n1"] > 0 and m["speedup_vs_n1"] > 0 and m["n1_value"] > 0
    full = json.loads(detail.read_text())
    assert full["config"]["frames_total"] == 32064 and full["config"]["sequences_per_gpu"] == 126      # 501 = 126 + 3 x 125
    assert sorted(r_["rank"] for r_ in full["multi_gpu"]["ranks_seen"]) == [0, 1, 2, 3]
    # (ii) every leg
    r = subprocess.run(cmd + ["--legs", "all"], env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1 and len(lines[0]) < 4096, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert b["multi_gpu"]["legs"] == ["config5", "gather_compare", "one_recording", "weak"]
    m = json.loads(detail.read_text())["multi_gpu"]
    assert {"peer", "rccl", "no_gather"} <= set(m["gather_compare"]) and "peer writes" in m["gather_compare"]["peer"]["ran_as"]
    assert "error" not in m["weak"] and m["weak"]["leg_frames_per_step_all_ranks"] == 4 * 32064 * 6
    assert len(m["weak"]["rank_ms_per_step"]["by_rank"]) == 4 and m["weak"]["efficiency_vs_n1"] > 0
    assert sum(m["one_recording"]["frames_per_rank"]) == 32064 and "error" not in m["one_recording"]
    assert m["one_recording"]["check"]["max_abs_vs_serial"] < 2e-5 and m["one_recording"]["n1_reference_ms"] > 0
    assert "error" not in m["config5"] and len(m["config5"]["one_recording"]["ranks"]) == 4
    assert m["config5"]["synthetic_sequences"]["value"] > 0 and len(m["config5"]["synthetic_sequences"]["by_rank"]) == 4


def _frame_shard_gpu_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from seqikpy_amd import _lib, frame_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)   # the ranks share the box's GPU; the library calls and the exchange are the real ones
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.tile(np.stack([z[f"{l}_pose"] for l in legs]), (1, 50, 1, 1))[None]   # 50 000 frames x 6 legs
    st = {}
    out = frame_sharding.solve_frame_sharded(pose, params, stats=st)                # automatic geometry of 50 000 frames
    res = dict(angles=out["angles"], fk=out["fk"], slab=np.array(st["slab"]), geometry=np.array([st["chunk"], st["halo"]]),
               rounds=st["boundary_rounds"], resumes=st["resume_calls"], repaired=st["chunks_repaired_after_exchange"])
    # a nasty one: the anipose LF episode with a short run-in -- repairs inside the slabs and across the boundary
    za = np.load(os.path.join(ROOT, "tests", "golden", "anipose_shipped.npz"))
    pa = [_lib.leg_params_from_arrays(za[f"{l}_seg"], za[f"{l}_bounds"], za[f"{l}_seeds"]) for l in ("LF", "RF")]
    pose_a = np.stack([za[f"{l}_pose"][200:520] for l in ("LF", "RF")])[None]
    out_a = frame_sharding.solve_frame_sharded(pose_a, pa, chunk=8, halo=2, stats=st)
    res.update(a_angles=out_a["angles"], a_fk=out_a["fk"])
    # rank boundaries next to / inside the LF singularity episode, chunks of it repaired after the exchange (LF_EDGE_CASES): the
    # lockstep protocol gives the one-rank call's bits; the round-2 protocol (lockstep=False) its own model's (tests/chunk_model.py)
    for start, chunk, halo in LF_EDGE_CASES[:4]:
        st_e = {}
        out_e = frame_sharding.solve_frame_sharded(za["LF_pose"][start:start + 320][None, None], pa[:1], chunk=chunk, halo=halo, stats=st_e)
        res[f"edge_{start}_{chunk}_{halo}"], res[f"edgefk_{start}_{chunk}_{halo}"] = out_e["angles"][0, 0], out_e["fk"][0, 0]
        res[f"edger_{start}_{chunk}_{halo}"] = np.array([st_e["boundary_rounds"], st_e["resume_calls"]])
        out_o = frame_sharding.solve_frame_sharded(za["LF_pose"][start:start + 320][None, None], pa[:1], chunk=chunk, halo=halo, lockstep=False)
        res[f"old_{start}_{chunk}_{halo}"] = out_o["angles"][0, 0]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_one_recording_sharded_by_frame_on_the_gpu_equals_one_rank(tmp_path, hiplib, oracle, world):
    """Round-2 review item 4: ONE recording (df3d x 50 = 50 000 frames x 6 legs), frame-sharded over 2 / 3 ranks that
    hand their slabs to the library's frame chunks (frame_lead, chunk_states, chunk_resume), == the one-rank chunked
    call with the same geometry, bit for bit -- angles and FK, on every rank.  Round-5 review, item 6: also with a rank boundary
    across the LF singularity episode, where repairs cascade over the boundary -- the ranks run the rounds of ONE call in
    lockstep (chunk_resume = 3 / 4 / 5); the round-2 protocol (`lockstep=False`) reproduces its own model there."""
    port = free_port()
    mp.spawn(_frame_shard_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.tile(np.stack([z[f"{l}_pose"] for l in legs]), (1, 50, 1, 1))[None]
    c, h, k = hiplib.frame_chunk_plan(50_000)
    one = hiplib.solve_seq(pose, params, frame_chunk=c, frame_halo=h)
    assert one["chunk_stats"]["chunks"] == 6 * k
    za = load_golden("anipose_shipped")
    pa = [hiplib.leg_params_from_arrays(za[f"{l}_seg"], za[f"{l}_bounds"], za[f"{l}_seeds"]) for l in ("LF", "RF")]
    one_a = hiplib.solve_seq(np.stack([za[f"{l}_pose"][200:520] for l in ("LF", "RF")])[None], pa, frame_chunk=8, frame_halo=2)
    assert one_a["chunk_stats"]["inconsistent_at_first_check"] > 0
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        assert tuple(got["geometry"]) == (c, h)
        assert np.array_equal(got["angles"], one["angles"]) and np.array_equal(got["fk"], one["fk"]), r
        assert int(got["resumes"]) == 0 and int(got["rounds"]) == 0       # well-posed: nothing inconsistent after the exchange
        assert np.array_equal(got["a_angles"], one_a["angles"]) and np.array_equal(got["a_fk"], one_a["fk"]), r
    from chunk_model import sharded_chunked_oracle
    par = (za["LF_seg"], za["LF_bounds"], za["LF_seeds"])
    saw_rounds = 0
    for start, chunk, halo in LF_EDGE_CASES[:4]:
        pose_e = za["LF_pose"][start:start + 320]
        one_e = hiplib.solve_seq(pose_e[None, None], pa[:1], frame_chunk=chunk, frame_halo=halo, want_fk=True)
        old_model = sharded_chunked_oracle(oracle, pose_e, *par, chunk, halo, world)
        for r in range(world):
            got = np.load(tmp_path / f"rank{r}.npz")
            # LOCKSTEP (default): the one-rank call's bits, angles and FK, whatever the number of ranks
            assert np.array_equal(got[f"edge_{start}_{chunk}_{halo}"], one_e["angles"][0, 0]), (start, chunk, halo, r)
            assert np.array_equal(got[f"edgefk_{start}_{chunk}_{halo}"], one_e["fk"][0, 0]), (start, chunk, halo, r)
            saw_rounds = max(saw_rounds, int(got[f"edger_{start}_{chunk}_{halo}"][0]))
            # the round-2 protocol: its own model's bits (not always the one-rank call's: test_frame_sharding_is_independent_...)
            assert np.array_equal(got[f"old_{start}_{chunk}_{halo}"], old_model["angles"]), (start, chunk, halo, r)
    assert saw_rounds >= 1      # repairs did cross the exchange


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_one_recording_frame_sharded_over_two_ranks(tmp_path):
    """`python3 bench.py --gpus 2 --one-recording`: config 3 literally -- one recording, contiguous frame slabs on the
    ranks' GPUs (here both on the box's GPU), one JSON line with the check against the serial walk."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEQIK_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--frames", "40000", "--one-recording"], env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and "ONE recording" in b["config"]["workload"]
    assert sum(b["config"]["frames_per_rank"]) == 40000 and b["config"]["boundary_rounds"] == 0
    assert b["check"]["first_chunk_equals_serial_bit_for_bit"] and b["check"]["max_abs_vs_serial"] < 2e-5
    assert b["check"]["max_abs_vs_reference_first_1000_frames"] < 1e-4
    assert abs(b["value"] - 40000 * 6 / (b["ms_per_step"] * 1e-3)) < 1e-5 * b["value"] and len(lines[0]) < 4096


def _config5_case():
    """A recording of 4000 frames x 6 legs (the df3d fixture repeated) as RAW key points through a made-up camera frame,
    in 8 pinned-style planar slabs of 500 frames, + the fused-alignment constants (scripts/stream_config5.py)."""
    from seqikpy_amd import _lib
    z = np.load(os.path.join(ROOT, "tests", "golden", "df3d_1000.npz"))
    legs = [str(l) for l in z["legs"]]
    L, T, n_slabs = len(legs), 500, 8
    params = [_lib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    rng = np.random.default_rng(5)
    scales, fixed = 1.0 + 0.4 * rng.random(L), rng.normal(0.0, 2.0, (L, 3))
    base = np.stack([z[f"{l}_pose"] for l in legs])
    al = np.tile(base, (1, 4, 1, 1))                                               # (L, 4000, 5, 3)
    tcs = [base[i, 0, 0].copy() for i in range(L)]
    affs = [_lib.make_affine(fixed[i], scales[i], tcs[i]) for i in range(L)]
    raw = np.stack([(al[i] - tcs[i]) / scales[i] + fixed[i] for i in range(L)])    # (L, 4000, 5, 3)
    slabs = [np.ascontiguousarray(raw[None, :, k * T:(k + 1) * T].transpose(0, 1, 3, 2, 4)) for k in range(n_slabs)]
    return params, affs, slabs, T, L, n_slabs


def _config5_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from seqikpy_amd import stream_sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    params, affs, slabs, T, L, n_slabs = _config5_case()
    outs = [(np.zeros((1, L, 7, T)), np.zeros((1, L, T, 9, 3))) for _ in range(n_slabs)]
    st = {}
    k0, k1 = stream_sharding.stream_recording_sharded(lambda k: slabs[k], lambda k: outs[k], n_slabs, T, params, affine=affs,
                                                      want_fk=True, stats=st)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), span=np.array([k0, k1]), rounds=st["boundary_rounds"], restreams=st["restreams"],
             angles=np.stack([outs[k][0] for k in range(k0, k1)]), fk=np.stack([outs[k][1] for k in range(k0, k1)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_config5_one_recording_streamed_over_ranks_equals_single_process_stream(tmp_path, hiplib, world):
    """Round-2 review item 6: config 5 on N ranks -- rank r streams its contiguous slabs of ONE recording from host
    buffers (carried warm start inside the rank, the first slab settled against the left neighbour's true end state) ==
    the single-process carried stream over all slabs, bit for bit (angles and FK), RAW key points with fused alignment."""
    from seqikpy_amd.streaming import SeqikStream
    params, affs, slabs, T, L, n_slabs = _config5_case()
    c, h, _ = hiplib.frame_chunk_plan(T)
    ref = [(np.zeros((1, L, 7, T)), np.zeros((1, L, T, 9, 3))) for _ in range(n_slabs)]
    # the single-process stream as scripts/stream_config5.py opens it: automatic geometry, per-chain guard on the carried slabs
    with SeqikStream(params, 1, T, affine=affs, layout=hiplib.planar_layout(T), want_fk=True, carry=True, frame_chunk=-1) as st:
        for k in range(n_slabs):
            st.submit(slabs[k], ref[k][0], ref[k][1])
        st.wait()
    mp.spawn(_config5_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    seen = 0
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        k0, k1 = got["span"]
        assert int(got["rounds"]) == 0 and int(got["restreams"]) == 0
        for i, k in enumerate(range(k0, k1)):
            assert np.array_equal(got["angles"][i], ref[k][0]), (r, k)
            assert np.array_equal(got["fk"][i], ref[k][1]), (r, k)
            seen += 1
    assert seen == n_slabs


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_stream_config5_script_four_ranks_on_one_gpu_rehearsal():
    """Round-3 review, item 4a, second half: `scripts/stream_config5.py --gpus N --one-recording` launches its own ranks (as
    bench.py does), every rank streams its slabs of ONE recording from its own pinned buffers, rank 0 prints one JSON line.  Four
    ranks share the box's GPU (the pool admits six GPU processes; process group on gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stream_config5.py"), "--gpus", "4", "--one-recording",
                        "--frames", "96000", "--slab-frames", "8000", "--gpu-stats"], env=env, capture_output=True, text=True, timeout=800)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 4 and b["slabs"] == 12 and b["frames_total"] == 96000
    assert [tuple(x["slabs"]) for x in sorted(b["ranks"], key=lambda x: x["rank"])] == [(0, 3), (3, 6), (6, 9), (9, 12)]
    assert all(x["boundary_rounds"] == 0 and x["restreams"] == 0 for x in b["ranks"]) and b["value"] > 0
    assert b["alignment_statistics_pass"]["frames_per_leg"] == 96000


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_frame_sharding_under_a_one_rank_rccl_group_with_the_default_device(hiplib):
    """Advisor, round 2: `solve_frame_sharded` with its default `device=-1` under an "nccl" (= RCCL) process group used to
    build `torch.device("cuda", -1)`.  One rank, default device: == the one-call chunked result, bit for bit."""
    import torch
    import torch.distributed as dist
    from seqikpy_amd import frame_sharding
    z = load_golden("df3d_1000")
    legs = [str(l) for l in z["legs"]][:3]
    params = [hiplib.leg_params_from_arrays(z[f"{l}_seg"], z[f"{l}_bounds"], z[f"{l}_seeds"]) for l in legs]
    pose = np.stack([z[f"{l}_pose"] for l in legs])[None]
    one = hiplib.solve_seq(pose, params, frame_chunk=16, frame_halo=8)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        st = {}
        out = frame_sharding.solve_frame_sharded(pose, params, chunk=16, halo=8, stats=st)
        assert st["slab"] == (0, 1000) and st["boundary_rounds"] == 0
        assert np.array_equal(out["angles"], one["angles"]) and np.array_equal(out["fk"], one["fk"])
    finally:
        dist.destroy_process_group()


def _uneven_gather_worker(rank, world, port, out_dir):
    for p in (PKG_PARENT, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from seqikpy_amd import peer_gather, sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    S_total = 7                                                  # 7 sequences over 2 / 3 ranks: shares differ by one
    lo, hi, _ = sharding.rank_share(S_total, world, rank, "strong")
    own = torch.full((hi - lo, 2, 7, 4), float(rank + 1), dtype=torch.float64)
    refused = False
    try:
        peer_gather.make_gather(dist, world, rank, own, n_buffers=2, prefer="rccl")
    except ValueError as exc:
        refused = "differ in shape" in str(exc)
    # the remedy bench.py uses: every rank's block padded to the largest share
    s_pad = max(sharding.rank_share(S_total, world, r, "strong")[1] - sharding.rank_share(S_total, world, r, "strong")[0] for r in range(world))
    bufs = [torch.zeros((s_pad, 2, 7, 4), dtype=torch.float64) for _ in range(2)]
    pipe, how = peer_gather.make_gather(dist, world, rank, bufs[0], n_buffers=2, prefer="rccl")
    for step in range(3):
        b = step % 2
        pipe.wait_buffer(b)
        bufs[b].zero_()
        bufs[b][: hi - lo] = 10.0 * step + rank
        pipe.submit(b, bufs[b])
    pipe.drain()
    if rank == 0:
        got = [[float(t[0, 0, 0, 0]), int((t != 0).any(-1).any(-1).any(-1).sum())] for t in pipe.recv[0]]
        np.save(os.path.join(out_dir, "got.npy"), np.array(got))
    np.save(os.path.join(out_dir, f"refused{rank}.npy"), np.array([int(refused)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_gather_refuses_uneven_blocks_and_works_on_padded_ones(tmp_path, world):
    """The shares of the fixed problem differ by one sequence between ranks (15 625 = 8 x 1 953 + 1).  Both gather
    pipelines move equal blocks: blocks of different shape are refused on every rank alike (on RCCL a size mismatch of a
    point-to-point transfer hangs), and blocks padded to the largest share -- what bench.py allocates -- gather correctly."""
    port = free_port()
    mp.spawn(_uneven_gather_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert int(np.load(tmp_path / f"refused{r}.npy")[0]) == 1, r
    got = np.load(tmp_path / "got.npy")                          # buffer 0 last carried step 2
    from seqikpy_amd.sharding import rank_share
    for r in range(world):
        lo, hi, _ = rank_share(7, world, r, "strong")
        assert got[r, 0] == 20.0 + r and got[r, 1] == hi - lo, (r, got[r])


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_headline_survives_a_leg_that_does_not_finish(tmp_path):
    """Everything an N > 1 run does is a collective; should a rank ever fail where the others do not, they wait for ever.  The
    measurement that is already made must still come out: every rank arms the same deadlines (bench_support.Lifeline) and rank 0
    prints the best line so far.  (i) the limit of the legs BEHIND the verified headline set below what they need: the final
    headline, `multi_gpu.timed_out`, exit code 0; (ii) the limit of the stage in front of the headline (calibrations) set below
    what it needs: the PROVISIONAL headline measured on the plainest path before any calibration -- never checked bit for bit, so
    the run leaves with exit code 75 (round-5 advice: a stuck run must not look healthy)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SEQIK_BENCH_BACKEND", "SEQIK_GATHER")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["SEQIK_BENCH_CONFIG5_FRAMES"] = "256000"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--frames", "8256",
           "--detail-path", str(tmp_path / "d.json")]
    r = subprocess.run(cmd + ["--legs", "all"], env=dict(env, SEQIK_BENCH_LEGS_TIMEOUT="1.5"), capture_output=True, text=True, timeout=500)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["value"] > 0 and b["roofline"]["frac"] > 0 and "verified" in b
    assert "did not finish" in b["multi_gpu"]["timed_out"] and b["multi_gpu"]["ranks_seen_n"] == 2 and "provisional" not in b["config"]
    assert "a leg behind the headline did not finish in time" in r.stderr
    r = subprocess.run(cmd, env=dict(env, SEQIK_BENCH_STAGE_TIMEOUT="0.3"), capture_output=True, text=True, timeout=500)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert r.returncode == 75 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(lines[0])
    assert b["config"]["provisional"] is True and b["n_gpus"] == 2 and b["value"] > 0 and b["steps"] == 4 and "verified" not in b
    assert abs(b["value"] - 8256 * 6 / (b["ms_per_step"] * 1e-3)) < 1e-5 * b["value"]
    assert "the calibrations / the headline did not finish in time" in r.stderr
