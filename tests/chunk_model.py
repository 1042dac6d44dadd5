"""TEST INFRASTRUCTURE -- model of the frame-chunk mode (``SeqikOptions.frame_chunk``, include/seqik.h) built on the
C oracle: the same sequence of solves the library enqueues (speculative pass, ``rounds`` x {verify, repair}, verify +
serial sweep), every solve being an oracle run over the frames of a chunk.  The HIP path must reproduce it bit for bit.
"""
import numpy as np


def chunked_oracle(oracle, pose, seg, bounds, seeds, chunk, halo, tol=1e-6, rounds=3, init=None):
    """One chain: pose (N, 5, 3) -> dict(angles (N, 7), fk (N, 9, 3), stats int32[8]) (stats as chunk_stats)."""
    N = pose.shape[0]
    C, h = int(chunk), int(halo)
    K = -(-N // C)
    angles, fk = np.zeros((N, 7)), np.zeros((N, 9, 3))
    ss = np.zeros((K, 7))
    stats = np.zeros(8, np.int32)
    stats[:3] = (K, C, h)

    def solve(k, repair):
        a, b = k * C, min((k + 1) * C, N)
        if repair:
            ss[k] = angles[a - 1]
            r = oracle.seq_leg(pose[a:b], seg, bounds, seeds, init=angles[a - 1].copy())
            off = 0
        else:
            t0 = max(0, a - h) if k > 0 else 0
            r = oracle.seq_leg(pose[t0:b], seg, bounds, seeds, init=init if k == 0 else None)
            off = a - t0
            if k > 0:
                ss[k] = r["angles"][off - 1]
        angles[a:b] = r["angles"][off:]
        fk[a:b] = r["fk"][off:]

    for k in range(K):
        solve(k, False)

    def inconsistent(k):
        return not np.all(np.abs(ss[k] - angles[k * C - 1]) <= tol)

    for r in range(rounds + 1):
        inc = [k for k in range(1, K) if inconsistent(k)]
        if r == 0:
            stats[7] = len(inc)
        if not inc:
            break
        if r < rounds:
            ready = [k for k in inc if not (k > 1 and (k - 1) in inc)]
            stats[3 + min(r, 2)] += len(ready)
            for k in ready:
                solve(k, True)
        else:
            for k in range(1, K):
                if inconsistent(k):
                    solve(k, True)
                    stats[6] += 1
    return dict(angles=angles, fk=fk, stats=stats)
