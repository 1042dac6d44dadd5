"""ORACLE ONLY.  Stand-in for ``ikpy.chain`` (see package docstring).

Reference call sites: ``seqikpy/leg_inverse_kinematics.py:66-69``
(``Chain.inverse_kinematics(target_position=, initial_position=)``) and
``:73`` (``Chain.forward_kinematics(q, full_kinematics=True)``).
"""
import numpy as np
import scipy.optimize


class Chain:
    # Set by oracle tooling to collect (status, nfev) per solve; None = off.
    solve_log = None

    def __init__(self, links, active_links_mask=None, name="chain", **_ignored):
        self.name = name
        self.links = list(links)
        if active_links_mask is None:
            active_links_mask = [True] * len(self.links)
        self.active_links_mask = np.asarray(active_links_mask, dtype=bool)

    def __len__(self):
        return len(self.links)

    def forward_kinematics(self, joints, full_kinematics=False):
        if len(joints) != len(self.links):
            raise ValueError("Your joints vector length is {} but you have {} links".format(
                len(joints), len(self.links)))
        frame = np.eye(4)
        frames = []
        for link, theta in zip(self.links, joints):
            frame = np.dot(frame, link.get_link_frame_matrix(theta))
            if full_kinematics:
                frames.append(frame)
        return frames if full_kinematics else frame

    def inverse_kinematics(self, target_position=None, initial_position=None, **_ignored):
        target = np.zeros(3) if target_position is None else np.asarray(target_position, float)
        if initial_position is None:
            initial_position = np.zeros(len(self.links))
        x0 = np.asarray(initial_position, dtype=float)

        def residual(x):
            return self.forward_kinematics(x)[:3, 3] - target

        lb = np.array([link.bounds[0] for link in self.links], dtype=float)
        ub = np.array([link.bounds[1] for link in self.links], dtype=float)
        res = scipy.optimize.least_squares(residual, x0, bounds=(lb, ub))
        if res.status == -1:
            raise ValueError("Inverse kinematic optimisation failed")
        if Chain.solve_log is not None:
            Chain.solve_log.append((int(res.status), int(res.nfev)))
        return res.x
