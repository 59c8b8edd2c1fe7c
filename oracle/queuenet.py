"""The QUEUE formulation of the open-network neighbour rules (M5 leader, O1 sticky follower, M3 insertion gap, M4
arrival) on a two-route merge, checked against the all-pairs statement of oracle/opennet.py at every sub-step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

oracle/opennet.py states who a vehicle's leader is by comparing every pair of vehicles (M5).  The HIP kernel
`k_merge_queue` (flow_amd/csrc/flowsim_queue.h) never compares pairs: it keeps the vehicles of a replica in two
queues and reads the leader from the neighbouring lane.  This module restates that bookkeeping in plain Python on top
of MergeOracle and asserts, after every sub-step, that it yields the leader, headway and follower of the all-pairs
rule -- so the kernel's structure is proven on the CPU before it is debugged on the GPU.

The structure (two routes, zipper_distance = 0: flow/networks/merge.py, one lane each):
  A   the vehicles of route 0 (highway) and every vehicle beyond the merge point, in driving order, head first
      (D = its prefix with x >= merge_x; the rest, U0, is still upstream on the highway)
  U1  the vehicles of route 1 (ramp) upstream of the merge point, head first
  leader(A[k]) = A[k-1];  leader(U1[k]) = U1[k-1], leader(U1[0]) = the last vehicle of D (M5: a vehicle on the other
  branch upstream of the merge point is not a leader);  follower candidates of X (O1: vehicles whose leader is X) = its
  successor in its queue and, for the last vehicle of D, the head of U1.
Events after a move, in this order: a vehicle no longer strictly behind its queue predecessor -> both queues are
re-sorted by (x descending, equal x: lower slot first); arrivals leave from the head of A; the head of U1 joins A at
the place its position gives when it has passed the merge point; an inserted vehicle becomes the tail of its queue.
"""
import numpy as np

from .opennet import BIG, NO_LEADER_HEADWAY, MergeOracle


class QueueMergeOracle(MergeOracle):
    """MergeOracle whose neighbour snapshot is ALSO derived from the two queues; any difference raises."""

    def __init__(self, spec, dtype=np.float64):
        super().__init__(spec, dtype)
        assert self.P == 2 and float(self.zip_d) == 0.0 and not self.lc_enabled
        self.A = [[] for _ in range(self.R)]
        self.U1 = [[] for _ in range(self.R)]
        self.resorts = 0
        self.joins = 0
        self.checks = 0
        self._pending = np.zeros(self.R, dtype=bool)

    # ------------------------------------------------------------------ structure
    def _sorted(self, r, labs):
        return sorted(labs, key=lambda i: (-self.x[r, i], i))

    def _in_a(self, r, i, route):
        return route == 0 or self.x[r, i] >= self.merge_x

    def _rebuild(self, r, labs, route_of):
        self.A[r] = self._sorted(r, [i for i in labs if self._in_a(r, i, route_of[i])])
        self.U1[r] = self._sorted(r, [i for i in labs if not self._in_a(r, i, route_of[i])])

    def _ordered(self, r, q):
        x = self.x[r]
        return all(x[a] > x[b] for a, b in zip(q[:-1], q[1:]))           # strictly: a tie is left to the re-sort

    def reset(self, mask=None):
        self._pending = np.ones(self.R, dtype=bool) if mask is None else np.asarray(mask, dtype=bool).copy()
        return super().reset(mask)

    def _update_neighbours(self, active):
        foll0, foll_h0 = self.foll.copy(), self.foll_h.copy()
        has = super()._update_neighbours(active)
        T = self.dt_.type
        for r in range(self.R):
            if not active[r]:
                continue
            alive_now = [int(i) for i in np.flatnonzero(self.route[r] >= 0)]
            known = self.A[r] + self.U1[r]
            if self._pending[r]:                      # the launch after a reset builds the queues from the slots
                self._pending[r] = False
                self._rebuild(r, alive_now, {i: int(self.route[r, i]) for i in alive_now})
            else:
                # routes of the vehicles the queues hold (an arrived vehicle's route is gone: it is beyond merge_x)
                route_of = {i: (int(self.route[r, i]) if self.route[r, i] >= 0 else 0) for i in known}
                if not (self._ordered(r, self.A[r]) and self._ordered(r, self.U1[r])):
                    self._rebuild(r, known, route_of)
                    self.resorts += 1
                # M4: arrivals are the head of A
                arrived = set(int(i) for i in np.flatnonzero(self._just_arrived[r]))
                n_arr = 0
                while n_arr < len(self.A[r]) and self.x[r, self.A[r][n_arr]] >= self.end_x and self.A[r][n_arr] in arrived:
                    n_arr += 1
                assert set(self.A[r][:n_arr]) == arrived, (r, self.A[r][:n_arr], arrived)
                self.A[r] = self.A[r][n_arr:]
                # the head of U1 passed the merge point: it joins A where its position puts it
                while self.U1[r] and self.x[r, self.U1[r][0]] >= self.merge_x:
                    e = self.U1[r].pop(0)
                    k = sum(1 for a in self.A[r] if (self.x[r, a] > self.x[r, e]) or (self.x[r, a] == self.x[r, e] and a < e))
                    self.A[r].insert(k, e)
                    self.joins += 1
                # M3: a vehicle inserted in this sub-step is the tail of its queue
                for i in alive_now:
                    if i not in known:
                        (self.A[r] if self.route[r, i] == 0 else self.U1[r]).append(i)
                assert sorted(self.A[r] + self.U1[r]) == alive_now
            self._check(r, has[r], foll0[r], foll_h0[r], T)
        return has

    # ------------------------------------------------------------------ the snapshot from the structure
    def _check(self, r, has, foll0, foll_h0, T):
        x, veh_len, seq = self.x[r], self.veh_len, self.seq[r]
        A, U1 = self.A[r], self.U1[r]
        n_d = sum(1 for a in A if x[a] >= self.merge_x)
        assert all(x[a] >= self.merge_x for a in A[:n_d]) and all(x[a] < self.merge_x for a in A[n_d:])
        lead = {}
        for k, a in enumerate(A):
            lead[a] = A[k - 1] if k > 0 else -1
        for k, u in enumerate(U1):
            lead[u] = U1[k - 1] if k > 0 else (A[n_d - 1] if n_d > 0 else -1)
        succ = {a: [] for a in A + U1}                                   # O1: the vehicles whose leader is X
        for i, ld in lead.items():
            if ld >= 0:
                succ[ld].append(i)
        track = bool(self.spec.get("track_followers", True))
        for i in A + U1:
            ld = lead[i]
            assert self.lead[r, i] == ld, ("leader", r, i, self.lead[r, i], ld)
            h = (x[ld] - x[i]) - veh_len[ld] if ld >= 0 else T(NO_LEADER_HEADWAY)
            assert self.h[r, i] == h, ("headway", r, i)
            assert bool(has[i]) == (ld >= 0)
            if not track:
                continue
            no_lead = ld < 0
            start_h = T(NO_LEADER_HEADWAY) if no_lead else foll_h0[i]
            start_f = -1 if no_lead else foll0[i]
            best, bseq, bj = T(BIG), None, -1
            for c in succ[i]:
                c_h = (x[i] - x[c]) - veh_len[i]
                if (not no_lead or seq[c] > seq[i]) and (c_h < best or (c_h == best and seq[c] < bseq)):
                    best, bseq, bj = c_h, seq[c], c
            better = best < start_h and best < T(BIG)
            assert self.foll[r, i] == (bj if better else start_f), ("follower", r, i)
            assert self.foll_h[r, i] == (best if better else start_h), ("follower headway", r, i)
        self.checks += 1

    # ------------------------------------------------------------------ M3: the insertion gap from the structure
    def insertion_leader(self, r, route):
        """The vehicle an insertion on ``route`` is checked against (-1: none): the tail of the route's queue."""
        A, U1 = self.A[r], self.U1[r]
        if route == 0:
            return A[-1] if A else -1
        if U1:
            return U1[-1]
        n_d = sum(1 for a in A if self.x[r, a] >= self.merge_x)
        return A[n_d - 1] if n_d > 0 else -1
