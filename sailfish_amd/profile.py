"""Time accounting of a subdomain runner in --mode=benchmark (reference sailfish/profile.py:11-160,
summary format controller.py:744-765).

Same event ids and the same `TimingInfo` summary as the reference, collected differently: the
reference times every step on the host, which needs a device synchronisation per step.  On an
MI355X a 256^3 sweep takes 0.4 ms, so a per-step sync would measure launch latency, not the
sweep.  Here steps are timed in *minibatches* (`--benchmark_minibatch`, the unit the reference
already uses for its variance estimate, profile.py:148-158): the streams are synchronised only at
minibatch boundaries, wall time / n gives the mean step time of the batch; HIP timing events are
recorded on every step (asynchronously) and read back at the end of the batch to split the steps
into bulk / boundary / pack / unpack time.
"""
import os
import time

from sailfish_amd import util


class TimeProfile(object):
    # GPU events.
    BULK = 0
    BOUNDARY = 1
    COLLECTION = 2
    DISTRIB = 3
    MACRO_BULK = 4
    MACRO_BOUNDARY = 5
    MACRO_COLLECTION = 6
    MACRO_DISTRIB = 7
    # CPU events.
    SEND_DISTS = 8
    RECV_DISTS = 9
    SEND_MACRO = 10
    RECV_MACRO = 11
    NET_RECV = 12
    STEP = 13
    STEP_SQ = 14

    def __init__(self, runner):
        self._runner = runner
        cfg = runner.config
        self._is_benchmark = getattr(cfg, 'mode', 'batch') == 'benchmark'
        self._minibatch = max(1, int(getattr(cfg, 'benchmark_minibatch', 50)))
        self._sample_from = int(getattr(cfg, 'benchmark_sample_from', 1000))
        n = self.STEP_SQ + 1
        self._timings = [0.0] * n
        self._min_timings = [1000.0] * n
        self._max_timings = [0.0] * n
        self._cpu_start = [0.0] * n
        self._cpu_batch = [0.0] * n
        self._events = []          # (event id, start, end) of the current batch
        self._open = {}
        self._in_batch = 0
        self._active = False
        self._all_steps = os.environ.get('SLF_PROFILE_ALL_STEPS', '0') == '1'
        self.samples = 0
        self.summary = None
        self.t_start = self.t_end = 0.0

    # ------------------------------------------------------------------ run brackets
    def record_start(self):
        self.t_start = time.time()
        if self._is_benchmark:
            max_iters = self._runner.config.max_iters
            if max_iters > 0 and self._sample_from >= max_iters:
                self._sample_from = max_iters // 2
                self._runner.config.logger.warning(
                    'benchmark: --benchmark_sample_from >= --max_iters, sampling from iteration %d', self._sample_from)

    def record_end(self):
        self.t_end = time.time()
        if not self._is_benchmark:
            return None
        self._close_batch()
        if self.samples == 0:
            return None
        mi = float(self.samples)
        sid = self._runner._spec.id
        t, lo, hi = self._timings, self._min_timings, self._max_timings

        def info(src, div, sq):
            return util.TimingInfo(comp=(src[self.BULK] + src[self.BOUNDARY]) / div, bulk=src[self.BULK] / div,
                                   bnd=src[self.BOUNDARY] / div, coll=src[self.COLLECTION] / div,
                                   net_wait=src[self.NET_RECV] / div, recv=src[self.RECV_DISTS] / div,
                                   send=src[self.SEND_DISTS] / div, total=src[self.STEP] / div, total_sq=sq,
                                   subdomain_id=sid)

        for i in range(len(lo)):
            if lo[i] == 1000.0:
                lo[i] = 0.0
        self.summary = (info(t, mi, t[self.STEP_SQ] / mi), info(lo, 1.0, 0.0), info(hi, 1.0, 0.0),
                        self._runner.num_fluid_nodes)
        return self.summary

    # ------------------------------------------------------------------ steps
    def _sync(self):
        r = self._runner
        r.backend.sync_stream(*r._all_streams())

    def start_step(self):
        it = self._runner._sim.iteration
        max_iters = self._runner.config.max_iters
        self._active = self._is_benchmark and it >= self._sample_from
        if self._active and max_iters > 0 and it + 1 >= max_iters:
            # the final step carries the end-of-run transfer of all fields to the host and their
            # verification (seconds of host work for a 512^3 subdomain): not part of the sample
            self._close_batch()
            self._active = False
        if not self._active:
            return
        if self._in_batch == 0:
            self._sync()
            self._t_batch = time.time()
            del self._events[:]
            for i in range(len(self._cpu_batch)):
                self._cpu_batch[i] = 0.0
        self._open.clear()

    def end_step(self, n=1):
        """n > 1: a graph replay of n steps."""
        if not self._active:
            return
        self._in_batch += n
        if self._in_batch >= self._minibatch:
            self._close_batch()

    def _account(self, i, total, per_step):
        self._timings[i] += total
        self._min_timings[i] = min(self._min_timings[i], per_step)
        self._max_timings[i] = max(self._max_timings[i], per_step)

    def _close_batch(self):
        n = self._in_batch
        if n == 0:
            return
        self._sync()
        dur = time.time() - self._t_batch
        per = dur / n
        self._account(self.STEP, dur, per)
        self._timings[self.STEP_SQ] += n * per * per
        sums, covered = {}, {}
        for i, ev0, ev1, steps in self._events:
            d = ev1.time_since(ev0) / 1e3
            sums[i] = sums.get(i, 0.0) + d
            covered[i] = covered.get(i, 0) + steps
            self._min_timings[i] = min(self._min_timings[i], d / steps)
            self._max_timings[i] = max(self._max_timings[i], d / steps)
        # an event id is recorded at most once per step (or once per graph replay of `steps` steps); where only some of
        # the batch's steps were timed (wants_gpu_events) their mean stands for the others
        timed = max(covered.values()) if covered else 0          # steps of this batch that carried timing events
        for i, total in sums.items():
            self._timings[i] += total * (float(n) / timed if 0 < timed < n else 1.0)
        del self._events[:]
        for i in (self.SEND_DISTS, self.RECV_DISTS, self.SEND_MACRO, self.RECV_MACRO, self.NET_RECV):
            if self._cpu_batch[i] > 0.0:
                self._account(i, self._cpu_batch[i], self._cpu_batch[i] / n)
        self.samples += n
        self._in_batch = 0

    def wants_gpu_events(self):
        """Does the coming step record HIP timing events?  A runner that replays its steps from a C-ABI plan
        (SubdomainRunner.step) takes the entry-by-entry route for such a step.  In an active benchmark batch: the first
        step of every minibatch (SLF_PROFILE_ALL_STEPS=1: every step); the split of a step into bulk / boundary /
        pack / unpack time is then scaled from the steps that were timed to the whole batch (_close_batch)."""
        if not self._active:
            return False
        return self._in_batch == 0 or self._all_steps

    # ------------------------------------------------------------------ events
    def record_gpu_start(self, event, stream):
        if not self._active:
            return None
        ev = self._runner.backend.make_event(stream, timing=True)
        self._open[event] = ev
        return ev

    def record_gpu_end(self, event, stream, need_event=False, steps=1):
        if self._active and event in self._open:
            ev = self._runner.backend.make_event(stream, timing=True)
            self._events.append((event, self._open.pop(event), ev, steps))
            return ev
        if need_event:
            return self._runner.backend.make_event(stream)
        return None

    def record_cpu_start(self, event):
        self._cpu_start[event] = time.time()

    def record_cpu_end(self, event):
        if self._active:
            self._cpu_batch[event] += time.time() - self._cpu_start[event]
