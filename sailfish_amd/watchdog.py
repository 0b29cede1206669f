"""Deadlines for multi-process runs: a run that stops making progress ends with a diagnosis instead of hanging.

The reference's master polls its subdomain processes and tears the run down when one of them dies
(master.py:268-312); sailfish_amd/launch.py does the same for the processes the controller starts.  This module is the
counterpart for ranks started by somebody else (torch.distributed.run: bench.py --gpus N) and for the failure the
reference cannot see: every process alive, none advancing -- a rendezvous that never completes, a communicator that
never initialises, an exchange whose partner is elsewhere.

Every rank names the PHASE it is in (`Watchdog.phase`) and a watcher thread holds it to the phase's deadline.  Each
rank keeps its last state in a small file of a directory all ranks of the run share (same node).  On expiry -- or when
another rank has reported a failure, has vanished, or the launcher has sent SIGTERM because a sibling died -- rank 0
prints ONE JSON line {"error": ..., "phase": ..., "ranks": [every rank's last state]} through `report` and the process
exits with a non-zero status; the other ranks write their state and exit likewise.  The watcher never touches the GPU
and never takes part in a collective: it works while the main thread is stuck inside one.
"""
import json
import os
import signal
import sys
import tempfile
import threading
import time

EXIT_STATUS = 86

DEADLINES = {           # seconds per phase; SLF_DEADLINE_<PHASE> overrides one, SLF_DEADLINE_SCALE scales all
    'start': 300.0,
    'rendezvous': 180.0,
    'transport': 180.0,
    'setup': 420.0,
    'first_exchange': 120.0,
    'warmup': 300.0,
    'timed': 300.0,
    'halo_timing': 180.0,
    'validate': 600.0,
    'report': 180.0,
    'finish': 120.0,
}


def deadline_of(phase):
    base = DEADLINES.get(phase.split(':')[0], 300.0)
    env = os.environ.get('SLF_DEADLINE_' + phase.split(':')[0].upper())
    if env:
        return float(env)
    return base * float(os.environ.get('SLF_DEADLINE_SCALE', '1'))


def run_directory(run_id=None):
    """The directory the ranks of one run share: keyed by the rendezvous address the launcher gave them and by the
    launcher's process id (the ranks of a run are children of one launcher; an earlier run on the same port is not)."""
    if run_id is None:
        run_id = '%s_%s_%s_%d' % (os.environ.get('TORCHELASTIC_RUN_ID', 'run'), os.environ.get('MASTER_ADDR', 'local'),
                                  os.environ.get('MASTER_PORT', '0'), os.getppid())
    safe = ''.join(ch if ch.isalnum() or ch in '._-' else '_' for ch in str(run_id))
    d = os.path.join(tempfile.gettempdir(), 'slf_watch_' + safe)
    os.makedirs(d, exist_ok=True)
    return d


def _alive(pid):
    """Is process `pid` still running?  A zombie (exited, not yet reaped by its launcher) is not."""
    try:
        with open('/proc/%d/stat' % int(pid)) as fh:
            return fh.read().rsplit(')', 1)[1].split()[0] not in ('Z', 'X')
    except (OSError, IndexError, ValueError):
        pass
    try:
        os.kill(int(pid), 0)
    except ProcessLookupError:
        return False
    except Exception:  # noqa: BLE001 -- no permission: it exists
        return True
    return True


class Watchdog(object):
    def __init__(self, rank, world, report, run_id=None, poll=0.25, extra=None, catch_sigterm=True):
        """report(dict): called once, on rank 0, with the diagnosis (bench.py prints it as its JSON line).
        extra(): optional, returns a dict added to this rank's state when a deadline expires (transport counters ...);
        it must not block."""
        self.rank, self.world, self.report, self.extra = int(rank), int(world), report, extra
        self.dir = run_directory(run_id)
        self.poll = poll
        self._lock = threading.Lock()
        self._phase, self._since, self._deadline, self._info = 'start', time.time(), deadline_of('start'), {}
        self._done = threading.Event()
        self._fired = False
        self._t0 = time.time()
        self._wake_r = self._wake_w = None
        self._clear_stale()
        self._write('running')
        if catch_sigterm and threading.current_thread() is threading.main_thread():
            # a launcher that has lost a rank sends SIGTERM to the others; the main thread may be inside a collective
            # that never returns, where no Python-level handler runs -- the wake-up descriptor is written by the C-level
            # handler at once and read by the watcher thread
            try:
                self._wake_r, self._wake_w = os.pipe()
                os.set_blocking(self._wake_w, False)
                os.set_blocking(self._wake_r, False)
                signal.signal(signal.SIGTERM, lambda *a: None)
                signal.set_wakeup_fd(self._wake_w, warn_on_full_buffer=False)
            except Exception:  # noqa: BLE001
                self._wake_r = self._wake_w = None
        self._thread = threading.Thread(target=self._watch, name='slf-watchdog', daemon=True)
        self._thread.start()

    # ------------------------------------------------------------------ state
    def _path(self, rank):
        return os.path.join(self.dir, 'rank%d.json' % rank)

    def _clear_stale(self):
        """A state file of an earlier run under the same rendezvous address must not be taken for a rank of this one."""
        try:
            os.remove(self._path(self.rank))
        except OSError:
            pass

    def _state(self, status, **more):
        with self._lock:
            st = {'rank': self.rank, 'pid': os.getpid(), 'status': status, 'phase': self._phase,
                  'in_phase_s': round(time.time() - self._since, 2), 'deadline_s': self._deadline,
                  'since_start_s': round(time.time() - self._t0, 2), 'started': self._t0}
            if self._info:
                st['info'] = dict(self._info)
        st.update(more)
        return st

    def _write(self, status, **more):
        st = self._state(status, **more)
        tmp = self._path(self.rank) + '.tmp%d' % os.getpid()
        try:
            with open(tmp, 'w') as fh:
                json.dump(st, fh)
            os.replace(tmp, self._path(self.rank))
        except OSError:
            pass
        return st

    def phase(self, name, deadline=None, **info):
        """The main thread enters phase `name` (deadline in seconds; default from DEADLINES)."""
        with self._lock:
            self._phase, self._since = name, time.time()
            self._deadline = float(deadline) if deadline is not None else deadline_of(name)
            self._info = info
        self._write('running')

    def note(self, **info):
        """More about the current phase (the step reached, the block being timed), kept in the state file."""
        with self._lock:
            self._info.update(info)
        self._write('running')

    def failed(self, what):
        """The main thread caught an exception it cannot recover from: tell the others before leaving."""
        self._write('failed', error=str(what)[:400])

    def close(self):
        self._done.set()
        self._write('done')
        if self._wake_w is not None:
            try:
                signal.set_wakeup_fd(-1)
            except Exception:  # noqa: BLE001
                pass

    # ------------------------------------------------------------------ watcher thread
    def _others(self):
        out = []
        for r in range(self.world):
            if r == self.rank:
                continue
            try:
                with open(self._path(r)) as fh:
                    st = json.load(fh)
            except (OSError, ValueError):
                out.append({'rank': r, 'status': 'no state file (not started, or gone before it wrote one)'})
                continue
            if st.get('started', 0) < self._t0 - 3600:
                out.append({'rank': r, 'status': 'stale state file'})
                continue
            if st.get('status') == 'running' and not _alive(st.get('pid', -1)):
                st['status'] = 'vanished (process %s no longer exists)' % st.get('pid')
            out.append(st)
        return out

    def _sigterm_seen(self):
        if self._wake_r is None:
            return False
        try:
            data = os.read(self._wake_r, 64)
        except (BlockingIOError, OSError):
            return False
        return signal.SIGTERM in data

    def _fire(self, error, status):
        with self._lock:
            first, self._fired = not self._fired, True
        if not first:
            # another thread of this process is writing the diagnosis and will end the process: do not return into code
            # that might end it first (the main thread re-raising its exception)
            while True:
                time.sleep(1.0)
        more = {}
        if self.extra is not None:
            try:
                more = {'detail': self.extra()}
            except Exception as e:  # noqa: BLE001
                more = {'detail': 'unavailable: %s' % str(e)[:100]}
        mine = self._write(status, **more)
        if self.rank == 0:
            time.sleep(min(1.0, 4 * self.poll))       # the others write their states in the same instant
            ranks = sorted([mine] + self._others(), key=lambda s: s.get('rank', 0))
            try:
                self.report({'error': error, 'phase': mine['phase'], 'in_phase_s': mine['in_phase_s'],
                             'deadline_s': mine['deadline_s'], 'ranks': ranks})
            finally:
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(EXIT_STATUS)
        sys.stderr.write('sailfish_amd watchdog: rank %d leaves (%s) in phase %s after %.1f s\n'
                         % (self.rank, error, mine['phase'], mine['in_phase_s']))
        sys.stderr.flush()
        time.sleep(2.0)          # rank 0 reads the states and prints the line first
        os._exit(EXIT_STATUS)

    def _watch(self):
        while not self._done.wait(self.poll):
            with self._lock:
                over = time.time() - self._since > self._deadline
                phase = self._phase
            if over:
                self._fire('deadline expired: rank %d has been in phase "%s" for more than %.0f s' % (self.rank, phase, self._deadline),
                           'expired')
            if self._sigterm_seen():
                self._fire('terminated by the launcher (SIGTERM): another rank has failed', 'terminated')
            for st in self._others():
                s = str(st.get('status', ''))
                if s in ('expired', 'failed', 'terminated') or s.startswith('vanished'):
                    what = st.get('error') or s
                    self._fire('rank %d: %s (phase "%s")' % (st.get('rank', -1), what, st.get('phase', '?')), 'stopped')
