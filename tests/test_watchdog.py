"""Deadlines of multi-process runs (sailfish_amd/watchdog.py; reference master.py:268-312 polls its subprocesses and tears
the run down): a rank that stops advancing, a rank that dies and a launcher that sends SIGTERM each end the run with ONE
diagnosis from rank 0 and a non-zero status everywhere -- while the main threads sit in something that never returns."""
import json
import multiprocessing as mp
import os
import signal
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _rank(rank, world, run_id, out_path, script):
    """script: [(phase, deadline, what)] -- what: 'pass' | 'hang' | 'die' | 'raise'."""
    from sailfish_amd.watchdog import Watchdog

    def report(diag):
        with open(out_path, 'w') as fh:
            json.dump(diag, fh)
    wd = Watchdog(rank, world, report, run_id=run_id, poll=0.05, extra=lambda: {'note': 'rank %d' % rank})
    for phase, deadline, what in script:
        wd.phase(phase, deadline, step=3)
        if what == 'hang':
            time.sleep(60)          # the main thread is stuck: only the watcher can end this
        elif what == 'die':
            os._exit(9)
        elif what == 'raise':
            wd.failed('ValueError: halo size mismatch')
            os._exit(1)
        time.sleep(0.2)
    wd.close()
    os._exit(0)


def _run(tmp_path, scripts, after_start=None):
    ctx = mp.get_context('spawn')
    out = str(tmp_path / 'line.json')
    run_id = 'test_%d_%s' % (os.getpid(), tmp_path.name)
    procs = [ctx.Process(target=_rank, args=(r, len(scripts), run_id, out, s)) for r, s in enumerate(scripts)]
    t0 = time.time()
    for p in procs:
        p.start()
    if after_start:
        after_start(procs)
    for p in procs:
        p.join(40)
        assert p.exitcode is not None, 'a rank is still running: the watchdog did not end it'
    diag = json.load(open(out)) if os.path.exists(out) else None
    return [p.exitcode for p in procs], diag, time.time() - t0


def test_a_run_that_advances_ends_quietly(tmp_path):
    codes, diag, _ = _run(tmp_path, [[('setup', 5, 'pass'), ('timed', 5, 'pass')]] * 2)
    assert codes == [0, 0] and diag is None


def test_deadline_of_a_stuck_rank_ends_every_rank_with_one_diagnosis(tmp_path):
    from sailfish_amd.watchdog import EXIT_STATUS
    scripts = [[('rendezvous', 5, 'pass'), ('first_exchange', 30, 'hang')],       # rank 0 waits for its neighbour ...
               [('rendezvous', 5, 'pass'), ('transport', 1.0, 'hang')]]           # ... which never leaves this phase
    codes, diag, took = _run(tmp_path, scripts)
    assert codes == [EXIT_STATUS, EXIT_STATUS] and took < 20
    assert 'rank 1' in diag['error'] and 'transport' in diag['error']
    states = dict((s['rank'], s) for s in diag['ranks'])
    assert states[1]['status'] == 'expired' and states[1]['phase'] == 'transport' and states[1]['detail'] == {'note': 'rank 1'}
    assert states[0]['phase'] == 'first_exchange' and states[0]['info'] == {'step': 3}


def test_rank_zero_reports_its_own_deadline(tmp_path):
    from sailfish_amd.watchdog import EXIT_STATUS
    codes, diag, _ = _run(tmp_path, [[('timed', 0.8, 'hang')], [('timed', 30, 'hang')]])
    assert codes == [EXIT_STATUS, EXIT_STATUS]
    assert diag['error'].startswith('deadline expired') and diag['phase'] == 'timed' and len(diag['ranks']) == 2


@pytest.mark.parametrize('how', ['die', 'raise'])
def test_a_rank_that_is_gone_is_noticed_before_any_deadline(tmp_path, how):
    from sailfish_amd.watchdog import EXIT_STATUS
    scripts = [[('setup', 5, 'pass'), ('timed', 300, 'hang')], [('setup', 5, 'pass'), ('timed', 300, how)]]
    codes, diag, took = _run(tmp_path, scripts)
    assert codes[0] == EXIT_STATUS and took < 20
    assert 'rank 1' in diag['error'] and ('vanished' in diag['error'] or 'halo size mismatch' in diag['error'])


def test_sigterm_from_the_launcher_produces_the_line_although_the_main_thread_is_stuck(tmp_path):
    from sailfish_amd.watchdog import EXIT_STATUS

    def terminate(procs):
        time.sleep(1.5)
        os.kill(procs[0].pid, signal.SIGTERM)
    codes, diag, took = _run(tmp_path, [[('timed', 300, 'hang')], [('timed', 300, 'hang')]], terminate)
    assert codes[0] == EXIT_STATUS and took < 20
    assert 'SIGTERM' in diag['error']


def test_deadlines_can_be_scaled_from_the_environment(monkeypatch):
    from sailfish_amd import watchdog
    base = watchdog.deadline_of('transport')
    monkeypatch.setenv('SLF_DEADLINE_SCALE', '0.5')
    assert watchdog.deadline_of('transport') == base / 2
    monkeypatch.setenv('SLF_DEADLINE_TRANSPORT', '7')
    assert watchdog.deadline_of('transport') == 7.0
