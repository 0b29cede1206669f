"""The peer transport through the C ABI (include/sailfish_hip.h "peer transport", sailfish_amd/peer.py): progress counters
and IPC-mappable buffers, with one process against itself and with two processes that share the one GPU of the box."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Opt(object):
    pass


def _backend():
    from sailfish_amd.backend_hip import HIPBackend
    return HIPBackend(Opt(), 0)


def test_ring_of_one_signals_wait_and_the_start_up_check():
    from sailfish_amd import peer
    b = _backend()
    t = peer.PeerTransport(b, 0, 1)            # runs the self-test: dense and one-word-per-workgroup stores through the "mapping"
    s = b.make_stream()
    for _ in range(5):
        t.signal([0], peer.CH_DIST, s)
    t.wait([0], peer.CH_DIST, s, 2)
    t.wait([0], peer.CH_DIST, s, 3)
    s.synchronize()
    pr = t.progress(0, peer.CH_DIST)
    assert pr == {'sent': 5, 'awaited': 5, 'arrived': 5}
    assert not t.status()['timed_out']
    t.check()
    snap = t.snapshot()
    assert {'rank': 0, 'channel': peer.CH_DIST, 'signals_enqueued': 5, 'waits_enqueued': 5} in snap['pairs']
    t.close()


def test_a_wait_nobody_answers_gives_up_and_says_whom_it_waited_for():
    from sailfish_amd import peer
    from sailfish_amd.backend_hip import HIPFatalError
    b = _backend()
    t = peer.PeerTransport(b, 0, 1, selftest=False)
    t.set_timeout(0.5)
    s = b.make_stream()
    t.signal([0], peer.CH_MACRO, s)
    t.wait([0], peer.CH_MACRO, s, 3)          # two signals short
    s.synchronize()                            # returns: the spin is bounded
    st = t.status()
    assert st['timed_out'] and st['rank'] == 0 and st['channel'] == peer.CH_MACRO and st['expected'] == 3 and st['seen'] == 1
    with pytest.raises(HIPFatalError, match='waited for signal 3 of rank 0'):
        t.check()
    t.close()


def test_step_plan_entries_for_signals_and_waits():
    from sailfish_amd import peer
    b = _backend()
    t = peer.PeerTransport(b, 0, 1, selftest=False)
    s1, s2 = b.make_stream(), b.make_stream()
    plan = b.make_plan()
    plan.peer_signal(t, [0], peer.CH_DIST, s1)
    plan.peer_signal(t, [0], peer.CH_DIST, s1)
    plan.peer_wait(t, [0], peer.CH_DIST, s2, 2)
    assert len(plan) == 3
    for it in range(4):
        plan.run(it)
    s1.synchronize()
    s2.synchronize()
    assert t.progress(0) == {'sent': 8, 'awaited': 8, 'arrived': 8} and not t.status()['timed_out']
    with pytest.raises(Exception):
        plan.peer_wait(t, [3], peer.CH_DIST, s2)          # rank out of range: refused when the plan is built
    t.close()


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from sailfish_amd.connector import init_distributed
from sailfish_amd.backend_hip import HIPBackend
from sailfish_amd import peer
class Opt(object): pass
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
init_distributed(force=True)
b = HIPBackend(Opt(), 0)
t = peer.PeerTransport(b, rank, world)          # self-test across the two processes
g = t.group()
n = 1 << 12
mine = g.alloc(4 * n)
b.memset_buf(mine, 0, 4 * n)
b.sync()
g.publish({'buf': mine})
other = 1 - rank
target = g.lookup(other, 'buf')
assert target != mine
s = b.make_stream()
src = b.alloc_buf(like=np.full(n, 100.0 + rank, dtype=np.float32))
ok = True
for rnd in range(3):
    b.to_buf(src, np.full(n, 100.0 * (rnd + 1) + rank, dtype=np.float32))
    b.copy_buf_async(target, src, 4 * n, s)         # a device copy INTO the other process's memory
    t.signal([other], peer.CH_DIST, s)
    t.wait([other], peer.CH_DIST, s)
    s.synchronize()
    got = np.empty(n, dtype=np.float32)
    b.from_buf(mine, got)
    ok = ok and bool(np.all(got == 100.0 * (rnd + 1) + other))
    t.signal([other], peer.CH_ACK, s)               # read: the other side may overwrite
    t.wait([other], peer.CH_ACK, s)
s.synchronize()
t.check()
print('PEER_WORKER rank %%d %%s' %% (rank, 'OK' if ok else 'MISMATCH'), flush=True)
g.release()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
'''


def test_two_processes_on_one_gpu_write_into_each_other(tmp_path):
    """hipIpcGetMemHandle / hipIpcOpenMemHandle between two processes that share the device, the start-up check across
    them, and data written through the mapping ordered by the counters."""
    from tests.test_gpu_two_ranks import _free_port
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, SLF_DIST_BACKEND='gloo', SLF_FORCE_DEVICE='0', HSA_ENABLE_IPC_MODE_LEGACY='0', GPU_MAX_HW_QUEUES='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), str(script)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0 and 'PEER_WORKER rank 0 OK' in out and 'PEER_WORKER rank 1 OK' in out, out[-3000:]
