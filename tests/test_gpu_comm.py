"""The RCCL entry points of the C ABI (slf_comm_*): a communicator of one rank on this GPU, send / receive to self on
a stream of the library.  What a host without torch would bind for the device-to-device halo exchange
(include/sailfish_hip.h; reference subdomain_runner.py:1064-1139 moves halos through host memory and zmq).

Runs in a process of its own (tests/_comm_worker.py): a process has ONE user of RCCL -- either torch.distributed (the
runner's connector) or these entry points -- and the rest of the GPU suite imports torch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('mode', ['destroy', 'leak'])
def test_comm_sendrecv_to_self(mode):
    res = subprocess.run([sys.executable, os.path.join(HERE, '_comm_worker.py'), mode], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode(errors='replace')
    assert res.returncode == 0, out[-2000:]
    assert '%s data ok True' % mode in out and '%s end of script' % mode in out, out[-2000:]
    assert '%s count ok True' % mode in out, out[-2000:]       # slf_comm_count: what RCCL reports for the communicator
