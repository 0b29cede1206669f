"""Host logic of the overlapped x-face exchange (sailfish_amd/xface.py: ChunkPlan): which planes of the face buffers
travel after which z-chunk of the sweep, and which transfer a chunk of the next step has to wait for."""
import pytest

from sailfish_amd.xface import ChunkPlan


@pytest.mark.parametrize('nz', [8, 12, 33, 64, 512])
@pytest.mark.parametrize('wrap', [True, False])
@pytest.mark.parametrize('nchunks', [1, 2, 4, 7])
def test_every_written_plane_travels_once_and_nothing_is_read_too_early(nz, wrap, nchunks):
    p = ChunkPlan(nz, wrap, nchunks)
    assert sorted(p.order) == list(range(len(p.chunks)))
    assert p.chunks[0][0] == 1 and p.chunks[-1][1] == nz + 1
    assert all(a[1] == b[0] for a, b in zip(p.chunks, p.chunks[1:]))
    for kind in ('push', 'own'):
        sent = {}
        for pos, runs in enumerate(p.batches[kind]):
            for p0, p1 in runs:
                assert p0 < p1
                for pl in range(p0, p1):
                    assert pl not in sent
                    sent[pl] = pos
        written = set()
        for c in range(len(p.chunks)):
            written.update(p.writes(kind, c))
        assert set(sent) == written
        # a plane travels as soon as its last writer has been swept
        done = {}
        for pos, c in enumerate(p.order):
            for pl in p.writes(kind, c):
                done[pl] = pos
        assert sent == done
        # the next step: chunk c may start once batch need[c] has arrived -- everything it reads is in it or earlier
        for c in range(len(p.chunks)):
            for pl in p.reads_after(kind, c):
                if pl in sent:
                    assert sent[pl] <= p.need[kind][c]
        assert set(range(1, nz + 1)) <= written


def test_the_first_chunk_of_a_step_never_waits_for_the_last_transfer_of_the_previous_one():
    """That is the point of the sweep order 0, K-1, 1, .. K-2 on a ring of chunks: the last batch travels while the
    next step's first chunk computes."""
    for nz in (64, 512):
        for wrap in (True, False):
            p = ChunkPlan(nz, wrap, 4)
            last = len(p.order) - 1
            for kind in ('push', 'own'):
                assert p.need[kind][p.order[0]] < last, (nz, wrap, kind, p.need[kind])


def test_thin_subdomains_are_not_cut():
    p = ChunkPlan(6, True, 4)
    assert p.chunks == [(1, 7)] and p.order == [0] and p.need['push'] == [0]


def test_two_transfers_per_step_on_request(monkeypatch):
    """SLF_XFACE_BATCHES=two: what is complete before the last chunk starts travels in one piece, the rest after it; the
    first chunk of the next step still does not wait for the last transfer."""
    monkeypatch.setenv('SLF_XFACE_BATCHES', 'two')
    for wrap in (True, False):
        p = ChunkPlan(512, wrap, 4)
        assert [p.exchanges_at(pos) for pos in range(4)] == [False, False, True, True]
        for kind in ('push', 'own'):
            sent = dict((pl, pos) for pos, runs in enumerate(p.batches[kind]) for p0, p1 in runs for pl in range(p0, p1))
            done = {}
            for pos, c in enumerate(p.order):
                for pl in p.writes(kind, c):
                    done[pl] = pos
            assert all(sent[pl] == max(done[pl], 2) for pl in done)
            assert p.need[kind][p.order[0]] == 2 and all(n == 3 for c, n in enumerate(p.need[kind]) if c != p.order[0] and
                                                         (wrap or c != 1))


@pytest.mark.parametrize('nz', [8, 12, 33, 64, 512])
@pytest.mark.parametrize('wrap', [True, False])
@pytest.mark.parametrize('nchunks', [1, 2, 4, 7])
@pytest.mark.parametrize('kinds', [('push', 'push'), ('own', 'push'), ('push', 'own')])
def test_uncopied_face_buffers_are_neither_read_too_early_nor_overwritten_too_early(nz, wrap, nchunks, kinds):
    """Peer transport / shared buffers: my send planes ARE the neighbour's receive planes, two sets alternating by step
    parity.  Chunk c of a step of kind `kind` starts when the neighbour's previous step (kind `prev`) has completed the
    chunks up to position peer_need[c]; by then (a) every plane c reads has been written by all its writers of that
    step, and (b) every plane c writes -- into the set the neighbour's previous step READ -- has been read by all its
    readers of that step."""
    kind, prev = kinds
    p = ChunkPlan(nz, wrap, nchunks)
    need = p.peer_need(kind, prev)
    pos_of = dict((c, pos) for pos, c in enumerate(p.order))
    for c in range(len(p.chunks)):
        done = set(c2 for c2 in range(len(p.chunks)) if pos_of[c2] <= need[c])       # neighbour's chunks of the previous step
        reads = set(p.reads_after(prev, c))
        for c2 in range(len(p.chunks)):
            if reads & set(p.writes(prev, c2)):
                assert c2 in done, 'chunk %d would read planes chunk %d of the previous step is still writing' % (c, c2)
        # the previous step read what the step before it (of THIS step's kind) had written into the same set
        writes = set(p.writes(kind, c))
        for c2 in range(len(p.chunks)):
            if writes & set(p.reads_after(kind, c2)):
                assert c2 in done, 'chunk %d would overwrite planes chunk %d of the previous step is still reading' % (c, c2)
        assert need[c] >= p.need[prev][c]


@pytest.mark.parametrize('nz,wrap,nchunks', [(512, True, 4), (512, True, 8), (64, False, 4), (33, True, 7), (12, False, 1)])
def test_signals_and_waits_of_the_peer_schedule_balance(nz, wrap, nchunks):
    """A step sends a signal after every position some chunk of the neighbours' NEXT step waits for, and its own chunks
    wait -- each in front of the chunk that needs it -- for exactly the signals of the step before: the counts add up,
    they never ask for a position beyond the one the chunk needs, and never for less."""
    p = ChunkPlan(nz, wrap, nchunks)
    for kind, prev in (('push', 'push'), ('own', 'push'), ('push', 'own')):
        signalled = p.peer_signals(prev, kind)
        need = p.peer_need(kind, prev)
        assert set(signalled) == set(need)
        counts = p.peer_counts(kind, prev)
        assert sum(n for _, n in counts) == len(signalled)
        consumed = 0
        by_pos = dict(counts)
        for pos, c in enumerate(p.order):
            consumed += by_pos.get(pos, 0)
            assert signalled[consumed - 1] == max(need[c2] for c2 in p.order[:pos + 1])     # exactly up to what is needed so far
