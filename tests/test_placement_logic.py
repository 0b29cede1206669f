"""Host logic of sailfish_amd/placement.py without a GPU: a fake backend records the virtual-memory calls.  Chunks and
spacers alternate, the spacers (and only they) are released, the span respects the free memory, host copies are cut
at chunk boundaries, and a failure in the middle gives everything back."""
import pytest

from sailfish_amd import placement


class FakeVmm(object):
    GRAN = 2 << 20

    def __init__(self, total=288 << 30, fail_after=None):
        self.total_memory, self.used, self.log = total, 0, []
        self.live, self.mapped, self.next = {}, {}, 1
        self.fail_after = fail_after

    def free_memory(self):
        return self.total_memory - self.used - sum(self.live.values())

    def vmm_granularity(self):
        return self.GRAN

    def vmm_reserve(self, n):
        assert n % self.GRAN == 0
        self.log.append(('reserve', n))
        return 0x7000000000 + 0x100000000 * len([e for e in self.log if e[0] == 'reserve'])

    def vmm_release_range(self, va, n):
        self.log.append(('free_range', n))

    def vmm_chunk_create(self, n):
        if self.fail_after is not None and len(self.live) >= self.fail_after:
            from sailfish_amd.backend_hip import HIPFatalError
            raise HIPFatalError('out of memory (fake)')
        h, self.next = self.next, self.next + 1
        self.live[h] = n
        self.log.append(('create', h, n))
        return h

    def vmm_chunk_release(self, h):
        self.log.append(('release', h))
        del self.live[h]

    def vmm_map(self, va, n, h):
        assert self.live[h] == n and va % self.GRAN == 0
        self.mapped[va] = h

    def vmm_unmap(self, va, n):
        del self.mapped[va]


def test_chunks_and_spacers_alternate_and_only_spacers_are_released():
    b = FakeVmm()
    nbytes = 19 * 544 * 514 * 514 * 4
    bufs = [placement.PlacedBuffer(b, nbytes, align_offset=124) for _ in range(2)]
    assert all(pb.addr == pb.va + 124 and pb.total >= nbytes + 256 and pb.part_bytes % b.GRAN == 0 for pb in bufs)
    info = placement.place(b, bufs)
    creates = [e for e in b.log if e[0] == 'create']
    sizes = [e[2] for e in creates]
    part, spacer = bufs[0].part_bytes, int(info['spacer_gib'] * 2 ** 30 + 0.5)
    # per round: part of A, part of B, spacer (none after the last round)
    assert sizes[:3] == [part, part, sizes[2]] and sizes[2] != part and len(creates) == 16 * 2 + 15
    assert abs(sizes[2] - spacer) < (1 << 20)
    released = [e[1] for e in b.log if e[0] == 'release']
    assert sorted(b.live[h] for h in b.live) == [part] * 32 and len(released) == 15
    assert all(h is not None for pb in bufs for h in pb.mapped) and len(b.mapped) == 32
    assert 60 < info['span_gib'] <= 73
    for pb in bufs:
        pb.release()
    assert not b.live and not b.mapped


def test_span_shrinks_with_the_free_memory_and_failures_give_everything_back():
    b = FakeVmm(total=40 << 30)
    b.used = 10 << 30
    pb = placement.PlacedBuffer(b, 11 << 30)
    info = placement.place(b, [pb])
    assert info['span_gib'] <= 11 + 0.8 * 19 + 0.1
    pb.release()
    b2 = FakeVmm(fail_after=5)
    pb2 = placement.PlacedBuffer(b2, 4 << 30)
    with pytest.raises(Exception):
        placement.place(b2, [pb2])
    pb2.release()
    assert not b2.live and not b2.mapped       # spacers released by place(), chunks by release()


def test_host_copies_are_cut_at_chunk_boundaries():
    from sailfish_amd.backend_hip import HIPBackend
    hb = HIPBackend.__new__(HIPBackend)          # no device: only the segment arithmetic
    fake = FakeVmm()
    pb = placement.PlacedBuffer(fake, 100 << 20, align_offset=124, parts=4)
    hb._placed = {pb.addr: pb}
    segs = hb._segments(pb.addr, 100 << 20)
    assert segs[0][0] == pb.addr and sum(n for _, _, n in segs) == 100 << 20
    assert [off for _, off, _ in segs] == [0] + [k * pb.part_bytes - 124 for k in range(1, len(segs))]
    for a, _, n in segs:
        assert (a - pb.va) // pb.part_bytes == (a + n - 1 - pb.va) // pb.part_bytes      # inside one chunk
    assert hb._segments(12345, 10) == [(12345, 0, 10)]


def test_spacers_of_a_group_are_held_until_every_member_is_placed():
    """Two simulations set up one after the other in one process (subdomain runners of a same-process group): inside
    placement.holding() the first one's spacers are still allocated while the second one is placed (its chunks cannot
    fall into the holes), and all of them are released when the block ends."""
    b1, b2 = FakeVmm(), FakeVmm()
    with placement.holding():
        p1 = placement.PlacedBuffer(b1, 3 << 30)
        placement.place(b1, [p1])
        assert len(b1.live) == 16 + 15 and not [e for e in b1.log if e[0] == 'release']
        p2 = placement.PlacedBuffer(b2, 3 << 30)
        placement.place(b2, [p2])
        assert len(b2.live) == 16 + 15
    assert len(b1.live) == 16 and len(b2.live) == 16          # chunks stay, spacers are gone
    p3 = placement.PlacedBuffer(b1, 3 << 30)
    placement.place(b1, [p3])                                   # outside: released on the spot
    assert len(b1.live) == 32


def test_spacers_of_eight_subdomains_in_one_process_fit_the_device():
    """BASELINE config 4 the way the reference runs it on ONE device: --subdomains=8 in one process (controller.LocalGroup).
    The spacers of every placement are held until the last subdomain is placed; with the full span for each (8 x 70 GiB)
    the device was full and the first plain allocation afterwards failed (round 5, tools/bench_configs.py --only 3g8).
    placement.holding(count) gives every member 1 / count of 60 % of the free memory for its spacers."""
    b = FakeVmm()                                   # one device: 288 GiB
    slab = 19 * 160 * 514 * 514 * 4                 # one 128 x 512 x 512 x-slab, in place
    with placement.holding(8):
        low = b.free_memory()
        for _ in range(8):
            pb = placement.PlacedBuffer(b, slab)
            info = placement.place(b, [pb])
            assert info['spacer_gib'] > 0.5          # still spread: 15 spacers of more than half a GiB between the 16 chunks
            low = min(low, b.free_memory())
        assert low > 0.3 * (288 << 30), low          # fields, node maps and halo buffers of eight subdomains still fit
    assert len(b.live) == 8 * 16                     # chunks stay, spacers are gone
    # one simulation by itself keeps the full span
    b1 = FakeVmm()
    with placement.holding():
        info = placement.place(b1, [placement.PlacedBuffer(b1, slab)])
    assert info['span_gib'] > 60


@pytest.mark.parametrize('times,chosen', [([3.3, 3.31], 0), ([3.8, 3.3, 3.31], 1), ([3.8, 3.7, 3.3], 2), ([3.3, 3.9, 3.8], 0)])
def test_placement_by_measurement_keeps_the_best_and_gives_everything_else_back(times, chosen):
    """placement.choose(): place a second set while the first stays allocated, stop when two sets agree with the best
    within 2 % (or after 3), return the best, release the rest."""
    b = FakeVmm()
    nbytes = 19 * 288 * 258 * 258 * 4
    made, released = [], []

    def make_set():
        bufs = [placement.PlacedBuffer(b, nbytes, align_offset=124) for _ in range(2)]
        placement.place(b, bufs)
        assert all(set(o.mapped).isdisjoint(pb.mapped) for old in made for o in old for pb in bufs)
        made.append(bufs)
        return bufs
    seq = iter(times)

    def release(bufs):
        released.append(bufs)
        for pb in bufs:
            pb.release()
    best, info = placement.choose(make_set, lambda bufs: next(seq) * 1e-3, release)
    assert info['times_ms'] == times[:len(info['times_ms'])] and info['chosen'] == chosen and best is made[chosen]
    assert len(made) == len(info['times_ms']) and len(released) == len(made) - 1 and best not in released
    mapped = sorted(h for pb in best for h in pb.mapped)
    assert sorted(b.live) == mapped and len(b.mapped) == len(mapped)   # nothing else survives


@pytest.mark.parametrize('fail', ['make', 'measure', 'room'])
def test_placement_by_measurement_survives_a_second_set_that_does_not_fit(fail):
    """arrays above half of the device memory -- the second set cannot be placed (or probed).  The first,
    valid set stays and nothing leaks; a failure of the FIRST set is still an error."""
    made, released = [], []

    def make_set():
        if fail == 'make' and made:
            raise RuntimeError('out of memory')
        made.append(object())
        return made[-1]

    def measure(bufs):
        if fail == 'measure' and len(made) > 1:
            raise RuntimeError('probe failed')
        return 3.3e-3 if len(made) == 1 else 3.0e-3

    best, info = placement.choose(make_set, measure, released.append, room=(lambda: False) if fail == 'room' else None)
    assert best is made[0] and info['chosen'] == 0 and info['times_ms'] == [3.3] and 'note' in info
    assert released == ([made[1]] if fail == 'measure' else [])
    with pytest.raises(RuntimeError):
        placement.choose(lambda: (_ for _ in ()).throw(RuntimeError('oom')), measure, released.append)
