"""Simulation controller (reference sailfish/controller.py): option surface,
domain decomposition, process / GPU mapping and the run loop.

Process model (MI355X-first, one node):
  * ``--gpus`` names ONE device (the default): this process drives every subdomain
    on it, stepped in lock-step with device-to-device halo copies (LocalGroup);
    a single subdomain runs its own loop (SubdomainRunner.main).
  * ``--gpus`` names SEVERAL devices and there are several subdomains: the
    controller starts one process per subdomain itself (sailfish_amd/launch.py;
    reference master.py:242-312), subdomain i on GPU gpus[i % len(gpus)]
    (master.py:106-117), each rank pinned to cores of its GPU's NUMA node; halos
    travel over RCCL/xGMI (connector.TorchDistConnector).  ``--debug_single_process``
    keeps everything in this process instead.
  * under ``python -m torch.distributed.run`` (WORLD_SIZE > 1) the same branch is
    entered directly: rank i owns subdomain i.  This replaces the reference's
    machine master + zmq connectors (master.py, connector.py); cluster launch
    (PBS/LSF/execnet) is out of scope.
"""
import logging
import math
import os
import pickle
import sys
import time

from sailfish_amd import config, io, subdomain_connection, util
from sailfish_amd.connector import LocalConnector, TorchDistConnector, init_distributed, make_connector  # noqa: F401
from sailfish_amd.geo import LBGeometry2D, LBGeometry3D


class GeometryError(Exception):
    pass


class LBGeometryProcessor(object):
    """Assigns ids, finds face neighbours (incl. periodic images) and local periodicity
    (reference controller.py:130-269)."""

    def __init__(self, subdomains, dim, gsize):
        self.subdomains = subdomains
        self.dim = dim
        self.gsize = gsize

    def transform(self, config):
        for i, s in enumerate(self.subdomains):
            s.id = i
        periodic = [config.periodic_x, config.periodic_y]
        if self.dim == 3:
            periodic.append(config.periodic_z)
        subdomain_connection.connect_subdomains(self.subdomains, self.gsize, periodic)
        if len(self.subdomains) > 1:
            for s in self.subdomains:
                if not s.neighbour_ids():
                    raise GeometryError('Not all subdomains are connected.')
        return self.subdomains


def lb_single_fluid():
    from sailfish_amd.lb_single import LBFluidSim
    return LBFluidSim


class LocalGroup(object):
    """Lock-step driver for runners that live in this process.  One step of the whole group is ONE program
    (stepqueue.py): the fronts of all runners (macro pass, sweep, pack), the device-to-device copies between their halo
    buffers, the backs (unpack) -- replayed from a C-ABI step plan where every subdomain sits on the same GPU, performed
    entry by entry otherwise (several GPUs in one process: peer copies) and on the steps that record timing events."""

    def __init__(self, runners):
        self.runners = runners
        self.by_id = dict((r._spec.id, r) for r in runners)
        for r in runners:
            r._group = self
        self._plans = {}
        self._plan_ok = None

    # ---------------------------------------------------------------- the group's step as a program
    def _program(self, q, it, reqs):
        rs = self.runners
        for r in rs:
            r._set_step_state(it)
        if rs[0].has_macro_exchange:
            for r, (_, fields_req, _) in zip(rs, reqs):
                r._program_macro(q, it, group=self, sync_req=fields_req)
            self._program_copies(q, it, 'macro')
            for r in rs:
                r._program_macro_back(q, it)
        for r, (_, fields_req, _) in zip(rs, reqs):
            r._program_front(q, it, fields_req, group=self)
        self._program_copies(q, it, 'dist')
        for r in rs:
            r._program_back(q, it)

    def _copy(self, q, dst_runner, src_runner, dst, src, nbytes):
        b, sh = dst_runner.backend, dst_runner._data_stream
        if src_runner.backend.gpu_id == b.gpu_id:
            q.copy(dst, src, nbytes, sh)
        else:       # another GPU of this process: a peer copy, which a step plan cannot hold
            q.call(lambda: b.copy_peer_async(dst, b.gpu_id, src, src_runner.backend.gpu_id, nbytes, sh))

    def _program_copies(self, q, it, kind):
        """Every packed send buffer into the matching receive buffer of the neighbour, on the receiver's data stream
        (kind: 'dist' populations, 'macro' fields of non-local models); event 'copied' after a runner's copies: its
        neighbours' next pack into the same send buffers waits for it."""
        par = it & 1
        pre = '' if kind == 'dist' else 'macro_'
        msgs = dict((r._spec.id, dict((m[0], m) for m in r.halo_messages(kind))) for r in self.runners)
        for r in self.runners:
            if kind == 'dist' and r._xface is not None:
                self._program_copies_xface(q, it, r)
                continue
            mine = msgs[r._spec.id]
            if not mine:
                continue
            if getattr(r, '_nnx', None) is not None and r._nnx_serial(self):
                continue        # shared planes, one calc stream: nothing to move and nothing to order
            sh = r._data_stream
            for nid, (_, _, _, recv_buf, n_recv) in sorted(mine.items()):
                src = self.by_id[nid]
                _, send_buf, n_send, _, _ = msgs[nid][r._spec.id]
                assert n_send == n_recv, 'halo size mismatch between subdomains %d and %d' % (nid, r._spec.id)
                if n_recv == 0:
                    continue
                q.wait(sh, src._pev[par][pre + 'packed'])
                if recv_buf != send_buf:        # shared planes (_share_xface_buffers): the order is all that is needed
                    self._copy(q, r, src, recv_buf, send_buf, n_recv * r.float().itemsize)
            q.record(r._pev[par][pre + 'copied'], sh)

    def _program_copies_xface(self, q, it, r):
        """1-D x decompositions (sailfish_amd/xface.py): batch i of a runner's face-buffer planes is ready with its i-th
        chunk event; the receiver copies it on its data stream and records the event its next step's chunks wait for."""
        par = it & 1
        sh = r._data_stream
        if self.single_calc_stream and r._xface.shared:
            return          # nothing to move and nothing to order: the one calc stream runs the sweeps in the order they need
        for pos in range(len(r._xchunks.order)):
            if not r._xchunks.exchanges_at(pos):
                continue
            mine = r.xface_pieces(pos)
            for nid in sorted(set(p[0] for p in mine)):
                src = self.by_id[nid]
                # pieces are listed in the order both sides post them: my k-th receive from this neighbour takes its
                # k-th send to me
                theirs = [p for p in src.xface_pieces(pos) if p[0] == r._spec.id]
                q.wait(sh, src._ev_chunk[par][pos])
                for (_, _, recv_addr, n), (_, send_addr, _, n_s) in zip([p for p in mine if p[0] == nid], theirs):
                    assert n == n_s, 'x-face piece mismatch between subdomains %d and %d' % (nid, r._spec.id)
                    if recv_addr != send_addr:          # shared face buffers (_share_xface_buffers): nothing to move
                        self._copy(q, r, src, recv_addr, send_addr, n * r.float().itemsize)
            q.record(r._ev_batch[par][pos], sh)
        q.record(r._pev[par]['copied'], sh)

    def _share_xface_buffers(self):
        """1-D x decompositions inside one process on one GPU: what subdomain A sends through a face is what its neighbour
        B receives through the opposite one, and both live in the same device memory -- so B's send buffer BECOMES A's
        receive buffer (per step parity), and the per-step device copies between them (a quarter of the GPU's time in a
        traced three-slab pipe run, profiles/r05/kernel_stats_pipe_3x_before.csv) disappear; the events that order A's
        reads after B's sweep stay.  A.recv keeps its memory and content (it may have been primed from a restored state)."""
        from sailfish_amd import xface
        if os.environ.get('SLF_XFACE_SHARE', '1') == '0':
            return
        done = 0
        for r in self.runners:
            # the binary Shan-Chen model's planes (xface.NNPlanes): the neighbour's send buffer of a link BECOMES my receive
            # buffer of the same link, per kind and step parity (a link's buffer holds its faces in the order both sides use)
            if getattr(r, '_nnx', None) is None:
                continue
            for kind, links in (('dist', r._links), ('macro', r._macro_links)):
                for nid, link in links.items():
                    nb = self.by_id.get(nid)
                    if nb is None or getattr(nb, '_nnx', None) is None or nb.backend.gpu_id != r.backend.gpu_id:
                        continue
                    nlink = (nb._links if kind == 'dist' else nb._macro_links)[r._spec.id]
                    for par in (0, 1):
                        nlink.send_bufs[par] = link.recv_bufs[par]
                        done += 1
                    nlink.send_buf = nlink.send_bufs[0]
                    nb._nnx_place(kind, r._spec.id)
                    r._nnx.shared = nb._nnx.shared = True
        for r in self.runners:
            if getattr(r, '_nnx', None) is not None and r._nnx.shared:
                r._nnx_prime()          # the send planes are final now: the densities the pass in front never rewrites
        for r in self.runners:
            if r._xface is None:
                continue
            sp = r._spec
            for face, nid in sp.connecting_subdomains():
                if face not in (sp.X_LOW, sp.X_HIGH):
                    continue
                nb = self.by_id.get(nid)
                if nb is None or nb._xface is None or nb.backend.gpu_id != r.backend.gpu_id:
                    continue
                f = xface.LOW if face == sp.X_LOW else xface.HIGH
                for par in (0, 1):
                    if r._xface.recv[par][f] and nb._xface.send[par][1 - f]:
                        nb._xface.send[par][1 - f] = r._xface.recv[par][f]
                        done += 1
                r._xface.shared = nb._xface.shared = True
                r._xface._bound = nb._xface._bound = None
        if done:
            self.runners[0].config.logger.debug('x-face buffers shared between the subdomains of this process: %d' % done)
        self._serialise_sweeps()

    single_calc_stream = False

    def _serialise_sweeps(self):
        """Every subdomain of the group on ONE device, all connected through shared x-face buffers: their sweeps go to ONE
        stream.  A sweep of a 128 .. 170-node slab fills the device by itself, so nothing is lost by running them one
        after the other -- and the order of the stream IS the order the face buffers need (B's step it + 1 after A's and
        C's step it), which otherwise takes an event recorded on one stream, waited for on a second and passed on to a
        third: 74 us between dependent sweeps, a third of the step of a pipe cut into three slabs
        (profiles/r06/trace_busy_pipe_3x.txt: 27.6 GMLUPS where the undivided pipe runs at 40).  SLF_GROUP_ONE_STREAM=0:
        one calc stream per subdomain and the events, as in round 5."""
        rs = self.runners
        if os.environ.get('SLF_GROUP_ONE_STREAM', '1') == '0' or len(rs) < 2:
            return
        def shared_planes(r):
            if getattr(r, '_nnx', None) is not None:
                return r._nnx.shared
            return r._xface is not None and r._xface.shared and len(r._xchunks.order) == 1

        if not all(shared_planes(r) for r in rs) or len(set(r.backend.gpu_id for r in rs)) != 1:
            return
        if any(r._bnd_stream is not r._calc_stream for r in rs):
            return
        for r in rs:
            r.backend.sync_stream(*r._all_streams())
        shared = rs[0]._calc_stream
        for r in rs[1:]:
            r._calc_stream = r._bnd_stream = shared
        self.single_calc_stream = True

    def step(self, reqs):
        from sailfish_amd.stepqueue import DirectQueue, NotPlannable
        rs = self.runners
        b0 = rs[0].backend
        it = rs[0]._sim.iteration
        if self._plan_ok is None:
            self._plan_ok = all(r._plan_ok for r in rs)
        timed = any(r._profile.wants_gpu_events() for r in rs)
        for r in rs:
            r._update_dynamic_params(it)
        plan = None
        if self._plan_ok and not timed:
            key = (it & 1, tuple(bool(req[1]) for req in reqs))
            plan = self._plans.get(key)
            if plan is None:
                plan = b0.make_plan()
                try:
                    self._program(plan, it, reqs)
                    self._plans[key] = plan
                except NotPlannable:
                    self._plan_ok, plan = False, None
        if plan is not None:
            for r in rs:
                r._set_step_state(it)
            plan.run(it)
        else:
            for r in rs:
                r.backend.set_iteration(it)
            self._program(DirectQueue(b0), it, reqs)
        for r in rs:
            if r._xface is not None:
                r._xface._bound = None
            r._sim.iteration += 1
            r.backend.set_iteration(r._sim.iteration)

    def run(self):
        runners = self.runners
        from sailfish_amd import placement
        if len(runners) == 1:
            # one subdomain: the runner's own loop (SubdomainRunner.main: HIP graphs for stretches without host
            # interaction, every other step replayed from its step plan)
            return runners[0].run()
        with placement.holding(len(runners)):   # several subdomains on one GPU: every one gets its own stretch of HBM
            for r in runners:
                r.prepare()
        self._share_xface_buffers()
        cfg = runners[0].config
        t_prev, it_prev = time.time(), runners[0]._sim.iteration
        t0, it0 = t_prev, it_prev
        for r in runners:
            r._profile.record_start()
        while not any(r.need_quit() for r in runners):
            reqs = [r.pre_step() for r in runners]
            for r in runners:
                r._profile.start_step()
            self.step(reqs)
            for r, (sync_req, fields_req, output_req) in zip(runners, reqs):
                r.post_step(sync_req, output_req)
            for r in runners:
                r._profile.end_step()
            it = runners[0]._sim.iteration
            if cfg.perf_stats_every > 0 and it % cfg.perf_stats_every == 0:
                for r in runners:
                    r.backend.sync_stream(*r._all_streams())
                now = time.time()
                nodes = sum(r.num_fluid_nodes for r in runners)
                cfg.logger.info('iteration:{0}  speed:{1:.2f} MLUPS'.format(
                    it, nodes * (it - it_prev) / (now - t_prev) * 1e-6))
                t_prev, it_prev = now, it
        for r in runners:
            r.summary = r._profile.record_end()
        for r in runners:
            r.finish()
        wall = time.time() - t0
        for r in runners:
            r.timing = {'steps': r._sim.iteration - it0, 'wall': wall}


class LBSimulationController(object):
    """Controls the execution of a LB simulation (reference controller.py:272-830)."""

    def __init__(self, lb_class, lb_geo=None, default_config=None):
        self._config_parser = config.LBConfigParser()
        self._lb_class = lb_class
        if lb_geo is None:
            lb_geo = LBGeometry2D if self.dim == 2 else LBGeometry3D
        self._lb_geo = lb_geo

        group = self._config_parser.add_group('Runtime mode settings')
        group.add_argument('--mode', help='runtime mode', type=str, choices=['batch', 'benchmark'],
                           default='batch')
        group.add_argument('--every', help='save simulation results every N iterations', metavar='N', type=int,
                           default=100)
        group.add_argument('--from', dest='from_', help='save simulation results from N iterations', metavar='N',
                           type=int, default=0)
        group.add_argument('--perf_stats_every', help='how often to display performance stats', metavar='N',
                           type=int, default=1000)
        group.add_argument('--max_iters', help='number of iterations to run; use 0 to run indefinitely',
                           type=int, default=0)
        group.add_argument('--output', help='save simulation results to FILE', metavar='FILE', type=str,
                           default='')
        group.add_argument('--output_format', help='output format', type=str,
                           choices=list(io.format_name_to_cls.keys()), default='npy')
        group.add_argument('--nooutput_compress', dest='output_compress', action='store_false', default=True,
                           help='store uncompressed .npz files')
        group.add_argument('--backends', type=str, default='hip',
                           help='computational backends to use (only "hip" exists)')
        group.add_argument('--gpus', nargs='+', default=0, type=int, help='which GPUs to use')
        group.add_argument('--debug_dump_dists', action='store_true', default=False,
                           help='dump the contents of the distribution arrays to files')
        group.add_argument('--debug_single_process', action='store_true', default=False,
                           help='run every subdomain in this process even when --gpus names several devices '
                                '(one process drives all of them in lock-step)')
        group.add_argument('--debug_dump_node_type_map', action='store_true', default=False,
                           help='dump the node type map into a file')
        group.add_argument('--base_name', type=str, default='',
                           help='base file name for logging, checkpoint and data output')
        group.add_argument('--log', type=str, default='', help='name of the file to which data is to be logged')
        group.add_argument('--loglevel', type=int, default=logging.INFO, help='minimum log level for the file logger')
        group.add_argument('--nobulk_boundary_split', dest='bulk_boundary_split', action='store_false',
                           default=True, help='Disable separate handling of bulk and boundary nodes')
        group.add_argument('--nocheck_invalid_results_host', action='store_false',
                           dest='check_invalid_results_host', default=True,
                           help='do not terminate when the host-side results contain inf / nan')
        group.add_argument('--nocheck_invalid_results_gpu', action='store_false',
                           dest='check_invalid_results_gpu', default=True,
                           help='If True, will terminate the simulation when invalid values (inf, nan) are detected '
                                'in the domain during the simulation.')
        group.add_argument('--seed', type=int, default=int(time.time()), help='PRNG seed value')

        group = self._config_parser.add_group('Checkpointing')
        group.add_argument('--checkpoint_file', type=str, help='Location of the checkpoint file.', metavar='PATH',
                           default='')
        group.add_argument('--single_checkpoint', action='store_true', default=False)
        group.add_argument('--restore_from', type=str, metavar='PATH', default='',
                           help='Location of a checkpoint file from which to start the simulation.')
        group.add_argument('--norestore_time', action='store_false', dest='restore_time', default=True)
        group.add_argument('--final_checkpoint', action='store_true', default=False,
                           help='Generates a checkpoint after the simulation is completed.')
        group.add_argument('--checkpoint_every', type=int, default=0, metavar='N',
                           help='Generates a checkpoint every N steps.')
        group.add_argument('--checkpoint_from', type=int, default=0, metavar='N')

        group = self._config_parser.add_group('Benchmarking')
        group.add_argument('--benchmark_sample_from', type=int, default=1000, metavar='N')
        group.add_argument('--benchmark_minibatch', type=int, default=50)

        group = self._config_parser.add_group('Simulation-specific settings')
        for base in lb_class.mro():
            if 'add_options' in base.__dict__:
                base.add_options(group, self.dim)

        group = self._config_parser.add_group('Geometry settings')
        lb_geo.add_options(group)

        group = self._config_parser.add_group('Code generator options')
        group.add_argument('--precision', type=str, choices=['single', 'double'], default='single',
                           help='precision (single, double)')
        group.add_argument('--block_size', type=int, default=64,
                           help='(accepted for compatibility; the HIP kernels choose the workgroup shape)')
        group.add_argument('--mem_alignment', type=int, default=32,
                           help='number of nodes to which the X dimension of the lattice is padded in memory')
        group.add_argument('--save_src', type=str, default='', help='(unused: kernels are pre-built)')
        group.add_argument('--use_src', type=str, default='', help='(unused: kernels are pre-built)')

        for backend in util.get_backends():
            group = self._config_parser.add_group("'{0}' backend options".format(backend.name))
            backend.add_options(group)

        defaults = {}
        lb_class.update_defaults(defaults)
        self._config_parser.set_defaults(defaults)
        if default_config is not None:
            self._config_parser.set_defaults(default_config)
        self.runners = []

    @property
    def dim(self):
        return self._lb_class.subdomain.dim

    def _init_subdomain_envelope(self, sim, subdomains):
        """Ghost envelope of 1 node (reference controller.py:482-494; nonlocality 0 models)."""
        envelope_size = max(1, getattr(sim, 'nonlocality', 0))
        for s in subdomains:
            s.set_actual_size(envelope_size)

    def save_subdomain_config(self, subdomains):
        """<output>.subdomains: pickled list of the subdomain specs (reference controller.py:769-775); read by
        utils/merge_subdomains.py."""
        if self.config.output:
            dname = os.path.dirname(self.config.output)
            if dname and not os.path.exists(dname):
                os.makedirs(dname)
            if int(os.environ.get('RANK', '0')) == 0:
                with open(io.subdomains_filename(self.config.output), 'wb') as f:
                    pickle.dump(subdomains, f)

    def run(self, ignore_cmdline=False):
        """Parses options, builds the subdomains and runs the simulation (the runners are afterwards
        available as .runners, the parsed options as .config)."""
        args = sys.argv[1:] if not ignore_cmdline else []
        self.config = self._config_parser.parse(args)
        cfg = self.config
        self._lb_class.modify_config(cfg)
        if getattr(cfg, 'minimize_roundoff', False) and not issubclass(self._lb_class, lb_single_fluid()):
            # the reference then stores delta-populations (templates/models/lb_single_fluid.mako:113, sym.py:656-661);
            # silently running the standard formulation would give different numbers under the same flag
            raise NotImplementedError('--minimize_roundoff: single-fluid BGK simulations only')
        if cfg.base_name:
            cfg.output = cfg.output or cfg.base_name
            cfg.checkpoint_file = cfg.checkpoint_file or cfg.base_name
            cfg.log = cfg.log or (cfg.base_name + '.log')
        cfg.logger = util.setup_logger(cfg)
        if cfg.mode == 'benchmark':
            cfg.output = ''

        geo = self._lb_geo(cfg)
        subdomains = geo.subdomains()
        assert subdomains is not None, 'Make sure the subdomain list is returned in geo_class.subdomains()'
        assert len(subdomains) > 0
        sim0 = self._lb_class(cfg)
        self._init_subdomain_envelope(sim0, subdomains)
        proc = LBGeometryProcessor(subdomains, self.dim, geo.gsize)
        subdomains = proc.transform(cfg)
        periodic = [cfg.periodic_x, cfg.periodic_y] + ([cfg.periodic_z] if self.dim == 3 else [])
        self.save_subdomain_config(subdomains)

        backend_cls = None
        for b in util.get_backends(cfg.backends.split(',')):
            backend_cls = b
            break
        if backend_cls is None:
            raise RuntimeError('no usable backend among: %s' % cfg.backends)

        world = int(os.environ.get('WORLD_SIZE', '1'))
        gpus = cfg.gpus if isinstance(cfg.gpus, (list, tuple)) else [cfg.gpus]
        output_cls = io.format_name_to_cls[cfg.output_format]

        def make_runner(spec, gpu, connector, backend=None):
            sim = self._lb_class(cfg)
            if backend is None:
                backend = backend_cls(cfg, gpu)
            output = output_cls(cfg, spec.id)
            runner_cls = sim.subdomain_runner
            runner = runner_cls(sim, spec, output, backend, None)
            runner.set_topology(subdomains, geo.gsize, periodic)
            runner._connector = connector
            return runner

        t0 = time.time()
        if world == 1 and len(subdomains) > 1 and len(gpus) > 1 and not cfg.debug_single_process:
            # several GPUs named, several subdomains: one process per subdomain, started here (reference
            # master.py:242-312 _run_subprocesses, GPUs dealt out round-robin as master.py:106-117); the children take
            # the WORLD_SIZE > 1 branch below.  --debug_single_process keeps everything in this process.
            from sailfish_amd import launch
            summary = launch.run_processes(self._lb_class, self._lb_geo, cfg, len(subdomains), gpus, log=cfg.logger.info)
            self.runners = []
            if cfg.mode == 'benchmark' and summary:
                self.mlups_total, self.mlups_comp = summary['mlups_total'], summary['mlups_comp']
                self.timing_infos = None
                return tuple(summary['timing']) + (subdomains,)
            return None, None
        if world > 1:
            rank, world = init_distributed()
            if len(subdomains) != world:
                raise GeometryError('torch.distributed run: need exactly one subdomain per rank '
                                    '(%d subdomains, %d ranks)' % (len(subdomains), world))
            # (SLF_FORCE_DEVICE: every rank on that GPU, with SLF_DIST_BACKEND=gloo -- tests/test_gpu_two_ranks.py)
            local_rank = int(os.environ.get('SLF_FORCE_DEVICE', os.environ.get('LOCAL_RANK', rank)))
            # the neighbours' receive buffers mapped into this process where the ranks can map each other's memory (one
            # node; also ranks that share a device), RCCL / torch.distributed otherwise
            backend = backend_cls(cfg, local_rank)
            connector = make_connector(dict((s.id, s.id) for s in subdomains), backend, rank, world)
            runner = make_runner(subdomains[rank], local_rank, connector, backend)
            self.runners = [runner]
            runner.run()
        else:
            connector = LocalConnector()
            self.runners = [make_runner(s, gpus[i % len(gpus)], connector) for i, s in enumerate(subdomains)]
            LocalGroup(self.runners).run()
        wall = time.time() - t0
        if cfg.mode == 'benchmark':
            # reference return value in benchmark mode (controller.py:765): timing_infos, min_timings,
            # max_timings, subdomains; (None, None) otherwise
            self.timing_infos = self._benchmark_summary(world)
            return ([s[0] for s in self.timing_infos], [s[1] for s in self.timing_infos],
                    [s[2] for s in self.timing_infos], subdomains)
        return None, None

    def _benchmark_summary(self, world):
        """Per-subdomain and total MLUPS, printed the reference's way (controller.py:740-765): eff = active
        nodes / mean step wall time, comp = active nodes / mean sweep kernel time."""
        cfg = self.config
        summaries = [r.summary for r in self.runners if getattr(r, 'summary', None) is not None]
        if world > 1:
            import torch.distributed as dist
            gathered = [None] * world
            dist.all_gather_object(gathered, summaries)
            summaries = [s for part in gathered for s in part]
            if dist.get_rank() != 0:
                return summaries
        mlups_total = mlups_comp = 0.0
        for ti, min_ti, max_ti, nodes in sorted(summaries, key=lambda s: s[0].subdomain_id):
            total = nodes / ti.total * 1e-6
            comp = nodes / ti.comp * 1e-6 if ti.comp > 0 else float('nan')
            mlups_total += total
            mlups_comp += comp
            stdev = math.sqrt(max(ti.total_sq - ti.total ** 2, 0.0))
            low = nodes / max(ti.total - stdev, 1e-12) * 1e-6 - total
            high = nodes / (ti.total + stdev) * 1e-6 - total
            if not cfg.quiet:
                print('Subdomain {0}: MLUPS eff:{1:.2f} +{2:.2f} -{3:.2f}  comp:{4:.2f}'.format(
                    ti.subdomain_id, total, abs(low), abs(high), comp))
        if not cfg.quiet:
            print('Total MLUPS: eff:{0:.2f}  comp:{1:.2f}'.format(mlups_total, mlups_comp))
        self.mlups_total, self.mlups_comp = mlups_total, mlups_comp
        return summaries
