"""Base classes of LB simulations (reference sailfish/lb_base.py): option surface,
field declarations, output / checkpoint scheduling hooks, body forces."""
from collections import namedtuple

import numpy as np

from sailfish_amd import hipabi

from sailfish_amd import node_type as nt
from sailfish_amd import sym, util

FieldPair = namedtuple('FieldPair', 'abstract buffer')
KernelPair = namedtuple('KernelPair', 'primary secondary')


class LBMixIn(object):
    pass


class Field(object):
    def __init__(self, name, expr=None, need_nn=False, init=0.0, gpu_array=False):
        self.name = name
        self.expr = expr
        self.init = init
        self.need_nn = need_nn
        self.gpu_array = gpu_array


class ScalarField(Field):
    pass


class VectorField(Field):
    pass


class LBSim(object):
    """Describes a type of LB simulation (reference lb_base.py:30-320)."""
    subdomain_runner = None
    nonlocality = 0

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--dt_per_lattice_time_unit', type=float, default=1.0,
                           help='physical time delta corresponding to one iteration of the simulation')
        grids = [x.__name__ for x in sym.KNOWN_GRIDS if x.dim == dim]
        group.add_argument('--grid', help='LB grid', type=str, choices=grids, default=grids[0])
        group.add_argument('--access_pattern', type=str, default='AB', choices=('AB', 'AA'),
                           help='Lattice access pattern: AB (two copies of the whole domain in memory), '
                                'AA (single domain copy in memory).')
        group.add_argument('--node_addressing', type=str, default='direct', choices=('direct', 'indirect'),
                           help='Node addressing mode: direct (dense arrays) or indirect (populations stored for the active '
                                'nodes only; single-fluid models).')
        group.add_argument('--minimize_roundoff', action='store_true', default=False,
                           help='tries to minimize round-off errors: the arrays hold f - w and the density field '
                                'rho - 1, so that O(1) and O(Ma) quantities are never added (reference sym.py:573-661); '
                                'BGK, fluid and bounce-back nodes')
        group.add_argument('--propagate_on_read', action='store_true', default=False,
                           help='(accepted for compatibility; the HIP kernels choose the streaming scheme)')
        group.add_argument('--propagate_with_shuffle', action='store_true', dest='propagate_with_shuffle',
                           default=False,
                           help='(accepted for compatibility; the tuned HIP kernels always shift in registers)')
        group.add_argument('--nouse_link_tags', action='store_false', dest='use_link_tags', default=True,
                           help='Disables link tagging for node types that support it (orientation only).')

    @classmethod
    def modify_config(cls, config):
        pass

    @classmethod
    def update_defaults(cls, defaults):
        pass

    @classmethod
    def fields(cls):
        return []

    def constants(self):
        return {}

    @property
    def grid(self):
        return max(self.grids, key=lambda grid: grid.Q)

    @property
    def dim(self):
        return self.grid.dim

    def _declared_fields(self):
        """The fields of this simulation class followed by those of the mix-ins in its MRO."""
        owners = [self] + [c for c in type(self).mro()[1:]
                           if issubclass(c, LBMixIn) and not issubclass(c, LBSim) and hasattr(c, 'fields')]
        return [field for owner in owners for field in owner.fields()]

    def init_fields(self, runner):
        """Creates the host mirrors of the macroscopic fields (pinned memory, registered for output) and exposes them
        as attributes: sim.rho, sim.v (list of components), sim.vx, sim.vy[, sim.vz]."""
        self._scalar_fields, self._vector_fields, self._fields = [], [], {}
        for field in self._declared_fields():
            if field.name in self._fields:
                raise AssertionError('Field %s defined more than once.' % field.name)
            if isinstance(field, VectorField):
                host = runner.make_vector_field(name=field.name, async_=True)
                for axis, component in zip('xyz', host):
                    setattr(self, field.name + axis, component)
                self._vector_fields.append(FieldPair(field, host))
            elif isinstance(field, ScalarField):
                host, _ = runner.make_scalar_field(name=field.name, async_=True)
                host[:] = field.init
                self._scalar_fields.append(FieldPair(field, host))
            else:
                raise AssertionError('Invalid field type %s' % type(field))
            setattr(self, field.name, host)
            self._fields[field.name] = FieldPair(field, host)

    def verify_fields(self):
        for name, field_pair in self._fields.items():
            assert getattr(self, name) is field_pair.buffer, \
                'Field {0} redefined (probably in initial_conditions())'.format(name)

    def __init__(self, config):
        self.config = config
        self.iteration = 0
        self.need_sync_flag = False
        self.need_fields_flag = False
        self.force_objects = []
        if config is not None:
            grid = util.get_grid_from_config(config)
            if grid is None:
                raise util.GridError('Invalid grid selected: {0}'.format(config.grid))
            self.grids = [grid]

    def get_state(self):
        return {'iteration': self.iteration}

    def set_state(self, state):
        self.iteration = state['iteration']

    def need_output(self):
        if self.config.output_required:
            return ((self.iteration + 1) % self.config.every) == 0 and self.config.from_ <= self.iteration
        return False

    def need_sync_fields(self):
        """(sync to host requested, macroscopic fields requested) -- reference lb_base.py:233-252."""
        sync = bool(self.need_sync_flag or self.need_output())
        fields = bool(sync or self.need_fields_flag)
        self.need_sync_flag = self.need_fields_flag = False      # one-shot requests
        return sync, fields

    def need_checkpoint(self):
        return (self.config.checkpoint_every > 0 and (self.iteration % self.config.checkpoint_every) == 0 and
                self.iteration >= self.config.checkpoint_from)

    def before_main_loop(self, runner):
        pass

    def after_step(self, runner):
        pass

    def after_main_loop(self, runner):
        pass

    def get_compute_kernels(self, runner, full_output, bulk):
        return KernelPair(None, None)

    def get_pbc_kernels(self, runner):
        return []

    def get_aux_kernels(self, runner):
        return KernelPair([], [])

    def initial_conditions(self, runner):
        pass

    def fill_module_desc(self, kw):
        """Contributes to the kernel-module descriptor (the counterpart of the reference's
        update_context(), lb_base.py:120-137)."""
        kw['relaxation_enabled'] = int(bool(getattr(self.config, 'relaxation_enabled', True)))


class LBForcedSim(LBSim):
    """Body forces (reference lb_base.py:300-395).  Mix-in: inherit from another LBSim first."""

    def __init__(self, config):
        super(LBForcedSim, self).__init__(config)
        self._forces = {}
        self._symbolic_forces = []      # DynamicValue accelerations (lattice 0) that depend on time

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--force_implementation', type=str, choices=['guo', 'edm'], default='guo',
                           help='How body / Shan-Chen forces enter the collision: Guo forcing or the exact '
                                'difference method (BGK only)')

    def add_body_force(self, force, grid=0, accel=True):
        """Adds a constant global acceleration (accel=True) acting on the fluid."""
        dim = self.grids[0].dim
        assert len(force) == dim
        if isinstance(force, nt.DynamicValue):
            # reference lb_base.py:346-353.  An acceleration that depends on TIME is evaluated on the host before every
            # step (it is an argument of the sweep launches: SubdomainRunner._update_dynamic_params); one that depends on
            # position would need a force field per node, which the kernels do not read
            if force.space_dependent():
                self.config.space_dependence = True
                raise NotImplementedError('body forces that depend on position are not supported by the HIP backend')
            if not accel or grid != 0:
                raise NotImplementedError('time-dependent body forces: accelerations on lattice 0')
            if force.time_dependent():
                self.config.time_dependence = True
            self._symbolic_forces.append(force)
            return
        if not accel:
            raise NotImplementedError('force (rather than acceleration) fields are not supported by the HIP backend')
        if grid not in (0, 1):
            raise NotImplementedError('the HIP backend handles body forces on lattices 0 and 1')
        self._forces.setdefault(grid, {}).setdefault(accel, np.zeros(dim, np.float64))
        self._forces[grid][accel] = self._forces[grid][accel] + np.float64(force)

    def body_force_at(self, iteration):
        """Acceleration on lattice 0 at LB iteration `iteration`: the constant part + the DynamicValue parts, or None when
        the simulation has no body force at all."""
        f = self._forces.get(0, {}).get(True)
        if not self._symbolic_forces:
            return f if (f is not None and np.any(f != 0.0)) else None
        dim = self.grids[0].dim
        total = np.zeros(dim) if f is None else np.array(f, dtype=np.float64)
        dt = getattr(self.config, 'dt_per_lattice_time_unit', 1.0)
        zero = tuple(np.zeros(1) for _ in range(dim))
        for dv in self._symbolic_forces:
            total = total + dv.evaluate(zero, iteration, dt)[0]
        return total

    @property
    def time_dependent_force(self):
        return any(dv.time_dependent() for dv in self._symbolic_forces)

    def fill_module_desc(self, kw):
        super(LBForcedSim, self).fill_module_desc(kw)
        f = self.body_force_at(0)
        if f is not None and (np.any(f != 0.0) or self._symbolic_forces):
            kw['has_force'] = 1
            kw['accel'] = list(f) + [0.0] * (3 - len(f))
        impl = getattr(self.config, 'force_implementation', 'guo')
        if impl not in ('guo', 'edm'):
            raise NotImplementedError('force_implementation=%s is not supported by the HIP backend' % impl)
        kw['force_implementation'] = hipabi.SLF_FORCE_EDM if impl == 'edm' else hipabi.SLF_FORCE_GUO
        f1 = self._forces.get(1, {}).get(True)
        if f1 is not None and np.any(f1 != 0.0):
            if len(self.grids) < 2:
                raise ValueError('add_body_force(grid=1) on a model with a single lattice')
            kw['accel1'] = list(f1) + [0.0] * (3 - len(f1))
