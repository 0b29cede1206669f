"""Binary-fluid models (reference sailfish/lb_binary.py): the Shan-Chen mixture."""
import os
from collections import defaultdict, namedtuple

import numpy as np

from sailfish_amd import hipabi, subdomain_runner, sym
from sailfish_amd.lb_base import LBForcedSim, LBSim, ScalarField, VectorField

MacroKernels = namedtuple('MacroKernels', 'distributions macro')

SHAN_CHEN_POTENTIALS = {'linear': 0, 'classic': 1}


class LBBinaryFluidBase(LBSim):
    """Two lattices (rho on lattice 0, the order parameter / second density phi on lattice 1), a shared
    velocity field and nearest-neighbour interactions (reference lb_binary.py:14-137)."""
    subdomain_runner = subdomain_runner.NNSubdomainRunner
    nonlocality = 1

    def __init__(self, config):
        super(LBBinaryFluidBase, self).__init__(config)
        self.grids.append(self.grid)

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--tau_phi', type=float, default=1.0, help='relaxation time for the phase field')

    def get_pbc_kernels(self, runner):
        """(distributions, macro) : copy -> axis -> kernels (reference lb_binary.py:40-105)."""
        dist_kernels = defaultdict(lambda: defaultdict(list))
        macro_kernels = defaultdict(lambda: defaultdict(list))
        if runner.indirect:      # periodic axes are wrapped inside the kernels; no ghost-layer kernels exist
            return MacroKernels(macro=macro_kernels, distributions=dist_kernels)
        d1a, d1b = runner.gpu_dist(0, 0), runner.gpu_dist(0, 1)
        d2a, d2b = runner.gpu_dist(1, 0), runner.gpu_dist(1, 1)
        nn_fields = [fp.buffer for fp in self._scalar_fields if fp.abstract.need_nn]
        for i in range(0, self.dim):
            dist_kernels[0][i] = [runner.get_kernel('ApplyPeriodicBoundaryConditions', [d1a, np.uint32(i)], 'Pi'),
                                  runner.get_kernel('ApplyPeriodicBoundaryConditions', [d2a, np.uint32(i)], 'Pi')]
        if self.config.access_pattern == 'AB':
            g1, g2, kernel = d1b, d2b, 'ApplyPeriodicBoundaryConditions'
        else:
            g1, g2, kernel = d1a, d2a, 'ApplyPeriodicBoundaryConditionsWithSwap'
        for i in range(0, self.dim):
            dist_kernels[1][i] = [runner.get_kernel(kernel, [g1, np.uint32(i)], 'Pi'),
                                  runner.get_kernel(kernel, [g2, np.uint32(i)], 'Pi')]
            for copy in (0, 1):
                macro_kernels[copy][i] = [runner.get_kernel('ApplyMacroPeriodicBoundaryConditions',
                                                            [runner.gpu_field(f), np.uint32(i)], 'Pi')
                                          for f in nn_fields]
        return MacroKernels(macro=macro_kernels, distributions=dist_kernels)

    def initial_conditions(self, runner):
        gpu_rho = runner.gpu_field(self.rho)
        gpu_phi = runner.gpu_field(self.phi)
        gpu_v = runner.gpu_field(self.v)
        gpu_map = runner.gpu_geo_map()
        args1 = [gpu_map, runner.gpu_dist(0, 0), runner.gpu_dist(1, 0)] + gpu_v + [gpu_rho, gpu_phi]
        runner.exec_kernel('SetInitialConditions', *runner.add_indirect_args(args1, 'P' * len(args1)))
        if self.config.access_pattern == 'AB':
            args2 = [gpu_map, runner.gpu_dist(0, 1), runner.gpu_dist(1, 1)] + gpu_v + [gpu_rho, gpu_phi]
            runner.exec_kernel('SetInitialConditions', *runner.add_indirect_args(args2, 'P' * len(args2)))

    def fill_module_desc(self, kw):
        super(LBBinaryFluidBase, self).fill_module_desc(kw)
        kw['tau_phi'] = self.config.tau_phi
        kw['model'] = hipabi.SLF_BGK


class LBBinaryFluidShanChen(LBBinaryFluidBase, LBForcedSim):
    """Binary fluid mixture using the Shan-Chen model (reference lb_binary.py:375-517)."""

    @classmethod
    def fields(cls):
        return [ScalarField('rho', need_nn=True), ScalarField('phi', need_nn=True), VectorField('v')]

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--visc', type=float, default=1.0, help='numerical viscosity')
        group.add_argument('--G11', type=float, default=0.0,
                           help='Shan-Chen component 1 self-interaction strength constant')
        group.add_argument('--G12', type=float, default=0.0,
                           help='Shan-Chen component 1<->2 interaction strength constant')
        group.add_argument('--G22', type=float, default=0.0,
                           help='Shan-Chen component 2 self-interaction strength constant')
        group.add_argument('--sc_potential', type=str, choices=sorted(SHAN_CHEN_POTENTIALS), default='linear',
                           help='Shan-Chen pseudopotential function to use')

    def constants(self):
        c = self.config
        return {'G11': c.G11, 'G12': c.G12, 'G21': c.G12, 'G22': c.G22}

    def fill_module_desc(self, kw):
        super(LBBinaryFluidShanChen, self).fill_module_desc(kw)
        c = self.config
        kw.update(lattice=self.grid.slf_id, simtype=hipabi.SLF_SIM_SHAN_CHEN_BINARY,
                  tau=sym.relaxation_time(c.visc), visc=c.visc, mrt_rates=sym.mrt_rates(self.grid, c.visc),
                  incompressible=0, sc_G=[c.G11, c.G12, c.G12, c.G22],
                  sc_potential=SHAN_CHEN_POTENTIALS[c.sc_potential])

    def get_compute_kernels(self, runner, full_output, bulk):
        """[(macro kernel, [collide kernels])] for the primary (A->B) and the secondary (B->A) half step
        (reference lb_binary.py:418-517)."""
        gpu_rho = runner.gpu_field(self.rho)
        gpu_phi = runner.gpu_field(self.phi)
        gpu_v = runner.gpu_field(self.v)
        gpu_map = runner.gpu_geo_map()
        d1a, d1b = runner.gpu_dist(0, 0), runner.gpu_dist(0, 1)
        d2a, d2b = runner.gpu_dist(1, 0), runner.gpu_dist(1, 1)
        options = np.uint32((1 if full_output else 0) | (2 if bulk else 0))
        tail = [gpu_rho, gpu_phi] + gpu_v + [options]
        sig = 'P' * (4 + len(gpu_v) + 1) + 'i'
        ni = self.config.needs_iteration_num

        def k(name, a, b):
            # indirect addressing: the dense-node -> slot table leads the list (reference lb_binary.py:457-465)
            args, s = runner.add_indirect_args([gpu_map, a, b] + tail, sig)
            return runner.get_kernel(name, args, s, needs_iteration=ni)

        # backends that have it sweep both lattices in ONE pass (rho, phi and the pseudopotential stencil are read once;
        # C ABI kernels "ShanChenCollideAndPropagateFused[V]"), otherwise the reference's two kernels.  SLF_SC_FUSED:
        # 0 = the reference's kernels, 1 = fused sweep reading the velocity the pass in front stored, 2 (default) = fused
        # sweep forming the node's densities and velocity itself + "ShanChenPrepareDensities" in front of it, which stores
        # the velocity only in the full_output kernels (what the host reads after an output step)
        mode = os.environ.get('SLF_SC_FUSED', '2')
        fused = getattr(runner.backend, 'supports_fused_shan_chen', False) and getattr(self.config, 'hip_sc_fused', True) and \
            mode != '0' and not runner.indirect
        local_v = fused and mode != '1' and getattr(runner.backend, 'supports_fused_shan_chen_local_velocity', False)
        sweep_name = 'ShanChenCollideAndPropagateFusedV' if local_v else 'ShanChenCollideAndPropagateFused'
        macro_name = 'ShanChenPrepareDensities' if local_v else 'ShanChenPrepareMacroFields'

        def sweeps(in1, out1, in2, out2):
            if fused:
                return [runner.get_kernel(sweep_name, [gpu_map, in1, out1, in2, out2] + tail, 'PP' + sig, needs_iteration=ni)]
            return [k('ShanChenCollideAndPropagate0', in1, out1), k('ShanChenCollideAndPropagate1', in2, out2)]

        macro1 = k(macro_name, d1a, d2a)
        primary = sweeps(d1a, d1b, d2a, d2b)
        if self.config.access_pattern == 'AB':
            macro2 = k(macro_name, d1b, d2b)
            secondary = sweeps(d1b, d1a, d2b, d2a)
        else:
            macro2, secondary = macro1, primary
        return [(macro1, primary), (macro2, secondary)]
