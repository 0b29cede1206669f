"""Single-fluid LB simulation (reference sailfish/lb_single.py:14-240)."""
from collections import defaultdict

import numpy as np

from sailfish_amd import hipabi, subdomain_runner, sym
from sailfish_amd.lb_base import KernelPair, LBForcedSim, LBSim, ScalarField, VectorField


class LBFluidSim(LBSim):
    """Simulates a single fluid."""
    subdomain_runner = subdomain_runner.SubdomainRunner

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--visc', type=float, default=1.0, help='numerical viscosity')
        group.add_argument('--incompressible', action='store_true', default=False,
                           help='use the incompressible model of Luo and He')
        group.add_argument('--model', help='LB collision model to use', type=str, choices=['bgk', 'mrt'],
                           default='bgk')
        # the two options of the reference's BGK relaxation preamble (lb_single.py:27-30, 38-42; relaxation_common.mako:166-237)
        group.add_argument('--regularized', action='store_true', default=False,
                           help='Apply the regularization procedure prior to the collision step.')
        group.add_argument('--subgrid', default='none', type=str, choices=['none', 'les-smagorinsky'],
                           help='subgrid model to use')
        group.add_argument('--smagorinsky_const', help='Smagorinsky constant', type=float, default=0.1)

    @classmethod
    def fields(cls):
        return [ScalarField('rho'), VectorField('v')]

    def fill_module_desc(self, kw):
        super(LBFluidSim, self).fill_module_desc(kw)
        cfg = self.config
        if not self.grid.model_supported(cfg.model):
            raise ValueError('model %s not supported on grid %s' % (cfg.model, self.grid.__name__))
        kw.update(lattice=self.grid.slf_id,
                  model=hipabi.SLF_MRT if cfg.model == 'mrt' else hipabi.SLF_BGK,
                  tau=sym.relaxation_time(cfg.visc), visc=cfg.visc,
                  mrt_rates=sym.mrt_rates(self.grid, cfg.visc),
                  incompressible=self.density_model(cfg))
        if getattr(cfg, 'regularized', False) or getattr(cfg, 'subgrid', 'none') != 'none':
            if cfg.model != 'bgk':
                # the reference's MRT relaxation never calls the preamble that implements them: it would ignore both silently
                raise ValueError('--regularized / --subgrid work with the BGK collision only')
            kw.update(regularized=int(bool(getattr(cfg, 'regularized', False))),
                      subgrid=hipabi.SLF_SUBGRID_LES_SMAGORINSKY if getattr(cfg, 'subgrid', 'none') == 'les-smagorinsky'
                      else hipabi.SLF_SUBGRID_NONE,
                      smagorinsky_const=float(getattr(cfg, 'smagorinsky_const', 0.1)))

    @staticmethod
    def density_model(cfg):
        """slf_module_desc::incompressible (SLF_DENSITY_*): the reference's three-way choice `incompressible` /
        `minimize_roundoff` / neither (sym.py:573-661, sym_equilibrium.py:100-118)."""
        if getattr(cfg, 'incompressible', False):
            return hipabi.SLF_DENSITY_INCOMPRESSIBLE
        if getattr(cfg, 'minimize_roundoff', False):
            if cfg.model != 'bgk':
                raise ValueError('--minimize_roundoff works with BGK-like models only (as in the reference)')
            return hipabi.SLF_DENSITY_ROUNDOFF
        return hipabi.SLF_DENSITY_COMPRESSIBLE

    ROUNDOFF_KINDS = (hipabi.SLF_NK_FLUID, hipabi.SLF_NK_GHOST, hipabi.SLF_NK_UNUSED, hipabi.SLF_NK_PROPAGATION_ONLY,
                      hipabi.SLF_NK_FULL_BB, hipabi.SLF_NK_HALF_BB, hipabi.SLF_NK_EQUILIBRIUM_DENSITY,
                      hipabi.SLF_NK_EQUILIBRIUM_VELOCITY)

    @classmethod
    def check_module_desc(cls, kw):
        """Refuses, on the host and with a clear message, what the kernels of the chosen formulation do not cover
        (the library refuses the same at module creation)."""
        if kw.get('incompressible') == hipabi.SLF_DENSITY_ROUNDOFF:
            bad = [k for k in kw.get('type_kind', []) if k not in cls.ROUNDOFF_KINDS]
            if bad:
                raise NotImplementedError('--minimize_roundoff: fluid, bounce-back and equilibrium density / velocity nodes '
                                          'only (node kinds %s are not covered: the reference\'s own regularized / Zou-He '
                                          'expressions are inconsistent under the option, DESIGN.md)' % sorted(set(bad)))

    def initial_conditions(self, runner):
        """f = feq(rho, v) on every copy of the distributions (reference lb_single.py:72-94)."""
        gpu_rho = runner.gpu_field(self.rho)
        gpu_v = runner.gpu_field(self.v)
        gpu_map = runner.gpu_geo_map()
        args1 = [runner.gpu_dist(0, 0)] + gpu_v + [gpu_rho, gpu_map]
        runner.exec_kernel('SetInitialConditions', *runner.add_indirect_args(args1, 'P' * len(args1)))
        if self.config.access_pattern == 'AB':
            args2 = [runner.gpu_dist(0, 1)] + gpu_v + [gpu_rho, gpu_map]
            runner.exec_kernel('SetInitialConditions', *runner.add_indirect_args(args2, 'P' * len(args2)))

    def _compute_kernels_arguments(self, runner, full_output, bulk):
        gpu_rho = runner.gpu_field(self.rho)
        gpu_v = runner.gpu_field(self.v)
        gpu_dist1a = runner.gpu_dist(0, 0)
        gpu_dist1b = runner.gpu_dist(0, 1)
        gpu_map = runner.gpu_geo_map()
        assert len(gpu_v) == self.dim
        args1 = [gpu_map, gpu_dist1a, gpu_dist1b, gpu_rho] + gpu_v
        args2 = [gpu_map, gpu_dist1b, gpu_dist1a, gpu_rho] + gpu_v
        options = 0
        if full_output:
            options |= 1
        if bulk:
            options |= 2
        if getattr(self.config, 'check_invalid_results_gpu', False):
            options |= 4      # on-GPU invalid value check (slf_module_poll_invalid)
        args1.append(np.uint32(options))
        args2.append(np.uint32(options))
        signature = 'P' * (len(args1) - 1) + 'i'
        args1, sig1 = runner.add_indirect_args(args1, signature)
        args2, _ = runner.add_indirect_args(args2, signature)
        return sig1, args1, args2

    def get_compute_kernels(self, runner, full_output, bulk):
        signature, args1, args2 = self._compute_kernels_arguments(runner, full_output, bulk)
        cnp_primary = runner.get_kernel('CollideAndPropagate', args1, signature,
                                        needs_iteration=self.config.needs_iteration_num)
        if self.config.access_pattern == 'AB':
            cnp_secondary = runner.get_kernel('CollideAndPropagate', args2, signature,
                                              needs_iteration=self.config.needs_iteration_num)
            return KernelPair([cnp_primary], [cnp_secondary])
        return KernelPair([cnp_primary], [cnp_primary])

    def get_resident_kernels(self, runner, scratch, steps, tile, halo):
        """(forward, backward): `steps` time steps inside ONE launch for launch-bound 2-D subdomains (library kernel
        CollideAndPropagateResident, csrc/slf_resident.hip; no counterpart in the reference, which launches
        CollideAndPropagate once per step: subdomain_runner.py:960-974).  `forward` reads the distribution arrays and
        writes the scratch copies `scratch` = [copy A, copy B or 0], `backward` the other way round; the step the launch
        starts with is the kernel's iteration argument."""
        gpu_map = runner.gpu_geo_map()
        ab = self.config.access_pattern == 'AB'
        a, b = runner.gpu_dist(0, 0), (runner.gpu_dist(0, 1) if ab else 0)
        options = 4 if getattr(self.config, 'check_invalid_results_gpu', False) else 0
        ints = [np.uint32(options), np.uint32(steps), np.uint32(tile[0]), np.uint32(tile[1]), np.uint32(halo)]
        sig = 'PPPPPiiiii'
        fwd = runner.get_kernel('CollideAndPropagateResident', [gpu_map, a, b, scratch[0], scratch[1]] + ints, sig,
                                needs_iteration=True)
        bwd = runner.get_kernel('CollideAndPropagateResident', [gpu_map, scratch[0], scratch[1], a, b] + ints, sig,
                                needs_iteration=True)
        return fwd, bwd

    def get_pbc_kernels(self, runner):
        """grid copy (0 primary, 1 secondary) -> axis -> kernels (reference lb_single.py:153-185)."""
        if runner.indirect:      # periodic axes are wrapped inside the sweep; no ghost-layer kernels exist
            return defaultdict(lambda: defaultdict(list))
        gpu_dist1a = runner.gpu_dist(0, 0)
        gpu_dist1b = runner.gpu_dist(0, 1)
        kernels = defaultdict(lambda: defaultdict(list))
        for i in range(0, self.dim):
            kernels[0][i] = [runner.get_kernel('ApplyPeriodicBoundaryConditions',
                                               [gpu_dist1a, np.uint32(i)], 'Pi')]
        if self.config.access_pattern == 'AB':
            gpu_dist, kernel = gpu_dist1b, 'ApplyPeriodicBoundaryConditions'
        else:
            gpu_dist, kernel = gpu_dist1a, 'ApplyPeriodicBoundaryConditionsWithSwap'
        for i in range(0, self.dim):
            kernels[1][i] = [runner.get_kernel(kernel, [gpu_dist, np.uint32(i)], 'Pi')]
        return kernels


class LBSingleFluidShanChen(LBFluidSim, LBForcedSim):
    """Single-component Shan-Chen model (reference lb_single.py:242-347): every step computes the density
    field first (PrepareMacroFields), then collides with the pseudopotential force added by Guo forcing."""
    nonlocality = 1
    subdomain_runner = subdomain_runner.NNSubdomainRunner

    @classmethod
    def add_options(cls, group, dim):
        group.add_argument('--G', type=float, default=1.0, help='Shan-Chen interaction strength constant')
        group.add_argument('--sc_potential', type=str, choices=['classic', 'linear'], default='linear',
                           help='Shan-Chen pseudopotential function to use')

    @classmethod
    def fields(cls):
        return [ScalarField('rho', need_nn=True), VectorField('v')]

    def constants(self):
        return {'SCG': self.config.G}

    def fill_module_desc(self, kw):
        super(LBSingleFluidShanChen, self).fill_module_desc(kw)
        if self.config.model != 'bgk':
            raise ValueError('the Shan-Chen model uses the BGK collision operator')
        kw.update(simtype=hipabi.SLF_SIM_SHAN_CHEN_SINGLE, sc_G=[self.config.G, 0.0, 0.0, 0.0],
                  sc_potential={'linear': 0, 'classic': 1}[self.config.sc_potential], tau_phi=kw['tau'])

    def get_pbc_kernels(self, runner):
        from sailfish_amd.lb_binary import MacroKernels
        dist_kernels = super(LBSingleFluidShanChen, self).get_pbc_kernels(runner)
        macro_kernels = defaultdict(lambda: defaultdict(list))
        for copy in (0, 1):
            for i in range(0, self.dim):
                macro_kernels[copy][i] = [runner.get_kernel('ApplyMacroPeriodicBoundaryConditions',
                                                            [runner.gpu_field(fp.buffer), np.uint32(i)], 'Pi')
                                          for fp in self._scalar_fields if fp.abstract.need_nn]
        return MacroKernels(macro=macro_kernels, distributions=dist_kernels)

    def get_compute_kernels(self, runner, full_output, bulk):
        gpu_rho = runner.gpu_field(self.rho)
        gpu_map = runner.gpu_geo_map()
        options = np.uint32((1 if full_output else 0) | (2 if bulk else 0))
        ni = self.config.needs_iteration_num
        macro_kernels = []
        for c in (0, 1):
            args, sig = runner.add_indirect_args([gpu_map, runner.gpu_dist(0, c), gpu_rho, options], 'PPPi')
            macro_kernels.append(runner.get_kernel('PrepareMacroFields', args, sig, needs_iteration=ni))
        sim_kernels = super(LBSingleFluidShanChen, self).get_compute_kernels(runner, full_output, bulk)
        return list(zip(macro_kernels, sim_kernels))
