#!/usr/bin/env python3
"""Static evidence of the shipped kernels, no GPU needed: for every HIP source of the library the
`-Rpass-analysis=kernel-resource-usage` listing (VGPRs, spills, occupancy, SGPRs, LDS: tools/resource_usage.py) and the
instruction mix of the device code (tools/asm_mix.py), written to <outdir>/kernels_resources.txt and
<outdir>/kernels_asm_mix.txt, with the sha256 of the sources they were made from (sailfish_amd/build.py source_hash()).

    python tools/static_evidence.py profiles/r05
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    from sailfish_amd import build as slf_build
    from tools import asm_mix, resource_usage
    import re
    os.makedirs(outdir, exist_ok=True)
    stamp = slf_build.source_hash()
    flags = [f for f in slf_build.HIPCC_FLAGS if f not in ('-shared', '-fPIC')]
    sources = [s for s in slf_build.SOURCES if s != 'slf_api.hip']
    tmp = tempfile.mkdtemp()
    procs = []
    for src in sources:
        asm = os.path.join(tmp, src + '.s')
        res = os.path.join(tmp, src + '.res')
        cmd = [slf_build._hipcc()] + flags + ['--cuda-device-only', '-S', '-Rpass-analysis=kernel-resource-usage', src, '-o', asm]
        procs.append((src, asm, res, subprocess.Popen(cmd, cwd=slf_build.CSRC, stderr=open(res, 'w'))))
    with open(os.path.join(outdir, 'kernels_resources.txt'), 'w') as fr, open(os.path.join(outdir, 'kernels_asm_mix.txt'), 'w') as fa:
        fr.write('# hipcc -Rpass-analysis=kernel-resource-usage of the kernel sources with sha256 %s (tools/static_evidence.py)\n' % stamp)
        fa.write('# static instruction mix per kernel (tools/asm_mix.py) of the kernel sources with sha256 %s\n' % stamp)
        for src, asm, res, p in procs:
            if p.wait() != 0:
                raise SystemExit('hipcc failed on %s' % src)
            fr.write('== %s\n' % src)
            for r in resource_usage.parse(res):
                name = re.sub(r'slf::|\(slf::[^)]*\)|void ', '', r[0])
                fr.write('%-100s vgpr %3s agpr %2s scratch %3s occ %s sgpr %3s lds %s\n' % ((name[:100],) + r[1:]))
            fa.write('== %s\n' % src)
            for name, c in asm_mix.kernels(asm):
                grp = lambda *pfx: sum(v for k, v in c.items() if k.startswith(pfx))   # noqa: E731
                fa.write('%-84s total %5d valu %5d (pk %4d, fp %4d, dpp %3d) salu %4d vmem %3d lds %3d exp %2d\n' % (
                    name[:84], sum(c.values()), grp('v_'), grp('v_pk'),
                    grp('v_add_f', 'v_sub_f', 'v_mul_f', 'v_fma', 'v_pk_add_f', 'v_pk_mul_f', 'v_pk_fma', 'v_mac', 'v_rcp', 'v_div'),
                    sum(v for k, v in c.items() if k.endswith('_dpp')),
                    grp('s_'), grp('global_', 'buffer_', 'flat_', 'scratch_'), grp('ds_'), grp('v_exp_f')))
    print('written to', outdir, 'sources', stamp[:16])


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'profiles/static')
