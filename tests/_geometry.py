"""Hand-encoded node maps for parity tests (test-only helper).

Encoding = the reference's bit layout orientation | scratch | param | type
(sailfish/geo_encoder.py:365-382) with a fixed dense type-id table."""
import numpy as np

from sailfish_amd import hipabi as h

TYPE_KIND = [h.SLF_NK_FLUID, h.SLF_NK_GHOST, h.SLF_NK_FULL_BB, h.SLF_NK_REGULARIZED_VELOCITY, h.SLF_NK_HALF_BB,
             h.SLF_NK_EQUILIBRIUM_DENSITY, h.SLF_NK_UNUSED, h.SLF_NK_ZOUHE_VELOCITY, h.SLF_NK_ZOUHE_DENSITY,
             h.SLF_NK_REGULARIZED_DENSITY, h.SLF_NK_EQUILIBRIUM_VELOCITY]
# with the outflow kinds, which exist for the two-copy access pattern only (a module with them refuses AA)
TYPE_KIND_OUTFLOW = TYPE_KIND + [h.SLF_NK_COPY, h.SLF_NK_YU_OUTFLOW]
(T_FLUID, T_GHOST, T_FULLBB, T_REGVEL, T_HALFBB, T_EQDENS, T_UNUSED, T_ZHVEL, T_ZHDENS, T_REGDENS,
 T_EQVEL, T_COPY, T_YU) = range(13)
# with the do-nothing outlet (which is a boundary condition under the in-place pattern only) and the full-slip wall:
# the same dense ids as the two outflow kinds in the table above
TYPE_KIND_INPLACE = TYPE_KIND + [h.SLF_NK_DO_NOTHING, h.SLF_NK_SLIP]
T_DONOTHING, T_SLIP = 11, 12
NT_BITS = (4, 3, 0)          # type bits, param bits, scratch bits
ORIENT_SHIFT = 7


def encode(type_id, orientation=0, param=0):
    return (orientation << ORIENT_SHIFT) | (param << NT_BITS[0]) | type_id


def empty_map(desc):
    m = np.full((desc.arr_nz, desc.arr_ny, desc.arr_nx), encode(T_GHOST), dtype=np.uint32)
    if desc.lat_nz > 1:
        m[1:desc.lat_nz - 1, 1:desc.lat_ny - 1, 1:desc.lat_nx - 1] = encode(T_FLUID)
    else:
        m[0, 1:desc.lat_ny - 1, 1:desc.lat_nx - 1] = encode(T_FLUID)
    return m


def link_tags(grid, m, z, y, x, wet_types=(T_FLUID, T_HALFBB, T_REGVEL, T_EQDENS)):
    """bit (i-1) set <=> direction i points to a wet node (reference subdomain.py:593-642)."""
    tag = 0
    for i in range(1, grid.Q):
        e = grid.basis[i]
        zz = z + (e[2] if grid.dim == 3 else 0)
        t = m[zz, y + e[1], x + e[0]] & ((1 << NT_BITS[0]) - 1)
        if t in wet_types:
            tag |= 1 << (i - 1)
    return tag


def cavity_3d(desc, lid_param=0):
    """Closed box of full-way bounce-back walls with a moving lid (regularized velocity) on y = max,
    like the reference's examples/ldc_3d.py."""
    m = empty_map(desc)
    nz, ny, nx = desc.lat_nz - 2, desc.lat_ny - 2, desc.lat_nx - 2
    w = encode(T_FULLBB)
    m[1, 1:ny + 1, 1:nx + 1] = w
    m[nz, 1:ny + 1, 1:nx + 1] = w
    m[1:nz + 1, 1, 1:nx + 1] = w
    m[1:nz + 1, 1:ny + 1, 1] = w
    m[1:nz + 1, 1:ny + 1, nx] = w
    # lid: inward normal = -y = D3Q19 orientation 4
    m[2:nz, ny, 2:nx] = encode(T_REGVEL, orientation=4, param=lid_param)
    m[1:nz + 1, ny, 1] = w
    m[1:nz + 1, ny, nx] = w
    m[1, ny, 1:nx + 1] = w
    m[nz, ny, 1:nx + 1] = w
    return m


def cavity_2d(desc, lid_param=0):
    """examples/ldc_2d.py-like: full-BB walls, regularized-velocity lid on y = max (D2Q9 orientation 4 = -y)."""
    m = empty_map(desc)
    ny, nx = desc.lat_ny - 2, desc.lat_nx - 2
    w = encode(T_FULLBB)
    m[0, 1, 1:nx + 1] = w
    m[0, 1:ny + 1, 1] = w
    m[0, 1:ny + 1, nx] = w
    m[0, ny, 2:nx] = encode(T_REGVEL, orientation=4, param=lid_param)
    m[0, ny, 1] = w
    m[0, ny, nx] = w
    return m


def channel_2d_halfbb(grid, desc):
    """x-periodic channel with half-way bounce-back walls (link tags) on y = 1 and y = max."""
    m = empty_map(desc)
    ny, nx = desc.lat_ny - 2, desc.lat_nx - 2
    m[0, 1, 1:nx + 1] = encode(T_HALFBB)
    m[0, ny, 1:nx + 1] = encode(T_HALFBB)
    # periodic in x: link tags must see the wrapped neighbours -> temporarily fill the x ghosts
    mm = m.copy()
    mm[0, :, 0] = mm[0, :, nx]
    mm[0, :, nx + 1] = mm[0, :, 1]
    for y in (1, ny):
        for x in range(1, nx + 1):
            m[0, y, x] = encode(T_HALFBB, orientation=link_tags(grid, mm, 0, y, x))
    return m


def channel_2d_pressure(desc, p_in=0, p_out=1):
    """Pressure-driven channel: NTEquilibriumDensity on x = 1 (normal +x: D2Q9 orientation 1) and
    x = max (normal -x: orientation 3); full-BB walls on y (like examples/poiseuille.py --drive=pressure)."""
    m = empty_map(desc)
    ny, nx = desc.lat_ny - 2, desc.lat_nx - 2
    m[0, 2:ny, 1] = encode(T_EQDENS, orientation=1, param=p_in)
    m[0, 2:ny, nx] = encode(T_EQDENS, orientation=3, param=p_out)
    m[0, 1, 1:nx + 1] = encode(T_FULLBB)
    m[0, ny, 1:nx + 1] = encode(T_FULLBB)
    return m


def channel_3d_fullbb(desc):
    """x,z-periodic duct with full-BB walls on y (force-driven Poiseuille, examples/poiseuille_3d.py-like)."""
    m = empty_map(desc)
    ny = desc.lat_ny - 2
    m[1:desc.lat_nz - 1, 1, 1:desc.lat_nx - 1] = encode(T_FULLBB)
    m[1:desc.lat_nz - 1, ny, 1:desc.lat_nx - 1] = encode(T_FULLBB)
    return m


def channel_inlet_outlet(desc, t_in, t_out, dim):
    """Open channel along x: inlet node type t_in on x = 1 (inward normal +x: orientation 1, parameter
    slot 0), outlet t_out on x = max (normal -x: orientation 3 in D2Q9 / 2 in D3Q19, parameter slot 1 for
    a density outlet, 0 for a velocity one is not used here), full-BB walls on y, z periodic."""
    m = empty_map(desc)
    ny, nx = desc.lat_ny - 2, desc.lat_nx - 2
    o_out = 3 if dim == 2 else 2
    zs = slice(0, 1) if dim == 2 else slice(1, desc.lat_nz - 1)
    m[zs, 2:ny, 1] = encode(t_in, orientation=1, param=0)
    m[zs, 2:ny, nx] = encode(t_out, orientation=o_out, param=3 if dim == 3 else 2)
    m[zs, 1, 1:nx + 1] = encode(T_FULLBB)
    m[zs, ny, 1:nx + 1] = encode(T_FULLBB)
    return m


def channel_slip_walls(desc, dim):
    """Channel along x between two full-slip walls on y (dry nodes; inward normal +y on the row y = 1, -y on y = max:
    directions 2 / 4 in D2Q9, 3 / 4 in D3Q19), x (and z) periodic, with one full-way bounce-back block in the middle so
    that the flow has something to go around."""
    m = empty_map(desc)
    ny, nx = desc.lat_ny - 2, desc.lat_nx - 2
    zs = slice(0, 1) if dim == 2 else slice(1, desc.lat_nz - 1)
    o_low, o_high = (2, 4) if dim == 2 else (3, 4)
    m[zs, 1, 1:nx + 1] = encode(T_SLIP, orientation=o_low)
    m[zs, ny, 1:nx + 1] = encode(T_SLIP, orientation=o_high)
    m[zs, ny // 2:ny // 2 + 2, nx // 3:nx // 3 + 3] = encode(T_FULLBB)
    return m
