"""BASELINE.json configs[2] and configs[4] at their FULL sizes (VERDICT r03 "What's weak" 2): BOMEX 256 x 256 x 128 and the
splitting-supercell shape 512 x 512 x 128.  The oracle cannot run these sizes in test time, so — as test_full_size_properties_512 does
for configs[1] — the tests check size-independent properties of the steps the bench times: finiteness, the discrete divergence left
by the projection, exact budgets of the flux-form operators (what the boundary fluxes put in is what the volume gains; closed walls),
identities of the moisture partition.  The same physics lists are compared with the oracle field by field at small sizes in
tests/test_closure.py::test_bomex_physics_list_matches_the_oracle and tests/test_gpu_compressible.py (Kessler model)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bomex(bz, size, forcing, closure=True):
    from test_forcings import _hip_forcing_kwargs
    Nx, Ny, Nz = size
    grid = bz.RectilinearGrid(size, x=(0.0, 6400.0), y=(0.0, 6400.0), z=(0.0, 3000.0))
    ref = bz.ReferenceState(grid, surface_pressure=101500.0, potential_temperature=299.1)
    kw = _hip_forcing_kwargs(bz, full=True) if forcing else {}
    if closure:
        kw["closure"] = bz.SmagorinskyLilly()
    m = bz.AtmosphereModel(grid, dynamics=bz.AnelasticDynamics(ref), advection=bz.WENO(order=5),
                           microphysics=bz.SaturationAdjustment(equilibrium=bz.WarmPhaseEquilibrium()), **kw)
    rng = np.random.default_rng(938)
    noise_t, noise_q = rng.standard_normal((Nz, Ny, Nx)), rng.standard_normal((Nz, Ny, Nx))

    def theta(x, y, z):      # Siebesma et al. (2003) profile + the example's noise below 1600 m (examples/bomex.jl)
        base = np.where(z < 520.0, 298.7, np.where(z < 1480.0, 298.7 + (z - 520.0) * (302.4 - 298.7) / 960.0,
                        np.where(z < 2000.0, 302.4 + (z - 1480.0) * (308.2 - 302.4) / 520.0, 308.2 + (z - 2000.0) * 3.65e-3)))
        return base + 0.1 * noise_t * (z < 1600.0)

    def qt(x, y, z):
        base = np.where(z < 520.0, 17.0 + z * (16.3 - 17.0) / 520.0, np.where(z < 1480.0, 16.3 + (z - 520.0) * (10.7 - 16.3) / 960.0,
                        np.where(z < 2000.0, 10.7 + (z - 1480.0) * (4.2 - 10.7) / 520.0, 4.2 + (z - 2000.0) * (-1.2e-3)))) * 1e-3
        return base + 2.5e-5 * noise_q * (z < 1600.0)

    m.set(θ=theta, qᵗ=qt, u=lambda x, y, z: np.where(z < 700.0, -8.75, -8.75 + (z - 700.0) * 1.8e-3) + 0 * x + 0 * y)
    return m


def test_config2_bomex_256x256x128_full_physics_list(bz):
    """configs[2] at full size, the physics list of examples/bomex.jl (saturation adjustment, SmagorinskyLilly, f-plane, geostrophic and
    subsidence forcing, drying, radiative cooling, bottom heat / moisture / drag fluxes), five steps of 2 s."""
    import torch
    m = _bomex(bz, (256, 256, 128), forcing=True)
    for _ in range(5):
        m.time_step(2.0)
    m.synchronize()
    for f in (m.temperature, m.potential_temperature, m.velocities["u"], m.velocities["w"], m.momentum["ρv"], m.moisture_density):
        assert bool(torch.isfinite(f.interior).all())
    scale = float(m.momentum["ρu"].interior.abs().max())
    assert m.max_abs_divergence() < 1e-11 * scale          # (kg m^-2 s^-1) / m: the projection closes the step
    qv, ql, q = m.microphysical_fields["qᵛ"].interior, m.microphysical_fields["qˡ"].interior, m.specific_moisture.interior
    assert float(ql.min()) >= 0.0 and float((qv + ql - q).abs().max()) < 1e-15          # q^e = q^v + q^l, liquid only where saturated
    assert float(qv.min()) > 0.0 and float(q.max()) < 0.02                                 # (the profile is still subsaturated ten seconds in: no cloud yet)
    assert float(m.momentum["ρw"].interior[0].abs().max()) == 0.0 and float(m.momentum["ρw"].interior[-1].abs().max()) == 0.0
    T = m.temperature.interior
    assert 270.0 < float(T.min()) and float(T.max()) < 305.0


def test_config2_bomex_256x256x128_conserves_what_advection_and_closure_must(bz):
    """the same grid and initial state without the forcing stack: WENO advection, the Smagorinsky flux divergence and the projection
    are all in flux form inside closed / periodic boundaries, so the volume sums of rho theta and rho q^e may not move (1e-13 relative
    over five steps), and the horizontal momentum sums only through the (zero) boundary stress"""
    import torch
    m = _bomex(bz, (256, 256, 128), forcing=False)
    s0 = [float(f.interior.sum(dtype=torch.float64)) for f in (m.potential_temperature_density, m.moisture_density)]
    for _ in range(5):
        m.time_step(2.0)
    m.synchronize()
    s1 = [float(f.interior.sum(dtype=torch.float64)) for f in (m.potential_temperature_density, m.moisture_density)]
    for a, b in zip(s0, s1):
        assert abs(b - a) <= 1e-13 * abs(a), (a, b)
    assert bool(torch.isfinite(m.velocities["w"].interior).all())


def test_config4_supercell_512x512x128_compressible_kessler(bz):
    """configs[4] at full size on one GPU (the 8-GPU split is the same kernels on y-slabs; tests/test_comm.py compares slab ranks with
    this seam): CompressibleDynamics + split-explicit WS-RK3 + DCMIP2016 Kessler, three steps of 2 s.  Dry mass is conserved by the
    flux-form density equation inside the closed lid (1e-12), total water changes only by what rains out at the surface (nothing yet,
    three steps in), the walls stay closed, the state stays physical."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    m = bench.supercell_model(bz, (512, 512, 128), "cuda:0")
    rho_d = m.dynamics.dry_density
    mass0 = float(rho_d.interior.sum(dtype=torch.float64))
    water = lambda: float((m.moisture_density.interior + m.microphysical_fields["ρqᶜˡ"].interior + m.microphysical_fields["ρqʳ"].interior).sum(dtype=torch.float64))      # noqa: E731
    w0 = water()
    for _ in range(3):
        m.time_step(2.0)
    m.synchronize()
    assert abs(float(rho_d.interior.sum(dtype=torch.float64)) - mass0) <= 1e-12 * mass0
    rained = float(m.microphysical_fields["precipitation_rate"].abs().max())
    # no rain has reached the ground; the transport is in flux form and the Kessler conversions move mass between the species, but the
    # scheme clips the WENO undershoots of the (initially zero) condensate fields at zero (dcmip2016_kessler.jl: max(0, .)), which
    # creates 1e-7 of the total per step at this resolution — the budget closes to that
    assert abs(water() - w0) <= 1e-6 * w0 + rained * 1e12
    for f in (m.temperature, m.velocities["u"], m.velocities["w"], m.dynamics.pressure):
        assert bool(torch.isfinite(f.interior).all())
    w = m.velocities["w"].interior
    assert float(w[0].abs().max()) == 0.0 and float(w[-1].abs().max()) == 0.0
    assert 0.05 < float(w.abs().max()) < 20.0          # the bubble has started to rise
    # a neutral (theta = 300 K) column reaches 300 - g z / c_p = 105 K and 2.5 kPa at its 20 km lid
    assert 100.0 < float(m.temperature.interior.min()) and float(m.temperature.interior.max()) < 310.0
    assert float(m.dynamics.pressure.interior.min()) > 2e3


def test_config3_per_rank_slab_1024x128x512(bz, monkeypatch):
    """configs[3] (dry bubble 1024 x 1024 x 512 over 8 GPUs) hands every rank a 1024 x 128 x 512 slab (round 6, VERDICT r05 item 2b: the shape
    existed only in tools/check_config3_slab_shape.py).  One rank of that shape runs the library-owned slab step — 1024-cell rows (teams of
    two wavefronts in the x transforms), 128-row slabs, 512 levels (the cooperative tridiagonal kernel's longest columns) — with every message
    addressed to itself (BZ_COMM_SELF_MESSAGES: the halo rows and the all-to-all blocks really travel through the transport): three steps stay
    finite, the projection leaves no discrete divergence (computed here from the momentum components, not by the library), rho theta is
    conserved to rounding and the lid and the floor stay closed.  Eight ranks of small shapes run in tests/test_comm.py."""
    import uuid
    import torch
    from breeze_jl_amd import distributed as bz_dist
    monkeypatch.setenv("BZ_COMM_SELF_MESSAGES", "1")
    size = (1024, 128, 512)
    G = bz.RectilinearGrid(size, x=(-20e3, 20e3), y=(-2.5e3, 2.5e3), z=(0.0, 10e3))

    def bubble(x, y, z):
        r = np.sqrt(x ** 2 + y ** 2 + (z - 3000.0) ** 2)
        return 300.0 * np.exp(1e-6 * z / 9.81) + 10.0 * np.maximum(0.0, 1.0 - r / 2000.0)

    m = bz_dist.SlabAtmosphereModel(G, 0, 1, advection=bz.WENO(order=5), surface_pressure=101325, potential_temperature=300,
                                    device="cuda:0", transport="local:" + uuid.uuid4().hex)
    m.set(θ=bubble, u=2.0)
    rth = m.potential_temperature_density
    s0 = float(rth.interior.sum(dtype=torch.float64))
    for _ in range(3):
        m.time_step(1.0)
    m.synchronize()
    assert m.comm_info()[1] > 0          # bytes went through the transport
    ru, rv, rw = (m.momentum[k].interior for k in ("ρu", "ρv", "ρw"))
    for f in (ru, rv, rw, rth.interior, m.temperature.interior):
        assert bool(torch.isfinite(f).all())
    dx, dy, dz = 40e3 / 1024, 5e3 / 128, 10e3 / 512
    div = (torch.roll(ru, -1, 2) - ru) / dx + (torch.roll(rv, -1, 1) - rv) / dy + (rw[1:] - rw[:-1]) / dz
    scale = float(ru.abs().max())
    assert float(div.abs().max()) < 1e-11 * scale
    assert abs(float(rth.interior.sum(dtype=torch.float64)) - s0) <= 1e-13 * abs(s0)
    assert float(rw[0].abs().max()) == 0.0 and float(rw[-1].abs().max()) == 0.0
    assert float(rw.abs().max()) > 1e-3          # the bubble rises
