"""Test doubles for the slab decomposition:
 * OracleSlab: SlabStepper backend whose rank-local operators are the CPU oracle's C functions (so the collective
   orchestration of breeze.jl_amd/distributed.py can run under gloo without a GPU);
 * ThreadedDecomposition: SlabDecomposition whose point-to-point exchange is an in-process mailbox between threads,
   so several slab ranks can share ONE GPU in a single process (exercises the wrap_y = 0 kernels on a 1-GPU box)."""
import ctypes as C
import threading

import numpy as np


def make_oracle_slab(orc, bz_dist, size, extent, rank, world, theta0=300.0, group=None, decomp=None):
    import torch
    Nx, Ny_g, Nz = size
    Ny = Ny_g // world
    dy = (extent[1][1] - extent[1][0]) / Ny_g
    y0 = extent[1][0] + rank * Ny * dy
    og = orc.Grid((Nx, Ny, Nz), x=extent[0], y=(y0, y0 + Ny * dy), z=extent[2], topology=("Periodic", "Slab", "Bounded"))
    og.dy = dy

    class OracleSlab(bz_dist.SlabStepper, orc.OracleModel):
        def __init__(self):
            orc.OracleModel.__init__(self, og, potential_temperature=theta0, initialize=False)
            d = decomp or bz_dist.SlabDecomposition(Nx, Ny, Nz, og.Hy, rank, world, group)
            bz_dist.SlabStepper.__init__(self, d)
            lx = orc.poisson_eigenvalues(Nx, og.dx, orc.PERIODIC)
            ly = orc.poisson_eigenvalues(Ny_g, dy, orc.PERIODIC)
            lam = np.ones((Ny_g, d.nkx))
            for c in range(d.nkx):
                if d.kx0 + c < d.nxh:
                    lam[:, c] = ly + lx[d.kx0 + c]
            self.lam_block = np.ascontiguousarray(lam)
            self.set_slab(theta=theta0)

        # ---- SlabStepper backend ----
        def _t(self, a):
            return torch.from_numpy(a)

        def momentum_fields(self):
            return [self._t(self.ru), self._t(self.rv), self._t(self.rw)]

        def tendency_halo_fields(self):
            return [self._t(getattr(self, n)) for n in ("ru", "rv", "rw", "u", "v", "w", "theta", "q", "T", "rtheta", "rq", "phi")]

        def local_rk3(self, dt, alpha, first):
            if first:
                for n in self.PROGNOSTIC:
                    self.U0[n][...] = getattr(self, n)
            self.rk3_substep(dt, alpha)

        def local_source(self, dt):
            g = self.grid
            # x and z halos of the momentum (the y halo came from the neighbours)
            self.fill_momentum_halos()
            rhs = np.zeros((g.Nz, g.Ny, g.Nx))
            self.lib.og_poisson_source(C.byref(self.cg), orc._p(rhs), orc._p(self.ru), orc._p(self.rv), orc._p(self.rw), C.c_double(dt))
            return torch.from_numpy(rhs)

        def local_spectral_solve(self, S, scale=1.0):
            d = self.decomp
            f = np.ascontiguousarray(S.permute(0, 2, 1).numpy()) * scale       # oracle layout: [k][ky][kx]; scale = 1/(Nx Ny)
            out = np.zeros_like(f)
            scratch = np.zeros(f.shape)
            dp = C.POINTER(C.c_double)
            self.lib.og_tridiagonal_solve(C.c_int(d.nkx), C.c_int(d.Ny_global), C.c_int(d.Nz), orc._p(self.lower), orc._p(self.diag0),
                                          orc._p(self.mass), orc._p(self.lam_block), f.view(np.float64).ctypes.data_as(dp),
                                          out.view(np.float64).ctypes.data_as(dp), orc._p(scratch))
            if d.kx0 == 0:
                out[:, 0, 0] -= out[:, 0, 0].mean()
            S.copy_(torch.from_numpy(out).permute(0, 2, 1))

        def local_project_diagnose(self, phi, below, dt):
            g = self.grid
            g.interior(self.phi)[...] = phi.numpy()
            self.phi[g.Hz:g.Hz + g.Nz, g.Hy - 1, g.Hx:g.Hx + g.Nx] = below.numpy()
            self._halo_center(self.phi)            # x + z (y is SLAB: untouched)
            self.make_pressure_correction(dt)
            self._local_update_state()

        def _local_update_state(self):
            cg = C.byref(self.cg)
            self.fill_momentum_halos()
            self._halo_center(self.rtheta)
            self._halo_center(self.rq)
            self.lib.og_compute_velocities(cg, orc._p(self.u), orc._p(self.v), orc._p(self.w), orc._p(self.ru), orc._p(self.rv), orc._p(self.rw))
            for f in (self.u, self.v, self.w):
                self._halo_velocity(f)
            self.lib.og_compute_thermo(cg, orc._p(self.theta), orc._p(self.q), orc._p(self.T), orc._p(self.rtheta), orc._p(self.rq))
            for f in (self.T, self.q, self.theta):
                self._halo_center(f)

        def local_tendencies(self):
            self.compute_tendencies()

        # ---- model API on a slab ----
        def set_slab(self, enforce_mass_conservation=True, **kw):
            g = self.grid
            Hz, Nz = g.Hz, g.Nz
            rho_c = self.ref.density[Hz:Hz + Nz][:, None, None]
            for name, value in kw.items():
                if name == "theta":
                    g.interior(self.theta)[...] = self._eval(value, "ccc")
                    g.interior(self.rtheta)[...] = rho_c * g.interior(self.theta)
                elif name in ("u", "v"):
                    loc = "fcc" if name == "u" else "cfc"
                    g.interior(getattr(self, name))[...] = self._eval(value, loc)
                    g.interior(getattr(self, "r" + name))[...] = rho_c * g.interior(getattr(self, name))
                else:
                    raise ValueError(name)
            self._local_update_state()
            self.decomp.exchange_y_halos(self.tendency_halo_fields())
            if enforce_mass_conservation:
                self.pressure_projection(1.0)

        def step(self, dt):
            if self.iteration == 0:
                self.local_tendencies()
            bz_dist.SlabStepper.time_step(self, dt)
            self.iteration += 1

    return OracleSlab()


class Mailbox:
    """Rendezvous for ThreadedDecomposition: every rank posts its sends, waits, takes its receives, waits."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.box = {}
        self.lock = threading.Lock()


def make_threaded_decomposition(bz_dist, mailbox, *args, **kw):
    class ThreadedDecomposition(bz_dist.SlabDecomposition):
        def _p2p(self, sends, recvs):
            # every rank is a thread with its own stream: the sender's pack kernels must have finished before another thread's stream
            # reads the buffer, and the receiver's copies before the sender recycles it (torch.distributed orders both by itself; this
            # stand-in has to say so — without the two synchronisations one run in four of the four-rank compressible test raced)
            import torch
            mb = mailbox
            cuda = any(t.is_cuda for t, _ in sends)
            if cuda:
                torch.cuda.current_stream().synchronize()
            with mb.lock:
                for t, dst in sends:
                    mb.box.setdefault((self.rank, dst), []).append(t)
            mb.barrier.wait()
            with mb.lock:
                for t, src in recvs:
                    t.copy_(mb.box[(src, self.rank)].pop(0))
            if cuda:
                torch.cuda.current_stream().synchronize()
            mb.barrier.wait()

    return ThreadedDecomposition(*args, **kw)


def make_oracle_compressible_slab(orc, oc, bz_dist, size, extent, rank, world, td=None, theta_ref=300.0, group=None, decomp=None,
                                  microphysics=None):
    """CompressibleOracleModel on the y-slab of `rank`: every halo fill of the oracle (x wrap + z boundary) is followed by
    the product's SlabDecomposition.exchange_y_halos, so the oracle's own sequence of fills defines the exchange points."""
    import torch
    Ny = size[1] // world
    dy = (extent[1][1] - extent[1][0]) / size[1]
    y0 = extent[1][0] + rank * Ny * dy
    grid = orc.Grid((size[0], Ny, size[2]), x=extent[0], y=(y0, y0 + Ny * dy), z=extent[2], topology=("Periodic", "Slab", "Bounded"))
    grid.dy = dy
    d = decomp or bz_dist.SlabDecomposition(size[0], Ny, size[2], grid.Hy, rank, world, group)

    class OracleCompressibleSlab(oc.CompressibleOracleModel):
        def _halo_center(self, f):
            super()._halo_center(f)
            d.exchange_y_halos([torch.from_numpy(f)])

        def _halo_w(self, f):
            super()._halo_w(f)
            d.exchange_y_halos([torch.from_numpy(f)])

    return OracleCompressibleSlab(grid, time_discretization=td or oc.SplitExplicit(substeps=6), reference_potential_temperature=theta_ref,
                                  microphysics=microphysics)
