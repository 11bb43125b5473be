"""One-process-per-GPU y-slab decomposition of the anelastic step.

The reference has no distributed code of its own (it re-exports Oceananigans' MPI-based `Distributed`,
src/Breeze.jl:172,183,209; no test or example uses it), so this is this repo's design, MI355X-first:

* 1-D slabs in y over the ranks of one node.  x stays whole and contiguous (coalescing, local x-FFT), z stays
  whole (tridiagonal solve, vertical WENO stencils, column physics).
* y-halo exchange = paired send/recv with the two ring neighbours, all fields of an exchange packed into one
  message per direction (xGMI is point-to-point: one message per link, not a ring collective).
* Poisson solve = local real-to-complex x-FFT -> transpose to kx-slabs (every rank sends one block to every
  other rank, so all xGMI links of a GPU are busy at once) -> y-FFT, Thomas solve in z, inverse y-FFT ->
  transpose back -> complex-to-real x-FFT.  The transposes are point-to-point batches (`batch_isend_irecv`), which
  RCCL runs as one grouped operation and gloo (CPU tests) also supports.
* Everything else is rank-local: the same HIP kernels as the single-GPU path, through the slab entry points of
  include/breeze_hip.h (`bz_create_slab`, `bz_ssp_rk3_substep_fused`, `bz_poisson_source_term`,
  `bz_spectral_tridiagonal_solve`, `bz_project_and_diagnose`, `bz_compute_tendencies`).

`SlabStepper` holds the collective orchestration and is backend-agnostic (any object providing the `local_*`
operations on torch tensors), which is how tests/test_distributed.py runs it on CPU with gloo, world size 2 and 4,
against the single-process oracle.  `SlabAtmosphereModel` is the GPU backend.

Status: verified on CPU (gloo).  The RCCL path has not run on multi-GPU hardware from this container (gpurun gives
one GPU); the round-end scaling run is its first exposure.
"""
import ctypes as C

import numpy as np

from . import _lib
from .grids import Bounded, Center, Face, Periodic, RectilinearGrid
from .model import AnelasticDynamics, AtmosphereModel
from .thermodynamics import (ReferenceState, ThermodynamicConstants, dry_air_gas_constant,
                             vapor_gas_constant)


class SlabDecomposition:
    """Communication pattern of a y-slab decomposition over `world` ranks (torch.distributed, any backend)."""

    def __init__(self, Nx, Ny_local, Nz, Hy, rank=0, world=1, group=None):
        self.Nx, self.Ny, self.Nz, self.Hy = Nx, Ny_local, Nz, Hy
        self.rank, self.world, self.group = rank, world, group
        self.Ny_global = Ny_local * world
        self.nxh = Nx // 2 + 1
        self.nkx = -(-self.nxh // world)            # kx columns per rank (zero-padded half spectrum)
        self.nxh_pad = self.nkx * world
        self.kx0 = rank * self.nkx
        self.upper = (rank + 1) % world
        self.lower = (rank - 1) % world
        if Ny_local < Hy:
            raise ValueError("a slab must hold at least Hy rows")
        # optional device kernel for the transposing packs: pack(src (Nz, A, ld) complex, c0, B, valid) -> (Nz, B, A) complex
        # (SlabAtmosphereModel installs bz_pack_transpose; CPU backends use the torch expression)
        self.pack = None
        self._comm_events = []
        # optional device kernel for the halo rows: rows(fields, row0, nrows, buffer, unpack) gathers / scatters parent rows of
        # every field through one contiguous buffer (bz_pack_rows); CPU backends use slicing
        self.rows = None

    def _pack(self, T, c0, B, valid=None):
        """(Nz, A, ld) complex -> contiguous (Nz, B, A) with out[k, b, a] = T[k, a, c0 + b] (zero for c0 + b >= valid)."""
        import torch
        valid = T.shape[2] if valid is None else valid
        if self.pack is not None:
            return self.pack(T, c0, B, valid)
        blk = T[:, :, c0:min(c0 + B, valid)]
        if blk.shape[2] < B:
            blk = torch.nn.functional.pad(blk, (0, B - blk.shape[2]))
        return blk.permute(0, 2, 1).contiguous()

    # -- point-to-point helper -------------------------------------------------
    def _p2p(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(tensor, src)] — all posted as one batch."""
        import torch.distributed as dist
        ops = [dist.P2POp(dist.isend, t, d, group=self.group) for t, d in sends]
        ops += [dist.P2POp(dist.irecv, t, s, group=self.group) for t, s in recvs]
        if not ops:
            return
        timed = self.profile
        if timed:
            import time
            import torch
            cuda = sends[0][0].is_cuda if sends else recvs[0][0].is_cuda
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if timed:
            self.comm_bytes += sum(t.numel() * t.element_size() for t, _ in sends)
            if cuda:
                e1.record()
                self._comm_events.append((e0, e1))
            else:
                self._comm_ms += 1e3 * (time.perf_counter() - t0)

    # -- instrumentation: time spent inside the point-to-point batches (HIP events on the stream the exchange runs on) --
    profile = False
    comm_bytes = 0
    _comm_ms = 0.0

    def profile_reset(self):
        self._comm_events, self._comm_ms, self.comm_bytes = [], 0.0, 0

    def comm_ms(self):
        """Milliseconds spent in exchanges since profile_reset() (synchronises the recorded events)."""
        ms = self._comm_ms
        for e0, e1 in getattr(self, "_comm_events", []):
            e1.synchronize()
            ms += e0.elapsed_time(e1)
        return ms

    # -- halos -----------------------------------------------------------------
    def exchange_y_halos(self, fields):
        """Fill the Hy y-halo rows of every parent tensor ((z, y, x)-shaped, full x width and all z levels)
        from the ring neighbours (periodic in y over the whole domain)."""
        import torch
        Hy, Ny = self.Hy, self.Ny
        if not fields:
            return
        if self.world == 1:      # direct row copies (measured faster than a gather + scatter through a buffer)
            for f in fields:
                f[:, :Hy, :] = f[:, Ny:Ny + Hy, :]
                f[:, Hy + Ny:, :] = f[:, Hy:2 * Hy, :]
            return
        if self.rows is not None and fields[0].is_cuda:
            total = sum(f.shape[0] * Hy * f.shape[2] for f in fields)
            up, down = (torch.empty(total, dtype=fields[0].dtype, device=fields[0].device) for _ in range(2))
            self.rows(fields, Ny, Hy, up, 0)                                     # my top rows -> upper's bottom halo
            self.rows(fields, Hy, Hy, down, 0)                                   # my bottom rows -> lower's top halo
            from_lower, from_upper = torch.empty_like(up), torch.empty_like(down)
            self._p2p([(up, self.upper), (down, self.lower)], [(from_lower, self.lower), (from_upper, self.upper)])
            self.rows(fields, 0, Hy, from_lower, 1)
            self.rows(fields, Hy + Ny, Hy, from_upper, 1)
            return
        up = torch.cat([f[:, Ny:Ny + Hy, :].reshape(-1) for f in fields])        # my top rows -> upper's bottom halo
        down = torch.cat([f[:, Hy:2 * Hy, :].reshape(-1) for f in fields])       # my bottom rows -> lower's top halo
        from_lower, from_upper = torch.empty_like(up), torch.empty_like(down)
        # order matters when lower == upper (world 2): sends "up then down", receives "from lower then from upper"
        self._p2p([(up, self.upper), (down, self.lower)], [(from_lower, self.lower), (from_upper, self.upper)])
        o = 0
        for f in fields:
            n = f.shape[0] * Hy * f.shape[2]
            f[:, :Hy, :] = from_lower[o:o + n].view(f.shape[0], Hy, f.shape[2])
            f[:, Hy + Ny:, :] = from_upper[o:o + n].view(f.shape[0], Hy, f.shape[2])
            o += n

    def row_from_lower(self, top_row):
        """Send my top interior row (any tensor) to the upper neighbour; return the lower neighbour's."""
        import torch
        if self.world == 1:
            return top_row.clone()
        buf = torch.empty_like(top_row)
        self._p2p([(top_row.contiguous(), self.upper)], [(buf, self.lower)])
        return buf

    # -- transposes of the distributed transform ------------------------------------
    # The kx-slab block is stored (Nz, nkx, Ny_global): ky fastest, so the y transform and the Thomas kernel's
    # column index run along contiguous memory (a strided torch.fft.ifft along dim 1 costs 7.9 ms at 512^3 on
    # MI355X, the contiguous one 0.8 ms).  The transposition rides on the pack copy that the exchange needs anyway.
    def to_kx_slabs(self, R):
        """(Nz, Ny_local, nxh) complex, y-slab  ->  (Nz, nkx, Ny_global) complex, kx-slab (columns >= nxh are the zero padding)."""
        import torch
        Nz, Ny, nkx, W = self.Nz, self.Ny, self.nkx, self.world
        if W == 1:
            return self._pack(R, 0, nkx, self.nxh)
        send = [torch.view_as_real(self._pack(R, p * nkx, nkx, self.nxh)) for p in range(W)]
        recv = [torch.empty_like(send[0]) for _ in range(W)]          # (Nz, nkx, Ny_local, 2) from each rank
        recv[self.rank].copy_(send[self.rank])
        peers = [p for p in range(W) if p != self.rank]
        self._p2p([(send[p], p) for p in peers], [(recv[p], p) for p in peers])
        return torch.view_as_complex(torch.cat(recv, dim=2))          # rank q's rows are global rows q*Ny ...

    def to_y_slabs(self, S):
        """(Nz, nkx, Ny_global) complex, kx-slab  ->  (Nz, Ny_local, nxh_pad) complex, y-slab."""
        import torch
        Ny, W = self.Ny, self.world
        if W == 1:
            return self._pack(S, 0, Ny)
        send = [torch.view_as_real(self._pack(S, q * Ny, Ny)) for q in range(W)]
        recv = [torch.empty_like(send[0]) for _ in range(W)]          # (Nz, Ny_local, nkx, 2) from each rank
        recv[self.rank].copy_(send[self.rank])
        peers = [p for p in range(W) if p != self.rank]
        self._p2p([(send[p], p) for p in peers], [(recv[p], p) for p in peers])
        return torch.view_as_complex(torch.cat(recv, dim=2))          # rank p's block is kx in [p*nkx, (p+1)*nkx)


class SlabStepper:
    """Collective orchestration of the SSP-RK3 anelastic step on y-slabs
    (call order of src/TimeSteppers/ssp_runge_kutta_3.jl:209-278 and
    src/AnelasticEquations/anelastic_time_stepping.jl:26-78).

    A backend supplies, on its own slab (torch tensors, any device):
      local_rk3(dt, alpha, first)            ssp_rk3_substep! (+ store_initial_state! when first)
      local_source(dt) -> rhs                (Nz, Ny, Nx) real; needs the y halo of the momentum current
      local_spectral_solve(S)                in place on (Nz, nkx, Ny_global) complex; zero-mean gauge
      local_project_diagnose(phi, below, dt) projection + velocities + theta, q, T + x/z halo fills
      local_tendencies()                     needs the y halos of tendency_halo_fields() current
      momentum_fields(), tendency_halo_fields()  -> lists of parent tensors
    """

    def __init__(self, decomp):
        self.decomp = decomp

    def poisson_solve(self, rhs):
        import torch
        d = self.decomp
        R = self.fft_x_forward(rhs)                                     # local x transform, (Nz, Ny, nxh)
        S = d.to_kx_slabs(R)                                            # (Nz, nkx, Ny_global); the pack pads kx >= nxh with zeros
        S = self.fft_y(S, inverse=False)                                # y transform, contiguous, all rows present
        # the inverse transforms run unnormalised (no scaling pass over the data); 1/(Nx Ny) rides on the right-hand side
        # inside the tridiagonal kernel
        self.local_spectral_solve(S, 1.0 / (d.Nx * d.Ny_global))
        S = self.fft_y(S, inverse=True)
        R = d.to_y_slabs(S)
        return self.fft_x_inverse(R)

    # local transforms: torch.fft by default (any device; the CPU tests), rocFFT plans of the C library on the GPU backend
    def fft_x_forward(self, rhs):
        import torch
        return torch.fft.rfft(rhs, dim=2)

    def fft_y(self, S, inverse):
        import torch
        return torch.fft.ifft(S, dim=2, norm="forward") if inverse else torch.fft.fft(S, dim=2)

    def fft_x_inverse(self, R):
        import torch
        d = self.decomp
        return torch.fft.irfft(R[:, :, :d.nxh], n=d.Nx, dim=2, norm="forward").contiguous()

    fused_rk = False      # backend folds the RK update into its tendency kernels (predictor momentum in separate arrays)

    def pressure_projection(self, dt, predictor=False):
        """compute_pressure_correction! + make_pressure_correction! + update_state!(compute_tendencies=false)."""
        d = self.decomp
        d.exchange_y_halos(self.predictor_fields() if predictor else self.momentum_fields())
        phi = self.poisson_solve(self.local_source(dt, predictor) if predictor else self.local_source(dt))
        below = d.row_from_lower(phi[:, d.Ny - 1, :])
        if predictor:
            self.local_project_diagnose(phi, below, dt, True)
        else:
            self.local_project_diagnose(phi, below, dt)
        d.exchange_y_halos(self.tendency_halo_fields())

    def stage(self, dt, alpha, first):
        if self.fused_rk:
            self.local_tend_rk(dt, alpha, first)          # tendencies of the current state + RK update in one pass
            self.pressure_projection(alpha * dt, predictor=True)
        else:
            self.local_rk3(dt, alpha, first)
            self.pressure_projection(alpha * dt)
            self.local_tendencies()

    def time_step(self, dt):
        for n, alpha in enumerate((1.0, 1.0 / 4.0, 2.0 / 3.0)):
            self.stage(dt, alpha, n == 0)


_LOC = {"ccc": (Center, Center, Center), "fcc": (Face, Center, Center),
        "cfc": (Center, Face, Center), "ccf": (Center, Center, Face)}


def attach_library_transport(self, transport, group):
    """Give the slab context `self._ctx` a library-owned communicator (csrc/bz_comm.hip): "rccl" (unique id made on rank 0 and
    broadcast over the torch.distributed group) or "local:<name>" (in-process transport between contexts of one process)."""
    import torch
    lib = self._lib
    if transport.startswith("local:"):
        self._check(lib.bz_comm_init_local(self._ctx, transport[6:].encode()), "bz_comm_init_local")
        return
    if transport != "rccl":
        raise ValueError(f"unknown transport {transport!r}")
    ident = torch.zeros(_lib.BZ_UNIQUE_ID_BYTES, dtype=torch.uint8)
    if self.rank == 0:
        buf = (C.c_ubyte * _lib.BZ_UNIQUE_ID_BYTES)()
        rc = lib.bz_comm_unique_id(buf)
        if rc != 0:
            raise _lib.BreezeHIPError(f"bz_comm_unique_id failed with code {rc}")
        ident = torch.frombuffer(bytearray(buf), dtype=torch.uint8).clone()
    if self.world > 1:
        import torch.distributed as dist
        ident = ident.to(self.device)
        dist.broadcast(ident, src=0, group=group)
        ident = ident.cpu()
    raw = (C.c_ubyte * _lib.BZ_UNIQUE_ID_BYTES)(*ident.tolist())
    # RCCL prints a version banner on C stdout when a communicator is created; a host that prints machine-readable results on
    # stdout (bench.py) must not get it there: point fd 1 at stderr for the duration of the call and flush C stdio inside it
    import os
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        rc = lib.bz_comm_init_rccl(self._ctx, raw)
        C.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    self._check(rc, "bz_comm_init_rccl")



class LibraryComm:
    """Stand-in for SlabDecomposition when the library owns the communicator: the only thing the host still asks for is a y-halo
    exchange of a list of parent arrays (bz_comm_exchange_y_halos)."""

    def __init__(self, model):
        self.model = model

    def exchange_y_halos(self, tensors):
        m = self.model
        n = len(tensors)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        levels = (C.c_int32 * n)(*[t.shape[0] for t in tensors])
        m._check(m._lib.bz_comm_exchange_y_halos(m._ctx, ptrs, levels, n), "bz_comm_exchange_y_halos")


class LibrarySlabAtmosphereModel(AtmosphereModel):
    """AtmosphereModel on one y-slab with the library-owned communicator (transport "rccl" or "local:<name>"): every component
    AtmosphereModel wires up — saturation adjustment, SmagorinskyLilly, column forcings, bottom fluxes, the physics list of
    BASELINE configs[2] — runs decomposed; `time_step` is one C call per step and rank (csrc/bz_comm.hip: the lean seam for the
    dry / vapour model; the fused-RK tier with halo exchanges, all-reduced horizontal averages and a viscosity kernel that covers
    the rows next to the slab for the BOMEX list and tracers; an operator-by-operator distributed step for StaticEnergy, Kessler
    species and WENO(order = 7 / 9))."""

    def __init__(self, global_grid, rank, world, transport="rccl", group=None, surface_pressure=101325, potential_temperature=288,
                 standard_pressure=1e5, thermodynamic_constants=None, device=None, **kw):
        import torch
        G = global_grid
        if G.topology != (Periodic, Periodic, Bounded):
            raise NotImplementedError("slab decomposition implements topology (Periodic, Periodic, Bounded)")
        if G.Ny % world:
            raise ValueError(f"Ny={G.Ny} is not divisible by {world} ranks")
        if transport == "torch":
            raise ValueError('LibrarySlabAtmosphereModel runs on the library-owned communicator: transport "rccl" or "local:<name>"')
        self.global_grid, self.rank, self.world, self.transport = G, rank, world, transport
        self._comm_ready = False
        Ny = G.Ny // world
        y0 = G.yᶠ[0] + rank * Ny * G.Δy
        z = (G.zᶠ[0], G.zᶠ[-1]) if G.regular_z else G.zᶠ
        grid = RectilinearGrid((G.Nx, Ny, G.Nz), x=(G.xᶠ[0], G.xᶠ[0] + G.Nx * G.Δx), y=(y0, y0 + Ny * G.Δy), z=z,
                               halo=(G.Hx, G.Hy, G.Hz), float_type=G.float_type)      # eltype(grid) travels with the slab
        grid.Δx, grid.Δy = G.Δx, G.Δy          # bit-identical spacings on every rank
        c = thermodynamic_constants or ThermodynamicConstants()
        ref = ReferenceState(grid, c, surface_pressure, potential_temperature, standard_pressure)
        dev = device if device is not None else f"cuda:{torch.cuda.current_device()}"
        super().__init__(grid, dynamics=AnelasticDynamics(ref), thermodynamic_constants=c, device=dev, **kw)
        attach_library_transport(self, transport, group)
        self._comm_ready = True
        self.set(θ=ref.potential_temperature)

    def _create_context(self, lib, bg, bc, br, order):
        return lib.bz_create_slab(C.byref(self._ctx), C.byref(bg), C.byref(bc), C.byref(br), order, self.world, self.rank)

    def _finish_set(self, enforce_mass_conservation):
        if not self._comm_ready:
            return                                   # the constructor's own set-up runs before the communicator exists
        self._check(self._lib.bz_comm_update_state_and_project(self._ctx, C.byref(self._state), C.byref(self._G), 1.0,
                                                               1 if enforce_mass_conservation else 0), "bz_comm_update_state_and_project")

    def time_step(self, Δt):
        if self.clock.iteration == 0:                 # maybe_prepare_first_time_step!: update_state! with the exchanges
            self._check(self._lib.bz_comm_update_state_and_project(self._ctx, C.byref(self._state), C.byref(self._G), 1.0, 0),
                        "bz_comm_update_state_and_project")
        self._check(self._lib.bz_time_step_anelastic(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G), float(Δt)),
                    "bz_time_step_anelastic")
        self.clock.time += float(Δt)
        self.clock.last_Δt = float(Δt)
        self.clock.iteration += 1

    def time_steps(self, Δt, n, diagnose_last=True):
        """n distributed steps in one C call (bz_time_steps_anelastic on a slab context): steps whose diagnostics nobody reads end with the
        momentum-only projection and exchange five halo fields instead of ten."""
        if n <= 0:
            return
        if self.clock.iteration == 0:
            self._check(self._lib.bz_comm_update_state_and_project(self._ctx, C.byref(self._state), C.byref(self._G), 1.0, 0),
                        "bz_comm_update_state_and_project")
        self._check(self._lib.bz_time_steps_anelastic(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G), float(Δt), int(n),
                                                      1 if diagnose_last else 0), "bz_time_steps_anelastic")
        self.clock.time += n * float(Δt)
        self.clock.last_Δt = float(Δt)
        self.clock.iteration += int(n)

    _collective_refresh = True      # Field reads of stale diagnostics raise instead of entering the collective alone (model.refresh_diagnostics())

    def _refresh_diagnostics(self):
        """update_state! with the y-halo exchanges — a collective: every rank of the communicator has to get here (all ranks read, or none)."""
        self._check(self._lib.bz_comm_update_state_and_project(self._ctx, C.byref(self._state), C.byref(self._G), 1.0, 0),
                    "bz_comm_update_state_and_project")

    def comm_info(self):
        """(transport name, bytes this rank has sent, number of exchanges) of the library-owned communicator."""
        name, nbytes, nex = C.c_char_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.bz_comm_info(self._ctx, C.byref(name), C.byref(nbytes), C.byref(nex)), "bz_comm_info")
        return name.value.decode(), nbytes.value, nex.value


class SlabAtmosphereModel(SlabStepper):
    """AtmosphereModel on one y-slab of a (Periodic, Periodic, Bounded) global grid: rank r of `world` owns rows
    [r*Ny/world, (r+1)*Ny/world).  Same fields, kernels and call order as `AtmosphereModel`; halos in y and the
    Poisson transposes go through torch.distributed (backend "nccl" = RCCL on ROCm)."""

    def __init__(self, global_grid, rank, world, advection=None, thermodynamic_constants=None,
                 surface_pressure=101325, potential_temperature=288, standard_pressure=1e5, device=None, group=None,
                 transport="torch"):
        """transport: who carries the y halos and the FFT transposes —
             "rccl"          the C library itself (bz_comm_init_rccl: RCCL send / recv groups on HIP streams; the whole step is ONE
                             C call, bz_time_step_anelastic, with the halo exchange overlapped with the interior tendency tiles).
                             torch.distributed is used once, to hand rank 0's 128-byte unique id to the other ranks.
             "local:<name>"  the C library's in-process transport (ranks = threads of this process sharing one or more GPUs)
             "torch"         this module's Python orchestration over torch.distributed (SlabStepper; also what the CPU tests drive)"""
        import torch
        from .model import Clock, Field, WENO
        if global_grid.topology != (Periodic, Periodic, Bounded):
            raise NotImplementedError("slab decomposition implements topology (Periodic, Periodic, Bounded)")
        if advection is None or getattr(advection, "order", None) != 5:
            raise NotImplementedError("the HIP path requires advection=WENO(order=5)")
        if global_grid.Ny % world:
            raise ValueError(f"Ny={global_grid.Ny} is not divisible by {world} ranks")
        if not torch.cuda.is_available():
            raise RuntimeError("SlabAtmosphereModel needs a GPU: the HIP path has no CPU fallback")
        self.global_grid = G = global_grid
        if G.ftype != 8:
            raise NotImplementedError('SlabAtmosphereModel(transport="torch") is the Float64 cross-check of the library-owned step; '
                                      "Float32 grids decompose through LibrarySlabAtmosphereModel")
        self.rank, self.world = rank, world
        Ny = G.Ny // world
        y0 = G.yᶠ[0] + rank * Ny * G.Δy
        z = (G.zᶠ[0], G.zᶠ[-1]) if G.regular_z else G.zᶠ
        self.grid = grid = RectilinearGrid((G.Nx, Ny, G.Nz), x=(G.xᶠ[0], G.xᶠ[0] + G.Nx * G.Δx),
                                           y=(y0, y0 + Ny * G.Δy), z=z, halo=(G.Hx, G.Hy, G.Hz))
        grid.Δx, grid.Δy = G.Δx, G.Δy          # bit-identical spacings on every rank
        SlabStepper.__init__(self, getattr(self, "_decomp_override", None) or
                             SlabDecomposition(G.Nx, Ny, G.Nz, G.Hy, rank, world, group))
        self.thermodynamic_constants = c = thermodynamic_constants or ThermodynamicConstants()
        self.reference_state = ref = ReferenceState(grid, c, surface_pressure, potential_temperature, standard_pressure)
        self.clock = Clock()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self._lib = lib = _lib.load()

        def fld(loc):
            return Field(grid, _LOC[loc], self.device)

        self.momentum = {"ρu": fld("fcc"), "ρv": fld("cfc"), "ρw": fld("ccf")}
        self.velocities = {"u": fld("fcc"), "v": fld("cfc"), "w": fld("ccf")}
        self.potential_temperature_density, self.potential_temperature = fld("ccc"), fld("ccc")
        self.moisture_density, self.specific_moisture = fld("ccc"), fld("ccc")
        self.temperature, self.pressure_anomaly = fld("ccc"), fld("ccc")
        prog = self.prognostic_fields()
        self.U0 = {k: Field(grid, f.loc, self.device) for k, f in prog.items()}
        self.G = {k: Field(grid, f.loc, self.device) for k, f in prog.items()}

        self._zf = np.ascontiguousarray(grid.zᶠ, dtype=np.float64)
        bg = _lib.bz_grid()
        bg.Nx, bg.Ny, bg.Nz = grid.Nx, grid.Ny, grid.Nz
        bg.Hx, bg.Hy, bg.Hz = grid.Hx, grid.Hy, grid.Hz
        for dd, t in enumerate(grid.topology_codes()):
            bg.topo[dd] = t
        bg.ftype = 8
        bg.dx, bg.dy = grid.Δx, grid.Δy
        bg.zf = self._zf.ctypes.data_as(C.POINTER(C.c_double))
        bg.regular_z = 1 if grid.regular_z else 0
        bc = _lib.bz_constants(c.gravitational_acceleration, dry_air_gas_constant(c), vapor_gas_constant(c),
                               c.dry_air_heat_capacity, c.vapor_heat_capacity)
        self._ref_arrays = [np.ascontiguousarray(a, dtype=np.float64) for a in (ref.density, ref.pressure, ref.temperature)]
        br = _lib.bz_reference_state(ref.surface_pressure, ref.potential_temperature, ref.standard_pressure,
                                     *[a.ctypes.data_as(C.POINTER(C.c_double)) for a in self._ref_arrays])
        self._ctx = C.c_void_p()
        rc = lib.bz_create_slab(C.byref(self._ctx), C.byref(bg), C.byref(bc), C.byref(br), advection.order, world, rank)
        if rc != 0:
            raise _lib.BreezeHIPError(f"bz_create_slab failed with code {rc}")
        self._check(lib.bz_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "bz_set_stream")
        self._state, self._U0, self._G = self._make_state(), self._make_prog(self.U0), self._make_prog(self.G)
        if getattr(self.decomp, "pack", None) is None and hasattr(self.decomp, "_pack"):
            self.decomp.pack = self._device_pack
            import os
            if not os.environ.get("BZ_NO_ROW_PACK"):
                self.decomp.rows = self._device_rows
        self.transport = transport
        if transport != "torch":
            self._attach_library_transport(transport, group)
        self.set(θ=ref.potential_temperature)

    def _attach_library_transport(self, transport, group):
        attach_library_transport(self, transport, group)

    def comm_info(self):
        """(transport name, bytes this rank has sent, number of exchanges) of the library-owned communicator."""
        name, nbytes, nex = C.c_char_p(), C.c_int64(), C.c_int32()
        self._check(self._lib.bz_comm_info(self._ctx, C.byref(name), C.byref(nbytes), C.byref(nex)), "bz_comm_info")
        return name.value.decode(), nbytes.value, nex.value

    # plumbing shared with AtmosphereModel -------------------------------------------------
    def _check(self, rc, what):
        _lib.check(self._lib, self._ctx, rc, what)

    def prognostic_fields(self):
        return {"ρu": self.momentum["ρu"], "ρv": self.momentum["ρv"], "ρw": self.momentum["ρw"],
                "ρθ": self.potential_temperature_density, "ρq": self.moisture_density}

    def _make_state(self):
        s = _lib.bz_state()
        s.rho_u, s.rho_v, s.rho_w = (self.momentum[k].ptr() for k in ("ρu", "ρv", "ρw"))
        s.rho_theta, s.rho_q = self.potential_temperature_density.ptr(), self.moisture_density.ptr()
        s.u, s.v, s.w = (self.velocities[k].ptr() for k in ("u", "v", "w"))
        s.theta, s.q, s.T = self.potential_temperature.ptr(), self.specific_moisture.ptr(), self.temperature.ptr()
        s.phi = self.pressure_anomaly.ptr()
        return s

    @staticmethod
    def _make_prog(d):
        p = _lib.bz_prognostic()
        p.rho_u, p.rho_v, p.rho_w, p.rho_theta, p.rho_q = (d[k].ptr() for k in ("ρu", "ρv", "ρw", "ρθ", "ρq"))
        return p

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.bz_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def synchronize(self):
        self._check(self._lib.bz_sync(self._ctx), "bz_sync")

    def profile_enable(self, on=True):
        self._check(self._lib.bz_profile_enable(self._ctx, 1 if on else 0), "bz_profile_enable")

    def profile_reset(self):
        self._check(self._lib.bz_profile_reset(self._ctx), "bz_profile_reset")

    def profile(self):
        out = {}
        for i in range(self._lib.bz_profile_count(self._ctx)):
            name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
            self._check(self._lib.bz_profile_get(self._ctx, i, C.byref(name), C.byref(ms), C.byref(n)), "bz_profile_get")
            out[name.value.decode()] = (ms.value, n.value)
        return out

    # SlabStepper backend ---------------------------------------------------------------------
    def momentum_fields(self):
        return [self.momentum[k].parent for k in ("ρu", "ρv", "ρw")]

    def tendency_halo_fields(self):
        """What the tendency kernels read across the slab edge: the advecting mass fluxes, the advected velocities and the two
        advected scalars.  T (buoyancy: own column only), rho theta / rho q (RK update: own cell) and the pressure anomaly
        (diagnostic) are never read in y-halo rows, so they do not travel."""
        return ([self.momentum[k].parent for k in ("ρu", "ρv", "ρw")] +
                [self.velocities[k].parent for k in ("u", "v", "w")] +
                [self.potential_temperature.parent, self.specific_moisture.parent])

    fused_rk = True

    def predictor_fields(self):
        return [self.G[k].parent for k in ("ρu", "ρv", "ρw")]

    def local_tend_rk(self, dt, alpha, first):
        self._check(self._lib.bz_tendencies_fused_rk(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G),
                                                     float(dt), float(alpha), 1 if first else 0), "bz_tendencies_fused_rk")

    def local_rk3(self, dt, alpha, first):
        self._check(self._lib.bz_ssp_rk3_substep_fused(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G),
                                                       float(dt), float(alpha), 1 if first else 0), "bz_ssp_rk3_substep_fused")

    def local_source(self, dt, predictor=False):
        import torch
        g = self.grid
        rhs = torch.empty((g.Nz, g.Ny, g.Nx), dtype=torch.float64, device=self.device)
        if predictor:
            self._check(self._lib.bz_poisson_source_term_from(self._ctx, C.byref(self._state), C.byref(self._G), float(dt),
                                                              C.c_void_p(rhs.data_ptr())), "bz_poisson_source_term_from")
        else:
            self._check(self._lib.bz_poisson_source_term(self._ctx, C.byref(self._state), float(dt), C.c_void_p(rhs.data_ptr())),
                        "bz_poisson_source_term")
        return rhs

    def local_spectral_solve(self, S, scale=1.0):
        assert S.is_contiguous()
        self._check(self._lib.bz_spectral_tridiagonal_solve(self._ctx, C.c_void_p(S.data_ptr()), float(scale)),
                    "bz_spectral_tridiagonal_solve")

    def fft_x_forward(self, rhs):
        import torch
        d = self.decomp
        assert rhs.is_contiguous()
        out = torch.empty((d.Nz, d.Ny, d.nxh), dtype=torch.complex128, device=rhs.device)
        self._check(self._lib.bz_slab_transform(self._ctx, 0, C.c_void_p(rhs.data_ptr()), C.c_void_p(out.data_ptr()), 0),
                    "bz_slab_transform")
        self._keep_fft = rhs
        return out

    def fft_y(self, S, inverse):
        assert S.is_contiguous()
        self._check(self._lib.bz_slab_transform(self._ctx, 2 if inverse else 1, C.c_void_p(S.data_ptr()), C.c_void_p(S.data_ptr()), 0),
                    "bz_slab_transform")
        return S

    def fft_x_inverse(self, R):
        import torch
        d = self.decomp
        assert R.is_contiguous() and R.shape[2] >= d.nxh
        out = torch.empty((d.Nz, d.Ny, d.Nx), dtype=torch.float64, device=R.device)
        self._check(self._lib.bz_slab_transform(self._ctx, 3, C.c_void_p(R.data_ptr()), C.c_void_p(out.data_ptr()), int(R.shape[2])),
                    "bz_slab_transform")
        self._keep_fft = R
        return out

    def _device_rows(self, fields, row0, nrows, buffer, unpack):
        n = len(fields)
        ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in fields])
        levels = (C.c_int32 * n)(*[f.shape[0] for f in fields])
        self._check(self._lib.bz_pack_rows(self._ctx, ptrs, levels, n, int(row0), int(nrows), C.c_void_p(buffer.data_ptr()),
                                           1 if unpack else 0), "bz_pack_rows")

    def _device_pack(self, T, c0, B, valid):
        import torch
        assert T.is_contiguous() and T.dtype == torch.complex128
        Nz, A, ld = T.shape
        out = torch.empty((Nz, B, A), dtype=torch.complex128, device=T.device)
        self._check(self._lib.bz_pack_transpose(self._ctx, C.c_void_p(T.data_ptr()), C.c_void_p(out.data_ptr()), Nz, A, ld,
                                                int(c0), int(B), int(valid)), "bz_pack_transpose")
        self._keep_pack = T          # the source must outlive the enqueued kernel
        return out

    def local_project_diagnose(self, phi, below, dt, predictor=False):
        assert phi.is_contiguous()
        below = below.contiguous()
        self._keep = (phi, below)        # keep alive until the stream has consumed them
        if predictor:
            self._check(self._lib.bz_project_and_diagnose_from(self._ctx, C.byref(self._state), C.byref(self._G),
                                                               C.c_void_p(phi.data_ptr()), C.c_void_p(below.data_ptr()),
                                                               float(dt)), "bz_project_and_diagnose_from")
        else:
            self._check(self._lib.bz_project_and_diagnose(self._ctx, C.byref(self._state), C.c_void_p(phi.data_ptr()),
                                                          C.c_void_p(below.data_ptr()), float(dt)), "bz_project_and_diagnose")

    def local_tendencies(self):
        self._check(self._lib.bz_compute_tendencies(self._ctx, C.byref(self._state), C.byref(self._G)), "bz_compute_tendencies")

    # model API ---------------------------------------------------------------------------------
    def set(self, enforce_mass_conservation=True, **kw):
        """set!(model; θ, u, v, w, qᵗ, ρu, ...) with functions of the GLOBAL coordinates."""
        import torch
        from .model import _ALIASES
        g, ref = self.grid, self.reference_state
        Hz, Nz = g.Hz, g.Nz
        ρc = torch.from_numpy(ref.density[Hz:Hz + Nz].copy()).to(self.device)[:, None, None]
        ρf = torch.from_numpy(0.5 * (ref.density[Hz - 1:Hz + Nz] + ref.density[Hz:Hz + Nz + 1])).to(self.device)[:, None, None]
        for name, value in kw.items():
            key = _ALIASES.get(name)
            if key is None:
                raise ValueError(f"Cannot set! {name} in AtmosphereModel")
            if key == "θ":
                self.potential_temperature.set_interior(value)
                self.potential_temperature_density.interior.copy_(ρc * self.potential_temperature.interior)
            elif key == "ρθ":
                self.potential_temperature_density.set_interior(value)
            elif key == "q":
                self.specific_moisture.set_interior(value)
                self.moisture_density.interior.copy_(ρc * self.specific_moisture.interior)
            elif key == "ρq":
                self.moisture_density.set_interior(value)
            elif key in ("u", "v"):
                self.velocities[key].set_interior(value)
                self.momentum["ρ" + key].interior.copy_(ρc * self.velocities[key].interior)
            elif key == "w":
                self.velocities["w"].set_interior(value)
                self.momentum["ρw"].interior.copy_(ρf * self.velocities["w"].interior)
            else:
                self.momentum[key].set_interior(value)
        if self.transport != "torch":
            self._check(self._lib.bz_comm_update_state_and_project(self._ctx, C.byref(self._state), C.byref(self._G), 1.0,
                                                                   1 if enforce_mass_conservation else 0),
                        "bz_comm_update_state_and_project")
            return
        # update_state!(compute_tendencies=false): x/z halos + diagnostics locally, y halos from the neighbours
        self._check(self._lib.bz_update_state(self._ctx, C.byref(self._state), C.byref(self._G), 0), "bz_update_state")
        self.decomp.exchange_y_halos(self.tendency_halo_fields())
        if enforce_mass_conservation:
            self.pressure_projection(1.0)

    def time_step(self, Δt):
        if self.transport != "torch":         # the library owns the exchanges: one call per step
            self._check(self._lib.bz_time_step_anelastic(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G),
                                                         float(Δt)), "bz_time_step_anelastic")
            self.clock.time += Δt
            self.clock.iteration += 1
            return
        if self.clock.iteration == 0 and not self.fused_rk:          # maybe_prepare_first_time_step!
            self.local_tendencies()
        SlabStepper.time_step(self, Δt)
        self.clock.time += Δt
        self.clock.iteration += 1

    def time_steps(self, Δt, n, diagnose_last=True):
        """n steps in one C call through the library transport (bz_time_steps_anelastic on a slab context: steps whose diagnostics nobody
        reads end with the momentum-only projection and exchange five halo fields instead of ten); the torch transport steps one by one."""
        n = int(n)
        if n <= 0:
            return
        if self.transport == "torch":
            for _ in range(n):
                self.time_step(Δt)
            return
        self._check(self._lib.bz_time_steps_anelastic(self._ctx, C.byref(self._state), C.byref(self._U0), C.byref(self._G), float(Δt), n,
                                                      1 if diagnose_last else 0), "bz_time_steps_anelastic")
        self.clock.time += n * Δt
        self.clock.iteration += n
