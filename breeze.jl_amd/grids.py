"""Host-side mirror of Oceananigans' RectilinearGrid for the hot path (regular x, y; regular or
stretched z).  Layout conventions follow include/breeze_hip.h: parent arrays are halo-inclusive,
i fastest; in Python they are exposed as (z, y, x)-shaped C-contiguous arrays."""
import numpy as np

Periodic, Bounded, Flat = "Periodic", "Bounded", "Flat"
_TOPO_CODE = {Periodic: 0, Bounded: 1, Flat: 2}


class Center:
    pass


class Face:
    pass


class RectilinearGrid:
    """RectilinearGrid(size=(Nx,Ny,Nz), x=(x0,x1), y=(y0,y1), z=(z0,z1) | z=faces,
    topology=(Periodic, Periodic, Bounded), halo=(3,3,3))

    Only the non-Flat dimensions are listed in `size`/`halo`, as in Oceananigans.
    """

    def __init__(self, size, x=None, y=None, z=None, topology=(Periodic, Periodic, Bounded), halo=None,
                 float_type=np.float64):
        if float_type in (np.float64, float):
            self.float_type, self.ftype = np.float64, 8
        elif float_type is np.float32:
            self.float_type, self.ftype = np.float32, 4          # eltype(grid) = Float32: the Float32 build of the library
        else:
            raise ValueError("float_type must be numpy.float64 or numpy.float32")
        for t in topology:
            if t not in _TOPO_CODE:
                raise ValueError(f"unknown topology {t!r}")
        self.topology = tuple(topology)
        size = (size,) if np.isscalar(size) else tuple(size)
        nonflat = [d for d in range(3) if topology[d] != Flat]
        if len(size) != len(nonflat):
            raise ValueError(f"size={size} must have one entry per non-Flat dimension ({len(nonflat)})")
        halo = (3,) * len(nonflat) if halo is None else ((halo,) * len(nonflat) if np.isscalar(halo) else tuple(halo))
        if len(halo) != len(nonflat):
            raise ValueError("halo must have one entry per non-Flat dimension")
        N, H = [1, 1, 1], [0, 0, 0]
        for n, h, d in zip(size, halo, nonflat):
            N[d], H[d] = int(n), int(h)
        self.Nx, self.Ny, self.Nz = N
        self.Hx, self.Hy, self.Hz = H
        for d in nonflat:       # Oceananigans' validate_halo: a halo cannot be wider than the domain it wraps
            if N[d] < H[d]:
                raise ValueError(f"halo ({H[d]}) must be ≤ size ({N[d]}) in dimension {'xyz'[d]}")
        ext = {0: x, 1: y, 2: z}
        for d in nonflat:
            if ext[d] is None:
                raise ValueError(f"extent of dimension {'xyz'[d]} is required")

        def regular(e, n, flat):
            if flat:
                return 1.0, np.zeros(1), np.zeros(1)
            a, b = e
            if not b > a:
                raise ValueError("extent must be increasing")
            d = (b - a) / n
            return d, a + d * np.arange(n), a + d * (np.arange(n) + 0.5)

        self.Δx, self.xᶠ, self.xᶜ = regular(x, self.Nx, topology[0] == Flat)
        self.Δy, self.yᶠ, self.yᶜ = regular(y, self.Ny, topology[1] == Flat)
        if topology[2] == Flat:
            raise ValueError("Flat z is not supported")
        Nz = self.Nz
        self.regular_z = isinstance(z, (tuple, list)) and len(z) == 2
        if self.regular_z:
            Δz = (z[1] - z[0]) / Nz
            self.zᶠ = z[0] + Δz * np.arange(Nz + 1)
            self.zᶠ[-1] = z[1]
            self.zᶜ = z[0] + Δz * (np.arange(Nz) + 0.5)
            self.Δz = Δz
        else:
            self.zᶠ = np.ascontiguousarray(z, dtype=np.float64)
            if self.zᶠ.shape != (Nz + 1,):
                raise ValueError("z must be a 2-tuple or an array of Nz+1 faces")
            self.zᶜ = 0.5 * (self.zᶠ[:-1] + self.zᶠ[1:])
            self.Δz = None
        self.Lx = 0.0 if x is None else x[1] - x[0]
        self.Ly = 0.0 if y is None else y[1] - y[0]
        self.Lz = self.zᶠ[-1] - self.zᶠ[0]

    # parent-array geometry
    @property
    def Sx(self):
        return self.Nx + 2 * self.Hx

    @property
    def Sy(self):
        return self.Ny + 2 * self.Hy

    def parent_shape(self, zface=False):
        return (self.Nz + 2 * self.Hz + (1 if zface else 0), self.Sy, self.Sx)

    def interior_slices(self, zface=False):
        return (slice(self.Hz, self.Hz + self.Nz + (1 if zface else 0)),
                slice(self.Hy, self.Hy + self.Ny), slice(self.Hx, self.Hx + self.Nx))

    def topology_codes(self):
        return tuple(_TOPO_CODE[t] for t in self.topology)

    def nodes(self, loc):
        """Broadcastable (x, y, z) node arrays shaped for (z, y, x) storage; loc like (Face, Center, Center)."""
        x = self.xᶠ if loc[0] is Face else self.xᶜ
        y = self.yᶠ if loc[1] is Face else self.yᶜ
        z = self.zᶠ if loc[2] is Face else self.zᶜ
        return x[None, None, :], y[None, :, None], z[:, None, None]

    def __repr__(self):
        return (f"RectilinearGrid({self.Nx}×{self.Ny}×{self.Nz}, halo=({self.Hx},{self.Hy},{self.Hz}), "
                f"topology={self.topology})")
