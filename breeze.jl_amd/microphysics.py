"""Host-side mirror of Breeze.Microphysics.SaturationAdjustment (src/Microphysics/saturation_adjustment.jl:20-60) and the
solver vocabulary of Breeze.Solvers (src/Solvers.jl:55-136): warm-phase equilibrium with the secant iteration."""


class WarmPhaseEquilibrium:
    pass


class MixedPhaseEquilibrium:
    def __init__(self, *a, **k):
        raise NotImplementedError("MixedPhaseEquilibrium is not implemented in the HIP path (warm phase only)")


class SecantSolver:
    def __init__(self, reltol=0, abstol=1e-4, maxiter=20):
        if reltol != 0:
            raise NotImplementedError("SecantSolver(reltol != 0) is not implemented in the HIP path")
        self.reltol, self.abstol, self.maxiter = 0.0, float(abstol), int(maxiter)


class SaturationAdjustment:
    """SaturationAdjustment(; solver = SecantSolver(abstol=1e-4, maxiter=20), equilibrium)."""

    def __init__(self, solver=None, equilibrium=None):
        if equilibrium is None or not isinstance(equilibrium, WarmPhaseEquilibrium):
            raise NotImplementedError("the HIP path implements SaturationAdjustment(equilibrium = WarmPhaseEquilibrium())")
        self.equilibrium = equilibrium
        self.solver = solver or SecantSolver()
