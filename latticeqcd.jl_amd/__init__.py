"""latticeqcd.jl_amd -- MI355X-native Dirac-solver hot path for LatticeQCD.jl (host-side mirror).

Only the hot path of SURVEY.md section 8 lives here: the HIP shared library (csrc/, C ABI in
include/lqcd_hip.h) and a thin host layer that mirrors the LatticeDiracOperators.jl / Gaugefields.jl
interface the reference calls at src/system/universe.jl:100-143, src/md/AbstractMD.jl:120-135,
src/md/standardMD.jl:82-101 and src/updates/standardHMC.jl:41-91.
"""
from . import gauge_io, lib, pegrid, rational  # noqa: F401
from .operators import *  # noqa: F401,F403
from .operators import (DdagD_operator, Dirac_operator, Fermionfields, Gaugefields, Initialize_Gaugefields,  # noqa: F401
                        Initialize_pseudofermion_fields, Lattice, calculate_Plaquette)
