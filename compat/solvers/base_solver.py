"""Reference name `solvers.base_solver` (solvers/base_solver.py:20-167)."""
from vince_amd.solvers.base_solver import BaseSolver  # noqa: F401
