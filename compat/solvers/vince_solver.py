"""Reference name `solvers.vince_solver` (solvers/vince_solver.py:33-706)."""
from vince_amd.solvers.vince_solver import VinceSolver, stack_dicts_in_list  # noqa: F401
