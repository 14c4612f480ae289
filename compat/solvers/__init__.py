"""Reference package `solvers` (solvers/__init__.py:1-12): the registry `arg_parser.solver_class` looks names up in.
Only the solver of the hot path is provided; the end-task solvers are out of scope."""
from vince_amd.solvers.vince_solver import VinceSolver

__all__ = ["VinceSolver"]
