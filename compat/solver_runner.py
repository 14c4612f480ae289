"""Reference name `solver_runner` (solver_runner.py:12-54) -> vince_amd.solver_runner."""
from vince_amd.solver_runner import main  # noqa: F401

if __name__ == "__main__":
    main()
