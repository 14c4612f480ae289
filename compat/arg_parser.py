"""Reference name `arg_parser` (arg_parser.py:14-241) -> vince_amd.arg_parser.  `parse_args()` reads sys.argv like the
reference's; the class registries resolve `--solver` / `--backbone` names to the HIP classes."""
from vince_amd.arg_parser import backbone_class, build_parser, finalize, parse_args, solver_class  # noqa: F401
