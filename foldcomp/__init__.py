"""`import foldcomp` -- the drop-in name of the MI355X codec's Python binding.

The reference's package is a C extension re-exported by foldcomp/__init__.py (module table foldcomp/foldcomp.cxx:702-709:
compress, decompress, get_data, open, error; plus util.split_pdb_by_chain). This package forwards the same names to
foldcomp_amd (host side in Python, every geometry step on the GPU through libfcz_hip.so), so a program written against the
reference runs unchanged with this repository on its path. `python -m foldcomp ...` is the command line. The reference's
`setup()` downloader (foldcomp/setup.py) is out of scope and raises.
"""
from foldcomp_amd import *  # noqa: F401,F403
from foldcomp_amd import __all__ as _names

__all__ = list(_names) + ["setup", "setup_async"]


def setup(*args, **kwargs):
    raise NotImplementedError("foldcomp.setup (database download) is not part of the MI355X codec; fetch the database files and use foldcomp.open(path)")


async def setup_async(*args, **kwargs):
    raise NotImplementedError("foldcomp.setup_async (database download) is not part of the MI355X codec")
