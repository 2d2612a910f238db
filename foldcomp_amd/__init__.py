"""foldcomp_amd -- MI355X-native Foldcomp codec hot path (host side).

`import foldcomp_amd as foldcomp` gives the reference module's surface
(compress / decompress / get_data / open / error); see api.py.
"""
from .api import (FoldcompDatabase, FoldcompError, compress, compress_many, decompress, decompress_many, error, get_data,
                  open, split_pdb_by_chain)

__all__ = ["compress", "decompress", "get_data", "open", "error", "FoldcompError", "FoldcompDatabase", "compress_many",
           "decompress_many", "split_pdb_by_chain"]
