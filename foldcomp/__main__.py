"""`python -m foldcomp compress|decompress|extract|check|rmsd ...` = `python -m foldcomp_amd ...`"""
import sys

from foldcomp_amd.__main__ import _main_guarded as main

sys.exit(main())
