"""foldcomp_amd -- MI355X-native Foldcomp codec hot path (host side)."""
