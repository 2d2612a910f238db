/* fcz_host.h -- the C++ host's structure readers as a library (host/libfcz_host.so, built from host/foldcomp_hip.cpp).
 *
 * Host code, no device work: what StructureReader::loadFromBuffer + readAllAtoms give the reference's driver
 * (src/structure_reader.cpp:31-97, src/main.cpp:455-462) -- the atoms of a PDB or mmCIF file in the order gemmi 0.5.1
 * hands them on, and the title -- by the same rules (format read off the content, lib/gemmi/mmread.hpp:31-47; read_pdb,
 * lib/gemmi/pdb.hpp:262-640; the CIF grammar and make_structure_from_block, lib/gemmi/cif.hpp:37-148, mmcif.hpp:560-680).
 * The caller is `python -m foldcomp_amd` (foldcomp_amd/_hostlib.py); a maintainer of the reference has its own reader. */
#ifndef FCZ_HOST_H
#define FCZ_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct fcz_host_atoms {
    uint64_t n;                        /* atoms, before removeAlternativePosition */
    char *atom, *residue, *chain;      /* n names each, NUL-separated (names of any length) */
    uint64_t atom_bytes, residue_bytes, chain_bytes;
    int32_t *atom_index, *res_index;   /* [n] serial numbers, residue numbers (-999: none given, mmCIF) */
    float *x, *y, *z, *bfac;           /* [n] */
    char* title; uint64_t title_len;   /* HEADER id code / _entry.id, else TITLE / _struct.title; "" when the file has none */
    char error[256];                   /* why the reader fails the file (return value 1) */
} fcz_host_atoms;

/* data: the bytes of the file; gz != 0: a gzip stream, inflated first. Returns 0, or 1 when the reader fails the file
 * (out->error says why; nothing to free). The arrays of a successful call are released by fcz_host_free. */
int  fcz_host_read_structure(const uint8_t* data, uint64_t len, int gz, fcz_host_atoms* out);
void fcz_host_free(fcz_host_atoms* atoms);

#ifdef __cplusplus
}
#endif
#endif
