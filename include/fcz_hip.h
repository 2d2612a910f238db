/* fcz_hip.h -- C-ABI of the MI355X-native Foldcomp codec hot path (libfcz_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types, integer status codes,
 * no exceptions. The reference (steineggerlab/foldcomp) has no FFI for this path -- it is entered
 * through the C++ class `Foldcomp`. Each entry point below names the reference interface it replaces
 * (file:line under the reference tree); INTEGRATION.md shows the binding a maintainer would add.
 *
 *   reference                                              this ABI
 *   ---------------------------------------------------    ------------------------------------------
 *   Foldcomp::compress(span<AtomCoordinate>)               fcz_compress_batch / fcz_compress_batch_dev
 *     src/foldcomp.cpp:562 (+preprocess :450)
 *   Foldcomp::writeStream(ostream&)  src/foldcomp.cpp:1038   (the FCZ bytes are what compress returns)
 *   Foldcomp::getSize()              src/foldcomp.cpp:1190   fcz_compress_sizes / fcz_compress_sizes_dev
 *   Foldcomp::read(istream&)         src/foldcomp.cpp:904    fcz_decompress_sizes (+ header parse)
 *   Foldcomp::decompress(vector<AtomCoordinate>&)          fcz_decompress_batch / fcz_decompress_batch_dev
 *     src/foldcomp.cpp:779
 *   Foldcomp::checkValidity()        src/foldcomp.cpp:1492   fcz_check
 *
 * Batch-first: one call handles C independent chains ("one wavefront per chain" on the device).
 * Data layout is structure-of-arrays; all offsets are element indices, not bytes, unless noted.
 *
 * Threading: an fcz_ctx owns one HIP stream + device scratch on one GPU; use one ctx per host thread.
 */
#ifndef FCZ_HIP_H
#define FCZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------ */
enum fcz_status {
    FCZ_OK = 0,
    FCZ_E_INVALID_ARG = -1,
    FCZ_E_NO_DEVICE = -2,       /* no HIP device / HIP runtime error; the library has NO CPU fallback */
    FCZ_E_HIP = -3,
    FCZ_E_BAD_MAGIC = -4,       /* Foldcomp::read returns -1 (src/foldcomp.cpp:911-915) */
    FCZ_E_TRUNCATED = -5,       /* FCZ entry shorter than its header promises */
    FCZ_E_RESIDUE = -6,         /* residue code the reference cannot process (AAS.at throws, src/sidechain.cpp:177) */
    FCZ_E_TOO_SHORT = -7,       /* chain with < 2 residues: undefined in the reference */
    FCZ_E_NOMEM = -8,
    FCZ_E_NONFINITE = -9        /* compress: a NaN or an infinity in a coordinate of a named atom (atom_code != 255) or in a CA
                                 * B-factor of the chain. The reference's readers can produce them (mmCIF `?` / `.` -> NaN,
                                 * lib/gemmi/numb.hpp:19-40; "nan" in a PDB column, lib/gemmi/pdb.hpp:49-54) and its compressor
                                 * then writes NaN quantiser parameters carrying the input's sign and payload
                                 * (src/discretizer.cpp:22-33): a record that decodes to no structure. Refused here, at every
                                 * level (C-ABI status, `[Error]` line of the hosts, foldcomp.error in Python). */
};

/* ---- vocabulary --------------------------------------------------------------------------- */
/* Residue codes: the reference's 5-bit codes (src/utility.h:133-206):
 *   ALA0 ARG1 ASN2 ASP3 CYS4 GLN5 GLU6 GLY7 HIS8 ILE9 LEU10 LYS11 MET12 PHE13 PRO14 SER15 THR16
 *   TRP17 TYR18 VAL19 ASX20 GLX21 STP22 UNK23.
 * Atom codes: 0=N 1=CA 2=C 3=O 4=CB ... 35=OH, 36=OXT, 255=any other name (see fcz_atom_code_name).
 */
#define FCZ_ATOM_CODE_OXT 36
#define FCZ_ATOM_CODE_OTHER 255

/* Compress input: C chains, R residues in total, M atoms in total. Caller-owned.
 * Replaces the tcb::span<AtomCoordinate> handed to Foldcomp::compress (src/main.cpp:485-488).
 * Preconditions (the reference's input domain): every residue holds exactly one atom
 * named N, one CA and one C, in that order (the reference counts residues on the flat list of every N / CA / C
 * atom, src/atom_coordinate.cpp:135-143, src/foldcomp.cpp:462: with one of each per residue that list is this
 * batch's residues); residues of a chain are gap-free; residue names are the 20 standard ones or UNK; the
 * chain's last atom belongs to its last residue by name as well (its residue name is header.lastResidue,
 * src/foldcomp.cpp:469). The structure ingest and the hosts of this repository refuse what does not comply. */
typedef struct fcz_chain_batch {
    uint32_t n_chains;              /* C */
    uint32_t n_residues;            /* R */
    uint32_t n_atoms;               /* M */
    int32_t  anchor_threshold;      /* `-b`, Foldcomp::anchorThreshold (default 25) */
    const uint32_t* res_off;        /* [C+1] residues of chain c = [res_off[c], res_off[c+1]) */
    const uint32_t* atom_off;       /* [R+1] atoms of residue r = [atom_off[r], atom_off[r+1]) (input order) */
    const float*    x;              /* [M] */
    const float*    y;              /* [M] */
    const float*    z;              /* [M] */
    const uint8_t*  atom_code;      /* [M] */
    const uint8_t*  res_code;       /* [R] */
    const float*    bfac_ca;        /* [R] tempFactor of the residue's CA atom (src/foldcomp.cpp:543-547) */
    const int32_t*  first_res_index;  /* [C] header.idxResidue (src/foldcomp.cpp:464) */
    const int32_t*  first_atom_index; /* [C] header.idxAtom */
    const char*     chain_id;       /* [C] */
    const char*     titles;         /* concatenated titles, no NULs */
    const uint32_t* title_off;      /* [C+1] byte offsets into titles */
} fcz_chain_batch;

/* Decompress output: SoA coordinates of all atoms of all chains in reference output order
 * (canonical atom order of src/amino_acid.h, or the `-a` order), OXT last when present.
 * Replaces the std::vector<AtomCoordinate>& filled by Foldcomp::decompress. Caller-owned. */
typedef struct fcz_atoms_out {
    float*    x;            /* [M] */
    float*    y;            /* [M] */
    float*    z;            /* [M] */
    float*    bfac_res;     /* [R] de-quantised B-factor of each residue (src/foldcomp.cpp:884-892) */
    uint8_t*  res_code;     /* [R] */
    uint8_t*  atom_code;    /* [M] optional (may be NULL) */
} fcz_atoms_out;

/* Parsed per-entry header fields a caller needs to rebuild AtomCoordinate records / PDB text. */
typedef struct fcz_entry_info {
    uint32_t n_residues;
    uint32_t n_atoms_out;       /* atoms the decompressor emits (incl. OXT) */
    uint32_t n_atoms_header;    /* header.nAtom */
    int32_t  first_res_index;
    int32_t  first_atom_index;
    uint32_t n_anchors;
    uint32_t n_sidechain_torsions;
    uint32_t title_off;         /* byte offset of the title inside the entry */
    uint32_t title_len;
    char     chain_id;
    char     first_residue;     /* one-letter */
    char     last_residue;
    uint8_t  has_oxt;
    int32_t  status;            /* FCZ_OK or the reason this entry is skipped */
} fcz_entry_info;

typedef struct fcz_ctx fcz_ctx;

/* ---- lifecycle ---------------------------------------------------------------------------- */
int  fcz_ctx_create(int device, fcz_ctx** out);
void fcz_ctx_destroy(fcz_ctx* ctx);
/* hipStream_t of the ctx as an opaque pointer (all *_dev entry points enqueue on it). */
void* fcz_ctx_stream(fcz_ctx* ctx);
int  fcz_ctx_synchronize(fcz_ctx* ctx);

/* Numerics of the decompress side (the compress side always writes the reference's bytes).
 *   FCZ_NUMERICS_EXACT (default): decompressed float32 coordinates are bit-identical to Foldcomp::decompress built with
 *     g++ -O3 on x86-64 / glibc 2.35 -- the reference's evaluation order, its double promotions and glibc's sinf/cosf are
 *     reproduced operation by operation (fcz_math.h).
 *   FCZ_NUMERICS_FAST: the same algorithm (Foldcomp::decompress src/foldcomp.cpp:779-900: forward NeRF, reverse NeRF from the
 *     next anchor, weighted average, side chains) in plain float arithmetic with FMA and hardware rsq, the backbone as a
 *     parallel composition of per-residue rigid transforms (fcz_backbone_fast.h). Coordinates differ from the exact path by
 *     float rounding only (< 1e-3 A, typically 1e-4 A; the reference's RMSD pins of build.sh:35,37 hold unchanged), which is
 *     what `foldcomp check` / the RMSD tolerance of the reference's own tests ask of a decoder. */
/* Host-side helpers for callers that feed the host-pointer entry points from their own threads (one ctx per thread and GPU):
 * the number of visible devices, and page-locked host memory (hipHostMalloc): copies from / to such buffers are true
 * asynchronous DMA on the ctx stream, so two ctxs on one GPU overlap one batch's transfers with the other's kernels. */
int   fcz_device_count(void);
void* fcz_pinned_alloc(size_t bytes);
void  fcz_pinned_free(void* p);

enum fcz_numerics { FCZ_NUMERICS_EXACT = 0, FCZ_NUMERICS_FAST = 1 };
int  fcz_ctx_set_numerics(fcz_ctx* ctx, int mode);
int  fcz_ctx_get_numerics(fcz_ctx* ctx);
const char* fcz_status_string(int status);
const char* fcz_atom_code_name(int atom_code);     /* "N", "CA", ... "OXT"; NULL if out of range */
int  fcz_atom_code_from_name(const char* name);    /* 255 for unknown names */
int  fcz_res_code_from_name(const char* three_letter); /* -1 for names the reference rejects */
const char* fcz_res_code_name(int res_code);
int  fcz_res_code_natoms(int res_code);            /* atoms emitted per residue (3 for UNK) */
/* canonical atom code of output position j of a residue (alt_order: the `-a` order) */
int  fcz_res_code_atom(int res_code, int j, int alt_order);

/* ---- compress ----------------------------------------------------------------------------- */
/* Exact FCZ size of every chain -> exclusive prefix in out_off[C+1] (bytes). Host arrays.
 * Pure host integer work (Foldcomp::getSize, src/foldcomp.cpp:1190-1214). */
int fcz_compress_sizes(const fcz_chain_batch* in, uint64_t* out_off);

/* Host-pointer convenience: copies the batch to the GPU, runs the kernels, copies FCZ bytes back
 * into out[out_off[c] .. out_off[c+1]). status[c] (may be NULL) receives a per-chain fcz_status.
 * Limit: the records of one batch total less than 2^39 bytes (512 GB; more than any device holds) -- a larger out_off[C] is
 * refused with FCZ_E_INVALID_ARG (the kernels hand side-chain byte addresses on in 39 bits). */
int fcz_compress_batch(fcz_ctx* ctx, const fcz_chain_batch* in, const uint64_t* out_off,
                       uint8_t* out, int32_t* status);

/* Device-resident variant: every pointer inside `in`, plus out_off/out/status, is a device pointer
 * (titles included). Work is enqueued on the ctx stream; no host synchronisation. */
int fcz_compress_sizes_dev(fcz_ctx* ctx, const fcz_chain_batch* in, uint64_t* out_off_dev);
int fcz_compress_batch_dev(fcz_ctx* ctx, const fcz_chain_batch* in, const uint64_t* out_off_dev,
                           uint8_t* out_dev, int32_t* status_dev);

/* Pre-quantisation backbone angles of a host batch -- what `get_data()` of the reference's Python module
 * returns for PDB input (foldcomp/foldcomp.cxx:633-662). angles_out = [6][R] floats, order phi, psi, omega,
 * n_ca_c, ca_c_n, c_n_ca; entry r0+k (k < n-1) belongs to packed word k; n_ca_c[r0+n-1] holds the first
 * residue's N-CA-C angle, which the FCZ format never stores (src/foldcomp.cpp:497). */
int fcz_compress_angles(fcz_ctx* ctx, const fcz_chain_batch* in, float* angles_out);

/* ---- decompress --------------------------------------------------------------------------- */
/* Parse the n entries blob[off[i] .. off[i+1]) (trailing bytes such as the MMseqs '\0' are ignored,
 * like Foldcomp::read). Fills info[n] and the exclusive prefixes res_off[n+1] / atom_off[n+1]
 * that size the output arrays. Entries that fail validation get info[i].status != FCZ_OK and
 * contribute zero residues/atoms. Host arrays. */
int fcz_decompress_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n,
                         fcz_entry_info* info, uint32_t* res_off, uint32_t* atom_off);

int fcz_decompress_batch(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n,
                         const uint32_t* res_off, const uint32_t* atom_off, int alt_order,
                         const fcz_atoms_out* out);

/* Device-resident variants. fcz_decompress_sizes_dev fills res_off_dev/atom_off_dev (n+1 each)
 * and returns the totals through pinned host words after a stream sync (the only sync on this path:
 * the caller needs R and M to allocate outputs). */
int fcz_decompress_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                             uint32_t* res_off_dev, uint32_t* atom_off_dev,
                             uint32_t* total_res, uint32_t* total_atoms);
int fcz_decompress_batch_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                             const uint32_t* res_off_dev, const uint32_t* atom_off_dev, int alt_order,
                             const fcz_atoms_out* out_dev);

/* ---- PDB text -------------------------------------------------------------------------------- */
/* The text `foldcomp decompress` writes for every entry (writeAtomCoordinatesToPDB, src/atom_coordinate.cpp:220-291,
 * called from src/main.cpp:625-637 and foldcomp/foldcomp.cxx:224-250): TITLE records wrapped at 70 columns, one
 * 81-byte ATOM record per atom (numbers by fast_ftoa<1000,3> / <100,2>, :185-218; printf widens a column that
 * overflows), one TER record. Inputs: the same FCZ entries plus the arrays fcz_decompress_batch_dev filled
 * (x, y, z, bfac_res, res_code required); everything else comes from the entry headers.
 * fcz_pdb_sizes_dev: exclusive prefix of the exact text sizes in text_off_dev[n+1] (bytes; skipped entries: 0). */
int fcz_pdb_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                      const uint32_t* res_off_dev, const uint32_t* atom_off_dev, const fcz_atoms_out* atoms_dev,
                      uint64_t* text_off_dev);
int fcz_pdb_format_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                       const uint32_t* res_off_dev, const uint32_t* atom_off_dev, const fcz_atoms_out* atoms_dev,
                       int alt_order, const uint64_t* text_off_dev, uint8_t* text_dev);
/* Host-pointer convenience: FCZ entries -> PDB text with every stage on the device. begin() fills text_off[n+1]
 * (host) and status[n] (may be NULL) and keeps the text in the ctx; fetch() copies text_off[n] bytes out.
 * alt_order here is a set of flags: FCZ_PDB_ALT_ORDER (--use-alt-order), FCZ_PDB_NUL_TERMINATED (every entry that decodes
 * is followed by one NUL, counted in text_off: the record `foldcomp decompress` appends to a database, src/main.cpp:656-664,
 * so that a job's records are one contiguous byte range of the data file). */
#define FCZ_PDB_ALT_ORDER      1
#define FCZ_PDB_NUL_TERMINATED 0x100
int fcz_decompress_pdb_begin(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int alt_order,
                             uint64_t* text_off, int32_t* status);
int fcz_decompress_pdb_fetch(fcz_ctx* ctx, uint8_t* text_out);
/* The sizes half of begin(): the entries are decoded on the device and text_off[n+1] / status[n] filled exactly as begin() fills
 * them (a line's width depends on the decoded numbers: printf widens a column that overflows), but no text is formatted or kept.
 * What a rank of a sharded `decompress --db` run calls over its range BEFORE it writes, so that the ranks exchange their record
 * and byte counts first and every record is appended once at its final offset -- writer_append, src/database_writer.cpp:36-58,
 * called once per record from src/main.cpp:656-664. */
int fcz_decompress_pdb_sizes(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int alt_order,
                             uint64_t* text_off, int32_t* status);

/* ---- structure ingest: PDB / mmCIF text -> fcz_chain_batch on the device ----------------------------- */
/* What the reference's driver does to every input file before Foldcomp::compress (src/main.cpp:455-508): StructureReader
 * (src/structure_reader.cpp:31-61; the fixed-column ATOM / HETATM record as foldcomp/foldcomp.cxx:259-278 reads it),
 * removeAlternativePosition (src/atom_coordinate.cpp:362-370), identifyChains (:469-497), identifyDiscontinousResInd (:506-530),
 * then per fragment the residue split, residue codes and CA B-factors of Foldcomp::preprocess (src/foldcomp.cpp:450-559).
 * Input: the bytes of n_files PDB files back to back (file i = text[file_off[i] .. file_off[i+1])), their base names
 * (names[name_off[i] .. name_off[i+1]); the first stem_len[i] characters are the stem that names the records and replaces an
 * absent title, src/main.cpp:465). Output: one fcz_chain_batch in HBM holding every fragment the codec can take, file order
 * then fragment order, ready for fcz_compress_sizes_dev / fcz_compress_batch_dev, plus per chain the file it came from and
 * how the reference names it:
 *   chain_meta = chain id | fragment ordinal << 8 | FCZ_INGEST_MULTI_CHAIN (the file holds several chains: the id is appended
 *                to the name) | FCZ_INGEST_MULTI_FRAG (the chain has gaps: "_<ordinal>" is appended).
 * The reading rules are those of the reference's reader (gemmi 0.5.1 read_pdb, lib/gemmi/pdb.hpp:262-365): record names on
 * four letters case-insensitively, END stops the reading, B-factor 20 on a line that ends before column 65, title = last
 * HEADER id code else the TITLE texts concatenated; MODEL n / atoms / ENDMDL groups under rising plain numbers (ensembles) are read in
 * file order with a new chain at every group, as the reader keeps them.
 * file_status[i]: FCZ_OK, FCZ_INGEST_NO_ATOMS, or FCZ_INGEST_HOST_*: something this path does not decide the way that reader
 * would on its own (a number field the exact fixed-point rule does not cover, a blank or hybrid-36 number, a two-character chain
 * name, an ATOM record shorter than its coordinates, an ANISOU record that does not stand directly behind its atom, models not numbered upwards or atoms outside them, a NUL byte, a residue whose (number,
 * insertion code) does not grow inside its run of one chain name -- the reader regroups such lines --, a title beyond 512 bytes,
 * more than 32 fragments) -- nothing of that file is in the batch and the caller's own reader has to take it (the hosts of this
 * repository restate every rule: foldcomp_amd/structure.py parse_pdb_gemmi, host/foldcomp_hip.cpp parse_pdb_gemmi). refused[2k], refused[2k+1] = file, chain_meta | reason << 24 of the
 * fragments that were left out (residue name the codec does not know, residue without N, CA, C in order or with a second one
 * of them, a last atom that carries another residue name than its residue, chain beyond the header's counts,
 * --skip-discontinuous). Gzipped files: fcz_ingest_gz_begin below (inflated on the device in front of this stage).
 * mmCIF text (round 4, k_ingest_parse_cif + k_ingest_rows_cif): a file that opens with data_ (gemmi::coor_format_from_content, lib/gemmi/mmread.hpp:31-47)
 * is read by gemmi's mmCIF rules (cif.hpp:37-148 grammar, mmcif.hpp:560-680 make_structure: the _atom_site loop's 23 columns by any
 * case, chain = auth_asym_id else label_asym_id, residue = auth_seq_id + comp id, atom name = auth_atom_id else label_atom_id,
 * title = _entry.id) when it has the shape every predicted-structure file has: one block, one item per line (or a tag line and its
 * value / text field on the following lines), loops of whole-line rows, _atom_site rows of one line each without quoted values
 * other than the atom name ("O5'"), chain names of up to four characters, integer residue numbers with an optional one-character
 * insertion code, models one after the other under rising plain numbers (1, 2, 3 ...), residues rising by (number, insertion code) inside a chain run, coordinates as plain decimals of at
 * most 15 digits (round 6: the PDB archive's shape beside AFDB's). Every other mmCIF file comes back as FCZ_INGEST_HOST_FIELD exactly
 * as a PDB file outside the fixed layout does (save_ frames, several blocks, comments after values, other quoted values, longer chain
 * names, a model that comes back or is not a plain number, '?' coordinates, duplicate tags, a _cell angle that is not plainly non-zero, lines beyond 255 characters ...). */
enum fcz_ingest_status { FCZ_INGEST_HOST_FIELD = 1, FCZ_INGEST_HOST_TITLE = 2, FCZ_INGEST_HOST_FRAGS = 3, FCZ_INGEST_NO_ATOMS = 4 };
enum fcz_ingest_reason { FCZ_INGEST_REF_RESNAME = 1, FCZ_INGEST_REF_BACKBONE = 2, FCZ_INGEST_REF_TOO_LONG = 3, FCZ_INGEST_REF_SKIP_DISC = 4,
                         FCZ_INGEST_REF_BACKBONE_TWICE = 5, FCZ_INGEST_REF_LAST_NAME = 6 };
#define FCZ_INGEST_MULTI_CHAIN (1u << 16)
#define FCZ_INGEST_MULTI_FRAG  (1u << 17)
#define FCZ_INGEST_SKIP_DISCONTINUOUS 1   /* flags: --skip-discontinuous (src/main.cpp:476-480) */
typedef struct fcz_ingest_result {        /* device pointers owned by the ctx, valid until its next ingest call */
    fcz_chain_batch batch;
    const uint32_t* chain_file;           /* [C] */
    const uint32_t* chain_meta;           /* [C] */
    const int32_t*  file_status;          /* [F] */
    const uint32_t* refused;              /* [2 * n_refused], in no particular order */
    uint32_t n_files, n_refused;
    const uint32_t* chain_name4;          /* [C] the chain's NAME, up to four characters packed little-endian (mmCIF auth_asym_id of large
                                           * complexes: "AA", "B2" ...; chain_meta's low byte is its first character, what the FCZ header
                                           * keeps): what the reference appends to a record's name when the file holds several chains
                                           * (src/main.cpp:489-491) */
} fcz_ingest_result;
/* Device-resident: text_dev / file_off_dev / names_dev / name_off_dev / stem_len_dev are device pointers; one stream
 * synchronisation (the totals). */
int fcz_ingest_pdb_dev(fcz_ctx* ctx, const uint8_t* text_dev, const uint64_t* file_off_dev, uint32_t n_files, uint64_t text_bytes,
                       const char* names_dev, const uint32_t* name_off_dev, const uint32_t* stem_len_dev, int anchor_threshold,
                       int flags, fcz_ingest_result* out);
/* Host-pointer convenience: copies the text in (true DMA from fcz_pinned_alloc memory) and runs the ingest; the batch stays
 * in the ctx. counts = {chains, residues, atoms, title bytes, refused fragments}. fetch() copies the batch arrays (every
 * pointer of host_batch must be caller-allocated for those counts; the struct's const is cast away) and the per-chain /
 * per-file / refusal arrays (any of them may be NULL) to the host. */
int fcz_ingest_pdb_begin(fcz_ctx* ctx, const uint8_t* text, const uint64_t* file_off, uint32_t n_files, const char* names,
                         const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5]);
int fcz_ingest_pdb_fetch(fcz_ctx* ctx, const fcz_chain_batch* host_batch, uint32_t* chain_file, uint32_t* chain_meta,
                         int32_t* file_status, uint32_t* refused);
/* the chains' names of the resident batch (fcz_ingest_result.chain_name4) copied to chain_name4[C]; after any *_begin call */
int fcz_ingest_chain_names_fetch(fcz_ctx* ctx, uint32_t* chain_name4);
/* Text in, FCZ records out: ingest + fcz_compress_sizes_dev + fcz_compress_batch_dev on the resident batch.
 * begin(): counts = {chains, residues, atoms, title bytes, refused fragments}, *fcz_bytes = size of the blob; fetch(): record
 * offsets out_off[C+1], per-chain status[C] and the arrays of fcz_ingest_pdb_fetch (any may be NULL), then the blob. */
int fcz_compress_pdb_begin(fcz_ctx* ctx, const uint8_t* text, const uint64_t* file_off, uint32_t n_files, const char* names,
                           const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags, uint32_t counts[5],
                           uint64_t* fcz_bytes);
int fcz_compress_pdb_fetch(fcz_ctx* ctx, uint64_t* out_off, int32_t* status, uint32_t* chain_file, uint32_t* chain_meta,
                           int32_t* file_status, uint32_t* refused, uint8_t* blob);

/* ---- inflate: gzip members -> text in HBM (round 6) -------------------------------------------------------- */
/* What the reference's reader does with zlib before it parses a `.pdb.gz` / `.cif.gz` input: gemmi::MaybeGzipped
 * (lib/gemmi/gz.hpp:105-133: gzread into a buffer sized from ISIZE, the member's last four bytes) for files, uncompressBuffer
 * (src/structure_reader.cpp:156-203: inflateInit2(15 | 32), inflate()) for database / tar entries. Here: one wavefront per member
 * (k_inflate, foldcomp_amd/csrc/fcz_inflate.h: RFC 1952 header and trailer, RFC 1951 stored / fixed / dynamic blocks, CRC-32 and ISIZE
 * verified on the device), the text written where the structure ingest reads it.
 * The device never guesses: status[i] != FCZ_INFLATE_OK means the member was NOT inflated here (its text range holds blanks) and
 * the caller's zlib has to take it -- that covers every stream zlib's inflate() rejects (invalid block type / stored lengths /
 * code-length set, over-subscribed or incomplete code, invalid literal/length or distance code, distance too far back, incorrect
 * data or length check, truncation) and streams this decoder does not read although zlib does: a header CRC (FHCRC), a header
 * beyond 252 bytes, an incomplete literal/length code, more than one member / bytes after the trailer, an ISIZE that is not the
 * text's size. FCZ_INFLATE_OK: the bytes are what zlib's inflate() returns for the member, checked by CRC-32 and ISIZE. */
enum fcz_inflate_status {
    FCZ_INFLATE_OK = 0,
    FCZ_INFLATE_HEADER = 1,     /* not a gzip member this decoder reads (magic, method, flags, header length, member < 18 bytes) */
    FCZ_INFLATE_BLOCK = 2,      /* a block header zlib rejects (or an incomplete literal/length code: zlib's to judge) */
    FCZ_INFLATE_CODE = 3,       /* invalid code / distance symbol, distance before the start of the text */
    FCZ_INFLATE_SIZE = 4,       /* the text is not text_off[i + 1] - text_off[i] bytes long */
    FCZ_INFLATE_INPUT = 5,      /* the stream runs past the member's end, or bytes are left before the trailer */
    FCZ_INFLATE_CHECK = 6       /* CRC-32 or ISIZE of the trailer do not match */
};
/* Host: exclusive prefix of the text sizes in text_off[n + 1], member i sized from its ISIZE (gemmi estimate_uncompressed_size,
 * lib/gemmi/gz.hpp:25-45). kind (may be NULL = every entry a gzip member): 1 gzip member, 0 plain bytes (size = length). A member
 * shorter than 18 bytes, or whose ISIZE exceeds what DEFLATE can expand its bytes to (1032 : 1), is sized 0 and will be refused. */
int fcz_inflate_sizes(const uint8_t* gz, const uint64_t* gz_off, uint32_t n, const uint8_t* kind, uint64_t* text_off);
/* Device-resident: every pointer a device pointer (kind_dev may be NULL); enqueued on the ctx stream, no synchronisation.
 * text_dev[text_off[i] .. text_off[i + 1]) receives member i's text (kind 0: a copy of its bytes), status_dev[i] its status. */
int fcz_inflate_dev(fcz_ctx* ctx, const uint8_t* gz_dev, const uint64_t* gz_off_dev, uint32_t n, const uint8_t* kind_dev,
                    const uint64_t* text_off_dev, uint8_t* text_dev, int32_t* status_dev);
/* Host-pointer convenience (tests, small callers): members in, text + status out. */
int fcz_inflate(fcz_ctx* ctx, const uint8_t* gz, const uint64_t* gz_off, uint32_t n, const uint8_t* kind, const uint64_t* text_off,
                uint8_t* text, int32_t* status);
/* Structure files as they lie on disk -> batch / FCZ records: fcz_ingest_pdb_begin / fcz_compress_pdb_begin with an inflate stage in
 * front. data = the files' bytes back to back, is_gz[i] != 0: file i is a gzip member (the reference decides by the name's `.gz`,
 * gemmi::MaybeGzipped::is_compressed) -- a fifth of the text's bytes cross the link. The results are fetched with
 * fcz_ingest_pdb_fetch / fcz_compress_pdb_fetch; file_status[i] == FCZ_INGEST_HOST_GZIP: the member was not inflated here (see
 * above), the caller inflates and reads the file itself. */
#define FCZ_INGEST_HOST_GZIP 5
int fcz_ingest_gz_begin(fcz_ctx* ctx, const uint8_t* data, const uint64_t* file_off, uint32_t n_files, const uint8_t* is_gz,
                        const char* names, const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags,
                        uint32_t counts[5]);
int fcz_compress_gz_begin(fcz_ctx* ctx, const uint8_t* data, const uint64_t* file_off, uint32_t n_files, const uint8_t* is_gz,
                          const char* names, const uint32_t* name_off, const uint32_t* stem_len, int anchor_threshold, int flags,
                          uint32_t counts[5], uint64_t* fcz_bytes);

/* ---- extract ---------------------------------------------------------------------------------- */
/* Foldcomp::extract (src/foldcomp.cpp:1260-1336) straight from the FCZ bytes, no reconstruction.
 *   mode 0: pLDDT (B-factor) of every residue with `digits` in 1..4 characters ("d", "dd", "dd.d", "dd.dd"; digit rules
 *           :1286-1325), joined by ',' when digits > 1;   mode 1: one-letter amino-acid sequence (digits ignored).
 * data_off[n+1] = exclusive prefix of the data sizes (entries that fail Foldcomp::read's checks: 0 bytes); the caller wraps
 * each string into the FASTA-like / TSV line (writeFASTALike / writeTSV, :1223-1237). */
int fcz_extract_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n, int mode, int digits, uint64_t* data_off);
int fcz_extract(fcz_ctx* ctx, const uint8_t* blob, const uint64_t* off, uint32_t n, int mode, int digits,
                const uint64_t* data_off, uint8_t* data_out);
int fcz_extract_sizes_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, int mode, int digits,
                          uint64_t* data_off_dev);
int fcz_extract_dev(fcz_ctx* ctx, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, int mode, int digits,
                    const uint64_t* data_off_dev, uint8_t* data_dev);

/* ---- check -------------------------------------------------------------------------------- */
/* Foldcomp::checkValidity (src/foldcomp.cpp:1492-1532) on one entry; returns the reference's
 * ValidityError value (0 = SUCCESS .. 6) or a negative fcz_status if the entry cannot be read. */
int fcz_check(const uint8_t* entry, uint64_t len);

/* ---- introspection for benchmarks --------------------------------------------------------- */
/* Accumulated device time (ms, HIP events on the ctx stream) and launch count of the named kernel
 * group since the last reset: "compress_sizes", "compress_index", "compress_angles", "compress_pack",
 * "decompress_sizes", "decompress_backbone", "decompress_index", "decompress_sidechain", "pdb_sizes", "pdb_format", "extract_sizes", "extract",
 * "ingest_parse", "ingest_parse_cif", "ingest_rows_cif", "ingest_frags", "ingest_fill", "inflate". */
int  fcz_ctx_enable_timing(fcz_ctx* ctx, int enable);
int  fcz_ctx_kernel_time(fcz_ctx* ctx, const char* name, double* ms, uint64_t* launches);
void fcz_ctx_reset_timing(fcz_ctx* ctx);

/* ---- diagnostics --------------------------------------------------------------------------- */
/* Device numerics self-test used by tests/test_device_math.py: evaluates one math primitive of the codec (mode 0..13:
 * acos->degrees, glibc sinf / cosf restatements, norm, cosine, NeRF placement, the kernels' paired and any-float sine / cosine ...) on `count` inputs generated from
 * the float bit patterns start_bits, start_bits + stride, ... and copies the float results to out_host. */
int fcz_selftest_math(fcz_ctx* ctx, int mode, uint32_t start_bits, uint32_t stride, uint32_t count, float* out_host);
/* Device copy ceiling used by bench.py beside the 8 TB/s peak: `reps` copies of `bytes` bytes (rounded down to 16) between two
 * scratch buffers by a float4-per-lane grid-stride kernel on a persistent grid (the kernel /opt/skills/guides/MI355X_MICROARCH.md
 * quotes its 6.29 TB/s "float4 copy" with), in a few launch shapes (8 / 16 / 32 blocks per CU, one or four loads in flight per lane,
 * default and non-temporal cache policy); *gb_per_s = the best (read + written bytes) / HIP-event time on the ctx stream. */
int fcz_selftest_copy(fcz_ctx* ctx, uint64_t bytes, int reps, double* gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* FCZ_HIP_H */
