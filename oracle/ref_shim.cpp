// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// A thin extern "C" driver around the *unmodified* reference sources as they lie under
// /root/reference/src (compiled in place by oracle/build_ref.sh; the result goes to oracle/_ref/).
// It exposes the reference's own per-chain entry points
//   Foldcomp::compress  (src/foldcomp.cpp:562)  + Foldcomp::writeStream (src/foldcomp.cpp:1038)
//   Foldcomp::read      (src/foldcomp.cpp:904)  + Foldcomp::decompress  (src/foldcomp.cpp:779)
// over plain arrays, so that tests can pin the C restatement (oracle/fcz_oracle.c) and the HIP path
// against the real thing, and bench.py can time the reference's CPU path ("cpu_baseline.kind":"reference").
// and, for the rows either side of the codec:
//   StructureReader::loadFromBuffer (src/structure_reader.cpp:74-98: gemmi PDB / mmCIF parser, gz inflate) followed by what
//   the compress lambda does before the codec (src/main.cpp:457-474: removeAlternativePosition, identifyChains,
//   identifyDiscontinousResInd), and the database container (src/database_writer.cpp, src/database_reader.cpp).
// Nothing of the reference is copied here: this file only *calls* it.
#include "foldcomp.h"
#include "atom_coordinate.h"
#include "amino_acid.h"
#include "structure_reader.h"
#include "database_reader.h"
#include "database_writer.h"

#include <chrono>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

static std::string trim_name(const char* p, int width) {
    int b = 0, e = width;
    while (b < e && (p[b] == ' ' || p[b] == '\0')) b++;
    while (e > b && (p[e - 1] == ' ' || p[e - 1] == '\0')) e--;
    return std::string(p + b, p + e);
}

extern "C" {

// atom_names: n_atoms x 4 chars (space/NUL padded), res_names: n_atoms x 3 chars.
// Returns the FCZ size written to out (or -needed if out_cap is too small, -1 on failure).
long ref_compress(int n_atoms, const char* atom_names, const char* res_names, const char* chain_ids,
                  const int* atom_index, const int* res_index,
                  const float* x, const float* y, const float* z, const float* bfac,
                  const char* title, int title_len, int anchor_threshold,
                  unsigned char* out, long out_cap) {
    std::vector<AtomCoordinate> atoms;
    atoms.reserve(n_atoms);
    for (int i = 0; i < n_atoms; i++) {
        atoms.emplace_back(trim_name(atom_names + 4 * i, 4), trim_name(res_names + 3 * i, 3),
                           std::string(1, chain_ids[i]), atom_index[i], res_index[i],
                           x[i], y[i], z[i], 1.0f, bfac[i]);
    }
    Foldcomp c;
    c.strTitle = std::string(title, title + title_len);
    c.anchorThreshold = anchor_threshold;
    try {
        tcb::span<AtomCoordinate> sp(atoms.data(), atoms.size());
        c.compress(sp);
    } catch (...) {
        return -1;
    }
    std::ostringstream oss;
    c.writeStream(oss);
    std::string s = oss.str();
    if ((long)s.size() > out_cap) return -(long)s.size();
    memcpy(out, s.data(), s.size());
    // The 4 struct-padding bytes of CompressedFileHeader are uninitialised in the reference
    // (src/foldcomp.h:118-136, src/foldcomp.cpp:1341); zero them so outputs are comparable.
    if (s.size() >= 24) { out[14] = out[15] = out[22] = out[23] = 0; }
    return (long)s.size();
}

// Pre-quantisation angles of one chain (Foldcomp::preprocess, src/foldcomp.cpp:450).
// out arrays need n_res entries each; returns number of values per array (n_res-1) or -1.
int ref_angles(int n_atoms, const char* atom_names, const char* res_names, const char* chain_ids,
               const int* atom_index, const int* res_index,
               const float* x, const float* y, const float* z, const float* bfac,
               float* phi, float* psi, float* omega, float* n_ca_c, float* ca_c_n, float* c_n_ca,
               float* sc_torsions, int sc_cap, int* n_sc) {
    std::vector<AtomCoordinate> atoms;
    for (int i = 0; i < n_atoms; i++) {
        atoms.emplace_back(trim_name(atom_names + 4 * i, 4), trim_name(res_names + 3 * i, 3),
                           std::string(1, chain_ids[i]), atom_index[i], res_index[i],
                           x[i], y[i], z[i], 1.0f, bfac[i]);
    }
    Foldcomp c;
    try {
        tcb::span<AtomCoordinate> sp(atoms.data(), atoms.size());
        c.preprocess(sp);
    } catch (...) {
        return -1;
    }
    size_t n = c.phi.size();
    for (size_t i = 0; i < n; i++) {
        phi[i] = c.phi[i]; psi[i] = c.psi[i]; omega[i] = c.omega[i];
        n_ca_c[i] = c.n_ca_c_angle[i]; ca_c_n[i] = c.ca_c_n_angle[i]; c_n_ca[i] = c.c_n_ca_angle[i];
    }
    int k = 0;
    for (auto& v : c.sideChainAnglesPerResidue)
        for (float f : v) { if (k < sc_cap) sc_torsions[k] = f; k++; }
    *n_sc = k;
    return (int)n;
}

// Returns number of atoms (or -needed if cap too small; -1000-k on read error k).
int ref_decompress(const unsigned char* fcz, long len, int alt_order,
                   float* x, float* y, float* z, float* bfac,
                   char* atom_names /*4 per atom*/, char* res_names /*3 per atom*/,
                   int* atom_index, int* res_index, char* chain_ids, int cap,
                   char* title_out, int title_cap, int* title_len) {
    std::string s((const char*)fcz, (size_t)len);
    std::istringstream iss(s);
    Foldcomp c;
    int flag = c.read(iss);
    if (flag != 0) return -1000 + flag;
    c.useAltAtomOrder = alt_order != 0;
    std::vector<AtomCoordinate> atoms;
    try {
        flag = c.decompress(atoms);
    } catch (...) {
        return -1003;
    }
    if (flag != 0) return -1004;
    if (title_len) {
        *title_len = (int)c.strTitle.size();
        if (title_out) memcpy(title_out, c.strTitle.data(), std::min<size_t>(title_cap, c.strTitle.size()));
    }
    if ((int)atoms.size() > cap) return -(int)atoms.size();
    for (size_t i = 0; i < atoms.size(); i++) {
        x[i] = atoms[i].coordinate.x; y[i] = atoms[i].coordinate.y; z[i] = atoms[i].coordinate.z;
        bfac[i] = atoms[i].tempFactor;
        memset(atom_names + 4 * i, ' ', 4);
        memcpy(atom_names + 4 * i, atoms[i].atom.data(), std::min<size_t>(4, atoms[i].atom.size()));
        memset(res_names + 3 * i, ' ', 3);
        memcpy(res_names + 3 * i, atoms[i].residue.data(), std::min<size_t>(3, atoms[i].residue.size()));
        atom_index[i] = atoms[i].atom_index;
        res_index[i] = atoms[i].residue_index;
        chain_ids[i] = atoms[i].chain.empty() ? ' ' : atoms[i].chain[0];
    }
    return (int)atoms.size();
}

// PDB text of a decompressed structure exactly as the reference writes it
// (writeAtomCoordinatesToPDB, src/atom_coordinate.cpp:220-291). Returns length or -needed.
long ref_decompress_pdb(const unsigned char* fcz, long len, int alt_order, char* out, long out_cap) {
    std::string s((const char*)fcz, (size_t)len);
    std::istringstream iss(s);
    Foldcomp c;
    if (c.read(iss) != 0) return -1;
    c.useAltAtomOrder = alt_order != 0;
    std::vector<AtomCoordinate> atoms;
    try { c.decompress(atoms); } catch (...) { return -1; }
    std::ostringstream oss;
    writeAtomCoordinatesToPDB(atoms, c.strTitle, oss);
    std::string o = oss.str();
    if ((long)o.size() > out_cap) return -(long)o.size();
    memcpy(out, o.data(), o.size());
    return (long)o.size();
}

// extract (src/foldcomp.cpp:1260): type 0 = plddt, 1 = fasta. Returns length or -needed.
long ref_extract(const unsigned char* fcz, long len, int type, int digits, char* out, long out_cap) {
    std::string s((const char*)fcz, (size_t)len);
    std::istringstream iss(s);
    Foldcomp c;
    if (c.read(iss) != 0) return -1;
    std::string data;
    c.extract(data, type, digits);
    if ((long)data.size() > out_cap) return -(long)data.size();
    memcpy(out, data.data(), data.size());
    return (long)data.size();
}


// ---- timed batch round trip for bench.py's cpu_baseline leg ("kind": "reference") ---------------------
// SoA batch in (same arrays as fcz_chain_batch, atom/residue names through code->name tables), the
// reference's own objects in between. AtomCoordinate vectors are built before the clock starts (the
// reference's parsers produce them; they are not part of the codec path timed here). Timed:
//   compress leg:   Foldcomp::compress + writeStream   (src/foldcomp.cpp:562, :1038)
//   decompress leg: Foldcomp::read + decompress          (src/foldcomp.cpp:904, :779)
// parallelised over chains with OpenMP exactly like the reference's `-t` (src/input_processor.h:85-89).
// With the chains' own titles / numbering / chain ids (all five `meta` pointers given) the records are the ones the product writes
// for the same batch, and hash_fcz[c] / hash_xyz[c] (may be null) return what the LIVE reference produced per chain: FNV-1a of the
// record with the 4 uninitialised header bytes zeroed, and of the bit patterns of the decoded x, y, z of every atom followed by the
// tempFactor of every CA atom (a third, untimed pass). ref_hash_records / ref_hash_atoms below hash a caller's arrays the same way.
static inline unsigned long long fnv_bytes(const unsigned char* p, size_t n, unsigned long long h) {
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
static inline unsigned long long fnv_word(unsigned w, unsigned long long h) { h ^= w; h *= 0x100000001b3ull; return h; }
static inline unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static const unsigned long long FNV0 = 0xcbf29ce484222325ull;
static unsigned long long hash_record_masked(const unsigned char* p, size_t n) {
    unsigned long long h = FNV0;
    for (size_t i = 0; i < n; i++) { const unsigned char b = (i == 14 || i == 15 || i == 22 || i == 23) ? 0 : p[i]; h ^= b; h *= 0x100000001b3ull; }
    return h;
}
void ref_hash_records(const unsigned char* blob, const unsigned long long* off, long n, unsigned long long* out) {
    for (long i = 0; i < n; i++) out[i] = hash_record_masked(blob + off[i], (size_t)(off[i + 1] - off[i]));
}
// chain c: atoms [atom_off[c], atom_off[c+1]) of x / y / z, residues [res_off[c], res_off[c+1]) of bfac_res
void ref_hash_atoms(const float* x, const float* y, const float* z, const unsigned* atom_off, const float* bfac_res, const unsigned* res_off,
                    long n, unsigned long long* out) {
    for (long c = 0; c < n; c++) {
        unsigned long long h = FNV0;
        for (unsigned a = atom_off[c]; a < atom_off[c + 1]; a++) { h = fnv_word(fbits(x[a]), h); h = fnv_word(fbits(y[a]), h); h = fnv_word(fbits(z[a]), h); }
        for (unsigned r = res_off[c]; r < res_off[c + 1]; r++) h = fnv_word(fbits(bfac_res[r]), h);
        out[c] = h;
    }
}

int ref_bench_roundtrip(int n_chains, const unsigned* res_off, const unsigned* atom_off,
                        const float* x, const float* y, const float* z,
                        const unsigned char* atom_code, const unsigned char* res_code, const float* bfac_ca,
                        const char* atom_names /*37 x 4, NUL padded*/, const char* res_names /*24 x 4*/,
                        int anchor_threshold, int n_threads, double* t_compress, double* t_decompress,
                        unsigned long long* fcz_bytes, unsigned long long* atoms_out,
                        const char* titles, const unsigned* title_off, const int* first_res, const int* first_atom, const char* chain_id,
                        unsigned long long* hash_fcz, unsigned long long* hash_xyz) {
    const bool meta = titles && title_off && first_res && first_atom && chain_id;
    std::vector<std::vector<AtomCoordinate>> chains(n_chains);
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
    for (int c = 0; c < n_chains; c++) {
        unsigned r0 = res_off[c], r1 = res_off[c + 1];
        int serial = meta ? first_atom[c] : 1;
        const int res0 = meta ? first_res[c] : 1;
        const std::string chain = meta ? std::string(1, chain_id[c]) : std::string("A");
        for (unsigned r = r0; r < r1; r++)
            for (unsigned a = atom_off[r]; a < atom_off[r + 1]; a++) {
                int code = atom_code[a];
                chains[c].emplace_back(std::string(code < 37 ? atom_names + 4 * code : "H"), std::string(res_names + 4 * res_code[r]),
                                       chain, serial++, (int)(r - r0) + res0, x[a], y[a], z[a], 1.0f, bfac_ca[r]);
            }
    }
    std::vector<std::string> fcz(n_chains);
    int fail = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
    for (int c = 0; c < n_chains; c++) {
        try {
            Foldcomp comp;
            comp.strTitle = meta ? std::string(titles + title_off[c], title_off[c + 1] - title_off[c]) : std::string("synth_0000000000");
            comp.anchorThreshold = anchor_threshold;
            tcb::span<AtomCoordinate> sp(chains[c].data(), chains[c].size());
            comp.compress(sp);
            std::ostringstream oss;
            comp.writeStream(oss);
            fcz[c] = oss.str();
        } catch (...) {
#pragma omp atomic
            fail++;
        }
    }
    double t1 = now();
    unsigned long long total_atoms = 0;
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads) reduction(+ : total_atoms)
    for (int c = 0; c < n_chains; c++) {
        try {
            std::istringstream iss(fcz[c]);
            Foldcomp comp;
            if (comp.read(iss) != 0) { continue; }
            std::vector<AtomCoordinate> atoms;
            comp.decompress(atoms);
            total_atoms += atoms.size();
        } catch (...) {
#pragma omp atomic
            fail++;
        }
    }
    double t2 = now();
    if (hash_fcz) for (int c = 0; c < n_chains; c++) hash_fcz[c] = hash_record_masked((const unsigned char*)fcz[c].data(), fcz[c].size());
    if (hash_xyz) {
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
        for (int c = 0; c < n_chains; c++) {
            hash_xyz[c] = 0;
            try {
                std::istringstream iss(fcz[c]);
                Foldcomp comp;
                if (comp.read(iss) != 0) continue;
                std::vector<AtomCoordinate> atoms;
                comp.decompress(atoms);
                unsigned long long h = FNV0;
                for (const AtomCoordinate& a : atoms) { h = fnv_word(fbits(a.coordinate.x), h); h = fnv_word(fbits(a.coordinate.y), h); h = fnv_word(fbits(a.coordinate.z), h); }
                for (const AtomCoordinate& a : atoms) if (a.atom == "CA") h = fnv_word(fbits(a.tempFactor), h);
                hash_xyz[c] = h;
            } catch (...) {}
        }
    }
    unsigned long long bytes = 0;
    for (auto& s : fcz) bytes += s.size();
    *t_compress = t1 - t0; *t_decompress = t2 - t1; *fcz_bytes = bytes; *atoms_out = total_atoms;
    return fail;
}

// ---- ingest: StructureReader + the fragmenting of src/main.cpp:457-474 -----------------------------------------------
// Parses a file image (PDB / mmCIF, optionally gzipped; `name` decides as in the reference) and reports the atom table after
// removeAlternativePosition plus the fragment ranges the compress lambda would hand to the codec.
// Returns the number of atoms (or -needed if cap_atoms is too small, -1 when the reference's reader fails).
// frag[4*k .. 4*k+3] = {first atom, end atom, chain ordinal, fragment ordinal within the chain}; n_chains = identifyChains count.
long ref_load_structure(const char* buf, long len, const char* name, long cap_atoms, char* atom_names, char* res_names,
                        char* chain_ids, int* atom_index, int* res_index, float* x, float* y, float* z, float* bfac,
                        char* title, int title_cap, int* title_len, int* frag, int frag_cap, int* n_frag, int* n_chains) {
    StructureReader reader;
    if (!reader.loadFromBuffer(buf, (size_t)len, std::string(name))) return -1;
    std::vector<AtomCoordinate> atoms;
    reader.readAllAtoms(atoms);
    removeAlternativePosition(atoms);
    const long n = (long)atoms.size();
    *title_len = (int)reader.title.size();
    if (title_cap > 0) { const int t = std::min<int>(*title_len, title_cap); memcpy(title, reader.title.data(), t); }
    *n_frag = 0; *n_chains = 0;
    if (atoms.empty()) return 0;      // src/main.cpp:459-462 stops here ("No atoms found"): identifyChains is never given an empty vector
    std::vector<std::pair<size_t, size_t>> chains = identifyChains(atoms);
    *n_chains = (int)chains.size();
    int nf = 0;
    for (size_t i = 0; i < chains.size(); i++) {
        std::vector<std::pair<size_t, size_t>> fr = identifyDiscontinousResInd(atoms, chains[i].first, chains[i].second);
        for (size_t j = 0; j < fr.size(); j++) {
            if (nf < frag_cap) { frag[4 * nf] = (int)fr[j].first; frag[4 * nf + 1] = (int)fr[j].second; frag[4 * nf + 2] = (int)i; frag[4 * nf + 3] = (int)j; }
            nf++;
        }
    }
    *n_frag = nf;
    if (n > cap_atoms) return -n;
    for (long i = 0; i < n; i++) {
        const AtomCoordinate& a = atoms[i];
        memset(atom_names + 4 * i, 0, 4); memcpy(atom_names + 4 * i, a.atom.data(), std::min<size_t>(4, a.atom.size()));
        memset(res_names + 3 * i, 0, 3); memcpy(res_names + 3 * i, a.residue.data(), std::min<size_t>(3, a.residue.size()));
        chain_ids[i] = a.chain.empty() ? ' ' : a.chain[0];
        atom_index[i] = a.atom_index; res_index[i] = a.residue_index;
        x[i] = a.coordinate.x; y[i] = a.coordinate.y; z[i] = a.coordinate.z; bfac[i] = a.tempFactor;
    }
    return n;
}

// ---- database container: make_writer / writer_append / free_writer, make_reader / reader_get_* -------------------------
// Writes n entries (blob + offsets, keys, NUL-separated names) with the reference's writer; 0 on success.
int ref_db_write(const char* data_path, const char* index_path, int n, const unsigned char* blob, const unsigned long long* off,
                 const unsigned* keys, const char* names) {
    void* w = make_writer(data_path, index_path);
    if (!w) return -1;
    const char* nm = names;
    for (int i = 0; i < n; i++) {
        writer_append(w, (const char*)blob + off[i], (size_t)(off[i + 1] - off[i]), keys[i], nm);
        nm += strlen(nm) + 1;
    }
    free_writer(w);
    return 0;
}
// Opens a database with the reference's reader (data + reverse lookup) and reports entry `id` in the reader's order: key,
// length, offset, name (looked up by key) and up to cap bytes of data. Returns the number of entries, or -1.
long ref_db_read(const char* data_path, const char* index_path, long id, unsigned* key, long long* length, long long* offset,
                 char* name, int name_cap, unsigned char* data, long cap) {
    void* r = make_reader(data_path, index_path, DB_READER_USE_DATA | DB_READER_USE_LOOKUP_REVERSE);
    if (!r) return -1;
    const long n = (long)reader_get_size(r);
    if (id >= 0 && id < n) {
        *key = reader_get_key(r, id); *length = reader_get_length(r, id); *offset = reader_get_offset(r, id);
        const char* nm = reader_lookup_name_alloc(r, *key);
        if (name_cap > 0) { name[0] = 0; if (nm) { strncpy(name, nm, name_cap - 1); name[name_cap - 1] = 0; } }
        if (nm) free((void*)nm);
        const char* d = reader_get_data(r, id);
        if (d && cap > 0) memcpy(data, d, (size_t)std::min<long long>(cap, *length));
    }
    free_reader(r);
    return n;
}
// name -> key -> id with the reference's lookup (DB_READER_USE_LOOKUP), as `foldcomp.open(ids=...)` does; -1 when missing
long ref_db_lookup(const char* data_path, const char* index_path, const char* name) {
    void* r = make_reader(data_path, index_path, DB_READER_USE_LOOKUP);
    if (!r) return -2;
    const uint32_t key = reader_lookup_entry(r, name);
    const long id = key == UINT32_MAX ? -1 : (long)reader_get_id(r, key);
    free_reader(r);
    return id;
}

// ---- end to end on files: what the compress lambda of src/main.cpp:438-536 does per input, under the reference's own
// `omp parallel for` (src/input_processor.h:85-101): read the file, StructureReader::loadFromBuffer, removeAlternativePosition,
// identifyChains / identifyDiscontinousResInd, Foldcomp::compress + writeStream per fragment (the bytes are kept in memory as the
// --db path does before writer_append). Wall time of the whole loop; returns the number of files that failed to load.
// file_hash (may be null): per file, the hashes (hash_record_masked) of its fragments' records chained in fragment order; 0 = failed.
int ref_compress_files(const char* paths, int n_files, int n_threads, int anchor_threshold, double* seconds,
                       unsigned long long* residues, unsigned long long* fcz_bytes, unsigned char* first_out, long first_cap, long* first_len,
                       unsigned long long* file_hash) {
    std::vector<std::string> files;
    const char* p = paths;
    for (int i = 0; i < n_files; i++) { files.emplace_back(p); p += files.back().size() + 1; }
    unsigned long long res = 0, bytes = 0; int failed = 0;
    *first_len = 0;
    const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : res, bytes, failed)
    for (int i = 0; i < n_files; i++) {
        FILE* f = fopen(files[i].c_str(), "rb");
        if (!f) { failed++; continue; }
        std::string buf; char tmp[1 << 16]; size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
        fclose(f);
        const size_t slash = files[i].find_last_of('/');
        const std::string base = slash == std::string::npos ? files[i] : files[i].substr(slash + 1);
        StructureReader reader;
        if (!reader.loadFromBuffer(buf.data(), buf.size(), base)) { failed++; continue; }
        std::vector<AtomCoordinate> atoms;
        reader.readAllAtoms(atoms);
        if (atoms.empty()) { failed++; continue; }
        const size_t dot = base.find_last_of('.');
        const std::string stem = dot == std::string::npos ? base : base.substr(0, dot);
        const std::string title = reader.title == base ? stem : reader.title;
        removeAlternativePosition(atoms);
        std::vector<std::pair<size_t, size_t>> chains = identifyChains(atoms);
        unsigned long long fh = 0;
        for (size_t c = 0; c < chains.size(); c++) {
            std::vector<std::pair<size_t, size_t>> fr = identifyDiscontinousResInd(atoms, chains[c].first, chains[c].second);
            for (size_t j = 0; j < fr.size(); j++) {
                tcb::span<AtomCoordinate> sp(&atoms[fr[j].first], atoms.data() + fr[j].second);
                Foldcomp comp;
                comp.strTitle = title;
                comp.anchorThreshold = anchor_threshold;
                try { comp.compress(sp); } catch (...) { failed++; continue; }
                std::ostringstream oss;
                comp.writeStream(oss);
                const std::string os = oss.str();
                res += comp.nResidue; bytes += os.size();
                fh = fnv_word((unsigned)(hash_record_masked((const unsigned char*)os.data(), os.size()) >> 32),
                              fnv_word((unsigned)hash_record_masked((const unsigned char*)os.data(), os.size()), fh ? fh : FNV0));
                if (i == 0 && c == 0 && j == 0 && (long)os.size() <= first_cap) {
                    memcpy(first_out, os.data(), os.size());
                    if (os.size() >= 24) first_out[14] = first_out[15] = first_out[22] = first_out[23] = 0;
                    *first_len = (long)os.size();
                }
            }
        }
        if (file_hash) file_hash[i] = fh;
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *residues = res; *fcz_bytes = bytes;
    return failed;
}

// ---- end to end on a database: what the decompress lambda of src/main.cpp:612-689 does per entry under the reference's own
// `omp for` over the database (DatabaseProcessor::run, src/input_processor.h:237-257): Foldcomp::read, decompress,
// writeAtomCoordinatesToPDB into a string + the '\0' of the --db output, then -- when out_path is given -- writer_append under
// `omp critical` into a database made by make_writer and closed by free_writer, exactly as src/main.cpp:571-575,656-664,683-685
// do (out_path == NULL: the text is counted and dropped). `passes` walks over the database. Wall time of the loop incl.
// free_writer; returns the number of entries that failed.
int ref_decompress_db(const char* data_path, const char* index_path, int n_threads, int alt_order, int passes, const char* out_path, double* seconds,
                      unsigned long long* residues, unsigned long long* text_bytes, char* first_out, long first_cap, long* first_len) {
    void* r = make_reader(data_path, index_path, DB_READER_USE_DATA | DB_READER_USE_LOOKUP_REVERSE);
    if (!r) return -1;
    void* wh = nullptr;
    unsigned int key = 0;
    if (out_path) wh = make_writer(out_path, (std::string(out_path) + ".index").c_str());
    const long n = (long)reader_get_size(r);
    unsigned long long res = 0, bytes = 0; int failed = 0;
    *first_len = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int pass = 0; pass < passes; pass++) {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : res, bytes, failed)
        for (long i = 0; i < n; i++) {
            Foldcomp comp;
            std::istringstream in(std::string(reader_get_data(r, i), (size_t)reader_get_length(r, i)));
            if (comp.read(in) != 0) { failed++; continue; }
            std::vector<AtomCoordinate> atoms;
            comp.useAltAtomOrder = alt_order != 0;
            if (comp.decompress(atoms) != 0) { failed++; continue; }
            std::ostringstream oss;
            writeAtomCoordinatesToPDB(atoms, comp.strTitle, oss);
            oss << '\0';
            const std::string os = oss.str();
            res += comp.nResidue; bytes += os.size();
            if (pass == 0 && i == 0 && (long)os.size() <= first_cap) { memcpy(first_out, os.data(), os.size()); *first_len = (long)os.size(); }
            if (wh) {
                const char* nm = reader_lookup_name_alloc(r, reader_get_key(r, i));
#pragma omp critical
                {
                    writer_append(wh, os.c_str(), os.size(), key, nm ? nm : "");
                    key++;
                }
                if (nm) free((void*)nm);
            }
        }
    }
    if (wh) free_writer(wh);
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *residues = res; *text_bytes = bytes;
    free_reader(r);
    return failed;
}

}  // extern "C"
