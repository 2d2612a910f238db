/* fcz_oracle.h -- TEST INFRASTRUCTURE ONLY. CPU restatement of the Foldcomp per-chain codec.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this. */
#ifndef FCZ_ORACLE_H
#define FCZ_ORACLE_H
#include <stdint.h>
#include "../include/fcz_hip.h"   /* struct definitions only (fcz_chain_batch, fcz_atoms_out, ...) */

#ifdef __cplusplus
extern "C" {
#endif

/* one chain: atoms in input order, residue r = [atom_off[r], atom_off[r+1]).
 * Returns the FCZ size (bytes written to out) or a negative fcz_status. */
long fcz_oracle_compress_chain(uint32_t n_res, const uint32_t* atom_off,
                               const float* x, const float* y, const float* z,
                               const uint8_t* atom_code, const uint8_t* res_code, const float* bfac_ca,
                               int32_t first_res_index, int32_t first_atom_index, char chain_id,
                               const char* title, uint32_t title_len, int32_t anchor_threshold,
                               uint8_t* out, long out_cap);

/* pre-quantisation angles of one chain (n_res-1 values each); sc gets sum(natoms-3) torsions */
int fcz_oracle_angles_chain(uint32_t n_res, const uint32_t* atom_off,
                            const float* x, const float* y, const float* z,
                            const uint8_t* atom_code, const uint8_t* res_code,
                            float* phi, float* psi, float* omega,
                            float* n_ca_c, float* ca_c_n, float* c_n_ca, float* sc);

/* one FCZ entry -> coordinates (reference output order). Arrays must hold info.n_atoms_out atoms /
 * info.n_residues residues (see fcz_oracle_entry_info). Returns atoms written or negative status. */
int fcz_oracle_entry_info(const uint8_t* entry, uint64_t len, fcz_entry_info* info);
int fcz_oracle_decompress_chain(const uint8_t* entry, uint64_t len, int alt_order,
                                float* x, float* y, float* z, float* bfac_res,
                                uint8_t* res_code, uint8_t* atom_code);

/* batch forms with the same argument meaning as the product C-ABI (include/fcz_hip.h);
 * n_threads > 1 uses OpenMP over chains like the reference's `-t` (src/input_processor.h:85-89). */
int fcz_oracle_compress_sizes(const fcz_chain_batch* in, uint64_t* out_off);
int fcz_oracle_compress_batch(const fcz_chain_batch* in, const uint64_t* out_off, uint8_t* out,
                              int32_t* status, int n_threads);
int fcz_oracle_decompress_sizes(const uint8_t* blob, const uint64_t* off, uint32_t n,
                                fcz_entry_info* info, uint32_t* res_off, uint32_t* atom_off);
int fcz_oracle_decompress_batch(const uint8_t* blob, const uint64_t* off, uint32_t n,
                                const uint32_t* res_off, const uint32_t* atom_off, int alt_order,
                                const fcz_atoms_out* out, int n_threads);
int fcz_oracle_check(const uint8_t* entry, uint64_t len);

/* restated libm pieces, exported so tests can pin them against the host libm */
float fcz_oracle_sinf(float x);   /* glibc 2.35 sinf algorithm, plain (non-FMA) double arithmetic */
float fcz_oracle_cosf(float x);
void  fcz_oracle_use_restated_trig(int on);
void  fcz_oracle_math_sweep(int mode, uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads);
long  fcz_oracle_trig_mismatches(uint32_t lo_bits, uint32_t hi_bits, int n_threads);
void  fcz_oracle_acos_deg_sweep(uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads);
void  fcz_oracle_sincos_sweep(int is_cos, uint32_t start_bits, uint32_t stride, uint32_t count, float* out, int n_threads); /* decompress: 0 = host libm sinf/cosf (default), 1 = restated */

#ifdef __cplusplus
}
#endif
#endif
