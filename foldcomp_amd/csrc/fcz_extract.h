// fcz_extract.h -- `foldcomp extract` straight from FCZ bytes, no reconstruction (Foldcomp::extract,
// reference src/foldcomp.cpp:1260-1336; digit rules :1286-1325). A byte scan: reads the B-factor bytes (pLDDT) or the
// 5-bit residue codes of the packed words (sequence) and writes a few characters per residue.
//   mode 0, digits 1..4: pLDDT per residue as "d", "dd", "dd.d", "dd.dd" (values <= 1 print their decimals when
//                        digits <= 2), joined by ',' when digits > 1
//   mode 1:              one-letter amino-acid sequence
// The caller wraps the data into the FASTA-like / TSV line (title and residue count are host metadata).
#pragma once
#include "fcz_kernels.h"

namespace fcz {

__host__ __device__ __forceinline__ uint32_t extract_width(int mode, int digits) {
    return mode == 1 ? 1u : (digits <= 1 ? 1u : digits == 2 ? 2u : digits == 3 ? 4u : 5u);
}
// data bytes of an entry with n residues
__host__ __device__ __forceinline__ uint64_t extract_bytes(uint32_t n, int mode, int digits) {
    if (n == 0) return 0;
    return (uint64_t)n * extract_width(mode, digits) + ((mode == 0 && digits > 1) ? (uint64_t)n - 1 : 0);
}
// the checks Foldcomp::read makes before extract touches the entry: magic and a complete record
__host__ __device__ __forceinline__ uint32_t extract_entry_residues(const uint8_t* e, uint64_t len) {
    if (len < 76 || !(e[0] == 'F' && e[1] == 'C' && e[2] == 'M' && e[3] == 'P')) return 0;
    const uint32_t n = (uint32_t)e[4] | ((uint32_t)e[5] << 8), n_anchor = e[12];
    const uint32_t n_sc = (uint32_t)e[16] | ((uint32_t)e[17] << 8) | ((uint32_t)e[18] << 16) | ((uint32_t)e[19] << 24);
    const uint32_t tl = (uint32_t)e[24] | ((uint32_t)e[25] << 8) | ((uint32_t)e[26] << 16) | ((uint32_t)e[27] << 24);
    if (tl > len || n_sc > len) return 0;
    const rec_layout L = make_layout(n, n_anchor, tl, n_sc);
    return (uint64_t)L.size <= len ? n : 0;
}

__global__ __launch_bounds__(BLOCK) void k_extract_sizes(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries,
                                                         int mode, int digits, uint64_t* __restrict__ sizes) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_entries) return;
    sizes[i] = extract_bytes(extract_entry_residues(blob + off[i], off[i + 1] - off[i]), mode, digits);
}

__global__ __launch_bounds__(BLOCK) void k_extract(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ off, uint32_t n_entries,
                                                   int mode, int digits, const uint64_t* __restrict__ data_off, uint8_t* __restrict__ data) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (i >= n_entries) return;
    const uint8_t* e = blob + off[i];
    const uint32_t n = extract_entry_residues(e, off[i + 1] - off[i]);
    if (n == 0) return;
    const entry_view v = view_entry(e);
    uint8_t* dst = data + data_off[i];
    if (mode == 1) {
        for (uint32_t k = lane; k < n; k += WAVE) {
            const uint32_t rc = e[v.L.o_words + 8 * (size_t)k] >> 3;
            dst[k] = (uint8_t)(rc < 24 ? fcz_res1[rc] : 'X');
        }
        return;
    }
    const float mn = ld_f32(e + v.L.o_tmp), cf = ld_f32(e + v.L.o_tmp + 4);
    const float maxval = (cf * 255.0f) + mn;
    const bool zero_one = maxval <= 1.0f && digits <= 2;
    const uint32_t w = extract_width(0, digits) + (digits > 1 ? 1u : 0u);   // characters per residue incl. the separator
    for (uint32_t k = lane; k < n; k += WAVE) {
        const float tf = dequant((uint32_t)e[v.L.o_tbytes + k], mn, cf);
        uint32_t d1, d2;
        float cl;
        if (zero_one) {
            cl = __builtin_fminf(__builtin_fmaxf(tf, 0.0f), 1.0f);
            d1 = (uint32_t)((int)(cl * 10.0f) % 10);
            d2 = (uint32_t)((int)(cl * 100.0f) % 10);
        } else {
            cl = __builtin_fminf(__builtin_fmaxf(tf, 0.0f), 100.0f);
            d1 = (uint32_t)(int)(cl / 10.0f);          // (char)(clamped / 10.0f): 100 -> 10 -> ':'
            d2 = (uint32_t)((int)cl % 10);
        }
        const uint32_t d3 = (uint32_t)((int)(cl * 10.0f) % 10), d4 = (uint32_t)((int)(cl * 100.0f) % 10);
        uint8_t* p = dst + (size_t)k * w;
        p[0] = (uint8_t)('0' + d1);
        if (digits > 1) p[1] = (uint8_t)('0' + d2);
        if (digits >= 3) { p[2] = '.'; p[3] = (uint8_t)('0' + d3); }
        if (digits == 4) p[4] = (uint8_t)('0' + d4);
        if (digits > 1 && k + 1 < n) p[w - 1] = ',';
    }
}

}  // namespace fcz
