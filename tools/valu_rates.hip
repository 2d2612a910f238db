// valu_rates.hip -- issue cost of the VALU instructions the FCZ kernels are made of, on the GPU this runs on.
// Each kernel runs ITER x 8 independent copies of one instruction per lane; cycles per wave-instruction per SIMD =
// time x clock x SIMDs / wave-instructions. Used to price kernel variants, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define ITER 4096

#define OP8(asmstr, c) \
    asm volatile(asmstr : "+" c(a0) : c(b0), c(b1)); asm volatile(asmstr : "+" c(a1) : c(b0), c(b1)); \
    asm volatile(asmstr : "+" c(a2) : c(b0), c(b1)); asm volatile(asmstr : "+" c(a3) : c(b0), c(b1)); \
    asm volatile(asmstr : "+" c(a4) : c(b0), c(b1)); asm volatile(asmstr : "+" c(a5) : c(b0), c(b1)); \
    asm volatile(asmstr : "+" c(a6) : c(b0), c(b1)); asm volatile(asmstr : "+" c(a7) : c(b0), c(b1));

#define KERNEL_F32(name, asmstr)                                                                    \
    __global__ __launch_bounds__(256) void name(float* out, float s) {                              \
        float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = s * 0.999f, b1 = s * 1.0001f;                                                    \
        for (int i = 0; i < ITER; i++) { OP8(asmstr, "v") }                                         \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
    }
#define KERNEL_F64(name, asmstr)                                                                    \
    __global__ __launch_bounds__(256) void name(float* out, float s) {                              \
        double a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        double b0 = s * 0.999, b1 = s * 1.0001;                                                     \
        for (int i = 0; i < ITER; i++) { OP8(asmstr, "v") }                                         \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
    }

KERNEL_F32(k_fma_f32, "v_fma_f32 %0, %1, %2, %0")
KERNEL_F32(k_mul_f32, "v_mul_f32 %0, %1, %0")
KERNEL_F32(k_add_f32, "v_add_f32 %0, %1, %0")
KERNEL_F32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F32(k_rsq_f32, "v_rsq_f32 %0, %0")
KERNEL_F32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F32(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL_F32(k_and_b32, "v_and_b32 %0, %1, %0")
KERNEL_F32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL_F32(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc")
KERNEL_F32(k_cmp_f32, "v_cmp_lt_f32 vcc, %1, %0")
KERNEL_F32(k_max3_f32, "v_max3_f32 %0, %1, %2, %0")
KERNEL_F32(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL_F32(k_div_scale, "v_div_scale_f32 %0, vcc, %1, %2, %0")
KERNEL_F32(k_div_fixup, "v_div_fixup_f32 %0, %1, %2, %0")
KERNEL_F32(k_div_fmas, "v_div_fmas_f32 %0, %1, %2, %0")
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %1, %2, %0")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %1, %0")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %1, %0")
KERNEL_F64(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_F64(k_cmp_f64, "v_cmp_lt_f64 vcc, %1, %0")
// packed f32 (two floats per lane in a VGPR pair): what two-items-per-lane kernels would be made of
KERNEL_F64(k_pk_fma_f32, "v_pk_fma_f32 %0, %1, %2, %0")
KERNEL_F64(k_pk_mul_f32, "v_pk_mul_f32 %0, %1, %0")
KERNEL_F64(k_pk_add_f32, "v_pk_add_f32 %0, %1, %0")
KERNEL_F64(k_pk_mov_b32, "v_pk_mov_b32 %0, %1, %2")
KERNEL_F32(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
KERNEL_F32(k_sub_f32, "v_sub_f32 %0, %1, %0")
KERNEL_F32(k_add_u32, "v_add_u32 %0, %1, %0")
KERNEL_F32(k_cmp_class, "v_cmp_class_f32 vcc, %0, %1")
KERNEL_F32(k_mov_b32, "v_mov_b32 %0, %1")

// v_cndmask with a mask the kernel itself produced (one v_cmp per 8 selects), vcc and SGPR-pair forms
__global__ __launch_bounds__(256) void k_cndmask_vcc(float* out, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = s * 0.999f, b1 = s * 1.0001f;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %8, %0, vcc\n v_cndmask_b32 %1, %8, %1, vcc\n v_cndmask_b32 %2, %8, %2, vcc\n"
                     "v_cndmask_b32 %3, %8, %3, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %5, %8, %5, vcc\n"
                     "v_cndmask_b32 %6, %8, %6, vcc\n v_cndmask_b32 %7, %8, %7, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(256) void k_cndmask_sgpr(float* out, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = s * 0.999f, b1 = s * 1.0001f;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cmp_lt_f32 s[20:21], %8, %9\n v_cndmask_b32 %0, %8, %0, s[20:21]\n v_cndmask_b32 %1, %8, %1, s[20:21]\n v_cndmask_b32 %2, %8, %2, s[20:21]\n"
                     "v_cndmask_b32 %3, %8, %3, s[20:21]\n v_cndmask_b32 %4, %8, %4, s[20:21]\n v_cndmask_b32 %5, %8, %5, s[20:21]\n"
                     "v_cndmask_b32 %6, %8, %6, s[20:21]\n v_cndmask_b32 %7, %8, %7, s[20:21]"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "s20", "s21");
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
#define SEL_KERNEL(name, setup, sel)                                                                                      \
    __global__ __launch_bounds__(256) void name(float* out, float s) {                                                    \
        float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b0 = s * 0.999f, b1 = s * 1.0001f;                                                                          \
        for (int i = 0; i < ITER; i++) {                                                                                  \
            asm volatile(setup "\n" sel(0) "\n" sel(1) "\n" sel(2) "\n" sel(3) "\n" sel(4) "\n" sel(5) "\n" sel(6) "\n" sel(7)     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1) : "vcc", "s20", "s21"); \
        }                                                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                      \
    }
#define SEL_E64_VCC(i) "v_cndmask_b32_e64 %" #i ", %8, %" #i ", vcc"
#define SEL_E32_VCC(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc"
#define SEL_E64_SGPR(i) "v_cndmask_b32_e64 %" #i ", %8, %" #i ", s[20:21]"
#define ADDC_VCC(i) "v_addc_co_u32_e32 %" #i ", vcc, %8, %" #i ", vcc"
#define ADDC_SGPR(i) "v_addc_co_u32_e64 %" #i ", s[20:21], %8, %" #i ", s[20:21]"
SEL_KERNEL(k_sel_e64_vcc_vcmp, "v_cmp_lt_f32_e32 vcc, %8, %9", SEL_E64_VCC)
SEL_KERNEL(k_sel_e32_vcc_smov, "s_mov_b64 vcc, 0x5555", SEL_E32_VCC)
SEL_KERNEL(k_sel_e32_vcc_none, "s_nop 0", SEL_E32_VCC)
SEL_KERNEL(k_sel_e64_sgpr_smov, "s_mov_b64 s[20:21], 0x5555", SEL_E64_SGPR)
SEL_KERNEL(k_addc_vcc, "s_nop 0", ADDC_VCC)
SEL_KERNEL(k_addc_sgpr, "s_nop 0", ADDC_SGPR)

// LDS: 4-byte reads / writes at lane-consecutive addresses, 8 per iteration
__global__ __launch_bounds__(256) void k_ds_read_b32(float* out, float s) {
    __shared__ float buf[256 * 9];
    for (int i = threadIdx.x; i < 256 * 9; i += 256) buf[i] = s + i;
    __syncthreads();
    float acc = 0.f;
    const float* p = buf + threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        float v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:1024\n ds_read_b32 %2, %8 offset:2048\n ds_read_b32 %3, %8 offset:3072\n"
                     "ds_read_b32 %4, %8 offset:4096\n ds_read_b32 %5, %8 offset:5120\n ds_read_b32 %6, %8 offset:6144\n ds_read_b32 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"((unsigned)(size_t)p));
        acc += v0 + v7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_ds_read_b128(float* out, float s) {
    __shared__ __attribute__((aligned(16))) float buf[256 * 4 * 8];
    for (int i = threadIdx.x; i < 256 * 32; i += 256) buf[i] = s + i;
    __syncthreads();
    float acc = 0.f;
    const float* p = buf + 4 * threadIdx.x;
    for (int i = 0; i < ITER; i++) {
        float4 v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:4096\n ds_read_b128 %2, %8 offset:8192\n ds_read_b128 %3, %8 offset:12288\n"
                     "ds_read_b128 %4, %8 offset:16384\n ds_read_b128 %5, %8 offset:20480\n ds_read_b128 %6, %8 offset:24576\n ds_read_b128 %7, %8 offset:28672\n s_waitcnt lgkmcnt(0)"
                     : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"((unsigned)(size_t)p));
        acc += v0.x + v7.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// conversions need mixed register classes
__global__ __launch_bounds__(256) void k_cvt_f64_f32(float* out, float s) {
    float a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(a0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d1) : "v"(a1));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d2) : "v"(a2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d3) : "v"(a3));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e0) : "v"(a0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e1) : "v"(a1));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e2) : "v"(a2)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e3) : "v"(a3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + e0 + e1 + e2 + e3);
}
__global__ __launch_bounds__(256) void k_cvt_f32_f64(float* out, float s) {
    double a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d0) : "v"(a0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d1) : "v"(a1));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d2) : "v"(a2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(d3) : "v"(a3));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(e0) : "v"(a0)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(e1) : "v"(a1));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(e2) : "v"(a2)); asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(e3) : "v"(a3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3 + e0 + e1 + e2 + e3;
}
__global__ __launch_bounds__(256) void k_cvt_i32_f64(float* out, float s) {
    double a0 = s + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int d0 = 0, d1 = 0, d2 = 0, d3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(d0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(d1) : "v"(a1));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(d2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(d3) : "v"(a3));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e0) : "v"(a0)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e1) : "v"(a1));
        asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e2) : "v"(a2)); asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(e3) : "v"(a3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + e0 + e1 + e2 + e3);
}
// a dependent chain: latency of back-to-back dependent instructions (one wave per SIMD shows it)
__global__ __launch_bounds__(256) void k_dep_fma_f32(float* out, float s) {
    float a = s + threadIdx.x, b = s * 0.999f;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
__global__ __launch_bounds__(256) void k_dep_fma_f64(float* out, float s) {
    double a = s + threadIdx.x, b = s * 0.999;
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a) : "v"(b));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)a;
}

typedef void (*kern_t)(float*, float);
struct entry { const char* name; kern_t k; };

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    float* out; hipMalloc(&out, sizeof(float) * 256 * cus * 16);
    const entry es[] = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_rsq_f32", k_rsq_f32},
        {"v_sqrt_f32", k_sqrt_f32}, {"v_sin_f32", k_sin_f32}, {"v_and_b32", k_and_b32}, {"v_lshl_add_u32", k_lshl_add}, {"v_cndmask_b32", k_cndmask},
        {"v_cmp_lt_f32", k_cmp_f32}, {"v_max3_f32", k_max3_f32}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_div_scale_f32", k_div_scale},
        {"v_div_fixup_f32", k_div_fixup}, {"v_div_fmas_f32", k_div_fmas},
        {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rsq_f64", k_rsq_f64}, {"v_rcp_f64", k_rcp_f64},
        {"v_sqrt_f64", k_sqrt_f64}, {"v_cmp_lt_f64", k_cmp_f64}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mov_b32", k_pk_mov_b32}, {"v_fmac_f32", k_fmac_f32}, {"v_sub_f32", k_sub_f32}, {"v_add_u32", k_add_u32}, {"v_cmp_class_f32", k_cmp_class}, {"v_mov_b32", k_mov_b32}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cvt_i32_f64", k_cvt_i32_f64}, {"cmp+8 cndmask vcc", k_cndmask_vcc}, {"cmp+8 cndmask sgpr", k_cndmask_sgpr},
        {"cmp+8 sel e64 vcc", k_sel_e64_vcc_vcmp}, {"smov+8 sel e32 vcc", k_sel_e32_vcc_smov}, {"8 sel e32 vcc", k_sel_e32_vcc_none},
        {"smov+8 sel e64 sgpr", k_sel_e64_sgpr_smov}, {"8 addc e32 vcc", k_addc_vcc}, {"8 addc e64 sgpr", k_addc_sgpr},
        {"ds_read_b32", k_ds_read_b32}, {"ds_read_b128", k_ds_read_b128}, {"dep v_fma_f32", k_dep_fma_f32}, {"dep v_fma_f64", k_dep_fma_f64},
    };
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    printf("%-18s %10s %10s %10s   (cycles per wave-instruction per SIMD at 1 / 2 / 4 waves per SIMD)\n", "instruction", "1w", "2w", "4w");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const entry& e : es) {
        printf("%-18s", e.name);
        for (int wps : {1, 2, 4}) {
            const int blocks = cus * wps;   // 256-thread blocks: 4 waves = one per SIMD
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
            const double winstr_per_simd = (double)ITER * 8 * wps;   // (the cmp+8 rows: 9 instructions per 8 counted)   // wave-instructions issued on one SIMD
            printf(" %10.2f", ms * 1e-3 * clk / winstr_per_simd);
        }
        printf("\n");
    }
    return 0;
}
