// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU ops used by the
// ray-march inner loop on gfx950.  Build & run:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define ITERS 512

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, int n)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
    const double c = seed * 1.000001;
    const float cf = (float)c;
    for (int it = 0; it < n; ++it) {
#define D8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
#define D8U(INS) asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define F8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf));
#define I8(INS) asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0 | 3));
        if (OP == 0) { D8("v_add_f64") }
        if (OP == 1) { D8("v_mul_f64") }
        if (OP == 2) { asm volatile("v_fma_f64 %0, %0, %8, %0\nv_fma_f64 %1, %1, %8, %1\nv_fma_f64 %2, %2, %8, %2\nv_fma_f64 %3, %3, %8, %3\nv_fma_f64 %4, %4, %8, %4\nv_fma_f64 %5, %5, %8, %5\nv_fma_f64 %6, %6, %8, %6\nv_fma_f64 %7, %7, %8, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (OP == 3) { D8U("v_floor_f64") }
        if (OP == 4) { D8U("v_ceil_f64") }
        if (OP == 5) { D8U("v_rndne_f64") }
        if (OP == 6) { D8U("v_fract_f64") }
        if (OP == 7) { asm volatile("v_cvt_i32_f64 %0, %8\nv_cvt_i32_f64 %1, %9\nv_cvt_i32_f64 %2, %10\nv_cvt_i32_f64 %3, %11\nv_cvt_i32_f64 %4, %12\nv_cvt_i32_f64 %5, %13\nv_cvt_i32_f64 %6, %14\nv_cvt_i32_f64 %7, %15\n" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (OP == 8) { asm volatile("v_cvt_f64_i32 %0, %8\nv_cvt_f64_i32 %1, %9\nv_cvt_f64_i32 %2, %10\nv_cvt_f64_i32 %3, %11\nv_cvt_f64_i32 %4, %12\nv_cvt_f64_i32 %5, %13\nv_cvt_f64_i32 %6, %14\nv_cvt_f64_i32 %7, %15\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7)); }
        if (OP == 9) { asm volatile("v_cvt_f32_f64 %0, %8\nv_cvt_f32_f64 %1, %9\nv_cvt_f32_f64 %2, %10\nv_cvt_f32_f64 %3, %11\nv_cvt_f32_f64 %4, %12\nv_cvt_f32_f64 %5, %13\nv_cvt_f32_f64 %6, %14\nv_cvt_f32_f64 %7, %15\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7)); }
        if (OP == 10) { asm volatile("v_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %9\nv_cvt_f64_f32 %2, %10\nv_cvt_f64_f32 %3, %11\nv_cvt_f64_f32 %4, %12\nv_cvt_f64_f32 %5, %13\nv_cvt_f64_f32 %6, %14\nv_cvt_f64_f32 %7, %15\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7)); }
        if (OP == 11) { F8("v_add_f32") }
        if (OP == 12) { F8("v_mul_f32") }
        if (OP == 13) { asm volatile("v_fma_f32 %0, %0, %8, %0\nv_fma_f32 %1, %1, %8, %1\nv_fma_f32 %2, %2, %8, %2\nv_fma_f32 %3, %3, %8, %3\nv_fma_f32 %4, %4, %8, %4\nv_fma_f32 %5, %5, %8, %5\nv_fma_f32 %6, %6, %8, %6\nv_fma_f32 %7, %7, %8, %7\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)); }
        if (OP == 14) { I8("v_mul_lo_u32") }
        if (OP == 15) { I8("v_add_u32") }
        if (OP == 16) { I8("v_lshlrev_b32") }
        if (OP == 17) { asm volatile("v_lshl_add_u32 %0, %0, 2, %8\nv_lshl_add_u32 %1, %1, 2, %8\nv_lshl_add_u32 %2, %2, 2, %8\nv_lshl_add_u32 %3, %3, 2, %8\nv_lshl_add_u32 %4, %4, 2, %8\nv_lshl_add_u32 %5, %5, 2, %8\nv_lshl_add_u32 %6, %6, 2, %8\nv_lshl_add_u32 %7, %7, 2, %8\n" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0 | 3)); }
        if (OP == 18) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(i0 | 3) : "vcc"); }
        if (OP == 19) { asm volatile("v_cmp_lt_f64 vcc, %0, %8\nv_cmp_lt_f64 vcc, %1, %8\nv_cmp_lt_f64 vcc, %2, %8\nv_cmp_lt_f64 vcc, %3, %8\nv_cmp_lt_f64 vcc, %4, %8\nv_cmp_lt_f64 vcc, %5, %8\nv_cmp_lt_f64 vcc, %6, %8\nv_cmp_lt_f64 vcc, %7, %8\n" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(c) : "vcc"); }
        if (OP == 20) { asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\nv_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)); }
        if (OP == 21) { asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\nv_mad_u64_u32 %0, vcc, %4, %5, %0\nv_mad_u64_u32 %1, vcc, %4, %5, %1\nv_mad_u64_u32 %2, vcc, %4, %5, %2\nv_mad_u64_u32 %3, vcc, %4, %5, %3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1) : "vcc"); }
        if (OP == 22) { F8("v_min_f32") }
        if (OP == 23) { asm volatile("v_cvt_i32_f32 %0, %8\nv_cvt_i32_f32 %1, %9\nv_cvt_i32_f32 %2, %10\nv_cvt_i32_f32 %3, %11\nv_cvt_i32_f32 %4, %12\nv_cvt_i32_f32 %5, %13\nv_cvt_i32_f32 %6, %14\nv_cvt_i32_f32 %7, %15\n" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP>
double run(const char *name, double *d_out, double clock_ghz)
{
    const int blocks = 256 * 4;   // 4 blocks/CU x 4 waves = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0, ITERS * 8);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)blocks * 4 /*waves*/ / (256.0 * 4) * ITERS * 8 * 8;
    const double cyc = ms * 1e-3 * clock_ghz * 1e9 / wave_instr_per_simd;
    printf("%-16s %8.3f ms  %6.2f cycles / wave-instr / SIMD (at %.2f GHz)\n", name, ms, cyc, clock_ghz);
    return cyc;
}

int main()
{
    double *d_out;
    hipMalloc(&d_out, 256 * 4 * 256 * sizeof(double));
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    run<11>("warmup", d_out, ghz);
    run<0>("v_add_f64", d_out, ghz);
    run<1>("v_mul_f64", d_out, ghz);
    run<2>("v_fma_f64", d_out, ghz);
    run<3>("v_floor_f64", d_out, ghz);
    run<4>("v_ceil_f64", d_out, ghz);
    run<5>("v_rndne_f64", d_out, ghz);
    run<6>("v_fract_f64", d_out, ghz);
    run<7>("v_cvt_i32_f64", d_out, ghz);
    run<8>("v_cvt_f64_i32", d_out, ghz);
    run<9>("v_cvt_f32_f64", d_out, ghz);
    run<10>("v_cvt_f64_f32", d_out, ghz);
    run<11>("v_add_f32", d_out, ghz);
    run<12>("v_mul_f32", d_out, ghz);
    run<13>("v_fma_f32", d_out, ghz);
    run<14>("v_mul_lo_u32", d_out, ghz);
    run<15>("v_add_u32", d_out, ghz);
    run<16>("v_lshlrev_b32", d_out, ghz);
    run<17>("v_lshl_add_u32", d_out, ghz);
    run<18>("v_cndmask_b32", d_out, ghz);
    run<19>("v_cmp_lt_f64", d_out, ghz);
    run<20>("v_pk_mul_f32", d_out, ghz);
    run<21>("v_mad_u64_u32", d_out, ghz);
    run<22>("v_min_f32", d_out, ghz);
    run<23>("v_cvt_i32_f32", d_out, ghz);
    return 0;
}
