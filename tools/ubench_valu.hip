// Micro-benchmark: issue cost (shader cycles per wave64 instruction per SIMD) of the VALU ops the ray-march
// inner loop uses, gfx950.  Round 2 rewrite -- answers "does a plain f32 / f64 VALU op issue every 2 or every 4
// cycles?" (MI355X_MICROARCH.md says SIMD-32, 2 cycles per wave64 f32 op; round 1's short 0.2 ms runs said 4):
//   * 16 independent dependency chains per wave (latency can never bound the issue rate);
//   * 1, 2, 4 and 8 waves per SIMD (blocks of 256 threads = one wave per SIMD of a CU, `w` blocks per CU);
//   * cycles are counted IN the kernel with s_memtime (shader clock, independent of DVFS and of launch ramps):
//     every wave records end - start, and the cost is  mean(end - start) / (instructions per wave * w);
//   * long runs (about 10^6 instructions per wave) and the wall-clock figure printed beside it.
// Build & run:  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CH16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

enum Op { ADD_F32, MUL_F32, FMA_F32, MIN_F32, ADD_U32, LSHL_ADD_U32, MUL_I24, CNDMASK, CVT_I32_F32, PK_FMA_F32,
          ADD_F64, MUL_F64, FMA_F64, FLOOR_F64, CEIL_F64, CVT_I32_F64, CVT_F64_I32, CVT_F32_F64, CVT_F64_F32,
          CMP_F64, MIX_MARCH, N_OPS };

static const char *kNames[N_OPS] = {"v_add_f32", "v_mul_f32", "v_fma_f32", "v_min_f32", "v_add_u32", "v_lshl_add_u32",
                                    "v_mul_i32_i24", "v_cndmask_b32", "v_cvt_i32_f32", "v_pk_fma_f32", "v_add_f64",
                                    "v_mul_f64", "v_fma_f64", "v_floor_f64", "v_ceil_f64", "v_cvt_i32_f64",
                                    "v_cvt_f64_i32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_cmp_lt_f64",
                                    "mix: 9 f64 + 7 f32 (march body ratio)"};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *cycles, double *sink, double seed, int iters)
{
    double a[16];
    float f[16];
    int i[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        a[j] = seed + threadIdx.x + j;
        f[j] = (float)a[j];
        i[j] = threadIdx.x + j;
    }
    const double c = seed * 1.000001;
    const float cf = (float)c;
    const int ci = (int)threadIdx.x | 3;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define F2(INS, j) asm volatile(INS " %0, %0, %1" : "+v"(f[j]) : "v"(cf));
#define D2(INS, j) asm volatile(INS " %0, %0, %1" : "+v"(a[j]) : "v"(c));
#define I2(INS, j) asm volatile(INS " %0, %0, %1" : "+v"(i[j]) : "v"(ci));
        if (OP == ADD_F32) {
#define M(j) F2("v_add_f32", j)
            CH16(M)
#undef M
        }
        if (OP == MUL_F32) {
#define M(j) F2("v_mul_f32", j)
            CH16(M)
#undef M
        }
        if (OP == FMA_F32) {
#define M(j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[j]) : "v"(cf));
            CH16(M)
#undef M
        }
        if (OP == MIN_F32) {
#define M(j) F2("v_min_f32", j)
            CH16(M)
#undef M
        }
        if (OP == ADD_U32) {
#define M(j) I2("v_add_u32", j)
            CH16(M)
#undef M
        }
        if (OP == LSHL_ADD_U32) {
#define M(j) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i[j]) : "v"(ci));
            CH16(M)
#undef M
        }
        if (OP == MUL_I24) {
#define M(j) I2("v_mul_i32_i24", j)
            CH16(M)
#undef M
        }
        if (OP == CNDMASK) {
#define M(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i[j]) : "v"(ci) : );
            CH16(M)
#undef M
        }
        if (OP == CVT_I32_F32) {
#define M(j) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i[j]) : "v"(f[j]));
            CH16(M)
#undef M
        }
        if (OP == PK_FMA_F32) {
#define M(j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[j]) : "v"(c));
            CH16(M)
#undef M
        }
        if (OP == ADD_F64) {
#define M(j) D2("v_add_f64", j)
            CH16(M)
#undef M
        }
        if (OP == MUL_F64) {
#define M(j) D2("v_mul_f64", j)
            CH16(M)
#undef M
        }
        if (OP == FMA_F64) {
#define M(j) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[j]) : "v"(c));
            CH16(M)
#undef M
        }
        if (OP == FLOOR_F64) {
#define M(j) asm volatile("v_floor_f64 %0, %0" : "+v"(a[j]));
            CH16(M)
#undef M
        }
        if (OP == CEIL_F64) {
#define M(j) asm volatile("v_ceil_f64 %0, %0" : "+v"(a[j]));
            CH16(M)
#undef M
        }
        if (OP == CVT_I32_F64) {
#define M(j) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i[j]) : "v"(a[j]));
            CH16(M)
#undef M
        }
        if (OP == CVT_F64_I32) {
#define M(j) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(i[j]));
            CH16(M)
#undef M
        }
        if (OP == CVT_F32_F64) {
#define M(j) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j]));
            CH16(M)
#undef M
        }
        if (OP == CVT_F64_F32) {
#define M(j) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[j]) : "v"(f[j]));
            CH16(M)
#undef M
        }
        if (OP == CMP_F64) {
#define M(j) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[j]), "v"(c) : "vcc");
            CH16(M)
#undef M
        }
        if (OP == MIX_MARCH) {  // the executed march body is ~144 f64 : 72 f32/int per group of 4 -> 9 : 7 per 16
            D2("v_mul_f64", 0) D2("v_add_f64", 1) F2("v_mul_f32", 0) D2("v_mul_f64", 2) D2("v_add_f64", 3)
            F2("v_add_f32", 1) asm volatile("v_floor_f64 %0, %0" : "+v"(a[4]));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[2]) : "v"(cf));
            D2("v_add_f64", 5) I2("v_add_u32", 0) D2("v_mul_f64", 6) F2("v_mul_f32", 3)
            asm volatile("v_ceil_f64 %0, %0" : "+v"(a[7]));
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[4]) : "v"(a[8]));
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[5]) : "v"(cf));
            I2("v_mul_i32_i24", 1)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        s += a[j] + (double)f[j] + (double)i[j];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cycles[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
        cycles[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
    }
}

template <int OP>
static void run(unsigned long long *d_cyc, double *d_sink, int cus, double wall_ghz)
{
    const int iters = 1 << 16;  // x 16 instructions = ~1.05 M instructions per wave
    printf("%-40s", kNames[OP]);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = cus * w;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 1.0, 256);  // warm-up
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink, 1.0, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4 * 2);
        hipMemcpy(h.data(), d_cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        double mean = 0.0, mean_real = 0.0;
        for (size_t q = 0; q < h.size(); q += 2) {
            mean += (double)h[q];
            mean_real += (double)h[q + 1];
        }
        mean /= (double)(h.size() / 2);
        mean_real /= (double)(h.size() / 2);
        const double per_instr = mean / ((double)iters * 16.0 * w);
        const double wall_cyc = ms * 1e-3 * wall_ghz * 1e9 / ((double)iters * 16.0 * w);
        printf("  w=%d: %5.2f (wall %5.2f, memtime %4.0f MHz)", w, per_instr, wall_cyc, mean / mean_real * 100.0);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
    }
    printf("\n");
}

template <int OP>
static void run_all(unsigned long long *d_cyc, double *d_sink, int cus, double ghz)
{
    run<OP>(d_cyc, d_sink, cus, ghz);
    if constexpr (OP + 1 < N_OPS)
        run_all<OP + 1>(d_cyc, d_sink, cus, ghz);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    unsigned long long *d_cyc;
    double *d_sink;
    hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 2 * sizeof(unsigned long long));
    hipMalloc(&d_sink, (size_t)cus * 8 * 256 * sizeof(double));
    printf("%s: %d CUs, nominal %.2f GHz.  Columns: s_memtime cycles per wave64 instruction per SIMD at w waves/SIMD\n"
           "(16 independent chains per wave, ~1.05 M instructions per wave; 'wall' = the same from the event time at the\n"
           "nominal clock -- larger when the chip clocks below nominal); 'memtime MHz' = s_memtime ticks per second, from the\n"
           "constant 100 MHz s_memrealtime counter read beside it.\n", prop.name, cus, ghz);
    run_all<0>(d_cyc, d_sink, cus, ghz);
    return 0;
}
