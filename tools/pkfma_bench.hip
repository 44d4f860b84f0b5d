// Issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_mov_b64) on gfx950: tools/pkfma_bench.hip
// build: hipcc -O2 --offload-arch=gfx950 -o tools/bin/pkfma_bench tools/pkfma_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s0, float s1, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {s0, s0}, c = {s1, s1};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0), "v"(s1));)
        } else if (MODE == 1) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q), "v"(c));)
        } else {  // scalar operand forms as the packet kernel would use them: sgpr pair x vgpr (op_sel broadcast) + vgpr pair
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[1,0,1]\n"
                         "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[1,0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(q), "v"(c));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, float* d, int waves_per_simd) {
    const int iters = 20000, blocks = 256 * waves_per_simd;  // 256 CUs x 4 waves per block = 1 wave per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 0.999f, 0.001f, 10);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 0.999f, 0.001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = 64.0 * iters;
    printf("%-28s %d waves/SIMD: %.3f ms, %.2f ns per instr per wave-slot => %.2f cycles/instr/SIMD at 2.4 GHz\n", name, waves_per_simd, ms,
           ms * 1e6 / (instr_per_wave * waves_per_simd), ms * 1e6 / (instr_per_wave * waves_per_simd) * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4);
    for (int w : {1, 2, 4}) { run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32 vgpr", d, w); run<2>("v_pk_fma_f32 sgpr,opsel", d, w); }
    return 0;
}
