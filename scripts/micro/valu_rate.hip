// Issue cost of f64 VALU on gfx950: N independent chains of v_fma_f64 / v_mul_f64 / v_rcp_f64 /
// v_mov_b64 per wave, W waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ __launch_bounds__(64) void k(double* out, unsigned long long* cyc, int iters) {
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // fma f64
            a0 = fma(a0, m, c); a1 = fma(a1, m, c); a2 = fma(a2, m, c); a3 = fma(a3, m, c);
            a4 = fma(a4, m, c); a5 = fma(a5, m, c); a6 = fma(a6, m, c); a7 = fma(a7, m, c);
        } else if (OP == 1) {  // rcp f64
            a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3);
            a4 = __builtin_amdgcn_rcp(a4); a5 = __builtin_amdgcn_rcp(a5); a6 = __builtin_amdgcn_rcp(a6); a7 = __builtin_amdgcn_rcp(a7);
        } else if (OP == 2) {  // fma f32
            float b0 = (float)a0, b1 = (float)a1, b2 = (float)a2, b3 = (float)a3, b4 = (float)a4, b5 = (float)a5, b6 = (float)a6, b7 = (float)a7;
            for (int j = 0; j < 1; ++j) {
                b0 = fmaf(b0, 1.0000001f, 1e-9f); b1 = fmaf(b1, 1.0000001f, 1e-9f); b2 = fmaf(b2, 1.0000001f, 1e-9f); b3 = fmaf(b3, 1.0000001f, 1e-9f);
                b4 = fmaf(b4, 1.0000001f, 1e-9f); b5 = fmaf(b5, 1.0000001f, 1e-9f); b6 = fmaf(b6, 1.0000001f, 1e-9f); b7 = fmaf(b7, 1.0000001f, 1e-9f);
            }
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
        } else if (OP == 3) {  // dependent chain fma f64
            a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
            a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c); a0 = fma(a0, m, c);
        } else if (OP == 4) {  // mul hi u32
            unsigned u0 = (unsigned)a0, u1 = (unsigned)a1, u2 = (unsigned)a2, u3 = (unsigned)a3;
            u0 = __umulhi(u0, 0xD2511F53u) ^ u1; u1 = __umulhi(u1, 0xCD9E8D57u) ^ u2; u2 = __umulhi(u2, 0xD2511F53u) ^ u3; u3 = __umulhi(u3, 0xCD9E8D57u) ^ u0;
            u0 = __umulhi(u0, 0xD2511F53u) ^ u1; u1 = __umulhi(u1, 0xCD9E8D57u) ^ u2; u2 = __umulhi(u2, 0xD2511F53u) ^ u3; u3 = __umulhi(u3, 0xCD9E8D57u) ^ u0;
            a0 = u0; a1 = u1; a2 = u2; a3 = u3;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP>
void run(const char* name, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd, iters = 20000;
    double* out; unsigned long long* cyc;
    (void)hipMalloc(&out, blocks * 64 * 8); (void)hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(out, cyc, 100);
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    // wave-instructions per SIMD = waves_per_simd * iters * 8
    const double n = (double)waves_per_simd * iters * 8;
    printf("%-14s waves/SIMD %d: kernel %.3f ms -> %.2f ns per wave-instr per SIMD; in-wave s_memtime cycles/instr %.2f\n",
           name, waves_per_simd, ms, ms * 1e6 / n, mean / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma_f64", w); run<1>("rcp_f64", w); run<2>("fma_f32(+cvt)", w); run<3>("fma_f64 dep", w); run<4>("mulhi_u32", w);
    }
    return 0;
}
