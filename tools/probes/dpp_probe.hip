// tools/probes/dpp_probe.hip -- does v_fmac_f64_dpp row_newbcast broadcast lane N of each 16-lane row, and at what rate?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/dpp_probe.hip -o gpurun_out/dpp_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define DPPF(N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc[(N) & 3]) : "v"(wv), "v"(xv))

__global__ void k_dpp(const double* w, const double* x, double* out, int iters)
{
    double wv = w[threadIdx.x & 15], xv = x[threadIdx.x & 63];
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
        asm volatile("s_nop 1");
        DPPF(0); DPPF(1); DPPF(2); DPPF(3); DPPF(4); DPPF(5); DPPF(6); DPPF(7);
        DPPF(8); DPPF(9); DPPF(10); DPPF(11); DPPF(12); DPPF(13); DPPF(14); DPPF(15);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ void k_sgpr(const double* w, const double* x, double* out, int iters)
{
    double xv = x[threadIdx.x & 63];
    double ws[16];
    for (int j = 0; j < 16; j++) ws[j] = w[j];
    double acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) { acc[j & 3] = fma(ws[j], xv, acc[j & 3]); asm volatile("" : "+v"(acc[j & 3])); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    const int nb = 256 * 8, nt = 256, iters = 20000;
    std::vector<double> hw(16), hx(64);
    for (int j = 0; j < 16; j++) hw[j] = 1.0 + 0.125 * j;
    for (int j = 0; j < 64; j++) hx[j] = 0.5 + 0.001 * j;
    double *dw, *dx, *dout;
    hipMalloc(&dw, 16 * 8); hipMalloc(&dx, 64 * 8); hipMalloc(&dout, (size_t)nb * nt * 8);
    hipMemcpy(dw, hw.data(), 16 * 8, hipMemcpyHostToDevice); hipMemcpy(dx, hx.data(), 64 * 8, hipMemcpyHostToDevice);
    std::vector<double> o1(nt), o2(nt);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_dpp, dim3(nb), dim3(nt), 0, 0, dw, dx, dout, iters);
            else hipLaunchKernelGGL(k_sgpr, dim3(nb), dim3(nt), 0, 0, dw, dx, dout, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 16 * iters * (double)nb * nt;
            if (rep) std::printf("%s: %.3f ms  %.2f TFLOP/s\n", which == 0 ? "dpp " : "sgpr", ms, fl / ms * 1e-9);
        }
        hipMemcpy(which == 0 ? o1.data() : o2.data(), dout, nt * 8, hipMemcpyDeviceToHost);
    }
    // one-iteration check of the broadcast semantics
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, dw, dx, dout, 1);
    hipMemcpy(o1.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) {
        double acc[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; j++) acc[j & 3] = std::fma(hw[j], hx[t], acc[j & 3]);
        const double ref = acc[0] + acc[1] + acc[2] + acc[3];
        if (ref != o1[t]) { if (bad < 4) std::printf("lane %d: %.17g vs %.17g\n", t, o1[t], ref); bad++; }
    }
    std::printf("broadcast check: %s (%d lanes differ)\n", bad ? "FAIL" : "ok", bad);
    return bad != 0;
}
