// tools/probes/fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS repository's access
// patterns (MI355X_MICROARCH.md, section HBM: FETCH_SIZE reports half of a 16 B/lane streaming read; "other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel here moves a KNOWN number of bytes
// of a buffer far larger than the 256 MiB Infinity Cache, once, so counter value x factor = bytes gives the factor.
//   rd4 / rd8 / rd16   : streaming reads, 4 / 8 / 16 B per lane, each wavefront instruction one contiguous run
//   rows8              : the strip kernels' staging pattern -- every wavefront loads 512-byte row segments (8 B per lane) that start
//                        at arbitrary 8-byte offsets of rows 4104 bytes apart (partial 128-byte lines at both ends, neighbouring
//                        wavefronts overlap by 40 %)
//   tile32             : the block products' operand pattern -- 32 B per lane (two dwordx4), 2 KiB contiguous per wavefront
//   wr8 / wr16         : streaming stores, 8 / 16 B per lane
// Usage (on the GPU box; tools/calibrate_pmc.sh):  rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib.bin ; rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib.bin
// The program prints the bytes each kernel moved; tools/pmc_calibrate.py divides them by the counters.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// each thread reads `per` consecutive chunks strided by the grid: every wavefront instruction covers one contiguous run
template <class T>
__global__ __launch_bounds__(256) void calib_read(const T* __restrict__ src, size_t n, float* __restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T acc = {};
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const T v = src[k];
        if constexpr (sizeof(T) == 16) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
        else acc += v;
    }
    float s;
    if constexpr (sizeof(T) == 16) s = acc.x + acc.y + acc.z + acc.w; else s = (float)acc;
    if (s == 12345.678f) sink[0] = s;   // never true on the zero-filled buffer: keeps the loads
}

template <class T>
__global__ __launch_bounds__(256) void calib_write(T* __restrict__ dst, size_t n, T v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) dst[k] = v;
}

// the strip kernels' staging: workgroup b, wavefront w loads, for r = 0 .. rows-1, the 64 doubles [row r][c0 + lane] with
// c0 = 1 + 57 * b' (b' = position of the group in its row band) -- rows `ld` doubles apart.  Distinct bands per 9 groups, so a band of
// `rows` x ld doubles is read by 9 overlapping groups exactly like one strip launch reads a sequence's table rows.
__global__ __launch_bounds__(512) void calib_rows8(const double* __restrict__ src, int ld, int rows, int groups_per_band, double* __restrict__ sink)
{
    const int band = blockIdx.x / groups_per_band, g = blockIdx.x % groups_per_band;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double* base = src + (size_t)band * rows * ld;
    double acc = 0.0;
    for (int r = w; r < rows; r += 8) {
        acc += base[(size_t)r * ld + 1 + 57 * g + lane];
        if (lane < 32) acc += base[(size_t)r * ld + 1 + 57 * g + 64 + lane];   // the 32-column tail of the 96-column window
    }
    if (acc == 12345.678) sink[0] = acc;
}
// bytes a band's groups touch at line granularity are what HBM must deliver at least once: rows x (1 + 57*(G-1) + 96) doubles

__global__ __launch_bounds__(256) void calib_tile32(const f4* __restrict__ src, size_t ntiles, float* __restrict__ sink)
{
    // one wavefront per 2 KiB tile: lane l reads bytes [32 l, 32 l + 32) as two dwordx4
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    f4 acc = {};
    for (size_t t = wave; t < ntiles; t += nw) {
        const f4 a = src[t * 128 + lane * 2], b = src[t * 128 + lane * 2 + 1];
        acc += a + b;
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) sink[0] = s;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: 8 x the Infinity Cache
    void* buf = nullptr;
    float* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    hipLaunchKernelGGL(calib_read<float>, dim3(grid), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_read<float> bytes_read=%zu\n", bytes);
    hipLaunchKernelGGL(calib_read<double>, dim3(grid), dim3(256), 0, 0, (const double*)buf, bytes / 8, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_read<double> bytes_read=%zu\n", bytes);
    hipLaunchKernelGGL(calib_read<f4>, dim3(grid), dim3(256), 0, 0, (const f4*)buf, bytes / 16, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_read<f4> bytes_read=%zu\n", bytes);
    // the same 16 B/lane streaming read from a base that is only 8-byte aligned (are unaligned dwordx4 loads served at the aligned rate?)
    hipLaunchKernelGGL(calib_read<f4>, dim3(grid), dim3(256), 0, 0, (const f4*)((const char*)buf + 8), bytes / 16 - 1, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_read<f4>+8 bytes_read=%zu\n", bytes - 16);
    {
        const int ld = 513, rows = 32, gpb = 8;   // a band = 32 rows of 513 doubles, read by 8 overlapping 96-column windows
        const size_t band_doubles = (size_t)rows * ld;
        const int bands = (int)(bytes / 8 / band_doubles);
        hipLaunchKernelGGL(calib_rows8, dim3(bands * gpb), dim3(512), 0, 0, (const double*)buf, ld, rows, gpb, (double*)sink);
        CHECK(hipDeviceSynchronize());
        const size_t requested = (size_t)bands * gpb * rows * 96 * 8, unique = (size_t)bands * rows * (1 + 57 * (gpb - 1) + 96) * 8;
        printf("calib_rows8 bytes_requested=%zu bytes_unique=%zu\n", requested, unique);
    }
    hipLaunchKernelGGL(calib_tile32, dim3(grid), dim3(256), 0, 0, (const f4*)buf, bytes / 2048, sink);
    CHECK(hipDeviceSynchronize());
    printf("calib_tile32 bytes_read=%zu\n", bytes);
    hipLaunchKernelGGL(calib_write<double>, dim3(grid), dim3(256), 0, 0, (double*)buf, bytes / 8, 0.0);
    CHECK(hipDeviceSynchronize());
    printf("calib_write<double> bytes_written=%zu\n", bytes);
    hipLaunchKernelGGL(calib_write<f4>, dim3(grid), dim3(256), 0, 0, (f4*)buf, bytes / 16, f4{0, 0, 0, 0});
    CHECK(hipDeviceSynchronize());
    printf("calib_write<f4> bytes_written=%zu\n", bytes);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
