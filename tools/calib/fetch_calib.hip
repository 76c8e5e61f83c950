// fetch_calib.hip -- what FETCH_SIZE reports on gfx950 for the access shapes of the tile kernel.
//
// rocprofv3's FETCH_SIZE counts the L2's read requests to memory at 64 bytes each.  A wide streaming
// read is known to come out at half its bytes (the requests are 128 bytes long).  The tile kernel
// mostly gathers: 16-byte texture footprints, 48-byte shading records, 32-byte raster records, one
// per lane, from hundreds of megabytes.  Whether a gathered 16 bytes costs memory a 64-byte or a
// 128-byte request decides whether the counter has to be doubled for them as well.  Each kernel
// here touches a known number of distinct 128-byte lines and 64-byte half lines of a 2 GiB buffer
// (eight times the last-level cache), so the counter can be priced against both.
//
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint64_t kBytes = 2ull << 30;
constexpr uint32_t kLines = (uint32_t)(kBytes / 128); // 2^24
constexpr uint32_t kMul = 0x9E3779B1u;                // odd: i -> i * kMul mod 2^24 is a permutation of the lines

__device__ inline void keep(uint4 v, uint32_t *sink)
{
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) *sink = v.x; // (never true for the zero-filled buffer's neighbours; keeps the loads)
}

// every byte once, 16 bytes per lane, lanes side by side
__global__ void stream16(const uint4 *p, uint32_t *sink)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    keep(p[i], sink);
}
// one 16-byte read from every line, lines in scattered order: a lane's line is nobody's neighbour
__global__ void gather16_one_per_line(const uint4 *p, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = (i * kMul) & (kLines - 1);
    keep(p[(uint64_t)line * 8], sink);
}
// the same lines, both halves: 16 bytes at offset 0 and 16 bytes at offset 64
__global__ void gather16_both_halves(const uint4 *p, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = (i * kMul) & (kLines - 1);
    const uint4 a = p[(uint64_t)line * 8], b = p[(uint64_t)line * 8 + 4];
    keep(make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w), sink);
}
// 48 bytes (three 16-byte loads) starting at byte 16 of every second line: one half line, then the next
__global__ void gather48_in_line(const uint4 *p, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = ((i * kMul) & (kLines / 2 - 1)) * 2;
    const uint4 *q = p + (uint64_t)line * 8 + 1;
    const uint4 a = q[0], b = q[1], c = q[2];
    keep(make_uint4(a.x ^ b.x ^ c.x, a.y ^ b.y ^ c.y, a.z ^ b.z ^ c.z, a.w ^ b.w ^ c.w), sink);
}
// 32 bytes (two 16-byte loads) at the start of every second line: one half line
__global__ void gather32_half_line(const uint4 *p, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t line = ((i * kMul) & (kLines / 2 - 1)) * 2;
    const uint4 a = p[(uint64_t)line * 8], b = p[(uint64_t)line * 8 + 1];
    keep(make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w), sink);
}

int main()
{
    uint4 *buf = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc((void **)&buf, kBytes));
    CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(buf, 0, kBytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct Case { const char *name; void (*k)(const uint4 *, uint32_t *); uint64_t threads, lines, halves, asked; };
    const Case cases[] = {
        {"stream16", stream16, kBytes / 16, kLines, 2ull * kLines, kBytes},
        {"gather16_one_per_line", gather16_one_per_line, kLines, kLines, kLines, 16ull * kLines},
        {"gather16_both_halves", gather16_both_halves, kLines, kLines, 2ull * kLines, 32ull * kLines},
        {"gather48_in_line", gather48_in_line, kLines / 2, kLines / 2, kLines, 48ull * (kLines / 2)},
        {"gather32_half_line", gather32_half_line, kLines / 2, kLines / 2, kLines / 2, 32ull * (kLines / 2)},
    };
    printf("{\n \"buffer_bytes\": %llu,\n \"cases\": {\n", (unsigned long long)kBytes);
    for (size_t c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
        const Case &k = cases[c];
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) { // three launches each: the counter pass averages them
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k.k, dim3((uint32_t)(k.threads / 256)), dim3(256), 0, 0, buf, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  \"%s\": {\"ms\": %.4f, \"lines_128B\": %llu, \"half_lines_64B\": %llu, \"bytes_asked\": %llu}%s\n", k.name, best,
               (unsigned long long)k.lines, (unsigned long long)k.halves, (unsigned long long)k.asked,
               c + 1 < sizeof(cases) / sizeof(cases[0]) ? "," : "");
    }
    printf(" }\n}\n");
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
