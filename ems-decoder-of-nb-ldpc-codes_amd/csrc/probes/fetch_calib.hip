// Developer probe (never part of the library): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the decode kernels' ACCESS WIDTHS?
// MI355X_MICROARCH.md calibrates FETCH_SIZE on wide coalesced streaming reads only (gfx950 reports half their bytes) and says "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute".  The GF(16) kernel
// (k_decode_q16t) reads and writes one 64-byte APP row PER LANE (four 16-byte accesses at a 64-byte-aligned address, rows scattered over the
// frame); the headline kernel moves 1 KiB rows, 8 lanes x 128 contiguous bytes.  Each mode below is a kernel of its own (one launch each,
// names k_calib<MODE>), touches every byte it counts exactly once in a region far larger than L2 + the 256 MiB Infinity Cache, and prints the
// bytes it knows it moved; scripts/pmc_calib.sh puts the counters next to them.
//   0  streaming read, 16 B per lane, fully coalesced                              (the guide's calibration point: expect 0.5)
//   1  one isolated 64-B row per lane = the FIRST half of a random 128-B line, the other half never touched
//   2  both halves of a random 128-B line by one lane, back to back (8 x 16 B)
//   3  the first halves of a workgroup's 256 random lines, a barrier, then their second halves (one workgroup per CU: the lines are still
//      in L2 if they came in whole): does the first access bring the whole 128-B line in?
//   4  one 1-KiB row per 8 lanes (8 x 16 B per lane, 128 B contiguous per lane group and k): the headline kernel's row access
//   5  streaming write, 16 B per lane                                              (the guide: WRITE_SIZE exact)
//   6  one isolated 64-B row WRITTEN per lane (first half of a random line)
//   7  both halves of a random line written by one lane
// build: make -C ems-decoder-of-nb-ldpc-codes_amd/csrc probes ; run: build/fetch_calib [log2 lanes, default 24]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// a permutation of [0, 2^bits): odd multiplier, then a xor-shift (both bijective modulo 2^bits)
__device__ __forceinline__ unsigned perm(unsigned i, int bits)
{
    const unsigned m = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    unsigned x = (i * 2654435761u) & m;
    x ^= x >> (bits / 2);
    return (x * 40503u * 2u + x) & m;      // x * (2 * 40503 + 1): odd
}

template <int MODE> __global__ void __launch_bounds__(256) k_calib(float *buf, float *sink, int bits)
{
    const unsigned t = blockIdx.x * 256u + threadIdx.x;           // one "lane unit" of work; grid covers 2^bits of them
    v4f *b = reinterpret_cast<v4f *>(buf);                          // 16-byte units; a 128-byte line = 8 of them
    float acc = 0.f;
    if (MODE == 0) {
        const v4f x = b[t];
        acc = x.x + x.w;
    } else if (MODE == 1) {
        const size_t line = perm(t, bits);
#pragma unroll
        for (int k = 0; k < 4; ++k) { const v4f x = b[line * 8 + k]; acc += x.x; }
    } else if (MODE == 2) {
        const size_t line = perm(t, bits);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const v4f x = b[line * 8 + k]; acc += x.x; }
    } else if (MODE == 3) {
        // the workgroup's 256 lines (32 KiB): all first halves, a barrier, all second halves.  Launched with 96 KiB of LDS = one workgroup
        // per CU, so that an XCD's 32 CUs have 1 MiB in flight between the two accesses of a line -- well inside its 4 MiB L2
        const size_t line = perm(t, bits);
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const v4f x = b[line * 8 + half * 4 + k]; acc += x.x; }
            if (acc == 654321.0f) sink[1] = acc;                    // (the loads of a half complete before the barrier)
            __syncthreads();
        }
    } else if (MODE == 4) {
        const size_t row = perm(t >> 3, bits - 3);                  // 1 KiB rows: 64 units; lane group of 8
        const int rl = t & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const v4f x = b[row * 64 + k * 8 + rl]; acc += x.x; }
    } else if (MODE == 5) {
        b[t] = v4f{(float)t, 1.f, 2.f, 3.f};
    } else if (MODE == 6) {
        const size_t line = perm(t, bits);
#pragma unroll
        for (int k = 0; k < 4; ++k) b[line * 8 + k] = v4f{(float)t, (float)k, 2.f, 3.f};
    } else if (MODE == 7) {
        const size_t line = perm(t, bits);
#pragma unroll
        for (int k = 0; k < 8; ++k) b[line * 8 + k] = v4f{(float)t, (float)k, 2.f, 3.f};
    }
    if (acc == 123456.789f) sink[t & 255] = acc;                    // keeps the loads alive, never true in practice
}

template <int MODE> static void run(float *buf, float *sink, int bits, double known, const char *what)
{
    const unsigned lanes = 1u << bits;
    const unsigned blocks = lanes / 256u;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    if (MODE == 3) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_calib<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    k_calib<MODE><<<blocks, 256, MODE == 3 ? 96 * 1024 : 0>>>(buf, sink, bits);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d  known_bytes %.0f  %.3f ms  %.1f GB/s  %s\n", MODE, known, ms, known / ms * 1e-6, what);
}

int main(int argc, char **argv)
{
    const int bits = argc > 1 ? atoi(argv[1]) : 24;                 // 2^24 lanes: 2 GiB of 128-byte lines
    const size_t bytes = (size_t)128 << bits;
    float *buf, *sink;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&sink, 1024));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    const double L = (double)(1u << bits);
    run<0>(buf, sink, bits, L * 16, "streaming read 16 B per lane");
    run<1>(buf, sink, bits, L * 64, "isolated 64-B row per lane (first half of a random line)");
    run<2>(buf, sink, bits, L * 128, "both halves of a random line by one lane");
    run<3>(buf, sink, bits, L * 128, "first halves of 256 lines per workgroup, barrier, their second halves (1 workgroup per CU)");
    run<4>(buf, sink, bits, L * 128, "1-KiB rows, 8 lanes x 128 B");
    run<5>(buf, sink, bits, L * 16, "streaming write 16 B per lane");
    run<6>(buf, sink, bits, L * 64, "isolated 64-B row written per lane");
    run<7>(buf, sink, bits, L * 128, "both halves of a random line written by one lane");
    CK(hipFree(buf)); CK(hipFree(sink));
    return 0;
}
