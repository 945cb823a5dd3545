// Developer probe (never part of the library): what HBM rate does the decode kernel's ACCESS PATTERN reach when nothing but
// the memory system is in the way?  One workgroup per "frame" (its own 8100 x 1 KiB row block, as the APP rows of the GF(256)
// headline code), waves of 64 lanes in 8-lane groups; per step a wave reads 8 rows picked pseudo-randomly inside its frame with
// the loads of load_row_nt (8 x 16 B per lane, non-temporal), optionally 8 records of 112 B, and writes the rows (and records)
// back -- the traffic of one two-visit wave step, with no arithmetic between.  LDS is requested like the decode kernel's so the
// same number of workgroups shares a CU.
// build: hipcc -O3 --offload-arch=gfx950 -o build/row_traffic csrc/probes/row_traffic.hip ; run: build/row_traffic [frames] [waves] [lds_kib] [steps] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 8100, ROW_F = 256, REC_B = 112, NREC = 16200;

typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ v4f ld(const v4f *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(v4f *p, v4f v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// mode bit 0: non-temporal, bit 1: records too, bit 2: sequential rows instead of random, bits 4-5: log2 of adjacent rows per pick, bit 3: no row writes, bit 6: no row reads
template <bool NT> __global__ void __launch_bounds__(256) k_rows(float *app, uint8_t *recs, int steps, int mode)
{
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rl = lane & 7, row = lane >> 3;
    float *fa = app + (size_t)blockIdx.x * ROWS * ROW_F;
    uint8_t *fr = recs + (size_t)blockIdx.x * NREC * REC_B;
    unsigned s = blockIdx.x * 2654435761u + wave * 40503u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < steps; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned r;
        if (mode & 4) r = ((unsigned)(it * nw + wave) * 8u + row) % ROWS;
        else {
            const int chunk = 1 << ((mode >> 4) & 3);        // mode bits 4-5: 1, 2, 4 or 8 adjacent rows per pick
            r = (((s >> 8) + (row / chunk) * 1009u) % (ROWS / chunk)) * chunk + row % chunk;   // 8 rows per wave step (same for the 8 lanes of a group)
        }
        r = __shfl(r, lane & ~7);
        v4f *p = reinterpret_cast<v4f *>(fa + (size_t)r * ROW_F) + rl;
        v4f x[8];
        if (!(mode & 64)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = ld<NT>(p + k * 8);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = v4f{(float)it, (float)k, (float)lane, 1.0f};
        }
        float rv[4] = {0, 0, 0, 0};
        uint8_t *rp = fr + (size_t)((r * 2u + (it & 1)) % NREC) * REC_B;
        if (mode & 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int k = rl + c * 8; rv[c] = reinterpret_cast<const float *>(rp)[k < 20 ? k : 0]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k].x += 1.0f; acc += x[k].y; }
        if (!(mode & 8)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) st<NT>(p + k * 8, x[k]);
        }
        if (mode & 2) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { const int k = rl + c * 8; if (k < 20) reinterpret_cast<float *>(rp)[k] = rv[c] + 1.0f; }
            if (rl < 5) reinterpret_cast<float *>(rp + 80)[rl] = acc;      // the 28 trailing bytes (default, hint, count, symbols)
        }
    }
    if (acc == 12345.678f) smem[0] = 1;                    // keep the loads
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 1536, waves = argc > 2 ? atoi(argv[2]) : 2, lds_kib = argc > 3 ? atoi(argv[3]) : 26;
    const int steps = argc > 4 ? atoi(argv[4]) : 4000, mode = argc > 5 ? atoi(argv[5]) : 3;
    float *app; uint8_t *recs;
    const size_t ab = (size_t)frames * ROWS * ROW_F * 4, rb = (size_t)frames * NREC * REC_B;
    CK(hipMalloc(&app, ab)); CK(hipMalloc(&recs, rb));
    CK(hipMemset(app, 0, ab)); CK(hipMemset(recs, 0, rb));
    auto kern = (mode & 1) ? k_rows<true> : k_rows<false>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kib * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(frames), dim3(waves * 64), lds_kib * 1024, 0, app, recs, steps, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_step = 8.0 * 1024 * ((mode & 8) ? 1 : 2) * ((mode & 64) ? 0.5 : 1.0) + ((mode & 2) ? 8.0 * REC_B * 2 : 0.0);
        const double bytes = (double)frames * waves * steps * per_step;
        printf("frames %d waves/wg %d lds %d KiB steps %d mode %d: %.2f ms, %.1f GB/s (read+write, algorithmic)\n", frames, waves, lds_kib, steps, mode, ms, bytes / ms / 1e6);
    }
    return 0;
}
