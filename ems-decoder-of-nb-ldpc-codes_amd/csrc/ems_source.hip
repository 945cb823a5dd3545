// ems_source.hip -- device-side frame source: information symbols, systematic
// encoder and BPSK/AWGN samples generated in HBM, so that a Monte-Carlo run needs no host work and no PCIe
// traffic per frame.
//
// What it replaces on the reference side: RandomBinaryGenerator (tools.c:124), Encoding (tools.c:232) and the
// noise draw of ModelChannel_AWGN_BPSK (channel.c:52-62).  The ENCODER is the reference's algorithm (back
// substitution on the upper-triangular matUT produced by GaussianElimination, tools.c:151, then the column
// permutation Perm) and is bit-exact against it.  The RANDOM STREAM is not the reference's: libc's drand48 is
// one sequential 48-bit LCG, and the reference's Box-Muller runs in double precision through libm, neither of
// which a GPU reproduces bit for bit at any useful rate.  This source uses a counter-based generator
// (Philox4x32-10, Salmon et al., SC'11) keyed by (seed) and indexed by (frame, draw), so frame f is the same
// whatever the batch size, rank count or launch order -- statistical, not bit, parity with the reference's
// stream (the bit-exact stream stays available on the host: ems_draw_frames).
//
// Layout: symbols are kept frame-minor, sym[n][Bs] u8 (n in matUT column order, Bs = batch stride), so that
// a wavefront = 64 frames reads one 64-byte line per matrix entry; the matrix itself (CSR split into the
// triangular part, columns < M, and the information part, columns >= M) is wave-uniform and comes through
// scalar loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ems_kernels.h"

namespace {

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10: 10 rounds of two 32x32->64 multiplies (0xD2511F53, 0xCD9E8D57) with the key bumped by the
// Weyl constants 0x9E3779B9 / 0xBB67AE85 between rounds.
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// GF(q) in the reference's symbol-index representation (init.c:37-100): 0 is zero, index i >= 1 is alpha^(i-1);
// MULGF/DIVGF are exponent arithmetic, ADDGF is XOR of the binary images.
__device__ __forceinline__ int gf_mul_idx(int a, int b, int qm1)
{
    int e = a + b - 2;
    e = e >= qm1 ? e - qm1 : e;
    return (a && b) ? e + 1 : 0;
}
__device__ __forceinline__ int gf_div_idx(int a, int b, int qm1)   // b != 0
{
    int e = a - b;
    e = e < 0 ? e + qm1 : e;
    return a ? e + 1 : 0;
}

// stream 0: information symbols.  Block j of frame f gives symbols 4j..4j+3; the top logq bits of a word are
// the binary image (uniform over the q images, like logq fair bits of RandomBinaryGenerator).
__global__ void k_src_info(uint64_t seed, uint64_t first_frame, int B, int Bs, int M, int K, int logq,
                           const uint8_t *__restrict__ bin2idx, uint8_t *__restrict__ sym)
{
    const int nblk = (K + 3) >> 2;
    const size_t total = (size_t)nblk * B;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(t % B), j = (int)(t / B);
        const uint64_t fr = first_frame + (uint64_t)f;
        const U4 r = philox4x32_10(U4{(uint32_t)j, 0u, (uint32_t)fr, (uint32_t)(fr >> 32)}, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * j + i;
            if (k < K) sym[(size_t)(M + k) * Bs + f] = bin2idx[w[i] >> (32 - logq)];
        }
    }
}

// rhs[m][f] = binary image of sum over the information columns of row m (the n >= M part of tools.c:241-247)
__global__ void k_src_rhs(int B, int Bs, int qm1, const int *__restrict__ rptr, const int *__restrict__ rcol,
                          const uint8_t *__restrict__ rcoef, const uint8_t *__restrict__ idx2bin,
                          const uint8_t *__restrict__ sym, uint8_t *__restrict__ rhs)
{
    __shared__ uint8_t i2b[256];
    for (int g = threadIdx.x; g <= qm1; g += blockDim.x) i2b[g] = idx2bin[g];
    __syncthreads();
    const int m = blockIdx.x;
    const int e0 = rptr[m], e1 = rptr[m + 1];
    for (int f = blockIdx.y * blockDim.x + threadIdx.x; f < B; f += gridDim.y * blockDim.x) {
        unsigned acc = 0;
        for (int e = e0; e < e1; ++e) acc ^= i2b[gf_mul_idx(rcoef[e], sym[(size_t)rcol[e] * Bs + f], qm1)];
        rhs[(size_t)m * Bs + f] = (uint8_t)acc;
    }
}

// back substitution over the triangular part, rows M-1..0 (tools.c:239-250), one lane per frame: the rows are
// a dependent chain, the frames are not.  sym is read back by the thread that wrote it (program order).
__global__ void k_src_solve(int B, int Bs, int M, int qm1, const int *__restrict__ tptr, const int *__restrict__ tcol,
                            const uint8_t *__restrict__ tcoef, const uint8_t *__restrict__ diag,
                            const uint8_t *__restrict__ idx2bin, const uint8_t *__restrict__ bin2idx,
                            const uint8_t *__restrict__ rhs, uint8_t *sym)
{
    __shared__ uint8_t i2b[256], b2i[256];
    for (int g = threadIdx.x; g <= qm1; g += blockDim.x) { i2b[g] = idx2bin[g]; b2i[g] = bin2idx[g]; }
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= B) return;
    unsigned nxt = rhs[(size_t)(M - 1) * Bs + f];
    for (int m = M - 1; m >= 0; --m) {
        unsigned acc = nxt;
        if (m > 0) nxt = rhs[(size_t)(m - 1) * Bs + f];             // independent of the chain: issued a row ahead
        for (int e = tptr[m]; e < tptr[m + 1]; ++e) acc ^= i2b[gf_mul_idx(tcoef[e], sym[(size_t)tcol[e] * Bs + f], qm1)];
        sym[(size_t)m * Bs + f] = (uint8_t)gf_div_idx(b2i[acc], diag[m], qm1);
    }
}

// CodeWord[Perm[n]] = sym[n] (tools.c:252; walked by codeword position through the inverse permutation), then
// noisy = (1 - 2 bit) + sigma * gauss per bit of the binary image (channel.c:52-62 with the device stream).  Stream 1, block j of frame f gives samples 2j and 2j+1
// (sample index = codeword position * logq + bit); u in (0,1], v in [0,1) on a 2^-24 grid; Box-Muller in f32.
__device__ __forceinline__ float box_muller(uint32_t wu, uint32_t wv)
{
    const float u = (float)((wu >> 8) + 1u) * 5.9604644775390625e-08f;
    const float v = (float)(wv >> 8) * 5.9604644775390625e-08f;
    return sqrtf(-2.0f * logf(u)) * cosf(6.28318530717958647692f * v);
}

__global__ void k_src_channel(uint64_t seed, uint64_t first_frame, int B, int Bs, int N, int logq, float sigma,
                              const int *__restrict__ iperm, const uint8_t *__restrict__ idx2bin,
                              const uint8_t *__restrict__ sym, int *__restrict__ cw, float *__restrict__ noisy)
{
    const size_t total = (size_t)B * N;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(t / N), pos = (int)(t % N);       // consecutive lanes = consecutive codeword positions: coalesced stores
        const int n = iperm[pos];
        const int s = sym[(size_t)n * Bs + f];
        cw[(size_t)f * N + pos] = s;
        const unsigned img = idx2bin[s];
        const uint64_t fr = first_frame + (uint64_t)f;
        float *out = noisy + ((size_t)f * N + pos) * logq;
        uint32_t have = 0xFFFFFFFFu;
        U4 r{};
        for (int l = 0; l < logq; ++l) {
            const uint32_t i = (uint32_t)pos * (uint32_t)logq + (uint32_t)l;
            if ((i >> 1) != have) {
                have = i >> 1;
                r = philox4x32_10(U4{have, 1u, (uint32_t)fr, (uint32_t)(fr >> 32)}, (uint32_t)seed, (uint32_t)(seed >> 32));
            }
            const float g = (i & 1u) ? box_muller(r.z, r.w) : box_muller(r.x, r.y);
            const float tx = (float)(1 - 2 * (int)((img >> l) & 1u));
            out[l] = tx + sigma * g;
        }
    }
}

// ================================================================================================
// The REFERENCE's own stream on the device (round 5): frames of libc's drand48 sequence, bit for bit.
//   * drand48 is the 48-bit LCG X <- 0x5DEECE66D X + 0xB mod 2^48 (exact integer arithmetic); My_drand48 returns (float)(X / 2^48)
//     (tools.c:71-73).  Frame f owns draws [f dpf, (f + 1) dpf), dpf = K logq + 2 N logq: the host hands every frame its state at the
//     frame's first draw (one multiply-add per frame), a thread positions itself inside the frame in O(log) steps through the powers
//     A^(2^i) / C_(2^i) (LcgPowers), and draws sequentially from there.
//   * information bits: floor((float)u * 1.9999) (tools.c:131) -- float to double, one IEEE double product, floor: exact on the device.
//   * noise: (float)(BPSK + sigma * sqrt(-2.0 * log(u)) * cos(2.0 * PI * v)) in double (channel.c:58).  sqrt, the products and the sum are
//     IEEE operations; log and cos are the host's libm there and the device's here, both within a few ulp of the true value -- but not
//     the SAME few ulp.  The double is only an intermediate: what the reference keeps is its rounding to float, and that is decided
//     unless the double lies within the libraries' disagreement of a rounding boundary.  The kernel rounds r - E and r + E with E = 2^-44
//     (1 + |noise|) -- some 500 ulp of the double, two orders of magnitude beyond what the two libraries can differ by (checked by
//     tests/test_gpu_source.py on millions of samples) -- and where the two disagree (1.5e-5 of the samples, mostly those near 0 where floats are dense:
//     about one per 64800-bit frame) it records the sample; the host recomputes exactly those with its own libm (ems_core.cpp: ems_source_draw_reference).
// The encoder between the two is k_src_rhs / k_src_solve above, bit-exact as it is.
// ================================================================================================
struct LcgPowers { uint64_t a[24], c[24]; };            // state after 2^i draws: a[i] x + c[i] (mod 2^48); offsets inside a frame stay below 2^24
constexpr uint64_t LCG_A = 0x5DEECE66DULL, LCG_C = 0xBULL, LCG_MASK = (1ULL << 48) - 1;

__device__ __forceinline__ uint64_t lcg_jump(uint64_t x, uint32_t n, const LcgPowers &P)
{
#pragma unroll 1
    for (int i = 0; n; ++i, n >>= 1)
        if (n & 1u) x = (P.a[i] * x + P.c[i]) & LCG_MASK;
    return x;
}
__device__ __forceinline__ float lcg_next(uint64_t &x)     // (float)drand48()
{
    x = (LCG_A * x + LCG_C) & LCG_MASK;
    return (float)((double)x * 0x1p-48);                    // exact scaling of a 48-bit integer, then one rounding to float
}

// information symbols: thread per (frame, symbol), frames fastest (sym is frame-minor)
__global__ void k_ref_info(const uint64_t *__restrict__ base, LcgPowers P, int B, int Bs, int M, int K, int logq,
                           const uint8_t *__restrict__ bin2idx, uint8_t *__restrict__ sym)
{
    const size_t total = (size_t)K * B;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(t % B), k = (int)(t / B);
        uint64_t x = lcg_jump(base[f], (uint32_t)k * (uint32_t)logq, P);
        unsigned img = 0;
        for (int l = 0; l < logq; ++l) img |= (unsigned)(int)floor((double)lcg_next(x) * 1.9999) << l;      // tools.c:131, bit l of the binary image
        sym[(size_t)(M + k) * Bs + f] = bin2idx[img];
    }
}

__global__ void k_ref_channel(const uint64_t *__restrict__ base, LcgPowers P, int B, int Bs, int N, int K, int logq, float sigma,
                              const int *__restrict__ iperm, const uint8_t *__restrict__ idx2bin, const uint8_t *__restrict__ sym,
                              int *__restrict__ cw, float *__restrict__ noisy, unsigned *__restrict__ flag_count, uint2 *__restrict__ flags,
                              unsigned flag_cap)
{
    const double TWO_PI = 2.0 * 3.1415926536;                   /* channel.c:18 */
    const size_t total = (size_t)B * N;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(t / N), pos = (int)(t % N);
        const int s = sym[(size_t)iperm[pos] * Bs + f];
        cw[(size_t)f * N + pos] = s;
        const unsigned img = idx2bin[s];
        uint64_t x = lcg_jump(base[f], (uint32_t)K * (uint32_t)logq + 2u * (uint32_t)pos * (uint32_t)logq, P);
        float *out = noisy + ((size_t)f * N + pos) * logq;
        for (int l = 0; l < logq; ++l) {
            const float u = lcg_next(x);
            const float v = lcg_next(x);
            const unsigned bit = (img >> l) & 1u;
            const double p = (double)sigma * sqrt(-2.0 * log((double)u)) * cos(TWO_PI * (double)v);
            const double r = (double)(1 - 2 * (int)bit) + p;
            const double E = (1.0 + fabs(p)) * 0x1p-44;
            const float lo = (float)(r - E), hi = (float)(r + E);
            out[l] = (float)r;
            if (!(lo == hi)) {                                   // undecided (or not a number): the host's libm has the last word
                const unsigned slot = atomicAdd(flag_count, 1u);
                if (slot < flag_cap) flags[slot] = uint2{(unsigned)f, ((unsigned)pos << 8) | ((unsigned)l << 1) | bit};
            }
        }
    }
}

// the host's answers for the undecided samples (same order as the flag list, which stays on the device)
__global__ void k_ref_patch(const uint2 *__restrict__ flags, const float *__restrict__ vals, unsigned n, int N, int logq, float *__restrict__ noisy)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 e = flags[i];
    noisy[((size_t)e.x * N + (e.y >> 8)) * logq + ((e.y >> 1) & 0x7Fu)] = vals[i];
}

} // namespace

hipError_t ems_launch_source_patch(const void *flags_dev, const float *vals_dev, unsigned n, int N, int logq, float *noisy_dev, hipStream_t st)
{
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_ref_patch, dim3((n + 255) / 256), dim3(256), 0, st, (const uint2 *)flags_dev, vals_dev, n, N, logq, noisy_dev);
    return hipGetLastError();
}

hipError_t ems_launch_source_reference(const EmsSourcePlan &S, const uint64_t *base_dev, const uint64_t pow_a[24], const uint64_t pow_c[24], int B, float sigma,
                                       int *cw_dev, float *noisy_dev, unsigned *flag_count_dev, void *flags_dev, unsigned flag_cap, hipStream_t st)
{
    LcgPowers P;
    for (int i = 0; i < 24; ++i) { P.a[i] = pow_a[i]; P.c[i] = pow_c[i]; }
    const int qm1 = S.q - 1;
    {
        const size_t total = (size_t)S.K * B;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(k_ref_info, dim3(blocks ? blocks : 1), dim3(256), 0, st, base_dev, P, B, S.Bs, S.M, S.K, S.logq, S.bin2idx, S.sym);
    }
    hipLaunchKernelGGL(k_src_rhs, dim3(S.M, (unsigned)((B + 255) / 256 < 16 ? (B + 255) / 256 : 16)), dim3(256), 0, st, B, S.Bs, qm1, S.rptr,
                       S.rcol, S.rcoef, S.idx2bin, S.sym, S.rhs);
    hipLaunchKernelGGL(k_src_solve, dim3((B + 63) / 64), dim3(64), 0, st, B, S.Bs, S.M, qm1, S.tptr, S.tcol, S.tcoef, S.diag, S.idx2bin,
                       S.bin2idx, S.rhs, S.sym);
    {
        const size_t total = (size_t)B * S.N;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
        hipLaunchKernelGGL(k_ref_channel, dim3(blocks), dim3(256), 0, st, base_dev, P, B, S.Bs, S.N, S.K, S.logq, sigma, S.iperm, S.idx2bin,
                           S.sym, cw_dev, noisy_dev, flag_count_dev, (uint2 *)flags_dev, flag_cap);
    }
    return hipGetLastError();
}

hipError_t ems_launch_source(const EmsSourcePlan &S, uint64_t seed, uint64_t first_frame, int B, float sigma, int *cw_dev,
                             float *noisy_dev, hipStream_t st)
{
    const int qm1 = S.q - 1;
    {
        const size_t total = (size_t)((S.K + 3) >> 2) * B;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        hipLaunchKernelGGL(k_src_info, dim3(blocks ? blocks : 1), dim3(256), 0, st, seed, first_frame, B, S.Bs, S.M, S.K, S.logq, S.bin2idx,
                           S.sym);
    }
    hipLaunchKernelGGL(k_src_rhs, dim3(S.M, (unsigned)((B + 255) / 256 < 16 ? (B + 255) / 256 : 16)), dim3(256), 0, st, B, S.Bs, qm1, S.rptr,
                       S.rcol, S.rcoef, S.idx2bin, S.sym, S.rhs);
    hipLaunchKernelGGL(k_src_solve, dim3((B + 63) / 64), dim3(64), 0, st, B, S.Bs, S.M, qm1, S.tptr, S.tcol, S.tcoef, S.diag, S.idx2bin,
                       S.bin2idx, S.rhs, S.sym);
    {
        const size_t total = (size_t)B * S.N;
        const unsigned blocks = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
        hipLaunchKernelGGL(k_src_channel, dim3(blocks), dim3(256), 0, st, seed, first_frame, B, S.Bs, S.N, S.logq, sigma, S.iperm, S.idx2bin,
                           S.sym, cw_dev, noisy_dev);
    }
    return hipGetLastError();
}
