/* oracle/ems_source_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * CPU statement of the device-side frame source (product: csrc/ems_source.hip):
 *   - srco_encode follows the reference's Encoding (tools.c:232-262): back substitution on GaussianElimination's
 *     matUT from row M-1 up, information symbols in columns M..N-1, then CodeWord[Perm[n]] = sym[n].  It is
 *     pinned against the real Encoding of oracle/_ref (tests/test_source_oracle.py).
 *   - the random stream has NO reference counterpart (the reference draws from libc's drand48; f2 asks for a
 *     counter-based generator with statistical parity): srco_philox restates Philox4x32-10 from its publication
 *     (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) and is pinned by that
 *     paper's known-answer vectors; srco_info / srco_noise restate how ems_source.hip maps words to symbols
 *     and Gaussian samples.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

void srco_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* information symbols of one frame: word i of block j (stream 0) -> symbol 4j+i, binary image = top logq bits */
void srco_info(uint64_t seed, uint64_t frame, int K, int logq, const int *bin2idx, int *ksymb)
{
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int j = 0; 4 * j < K; j++) {
        const uint32_t ctr[4] = {(uint32_t)j, 0u, (uint32_t)frame, (uint32_t)(frame >> 32)};
        uint32_t w[4];
        srco_philox(ctr, key, w);
        for (int i = 0; i < 4 && 4 * j + i < K; i++) ksymb[4 * j + i] = bin2idx[w[i] >> (32 - logq)];
    }
}

/* tools.c:232-252 (tables as q x q int arrays, ADDGF/MULGF/DIVGF of init.c) */
void srco_encode(int N, int M, int q, const int *matUT, const int *perm, const int *add, const int *mul, const int *dv,
                 const int *ksymb, int *sym_scratch, int *codeword)
{
    int *sym = sym_scratch;
    for (int k = 0; k < N - M; k++) sym[M + k] = ksymb[k];
    for (int m = M - 1; m >= 0; m--) {
        const int *row = matUT + (size_t)m * N;
        int acc = 0;
        for (int n = m + 1; n < N; n++)
            if (row[n]) acc = add[acc * q + mul[row[n] * q + sym[n]]];
        sym[m] = dv[acc * q + row[m]];
    }
    for (int n = 0; n < N; n++) codeword[perm[n]] = sym[n];
}

static float box_muller(uint32_t wu, uint32_t wv)
{
    const float u = (float)((wu >> 8) + 1u) * 5.9604644775390625e-08f;
    const float v = (float)(wv >> 8) * 5.9604644775390625e-08f;
    return sqrtf(-2.0f * logf(u)) * cosf(6.28318530717958647692f * v);
}

/* noisy[n][l] = (1 - 2 bit) + sigma * gauss, sample index i = n*logq + l -> block i/2 of stream 1, half i&1 */
void srco_noise(uint64_t seed, uint64_t frame, int N, int logq, const int *idx2bin, const int *codeword, float sigma, float *noisy)
{
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int n = 0; n < N; n++)
        for (int l = 0; l < logq; l++) {
            const uint32_t i = (uint32_t)n * (uint32_t)logq + (uint32_t)l;
            const uint32_t ctr[4] = {i >> 1, 1u, (uint32_t)frame, (uint32_t)(frame >> 32)};
            uint32_t w[4];
            srco_philox(ctr, key, w);
            const float g = (i & 1u) ? box_muller(w[2], w[3]) : box_muller(w[0], w[1]);
            const float tx = (float)(1 - 2 * ((idx2bin[codeword[n]] >> l) & 1));
            noisy[(size_t)n * logq + l] = tx + sigma * g;
        }
}
