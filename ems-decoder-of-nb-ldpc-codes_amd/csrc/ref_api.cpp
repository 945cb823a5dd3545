// ref_api.cpp -- the reference simulator's C boundary (include/nbldpc_ref_api.h) on top of the batched
// GPU library.  Host-side pieces (loader, tables, allocators, RNG, encoder) are plain C++ because the
// reference keeps them on the harness side; every hot-path symbol forwards to the HIP kernels through
// the C ABI of ems_abi.h and never computes on the CPU.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ems_abi.h"
#include "../../include/nbldpc_ref_api.h"
#include "ems_constellations.h"

namespace {

// One GPU decoder context per code_t the caller has shown us (created lazily, destroyed by FreeCode).
struct Ctx {
    ems_handle *h = nullptr;
    ems_params prm{};
    int max_batch = 1;
};
std::map<const code_t *, Ctx> g_ctx;
std::mutex g_mu;

int env_device()
{
    const char *e = getenv("EMS_DEVICE");
    return e ? atoi(e) : 0;
}

void complain(const char *who, int rc)
{
    fprintf(stderr, "\n[libems_hip] %s failed (%d): %s\n", who, rc, ems_last_error());
}

// row-pointer-over-one-block 2-D array, the allocation shape of init.c:310-415
template <typename T> T **alloc2d(size_t rows, size_t cols)
{
    T **p = (T **)calloc(rows ? rows : 1, sizeof(T *));
    T *blk = (T *)calloc((rows * cols) ? rows * cols : 1, sizeof(T));
    for (size_t r = 0; r < rows; ++r) p[r] = blk + r * cols;
    if (!rows) p[0] = blk;
    return p;
}
template <typename T> void free2d(T **p)
{
    if (!p) return;
    free(p[0]);
    free(p);
}

void flatten(const code_t *code, std::vector<int> &col, std::vector<int> &coef, ems_code_desc &d)
{
    col.clear();
    coef.clear();
    for (int m = 0; m < code->M; ++m)
        for (int k = 0; k < code->rowDegree[m]; ++k) {
            col.push_back(code->mat[m][k]);
            coef.push_back(code->matValue[m][k]);
        }
    d.N = code->N; d.M = code->M; d.GF = code->GF; d.logGF = code->logGF;
    d.nbBranch = (int)col.size();
    d.rowDegree = code->rowDegree; d.col = col.data(); d.coef = coef.data();
}

// Fetch (or build) the GPU context of `code` for message length nm.
int get_ctx(const code_t *code, int nm, int NbIterMax, int NbOper, float offset, Ctx **out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    Ctx &c = g_ctx[code];
    ems_params p{nm, NbOper, offset, NbIterMax, 1};
    if (c.h && c.prm.nm != nm) {
        ems_destroy(c.h);
        c.h = nullptr;
    }
    if (!c.h) {
        std::vector<int> col, coef;
        ems_code_desc d;
        flatten(code, col, coef, d);
        int rc = ems_create(&c.h, &d, &p, c.max_batch, env_device());
        if (rc) { g_ctx.erase(code); return rc; }
    } else {
        int rc = ems_set_params(c.h, &p);
        if (rc) return rc;
    }
    c.prm = p;
    *out = &c;
    return EMS_OK;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------
// field tables (init.c:37-130, :427-526)
// ---------------------------------------------------------------------------------------------------
void Table_Add_GF(table_t *table, int GF, int logGF)   /* init.c:37: XOR of binary images */
{
    std::vector<int> img(GF), inv(GF);
    for (int g = 0; g < GF; ++g) {
        int b = 0;
        for (int l = 0; l < logGF; ++l) b |= (table->BINGF[g][l] & 1) << l;
        img[g] = b;
        inv[b] = g;
    }
    for (int a = 0; a < GF; ++a)
        for (int b = 0; b < GF; ++b) table->ADDGF[a][b] = inv[img[a] ^ img[b]];
}

void Table_Mul_GF(int **MULGF, int GF)                  /* init.c:65: exponents add modulo GF-1 */
{
    for (int a = 0; a < GF; ++a)
        for (int b = 0; b < GF; ++b) MULGF[a][b] = (a && b) ? (a + b - 2) % (GF - 1) + 1 : 0;
}

void Table_Div_GF(int **DIVGF, int GF)                  /* init.c:100: -1 on division by zero */
{
    for (int a = 0; a < GF; ++a)
        for (int b = 0; b < GF; ++b) {
            if (b == 0) DIVGF[a][b] = -1;
            else if (a == 0) DIVGF[a][b] = 0;
            else DIVGF[a][b] = ((a - b) % (GF - 1) + (GF - 1)) % (GF - 1) + 1;
        }
}

void LoadTables(table_t *table, int GF, int logGF)      /* init.c:427 */
{
    uint8_t i2b[256], b2i[256];
    if (ems_gf_tables(GF, i2b, b2i) != EMS_OK) {
        // the reference prints and exit()s here (init.c:431-435)
        printf("The binary image of GF(%d) is not available in this version of the program. Please try GF(64) or GF(256)\n", GF);
        exit(EXIT_FAILURE);
    }
    table->BINGF = alloc2d<int>(GF, logGF);
    table->ADDGF = alloc2d<int>(GF, GF);
    table->MULGF = alloc2d<int>(GF, GF);
    table->DIVGF = alloc2d<int>(GF, GF);
    for (int g = 0; g < GF; ++g)
        for (int l = 0; l < logGF; ++l) table->BINGF[g][l] = (i2b[g] >> l) & 1;
    Table_Add_GF(table, GF, logGF);
    Table_Mul_GF(table->MULGF, GF);
    Table_Div_GF(table->DIVGF, GF);
}

void FreeTable(table_t *table)                          /* init.c:515 */
{
    free2d(table->ADDGF); free2d(table->MULGF); free2d(table->DIVGF); free2d(table->BINGF);
    table->ADDGF = table->MULGF = table->DIVGF = table->BINGF = nullptr;
}

// ---------------------------------------------------------------------------------------------------
// code + decoder storage (init.c:143-415)
// ---------------------------------------------------------------------------------------------------
void LoadCode(char *FileMatrix, code_t *code)           /* init.c:143, both alist layouts */
{
    ems_code_desc d;
    int rc = ems_load_code_file(FileMatrix, 0, &d);
    if (rc) {
        complain("LoadCode", rc);
        exit(EXIT_FAILURE);
    }
    memset(code, 0, sizeof *code);
    code->N = d.N; code->M = d.M; code->K = d.N - d.M; code->GF = d.GF; code->logGF = d.logGF;
    code->rate = (float)(d.N - d.M) / d.N;
    code->nbBranch = d.nbBranch;
    code->rowDegree = (int *)calloc(d.M, sizeof(int));
    code->columnDegree = (int *)calloc(d.N, sizeof(int));
    code->mat = (int **)calloc(d.M, sizeof(int *));
    code->matValue = (int **)calloc(d.M, sizeof(int *));
    int e = 0;
    for (int m = 0; m < d.M; ++m) {
        const int dg = d.rowDegree[m];
        code->rowDegree[m] = dg;
        code->mat[m] = (int *)calloc(dg, sizeof(int));      // per-row blocks, as init.c:181-190
        code->matValue[m] = (int *)calloc(dg, sizeof(int));
        for (int k = 0; k < dg; ++k, ++e) {
            code->mat[m][k] = d.col[e];
            code->matValue[m][k] = d.coef[e];
            code->columnDegree[d.col[e]]++;
        }
    }
    ems_free_code_desc(&d);
    printf("LDPC code parameters: \n");
    printf(" \t N \t:%d \n \t K \t:%d \n \t M\t:%d \n \t CR\t:%g \n \t GF \t:%d \n \t logGF \t:%d\n", code->N, code->K,
           code->M, code->rate, code->GF, code->logGF);
    fflush(stdout);
}

void FreeCode(code_t *code)                             /* init.c:289 */
{
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(code);
        if (it != g_ctx.end()) {
            if (it->second.h) ems_destroy(it->second.h);
            g_ctx.erase(it);
        }
    }
    for (int m = 0; m < code->M; ++m) {
        free(code->mat[m]);
        free(code->matValue[m]);
    }
    if (code->matUT) { free(code->matUT[0]); free(code->matUT); }
    free(code->mat); free(code->matValue); free(code->rowDegree); free(code->columnDegree); free(code->Perm);
    memset(code, 0, sizeof *code);
}

void AllocateDecoder(code_t *code, decoder_t *decoder)  /* init.c:310; decoder->nbMax set by caller */
{
    const int dc0 = code->rowDegree[0];                 // scratch sized by the first row (init.c:320)
    decoder->nbBranch = code->nbBranch;
    decoder->N = code->N;
    decoder->CtoV = alloc2d<softdata_t>(code->nbBranch, code->GF);
    decoder->M_CtoV_LLR = alloc2d<softdata_t>(dc0, code->GF);
    decoder->M_VtoC_LLR = alloc2d<softdata_t>(dc0, decoder->nbMax);
    decoder->M_CtoV_GF = alloc2d<int>(dc0, code->GF);
    decoder->M_VtoC_GF = alloc2d<int>(dc0, decoder->nbMax);
    decoder->APP = alloc2d<softdata_t>(code->N, code->GF);
    decoder->intrinsic_LLR = alloc2d<softdata_t>(code->N, code->GF);
    decoder->intrinsic_GF = alloc2d<int>(code->N, code->GF);
}

void FreeDecoder(decoder_t *decoder)                    /* init.c:395 */
{
    free2d(decoder->CtoV); free2d(decoder->M_CtoV_LLR); free2d(decoder->M_VtoC_LLR); free2d(decoder->M_CtoV_GF);
    free2d(decoder->M_VtoC_GF); free2d(decoder->APP); free2d(decoder->intrinsic_LLR); free2d(decoder->intrinsic_GF);
    memset(decoder, 0, sizeof *decoder);
}

// ---------------------------------------------------------------------------------------------------
// RNG, source, encoder (tools.c:40-268) -- harness side, host
// ---------------------------------------------------------------------------------------------------
float My_drand48(int *initialise)                       /* tools.c:40, the __linux__ branch */
{
    (void)initialise;
    return (float)drand48();
}

int Bin2GF(int *U, int GF, int logGF, int **BINGF)      /* tools.c:95 */
{
    for (int k = 0; k < GF; ++k) {
        int l = 0;
        while (l < logGF && U[l] == BINGF[k][l]) ++l;
        if (l == logGF) return k;
    }
    return GF;
}

void RandomBinaryGenerator(int N, int M, int GF, int logGF, int **KBIN, int *KSYMB, int **BINGF, int *init_rand)
{                                                       /* tools.c:124 */
    for (int k = 0; k < N - M; ++k) {
        for (int l = 0; l < logGF; ++l) KBIN[k][l] = (int)floor(My_drand48(init_rand) * 1.9999);
        KSYMB[k] = Bin2GF(KBIN[k], GF, logGF, BINGF);
    }
}

void GaussianElimination(code_t *code, table_t *table)  /* tools.c:151 */
{
    const int N = code->N, M = code->M;
    code->matUT = alloc2d<int>(M, N);
    code->Perm = (int *)calloc(N, sizeof(int));
    for (int n = 0; n < N; ++n) code->Perm[n] = n;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < code->rowDegree[m]; ++k) code->matUT[m][code->mat[m][k]] = code->matValue[m][k];
    for (int m = 0; m < M; ++m) {
        int *pr = code->matUT[m];
        int piv = m;
        while (piv < N && pr[piv] == 0) ++piv;
        if (piv == N) {
            printf("The matrix is not full rank (%d,%d)\n", m, piv);
            exit(EXIT_FAILURE);
        }
        std::swap(code->Perm[piv], code->Perm[m]);
        for (int r = 0; r < M; ++r) std::swap(code->matUT[r][m], code->matUT[r][piv]);
        const int d = pr[m];
        for (int r = m + 1; r < M; ++r) {
            int *row = code->matUT[r];
            const int lead = row[m];
            if (!lead) continue;
            // row <- (row / lead) * d + pivot_row over GF(q)
            for (int n = m; n < N; ++n) {
                int x = row[n];
                if (x) x = table->MULGF[table->DIVGF[x][lead]][d];
                row[n] = table->ADDGF[x][pr[n]];
            }
        }
    }
}

int Encoding(code_t *code, table_t *table, int *CodeWord, int **NBIN, int *KSYMB)  /* tools.c:232 */
{
    const int N = code->N, M = code->M;
    std::vector<int> sym(N);
    for (int k = 0; k < N - M; ++k) sym[M + k] = KSYMB[k];
    for (int m = M - 1; m >= 0; --m) {
        const int *row = code->matUT[m];
        int acc = 0;
        for (int n = m + 1; n < N; ++n)
            if (row[n]) acc = table->ADDGF[acc][table->MULGF[row[n]][sym[n]]];
        sym[m] = table->DIVGF[acc][row[m]];
    }
    for (int n = 0; n < N; ++n) CodeWord[code->Perm[n]] = sym[n];
    for (int n = 0; n < N; ++n)
        for (int l = 0; l < code->logGF; ++l) NBIN[n][l] = table->BINGF[CodeWord[n]][l];
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// channel (channel.c:38-98): noise on the host (same drand48 stream), LLR + sort on the GPU
// ---------------------------------------------------------------------------------------------------
float ems_awgn_bpsk_noise(code_t *code, int **NBIN, float EbN, int *init_rand, float *noisy)
{
    const double PI = 3.1415926536;                                            /* channel.c:18 */
    const float sigma = (float)sqrt(1.0 / (2.0 * (double)code->rate * pow(10.0, (double)EbN / 10.0))); /* channel.c:51 */
    for (int n = 0; n < code->N; ++n)
        for (int l = 0; l < code->logGF; ++l) {
            const float u = My_drand48(init_rand);
            const float v = My_drand48(init_rand);
            noisy[(size_t)n * code->logGF + l] = (float)((double)(1 - 2 * NBIN[n][l]) + (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * PI * (double)v));
        }
    return sigma;
}

// ------------------------------------------------------------------------------------------------
// Frame generation for the batched harness: frames [first_frame, first_frame + count) of the reference's
// drand48 stream (X0 = 0, never reseeded), i.e. exactly what RandomBinaryGenerator + Encoding +
// ModelChannel_AWGN_BPSK's noise draw would have produced frame after frame (NB_LDPC.c:255-261), but
// generated by `threads` host threads at once: every frame owns draws [f*dpf, (f+1)*dpf) of the stream
// (dpf = K*logGF + 2*N*logGF) and the LCG X <- a*X + c mod 2^48 can be positioned anywhere in O(log n).
// Each thread takes a contiguous chunk of frames and runs the back-substitution of Encoding (tools.c:232)
// for the whole chunk in one pass over matUT (the 131 MB of cfg 4 are streamed once per chunk, not per frame).
// libc's drand48 state is neither used nor advanced.
// ------------------------------------------------------------------------------------------------
namespace {
struct Lcg48 {
    static constexpr uint64_t A = 0x5DEECE66DULL, C = 0xBULL, MASK = (1ULL << 48) - 1;
    uint64_t x;
    static uint64_t jump(uint64_t x0, uint64_t n)       // state after n draws from x0
    {
        uint64_t a = 1, c = 0, ba = A, bc = C;
        while (n) {
            if (n & 1) { a = (ba * a) & MASK; c = (ba * c + bc) & MASK; }
            bc = (ba * bc + bc) & MASK;
            ba = (ba * ba) & MASK;
            n >>= 1;
        }
        return (a * x0 + c) & MASK;
    }
    float next()                                          // (float)drand48(), tools.c:71-73
    {
        x = (A * x + C) & MASK;
        return (float)ldexp((double)x, -48);
    }
};
} // namespace

int ems_draw_frames(code_t *code, table_t *table, long first_frame, int count, float EbN, int threads, int *codewords,
                    float *noisy, float *sigma_out)
{
    if (!code || !table || !code->matUT || !code->Perm || count < 0 || first_frame < 0 || !codewords || !noisy) return -1;
    const int N = code->N, M = code->M, K = N - M, GF = code->GF, logGF = code->logGF;
    const uint64_t dpf = (uint64_t)K * logGF + 2ULL * N * logGF;
    const float sigma = (float)sqrt(1.0 / (2.0 * (double)code->rate * pow(10.0, (double)EbN / 10.0)));   /* channel.c:51 */
    if (sigma_out) *sigma_out = sigma;
    if (count == 0) return 0;
    std::vector<int> i2b(GF), b2i(GF);                     // binary image <-> symbol index (ADDGF = XOR of images, init.c:37)
    for (int g = 0; g < GF; ++g) {
        int x = 0;
        for (int l = 0; l < logGF; ++l) x |= table->BINGF[g][l] << l;
        i2b[g] = x; b2i[x] = g;
    }
    if (threads < 1) threads = 1;
    threads = std::min(threads, count);
    const int chunk = (count + threads - 1) / threads;
    auto work = [&](int c0, int c1) {
        const int C = c1 - c0;
        const double PI = 3.1415926536;                    /* channel.c:18 */
        std::vector<uint8_t> sym((size_t)N * C);           // [n][frame in chunk]
        std::vector<Lcg48> rng(C);
        std::vector<int> bits(logGF);
        for (int fi = 0; fi < C; ++fi) {                   // RandomBinaryGenerator (tools.c:124)
            rng[fi].x = Lcg48::jump(0, (uint64_t)(first_frame + c0 + fi) * dpf);
            for (int k = 0; k < K; ++k) {
                for (int l = 0; l < logGF; ++l) bits[l] = (int)floor(rng[fi].next() * 1.9999);
                sym[(size_t)(M + k) * C + fi] = (uint8_t)Bin2GF(bits.data(), GF, logGF, table->BINGF);
            }
        }
        std::vector<int> acc(C);
        for (int m = M - 1; m >= 0; --m) {                 // Encoding (tools.c:232), all frames of the chunk per row
            const int *row = code->matUT[m];
            std::fill(acc.begin(), acc.end(), 0);
            for (int n = m + 1; n < N; ++n) {
                const int r = row[n];
                if (!r) continue;
                const int *mul = table->MULGF[r];
                const uint8_t *sn = &sym[(size_t)n * C];
                for (int fi = 0; fi < C; ++fi) acc[fi] ^= i2b[mul[sn[fi]]];
            }
            const int piv = row[m];
            for (int fi = 0; fi < C; ++fi) sym[(size_t)m * C + fi] = (uint8_t)table->DIVGF[b2i[acc[fi]]][piv];
        }
        for (int fi = 0; fi < C; ++fi) {
            int *cw = codewords + (size_t)(c0 + fi) * N;
            float *ny = noisy + (size_t)(c0 + fi) * N * logGF;
            for (int n = 0; n < N; ++n) cw[code->Perm[n]] = sym[(size_t)n * C + fi];
            for (int n = 0; n < N; ++n)                    // noise draw of ModelChannel_AWGN_BPSK (channel.c:52-62)
                for (int l = 0; l < logGF; ++l) {
                    const float u = rng[fi].next();
                    const float v = rng[fi].next();
                    const int bit = table->BINGF[cw[n]][l];
                    ny[(size_t)n * logGF + l] = (float)((double)(1 - 2 * bit) + (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * PI * (double)v));
                }
        }
    };
    std::vector<std::thread> pool;
    for (int c0 = chunk; c0 < count; c0 += chunk) pool.emplace_back(work, c0, std::min(count, c0 + chunk));
    work(0, std::min(count, chunk));
    for (auto &t : pool) t.join();
    return 0;
}

void ModelChannel_AWGN_BPSK(code_t *code, decoder_t *decoder, table_t *table, int **NBIN, float EbN, int *init_rand)
{
    (void)table;
    std::vector<float> noisy((size_t)code->N * code->logGF);
    const float sigma = ems_awgn_bpsk_noise(code, NBIN, EbN, init_rand, noisy.data());
    Ctx *c = nullptr;
    int rc = get_ctx(code, decoder->nbMax, 2, 2 * decoder->nbMax, 0.f, &c);
    if (!rc) rc = ems_llr_init(c->h, noisy.data(), sigma, 1);
    if (!rc) rc = ems_get_sorted_llr(c->h, 0, decoder->intrinsic_LLR[0], decoder->intrinsic_GF[0]);
    if (rc) {
        complain("ModelChannel_AWGN_BPSK", rc);
        exit(EXIT_FAILURE); // void signature: no way to report; never continue on stale LLRs
    }
}

void ModelChannelLLR(code_t *code, decoder_t *decoder, const float *obs, const float *att, const float *table, int D, int mode,
                     float sigma)
{
    Ctx *c = nullptr;
    int rc = get_ctx(code, decoder->nbMax, 2, 2 * decoder->nbMax, 0.f, &c);
    if (!rc) rc = ems_llr_init_constellation(c->h, obs, att, table, D, mode, sigma, 1);
    if (!rc) rc = ems_get_sorted_llr(c->h, 0, decoder->intrinsic_LLR[0], decoder->intrinsic_GF[0]);
    if (rc) {
        complain("ModelChannelLLR", rc);
        exit(EXIT_FAILURE); // void signature, as ModelChannel_AWGN_BPSK: never continue on stale LLRs
    }
}

// ---------------------------------------------------------------------------------------------------
// The reference's other channel front ends (channel.h:20-26) under their own names, so that a harness which
// calls them (NB_LDPC.c:262-264, commented out in the reference's main()) links against this library alone.  As with
// ModelChannel_AWGN_BPSK the split is: constellation construction and the draws on the host, in the reference's drand48 order and with
// its float / double promotions (every observation is what the reference's NoisyBin holds), the q-wide LLR rows and their ascending
// sort on the GPU (ModelChannelLLR above -> k_llr_const).  The switches are those the reference ships (channel.c:20-26): rotated QAM,
// independent Rayleigh fading per axis (rayleigh_fading_SSD), erasures.  Raw constellation data: ems_constellations.h.
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr double CH_PI = 3.1415926536;                                   /* channel.c:18 */

// channel.c:206-221 / :565-577 / :791-810: scale so that the average energy per two dimensions is one (float running sum in the
// reference's association, the square root in double, the product in float)
void normalise_constellation(std::vector<float> &t, int q, int D)
{
    float acc = 0.0f;
    for (int i = 0; i < q; ++i) {
        const float *p = &t[(size_t)i * D];
        acc = p[0] * p[0] + p[1] * p[1] + acc;
        if (D == 4) acc = p[2] * p[2] + p[3] * p[3] + acc;
    }
    const float scale = (float)sqrt((double)((float)((D == 4 ? 2 : 1) * q) / acc));
    for (float &x : t) x = scale * x;
}

// channel.c:346-358 / :404-416 / :524-538: the DVB-T2 QAM mapping turned by alpha = 31.7 degrees (alpha is a float, cos / sin in double)
void rotated_qam(const float *qam, int q, std::vector<float> &t)
{
    const float alpha = (float)(31.7 * CH_PI / 180);
    t.assign((size_t)q * 2, 0.0f);
    for (int i = 0; i < q; ++i) {
        const double x = (double)qam[2 * i], y = (double)qam[2 * i + 1];
        t[2 * i] = (float)(x * cos((double)alpha) - y * sin((double)alpha));
        t[2 * i + 1] = (float)(y * cos((double)alpha) + x * sin((double)alpha));
    }
}

enum ChannelDraw { DRAW_AWGN, DRAW_FADING_ERASURE, DRAW_FADING_RAW_THEN_ERASURE };

// One frame's observations and attenuations, [N][D] flat.  Per node the binary image of its symbol (`som`, channel.c:249-252), per axis
// the uniforms u, v of the Box-Muller sample (channel.c:255-261 / :636-640 / :831-833); with fading a third uniform for the attenuation
// sqrt(-log w) and a fourth for the erasure (probability 0.1: attenuation 0, else rescaled by 1 / sqrt(0.9); channel.c:648-669).
// ModelChannel forms the observation with the erased / rescaled attenuation (:674), the 4-D front end with the raw one and only
// STORES the erased one for the LLR (:836-860) -- kept as it is.
void draw_observations(ChannelDraw kind, const std::vector<float> &table, int D, code_t *code, int **NBIN, float sigma, int *init_rand,
                       std::vector<float> &obs, std::vector<float> &att)
{
    const float erasure_proba = (float)0.1;                              /* channel.c:593 / :771 */
    const int N = code->N;
    obs.assign((size_t)N * D, 0.0f);
    att.assign((size_t)N * D, 1.0f);
    for (int n = 0; n < N; ++n) {
        int image = 0;
        for (int l = 0; l < code->logGF; ++l) image += NBIN[n][l] << l;
        for (int d = 0; d < D; ++d) {
            const float u = My_drand48(init_rand);
            const float v = My_drand48(init_rand);
            const double noise = (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * CH_PI * (double)v);
            const float point = table[(size_t)image * D + d];
            float a = 1.0f;
            if (kind == DRAW_AWGN) {
                obs[(size_t)n * D + d] = (float)((double)point + noise);
            } else {
                a = (float)sqrt(-log((double)My_drand48(init_rand)));
                const float raw = a;
                a = My_drand48(init_rand) < erasure_proba ? 0.0f : (float)((double)a / sqrt((double)(1 - erasure_proba)));
                obs[(size_t)n * D + d] = (float)((double)(point * (kind == DRAW_FADING_RAW_THEN_ERASURE ? raw : a)) + noise);
            }
            att[(size_t)n * D + d] = a;
        }
    }
}

float snr_sigma(float EbN) { return (float)sqrt(1.0 / (2.0 * pow(10, (double)EbN / 10.0))); }   /* channel.c:233 / :596 / :812: no code rate here */
} // namespace

void ModelChannel_AWGN_64(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand)      /* channel.c:112-310 */
{
    if (code->GF != 64) { fprintf(stderr, "ModelChannel_AWGN_64: the 64-APSK front end needs GF(64), not GF(%d)\n", code->GF); exit(EXIT_FAILURE); }
    std::vector<float> table(64 * 2), obs, att;
    for (int i = 0; i < 64; ++i) {                                       /* :132-199: double radius * cos / sin, stored as float */
        table[2 * i] = (float)(kApsk64Radius[i] * cos(CH_PI * kApsk64Num[i] / kApsk64Den[i]));
        table[2 * i + 1] = (float)(kApsk64Radius[i] * sin(CH_PI * kApsk64Num[i] / kApsk64Den[i]));
    }
    normalise_constellation(table, 64, 2);
    const float sigma = snr_sigma(EbN);
    draw_observations(DRAW_AWGN, table, 2, code, NBIN, sigma, init_rand, obs, att);
    ModelChannelLLR(code, decoder, obs.data(), nullptr, table.data(), 2, EMS_LLR_SUM, sigma);             /* :270-304 */
}

void ModelChannel(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand)             /* channel.c:328-736 */
{
    const int q = code->GF;
    if (q != 16 && q != 64 && q != 256) { fprintf(stderr, "ModelChannel: GF(%d) has no constellation (16, 64, 256)\n", q); exit(EXIT_FAILURE); }
    std::vector<float> table, obs, att;
    rotated_qam(q == 256 ? kQam256 : (q == 64 ? kQam64 : kQam16), q, table);
    normalise_constellation(table, q, 2);
    const float sigma = snr_sigma(EbN);
    draw_observations(DRAW_FADING_ERASURE, table, 2, code, NBIN, sigma, init_rand, obs, att);
    ModelChannelLLR(code, decoder, obs.data(), att.data(), table.data(), 2, EMS_LLR_SSD, sigma);          /* :676-730 */
}

void ModelChannel_AWGN_256QAM_4D(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand)   /* channel.c:749-929 */
{
    if (code->GF != 256) { fprintf(stderr, "ModelChannel_AWGN_256QAM_4D: needs GF(256), not GF(%d)\n", code->GF); exit(EXIT_FAILURE); }
    std::vector<float> table(kQam4d16R, kQam4d16R + 256 * 4), obs, att;                                   /* :784-787 */
    normalise_constellation(table, 256, 4);
    const float sigma = snr_sigma(EbN);
    draw_observations(DRAW_FADING_RAW_THEN_ERASURE, table, 4, code, NBIN, sigma, init_rand, obs, att);
    ModelChannelLLR(code, decoder, obs.data(), att.data(), table.data(), 4, EMS_LLR_SSD, sigma);          /* :861-915 */
}

void build_QAM_table(int logGF)                                          /* channel.c:931-942: prints the half-integer grid of a square QAM */
{
    const int half = 1 << ((logGF >> 1) - 1);
    for (int i = -half; i < half; ++i)
        for (int j = -half; j < half; ++j) printf("%f \t %f \n", i + 0.5, j + 0.5);
}

// ---------------------------------------------------------------------------------------------------
// check node, decision, syndrome: device-backed single-node calls
// ---------------------------------------------------------------------------------------------------
int maximum(float *Y, int nl)                           /* bubble_decoder.c:18 */
{
    float best = 0.0f;
    int pos = 0;
    for (int i = 0; i < nl; ++i)
        if (Y[i] > best) { best = Y[i]; pos = i; }
    return pos;
}

int minimum(float *Y, int nl)                           /* bubble_decoder.c:38 */
{
    float best = 1e5f;
    int pos = 0;
    for (int i = 0; i < nl; ++i)
        if (Y[i] < best) { best = Y[i]; pos = i; }
    return pos;
}

int ElementaryStep(float *Input1, float *Input2, int *IndiceInput1, int *IndiceInput2, float *Output, int *IndiceOut,
                   int **ADDGF, int GF, int nbMax, int nbOper)
{
    (void)ADDGF; // GF addition is an XOR of binary images on the device
    int rc = ems_elementary_step(GF, nbMax, nbOper, Input1, Input2, IndiceInput1, IndiceInput2, Output, IndiceOut, env_device());
    if (rc) complain("ElementaryStep", rc);
    return rc; // the reference always returns 0 (bubble_decoder.c:592)
}

void CheckPassLogEMS(int node, decoder_t *decoder, code_t *code, table_t *table, int NbOper, float offset)
{
    const int dc = code->rowDegree[node], nm = decoder->nbMax, GF = code->GF;
    int rc = ems_check_node(GF, dc, nm, NbOper, offset, code->matValue[node], decoder->M_VtoC_LLR[0], decoder->M_VtoC_GF[0],
                            decoder->M_CtoV_LLR[0], env_device());
    if (rc) {
        complain("CheckPassLogEMS", rc);
        exit(EXIT_FAILURE);
    }
    // visible side effects of the reference: inputs left rotated (:133-152), output symbols = identity (:274-278)
    for (int t = 0; t < dc; ++t) {
        for (int k = 0; k < nm; ++k)
            if (decoder->M_VtoC_GF[t][k] != -1) decoder->M_VtoC_GF[t][k] = table->MULGF[decoder->M_VtoC_GF[t][k]][code->matValue[node][t]];
        for (int g = 0; g < GF; ++g) decoder->M_CtoV_GF[t][g] = g;
    }
}

void Decision(int *decision, float **APP, int N, int GF)    /* tools.c:312 */
{
    std::vector<float> flat((size_t)N * GF);
    for (int n = 0; n < N; ++n) memcpy(&flat[(size_t)n * GF], APP[n], sizeof(float) * GF);
    int rc = ems_decision(GF, N, flat.data(), decision, env_device());
    if (rc) {
        complain("Decision", rc);
        exit(EXIT_FAILURE);
    }
}

int Syndrom(code_t *code, int *decide, table_t *tableGF)    /* tools.c:284 */
{
    (void)tableGF;
    std::vector<int> col, coef;
    ems_code_desc d;
    flatten(code, col, coef, d);
    int synd = 0;
    int rc = ems_syndrome(&d, decide, &synd, env_device());
    if (rc) {
        complain("Syndrom", rc);
        exit(EXIT_FAILURE);
    }
    return synd;
}

// ---------------------------------------------------------------------------------------------------
// Decoding(): NB_LDPC.c:266-472 hoisted; one frame through the batched GPU path
// ---------------------------------------------------------------------------------------------------
int Decoding(code_t *code, table_t *table, decoder_t *decoder, int NbIterMax, int NbOper, float offset, int *decide, int *synd)
{
    (void)table;
    Ctx *c = nullptr;
    int rc = get_ctx(code, decoder->nbMax, NbIterMax, NbOper, offset, &c);
    if (rc) { complain("Decoding", rc); return rc; }
    const int N = code->N, GF = code->GF;
    // NB_LDPC.c:281-288: APP[n][intrinsic_GF[n][k]] = intrinsic_LLR[n][k]
    std::vector<float> llr((size_t)N * GF, 1e5f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < GF; ++k) {
            const int g = decoder->intrinsic_GF[n][k];
            if (g >= 0 && g < GF) llr[(size_t)n * GF + g] = decoder->intrinsic_LLR[n][k];
        }
    int iters = 0, s = 0;
    if ((rc = ems_load_llr(c->h, llr.data(), 1)) || (rc = ems_decode(c->h, 1)) ||
        (rc = ems_get_results(c->h, 1, decide, &iters, &s)) || (rc = ems_get_app(c->h, 0, decoder->APP[0])) ||
        (rc = ems_get_ctov(c->h, 0, decoder->CtoV[0]))) {
        complain("Decoding", rc);
        return rc;
    }
    if (synd) *synd = s;
    return iters;
}

} // extern "C"
