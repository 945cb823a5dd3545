/*
 * ems_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ems_oracle.h).
 *
 * CPU restatement of the reference EMS NB-LDPC hot path, written from the reference's observable
 * behaviour (file:line cited per function).  Deliberately literal where literal matters for
 * bit-exact parity (float/double promotion order, strict '<' against the 1e5 sentinel, tie -> lowest
 * index, the three stop rules of the bubble loop), and deliberately NOT clever: it is the checker.
 */
#define _POSIX_C_SOURCE 200809L
#include "ems_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SENTINEL 1e5f

/* ------------------------------------------------------------------------------------------------
 * GF(q)
 * ---------------------------------------------------------------------------------------------- */

/* struct.h:117 (x^4+x+1), :145 (x^6+x+1), :217 (x^8+x^4+x^3+x^2+1): index i>=1 is alpha^(i-1), the
 * binary image stores the coefficient of x^l in bit l. */
int orc_gf_init(orc_gf *gf, int q)
{
    unsigned poly;
    int logq;
    switch (q) {
    case 16: poly = 0x13; logq = 4; break;
    case 64: poly = 0x43; logq = 6; break;
    case 256: poly = 0x11D; logq = 8; break;
    default: return -1; /* init.c:431-435 exits here */
    }
    memset(gf, 0, sizeof *gf);
    gf->q = q;
    gf->logq = logq;
    unsigned b = 1;
    gf->idx2bin[0] = 0;
    gf->bin2idx[0] = 0;
    for (int i = 1; i < q; i++) {
        gf->idx2bin[i] = (uint8_t)b;
        gf->bin2idx[b] = (uint8_t)i;
        b <<= 1;
        if (b & (unsigned)q) b ^= poly;
    }
    return 0;
}

/* init.c:37-55: XOR of the binary images, mapped back through Bin2GF (tools.c:95) */
int orc_gf_add(const orc_gf *gf, int a, int b)
{
    return gf->bin2idx[gf->idx2bin[a] ^ gf->idx2bin[b]];
}

/* init.c:65-89 */
int orc_gf_mul(const orc_gf *gf, int a, int b)
{
    if (a == 0 || b == 0) return 0;
    return (a + b - 2) % (gf->q - 1) + 1;
}

/* init.c:100-130: DIVGF[a][0] = -1, DIVGF[0][b] = 0, else exponent difference */
int orc_gf_div(const orc_gf *gf, int a, int b)
{
    if (b == 0) return -1;
    if (a == 0) return 0;
    int m = gf->q - 1;
    return ((a - b) % m + m) % m + 1;
}

/* ------------------------------------------------------------------------------------------------
 * alist loader (init.c:143-272), both on-disk layouts
 * ---------------------------------------------------------------------------------------------- */

static int columns_match(const int *cols, int E, int N, const int *col_deg)
{
    int *cnt = calloc((size_t)N, sizeof(int));
    int ok = 1;
    for (int e = 0; e < E && ok; e++) {
        if (cols[e] < 0 || cols[e] >= N) ok = 0;
        else cnt[cols[e]]++;
    }
    for (int n = 0; n < N && ok; n++)
        if (cnt[n] != col_deg[n]) ok = 0;
    free(cnt);
    return ok;
}

/* Full non-binary alist (matrices/KN/N64800_*.txt): "N M q", "dvmax dcmax", N column degrees, M row degrees, then per
 * COLUMN its (row+1, exponent) pairs and per ROW its (col+1, exponent) pairs.  Neither loader of the reference reads it
 * (SURVEY.md 8(f) f4); there is therefore no reference behaviour to follow, only the format: the row section is used,
 * the column section must describe the same matrix. */
static orc_code *load_full_alist(const int *T, size_t nT, int N, int M, int q)
{
    if (nT < (size_t)2 + N + M) return NULL;
    const int dvmax = T[0], dcmax = T[1];
    const int *cdeg = T + 2, *rdeg = T + 2 + N;
    long E = 0, sumc = 0;
    for (int m = 0; m < M; m++) { if (rdeg[m] <= 0 || rdeg[m] > dcmax) return NULL; E += rdeg[m]; }
    for (int n = 0; n < N; n++) { if (cdeg[n] < 0 || cdeg[n] > dvmax) return NULL; sumc += cdeg[n]; }
    if (sumc != E || nT != (size_t)2 + N + M + 4 * (size_t)E) return NULL;
    const int *csec = T + 2 + N + M, *rsec = csec + 2 * E;
    orc_code *c = calloc(1, sizeof *c);
    c->N = N; c->M = M; c->q = q;
    c->logq = (int)rint(log((double)q) / log(2.0));
    c->K = N - M;
    c->rate = (float)(N - M) / N;
    c->col_deg = calloc((size_t)N, sizeof(int));
    c->row_deg = calloc((size_t)M, sizeof(int));
    c->row_ptr = calloc((size_t)M + 1, sizeof(int));
    c->col = malloc(sizeof(int) * (size_t)E);
    c->coef = malloc(sizeof(int) * (size_t)E);
    c->E = (int)E;
    int ok = 1;
    for (int m = 0; m < M; m++) {
        c->row_deg[m] = rdeg[m];
        c->row_ptr[m + 1] = c->row_ptr[m] + rdeg[m];
        if (rdeg[m] > c->dc_max) c->dc_max = rdeg[m];
    }
    for (long e = 0; e < E && ok; e++) {
        c->col[e] = rsec[2 * e] - 1;
        c->coef[e] = rsec[2 * e + 1] + 1;
        if (c->col[e] < 0 || c->col[e] >= N || rsec[2 * e + 1] < 0 || rsec[2 * e + 1] > q - 2) ok = 0;
        else c->col_deg[c->col[e]]++;
    }
    for (int n = 0; n < N && ok; n++) ok = c->col_deg[n] == cdeg[n];
    /* cross-check with the column section: walk the columns, every (row, exponent) must be found in that row */
    long e = 0;
    for (int n = 0; n < N && ok; n++)
        for (int k = 0; k < cdeg[n] && ok; k++, e++) {
            const int r = csec[2 * e] - 1;
            int found = 0;
            if (r >= 0 && r < M)
                for (int j = c->row_ptr[r]; j < c->row_ptr[r + 1]; j++)
                    if (c->col[j] == n && c->coef[j] == csec[2 * e + 1] + 1) found = 1;
            ok = found;
        }
    if (!ok) { orc_free_code(c); return NULL; }
    return c;
}

orc_code *orc_load_code(const char *path, int fmt)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    orc_code *c = calloc(1, sizeof *c);
    if (fscanf(f, "%d %d %d", &c->N, &c->M, &c->q) != 3 || c->N <= 0 || c->M <= 0 || c->M >= c->N) {
        fclose(f);
        free(c);
        return NULL;
    }
    /* every integer after the header */
    size_t cap = 1 << 16, nT = 0;
    int *T = malloc(cap * sizeof(int)), x;
    while (fscanf(f, "%d", &x) == 1) {
        if (nT == cap) { cap *= 2; T = realloc(T, cap * sizeof(int)); }
        T[nT++] = x;
    }
    fclose(f);
    c->logq = (int)rint(log((double)c->q) / log(2.0)); /* init.c:163-166 */
    c->K = c->N - c->M;
    c->rate = (float)(c->N - c->M) / c->N;              /* init.c:170 */
    c->col_deg = calloc((size_t)c->N, sizeof(int));
    c->row_deg = calloc((size_t)c->M, sizeof(int));
    c->row_ptr = calloc((size_t)c->M + 1, sizeof(int));
    int ok = nT >= (size_t)c->N + c->M;
    for (int n = 0; n < c->N && ok; n++) c->col_deg[n] = T[n];
    for (int m = 0; m < c->M && ok; m++) c->row_deg[m] = T[c->N + m];
    for (int m = 0; m < c->M && ok; m++) {
        if (c->row_deg[m] <= 0) { ok = 0; break; }
        c->row_ptr[m + 1] = c->row_ptr[m] + c->row_deg[m];
        if (c->row_deg[m] > c->dc_max) c->dc_max = c->row_deg[m];
    }
    c->E = ok ? c->row_ptr[c->M] : 0;
    int E = c->E;
    ok = ok && nT >= (size_t)c->N + c->M + 2 * (size_t)E;
    const int *tok = T + c->N + c->M;
    int use = -1;
    if (ok && fmt != ORC_FMT_FULL) {
        c->col = malloc(sizeof(int) * (size_t)E);
        c->coef = malloc(sizeof(int) * (size_t)E);
        /* UBS block layout (init.c:195-207): E column indices (0-based) then E coefficients (symbol idx) */
        int ubs_ok = columns_match(tok, E, c->N, c->col_deg);
        for (int e = 0; e < E && ubs_ok; e++)
            if (tok[E + e] < 1 || tok[E + e] >= c->q) ubs_ok = 0;
        /* KN interleaved layout (init.c:211-227): (col+1, exponent) pairs */
        int kn_ok = 1;
        for (int e = 0; e < E; e++) {
            c->col[e] = tok[2 * e] - 1;
            if (tok[2 * e + 1] < 0 || tok[2 * e + 1] > c->q - 2) kn_ok = 0;
        }
        kn_ok = kn_ok && columns_match(c->col, E, c->N, c->col_deg);
        use = fmt;
        if (use == ORC_FMT_AUTO) use = ubs_ok ? ORC_FMT_UBS : (kn_ok ? ORC_FMT_KN : -1);
        if (use == ORC_FMT_UBS) {
            for (int e = 0; e < E; e++) { c->col[e] = tok[e]; c->coef[e] = tok[E + e]; }
        } else if (use == ORC_FMT_KN) {
            for (int e = 0; e < E; e++) { c->col[e] = tok[2 * e] - 1; c->coef[e] = tok[2 * e + 1] + 1; }
        }
    }
    if (use != ORC_FMT_UBS && use != ORC_FMT_KN) {
        const int N = c->N, M = c->M, q = c->q;
        orc_free_code(c);
        c = (fmt == ORC_FMT_AUTO || fmt == ORC_FMT_FULL) ? load_full_alist(T, nT, N, M, q) : NULL;
    }
    free(T);
    return c;
}

void orc_free_code(orc_code *c)
{
    if (!c) return;
    free(c->row_deg); free(c->col_deg); free(c->row_ptr); free(c->col); free(c->coef);
    free(c);
}

/* ------------------------------------------------------------------------------------------------
 * RNG: glibc drand48 (tools.c:71-73): X <- (0x5DEECE66D * X + 0xB) mod 2^48, value X / 2^48
 * ---------------------------------------------------------------------------------------------- */
double orc_drand48(orc_rng *r)
{
    r->x = (0x5DEECE66DULL * r->x + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)r->x / 281474976710656.0;
}

static float my_drand48(orc_rng *r) { return (float)orc_drand48(r); } /* float return type, tools.c:40 */

/* ------------------------------------------------------------------------------------------------
 * encoder: GaussianElimination (tools.c:151-229) + Encoding (tools.c:232-268)
 * ---------------------------------------------------------------------------------------------- */
struct orc_encoder {
    int N, M, q;
    int *perm;            /* tools.c:165 Perm */
    int *ut_ptr;          /* CSR of the upper-triangular matrix rows: strictly-right-of-diagonal */
    int *ut_col;
    uint8_t *ut_val;
    uint8_t *diag;        /* matUT[m][m] */
    const orc_gf *gf;
};

orc_encoder *orc_encoder_create(const orc_code *c, const orc_gf *gf)
{
    const int N = c->N, M = c->M, q = c->q;
    uint8_t *ut = calloc((size_t)M * N, 1);
    if (!ut) return NULL;
    uint8_t *mul = malloc((size_t)q * q), *dv = malloc((size_t)q * q);
    for (int a = 0; a < q; a++)
        for (int b = 0; b < q; b++) {
            mul[a * q + b] = (uint8_t)orc_gf_mul(gf, a, b);
            dv[a * q + b] = (uint8_t)(b ? orc_gf_div(gf, a, b) : 0);
        }
    orc_encoder *e = calloc(1, sizeof *e);
    e->N = N; e->M = M; e->q = q; e->gf = gf;
    e->perm = malloc(sizeof(int) * (size_t)N);
    for (int n = 0; n < N; n++) e->perm[n] = n;
    for (int m = 0; m < M; m++)
        for (int k = c->row_ptr[m]; k < c->row_ptr[m + 1]; k++) ut[(size_t)m * N + c->col[k]] = (uint8_t)c->coef[k];

    for (int m = 0; m < M; m++) {
        uint8_t *rm = ut + (size_t)m * N;
        int ind = m;
        while (ind < N && rm[ind] == 0) ind++;       /* tools.c:177-180 first non-zero at/after m */
        if (ind == N) {                              /* tools.c:181-185 rank deficient */
            free(ut); free(mul); free(dv); orc_encoder_free(e);
            return NULL;
        }
        int t = e->perm[ind]; e->perm[ind] = e->perm[m]; e->perm[m] = t;
        for (int m1 = 0; m1 < M; m1++) {             /* tools.c:189-194 column swap */
            uint8_t *r1 = ut + (size_t)m1 * N;
            uint8_t b = r1[m]; r1[m] = r1[ind]; r1[ind] = b;
        }
        const uint8_t piv = rm[m];
        for (int m1 = m + 1; m1 < M; m1++) {         /* tools.c:196-216 */
            uint8_t *r1 = ut + (size_t)m1 * N;
            const uint8_t buf = r1[m];
            if (!buf) continue;
            for (int n = m; n < N; n++) {
                uint8_t x = r1[n];
                if (x) x = dv[x * q + buf];
                if (x) x = mul[x * q + piv];
                r1[n] = gf->bin2idx[gf->idx2bin[x] ^ gf->idx2bin[rm[n]]];
            }
        }
    }
    e->diag = malloc((size_t)M);
    e->ut_ptr = malloc(sizeof(int) * ((size_t)M + 1));
    size_t nnz = 0;
    for (int m = 0; m < M; m++)
        for (int n = m + 1; n < N; n++) nnz += ut[(size_t)m * N + n] != 0;
    e->ut_col = malloc(sizeof(int) * (nnz ? nnz : 1));
    e->ut_val = malloc(nnz ? nnz : 1);
    size_t p = 0;
    for (int m = 0; m < M; m++) {
        e->ut_ptr[m] = (int)p;
        e->diag[m] = ut[(size_t)m * N + m];
        for (int n = m + 1; n < N; n++)
            if (ut[(size_t)m * N + n]) { e->ut_col[p] = n; e->ut_val[p] = ut[(size_t)m * N + n]; p++; }
    }
    e->ut_ptr[M] = (int)p;
    free(ut); free(mul); free(dv);
    return e;
}

void orc_encoder_free(orc_encoder *e)
{
    if (!e) return;
    free(e->perm); free(e->ut_ptr); free(e->ut_col); free(e->ut_val); free(e->diag);
    free(e);
}

/* tools.c:124-136: K*logq draws, bit = floor((float)drand48 * 1.9999), symbol = Bin2GF(bits) */
void orc_random_info(orc_rng *r, const orc_code *c, const orc_gf *gf, int *ksymb)
{
    for (int k = 0; k < c->K; k++) {
        unsigned bits = 0;
        for (int l = 0; l < c->logq; l++) {
            int b = (int)floor(my_drand48(r) * 1.9999);
            bits |= (unsigned)b << l;
        }
        ksymb[k] = gf->bin2idx[bits];
    }
}

/* tools.c:232-268: back-substitution over the upper-triangular form, then de-interleave by Perm */
void orc_encode(const orc_encoder *e, const int *ksymb, int *codeword)
{
    const int N = e->N, M = e->M;
    const orc_gf *gf = e->gf;
    int *ns = malloc(sizeof(int) * (size_t)N);
    for (int k = 0; k < N - M; k++) ns[M + k] = ksymb[k];
    for (int m = M - 1; m >= 0; m--) {
        unsigned acc = 0; /* binary image of the running GF sum */
        for (int p = e->ut_ptr[m]; p < e->ut_ptr[m + 1]; p++)
            acc ^= gf->idx2bin[orc_gf_mul(gf, e->ut_val[p], ns[e->ut_col[p]])];
        ns[m] = orc_gf_div(gf, gf->bin2idx[acc], e->diag[m]);
    }
    for (int n = 0; n < N; n++) codeword[e->perm[n]] = ns[n];
    free(ns);
}

/* ------------------------------------------------------------------------------------------------
 * channel (channel.c:38-98)
 * ---------------------------------------------------------------------------------------------- */

/* channel.c:51: double expression stored into a float */
float orc_sigma(float rate, float EbN)
{
    return (float)sqrt(1.0 / (2.0 * (double)rate * pow(10, (double)EbN / 10.0)));
}

/* channel.c:52-62: two float-rounded uniforms per bit, Box-Muller in double, PI = 3.1415926536 */
void orc_awgn_bpsk(orc_rng *r, const orc_code *c, const orc_gf *gf, const int *codeword, float sigma,
                   float *noisy)
{
    const double PI = 3.1415926536;
    for (int n = 0; n < c->N; n++)
        for (int l = 0; l < c->logq; l++) {
            float u = my_drand48(r);
            float v = my_drand48(r);
            int bit = (gf->idx2bin[codeword[n]] >> l) & 1;
            noisy[(size_t)n * c->logq + l] =
                (float)((double)(1 - 2 * bit) + (double)sigma * sqrt(-2.0 * log((double)u)) * cos(2.0 * PI * (double)v));
        }
}

/* channel.c:66-76: (y-s) and its square and sigma*sigma are float ops; the division and the
 * accumulation are double; the running sum is rounded to float after every bit. */
void orc_llr(const orc_code *c, const orc_gf *gf, const float *noisy, float sigma, float *llr)
{
    const int q = c->q, logq = c->logq;
    for (int n = 0; n < c->N; n++)
        for (int g = 0; g < q; g++) {
            float t = 0.0f;
            for (int l = 0; l < logq; l++) {
                int bit = (gf->idx2bin[g] >> l) & 1;
                float a = noisy[(size_t)n * logq + l] - (float)(1 - 2 * bit);
                float sq = a * a;
                t = (float)((double)t + (double)sq / (2.0 * (double)(sigma * sigma)));
            }
            llr[(size_t)n * q + g] = t;
        }
}

/* channel.c:78-91: q rounds of "first strict minimum below 1e5, then mark it 1e5" */
void orc_sort_llr(const float *llr, int N, int q, float *s_llr, int *s_gf)
{
    float *tmp = malloc(sizeof(float) * (size_t)q);
    for (int n = 0; n < N; n++) {
        memcpy(tmp, llr + (size_t)n * q, sizeof(float) * (size_t)q);
        for (int k = 0; k < q; k++) {
            float best = SENTINEL;
            int bg = -1;
            for (int g = 0; g < q; g++)
                if (tmp[g] < best) { best = tmp[g]; bg = g; }
            s_llr[(size_t)n * q + k] = best;
            s_gf[(size_t)n * q + k] = bg;
            if (bg >= 0) tmp[bg] = SENTINEL;
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * L-bubble elementary check node (bubble_decoder.c:316-593) and its minimum() (:38-56)
 * ---------------------------------------------------------------------------------------------- */
static int first_min8(const float *y)
{
    float aux = SENTINEL;
    int pos = 0;
    for (int i = 0; i < 8; i++)
        if (y[i] < aux) { aux = y[i]; pos = i; }
    return pos;
}

int orc_elementary_step(const float *in1, const float *in2, const int *idx1, const int *idx2,
                        float *out, int *idxout, const orc_gf *gf, int nm, int nbOper)
{
    float lo[64];
    int li[64];
    uint8_t seen[256];
    float cv[8];
    int cr[8], cc[8];
    if (nm > 64 || nm < 5) return -1;
    memset(seen, 0, sizeof seen);
    for (int i = 0; i < nm; i++) { lo[i] = SENTINEL; li[i] = -1; }           /* :370-374 */
    /* :445-460 competitors 0..3 sit on rows 0..3 at column 0, 4..7 on row 4 at columns 0..3;
     * every cell they can ever reach lies inside the pre-computed "L" (:386-423) and equals
     * in1[r] + in2[c] as a float add. */
    for (int p = 0; p < 4; p++) { cr[p] = p; cc[p] = 0; cr[p + 4] = 4; cc[p + 4] = p; }
    for (int p = 0; p < 8; p++) cv[p] = in1[cr[p]] + in2[cc[p]];

    int s = 0;
    for (int ss = 0; ss < nbOper; ss++) {                                     /* :472 */
        int pos = first_min8(cv);
        int r = cr[pos], c = cc[pos];
        if (idx1[r] == -1 || idx2[c] == -1) break;                            /* :478-484 */
        int sym = orc_gf_add(gf, idx1[r], idx2[c]);
        if (!seen[sym]) {                                                     /* :490-496 */
            lo[s] = cv[pos];
            li[s] = sym;
            seen[sym] = 1;
            s++;
        }
        if (s == nm) break;                                                   /* :502 */
        if (r >= nm - 1 || c >= nm - 1) break;                                /* :506-544 */
        if (pos > 3) cr[pos]++; else cc[pos]++;                               /* :548-555 */
        cv[pos] = in1[cr[pos]] + in2[cc[pos]];                                /* :557 */
    }
    for (int i = 0; i < nm; i++) { out[i] = lo[i]; idxout[i] = li[i]; }       /* :564-568 */
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * forward/backward check node (bubble_decoder.c:72-305)
 * ---------------------------------------------------------------------------------------------- */
void orc_check_node(int dc, int nm, const orc_gf *gf, const int *coef, const float *vtoc_llr,
                    const int *vtoc_gf, int nbOper, float offset, float *ctov)
{
    const int q = gf->q;
    const int S = 2 * (dc - 2);
    float *inl = malloc(sizeof(float) * (size_t)dc * nm);
    int *ing = malloc(sizeof(int) * (size_t)dc * nm);
    float *mil = malloc(sizeof(float) * (size_t)(S > 0 ? S : 1) * nm);
    int *mig = malloc(sizeof(int) * (size_t)(S > 0 ? S : 1) * nm);
    float *outl = malloc(sizeof(float) * (size_t)dc * nm);
    int *outg = malloc(sizeof(int) * (size_t)dc * nm);
    float F[64], B[64], t1[64];
    int Fg[64], Bg[64], t1g[64];

    /* :133-152 rotate every input symbol by its edge coefficient */
    for (int t = 0; t < dc; t++)
        for (int k = 0; k < nm; k++) {
            inl[t * nm + k] = vtoc_llr[t * nm + k];
            int g = vtoc_gf[t * nm + k];
            ing[t * nm + k] = (g != -1) ? orc_gf_mul(gf, g, coef[t]) : -1;
        }
    for (int k = 0; k < S * nm; k++) { mil[k] = SENTINEL; mig[k] = -1; }      /* :100-120 */
    for (int k = 0; k < nm; k++) {                                            /* :157-163 */
        F[k] = inl[k]; Fg[k] = ing[k];
        B[k] = inl[(dc - 1) * nm + k]; Bg[k] = ing[(dc - 1) * nm + k];
    }
    for (int kk = 1; kk < dc - 1; kk++) {                                     /* :166-200 */
        memcpy(mil + (kk - 1) * nm, F, sizeof(float) * (size_t)nm);
        memcpy(mig + (kk - 1) * nm, Fg, sizeof(int) * (size_t)nm);
        memcpy(mil + (S - kk) * nm, B, sizeof(float) * (size_t)nm);
        memcpy(mig + (S - kk) * nm, Bg, sizeof(int) * (size_t)nm);
        orc_elementary_step(F, inl + kk * nm, Fg, ing + kk * nm, F, Fg, gf, nm, nbOper);
        orc_elementary_step(B, inl + (dc - kk - 1) * nm, Bg, ing + (dc - kk - 1) * nm, B, Bg, gf, nm, nbOper);
    }
    memcpy(outl + (dc - 1) * nm, F, sizeof(float) * (size_t)nm);              /* :204-213 */
    memcpy(outg + (dc - 1) * nm, Fg, sizeof(int) * (size_t)nm);
    memcpy(outl, B, sizeof(float) * (size_t)nm);
    memcpy(outg, Bg, sizeof(int) * (size_t)nm);
    for (int k = 0; k < dc - 2; k++) {                                        /* :217-227 merge */
        orc_elementary_step(mil + k * nm, mil + (dc - 2 + k) * nm, mig + k * nm, mig + (dc - 2 + k) * nm,
                            t1, t1g, gf, nm, nbOper);
        memcpy(outl + (k + 1) * nm, t1, sizeof(float) * (size_t)nm);
        memcpy(outg + (k + 1) * nm, t1g, sizeof(int) * (size_t)nm);
    }
    /* :231-281 inverse rotation of the valid prefix, then expand to a q-wide vector whose default is
     * (last valid LLR + offset) */
    for (int t = 0; t < dc; t++) {
        int stp = nm;
        for (int k = 0; k < nm; k++)
            if (outg[t * nm + k] == -1) { stp = k; break; }
        float dflt = outl[t * nm + stp - 1] + offset; /* stp >= 1 always: ES emits its first pick */
        for (int g = 0; g < q; g++) ctov[t * q + g] = dflt;
        for (int k = 0; k < stp; k++)
            ctov[t * q + orc_gf_div(gf, outg[t * nm + k], coef[t])] = outl[t * nm + k];
    }
    free(inl); free(ing); free(mil); free(mig); free(outl); free(outg);
}

/* ------------------------------------------------------------------------------------------------
 * hard decision + syndrome (tools.c:284-330)
 * ---------------------------------------------------------------------------------------------- */
void orc_decision(const float *app, int N, int q, int *decide)
{
    for (int n = 0; n < N; n++) {
        float mn = SENTINEL;
        int ind = 0;
        for (int g = 0; g < q; g++)
            if (app[(size_t)n * q + g] < mn) { mn = app[(size_t)n * q + g]; ind = g; }
        decide[n] = ind;
    }
}

int orc_syndrome(const orc_code *c, const orc_gf *gf, const int *decide)
{
    int synd = 0;
    for (int k = 0; k < c->M; k++) {
        for (int e = c->row_ptr[k]; e < c->row_ptr[k + 1]; e++)
            synd = orc_gf_add(gf, synd, orc_gf_mul(gf, c->coef[e], decide[c->col[e]]));
        if (synd != 0) break;
    }
    return synd;
}

/* ------------------------------------------------------------------------------------------------
 * CRC-32 (IEEE 802.3, reflected) for golden trajectories
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_crc32(const void *buf, uint64_t len)
{
    static uint32_t tab[256];
    static int init = 0;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            tab[i] = c;
        }
        init = 1;
    }
    const uint8_t *p = buf;
    uint32_t c = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < len; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------------------------------------
 * layered decode of one frame (NB_LDPC.c:266-474)
 * ---------------------------------------------------------------------------------------------- */
int orc_decode_frame(const orc_code *c, const orc_gf *gf, const float *llr, int nm, int nbOper,
                     float offset, int NbIterMax, const int *order, int early_exit, int *decide,
                     int *synd_out, float *app_out, float *ctov_out, uint8_t *sweep_decide,
                     uint32_t *sweep_crc, int *sweep_synd)
{
    const int N = c->N, M = c->M, q = c->q, E = c->E, dcm = c->dc_max;
    float *app = malloc(sizeof(float) * (size_t)N * q);
    float *ctov = calloc((size_t)E * q, sizeof(float));                      /* :273-279 */
    float *mvc = malloc(sizeof(float) * (size_t)dcm * q);
    float *mvc2 = malloc(sizeof(float) * (size_t)dcm * q);
    float *mcv = malloc(sizeof(float) * (size_t)dcm * q);
    float *vl = malloc(sizeof(float) * (size_t)dcm * nm);
    int *vg = malloc(sizeof(int) * (size_t)dcm * nm);
    memcpy(app, llr, sizeof(float) * (size_t)N * q);                         /* :281-288 */
    int synd = 0, iter;

    for (iter = 0; iter < NbIterMax - 1; iter++) {                           /* :314 */
        for (int v = 0; v < M; v++) {                                        /* :320 */
            const int node = order ? order[v] : v;
            const int numB = c->row_ptr[node], dc = c->row_deg[node];
            for (int i = 0; i < dc; i++) {                                   /* :329-337 */
                const float *a = app + (size_t)c->col[numB + i] * q;
                const float *cv = ctov + (size_t)(numB + i) * q;
                for (int g = 0; g < q; g++) mvc2[i * q + g] = mvc[i * q + g] = a[g] - cv[g];
            }
            for (int i = 0; i < dc; i++) {                                   /* :354-374 */
                for (int k = 0; k < nm; k++) {
                    float best = SENTINEL;
                    int bg = 0;
                    for (int g = 0; g < q; g++)
                        if (mvc2[i * q + g] < best) { best = mvc2[i * q + g]; bg = g; }
                    vl[i * nm + k] = best;
                    vg[i * nm + k] = bg;
                    mvc2[i * q + bg] = SENTINEL;
                }
                for (int g = 1; g < nm; g++) vl[i * nm + g] = vl[i * nm + g] - vl[i * nm];
                vl[i * nm] = 0.0f;
            }
            orc_check_node(dc, nm, gf, c->coef + numB, vl, vg, nbOper, offset, mcv); /* :392 */
            for (int i = 0; i < dc; i++) {                                   /* :415-450 */
                float *a = app + (size_t)c->col[numB + i] * q;
                float *cv = ctov + (size_t)(numB + i) * q;
                for (int g = 0; g < q; g++) {
                    cv[g] = mcv[i * q + g];
                    a[g] = mcv[i * q + g] + mvc[i * q + g];
                }
            }
        }
        orc_decision(app, N, q, decide);                                     /* :468 */
        synd = orc_syndrome(c, gf, decide);                                  /* :469 */
        if (sweep_decide)
            for (int n = 0; n < N; n++) sweep_decide[(size_t)iter * N + n] = (uint8_t)decide[n];
        if (sweep_crc) sweep_crc[iter] = orc_crc32(app, sizeof(float) * (uint64_t)N * q);
        if (sweep_synd) sweep_synd[iter] = synd;
        if (early_exit && synd == 0) break;                                  /* :470-471 */
    }
    if (synd_out) *synd_out = synd;
    if (app_out) memcpy(app_out, app, sizeof(float) * (size_t)N * q);
    if (ctov_out) memcpy(ctov_out, ctov, sizeof(float) * (size_t)E * q);
    free(app); free(ctov); free(mvc); free(mvc2); free(mcv); free(vl); free(vg);
    return iter + 1;                                                         /* :474 */
}

/* NB_LDPC.c:479-485: bit mismatches between decide[k] and the CODEWORD bits for k < K */
int orc_bit_errors(const orc_code *c, const orc_gf *gf, const int *decide, const int *codeword)
{
    int nb = 0;
    for (int k = 0; k < c->K; k++)
        nb += __builtin_popcount(gf->idx2bin[decide[k]] ^ gf->idx2bin[codeword[k]]);
    return nb;
}

void orc_draw_frame(const orc_code *c, const orc_gf *gf, const orc_encoder *enc, orc_rng *rng,
                    float sigma, int *codeword, float *noisy)
{
    int *ks = malloc(sizeof(int) * (size_t)c->K);
    orc_random_info(rng, c, gf, ks);                                         /* NB_LDPC.c:255 */
    orc_encode(enc, ks, codeword);                                           /* :258 */
    orc_awgn_bpsk(rng, c, gf, codeword, sigma, noisy);                       /* :261 */
    free(ks);
}

void orc_sim_frames(const orc_code *c, const orc_gf *gf, const orc_encoder *enc, orc_rng *rng,
                    float EbN, int nm, int nbOper, float offset, int NbIterMax, int early_exit,
                    int nframes, int *rec)
{
    const float sigma = orc_sigma(c->rate, EbN);
    int *cw = malloc(sizeof(int) * (size_t)c->N);
    int *decide = calloc((size_t)c->N, sizeof(int));
    float *noisy = malloc(sizeof(float) * (size_t)c->N * c->logq);
    float *llr = malloc(sizeof(float) * (size_t)c->N * c->q);
    for (int f = 0; f < nframes; f++) {
        int synd = 0;
        orc_draw_frame(c, gf, enc, rng, sigma, cw, noisy);
        orc_llr(c, gf, noisy, sigma, llr);
        int it = orc_decode_frame(c, gf, llr, nm, nbOper, offset, NbIterMax, NULL, early_exit, decide,
                                  &synd, NULL, NULL, NULL, NULL, NULL);
        rec[3 * f + 0] = orc_bit_errors(c, gf, decide, cw);
        rec[3 * f + 1] = synd;
        rec[3 * f + 2] = it;
    }
    free(cw); free(decide); free(noisy); free(llr);
}
