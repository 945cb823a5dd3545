/*
 * ems_sim_v2.c -- the Monte-Carlo harness of the OLDER generation (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1980-2311) driving the
 * MI355X decoder library: same seven arguments, same drand48 draw order (information bits v2.c:1025, Rayleigh-faded 256-point grid
 * constellation + AWGN v2.c:1167-1470, frame after frame, never reseeded), same statistics and the same five printed lines per
 * frame -- the bit errors sampled when iterations 9 / 19 / 29 / 39 / 49 begin, a frame counted as erroneous when a sample exceeds
 * 12 bit errors ("fake BCH"), `nbUndetectedErrors` raised once per such sample when the frame ends on a zero syndrome, the stop at
 * 100 erroneous frames of the fifth sample (v2.c:2174-2291).  What changes is what north_star asks for: the per-frame block
 * v2.c:1472-1497 (likelihoods + sort) and v2.c:2093-2253 (the decode loop) is one batched call sequence into the GPU library --
 * frames are drawn EMS_BATCH at a time on the host, decoded together (ems_v2_channel_likelihood, ems_v2_decode with snapshots),
 * and reported in frame order; frames drawn past the stop are dropped.
 *
 *   ems_sim_v2 NbMonteCarlo NbIterMax FileMatrix EbN NbMax Offset NbOper
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "ems_abi.h"
#include "nbldpc_ref_api.h"

#define PI 3.1415926536 /* v2.c:92 */

static void die(const char *what, int rc)
{
    fprintf(stderr, "\n%s failed (%d): %s\n", what, rc, ems_last_error());
    exit(EXIT_FAILURE);
}

int main(int argc, char *argv[])
{
    if (argc < 8) {
        printf("Syntax:\n %s\n ", argv[0]);
        printf("\n\t NbMonteCarlo     : # simulated frames ");
        printf("\n\t NbIterMax        : # of maximum decoding iterations");
        printf("\n\t FileMatrix       : File name of the parity-check matrix");
        printf("\n\t EbN              : Eb/No (dB) ");
        printf("\n\t NbMax            : size of truncated messages ");
        printf("\n\t Offset           : offset correction factor (0.4 -- 1) ");
        printf("\n\t NbOper           : Maximum number of operations for sorting \n");
        return EXIT_FAILURE;
    }
    const int NbMonteCarlo = atoi(argv[1]);
    const int NbIterMax = atoi(argv[2]);
    char *FileMatrix = argv[3];
    const float EbN = (float)atof(argv[4]);
    const int nbMax = atoi(argv[5]);
    const float offset = (float)atof(argv[6]);
    const int NbOper = atoi(argv[7]);
    const int device = getenv("EMS_DEVICE") ? atoi(getenv("EMS_DEVICE")) : 0;
    printf("\n\t NbMonteCarlo     : %d", NbMonteCarlo);
    printf("\n\t NbIterMax        : %d", NbIterMax);
    printf("\n\t FileMatrix       : %s", FileMatrix);
    printf("\n\t Eb/No (dB)       : %g", EbN);
    printf("\n\t NbMax            : %d", nbMax);
    printf("\n\t Offset           : %g", offset);
    printf("\n\t NbOper           : %d\n", NbOper);

    code_t code;
    table_t table;
    LoadCode(FileMatrix, &code);
    if (code.GF != 256) {                                   /* v2.c:937,972: the binary images and the constellation are GF(256)'s */
        fprintf(stderr, "the older generation's channel is a 256-point constellation: GF(256) codes only\n");
        return EXIT_FAILURE;
    }
    LoadTables(&table, code.GF, code.logGF);
    GaussianElimination(&code, &table);
    const int N = code.N, M = code.M, K = code.K, logGF = code.logGF;

    int *col = malloc(sizeof(int) * (size_t)code.nbBranch), *coef = malloc(sizeof(int) * (size_t)code.nbBranch);
    for (int m = 0, e = 0; m < M; m++)
        for (int k = 0; k < code.rowDegree[m]; k++, e++) {
            col[e] = code.mat[m][k];
            coef[e] = code.matValue[m][k];
        }
    ems_code_desc desc = {N, M, code.GF, logGF, code.nbBranch, code.rowDegree, col, coef};
    ems_params prm = {nbMax, NbOper, offset, NbIterMax, 1};
    int rc, batch = getenv("EMS_BATCH") ? atoi(getenv("EMS_BATCH")) : 0;
    /* ems_suggest_batch budgets the main path's per-frame state (APP rows + C->V records, ~10.5 MB for the 64800-bit GF(256) code); the v2
     * flavour allocates four more [N][q] blocks and the message records per frame on top (ems_core.cpp: v2_prepare, ~4.6 x): ask for the
     * share of five handles so that the HBM cap of the suggestion holds for this flavour too */
    if (batch <= 0 && (rc = ems_suggest_batch(&desc, &prm, device, 5, &batch))) die("ems_suggest_batch", rc);
    if (batch > NbMonteCarlo) batch = NbMonteCarlo;
    if (batch < 1) batch = 1;
    ems_handle *h;
    if ((rc = ems_create(&h, &desc, &prm, batch, device))) die("ems_create", rc);
    if ((rc = ems_v2_enable_snapshots(h, 1))) die("ems_v2_enable_snapshots", rc);
    fprintf(stderr, "[ems_sim_v2] batch %d frames per launch, device %d\n", batch, device);

    /* the 16 x 16 grid of v2.c:1185-1443, scaled by R (float product, :1446) */
    const float R = 0.433860915637f;
    float (*TABLEAU)[2] = malloc(sizeof(float[2]) * 256);
    for (int i = 0; i < 256; i++) {
        TABLEAU[i][0] = R * (-7.5f + (float)(i >> 4));
        TABLEAU[i][1] = R * (-7.5f + (float)(i & 15));
    }
    const float sigma = (float)sqrt(1.0 / (2.0 * pow(10, EbN / 10.0)));        /* v2.c:1451 */

    int **NBIN = calloc(N, sizeof(int *)), **KBIN = calloc(K, sizeof(int *));
    for (int n = 0; n < N; n++) NBIN[n] = calloc(logGF, sizeof(int));
    for (int k = 0; k < K; k++) KBIN[k] = calloc(logGF, sizeof(int));
    int *KSYMB = calloc(K, sizeof(int));
    int *cw = malloc(sizeof(int) * (size_t)batch * N);
    float *obs = malloc(sizeof(float) * (size_t)batch * N * 2), *att = malloc(sizeof(float) * (size_t)batch * N * 2);
    int *iters = malloc(sizeof(int) * batch), *synd = malloc(sizeof(int) * batch), *snap = malloc(sizeof(int) * 5 * (size_t)batch);

    int nbErroneousFrames[5] = {0, 0, 0, 0, 0}, nb_err[5] = {0, 0, 0, 0, 0}, nbUndetectedErrors = 0, nb = 0, stop = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int done = 0; done < NbMonteCarlo && !stop; done += batch) {
        const int B = NbMonteCarlo - done < batch ? NbMonteCarlo - done : batch;
        for (int b = 0; b < B; b++) {
            /* RandomBinaryGenerator (v2.c:1025): information symbols M..N-1, floor(drand48() * 2.0) per bit */
            for (int k = 0; k < K; k++) {
                for (int q = 0; q < logGF; q++) KBIN[k][q] = (int)floor(drand48() * 2.0);
                KSYMB[k] = Bin2GF(KBIN[k], code.GF, logGF, table.BINGF);
            }
            int *CodeWord = cw + (size_t)b * N;
            Encoding(&code, &table, CodeWord, NBIN, KSYMB);                     /* v2.c:1126: the same back substitution */
            /* ModelChannelAWGN, the draws (v2.c:1452-1468): one fading draw per node, two Box-Muller samples */
            for (int n = 0; n < N; n++) {
                int som = 0;
                for (int q = 0; q < 8; q++) som += NBIN[n][q] << q;
                const float w = (float)drand48();
                const float A = (float)sqrt(-log(w));
                float *o = obs + ((size_t)b * N + n) * 2, *a = att + ((size_t)b * N + n) * 2;
                a[0] = a[1] = A;                                                /* ATT[n] on both axes */
                for (int q = 0; q < 2; q++) {
                    const float u = (float)drand48();
                    const float v = (float)drand48();
                    o[q] = (float)(TABLEAU[som][q] * A + sigma * sqrt(-2.0 * log(u)) * cos(2.0 * PI * v));
                }
            }
        }
        /* v2.c:1472-1497 (likelihoods, decreasing sort, V->C initialisation) and v2.c:2119-2253 (iterations) on the GPU */
        if ((rc = ems_set_codewords(h, cw, B))) die("ems_set_codewords", rc);
        if ((rc = ems_v2_channel_likelihood(h, obs, att, &TABLEAU[0][0], sigma, B))) die("ems_v2_channel_likelihood", rc);
        if ((rc = ems_v2_decode(h, B))) die("ems_v2_decode", rc);
        if ((rc = ems_get_results(h, B, NULL, iters, synd))) die("ems_get_results", rc);
        if ((rc = ems_v2_get_snapshots(h, B, snap))) die("ems_v2_get_snapshots", rc);
        for (int b = 0; b < B; b++) {                                           /* v2.c:2255-2309, frame order */
            nb++;
            for (int j = 0; j < 5; j++)
                if (snap[b * 5 + j] > 12) {                                     /* "fake BCH": more than 12 bit errors */
                    nbErroneousFrames[j]++;
                    if (synd[b] == 0) nbUndetectedErrors++;
                }
            for (int j = 0; j < 5; j++)
                if (snap[b * 5 + j] > 12) nb_err[j] += snap[b * 5 + j];
            for (int j = 0; j < 5; j++) {
                printf("\r<%d> FER= %d/ %d = %g BER= %d/ %d = %g \n", nbUndetectedErrors, nbErroneousFrames[j], nb, (double)nbErroneousFrames[j] / nb,
                       nb_err[j], nb, (double)nb_err[j] / nb / N / logGF);
                fflush(stdout);
            }
            if (nbErroneousFrames[4] == 100) { stop = 1; break; }
        }
    }
    printf("\n");
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "[ems_sim_v2] %d frames, simulation loop wall %.3f s (%.1f frames/s incl. host generation and PCIe), batch %d\n", nb, dt,
            dt > 0 ? nb / dt : 0.0, batch);

    ems_destroy(h);
    free(col); free(coef); free(cw); free(obs); free(att); free(iters); free(synd); free(snap); free(KSYMB); free(TABLEAU);
    for (int n = 0; n < N; n++) free(NBIN[n]);
    for (int k = 0; k < K; k++) free(KBIN[k]);
    free(NBIN); free(KBIN);
    FreeCode(&code);
    FreeTable(&table);
    return EXIT_SUCCESS;
}
