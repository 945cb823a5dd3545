/* oracle/ref_v2_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Driver around the REAL v2 reference (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c), compiled by oracle/Makefile
 * into oracle/_ref/libref_v2.so from the source where it lies under /root/reference: the reference file is
 * #included (its main() renamed), nothing of it is copied into this repository.  Everything below the
 * #include is this repo's own glue: plain-pointer entry points that (a) call the reference's
 * ElementaryStep / CheckPassLogEMS / DataPassLog / ComputeAPPLog / Decision / Syndrom on caller-supplied
 * data and (b) run the decode loop of the reference's main() (v2.c:2101-2253) on a given likelihood set --
 * main() itself cannot be used for that: it draws its own frames and only prints statistics.
 * tests/test_v2_oracle_vs_reference.py uses it to pin oracle/ems_v2_oracle.c.
 */
#define main ems_refv2_main
#include REFV2_SRC
#undef main

typedef struct {
    code_t code;
    table_t table;
    decoder_t dec;
    int *Flag, *TMPindex, *decide;
    float *TMP;
} refv2_ctx;

void *refv2_open(const char *matrix, int nbMax)
{
    refv2_ctx *c = calloc(1, sizeof *c);
    LoadCode((char *)matrix, &c->code);
    LoadTables(&c->table, c->code.GF, c->code.logGF);
    c->dec.nbMax = nbMax;
    AllocateDecoder(&c->code, &c->dec);
    c->Flag = calloc(c->code.N, sizeof(int));
    c->TMP = calloc(c->code.GF, sizeof(float));
    c->TMPindex = calloc(c->code.GF, sizeof(int));
    c->decide = calloc(c->code.N, sizeof(int));
    return c;
}

void refv2_info(void *h, int *out /* N M E GF logGF dc0 */)
{
    refv2_ctx *c = h;
    out[0] = c->code.N; out[1] = c->code.M; out[2] = c->code.nbBranch; out[3] = c->code.GF; out[4] = c->code.logGF;
    out[5] = c->code.rowDegree[0];
}

/* The reference fills BINGF only for GF(256) (v2.c:972-978); for other fields the caller supplies the tables. */
void refv2_set_tables(void *h, const int *add, const int *mul, const int *div)
{
    refv2_ctx *c = h;
    const int q = c->code.GF;
    for (int a = 0; a < q; a++)
        for (int b = 0; b < q; b++) {
            c->table.ADDGF[a][b] = add[a * q + b];
            c->table.MULGF[a][b] = mul[a * q + b];
            c->table.DIVGF[a][b] = div[a * q + b];
        }
}
void refv2_get_tables(void *h, int *add, int *mul, int *div, int *bin)
{
    refv2_ctx *c = h;
    const int q = c->code.GF;
    for (int a = 0; a < q; a++) {
        for (int b = 0; b < q; b++) {
            add[a * q + b] = c->table.ADDGF[a][b];
            mul[a * q + b] = c->table.MULGF[a][b];
            div[a * q + b] = c->table.DIVGF[a][b];
        }
        for (int l = 0; l < c->code.logGF; l++) bin[a * c->code.logGF + l] = c->table.BINGF[a][l];
    }
}
void refv2_get_graph(void *h, int *interleaver, int *ntob /* [N][2] */)
{
    refv2_ctx *c = h;
    for (int e = 0; e < c->code.nbBranch; e++) interleaver[e] = c->code.interleaver[e];
    for (int n = 0; n < c->code.N; n++)
        for (int k = 0; k < c->code.columnDegree[n] && k < 2; k++) ntob[2 * n + k] = c->code.NtoB[n][k];
}

int refv2_elementary_step(void *h, float *in1, float *in2, int *i1, int *i2, float *out, int *iout, int nbMax, int nbOper,
                          float offset)
{
    refv2_ctx *c = h;
    return ElementaryStep(in1, in2, i1, i2, out, iout, c->table.ADDGF, c->code.GF, nbMax, nbOper, offset);
}

/* one check node: vc/ivc [dc][nbMax] in, cv/icv [dc][nbMax] out (v2.c:2125-2141 around CheckPassLogEMS) */
void refv2_check_node(void *h, int node, const float *vc, const int *ivc, float *cv, int *icv, float offset, int NbOper)
{
    refv2_ctx *c = h;
    const int dc = c->code.rowDegree[node], nm = c->dec.nbMax;
    for (int i = 0; i < dc; i++)
        for (int k = 0; k < nm; k++) {
            c->dec.VC_arr[i][k] = vc[i * nm + k];
            c->dec.IndiceVC_arr[i][k] = ivc[i * nm + k];
        }
    CheckPassLogEMS(node, &c->dec, &c->code, &c->table, offset, NbOper);
    for (int i = 0; i < dc; i++)
        for (int k = 0; k < nm; k++) {
            cv[i * nm + k] = c->dec.CV_arr[i][k];
            icv[i * nm + k] = c->dec.IndiceCV_arr[i][k];
        }
}

/* one variable-node update: sorted likelihood row + one C->V message in, V->C message out (v2.c:2146-2157) */
void refv2_data_pass(void *h, const float *like, const int *like_idx, const float *cv, const int *icv, float offset, float *vc,
                     int *ivc)
{
    refv2_ctx *c = h;
    const int q = c->code.GF, nm = c->dec.nbMax;
    for (int g = 0; g < q; g++) { c->TMP[g] = like[g]; c->TMPindex[g] = like_idx[g]; }
    for (int k = 0; k < nm; k++) { c->dec.CV[k] = cv[k]; c->dec.IndiceCV[k] = icv[k]; }
    DataPassLog(&c->code, &c->dec, c->TMP, c->TMPindex, offset);
    for (int k = 0; k < nm; k++) { vc[k] = c->dec.VC[k]; ivc[k] = c->dec.IndiceVC[k]; }
}

void refv2_set_likelihood(void *h, const float *like, const int *idx)
{
    refv2_ctx *c = h;
    const int N = c->code.N, q = c->code.GF;
    ClearDecoder(&c->dec, q);                                   /* v2.c:2084 */
    for (int n = 0; n < N; n++)
        for (int g = 0; g < q; g++) {
            c->dec.likelihood[n][g] = like[(size_t)n * q + g];
            c->dec.likelihood_index[n][g] = idx[(size_t)n * q + g];
        }
}

/* the reference's own channel on the all-zero codeword (a codeword of every linear code), seeded stream */
void refv2_channel_zero_codeword(void *h, float EbN, long seed, float *like, int *idx)
{
    refv2_ctx *c = h;
    const int N = c->code.N, q = c->code.GF;
    int **NBIN = calloc(N, sizeof(int *));
    for (int n = 0; n < N; n++) NBIN[n] = calloc(c->code.logGF, sizeof(int));
    srand48(seed);
    ClearDecoder(&c->dec, q);
    ModelChannelAWGN(&c->code, &c->dec, &c->table, NBIN, EbN);
    for (int n = 0; n < N; n++) {
        for (int g = 0; g < q; g++) {
            like[(size_t)n * q + g] = c->dec.likelihood[n][g];
            idx[(size_t)n * q + g] = c->dec.likelihood_index[n][g];
        }
        free(NBIN[n]);
    }
    free(NBIN);
}

/* The decode loop of the reference's main() (v2.c:2093-2253) on the likelihoods set by refv2_set_likelihood:
 * returns the number of iterations executed (iter+1 at the break, NbIterMax+1 if the syndrome never cleared). */
int refv2_decode(void *h, int NbIterMax, int NbOper, float offset, int *decide_out, int *synd_out)
{
    refv2_ctx *c = h;
    code_t *code = &c->code;
    decoder_t *d = &c->dec;
    const int nm = d->nbMax;
    int numB = 0, synd = 0, iter, done = 0;
    for (int n = 0; n < code->N; n++) {                                         /* v2.c:2101-2113 */
        for (int t = 0; t < code->columnDegree[n]; t++) {
            const int b = code->interleaver[numB + t];
            for (int k = 0; k < nm; k++) { d->VtoC[b][k] = d->likelihood[n][k]; d->VtoCindex[b][k] = d->likelihood_index[n][k]; }
        }
        numB += code->columnDegree[n];
    }
    for (iter = 0; iter <= NbIterMax; iter++) {                                 /* v2.c:2119 */
        numB = 0;
        for (int node = 0; node < code->M; node++) {
            const int dc = code->rowDegree[node];
            for (int i = 0; i < dc; i++)
                for (int k = 0; k < nm; k++) { d->VC_arr[i][k] = d->VtoC[numB + i][k]; d->IndiceVC_arr[i][k] = d->VtoCindex[numB + i][k]; }
            CheckPassLogEMS(node, d, code, &c->table, offset, NbOper);
            for (int i = 0; i < dc; i++)
                for (int k = 0; k < nm; k++) { d->CtoV[numB + i][k] = d->CV_arr[i][k]; d->CtoVindex[numB + i][k] = d->IndiceCV_arr[i][k]; }
            for (int i = 0; i < dc; i++) {
                const int vn = code->mat[node][i];
                for (int k = 0; k < code->GF; k++) { c->TMP[k] = d->likelihood[vn][k]; c->TMPindex[k] = d->likelihood_index[vn][k]; }
                for (int k = 0; k < nm; k++) { d->CV[k] = d->CtoV[i + numB][k]; d->IndiceCV[k] = d->CtoVindex[i + numB][k]; }
                DataPassLog(code, d, c->TMP, c->TMPindex, offset);
                c->Flag[vn] = 1 ^ c->Flag[vn];
                const int dst = code->NtoB[vn][c->Flag[vn]];
                for (int k = 0; k < nm; k++) { d->VtoC[dst][k] = d->VC[k]; d->VtoCindex[dst][k] = d->IndiceVC[k]; }
            }
            numB += dc;
        }
        ComputeAPPLog(code, d, offset);
        Decision(c->decide, d->APP, code->N, code->GF);
        synd = Syndrom(code, c->decide, &c->table);
        done = iter + 1;
        if (synd == 0) break;
    }
    for (int n = 0; n < code->N; n++) decide_out[n] = c->decide[n];
    *synd_out = synd;
    return done;
}

void refv2_get_state(void *h, float *vtoc, int *vtoci, float *ctov, int *ctovi, float *app)
{
    refv2_ctx *c = h;
    const size_t ne = (size_t)c->code.nbBranch * c->dec.nbMax, na = (size_t)c->code.N * c->code.GF;
    memcpy(vtoc, c->dec.VtoC[0], ne * sizeof(float));
    memcpy(vtoci, c->dec.VtoCindex[0], ne * sizeof(int));
    memcpy(ctov, c->dec.CtoV[0], ne * sizeof(float));
    memcpy(ctovi, c->dec.CtoVindex[0], ne * sizeof(int));
    memcpy(app, c->dec.APP[0], na * sizeof(float));
}

void refv2_close(void *h)
{
    refv2_ctx *c = h;
    /* (the reference's FreeCode/FreeTable/FreeDecoder are not called: FreeCode faults on small codes; a test
     *  process can afford to leave those blocks to exit) */
    free(c->Flag); free(c->TMP); free(c->TMPindex); free(c->decide);
    free(c);
}
