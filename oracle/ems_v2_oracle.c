/* oracle/ems_v2_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Plain-C restatement of the decode core of the reference's older, self-contained generation
 * EMS_HS_L-BubbleCheck_UBS_decoder_v2.c: max-is-best log-likelihoods, truncated
 * nm-long messages in both directions, 4-candidate bubble check, horizontal (layered) schedule.
 * Every function cites the reference lines it follows; tests/test_v2_oracle_vs_reference.py pins each of
 * them, and whole-frame decodes, against the real functions (oracle/_ref/libref_v2.so).
 * Flat arrays only; GF arithmetic through q x q index tables (add, mul, div) supplied by the caller.
 */
#include <stdlib.h>
#include <string.h>

#define V2_MAX_NM 64
#define V2_MAX_DC 32
#define V2_MAX_Q 256

/* v2.c:1705-1720: first strict maximum ABOVE 0.0 (the scan starts from aux = 0.0), competitor 0 otherwise */
static int v2o_maximum(const float *y, int nl)
{
    float aux = 0.0f;
    int pos = 0;
    for (int i = 0; i < nl; i++)
        if (y[i] > aux) { aux = y[i]; pos = i; }
    return pos;
}

/* v2.c:1547-1701  ElementaryStep, nl = 4 competitors.  Output may alias Input1 (the reference works on local
 * copies, :1589-1592 / :1696).  Returns the number of valid outputs (the reference returns nothing useful). */
int v2o_elementary_step(const float *in1, const float *in2, const int *idx1, const int *idx2, float *out, int *iout,
                        const int *add, int q, int nm, int nbOper)
{
    enum { NL = 4 };
    float loc_out[V2_MAX_NM];
    int loc_idx[V2_MAX_NM];
    static _Thread_local float tab_aux[V2_MAX_NM][V2_MAX_NM];
    unsigned char seen[V2_MAX_Q];
    float comp_v[NL];
    int comp_r[NL], comp_c[NL];
    memset(seen, 0, sizeof seen);                                        /* :1588 */
    for (int i = 0; i < nm; i++) { loc_out[i] = -1000.0f; loc_idx[i] = -1; }   /* :1590-1593 */
    for (int i = 0; i < nm; i++)
        for (int j = 0; j < nm; j++) tab_aux[i][j] = in1[i] + in2[j];           /* :1598-1604 */
    for (int j = 0; j < NL - 1; j++) { comp_v[j] = tab_aux[j][0]; comp_r[j] = j; comp_c[j] = 0; }   /* :1620-1626 */
    comp_v[NL - 1] = tab_aux[NL - 2][1]; comp_r[NL - 1] = NL - 2; comp_c[NL - 1] = 1;               /* :1628-1631 */
    int s = 0;
    for (int ss = 0; ss < nbOper; ss++) {                                 /* :1648 */
        const int pos = v2o_maximum(comp_v, NL);
        if (idx1[comp_r[pos]] == -1 || idx2[comp_c[pos]] == -1) break;    /* :1653-1656 */
        const int sym = add[idx1[comp_r[pos]] * q + idx2[comp_c[pos]]];
        if (!seen[sym]) {                                                 /* :1662-1667 */
            loc_out[s] = comp_v[pos];
            loc_idx[s] = sym;
            seen[sym] = 1;
            s++;
        }
        if (s == nm) break;                                               /* :1669 */
        if (comp_r[pos] >= nm - 1 || comp_c[pos] >= nm - 1) break;        /* :1674-1677 */
        const int u = pos >> 1;                                           /* :1682-1688: competitors 0,1 walk right, 2,3 walk down */
        comp_r[pos] += u;
        comp_c[pos] += 1 - u;
        comp_v[pos] = tab_aux[comp_r[pos]][comp_c[pos]];
    }
    const int stp = s;                                                    /* :1692 */
    for (int i = 0; i < stp; i++) loc_out[i] = loc_out[i] - loc_out[stp - 1];   /* :1693 (in order: the last entry is zeroed last) */
    for (int i = 0; i < nm; i++) { out[i] = loc_out[i]; iout[i] = loc_idx[i]; }  /* :1695 */
    return s;
}

/* v2.c:1732-1876  CheckPassLogEMS on one check node of degree dc with coefficients coef[dc]:
 * vc/ivc [dc][nm] in (not modified: the rotation works on a copy, as the reference's VC_arr copy does),
 * cv/icv [dc][nm] out. */
void v2o_check_node(int dc, const int *coef, const float *vc, const int *ivc, float *cv, int *icv, const int *add,
                    const int *mul, const int *divt, int q, int nm, int nbOper)
{
    const int S = 2 * (dc - 2);
    static _Thread_local float inter[2 * V2_MAX_DC][V2_MAX_NM];
    static _Thread_local int interi[2 * V2_MAX_DC][V2_MAX_NM];
    static _Thread_local int rot[V2_MAX_DC][V2_MAX_NM];
    float fw[V2_MAX_NM], bw[V2_MAX_NM], fw1[V2_MAX_NM], bw1[V2_MAX_NM];
    int fwi[V2_MAX_NM], bwi[V2_MAX_NM], fwi1[V2_MAX_NM], bwi1[V2_MAX_NM];
    for (int k = 0; k < S; k++)
        for (int k1 = 0; k1 < nm; k1++) { inter[k][k1] = -1e5f; interi[k][k1] = -1; }      /* :1759-1765 */
    for (int t = 0; t < dc; t++)                                                          /* :1772-1784 rotation */
        for (int k = 0; k < nm; k++) rot[t][k] = ivc[t * nm + k] != -1 ? mul[ivc[t * nm + k] * q + coef[t]] : -1;
    for (int k = 0; k < nm; k++) {                                                        /* :1789-1795 */
        fw[k] = vc[0 * nm + k]; fwi[k] = rot[0][k];
        bw[k] = vc[(dc - 1) * nm + k]; bwi[k] = rot[dc - 1][k];
    }
    for (int kk = 1; kk < dc - 1; kk++) {                                                 /* :1798 */
        for (int k = 0; k < nm; k++) {
            fw1[k] = vc[kk * nm + k]; fwi1[k] = rot[kk][k];
            bw1[k] = vc[(dc - kk - 1) * nm + k]; bwi1[k] = rot[dc - kk - 1][k];
        }
        for (int k = 0; k < nm; k++) {                                                    /* :1811-1818 */
            inter[kk - 1][k] = fw[k]; interi[kk - 1][k] = fwi[k];
            inter[S - kk][k] = bw[k]; interi[S - kk][k] = bwi[k];
        }
        v2o_elementary_step(fw, fw1, fwi, fwi1, fw, fwi, add, q, nm, nbOper);             /* :1822 */
        v2o_elementary_step(bw, bw1, bwi, bwi1, bw, bwi, add, q, nm, nbOper);             /* :1827 */
    }
    for (int k = 0; k < nm; k++) {                                                        /* :1833-1842 */
        cv[(dc - 1) * nm + k] = fw[k]; icv[(dc - 1) * nm + k] = fwi[k];
        cv[0 * nm + k] = bw[k]; icv[0 * nm + k] = bwi[k];
    }
    for (int k = 0; k < dc - 2; k++) {                                                    /* :1845-1850 merges */
        v2o_elementary_step(inter[k], inter[(dc - 2) + k], interi[k], interi[(dc - 2) + k], fw, fwi, add, q, nm, nbOper);
        for (int g = 0; g < nm; g++) { cv[(k + 1) * nm + g] = fw[g]; icv[(k + 1) * nm + g] = fwi[g]; }
    }
    for (int t = 0; t < dc; t++) {                                                        /* :1854-1870 inverse rotation */
        int stp = nm;
        for (int k = 0; k < nm; k++)
            if (icv[t * nm + k] == -1) { stp = k; break; }
        for (int k = 0; k < stp; k++) icv[t * nm + k] = divt[icv[t * nm + k] * q + coef[t]];
    }
}

/* v2.c:1513-1545  DataPassLog: like/like_idx = the node's q sorted likelihoods and their symbols,
 * cv/icv = one incoming C->V message; vc/ivc = the nm largest of the sum, normalised to the last one. */
void v2o_data_pass(const float *like, const int *like_idx, const float *cv, const int *icv, float offset, float *vc, int *ivc,
                   int q, int nm)
{
    float tmpp[V2_MAX_Q];
    for (int g = 0; g < V2_MAX_Q; g++) tmpp[g] = 0.0f;                                    /* :1522 */
    for (int g = 0; g < q; g++) tmpp[like_idx[g]] = like[g] - offset;                     /* :1524 */
    for (int k = 0; k < nm; k++) {                                                        /* :1526-1531 */
        if (icv[k] != -1) tmpp[icv[k]] = tmpp[icv[k]] + cv[k] + offset;
        else break;
    }
    for (int k = 0; k < nm; k++) {                                                        /* :1536-1541 */
        vc[k] = -1e5f; ivc[k] = 0;
        for (int g = 0; g < q; g++)
            if (tmpp[g] > vc[k]) { vc[k] = tmpp[g]; ivc[k] = g; }
        tmpp[ivc[k]] = -1e5f;
    }
    for (int g = 0; g < nm; g++) vc[g] = vc[g] - vc[nm - 1];                              /* :1545 */
}

/* graph helpers of LoadCode (v2.c:682-706): edges are numbered row by row; ntob[n][k] = k-th edge of column n in
 * that numbering, interleaver = the same read column by column.  Returns -1 if some column has more than dv_max edges. */
int v2o_graph(int N, int M, const int *row_deg, const int *col, int *col_deg, int *ntob, int dv_max, int *interleaver)
{
    int e = 0;
    for (int n = 0; n < N; n++) col_deg[n] = 0;
    for (int m = 0; m < M; m++)
        for (int k = 0; k < row_deg[m]; k++, e++) {
            const int n = col[e];
            if (col_deg[n] >= dv_max) return -1;
            ntob[n * dv_max + col_deg[n]++] = e;
        }
    int b = 0;
    for (int n = 0; n < N; n++)
        for (int k = 0; k < col_deg[n]; k++) interleaver[b++] = ntob[n * dv_max + k];
    return 0;
}

/* v2.c:1887-1921  ComputeAPPLog */
void v2o_compute_app(int N, int E, int q, int nm, const int *col_deg, const int *interleaver, const float *like,
                     const int *like_idx, const float *ctov, const int *ctovi, float offset, float *app, float *tmp /* [E][q] scratch */)
{
    for (int e = 0; e < E; e++) {
        for (int g = 0; g < q; g++) tmp[(size_t)e * q + g] = -offset;
        for (int g = 0; g < nm; g++)
            if (ctovi[(size_t)e * nm + g] != -1) tmp[(size_t)e * q + ctovi[(size_t)e * nm + g]] = ctov[(size_t)e * nm + g];
    }
    int numB = 0;
    for (int n = 0; n < N; n++) {
        for (int g = 0; g < q; g++) app[(size_t)n * q + like_idx[(size_t)n * q + g]] = like[(size_t)n * q + g];
        for (int g = 0; g < q; g++)
            for (int t = numB; t < numB + col_deg[n]; t++)
                app[(size_t)n * q + g] = app[(size_t)n * q + g] + tmp[(size_t)interleaver[t] * q + g];
        numB += col_deg[n];
    }
}

/* v2.c:1961-1975  Decision: first strict maximum, starting from symbol 0 */
void v2o_decision(int N, int q, const float *app, int *decide)
{
    for (int n = 0; n < N; n++) {
        float max = app[(size_t)n * q];
        int ind = 0;
        for (int g = 1; g < q; g++)
            if (app[(size_t)n * q + g] > max) { max = app[(size_t)n * q + g]; ind = g; }
        decide[n] = ind;
    }
}

/* v2.c:1934-1949  Syndrom: running sum, stop after the first row that leaves it non-zero */
int v2o_syndrome(int M, const int *row_deg, const int *col, const int *coef, const int *decide, const int *add, const int *mul, int q)
{
    int synd = 0, e = 0;
    for (int k = 0; k < M; k++) {
        for (int l = 0; l < row_deg[k]; l++, e++) synd = add[synd * q + mul[coef[e] * q + decide[col[e]]]];
        if (synd != 0) break;
    }
    return synd;
}

/* The frame decode of main() (v2.c:2093-2253): VtoC initialised from the nm best likelihoods of each node,
 * then per iteration every check node in order: CheckPassLogEMS, and for each of its edges DataPassLog into
 * the OTHER edge of that variable node (Flag toggles, dv = 2), then ComputeAPPLog + Decision + Syndrom with
 * exit on a zero syndrome.  iter runs 0..NbIterMax inclusive.  Returns the number of iterations executed.
 * State arrays (vtoc.. app) are outputs; early_exit = 0 runs all NbIterMax+1 iterations (fixed work). */
int v2o_decode(int N, int M, int q, int nm, const int *row_deg, const int *col, const int *coef, const float *like,
               const int *like_idx, int NbIterMax, int nbOper, float offset, int early_exit, const int *add, const int *mul,
               const int *divt, int *decide, int *synd_out, float *vtoc, int *vtoci, float *ctov, int *ctovi, float *app)
{
    int E = 0;
    for (int m = 0; m < M; m++) E += row_deg[m];
    int *col_deg = calloc(N, sizeof(int)), *ntob = calloc((size_t)N * 2, sizeof(int)), *interleaver = calloc(E, sizeof(int));
    int *flag = calloc(N, sizeof(int));
    float *tmp = malloc((size_t)E * q * sizeof(float));
    int done = -1, synd = 0;
    if (v2o_graph(N, M, row_deg, col, col_deg, ntob, 2, interleaver) == 0) {
        for (size_t i = 0; i < (size_t)E * nm; i++) { vtoc[i] = 0; ctov[i] = 0; vtoci[i] = -1; ctovi[i] = -1; }   /* ClearDecoder :866 */
        for (size_t i = 0; i < (size_t)N * q; i++) app[i] = 0;
        int numB = 0;
        for (int n = 0; n < N; n++) {                                           /* :2101-2113 */
            for (int t = 0; t < col_deg[n]; t++) {
                const int b = interleaver[numB + t];
                for (int k = 0; k < nm; k++) { vtoc[(size_t)b * nm + k] = like[(size_t)n * q + k]; vtoci[(size_t)b * nm + k] = like_idx[(size_t)n * q + k]; }
            }
            numB += col_deg[n];
        }
        float vcb[V2_MAX_DC * V2_MAX_NM], cvb[V2_MAX_DC * V2_MAX_NM], vc[V2_MAX_NM];
        int ivcb[V2_MAX_DC * V2_MAX_NM], icvb[V2_MAX_DC * V2_MAX_NM], ivc[V2_MAX_NM];
        for (int iter = 0; iter <= NbIterMax; iter++) {                          /* :2119 */
            numB = 0;
            for (int node = 0; node < M; node++) {
                const int dc = row_deg[node];
                memcpy(vcb, vtoc + (size_t)numB * nm, sizeof(float) * dc * nm);
                memcpy(ivcb, vtoci + (size_t)numB * nm, sizeof(int) * dc * nm);
                v2o_check_node(dc, coef + numB, vcb, ivcb, cvb, icvb, add, mul, divt, q, nm, nbOper);
                memcpy(ctov + (size_t)numB * nm, cvb, sizeof(float) * dc * nm);
                memcpy(ctovi + (size_t)numB * nm, icvb, sizeof(int) * dc * nm);
                for (int i = 0; i < dc; i++) {                                   /* :2146-2163 */
                    const int vn = col[numB + i];
                    v2o_data_pass(like + (size_t)vn * q, like_idx + (size_t)vn * q, ctov + (size_t)(numB + i) * nm,
                                  ctovi + (size_t)(numB + i) * nm, offset, vc, ivc, q, nm);
                    flag[vn] ^= 1;
                    const int dst = ntob[vn * 2 + flag[vn]];
                    memcpy(vtoc + (size_t)dst * nm, vc, sizeof(float) * nm);
                    memcpy(vtoci + (size_t)dst * nm, ivc, sizeof(int) * nm);
                }
                numB += dc;
            }
            v2o_compute_app(N, E, q, nm, col_deg, interleaver, like, like_idx, ctov, ctovi, offset, app, tmp);
            v2o_decision(N, q, app, decide);
            synd = v2o_syndrome(M, row_deg, col, coef, decide, add, mul, q);
            done = iter + 1;
            if (synd == 0 && early_exit) break;
        }
    }
    *synd_out = synd;
    free(col_deg); free(ntob); free(interleaver); free(flag); free(tmp);
    return done;
}
