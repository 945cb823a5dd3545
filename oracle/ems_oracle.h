/*
 * ems_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (CPU, single-thread) restatement of the reference's EMS NB-LDPC hot path, used ONLY as the
 * checker by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.  The product
 * library (ems-decoder-of-nb-ldpc-codes_amd/csrc) never includes, links or calls anything in this
 * directory and fails loudly when its HIP device code cannot run.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_vs_reference.py, run in
 * the build container where /root/reference exists) against the real reference compiled by
 * oracle/Makefile into oracle/_ref/, and against the golden vectors generated from that build and
 * committed under tests/golden/ (generator: tests/golden/make_golden.py).
 *
 * Each function cites the reference file:line it restates (paths relative to the reference root).
 */
#ifndef EMS_ORACLE_H
#define EMS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GF(q), q in {16,64,256}: symbol index 0 = zero, index i>=1 = alpha^(i-1)  (struct.h:117-476,
 *      init.c:37-130).  Two q-byte tables replace the reference's three q*q int tables. ---- */
typedef struct {
    int q, logq;
    uint8_t idx2bin[256]; /* binary image, bit l = BINGF[idx][l]            (struct.h BinGF_*) */
    uint8_t bin2idx[256]; /* inverse                                       (tools.c:95 Bin2GF) */
} orc_gf;

int orc_gf_init(orc_gf *gf, int q);                 /* 0 ok, -1 unsupported q (init.c:431-435)  */
int orc_gf_add(const orc_gf *gf, int a, int b);     /* ADDGF[a][b]           (init.c:37-55)     */
int orc_gf_mul(const orc_gf *gf, int a, int b);     /* MULGF[a][b]           (init.c:65-89)     */
int orc_gf_div(const orc_gf *gf, int a, int b);     /* DIVGF[a][b], -1 if b==0 (init.c:100-130) */

/* ---- parity-check matrix (init.c:143-272).  Flat CSR-by-row instead of int** rows. ---- */
typedef struct {
    int N, M, K, q, logq, E, dc_max;
    float rate;        /* (float)(N-M)/N  (init.c:168) */
    int *row_deg;      /* [M]   */
    int *col_deg;      /* [N]   */
    int *row_ptr;      /* [M+1] edge offset of each check node == the reference's running numB */
    int *col;          /* [E]   mat[m][k]      (0-based variable-node index)                    */
    int *coef;         /* [E]   matValue[m][k] (symbol index 1..q-1)                            */
} orc_code;

enum { ORC_FMT_AUTO = 0, ORC_FMT_UBS = 1, ORC_FMT_KN = 2, ORC_FMT_FULL = 3 };
orc_code *orc_load_code(const char *path, int fmt); /* NULL on failure */
void orc_free_code(orc_code *c);

/* ---- drand48-compatible LCG; glibc unseeded => X0 = 0 (tools.c:71-73) ---- */
typedef struct { uint64_t x; } orc_rng;
double orc_drand48(orc_rng *r);

/* ---- source + encoder (tools.c:124-268) ---- */
typedef struct orc_encoder orc_encoder;
orc_encoder *orc_encoder_create(const orc_code *c, const orc_gf *gf); /* GaussianElimination */
void orc_encoder_free(orc_encoder *e);
void orc_random_info(orc_rng *r, const orc_code *c, const orc_gf *gf, int *ksymb); /* tools.c:124 */
void orc_encode(const orc_encoder *e, const int *ksymb, int *codeword);           /* tools.c:232 */

/* ---- channel (channel.c:38-98) ---- */
float orc_sigma(float rate, float EbN);                                           /* channel.c:51 */
void orc_awgn_bpsk(orc_rng *r, const orc_code *c, const orc_gf *gf, const int *codeword,
                   float sigma, float *noisy /*[N][logq]*/);                      /* channel.c:52-62 */
void orc_llr(const orc_code *c, const orc_gf *gf, const float *noisy, float sigma,
             float *llr /*[N][q], indexed by symbol*/);                           /* channel.c:66-76 */
void orc_sort_llr(const float *llr, int N, int q, float *s_llr, int *s_gf);       /* channel.c:78-91 */

/* ---- check node (bubble_decoder.c) ---- */
int orc_elementary_step(const float *in1, const float *in2, const int *idx1, const int *idx2,
                        float *out, int *idxout, const orc_gf *gf, int nm, int nbOper); /* :316 */
void orc_check_node(int dc, int nm, const orc_gf *gf, const int *coef /*[dc]*/,
                    const float *vtoc_llr /*[dc][nm]*/, const int *vtoc_gf /*[dc][nm], unrotated*/,
                    int nbOper, float offset, float *ctov /*[dc][q] dense, by symbol*/);  /* :72 */

/* ---- hard decision + syndrome (tools.c:284-330) ---- */
void orc_decision(const float *app, int N, int q, int *decide);
int orc_syndrome(const orc_code *c, const orc_gf *gf, const int *decide);

/* ---- layered decode of one frame (NB_LDPC.c:266-474) ----
 * llr     [N][q]  channel LLR by symbol (== APP init, NB_LDPC.c:281-288)
 * order   NULL = reference order 0..M-1, else a permutation of check nodes (schedule tests)
 * early_exit 1 = reference behaviour (stop when syndrome==0), 0 = fixed-work benchmark mode
 * returns the value the harness adds to sum_it (iter+1, NB_LDPC.c:474).
 * Optional outputs (may be NULL): app_out [N][q], ctov_out [E][q] final state;
 * sweep_decide [NbIterMax-1][N] uint8 hard decisions after each executed sweep,
 * sweep_crc [NbIterMax-1] CRC-32 of the APP block after each executed sweep, sweep_synd likewise. */
int orc_decode_frame(const orc_code *c, const orc_gf *gf, const float *llr, int nm, int nbOper,
                     float offset, int NbIterMax, const int *order, int early_exit,
                     int *decide, int *synd_out, float *app_out, float *ctov_out,
                     uint8_t *sweep_decide, uint32_t *sweep_crc, int *sweep_synd);

/* ---- Monte-Carlo frames exactly as the reference's main loop draws them (NB_LDPC.c:250-507) ----
 * rec[f] = { bit errors over the first K symbols, syndrome, iter+1 } */
void orc_sim_frames(const orc_code *c, const orc_gf *gf, const orc_encoder *enc, orc_rng *rng,
                    float EbN, int nm, int nbOper, float offset, int NbIterMax, int early_exit,
                    int nframes, int *rec /*[nframes][3]*/);

/* one frame's channel draw only (source + encode + noise), for feeding the GPU the same stream */
void orc_draw_frame(const orc_code *c, const orc_gf *gf, const orc_encoder *enc, orc_rng *rng,
                    float sigma, int *codeword /*[N]*/, float *noisy /*[N][logq]*/);

int orc_bit_errors(const orc_code *c, const orc_gf *gf, const int *decide, const int *codeword);
uint32_t orc_crc32(const void *buf, uint64_t len);

#ifdef __cplusplus
}
#endif
/* ---- constellation channel front ends (ems_channel_oracle.c; channel.c:112-929) ---- */
void orc_const_normalise(float *table /*[q][D] in place*/, int q, int D);                /* channel.c:206-221, :565-577, :791-807 */
void orc_const_rotate(const float *qam /*[q][2]*/, int q, float *out /*[q][2]*/);        /* channel.c:347-358 (QAM_R)            */
void orc_const_apsk64(const double *radius, const int *num, const int *den, float *out /*[64][2]*/); /* channel.c:132-199 */
void orc_const_apsk256(const float *radius8, float *out /*[256][2]*/);                              /* channel.c:361-375 */
float orc_chan_sigma(float EbN);                                                         /* channel.c:233 / :596 / :810          */
void orc_chan_draw(orc_rng *r, int kind, const float *table, int D, int N, const int *images, float sigma,
                   float *obs /*[N][D]*/, float *att /*[N][D]*/);
void orc_const_llr(const orc_gf *gf, const float *table, int q, int D, int mode, const float *obs, const float *att, int N,
                   float sigma, float *llr /*[N][q]*/);
/* the older generation's own channel (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1167-1499): grid table, draws, decreasing sort */
void orc_v2_table(float *out /*[256][2]*/);
void orc_v2_chan_draw(orc_rng *r, const float *table, int N, const int *images, float sigma, float *obs, float *att);
void orc_v2_sort_likelihood(const float *like, int N, int q, float *s_like, int *s_idx);

#endif
