/*
 * nbldpc_ref_api.h -- the reference simulator's own C boundary, re-exported by libems_hip.so.
 *
 * The reference (Lcrypto/EMS-decoder-of-NB-LDPC-codes) has no plugin or FFI layer: its boundary is the
 * set of plain C functions NB_LDPC.c calls through its include/ headers.  This header declares the same symbols
 * with the same signatures and the same struct layouts (binary compatible with include/struct.h), so a
 * maintainer can link NB_LDPC.c against libems_hip.so instead of init.o/tools.o/channel.o/
 * bubble_decoder.o.  Each declaration cites the reference interface it replaces.
 *
 * Where the work happens:
 *   host  (harness side, not accelerated): LoadCode, LoadTables, allocators, RNG, encoder;
 *   GPU   (hand-written HIP, gfx950):      ModelChannel_AWGN_BPSK's LLR + sort, CheckPassLogEMS,
 *                                          ElementaryStep, Decision, Syndrom, Decoding.
 * The GPU-backed symbols return after the device work has completed and fail loudly (message on
 * stderr, non-zero status where the signature has one) when no HIP device is usable; none of them
 * falls back to a CPU implementation.
 *
 * This layer is a COMPATIBILITY SHIM, not the fast path: the reference's loop hands over one frame at a time, so
 * Decoding() is a batch of ONE on a decoder context the library keeps per code_t (one global table under a mutex,
 * created on first use, destroyed by FreeCode) -- about 105 ms per frame of the 64800-bit GF(256) code, the latency of a
 * frame's 441-level dependency chain, with the other 255 compute units idle.  It exists so that the reference's own main()
 * runs unchanged (oracle/Makefile builds NB_LDPC.c against it: oracle/_ref/essai_gpu); throughput needs the batched handle
 * API of ems_abi.h, which ems_sim.c shows.
 */
#ifndef NBLDPC_REF_API_H
#define NBLDPC_REF_API_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types: layout-identical to struct.h:48-108 (oracle/abi_layout_check.c asserts it at compile time against the reference's
 *      own header).  A translation unit that already has the reference's struct.h defines NBLDPC_REF_HAVE_STRUCT_H and gets the
 *      prototypes below declared on the reference's types (oracle/abi_proto_check.c: the compiler then checks every signature
 *      against init.h / tools.h / channel.h / bubble_decoder.h). ---- */
#ifndef NBLDPC_REF_HAVE_STRUCT_H
typedef float softdata_t; /* struct.h:82 */

typedef struct {          /* struct.h:48-64 */
    int N;                /* columns of H (code length in GF symbols)          */
    int M;                /* rows of H                                         */
    int K;                /* N - M                                             */
    int GF;               /* field order                                       */
    int logGF;            /* bits per symbol                                   */
    int **mat;            /* [M][rowDegree[m]] variable node of each edge      */
    int **matValue;       /* [M][rowDegree[m]] coefficient (symbol index)      */
    int *rowDegree;       /* [M]                                               */
    int *columnDegree;    /* [N]                                               */
    int nbBranch;         /* number of edges                                   */
    float rate;
    int **matUT;          /* upper-triangular form used by Encoding            */
    int *Perm;            /* column permutation of the elimination             */
} code_t;

typedef struct {          /* struct.h:71-77; each is a row-pointer array over one block (X[0]) */
    int **BINGF;          /* [GF][logGF] */
    int **ADDGF;          /* [GF][GF]    */
    int **MULGF;          /* [GF][GF]    */
    int **DIVGF;          /* [GF][GF], DIVGF[a][0] = -1 */
} table_t;

typedef struct {          /* struct.h:88-108 */
    int N;
    int nbMax;            /* nm; set by the caller BEFORE AllocateDecoder (NB_LDPC.c:109) */
    int nbBranch;
    softdata_t **CtoV;           /* [nbBranch][GF] */
    softdata_t **intrinsic_LLR;  /* [N][GF] ascending */
    int **intrinsic_GF;          /* [N][GF] */
    softdata_t **APP;            /* [N][GF] */
    softdata_t **M_VtoC_LLR;     /* [rowDegree[0]][nbMax] */
    int **M_VtoC_GF;
    softdata_t **M_CtoV_LLR;     /* [rowDegree[0]][GF] */
    int **M_CtoV_GF;
} decoder_t;
#endif /* NBLDPC_REF_HAVE_STRUCT_H */

/* ---- init.h:11-29 (host) ---- */
void Table_Add_GF(table_t *table, int GF, int logGF);          /* init.h:11 */
void Table_Mul_GF(int **MULGF, int GF);                        /* init.h:13 */
void Table_Div_GF(int **DIVGF, int GF);                        /* init.h:15 */
void LoadCode(char *FileMatrix, code_t *code);                 /* init.h:19; reads BOTH alist layouts */
void FreeCode(code_t *code);                                   /* init.h:21 */
void AllocateDecoder(code_t *code, decoder_t *decoder);        /* init.h:23 */
void FreeDecoder(decoder_t *decoder);                          /* init.h:25 */
void LoadTables(table_t *table, int GF, int logGF);            /* init.h:27 */
void FreeTable(table_t *table);                                /* init.h:29 */

/* ---- tools.h:18-30 ---- */
float My_drand48(int *initialise);                                                         /* tools.h:18 host */
int Bin2GF(int *U, int GF, int logGF, int **BINGF);                                        /* tools.h:20 host */
void RandomBinaryGenerator(int N, int M, int GF, int logGF, int **KBIN, int *KSYMB,
                           int **BINGF, int *init_rand);                                   /* tools.h:22 host */
void GaussianElimination(code_t *code, table_t *table);                                    /* tools.h:24 host */
int Encoding(code_t *code, table_t *table, int *CodeWord, int **NBIN, int *KSYMB);         /* tools.h:26 host */
int Syndrom(code_t *code, int *decide, table_t *tableGF);                                  /* tools.h:28 GPU  */
void Decision(int *decision, float **APP, int N, int GF);                                  /* tools.h:30 GPU  */

/* ---- channel.h:17: noise drawn on the host from drand48 (same stream as the reference), LLR
 *      computation and the ascending sort on the GPU ---- */
void ModelChannel_AWGN_BPSK(code_t *code, decoder_t *decoder, table_t *table, int **NBIN, float EbN,
                            int *init_rand);

/* ---- NEW: the LLR + sort part of the constellation front ends -- ModelChannel_AWGN_64 (channel.c:270-304),
 *      ModelChannel (:676-730) and ModelChannel_AWGN_256QAM_4D (:861-915) -- hoisted into one function (SURVEY.md
 *      8(f) row f3).  The draws, the constellation construction and its normalisation stay where they are in
 *      channel.c; INTEGRATION.md section 5 shows the three-line forward.  obs / att: [N][D] flat (NoisyBin rows,
 *      attenuation[n][d]; att may be NULL with mode EMS_LLR_SUM), table: [GF][D] indexed by the binary image `som`,
 *      mode: EMS_LLR_SUM (0) or EMS_LLR_SSD (1) of ems_abi.h.  Fills decoder->intrinsic_LLR / intrinsic_GF. ---- */
void ModelChannelLLR(code_t *code, decoder_t *decoder, const float *obs, const float *att, const float *table, int D,
                     int mode, float sigma);

/* ---- channel.h:20-26 (round 5): the reference's other channel front ends under their own names -- the calls NB_LDPC.c:262-264 keeps
 *      commented out next to ModelChannel_AWGN_BPSK.  Constellation construction and the draws on the host (the reference's drand48
 *      order and promotions, channel.c:112-266 / :328-680 / :749-860, as its shipped switches channel.c:20-26 compile them: rotated QAM,
 *      independent Rayleigh fading per axis, erasures), LLR rows + ascending sort on the GPU through ModelChannelLLR.  GF(64) /
 *      GF(16, 64, 256) / GF(256); anything else ends the process with a message, as the void signatures leave no other way. ---- */
void ModelChannel_AWGN_64(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand);        /* channel.h:20 */
void ModelChannel(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand);                /* channel.h:22 */
void ModelChannel_AWGN_256QAM_4D(code_t *code, decoder_t *decoder, int **NBIN, float EbN, int *init_rand); /* channel.h:24 */
void build_QAM_table(int logGF);                                                                           /* channel.h:26 (prints a grid) */

/* ---- bubble_decoder.h:15-17,26 ---- */
int maximum(float *Y, int nl);                                                             /* host helper */
int minimum(float *Y, int nl);                                                             /* host helper */
void CheckPassLogEMS(int node, decoder_t *decoder, code_t *code, table_t *table, int NbOper,
                     float offset);                                                        /* GPU */
int ElementaryStep(float *Input1, float *Input2, int *IndiceInput1, int *IndiceInput2, float *Output,
                   int *IndiceOut, int **ADDGF, int GF, int nbMax, int nbOper);            /* GPU */

/* ---- NEW: the per-frame decode block the reference keeps inline in main() (NB_LDPC.c:266-472),
 *      hoisted into a function.  Reads decoder->intrinsic_LLR/GF (as ModelChannel_* left them), runs
 *      NbIterMax-1 layered sweeps with per-sweep Decision + Syndrom and early exit on the GPU, leaves
 *      decoder->APP and decoder->CtoV as the reference would, fills decide[N] and *synd, and returns
 *      the value main() adds to sum_it (iter + 1, NB_LDPC.c:474); negative on failure. ---- */
int Decoding(code_t *code, table_t *table, decoder_t *decoder, int NbIterMax, int NbOper, float offset,
             int *decide, int *synd);

/* ---- NEW helper for batched harnesses: channel.c:51-62 only (sigma + BPSK + Box-Muller noise from
 *      drand48), observations written to noisy[N][logGF]; returns sigma. ---- */
float ems_awgn_bpsk_noise(code_t *code, int **NBIN, float EbN, int *init_rand, float *noisy);
/* NEW (batched harness): frames [first_frame, first_frame+count) (0-based) of the reference's drand48 stream --
 * the codewords RandomBinaryGenerator + Encoding (tools.c:124, :232) and the samples the noise draw of
 * ModelChannel_AWGN_BPSK (channel.c:51-62) would have produced frame after frame from X0 = 0 -- generated by
 * `threads` host threads (LCG jump-ahead; libc's drand48 state is not touched).  codewords[count][N],
 * noisy[count][N][logGF]; needs code->matUT/Perm (GaussianElimination).  Returns 0, or -1 on bad arguments. */
int ems_draw_frames(code_t *code, table_t *table, long first_frame, int count, float EbN, int threads,
                    int *codewords, float *noisy, float *sigma_out);

#ifdef __cplusplus
}
#endif
#endif /* NBLDPC_REF_API_H */
