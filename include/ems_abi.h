/*
 * ems_abi.h -- C ABI of the MI355X EMS NB-LDPC decoder library (libems_hip.so).
 *
 * Plain pointers and sizes only; no C++ / torch / HIP types cross this boundary.  All functions
 * return 0 on success or a negative EMS_E* code (the reference printf()s and exit()s instead, e.g.
 * init.c:431-435, tools.c:181-185; this library never exits the process).
 *
 * The batched entry points below are NEW: the reference decodes one frame at a time inline in main()
 * (NB_LDPC.c:266-472).  They are what the GPU path really uses; the reference-shaped single-frame
 * symbols (Decoding(), CheckPassLogEMS(), ... see nbldpc_ref_api.h) are thin wrappers over them.
 *
 * Every compute entry point runs hand-written HIP kernels on gfx950.  There is no CPU fallback: if
 * no HIP device is usable the calls fail with EMS_ENODEV.
 */
#ifndef EMS_ABI_H
#define EMS_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    EMS_OK = 0,
    EMS_EINVAL = -1,     /* bad argument / unsupported parameter combination                         */
    EMS_ENODEV = -2,     /* no usable HIP device (the library has NO CPU fallback)                   */
    EMS_EHIP = -3,       /* a HIP runtime call or kernel launch failed (see ems_last_error)          */
    EMS_EIO = -4,        /* matrix file unreadable / malformed                                       */
    EMS_ENOMEM = -5,
    EMS_EUNSUPPORTED = -6 /* field order, check degree or nm outside what the kernels are built for */
};

/* ---- parity-check matrix, flat CSR by check node (what init.c:143-272 LoadCode parses) ---- */
typedef struct {
    int N, M, GF, logGF;   /* code_t.N/.M/.GF/.logGF (struct.h:48-64)                               */
    int nbBranch;          /* number of edges E                                                     */
    const int *rowDegree;  /* [M]                                                                   */
    const int *col;        /* [E] variable-node index of each edge, check-node major (code_t.mat)   */
    const int *coef;       /* [E] GF coefficient as symbol index 1..GF-1 (code_t.matValue)          */
} ems_code_desc;

/* ---- decoder parameters: the reference's argv (NB_LDPC.c:105-111) ---- */
typedef struct {
    int nm;          /* decoder.nbMax: length of the truncated messages                            */
    int nbOper;      /* NbOper: max bubble-check steps per elementary check node                    */
    float offset;    /* offset added to the default C->V value (bubble_decoder.c:264)               */
    int NbIterMax;   /* reference semantics: NbIterMax-1 sweeps are run (NB_LDPC.c:314)             */
    int early_exit;  /* 1 = stop a frame at the first zero syndrome (NB_LDPC.c:470); 0 = fixed work */
} ems_params;

typedef struct ems_handle ems_handle;

/* Loader that accepts BOTH on-disk layouts the reference needs a recompile for (init.c:25):
 * fmt 0 = auto-detect, 1 = "UBS" block layout (init.c:195-207), 2 = "KN" interleaved (init.c:211-227), 3 = the full
 * non-binary alist (dvmax dcmax / degrees / per-column (row+1, exponent) / per-row (col+1, exponent) pairs) of
 * matrices/KN/N64800_*.txt, which neither loader of the reference reads.
 * Arrays are malloc()ed into *desc; release with ems_free_code_desc. */
int ems_load_code_file(const char *path, int fmt, ems_code_desc *desc);
void ems_free_code_desc(ems_code_desc *desc);

/* GF(q) tables as the reference builds them (init.c:37-130, struct.h:117-476); q in {16,64,256}.
 * idx2bin[i] = binary image of symbol index i (bit l = BINGF[i][l]); bin2idx = inverse. */
int ems_gf_tables(int q, uint8_t *idx2bin /*[q]*/, uint8_t *bin2idx /*[q]*/);

/* Create a decoder for one code on HIP device `device`, sized for up to max_batch frames in flight.
 * What the kernels are built for, next to what the reference allows (it caps none of these; EMS_EUNSUPPORTED / EMS_EINVAL beyond):
 *   - field order GF(16), GF(64), GF(256): the reference's own list (init.c:431-435);
 *   - check degree 2..64, regular or irregular (reference: scratch sized by rowDegree[0], no cap, init.c:320; its matrices go up to 20);
 *     degrees above 16 walk the edges of a visit in passes (k_decode_wide), LDS permitting: (4 dc - 6) lists of nm + 2 entries per check node;
 *   - 5 <= nm <= 64 (EMS_MAX_NM; reference: allocates by nbMax with no cap, init.c:310-328; start.sh runs 30); nm <= GF;
 *     above 32 the literal selection and bubble loop run;
 *   - a check node may list one variable node twice (none of the reference's matrices does; nothing in it forbids it): both edges read the
 *     same APP row (NB_LDPC.c:329-337), the later edge's APP update and decision stay (:444-450); such codes run the generic kernels;
 *   - the v2 flavour (ems_v2_*): GF(64) / GF(256), variable degree 2 on two different check nodes (v2.c:2157 handles nothing else); check
 *     degree 2..64, regular or irregular as above (v2.c:1756-1852 loops over rowDegree[node], :750 / :794-800 size the scratch by
 *     rowDegree[0]): regular degrees up to 16 (GF(256): up to 8) on lane rows, everything else in passes (k_decode_v2_wide, round 5). */
int ems_create(ems_handle **out, const ems_code_desc *code, const ems_params *prm, int max_batch, int device);
void ems_destroy(ems_handle *h);
int ems_set_params(ems_handle *h, const ems_params *prm); /* nm may not exceed the value given at create */

/* ---- the hot path, batched.  B <= max_batch frames.  Host-pointer variants copy in/out. ----
 * Every handle owns one HIP stream.  ems_llr_init*, ems_source_draw and ems_decode only queue work on it and
 * return (keep a host buffer handed to ems_llr_init untouched until the next waiting call on the handle: with
 * pinned memory the copy is asynchronous too); ems_get_results,
 * ems_count_errors*, the ems_get_* inspectors and ems_synchronize wait for that stream only.  Several handles
 * therefore run concurrently on one device -- the way the harness overlaps the tail of an early-exit batch with
 * the next batches.  A handle is not thread-safe; use one host thread per handle. */

/* channel.c:66-76 (GF(q) LLR of every symbol of every variable node) -> APP (NB_LDPC.c:281-288).
 * noisy: [B][N][logGF] float BPSK+AWGN observations (host-generated so the drand48 stream is kept). */
int ems_llr_init(ems_handle *h, const float *noisy_host, float sigma, int B);
int ems_llr_init_dev(ems_handle *h, const float *noisy_dev, float sigma, int B);
/* The constellation front ends of channel.c:112-929 -- LLR part: one GF(q) symbol = one point of a
 * D-dimensional constellation (D = 2: 64-APSK, rotated 16/64/256-QAM and ModelChannel's other compile-time constellations; D = 4: the
 * 4-D 256-QAM), q = 16, 64 or 256.
 *   obs   [B][N][D]  received coordinates (the reference's NoisyBin, channel.c:255-266 / :636-670 / :829-856)
 *   att   [B][N][D]  per-axis attenuation as the LLR sees it (fading, 0 = erased axis); may be NULL with EMS_LLR_SUM
 *                    (no fading); with it, EMS_LLR_SUM gives 0 to a node whose attenuation is <= 0 (channel.c:681-684)
 *   table [q][D]     the normalised constellation, indexed by the symbol's BINARY IMAGE (the reference's `som`)
 *   mode  EMS_LLR_SUM: SQR(y0-t0)/(2 sigma^2) + SQR(y1-t1)/(2 sigma^2), both quotients double, one rounding (channel.c:277)
 *         EMS_LLR_SSD: axes with attenuation > 0 accumulate SQR(y-a*t)/(2 sigma^2) into a float, others are skipped
 *                      (channel.c:690-700, :870-886) -- a node with every axis erased gets an all-zero row
 * The draws (RNG, modulation, fading, erasures) stay on the harness side, as for ems_llr_init. */
enum { EMS_LLR_SUM = 0, EMS_LLR_SSD = 1, EMS_LLR_NEG = 2 /* the v2 flavour's max-is-best form, see ems_v2_channel_likelihood */ };
int ems_llr_init_constellation(ems_handle *h, const float *obs_host, const float *att_host, const float *table_host, int D, int mode,
                               float sigma, int B);
/* Alternative input for tests: load APP directly from [B][N][GF] LLRs indexed by symbol. */
int ems_load_llr(ems_handle *h, const float *llr_host, int B);

/* NB_LDPC.c:266-472 for B frames: all sweeps, per-sweep Decision + Syndrom, per-frame early exit. */
int ems_decode(ems_handle *h, int B);

/* Results of the last ems_decode: decide [B][N] (tools.c:312), iters [B] = iter+1 (NB_LDPC.c:474),
 * synd [B] = value Syndrom() returns (tools.c:284).  Any pointer may be NULL.
 * Returns EMS_EHIP (and the outputs must not be used) if the kernel flagged a frame whose dependency wait gave up
 * -- the device marks such a frame with iters = -1; it cannot happen with a schedule built by ems_create. */
int ems_get_results(ems_handle *h, int B, int *decide, int *iters, int *synd);

/* llr_init + decode + get_results in one call (what Decoding() and the harness use). */
int ems_decode_batch(ems_handle *h, const float *noisy_host, float sigma, int B,
                     int *decide, int *iters, int *synd);

/* NB_LDPC.c:479-485 bit errors over the first K symbols against the transmitted codewords [B][N]. */
int ems_count_errors(ems_handle *h, const int *codewords_host, int B, int *nb_errors /*[B]*/);

/* Device-resident variant for benchmark drivers: upload the codewords once, then count per step;
 * nb_errors may be NULL (result stays on the device). */
int ems_set_codewords(ems_handle *h, const int *codewords_host, int B);
int ems_count_errors_resident(ems_handle *h, int B, int *nb_errors /*[B] or NULL*/);

/* ---- device-side frame source ----
 * Replaces, for throughput runs, the host calls RandomBinaryGenerator (tools.c:124) + Encoding (tools.c:232) +
 * the noise draw of ModelChannel_AWGN_BPSK (channel.c:52-62): information symbols, codewords and noisy BPSK
 * samples of frames [first_frame, first_frame + B) are produced in HBM, the codewords stay resident for
 * ems_count_errors_resident and the channel LLRs are initialised, i.e. the handle is left exactly as after
 * ems_set_codewords + ems_llr_init.  The encoder is the reference's (bit-exact); the random stream is a
 * counter-based Philox4x32-10 keyed by `seed` and indexed by (frame, draw) -- frame f does not depend on the
 * batch size or on which rank draws it -- NOT libc's drand48: statistical, not bit, parity with the reference's
 * frames (ems_draw_frames keeps the bit-exact stream on the host).
 * ems_source_attach takes GaussianElimination's outputs: matUT as one dense row-major [M][N] block (code_t.matUT[0])
 * and Perm [N]. */
int ems_source_attach(ems_handle *h, const int *matUT, const int *perm);
int ems_source_draw(ems_handle *h, unsigned long long seed, unsigned long long first_frame, int B, float sigma);
/* Round 5: the REFERENCE's own stream on the device -- frames [first_frame, first_frame + B) (0-based) of the drand48 sequence the
 * reference's main() consumes (X0 = 0; per frame K logGF draws of RandomBinaryGenerator, tools.c:124-136, then 2 N logGF of the noise draw,
 * channel.c:52-62), bit for bit what ems_draw_frames generates on the host cores: the LCG in exact 48-bit integer arithmetic with
 * jump-ahead per frame and per thread, the encoder of ems_source_attach, the Box-Muller sample in double on the device -- and, because the
 * device's log / cos are not the host's to the last bit, every sample whose rounding to float they could move (about one in 60 000, mostly samples near 0 where floats are dense) recomputed
 * on the host with the reference's expression before anything reads it (*recomputed, may be NULL, receives how many).  Leaves the handle
 * as ems_set_codewords + ems_llr_init would.  Unlike ems_source_draw it waits for its own kernels (the recomputed samples are patched in
 * before the LLR kernel is queued). */
int ems_source_draw_reference(ems_handle *h, unsigned long long first_frame, int B, float sigma, int *recomputed);
/* copies of what ems_source_draw / ems_source_draw_reference produced (tests): codewords [B][N] int, noisy [B][N][logGF] float; either may be NULL */
int ems_source_get(ems_handle *h, int B, int *codewords, float *noisy);

/* ---- introspection for parity tests ---- */
int ems_get_app(ems_handle *h, int frame, float *app /*[N][GF]*/);   /* decoder_t.APP after decode      */
int ems_get_ctov(ems_handle *h, int frame, float *ctov /*[E][GF]*/); /* decoder_t.CtoV, expanded dense */
/* channel.c:78-91: the ascending (LLR, symbol) form of frame `frame`'s channel LLRs, as
 * decoder_t.intrinsic_LLR / intrinsic_GF hold it ([N][GF] each); valid after ems_llr_init. */
int ems_get_sorted_llr(ems_handle *h, int frame, float *s_llr, int *s_gf);
/* ---- v2 flavour: the decode core of EMS_HS_L-BubbleCheck_UBS_decoder_v2.c ----
 * Max-is-best log-likelihoods, nm-long messages both ways, 4-candidate bubble check (v2.c:1547), horizontal
 * schedule (v2.c:2119-2253).  Same handle (ems_create: nm, NbOper, offset, NbIterMax, early_exit); the decode
 * runs iterations iter = 0..NbIterMax inclusive like the reference's loop.  Needs dv = 2 (the reference's Flag
 * trick, v2.c:2157), GF(64) or GF(256); any check degree 2..64, regular or irregular (the reference's dc = 20 matrix included).
 *   like / like_idx: [B][N][GF], per node the GF likelihoods in decreasing order and their symbols, i.e.
 *   decoder.likelihood / decoder.likelihood_index as ModelChannelAWGN leaves them (v2.c:1483-1497).
 *   Results through ems_get_results (decide, iterations executed, syndrome). */
int ems_v2_load_likelihood(ems_handle *h, const float *like, const int *like_idx, int B);
int ems_v2_decode(ems_handle *h, int B);
/* The older generation's own channel front end, LLR part (ModelChannelAWGN, v2.c:1472-1497) instead of
 * ems_v2_load_likelihood: obs [B][N][2] received coordinates, att [B][N][2] the node's fading on both axes (ATT[n]),
 * table [GF][2] the scaled constellation by binary image (TABLEAU), sigma (v2.c:1451).  The likelihoods
 * -(SQR(y0-t0*A)/(2 sigma^2)) - SQR(y1-A*t1)/(2 sigma^2) (0 where A <= 0) go straight into the decoder's symbol-indexed block
 * and the V->C messages start from each node's nm best (v2.c:2101-2113); ems_v2_get_sorted_likelihood returns what the
 * reference keeps as decoder.likelihood / likelihood_index ([N][GF] each, decreasing) for one frame. */
int ems_v2_channel_likelihood(ems_handle *h, const float *obs_host, const float *att_host, const float *table_host, float sigma, int B);
int ems_v2_get_sorted_likelihood(ems_handle *h, int frame, float *like, int *like_idx);
/* The statistics the older generation's main() keeps (v2.c:2174-2291): with snapshots enabled, ems_v2_decode also counts, per frame,
 * the bit errors (first K symbols, against the codewords of ems_set_codewords) of the hard decisions in force when iterations 9, 19,
 * 29, 39 and 49 BEGIN -- v2.c samples before that iteration's ComputeAPPLog / Decision, so these are the decisions of iterations
 * 8, 18, ... -- and 0 where the frame had stopped (or NbIterMax ended) before.  snap: [B][5].  The "> 12 bit errors" frame rule and
 * the printed lines stay in the harness (csrc/harness/ems_sim_v2.c). */
int ems_v2_enable_snapshots(ems_handle *h, int on);
int ems_v2_get_snapshots(ems_handle *h, int B, int *snap /*[B][5]*/);
/* initialisation again from the likelihoods already staged in HBM by the last ems_v2_load_likelihood (timed with
 * ems_last_llr_ms): lets a harness repeat a decode without the host upload */
int ems_v2_reinit(ems_handle *h, int B);
/* decoder.VtoC / VtoCindex / CtoV / CtoVindex ([E][nm], edges numbered row by row) and decoder.APP ([N][GF]) of one
 * frame after ems_v2_decode; any pointer may be NULL. */
int ems_v2_get_state(ems_handle *h, int frame, float *vtoc, int *vtoci, float *ctov, int *ctovi, float *app);

/* Level schedule of the layered sweep: check nodes grouped so that no two in a level share a variable
 * node; visiting levels in order reproduces the sequential order of NB_LDPC.c:320 exactly. */
int ems_get_schedule(ems_handle *h, int *n_levels, int *level_ptr /*[M+1] or NULL*/, int *order /*[M] or NULL*/);
/* The same schedule computed on the host only (no device needed). */
int ems_level_schedule(const ems_code_desc *code, int *n_levels, int *level_ptr /*[M+1] or NULL*/, int *order /*[M] or NULL*/);

/* The pairs of check nodes the two-visits-per-wave kernels claim, in claim order, computed on the host only (check degree 4
 * or 8): pair_cn [2 * n_pairs] (or NULL), second member -1 for a single.  Members of a pair share no variable node and
 * every earlier visitor (reference order, NB_LDPC.c:320) of their variable nodes sits in an earlier pair, so walking the
 * pairs in order -- both members at once -- is the sequential sweep exactly. */
int ems_pair_schedule(const ems_code_desc *code, int *n_pairs, int *pair_cn /*[2*M] or NULL*/);

/* ---- single-node entry points on the device (unit parity of bubble_decoder.c) ---- */
/* bubble_decoder.c:316 ElementaryStep: index-domain symbols, -1 = invalid, 1e5 = padding value. */
int ems_elementary_step(int GF, int nm, int nbOper, const float *in1, const float *in2, const int *idx1,
                        const int *idx2, float *out, int *idxout, int device);
/* bubble_decoder.c:72 CheckPassLogEMS: vtoc_* [dc][nm] sorted+normalised inputs (unrotated symbols),
 * coef [dc] symbol indices, ctov [dc][GF] dense output indexed by symbol. */
int ems_check_node(int GF, int dc, int nm, int nbOper, float offset, const int *coef, const float *vtoc_llr,
                   const int *vtoc_gf, float *ctov, int device);
/* tools.c:312 / tools.c:284 on the device for one frame */
int ems_decision(int GF, int N, const float *app /*[N][GF]*/, int *decide, int device);
int ems_syndrome(const ems_code_desc *code, const int *decide, int *synd, int device);

/* ---- launch shape ----
 * The library picks the decode kernel and its workgroup shape from the code, the message size and the batch
 * (DESIGN.md 4.1).  A harness that wants another one says so HERE, once, not through the environment of every call:
 * each field is -1 for "the library's choice".  ems_create initialises the handle's tuning from the developer variables
 * EMS_DUAL / EMS_WAVES / EMS_OCC4 / EMS_TWIN (read once, at create); the hot entry points never call getenv.
 *   dual   0 = never the two-visits-per-wave kernels (nor two frames per wave); n > 0 = force them, n waves per workgroup
 *   waves  n > 0 = force the one-visit kernel with n waves per workgroup (occ4 != 0: its 128-VGPR build)
 *   twin   0 = never two FRAMES per wave (tiny codes); 1 = force it */
typedef struct { int dual, waves, occ4, twin; } ems_tuning;
int ems_set_tuning(ems_handle *h, const ems_tuning *t);
int ems_get_tuning(const ems_handle *h, ems_tuning *t);
/* Frames per launch that fill device `device` for this code and message size with the kernel the library would pick
 * (n_CU x the tuned frames per CU of that kernel), capped so that `handles` decoder handles of that size fit the
 * free HBM.  What ems_sim uses when EMS_BATCH is not given. */
int ems_suggest_batch(const ems_code_desc *code, const ems_params *prm, int device, int handles, int *batch);

/* ---- measurement ---- */
/* Duration in ms of the kernels of the last ems_llr_init / ems_decode call, from HIP events recorded
 * on the handle's own stream around the launches. */
float ems_last_llr_ms(const ems_handle *h);
float ems_last_decode_ms(const ems_handle *h);
/* The decode kernel the last ems_decode / ems_v2_decode launched, as the profiler names it without its argument list
 * (e.g. "k_decode_dual<256, 20, 4>"), its workgroup size in waves, and the device's compute-unit count: what a bench
 * reports instead of restating the library's shape rule.  "" / 0 before the first decode. */
const char *ems_last_kernel_name(const ems_handle *h);
int ems_last_workgroup_waves(const ems_handle *h);
int ems_device_cu_count(const ems_handle *h);
/* Raw device access for zero-copy drivers (bench.py): device pointer of the noisy-sample staging
 * buffer [max_batch][N][logGF] float and the handle's stream (hipStream_t as void*). */
void *ems_noisy_dev_ptr(ems_handle *h);
void *ems_stream(ems_handle *h);
int ems_synchronize(ems_handle *h);
/* Developer diagnostics of builds compiled with -DEMS_STAMPS (all zero in the product build): out24 receives
 * 24 values -- [0..15] per-phase cycle accumulators, [16..23] event counters. */
int ems_debug_stamps(ems_handle *h, unsigned long long *out24 /*[24]*/, int reset);
const char *ems_last_error(void);
const char *ems_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EMS_ABI_H */
