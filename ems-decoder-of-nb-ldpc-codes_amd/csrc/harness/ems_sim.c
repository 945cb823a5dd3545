/*
 * ems_sim.c -- Monte-Carlo BER/FER harness in C driving the MI355X decoder library.
 *
 * Same seven positional arguments, same drand48 draw order (source -> encoder -> channel, frame after
 * frame, never reseeded), same progress line, same 40-erroneous-frame stop and same result-file line as
 * the reference's main() (NB_LDPC.c:63-604).  The only structural change is the one north_star asks
 * for: the inline per-frame decode block (NB_LDPC.c:266-472) is replaced by one batched call into the
 * GPU library -- frames are generated EMS_BATCH at a time (default: what fills the device, ems_suggest_batch) by EMS_THREADS host
 * threads (default: the cores, at most 16; 1 = the reference's own serial call sequence, same bits), decoded
 * together, and reported in frame order, truncated at the frame where the 40th erroneous frame occurs.
 * Several decoder handles (EMS_INFLIGHT, default 3) take turns so that the tail of one batch overlaps the next.
 * Frame source (round 5): by default the reference's drand48 frames are generated ON THE DEVICE, bit for bit (ems_source_draw_reference: exact
 * 48-bit LCG with jump-ahead, the reference's encoder, Box-Muller in double with the undecidable float roundings recomputed on the host);
 * EMS_HOST_SOURCE=1 keeps the host generator described above (same bits); EMS_GPU_SOURCE=<seed> switches to the device-side counter-based
 * generator (ems_source_draw): statistical instead of bit parity with the reference's frames.
 *
 * EMS_DEVICES=0,1,...,7 runs ONE simulation on several GPUs of the node (run_multi below, ems_multi.h): one host thread per
 * device, batch j of the stream on device j mod D, and per round one RCCL all-reduce (ncclAllReduce, ncclInt64, ncclSum) of the
 * five counters {frames, erroneous frames, bit errors, undetected, sum of iterations} over xGMI -- the only collective of the
 * path -- on which every device takes the same "40 erroneous frames" stop; the per-frame results stay on the host and device 0's
 * thread prints the reference's progress line in frame order, cut at exactly the 40th erroneous frame.  EMS_DEVICES=0 runs the
 * same code, collective included, on one GPU.
 *
 *   ems_sim NbMonteCarlo NbIterMax FileMatrix EbN NbMax Offset NbOper
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <dlfcn.h>
#include <hip/hip_runtime_api.h>            /* (plain C: device selection, the 40-byte staging copies and the stream of the all-reduce) */

#include "ems_abi.h"
#include "ems_multi.h"
#include "nbldpc_ref_api.h"

static void die(const char *what, int rc)
{
    fprintf(stderr, "\n%s failed (%d): %s\n", what, rc, ems_last_error());
    exit(EXIT_FAILURE);
}

/* One batch of frames, drawn on the host (the reference's stream, NB_LDPC.c:255-261) while the GPU decodes the
 * previous one. */
typedef struct {
    code_t *code; table_t *table;
    long first;                             /* frames [first, first+B) of the stream */
    int B, threads;
    float EbN, sigma;
    int *cw; float *noisy;                  /* [B][N], [B][N][logGF] */
    int **NBIN, **KBIN, *KSYMB, *Idum;      /* scratch of the serial (threads == 1) path */
    double seconds;
    int rc;
    int offset_run;                         /* EMS_FIRST_FRAME != 0: libc's drand48 cannot be positioned, use the library's generator */
} gen_job;

static void *generate(void *arg)
{
    gen_job *j = (gen_job *)arg;
    struct timespec g0, g1;
    clock_gettime(CLOCK_MONOTONIC, &g0);
    j->rc = 0;
    if (j->threads > 1 || j->offset_run) {  /* same bits, frame-parallel, any starting frame (nbldpc_ref_api.h) */
        j->rc = ems_draw_frames(j->code, j->table, j->first, j->B, j->EbN, j->threads, j->cw, j->noisy, &j->sigma);
    } else {
        const int N = j->code->N, logGF = j->code->logGF;
        for (int b = 0; b < j->B; b++) {    /* the reference's own call sequence on libc's drand48 */
            RandomBinaryGenerator(N, j->code->M, j->code->GF, logGF, j->KBIN, j->KSYMB, j->table->BINGF, j->Idum);
            Encoding(j->code, j->table, j->cw + (size_t)b * N, j->NBIN, j->KSYMB);
            j->sigma = ems_awgn_bpsk_noise(j->code, j->NBIN, j->EbN, j->Idum, j->noisy + (size_t)b * N * logGF);
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &g1);
    j->seconds = (g1.tv_sec - g0.tv_sec) + 1e-9 * (g1.tv_nsec - g0.tv_nsec);
    return NULL;
}

/* the end of the reference's main() (NB_LDPC.c:563-583): result-file line and closing messages */
static void report(const char *file_name, float EbN, int nbErroneousFrames, int nb_logged, int total_errors, long long sum_it, int K, int logGF)
{
    printf(" \n results are printed in file %s \n", file_name);
    time_t now = time(NULL);
    char *ts = ctime(&now);
    FILE *op = fopen(file_name, "a");
    if (!op) {
        printf(" \n !! file not found \n ");
    } else {
        fprintf(op, " SNR:%.2f: \t FER= %d / %d = %f ", EbN, nbErroneousFrames, nb_logged, (double)nbErroneousFrames / nb_logged);
        fprintf(op, " \t BER= %d / x = \t %f  avr_it= \t %.2f \t time: %s", total_errors,
                (double)total_errors / ((double)nb_logged * K * logGF), (double)sum_it / nb_logged, ts);
        fclose(op);
    }
    printf(" \n results printed \n ");
    printf("\n");
    printf("Simulation complete at time: %s", ts);
}

/* ------------------------------------------------------------------------------------------------------------------------
 * several GPUs, one simulation (EMS_DEVICES)
 * ---------------------------------------------------------------------------------------------------------------------- */
/* RCCL is a 570 MB library: it is loaded only when EMS_DEVICES asks for it.  Prototypes as in rccl.h:236 / :260 / :339 / :611. */
typedef struct {
    void *lib;
    int (*CommInitAll)(void **comm, int ndev, const int *devlist);
    int (*AllReduce)(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, hipStream_t stream);
    int (*CommDestroy)(void *comm);
    const char *(*GetErrorString)(int result);
} rccl_api;
enum { RCCL_INT64 = 4 /* ncclInt64, rccl.h:463 */, RCCL_SUM = 0 /* ncclSum, rccl.h:448 */ };

static int rccl_load(rccl_api *r)
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", NULL};
    r->lib = NULL;
    for (int i = 0; names[i] && !r->lib; i++) r->lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r->lib) { fprintf(stderr, "cannot load RCCL: %s\n", dlerror()); return -1; }
    *(void **)&r->CommInitAll = dlsym(r->lib, "ncclCommInitAll");
    *(void **)&r->AllReduce = dlsym(r->lib, "ncclAllReduce");
    *(void **)&r->CommDestroy = dlsym(r->lib, "ncclCommDestroy");
    *(void **)&r->GetErrorString = dlsym(r->lib, "ncclGetErrorString");
    if (!r->CommInitAll || !r->AllReduce || !r->CommDestroy || !r->GetErrorString) { fprintf(stderr, "RCCL: symbols missing\n"); return -1; }
    return 0;
}

typedef struct {
    int hip_dev;                            /* device ordinal */
    ems_handle *h[2];                       /* two batches in flight per device (ems_multi.h) */
    int *cw[2];
    float *noisy[2];
    long long first[2];
    int *iters, *synd, *nberr;
    void *comm;                             /* ncclComm_t */
    hipStream_t cstream;                    /* the all-reduce's own stream */
    long long *d_cnt;                       /* [2][5] send | receive */
    long long *h_cnt;                       /* [2][5] pinned host staging of the same (an asynchronous copy to pageable memory would wait) */
} dev_ctx;

typedef struct {
    dev_ctx *d;
    int ndev, threads_per_dev, gpu_source, K, logGF;
    code_t *code;
    table_t *table;
    float EbN, sigma_dev;
    unsigned long long gpu_seed;
    long long first_frame;
    double allreduce_timeout_s;             /* watchdog of the per-round all-reduce (EMS_ALLREDUCE_TIMEOUT) */
    rccl_api nccl;
    int host_sum;                           /* EMS_ALLREDUCE=host: the five counters summed in host memory under a barrier (no RCCL) */
    pthread_barrier_t hbar;
    long long (*hslot)[EMS_NCOUNTERS];      /* [ndev] */
} multi_ctx;

static int m_launch(void *c, int dev, int slot, long long first, int count)
{
    multi_ctx *m = (multi_ctx *)c;
    dev_ctx *d = &m->d[dev];
    int rc;
    d->first[slot] = first;
    if (m->gpu_source == 2) {
        if ((rc = ems_source_draw_reference(d->h[slot], (unsigned long long)(m->first_frame + first), count, m->sigma_dev, NULL))) goto bad;
    } else if (m->gpu_source) {
        if ((rc = ems_source_draw(d->h[slot], m->gpu_seed, (unsigned long long)(m->first_frame + first), count, m->sigma_dev))) goto bad;
    } else {
        float sigma = 0.0f;                 /* the reference's stream from any frame on (LCG jump-ahead), frame-parallel on this device's share of the cores */
        if (ems_draw_frames(m->code, m->table, (long)(m->first_frame + first), count, m->EbN, m->threads_per_dev, d->cw[slot], d->noisy[slot], &sigma)) {
            fprintf(stderr, "\nems_draw_frames failed\n");
            return -1;
        }
        if ((rc = ems_llr_init(d->h[slot], d->noisy[slot], sigma, count))) goto bad;
    }
    if ((rc = ems_decode(d->h[slot], count))) goto bad;
    return 0;
bad:
    fprintf(stderr, "\ndevice %d: launch failed (%d): %s\n", d->hip_dev, rc, ems_last_error());
    return -1;
}

static int m_collect(void *c, int dev, int slot, int count, ems_frame_result *res)
{
    multi_ctx *m = (multi_ctx *)c;
    dev_ctx *d = &m->d[dev];
    int rc;
    if (!res) return ems_synchronize(d->h[slot]) ? -1 : 0;           /* launched ahead of the stop: dropped */
    if ((rc = ems_get_results(d->h[slot], count, NULL, d->iters, d->synd)) ||
        (rc = m->gpu_source ? ems_count_errors_resident(d->h[slot], count, d->nberr) : ems_count_errors(d->h[slot], d->cw[slot], count, d->nberr))) {
        fprintf(stderr, "\ndevice %d: collect failed (%d): %s\n", d->hip_dev, rc, ems_last_error());
        return -1;
    }
    for (int b = 0; b < count; b++) {
        res[b].nberr = d->nberr[b]; res[b].synd = d->synd[b]; res[b].iters = d->iters[b];
    }
    return 0;
}

/* the path's one collective: ncclAllReduce(5 x int64, sum) on the device's own communicator and stream.
 * A device that cannot join the collective cannot be recovered in-process -- its peers are already inside it and would wait for ever --
 * so every failure here, and an all-reduce that does not complete within EMS_ALLREDUCE_TIMEOUT seconds (default 120; the counters are
 * 40 bytes), ends the whole process with a message naming the device instead of hanging the node. */
static void m_allreduce_fatal(const dev_ctx *d, const char *what, const char *detail)
{
    fprintf(stderr, "\nems_sim: device %d: %s%s%s -- the per-round counter all-reduce cannot complete, stopping\n", d->hip_dev, what,
            detail ? ": " : "", detail ? detail : "");
    fflush(stderr);
    _exit(EXIT_FAILURE);                  /* (not exit(): atexit handlers of RCCL / HIP could wait for the stuck collective) */
}

/* Issued right after a round is collected, waited for one round later (ems_multi.h: the collective's kernel needs a CU, and the next
 * round's persistent decode workgroups may hold all of them until that round ends). */
static int m_allreduce_begin(void *c, int dev, const long long v[EMS_NCOUNTERS])
{
    multi_ctx *m = (multi_ctx *)c;
    dev_ctx *d = &m->d[dev];
    int nrc;
    hipError_t he;
    if ((he = hipSetDevice(d->hip_dev)) != hipSuccess) m_allreduce_fatal(d, "hipSetDevice", hipGetErrorString(he));
    memcpy(d->h_cnt, v, sizeof(long long) * EMS_NCOUNTERS);
    if ((he = hipMemcpyAsync(d->d_cnt, d->h_cnt, sizeof(long long) * EMS_NCOUNTERS, hipMemcpyHostToDevice, d->cstream)) != hipSuccess)
        m_allreduce_fatal(d, "staging copy", hipGetErrorString(he));
    if ((nrc = m->nccl.AllReduce(d->d_cnt, d->d_cnt + EMS_NCOUNTERS, EMS_NCOUNTERS, RCCL_INT64, RCCL_SUM, d->comm, d->cstream)))
        m_allreduce_fatal(d, "ncclAllReduce", m->nccl.GetErrorString(nrc));
    if ((he = hipMemcpyAsync(d->h_cnt + EMS_NCOUNTERS, d->d_cnt + EMS_NCOUNTERS, sizeof(long long) * EMS_NCOUNTERS, hipMemcpyDeviceToHost, d->cstream)) != hipSuccess)
        m_allreduce_fatal(d, "result copy", hipGetErrorString(he));
    return 0;
}

static int m_allreduce_end(void *c, int dev, long long v[EMS_NCOUNTERS])
{
    multi_ctx *m = (multi_ctx *)c;
    dev_ctx *d = &m->d[dev];
    hipError_t he;
    if ((he = hipSetDevice(d->hip_dev)) != hipSuccess) m_allreduce_fatal(d, "hipSetDevice", hipGetErrorString(he));
    /* watchdog: poll instead of hipStreamSynchronize, which has no deadline */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; spins++) {
        he = hipStreamQuery(d->cstream);
        if (he == hipSuccess) break;
        if (he != hipErrorNotReady) m_allreduce_fatal(d, "all-reduce stream", hipGetErrorString(he));
        if (spins > 200) { const struct timespec nap = {0, 200000}; nanosleep(&nap, NULL); }     /* 0.2 ms once the fast path is over */
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > m->allreduce_timeout_s) {
            char msg[96];
            snprintf(msg, sizeof msg, "no completion after %.0f s (EMS_ALLREDUCE_TIMEOUT)", m->allreduce_timeout_s);
            m_allreduce_fatal(d, "all-reduce watchdog", msg);
        }
    }
    memcpy(v, d->h_cnt + EMS_NCOUNTERS, sizeof(long long) * EMS_NCOUNTERS);
    return 0;
}

/* EMS_ALLREDUCE=host: the device threads of this one process sum the five counters in host memory (synchronous form of ems_multi.h) */
static int m_allreduce_host(void *c, int dev, long long v[EMS_NCOUNTERS])
{
    multi_ctx *m = (multi_ctx *)c;
    memcpy(m->hslot[dev], v, sizeof m->hslot[dev]);
    pthread_barrier_wait(&m->hbar);
    for (int i = 0; i < EMS_NCOUNTERS; i++) {
        long long sum = 0;
        for (int k = 0; k < m->ndev; k++) sum += m->hslot[k][i];
        v[i] = sum;
    }
    pthread_barrier_wait(&m->hbar);                                  /* (the slots are free for the next round) */
    return 0;
}

static void m_line(const ems_counters *run, void *c)                   /* NB_LDPC.c:498 */
{
    multi_ctx *m = (multi_ctx *)c;
    const long long nb = run->c[EMS_CNT_FRAMES];
    printf("\r<%d> FER= %d / %d = %f BER= %d / x = %f  avr_it=%.2f", (int)run->c[EMS_CNT_UNDETECTED], (int)run->c[EMS_CNT_ERR_FRAMES], (int)nb,
           (double)run->c[EMS_CNT_ERR_FRAMES] / nb, (int)run->c[EMS_CNT_BIT_ERRORS],
           (double)run->c[EMS_CNT_BIT_ERRORS] / ((double)nb * m->K * m->logGF), (double)run->c[EMS_CNT_SUM_IT] / nb);
    fflush(stdout);
}

/* The all-reduce's stream gets the highest priority the device offers: its one small kernel is queued while the next round's decode
 * workgroups (one per frame, persistent) hold every CU, and should take the first slot a finished frame frees rather than queue behind
 * the decode's own backlog. */
static hipError_t create_collective_stream(hipStream_t *s)
{
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

/* EMS_DEVICES: the whole simulation loop on the listed devices; fills the reference's counters.  Returns 0 or exits. */
static int run_multi(const char *devlist, code_t *code, table_t *table, const ems_code_desc *desc, const ems_params *prm, int batch_env,
                     long long n_frames, float EbN, int threads, int gpu_source, unsigned long long gpu_seed, long long first_frame,
                     ems_counters *fin, int *batch_out)
{
    enum { MAXDEV = 64 };
    int devs[MAXDEV], nd = 0;
    for (const char *p = devlist; *p && nd < MAXDEV;) {
        char *e;
        const long v = strtol(p, &e, 10);
        if (e == p) break;
        devs[nd++] = (int)v;
        p = (*e == ',') ? e + 1 : e;
    }
    if (nd < 1) { fprintf(stderr, "EMS_DEVICES=%s: no device\n", devlist); exit(EXIT_FAILURE); }
    multi_ctx m;
    memset(&m, 0, sizeof m);
    m.host_sum = getenv("EMS_ALLREDUCE") && strcmp(getenv("EMS_ALLREDUCE"), "host") == 0;
    if (!m.host_sum && rccl_load(&m.nccl)) exit(EXIT_FAILURE);
    m.ndev = nd; m.code = code; m.table = table; m.EbN = EbN; m.gpu_source = gpu_source; m.gpu_seed = gpu_seed; m.first_frame = first_frame;
    m.K = code->K; m.logGF = code->logGF;
    m.allreduce_timeout_s = getenv("EMS_ALLREDUCE_TIMEOUT") ? atof(getenv("EMS_ALLREDUCE_TIMEOUT")) : 120.0;
    if (!(m.allreduce_timeout_s > 0.0)) m.allreduce_timeout_s = 120.0;
    m.threads_per_dev = threads / nd > 0 ? threads / nd : 1;
    m.sigma_dev = (float)sqrt(1.0 / (2.0 * (double)code->rate * pow(10.0, (double)EbN / 10.0)));   /* channel.c:51 */
    int batch = batch_env, rc;
    if (batch <= 0 && (rc = ems_suggest_batch(desc, prm, devs[0], 2, &batch))) die("ems_suggest_batch", rc);
    if ((long long)batch * nd > n_frames) batch = (int)((n_frames + nd - 1) / nd);      /* (no buffers for frames that are never drawn) */
    if (batch < 1) batch = 1;
    *batch_out = batch;
    m.d = calloc((size_t)nd, sizeof *m.d);
    void **comms = calloc((size_t)nd, sizeof *comms);
    int nrc;
    if (m.host_sum) {
        m.hslot = calloc((size_t)nd, sizeof *m.hslot);
        if (!m.hslot || pthread_barrier_init(&m.hbar, NULL, (unsigned)nd)) { fprintf(stderr, "cannot set up the host-side sum\n"); exit(EXIT_FAILURE); }
    } else if ((nrc = m.nccl.CommInitAll(comms, nd, devs))) {           /* one communicator per device, this process */
        fprintf(stderr, "ncclCommInitAll(%d devices): %s\n", nd, m.nccl.GetErrorString(nrc));
        exit(EXIT_FAILURE);
    }
    for (int i = 0; i < nd; i++) {
        dev_ctx *d = &m.d[i];
        d->hip_dev = devs[i];
        d->comm = comms[i];
        for (int s = 0; s < 2; s++) {
            if ((rc = ems_create(&d->h[s], desc, prm, batch, devs[i]))) die("ems_create", rc);
            if (gpu_source && (rc = ems_source_attach(d->h[s], code->matUT[0], code->Perm))) die("ems_source_attach", rc);
            d->cw[s] = malloc(sizeof(int) * (size_t)batch * code->N);
            d->noisy[s] = malloc(sizeof(float) * (size_t)batch * code->N * code->logGF);
        }
        d->iters = malloc(sizeof(int) * batch); d->synd = malloc(sizeof(int) * batch); d->nberr = malloc(sizeof(int) * batch);
        if (hipSetDevice(devs[i]) != hipSuccess || create_collective_stream(&d->cstream) != hipSuccess ||
            hipMalloc((void **)&d->d_cnt, sizeof(long long) * 2 * EMS_NCOUNTERS) != hipSuccess ||
            hipHostMalloc((void **)&d->h_cnt, sizeof(long long) * 2 * EMS_NCOUNTERS, hipHostMallocDefault) != hipSuccess) {
            fprintf(stderr, "device %d: cannot set up the all-reduce buffers\n", devs[i]);
            exit(EXIT_FAILURE);
        }
    }
    fprintf(stderr, "[ems_sim] %d device(s) [%s], one %s of %d int64 counters per round, batch %d frames per device and round, %d generator thread(s) per device\n",
            nd, devlist, m.host_sum ? "host-side sum (EMS_ALLREDUCE=host)" : "RCCL all-reduce (issued when a round is collected, waited for a round later)",
            EMS_NCOUNTERS, batch, m.threads_per_dev);
    const ems_multi_ops ops = {m_launch, m_collect, m.host_sum ? m_allreduce_host : NULL, m_line, &m, m.host_sum ? NULL : m_allreduce_begin,
                               m.host_sum ? NULL : m_allreduce_end};
    if (ems_multi_run(&ops, nd, n_frames, batch, 40, fin)) { fprintf(stderr, "\nmulti-device run failed\n"); exit(EXIT_FAILURE); }
    for (int i = 0; i < nd; i++) {
        dev_ctx *d = &m.d[i];
        (void)hipSetDevice(d->hip_dev);
        (void)hipStreamSynchronize(d->cstream);
        if (!m.host_sum) (void)m.nccl.CommDestroy(d->comm);
        (void)hipFree(d->d_cnt);
        (void)hipHostFree(d->h_cnt);
        (void)hipStreamDestroy(d->cstream);
        for (int s = 0; s < 2; s++) { ems_destroy(d->h[s]); free(d->cw[s]); free(d->noisy[s]); }
        free(d->iters); free(d->synd); free(d->nberr);
    }
    if (m.host_sum) { pthread_barrier_destroy(&m.hbar); free(m.hslot); }
    free(m.d); free(comms);
    return 0;
}

int main(int argc, char *argv[])
{
    if (argc < 8) {
        printf("File:\n %s\n ", argv[0]);
        printf("\n\tSyntax:\n\t\n\t NbMonteCarlo     : # simulated frames"
               "\n\t NbIterMax        : # of maximum decoding iterations"
               "\n\t FileMatrix       : File name of the parity-check matrix"
               "\n\t EbN              : Eb/No (dB)"
               "\n\t NbMax            : size of truncated messages"
               "\n\t Offset           : offset correction factor (0.4 -- 1)"
               "\n\t NbOper           : Maximum number of operations for sorting \n");
        return EXIT_FAILURE;
    }
    /* one HIP stream per decoder handle: keep them on distinct hardware queues (ROCm's default is 4 queues for all
     * streams of the process; streams that share one serialise).  Must be set before the first HIP call. */
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    const int NbMonteCarlo = atoi(argv[1]);
    const int NbIterMax = atoi(argv[2]);
    char *FileMatrix = argv[3];
    const float EbN = (float)atof(argv[4]);
    const int nbMax = atoi(argv[5]);
    const float offset = (float)atof(argv[6]);
    const int NbOper = atoi(argv[7]);
    int batch = getenv("EMS_BATCH") ? atoi(getenv("EMS_BATCH")) : 0;   /* 0: derived from the device once the code is loaded (ems_suggest_batch) */
    const int device = getenv("EMS_DEVICE") ? atoi(getenv("EMS_DEVICE")) : 0;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int threads = getenv("EMS_THREADS") ? atoi(getenv("EMS_THREADS")) : (int)(ncpu > 16 ? 16 : (ncpu < 1 ? 1 : ncpu));
    if (threads < 1) threads = 1;
    /* EMS_GPU_SOURCE=<seed>: frames are generated on the GPU from the counter-based stream <seed> (ems_source_draw:
     * no host generation, no PCIe traffic) instead of the reference's drand48 stream -- statistical, not bit,
     * parity with the reference's trace. */
    /* frame source: 2 = the reference's drand48 stream generated on the DEVICE (default since round 5: ems_source_draw_reference, bit for bit
     * the frames of the host generator below, at a few per cent of its time); 0 = the same stream on the host cores (EMS_HOST_SOURCE=1:
     * ems_draw_frames, or the reference's serial calls with EMS_THREADS=1); 1 = the counter-based device stream (EMS_GPU_SOURCE=<seed>) */
    const int gpu_source = getenv("EMS_GPU_SOURCE") != NULL ? 1 : (getenv("EMS_HOST_SOURCE") && atoi(getenv("EMS_HOST_SOURCE")) ? 0 : 2);
    const unsigned long long gpu_seed = gpu_source == 1 ? strtoull(getenv("EMS_GPU_SOURCE"), NULL, 0) : 0ULL;
    /* EMS_FIRST_FRAME=<f>: this process simulates frames [f, f + NbMonteCarlo) of the stream (0-based), e.g. one process
     * per GPU on disjoint ranges; 0 = the reference's run. */
    const long first_frame = getenv("EMS_FIRST_FRAME") ? atol(getenv("EMS_FIRST_FRAME")) : 0L;
    int Idum = -1;

    printf(" Monte-Carlo simulation of Non-Binary LDPC decoder \n\n");
    printf("Simulation parameters:\n");
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
    LoadTables(&table, code.GF, code.logGF);
    GaussianElimination(&code, &table);

    /* flat description for the batched GPU decoder */
    int *col = malloc(sizeof(int) * (size_t)code.nbBranch), *coef = malloc(sizeof(int) * (size_t)code.nbBranch);
    for (int m = 0, e = 0; m < code.M; m++)
        for (int k = 0; k < code.rowDegree[m]; k++, e++) {
            col[e] = code.mat[m][k];
            coef[e] = code.matValue[m][k];
        }
    ems_code_desc desc = {code.N, code.M, code.GF, code.logGF, code.nbBranch, code.rowDegree, col, coef};
    ems_params prm = {nbMax, NbOper, offset, NbIterMax, 1};
    if (getenv("EMS_DEVICES")) {                                        /* one simulation on several GPUs of the node (run_multi) */
        char note_m[40] = "FB30", file_m[128];
        printf("\n\t Note             : %s\n", note_m);
        sprintf(file_m, "./data/results_N%d_CR%0.2f_GF%d_IT%d_Offset%0.1f_nm%d_%s.txt", code.N, code.rate, code.GF, NbIterMax, offset, nbMax, note_m);
        time_t now_m = time(NULL);
        printf("Simulation started at time: %s \n", ctime(&now_m));
        ems_counters fin;
        memset(&fin, 0, sizeof fin);
        struct timespec m0, m1;
        clock_gettime(CLOCK_MONOTONIC, &m0);
        int batch_m = 0;
        run_multi(getenv("EMS_DEVICES"), &code, &table, &desc, &prm, batch, NbMonteCarlo, EbN, threads, gpu_source, gpu_seed, first_frame, &fin, &batch_m);
        clock_gettime(CLOCK_MONOTONIC, &m1);
        const double tm = (m1.tv_sec - m0.tv_sec) + 1e-9 * (m1.tv_nsec - m0.tv_nsec);
        const int nbm = (int)fin.c[EMS_CNT_FRAMES], stopped = fin.c[EMS_CNT_ERR_FRAMES] == 40;
        report(file_m, EbN, (int)fin.c[EMS_CNT_ERR_FRAMES], stopped ? nbm : nbm + 1, (int)fin.c[EMS_CNT_BIT_ERRORS], fin.c[EMS_CNT_SUM_IT], code.K, code.logGF);
        fprintf(stderr, "[ems_sim] %d frames on EMS_DEVICES=%s, wall %.3f s incl. set-up (%.1f frames/s), batch %d per device\n", nbm, getenv("EMS_DEVICES"),
                tm, tm > 0 ? nbm / tm : 0.0, batch_m);
        free(col); free(coef);
        FreeCode(&code);
        FreeTable(&table);
        return EXIT_SUCCESS;
    }
    /* NH decoder handles (own streams) used round-robin: up to NH batches are in flight, the results of batch k
     * are fetched only after batch k+NH-1 has been launched, so the tail of a batch (its few frames that run every
     * iteration) overlaps the next ones.  EMS_INFLIGHT=1..4 (default 3; 2.0 dB, GF(256): 5.5 / 6.7 / 7.4 k frames/s). */
    enum { NH_MAX = 4 };
    int NH = getenv("EMS_INFLIGHT") ? atoi(getenv("EMS_INFLIGHT")) : 3;
    if (NH < 1) NH = 1;
    if (NH > NH_MAX) NH = NH_MAX;
    ems_handle *hd[NH_MAX] = {NULL, NULL, NULL, NULL};
    int rc;
    const int batch_given = batch > 0;
    /* default batch: the frames that fill this device with the kernel the library picks for this code (e.g. 6 per CU for the
     * GF(256) dc = 4 codes), within the free HBM for NH handles */
    if (!batch_given && (rc = ems_suggest_batch(&desc, &prm, device, NH, &batch))) die("ems_suggest_batch", rc);
    if (batch > NbMonteCarlo) batch = NbMonteCarlo;            /* (no buffers for frames that are never drawn) */
    if (batch < 1) batch = 1;
    fprintf(stderr, "[ems_sim] batch %d frames per launch (%s), %d handle(s) in flight, device %d\n", batch,
            batch_given ? "EMS_BATCH" : "derived from the device", NH, device);
    for (int i = 0; i < NH; i++)
        if ((rc = ems_create(&hd[i], &desc, &prm, batch, device))) die("ems_create", rc);
    if (gpu_source)
        for (int i = 0; i < NH; i++)
            if ((rc = ems_source_attach(hd[i], code.matUT[0], code.Perm))) die("ems_source_attach", rc);
    const float sigma_dev = (float)sqrt(1.0 / (2.0 * (double)code.rate * pow(10.0, (double)EbN / 10.0)));   /* channel.c:51 */

    char note[40] = "FB30";
    char file_name[128];
    printf("\n\t Note             : %s\n", note);
    sprintf(file_name, "./data/results_N%d_CR%0.2f_GF%d_IT%d_Offset%0.1f_nm%d_%s.txt", code.N, code.rate, code.GF,
            NbIterMax, offset, nbMax, note);
    time_t now = time(NULL);
    printf("Simulation started at time: %s \n", ctime(&now));

    const int N = code.N, K = code.K, logGF = code.logGF;
    int **NBIN = calloc(N, sizeof(int *)), **KBIN = calloc(K, sizeof(int *));
    for (int n = 0; n < N; n++) NBIN[n] = calloc(logGF, sizeof(int));
    for (int k = 0; k < K; k++) KBIN[k] = calloc(logGF, sizeof(int));
    int *KSYMB = calloc(K, sizeof(int));
    enum { NBUF_MAX = NH_MAX + 1 };
    const int NBUF = NH + 1;                                            /* NH in flight or being tallied + one being drawn */
    int *cwbuf[NBUF_MAX];
    float *nybuf[NBUF_MAX];
    for (int i = 0; i < NBUF; i++) {
        cwbuf[i] = malloc(sizeof(int) * (size_t)batch * N);
        nybuf[i] = malloc(sizeof(float) * (size_t)batch * N * logGF);
    }
    int *iters = malloc(sizeof(int) * batch), *synd = malloc(sizeof(int) * batch), *nberr = malloc(sizeof(int) * batch);

    int nb = 0, sum_it = 0, total_errors = 0, nbErroneousFrames = 0, nbUndetectedErrors = 0, stop = 0;
    double t_decode = 0, t_gen = 0;
    gen_job job[NBUF_MAX];
    for (int i = 0; i < NBUF; i++)
        job[i] = (gen_job){&code, &table, 0, 0, threads, EbN, 0.0f, cwbuf[i], nybuf[i], NBIN, KBIN, KSYMB, &Idum, 0.0, 0, first_frame != 0};
    pthread_t producer;
    int drawn = 0;                                                      /* frames handed to the generator so far */
    job[0].first = first_frame;
    job[0].B = NbMonteCarlo < batch ? NbMonteCarlo : batch;
    drawn = job[0].B;
    if (!gpu_source) generate(&job[0]);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int launched = 0, tallied = 0;                                      /* batch counters; batch k uses job[k % NBUF], hd[k % NH] */
    for (;;) {
        int have_next = 0, more = 0;
        if ((long)launched * batch < NbMonteCarlo) {
            gen_job *j = &job[launched % NBUF];
            if (gpu_source) {
                j->first = first_frame + (long)launched * batch;
                j->B = (NbMonteCarlo - launched * batch) < batch ? NbMonteCarlo - launched * batch : batch;
            }
            if (j->rc) die("ems_draw_frames", j->rc);
            t_gen += j->seconds;
            if (!gpu_source && drawn < NbMonteCarlo) {                  /* draw the next batch while this one decodes */
                gen_job *n = &job[(launched + 1) % NBUF];
                n->first = first_frame + drawn;
                n->B = (NbMonteCarlo - drawn) < batch ? (NbMonteCarlo - drawn) : batch;
                drawn += n->B;
                have_next = pthread_create(&producer, NULL, generate, n) == 0;
                if (!have_next) generate(n);
            }
            ems_handle *h = hd[launched % NH];                          /* asynchronous: upload, LLR kernel, decode kernel */
            if (gpu_source == 2) {
                struct timespec g0, g1;
                clock_gettime(CLOCK_MONOTONIC, &g0);
                if ((rc = ems_source_draw_reference(h, (unsigned long long)j->first, j->B, sigma_dev, NULL))) die("ems_source_draw_reference", rc);
                clock_gettime(CLOCK_MONOTONIC, &g1);
                t_gen += (g1.tv_sec - g0.tv_sec) + 1e-9 * (g1.tv_nsec - g0.tv_nsec);   /* (it waits for its own kernels: the host patches undecided samples) */
            } else if (gpu_source) {
                if ((rc = ems_source_draw(h, gpu_seed, (unsigned long long)j->first, j->B, sigma_dev))) die("ems_source_draw", rc);
            } else if ((rc = ems_llr_init(h, j->noisy, j->sigma, j->B))) die("ems_llr_init", rc);
            if ((rc = ems_decode(h, j->B))) die("ems_decode", rc);
            launched++;
            more = 1;
        }
        if (launched - tallied == NH || (!more && launched > tallied)) { /* fetch the older batch (blocks on its stream) */
            gen_job *j = &job[tallied % NBUF];
            ems_handle *h = hd[tallied % NH];
            const int B = j->B;
            if ((rc = ems_get_results(h, B, NULL, iters, synd))) die("ems_get_results", rc);   /* decisions stay in HBM */
            if ((rc = gpu_source ? ems_count_errors_resident(h, B, nberr) : ems_count_errors(h, j->cw, B, nberr))) die("ems_count_errors", rc);
            tallied++;
            for (int b = 0; b < B; b++) {                               /* NB_LDPC.c:474-507, frame order */
                nb++;
                sum_it += iters[b];
                total_errors += nberr[b];
                if (nberr[b] != 0) {
                    nbErroneousFrames++;
                    if (synd[b] == 0) nbUndetectedErrors++;
                }
                printf("\r<%d> FER= %d / %d = %f BER= %d / x = %f  avr_it=%.2f", nbUndetectedErrors, nbErroneousFrames, nb,
                       (double)nbErroneousFrames / nb, total_errors, (double)total_errors / ((double)nb * K * logGF),
                       (double)sum_it / nb);
                fflush(stdout);
                if (nbErroneousFrames == 40) { stop = 1; break; }
            }
        }
        if (have_next) pthread_join(producer, NULL);                    /* (dropped if the 40-error stop came first) */
        if (stop || (!more && launched == tallied)) break;
    }
    for (int i = 0; i < NH; i++) ems_synchronize(hd[i]);                 /* a batch launched ahead of the stop is dropped */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    t_decode = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    /* the reference's `nb` is one past the last frame when the loop runs to completion (NB_LDPC.c:576) */
    report(file_name, EbN, nbErroneousFrames, stop ? nb : nb + 1, total_errors, sum_it, K, logGF);
    fprintf(stderr, "[ems_sim] %d frames, simulation loop wall %.3f s (%.1f frames/s incl. PCIe and generation), batch %d; %s frame generation %.3f s on %d thread(s)\n",
            nb, t_decode, t_decode > 0 ? nb / t_decode : 0.0, batch, gpu_source == 2 ? "device-side drand48 stream," : (gpu_source ? "device-side (no host)" : "host"), t_gen, gpu_source ? 0 : threads);

    for (int i = 0; i < NH; i++) ems_destroy(hd[i]);
    for (int i = 0; i < NBUF; i++) { free(cwbuf[i]); free(nybuf[i]); }
    free(col); free(coef); free(iters); free(synd); free(nberr); free(KSYMB);
    for (int n = 0; n < N; n++) free(NBIN[n]);
    for (int k = 0; k < K; k++) free(KBIN[k]);
    free(NBIN); free(KBIN);
    FreeCode(&code);
    FreeTable(&table);
    return EXIT_SUCCESS;
}
