/* test_tally.c -- CPU unit test of the multi-GPU Monte-Carlo loop of ems_sim (csrc/harness/ems_multi.h + ems_tally.h) with a FAKE
 * N-device decode (per-frame results are a fixed function of the frame index) and a barrier-based all-reduce standing in for RCCL:
 * for any device count and batch size the ordered tally must be the single-device one -- same line after every frame, truncated
 * at exactly the frame that brings the erroneous frames to 40 (NB_LDPC.c:491-506) -- and every device must see the same
 * all-reduced counters.  Usage: test_tally <ndev> <batch> <n_frames> <error rate per 1000>; prints the trace; exit code 0 = equal
 * to the sequential tally.
 * Failure injection: three more arguments <device> <launch|collect> <k> make that device's k-th call (0-based) of that callback fail.
 * The loop must then END on every thread -- nobody left in a barrier or in the all-reduce -- and ems_multi_run must return -1:
 * exit code 4.  (A hang shows as the caller's timeout.)
 * EMS_TALLY_LAGGED=1 in the environment: the split all-reduce of ems_multi.h (allreduce_begin / allreduce_end: issued when a round is
 * collected, waited for one round later) instead of the synchronous one.  Every callback is then logged per device and the ORDER is
 * asserted: launch(r+1) < collect(r) < [end(r-1)] < begin(r), i.e. the all-reduce of round r is never waited for before round r+1 has been
 * collected, and none is outstanding when the run ends. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ems_multi.h"

static unsigned mix(unsigned long long f)
{
    f = (f + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;
    f ^= f >> 29;
    f *= 0x94D049BB133111EBull;
    return (unsigned)(f >> 33);
}
static int g_rate = 30;
static int g_fail_dev = -1, g_fail_kind = 0, g_fail_at = 0;     /* kind 1: launch, 2: collect */
static int g_calls[64][3];
static ems_frame_result fake_result(long long f)
{
    const unsigned h = mix((unsigned long long)f);
    ems_frame_result r;
    const int bad = (int)(h % 1000u) < g_rate;
    r.nberr = bad ? 1 + (int)((h >> 10) % 97u) : 0;
    r.synd = bad ? (((h >> 20) & 7u) ? 1 + (int)((h >> 23) % 63u) : 0) : 0;      /* one bad frame in eight is undetected */
    r.iters = bad ? 30 : 2 + (int)((h >> 5) % 9u);
    return r;
}

typedef struct {
    int ndev;
    long long first[64][2];
    int count[64][2];
    pthread_barrier_t bar;
    long long slots[64][EMS_NCOUNTERS];
    long long seen_total[64][EMS_NCOUNTERS];      /* what each device accumulated from the all-reduce */
    char *trace; size_t len, cap;
    int launches, drains;
    pthread_mutex_t mu;
    long long pend[64][EMS_NCOUNTERS];            /* split form: what begin() left for end() */
    int outstanding[64];
    char *ev[64]; size_t evn[64], evcap[64];     /* per-device event log: 'L' launch, 'C' collect, 'D' drain, 'B' begin, 'E' end */
} fake_ctx;

static void logev(fake_ctx *x, int dev, char e)
{
    if (x->evn[dev] + 1 > x->evcap[dev]) { x->evcap[dev] = 2 * x->evcap[dev] + 256; x->ev[dev] = (char *)realloc(x->ev[dev], x->evcap[dev]); }
    x->ev[dev][x->evn[dev]++] = e;
}

static int f_launch(void *c, int dev, int slot, long long first, int count)
{
    fake_ctx *x = (fake_ctx *)c;
    if (dev == g_fail_dev && g_fail_kind == 1 && g_calls[dev][1]++ == g_fail_at) return -1;
    x->first[dev][slot] = first; x->count[dev][slot] = count;
    logev(x, dev, 'L');
    pthread_mutex_lock(&x->mu); x->launches++; pthread_mutex_unlock(&x->mu);
    return 0;
}
static int f_collect(void *c, int dev, int slot, int count, ems_frame_result *res)
{
    fake_ctx *x = (fake_ctx *)c;
    if (count != x->count[dev][slot]) return -1;
    if (dev == g_fail_dev && g_fail_kind == 2 && res && g_calls[dev][2]++ == g_fail_at) return -1;
    if (!res) { logev(x, dev, 'D'); pthread_mutex_lock(&x->mu); x->drains++; pthread_mutex_unlock(&x->mu); return 0; }
    logev(x, dev, 'C');
    for (int b = 0; b < count; b++) res[b] = fake_result(x->first[dev][slot] + b);
    return 0;
}
static int f_allreduce(void *c, int dev, long long v[EMS_NCOUNTERS])
{
    fake_ctx *x = (fake_ctx *)c;
    memcpy(x->slots[dev], v, sizeof x->slots[dev]);
    pthread_barrier_wait(&x->bar);
    for (int i = 0; i < EMS_NCOUNTERS; i++) {
        long long s = 0;
        for (int d = 0; d < x->ndev; d++) s += x->slots[d][i];
        v[i] = s;
        x->seen_total[dev][i] += s;
    }
    pthread_barrier_wait(&x->bar);
    return 0;
}
/* split form: begin only deposits the contribution (nothing may wait in it), end meets the other devices */
static int f_begin(void *c, int dev, const long long v[EMS_NCOUNTERS])
{
    fake_ctx *x = (fake_ctx *)c;
    if (x->outstanding[dev]) return -1;                         /* at most one outstanding per device */
    memcpy(x->pend[dev], v, sizeof x->pend[dev]);
    x->outstanding[dev] = 1;
    logev(x, dev, 'B');
    return 0;
}
static int f_end(void *c, int dev, long long v[EMS_NCOUNTERS])
{
    fake_ctx *x = (fake_ctx *)c;
    if (!x->outstanding[dev]) return -1;
    logev(x, dev, 'E');
    memcpy(v, x->pend[dev], sizeof x->pend[dev]);
    x->outstanding[dev] = 0;
    return f_allreduce(c, dev, v);
}
/* the order ems_multi.h promises for the split form, on one device's log: (L) then per round [L] C [E] B, drains and the last E at the end */
static int order_ok(const char *ev, size_t n)
{
    size_t i = 0;
    int round = 0;
    if (n == 0) return 1;
    if (ev[i] == 'L') i++;                                      /* round 0 goes out */
    while (i < n) {
        if (ev[i] == 'L') i++;                                  /* launch(r+1) BEFORE collect(r) */
        if (i < n && ev[i] == 'C') i++;                         /* (a device whose stream has run out has nothing to collect) */
        if (round > 0) { if (i >= n || ev[i] != 'E') return 0; i++; }   /* end(r-1) only now: round r has been collected */
        if (i >= n || ev[i] != 'B') return 0;
        i++;
        round++;
        if (i < n && ev[i] == 'D') i++;                         /* dropped round at the stop */
        if (i + 1 == n && ev[i] == 'E') return 1;               /* the last all-reduce is waited for when the loop is left */
    }
    return 0;
}
static void f_line(const ems_counters *run, void *c)
{
    fake_ctx *x = (fake_ctx *)c;
    char buf[160];
    const int n = snprintf(buf, sizeof buf, "<%lld> FER= %lld / %lld BER= %lld avr_it=%.2f\n", run->c[EMS_CNT_UNDETECTED], run->c[EMS_CNT_ERR_FRAMES],
                           run->c[EMS_CNT_FRAMES], run->c[EMS_CNT_BIT_ERRORS], (double)run->c[EMS_CNT_SUM_IT] / (double)run->c[EMS_CNT_FRAMES]);
    if (x->len + (size_t)n + 1 > x->cap) { x->cap = 2 * x->cap + 4096; x->trace = (char *)realloc(x->trace, x->cap); }
    memcpy(x->trace + x->len, buf, (size_t)n + 1);
    x->len += (size_t)n;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s ndev batch n_frames rate_per_1000\n", argv[0]); return 2; }
    const int ndev = atoi(argv[1]), batch = atoi(argv[2]);
    const long long n_frames = atoll(argv[3]);
    g_rate = atoi(argv[4]);
    if (ndev < 1 || ndev > 64) return 2;
    if (argc >= 8) {
        g_fail_dev = atoi(argv[5]);
        g_fail_kind = strcmp(argv[6], "launch") == 0 ? 1 : 2;
        g_fail_at = atoi(argv[7]);
    }
    fake_ctx x;
    memset(&x, 0, sizeof x);
    x.ndev = ndev;
    pthread_barrier_init(&x.bar, NULL, (unsigned)ndev);
    pthread_mutex_init(&x.mu, NULL);
    const int lagged = getenv("EMS_TALLY_LAGGED") != NULL;
    ems_multi_ops ops = {f_launch, f_collect, lagged ? NULL : f_allreduce, f_line, &x, lagged ? f_begin : NULL, lagged ? f_end : NULL};
    ems_counters fin;
    if (ems_multi_run(&ops, ndev, n_frames, batch, 40, &fin)) {
        fprintf(stderr, "ems_multi_run failed%s\n", g_fail_dev >= 0 ? " (as injected): every device thread came back" : "");
        free(x.trace);
        for (int d = 0; d < ndev; d++) free(x.ev[d]);
        return g_fail_dev >= 0 ? 4 : 3;
    }

    /* the single-device loop of the reference, frame after frame */
    fake_ctx y;
    memset(&y, 0, sizeof y);
    ems_counters seq;
    memset(&seq, 0, sizeof seq);
    for (long long f = 0; f < n_frames; f++) {
        const ems_frame_result r = fake_result(f), *pr = &r;
        const int one = 1;
        if (ems_tally_round(&seq, &pr, &one, 1, 40, f_line, &y)) break;
    }
    fputs(x.trace ? x.trace : "", stdout);
    int ok = x.len == y.len && (x.len == 0 || memcmp(x.trace, y.trace, x.len) == 0) && memcmp(&fin, &seq, sizeof fin) == 0;
    for (int d = 1; d < ndev; d++) ok = ok && memcmp(x.seen_total[d], x.seen_total[0], sizeof x.seen_total[0]) == 0;
    /* the all-reduced totals cover whole rounds: at least the ordered tally, and equal to it when the run was not truncated */
    ok = ok && x.seen_total[0][EMS_CNT_FRAMES] >= fin.c[EMS_CNT_FRAMES] && x.seen_total[0][EMS_CNT_ERR_FRAMES] >= fin.c[EMS_CNT_ERR_FRAMES];
    if (fin.c[EMS_CNT_ERR_FRAMES] < 40) ok = ok && memcmp(x.seen_total[0], fin.c, sizeof fin.c) == 0 && fin.c[EMS_CNT_FRAMES] == n_frames;
    if (lagged)
        for (int d = 0; d < ndev; d++) {
            if (!order_ok(x.ev[d], x.evn[d]) || x.outstanding[d]) {
                fprintf(stderr, "device %d: callback order %.*s breaks launch(r+1) < collect(r) < end(r-1) < begin(r)\n", d, (int)x.evn[d], x.ev[d]);
                ok = 0;
            }
        }
    for (int d = 0; d < ndev; d++) free(x.ev[d]);
    fprintf(stderr, "ndev %d batch %d: %lld frames tallied, %lld erroneous, %d launches, %d dropped -> %s\n", ndev, batch, fin.c[EMS_CNT_FRAMES],
            fin.c[EMS_CNT_ERR_FRAMES], x.launches, x.drains, ok ? "EQUAL to the sequential tally" : "MISMATCH");
    free(x.trace); free(y.trace);
    return ok ? 0 : 1;
}
