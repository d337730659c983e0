/* Per-image latency of the drop-in entry point, measured without Python in the way: T caller threads (the
 * reference's tokio workers, src/main.rs:33) call flgpu_transform concurrently with HOST buffers; the library's
 * request queue packs them into shared launches.  Prints one JSON object.
 *   latency_probe <threads> <requests> <src_w> <src_h> <query> [front_end] [queue_lanes] [max_batch] [pinned] [jpeg files...]
 * With JPEG files (each src_w x src_h) the requests carry the FILE bytes (FLGPU_IMG_JPEG_SOURCE): the library decodes them. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <sys/resource.h>

#include "fanlin_gpu.h"

static flgpu_ctx *g_ctx;
static flgpu_params g_params;
static flgpu_plan g_plan;
static uint32_t g_w, g_h;
static uint8_t *g_src[8];
static size_t g_src_len[8];
static int g_jpeg, g_nsrc = 8;
static uint32_t g_c = 3;
static int g_requests, g_next, g_failed, g_pinned;
static double *g_lat;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static uint8_t **g_dst; /* one result buffer per caller thread, allocated BEFORE the clock starts: round 2 allocated (and
                           freed) them inside the timed region, and 64 hipHostMalloc / hipHostFree calls behind the driver's
                           lock were what made the page-locked run look slower than the pageable one */

/* The caller threads live through both phases, as a server's worker threads do: until round 5 the timed phase started 64 NEW threads, and the
   first hundred-odd requests of every run (each thread's first call: its thread-local decode arrays, its first pages) took 20-30 ms -- the
   whole p99 of the run was that start (tools/experiments/jh_tail.sh). */
static pthread_barrier_t g_phase;

static void *caller(void *arg)
{
    uint8_t *dst = g_dst[(size_t)(uintptr_t)arg];
    const uint32_t fl = g_pinned ? FLGPU_IMG_PINNED : 0u;
    for (int phase = 0; phase < 2; ++phase) {
    if (phase == 1) { pthread_barrier_wait(&g_phase); /* warm-up done; main resets the counters */ pthread_barrier_wait(&g_phase); }
    for (;;) {
        pthread_mutex_lock(&g_mu);
        const int i = g_next++;
        pthread_mutex_unlock(&g_mu);
        if (i >= g_requests) break;
        flgpu_image in = {g_src[i % g_nsrc], g_jpeg ? (uint64_t)g_src_len[i % g_nsrc] : (uint64_t)g_w * g_h * 3, g_w, g_h, g_c, g_jpeg ? FLGPU_IMG_JPEG_SOURCE : fl, 0}, out = {dst, g_plan.max_out_bytes, 0, 0, 0, fl, 0};
        const double t0 = now_ms();
        if (flgpu_transform(g_ctx, &in, &g_params, &out) != FLGPU_OK) { pthread_mutex_lock(&g_mu); g_failed++; pthread_mutex_unlock(&g_mu); }
        g_lat[i] = now_ms() - t0;
    }
    }
    return NULL;
}

static int cmp(const void *a, const void *b) { const double x = *(const double *)a, y = *(const double *)b; return (x > y) - (x < y); }

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const int threads = atoi(argv[1]);
    g_requests = atoi(argv[2]);
    g_w = (uint32_t)atoi(argv[3]); g_h = (uint32_t)atoi(argv[4]);
    flgpu_query q;
    int fmt = 0;
    if (flgpu_query_parse(argv[5], &q) != FLGPU_OK || flgpu_params_from_query(&q, 0, 0, &g_params, &fmt) != FLGPU_OK) return 3;
    g_params.front_end = argc > 6 ? (uint8_t)atoi(argv[6]) : FLGPU_FE_NONE;
    if (flgpu_plan_output(&g_params, g_w, g_h, 3, &g_plan) != FLGPU_OK) return 4;
    flgpu_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = -1;
    cfg.queue_lanes = argc > 7 ? (uint32_t)atoi(argv[7]) : 0;
    cfg.max_batch = argc > 8 ? (uint32_t)atoi(argv[8]) : 0;
    if (getenv("FL_PROBE_FLUSH_US")) cfg.flush_timeout_us = (uint32_t)atoi(getenv("FL_PROBE_FLUSH_US")); /* the queue's flush timer (experiments) */
    g_pinned = argc > 9 ? atoi(argv[9]) : 0;
    int st = 0;
    g_ctx = flgpu_create(&cfg, &st);
    if (!g_ctx) { fprintf(stderr, "flgpu_create: %s\n", flgpu_strerror(st)); return 5; }
    const size_t n = (size_t)g_w * g_h * 3;
    uint32_t s = 0xFA171200u;
    if (argc > 10) {
        g_jpeg = 1;
        g_nsrc = argc - 10 > 8 ? 8 : argc - 10;
        for (int k = 0; k < g_nsrc; ++k) {
            FILE *f = fopen(argv[10 + k], "rb");
            if (!f) { fprintf(stderr, "cannot open %s\n", argv[10 + k]); return 7; }
            fseek(f, 0, SEEK_END);
            g_src_len[k] = (size_t)ftell(f);
            fseek(f, 0, SEEK_SET);
            g_src[k] = (uint8_t *)malloc(g_src_len[k]);
            if (fread(g_src[k], 1, g_src_len[k], f) != g_src_len[k]) return 7;
            fclose(f);
            flgpu_jpeg_info info;
            if (flgpu_jpeg_info_of(g_src[k], g_src_len[k], &info) != FLGPU_OK || !info.supported || info.width != g_w || info.height != g_h) {
                fprintf(stderr, "%s is not a supported %ux%u JPEG\n", argv[10 + k], g_w, g_h);
                return 7;
            }
            g_c = info.channels;
        }
        if (flgpu_plan_output(&g_params, g_w, g_h, g_c, &g_plan) != FLGPU_OK) return 4;
    }
    for (int k = 0; k < 8 && !g_jpeg; ++k) {
        g_src[k] = (uint8_t *)(g_pinned ? flgpu_host_alloc(g_ctx, n) : malloc(n));
        for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; g_src[k][i] = (uint8_t)(s >> 24); }
    }
    g_lat = (double *)calloc((size_t)(g_requests > threads * 16 ? g_requests : threads * 16), sizeof(double));
    pthread_t *ts = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    g_dst = (uint8_t **)calloc((size_t)threads, sizeof(uint8_t *));
    for (int t = 0; t < threads; ++t) g_dst[t] = (uint8_t *)(g_pinned ? flgpu_host_alloc(g_ctx, g_plan.max_out_bytes) : malloc(g_plan.max_out_bytes));
    /* warm up: tables, pinned blocks, lanes */
    const int total = g_requests;
    g_requests = threads * 16; g_next = 0; /* until the staging pools have every block they will use: a page-locked allocation takes 80 ms and holds up every copy */
    pthread_barrier_init(&g_phase, NULL, (unsigned)threads + 1u);
    for (int t = 0; t < threads; ++t) pthread_create(&ts[t], NULL, caller, (void *)(uintptr_t)t);
    pthread_barrier_wait(&g_phase); /* every caller has left the warm-up loop */
    g_requests = total; g_next = 0;
    flgpu_reset_stats(g_ctx);
    struct rusage ru0, ru1;
    getrusage(RUSAGE_SELF, &ru0);
    const double t0 = now_ms();
    pthread_barrier_wait(&g_phase);
    for (int t = 0; t < threads; ++t) pthread_join(ts[t], NULL);
    const double wall = now_ms() - t0;
    getrusage(RUSAGE_SELF, &ru1);
    /* CPU the whole process spent in the timed region (callers: decode + staging copies; the library's lane threads; the runtime's waits) */
    const double cpu_ms = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) * 1e3 + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-3 +
                          (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) * 1e3 + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-3;
    const double user_ms = (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) * 1e3 + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-3;
    flgpu_stats stats;
    flgpu_get_stats(g_ctx, &stats);
    if (getenv("FL_PROBE_DUMP")) { /* latencies in request order, for looking at where the slow ones are */
        FILE *f = fopen(getenv("FL_PROBE_DUMP"), "w");
        if (f) { for (int i = 0; i < g_requests; ++i) fprintf(f, "%.3f\n", g_lat[i]); fclose(f); }
    }
    qsort(g_lat, (size_t)g_requests, sizeof(double), cmp);
    printf("{\"p50_ms\": %.3f, \"p99_ms\": %.3f, \"requests\": %d, \"caller_threads\": %d, \"images_per_s\": %.1f, \"queue_flushes\": %llu, "
           "\"failed\": %d, \"pinned\": %d, \"jpeg_sources\": %llu, \"mean_file_bytes\": %.0f, \"mean_upload_bytes\": %.0f, \"entropy_decoded_on_device\": %llu, \"device_entropy_retries\": %llu, \"host_cpu_ms_per_request\": %.3f, \"host_cpu_user_ms_per_request\": %.3f, "
           "\"warmup_requests\": %d, \"path\": \"flgpu_transform from C threads that live through warm-up and timed phase, host buffers (H2D + kernels + D2H), request-batching queue\"}\n",
           g_lat[g_requests / 2], g_lat[(int)(g_requests * 0.99) < g_requests ? (int)(g_requests * 0.99) : g_requests - 1], g_requests, threads,
           g_requests / wall * 1e3, (unsigned long long)stats.queue_flushes, g_failed, g_pinned, (unsigned long long)stats.jpeg_sources,
           stats.jpeg_sources ? (double)stats.jpeg_file_bytes / (double)stats.jpeg_sources : 0.0, stats.jpeg_sources ? (double)stats.jpeg_upload_bytes / (double)stats.jpeg_sources : 0.0,
           (unsigned long long)stats.jpeg_device_huffman, (unsigned long long)stats.jpeg_device_huffman_retries, cpu_ms / g_requests, user_ms / g_requests, threads * 16);
    flgpu_destroy(g_ctx);
    return g_failed ? 6 : 0;
}
