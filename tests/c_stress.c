/* Thread-safety of the request queue under a MIXED load: T threads submit R different requests each (sizes, crop,
 * blur, grayscale, JPEG encode, orientation all vary) through flgpu_transform; every result is hashed.  Then the same
 * requests run again one at a time.  The two hash tables must be identical: batching, lane scheduling and scratch
 * reuse must never leak between requests.      c_stress <threads> <requests per thread>                                */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fanlin_gpu.h"

static flgpu_ctx *g_ctx;
static int g_threads, g_per;
static uint64_t *g_hash;
static int g_failed;

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static uint64_t run_request(int id)
{
    uint32_t s = 0x5EED0000u + (uint32_t)id;
    const uint32_t sw = 64 + lcg(&s) % 900, sh = 48 + lcg(&s) % 600, c = 1 + lcg(&s) % 4;
    flgpu_params p;
    memset(&p, 0, sizeof(p));
    p.has_dims = lcg(&s) % 8 != 0;
    p.w = 20 + lcg(&s) % 400; p.h = 20 + lcg(&s) % 300;
    p.fill_r = (uint8_t)lcg(&s); p.fill_g = (uint8_t)lcg(&s); p.fill_b = (uint8_t)lcg(&s);
    p.crop = lcg(&s) % 2; p.grayscale = lcg(&s) % 4 == 0; p.inverse = lcg(&s) % 4 == 0;
    p.blur_sigma = lcg(&s) % 5 == 0 ? 10.0f + (float)(lcg(&s) % 11) : 0.0f;
    p.quality = (uint8_t)(1 + lcg(&s) % 100);
    p.front_end = lcg(&s) % 3 == 0 ? FLGPU_FE_JPEG : (lcg(&s) % 5 == 0 ? FLGPU_FE_JFIF444 : FLGPU_FE_NONE);
    p.orientation = (uint8_t)(1 + lcg(&s) % 8);
    flgpu_plan plan;
    if (flgpu_plan_output(&p, sw, sh, c, &plan) != FLGPU_OK) return 1;
    const size_t n = (size_t)sw * sh * c, cap = plan.out_bytes + (size_t)plan.plane_w * plan.plane_h * 8 + 4096;
    /* every third request uses page-locked buffers (FLGPU_IMG_PINNED: no staging copy inside the library) */
    const int pinned = id % 3 == 0;
    uint8_t *src = (uint8_t *)(pinned ? flgpu_host_alloc(g_ctx, n) : malloc(n)), *dst = (uint8_t *)(pinned ? flgpu_host_alloc(g_ctx, cap) : malloc(cap));
    for (size_t i = 0; i < n; ++i) src[i] = (uint8_t)(lcg(&s) >> 3);
    const uint32_t fl = pinned ? FLGPU_IMG_PINNED : 0u;
    flgpu_image in = {src, n, sw, sh, c, fl, 0}, out = {dst, cap, 0, 0, 0, fl, 0};
    uint64_t h = 0xcbf29ce484222325ull;
    const int st = flgpu_transform(g_ctx, &in, &p, &out);
    if (st != FLGPU_OK) {
        if (__sync_fetch_and_add(&g_failed, 1) < 5) fprintf(stderr, "request %d failed: %s (%s)\n", id, flgpu_strerror(st), flgpu_last_error(g_ctx)); /* which of the two runs, and why, if the hashes differ below */
        h = 0xDEAD0000u + (uint64_t)st;
    }
    else {
        for (uint64_t i = 0; i < out.bytes; ++i) h = (h ^ dst[i]) * 0x100000001b3ull;
        h ^= ((uint64_t)out.width << 40) ^ ((uint64_t)out.height << 20) ^ out.channels ^ (out.bytes << 3);
    }
    if (pinned) { flgpu_host_free(g_ctx, src); flgpu_host_free(g_ctx, dst); } else { free(src); free(dst); }
    return h;
}

static void *worker(void *arg)
{
    const int t = (int)(intptr_t)arg;
    for (int r = 0; r < g_per; ++r) g_hash[t * g_per + r] = run_request(t * g_per + r);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc != 3) return 2;
    g_threads = atoi(argv[1]); g_per = atoi(argv[2]);
    const int total = g_threads * g_per;
    flgpu_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = -1;
    int st = 0;
    g_ctx = flgpu_create(&cfg, &st);
    if (!g_ctx) { fprintf(stderr, "flgpu_create: %s\n", flgpu_strerror(st)); return 3; }
    g_hash = (uint64_t *)calloc((size_t)total, sizeof(uint64_t));
    pthread_t *ts = (pthread_t *)calloc((size_t)g_threads, sizeof(pthread_t));
    for (int t = 0; t < g_threads; ++t) pthread_create(&ts[t], NULL, worker, (void *)(intptr_t)t);
    for (int t = 0; t < g_threads; ++t) pthread_join(ts[t], NULL);
    int bad = 0;
    for (int id = 0; id < total; ++id) {
        const uint64_t h = run_request(id); /* alone: a batch of one */
        if (h != g_hash[id]) { if (bad < 5) fprintf(stderr, "request %d: concurrent %016llx alone %016llx\n", id, (unsigned long long)g_hash[id], (unsigned long long)h); bad++; }
    }
    flgpu_stats stats;
    flgpu_get_stats(g_ctx, &stats);
    printf("{\"requests\": %d, \"mismatches\": %d, \"failed\": %d, \"queue_flushes\": %llu, \"images\": %llu}\n", total, bad, g_failed,
           (unsigned long long)stats.queue_flushes, (unsigned long long)stats.images);
    flgpu_destroy(g_ctx);
    return bad ? 4 : 0;
}
