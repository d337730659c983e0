/* A host with no Python and no PyTorch in the process: plain C against include/fanlin_gpu.h, the way the Rust shim
 * of INTEGRATION.md binds it.  Used by tests/test_c_client.py (GPU box): parses a query string, transforms one
 * synthetic picture through flgpu_transform and writes the result for comparison with the oracle.
 *   c_client <query> <w> <h> <channels> <out file>                                                           */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fanlin_gpu.h"

int main(int argc, char **argv)
{
    if (argc != 6) return 2;
    const uint32_t w = (uint32_t)atoi(argv[2]), h = (uint32_t)atoi(argv[3]), c = (uint32_t)atoi(argv[4]);
    flgpu_query q;
    if (flgpu_query_parse(argv[1], &q) != FLGPU_OK) { fprintf(stderr, "query rejected\n"); return 3; }
    if (flgpu_query_as_is(&q)) { fprintf(stderr, "as_is: nothing to do\n"); return 4; }
    flgpu_params p;
    int out_format = 0;
    if (flgpu_params_from_query(&q, 0, 0, &p, &out_format) != FLGPU_OK) return 5;
    p.front_end = FLGPU_FE_NONE;
    flgpu_plan plan;
    if (flgpu_plan_output(&p, w, h, c, &plan) != FLGPU_OK) return 6;
    const size_t n = (size_t)w * h * c;
    uint8_t *src = (uint8_t *)malloc(n), *dst = (uint8_t *)malloc(plan.out_bytes);
    uint32_t s = 0xFA171200u;                       /* the test regenerates the same bytes */
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; src[i] = (uint8_t)(s >> 24); }
    flgpu_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.device = -1;
    int st = 0;
    flgpu_ctx *ctx = flgpu_create(&cfg, &st);
    if (!ctx) { fprintf(stderr, "flgpu_create: %s\n", flgpu_strerror(st)); return 7; }
    flgpu_image in = {src, n, w, h, c, 0, 0}, out = {dst, plan.out_bytes, 0, 0, 0, 0, 0};
    st = flgpu_transform(ctx, &in, &p, &out);
    if (st != FLGPU_OK) { fprintf(stderr, "flgpu_transform: %s (%s)\n", flgpu_strerror(st), flgpu_last_error(ctx)); return 8; }
    FILE *f = fopen(argv[5], "wb");
    if (!f) return 9;
    fprintf(f, "%u %u %u %llu\n", out.width, out.height, out.channels, (unsigned long long)out.bytes);
    fwrite(dst, 1, out.bytes, f);
    fclose(f);
    flgpu_destroy(ctx);
    free(src); free(dst);
    return 0;
}
