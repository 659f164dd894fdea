/* TEST INFRASTRUCTURE ONLY -- command-line front of the CPU restatement (mfa_oracle.c).
 *   oracle_cli match <image.blob>   strings on stdin, one per line (empty line = empty string) -> 0/1 lines
 *   oracle_cli stats <image.blob>   same input -> one line of counters
 *   oracle_cli time  <image.blob>   same input -> "<n> <bytes> <seconds> <accepted>"
 */
#define _POSIX_C_SOURCE 200809L
#include "mfa_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static void* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)sz + 1);
    if (fread(p, 1, (size_t)sz, f) != (size_t)sz) { perror("read"); exit(2); }
    fclose(f); *n = (size_t)sz; return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: oracle_cli match|stats|time <image.blob>\n"); return 2; }
    size_t nb; void* blob = slurp(argv[2], &nb);
    mfa_oracle_image* img = NULL;
    int rc = mfa_oracle_image_load(blob, nb, &img);
    if (rc) { fprintf(stderr, "image load failed: %d\n", rc); return 3; }
    /* read all of stdin */
    size_t cap = 1 << 20, len = 0; char* buf = malloc(cap);
    for (;;) { if (len == cap) { cap *= 2; buf = realloc(buf, cap); } size_t r = fread(buf + len, 1, cap - len, stdin); if (!r) break; len += r; }
    size_t nstr = 0; for (size_t k = 0; k < len; k++) nstr += buf[k] == '\n';
    if (len && buf[len - 1] != '\n') nstr++;
    uint64_t* off = malloc((nstr + 1) * sizeof *off);
    uint8_t* bytes = malloc(len + 1); uint8_t* res = malloc(nstr + 1);
    size_t w = 0, s = 0; off[0] = 0;
    for (size_t k = 0; k < len; k++) { if (buf[k] == '\n') off[++s] = w; else bytes[w++] = (uint8_t)buf[k]; }
    if (len && buf[len - 1] != '\n') off[++s] = w;
    mfa_oracle_stats st; memset(&st, 0, sizeof st);
    int want_stats = !strcmp(argv[1], "stats");
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    rc = mfa_oracle_match_batch(img, bytes, off, nstr, res, want_stats ? &st : NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (rc) { fprintf(stderr, "match failed: %d\n", rc); return 4; }
    if (!strcmp(argv[1], "match")) {
        for (size_t k = 0; k < nstr; k++) { putchar('0' + res[k]); putchar('\n'); }
    } else if (want_stats) {
        printf("steps %llu evaluations %llu cell_reads %llu compare_bytes %llu max_states %llu variables %llu\n",
               (unsigned long long)st.steps, (unsigned long long)st.evaluations, (unsigned long long)st.cell_reads,
               (unsigned long long)st.compare_bytes, (unsigned long long)st.max_states, (unsigned long long)st.variables);
    } else {
        unsigned acc = 0; for (size_t k = 0; k < nstr; k++) acc += res[k];
        printf("%zu %zu %.6f %u\n", nstr, w, (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec), acc);
    }
    return 0;
}
