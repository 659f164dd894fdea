/* A plain C11 caller of the C-ABI (include/mfa_hip.h): what a maintainer of the reference's match loop
 * (matchers/match.cpp:21-31: compile once, then one match() per input string) links against when the loop moves to the GPU.
 *
 *   capi_smoke <image.blob> <strings.txt> <expected.bits>
 *
 * image.blob     an automaton image (include/mfa_image_format.h), e.g. the output of `diploma -dump` frozen by the host mirror
 * strings.txt    one input string per line (an empty line is the empty string)
 * expected.bits  the reference's answers, one '0' / '1' character per string (tests/golden/results)
 *
 * Loads the blob, matches the whole file with ONE mfa_match_batch_host call and compares; exit code 0 = every answer equal,
 * 1 = a difference, 2 = usage / I/O, 3 = the library returned an error (printed with mfa_strerror).  Built with
 * `cc -std=c11 -Wall -Wextra -Werror -pedantic` by tests/test_capi_from_c.py: the header is C, not C++.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mfa_hip.h"

static unsigned char* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return NULL; }
    size_t cap = 1u << 16, len = 0;
    unsigned char* buf = malloc(cap);
    if (!buf) { fclose(f); return NULL; }
    for (;;) {
        if (len == cap) {
            unsigned char* nb = realloc(buf, cap *= 2);
            if (!nb) { free(buf); fclose(f); return NULL; }
            buf = nb;
        }
        size_t got = fread(buf + len, 1, cap - len, f);
        len += got;
        if (got == 0) break;
    }
    fclose(f);
    *n = len;
    return buf;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: capi_smoke image.blob strings.txt expected.bits\n"); return 2; }
    size_t n_blob = 0, n_text = 0, n_bits = 0;
    unsigned char* blob = slurp(argv[1], &n_blob);
    unsigned char* text = slurp(argv[2], &n_text);
    unsigned char* bits = slurp(argv[3], &n_bits);
    if (!blob || !text || !bits) return 2;

    /* the batch: all strings back to back (the newlines squeezed out in place), offsets[k] .. offsets[k+1] = string k */
    uint64_t n = 0;
    for (size_t k = 0; k < n_text; k++) n += text[k] == '\n';
    uint64_t* offsets = malloc((size_t)(n + 1) * sizeof *offsets);
    uint8_t* results = malloc((size_t)(n ? n : 1));
    uint8_t* bytes = malloc(n_text + 64);                 /* (the device copy is padded by the library; 64 spare bytes here are not required) */
    if (!offsets || !results || !bytes) return 2;
    uint64_t s = 0, w = 0;
    offsets[0] = 0;
    for (size_t k = 0; k < n_text; k++) {
        if (text[k] == '\n') offsets[++s] = w;
        else bytes[w++] = text[k];
    }

    printf("%s, %d device(s)\n", mfa_version(), mfa_device_count());
    mfa_image_t* img = NULL;
    int rc = mfa_image_create(blob, n_blob, &img);
    if (rc != MFA_OK) { fprintf(stderr, "mfa_image_create: %s\n", mfa_strerror(rc)); return 3; }
    mfa_image_info info;
    rc = mfa_image_get_info(img, &info);
    if (rc != MFA_OK) { fprintf(stderr, "mfa_image_get_info: %s\n", mfa_strerror(rc)); return 3; }
    rc = mfa_match_batch_host(img, bytes, offsets, n, results, 0);
    if (rc != MFA_OK) {
        fprintf(stderr, "mfa_match_batch_host: %s (hip error %d)\n", mfa_strerror(rc), mfa_last_hip_error());
        mfa_image_destroy(img);
        return 3;
    }
    uint64_t checked = 0, wrong = 0, accepted = 0;
    for (uint64_t k = 0; k < n; k++) {
        if (k >= n_bits || (bits[k] != '0' && bits[k] != '1')) break;
        checked++;
        accepted += results[k] == 1;
        if (results[k] != (uint8_t)(bits[k] - '0')) {
            if (wrong++ < 5) fprintf(stderr, "string %llu: got %u, the reference says %c\n", (unsigned long long)k, (unsigned)results[k], bits[k]);
        }
    }
    printf("%llu strings matched in one call, %llu compared with the reference's answers, %llu accepted, %llu differences\n",
           (unsigned long long)n, (unsigned long long)checked, (unsigned long long)accepted, (unsigned long long)wrong);
    mfa_image_destroy(img);
    free(blob); free(text); free(bits); free(offsets); free(results); free(bytes);
    return (wrong == 0 && checked == n) ? 0 : 1;
}
