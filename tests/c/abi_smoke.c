/* A host program in plain C against include/pyvb_hip.h: no Python, no C++ in sight.  Builds with
 *   gcc -std=c99 -I include tests/c/abi_smoke.c -o abi_smoke -L pyvb_amd -lpyvb_hip -lm -Wl,-rpath,$PWD/pyvb_amd
 * and runs two variational iterations of a small LDS batch on device 0; prints the lower bound of both iterations (which
 * tests/test_c_abi_gpu.py compares with the Python front end on the same inputs, read from a file the test writes).
 * File format (doubles, native endian): header N T D K, then Y[N][T][K], X[N][T][D], A_mean, A_colvar, C_mean, C_colvar,
 * Q_b, R_b in the layouts of the header. */
#include <stdio.h>
#include <stdlib.h>
#include "pyvb_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != PYVB_OK) { fprintf(stderr, "%s failed: %d %s\n", #call, rc_, pyvb_last_error()); return 1; } } while (0)

static double* rd(FILE* f, size_t n) {
    double* p = (double*)malloc(n * sizeof(double));
    if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: abi_smoke problem.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    double* hdr = rd(f, 4);
    const int N = (int)hdr[0], T = (int)hdr[1], D = (int)hdr[2], K = (int)hdr[3];
    double* Y = rd(f, (size_t)N * T * K);
    double* X = rd(f, (size_t)N * T * D);
    double* A_mean = rd(f, (size_t)N * D * D); double* A_var = rd(f, (size_t)N * D * D);
    double* C_mean = rd(f, (size_t)N * K * D); double* C_var = rd(f, (size_t)N * D * K);
    double* Q_b = rd(f, (size_t)N * D); double* R_b = rd(f, (size_t)N * K);
    fclose(f);

    /* priors of examples/Linear_Dynamic_System.py:47-58 */
    double* x0_mean = (double*)calloc(D, sizeof(double));
    double* x0_prec = (double*)calloc((size_t)D * D, sizeof(double));
    double* A_pm = (double*)calloc((size_t)D * D, sizeof(double)); double* A_pp = (double*)malloc((size_t)D * D * sizeof(double));
    double* C_pm = (double*)calloc((size_t)K * D, sizeof(double)); double* C_pp = (double*)malloc((size_t)D * K * sizeof(double));
    double* qa0 = (double*)malloc(D * sizeof(double)); double* ra0 = (double*)malloc(K * sizeof(double));
    for (int i = 0; i < D; ++i) { x0_prec[i * D + i] = 1.0; qa0[i] = 1e-3; }
    for (int i = 0; i < D * D; ++i) A_pp[i] = 1e-3;
    for (int i = 0; i < D * K; ++i) C_pp[i] = 1e-3;
    for (int i = 0; i < K; ++i) ra0[i] = 1e-3;

    int ndev = 0;
    CHECK(pyvb_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 3; }
    pyvb_lds* h = NULL;
    CHECK(pyvb_lds_create(&h, 0, N, T, D, K, PYVB_NOISE_DIAGONAL_GAMMA));
    CHECK(pyvb_lds_set_priors(h, x0_mean, x0_prec, A_pm, A_pp, C_pm, C_pp, qa0, qa0, ra0, ra0));
    CHECK(pyvb_lds_set_observations(h, Y));
    CHECK(pyvb_lds_set_state(h, X, A_mean, A_var, C_mean, C_var, Q_b, R_b));
    CHECK(pyvb_lds_iterate(h, 2));
    double hist[2 * 6];
    int count = 0;
    CHECK(pyvb_lds_get_elbo_history(h, hist, 2, &count));
    if (count != 2) { fprintf(stderr, "history holds %d rows\n", count); return 4; }
    for (int it = 0; it < 2; ++it) {
        double tot = 0.0;
        for (int p = 0; p < 6; ++p) tot += hist[it * 6 + p];
        printf("iteration %d lower bound %.17g\n", it + 1, tot);
    }
    CHECK(pyvb_lds_get_state(h, X, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL));
    printf("x[0][T-1][0] %.17g\n", X[(size_t)(T - 1) * D]);
    CHECK(pyvb_lds_destroy(h));
    return 0;
}
